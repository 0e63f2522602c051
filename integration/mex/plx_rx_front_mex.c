/* RxSamples = plx_rx_front_mex(sigx, sigy, Hopt, Hel, Elo, balanced, adcbits, r, b, shift)
 * the per-sample part of RxPdmCohQpsk.m:19-72 (receiver_cohmix + ADC + fastshift + decimate + I/Q recombination) behind
 * the unchanged .m signature; the tables come from the .m code (myfilter, fastexp, fir1), see INTEGRATION.md. */
#include "plx_mex_common.h"
#include <string.h>
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    (void)nlhs;
    if (plx_mex_release_request(nrhs, prhs)) return;
    if (nrhs != 10) mexErrMsgTxt("Ten inputs required.");
    plx_mex_once();
    plx_front_desc d;
    memset(&d, 0, sizeof(d));
    d.nfft = (int64_t)mxGetNumberOfElements(prhs[0]);
    d.dual_pol = !mxIsEmpty(prhs[1]);
    d.max_frames = 1;
    d.hopt_re = mxGetPr(prhs[2]); d.hopt_im = mxGetPi(prhs[2]);       /* a NULL imaginary plane means a real table */
    d.hel_re = mxGetPr(prhs[3]); d.hel_im = mxGetPi(prhs[3]);
    if (mxGetNumberOfElements(prhs[4]) == 1 && !mxGetPi(prhs[4])) d.elo_scalar = mxGetScalar(prhs[4]);
    else { d.elo_re = mxGetPr(prhs[4]); d.elo_im = mxGetPi(prhs[4]); }
    d.balanced = (int32_t)mxGetScalar(prhs[5]);
    d.adcbits = (int32_t)mxGetScalar(prhs[6]);
    d.decim = (int32_t)mxGetScalar(prhs[7]);
    d.ntaps = (int32_t)mxGetNumberOfElements(prhs[8]);
    d.fir = mxGetPr(prhs[8]);
    size_t ns = mxGetNumberOfElements(prhs[9]);
    int64_t sh[2] = {(int64_t)mxGetPr(prhs[9])[0], (int64_t)mxGetPr(prhs[9])[ns - 1]};
    size_t nout = (size_t)((d.nfft + d.decim - 1) / (d.decim > 0 ? d.decim : 1));
    plhs[0] = mxCreateDoubleMatrix(nout, (size_t)(1 + d.dual_pol), mxCOMPLEX);
    if (plx_rx_front(mxGetPr(prhs[0]), mxGetPi(prhs[0]), d.dual_pol ? mxGetPr(prhs[1]) : NULL, d.dual_pol ? mxGetPi(prhs[1]) : NULL,
                     &d, sh, mxGetPr(plhs[0]), mxGetPi(plhs[0]), NULL, NULL))
        mexErrMsgTxt(plx_last_error());
}

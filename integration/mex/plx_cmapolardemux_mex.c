/* [y,h1,h2,passes] = plx_cmapolardemux_mex(x, R, mu, taps, M)
 * the whole driver loop of cmapolardemux (/root/reference/DspPdmCohQpsk.m:142-192 == dsp4cohdec.m:374-425) in ONE call: the
 * .m keeps its lines :143-160 (R, mu, taps and the initial centre-tap matrix M from params.mat / params.phizero / the
 * single-polarisation ratio) and hands x [L x 2] over once; the cyclic extension (:161-165), the pass loop (:175-191, up to
 * 299 calls of the per-pass MEX for L = 1024, mu = 1/6000) and the 5e-5 test (:187) run on the device.  INTEGRATION.md shows
 * the edited function body. */
#include "plx_mex_common.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (plx_mex_release_request(nrhs, prhs)) return;
    if (nrhs != 5) mexErrMsgTxt("Five inputs required.");
    if (mxGetN(prhs[0]) != 2) mexErrMsgTxt("x must have two columns.");
    if (mxGetNumberOfElements(prhs[1]) != 2) mexErrMsgTxt("R must have two elements.");
    if (mxGetM(prhs[4]) != 2 || mxGetN(prhs[4]) != 2) mexErrMsgTxt("M must be 2 x 2.");
    plx_mex_once();
    size_t L = mxGetM(prhs[0]);
    int taps = (int)mxGetScalar(prhs[3]);
    const double *mr = mxGetPr(prhs[4]), *mi = mxGetPi(prhs[4]);
    double M[8];                                     /* row-major (re, im) from MATLAB's column-major planes */
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 2; c++) { M[2 * (2 * r + c)] = mr[r + 2 * c]; M[2 * (2 * r + c) + 1] = mi ? mi[r + 2 * c] : 0.0; }
    plhs[0] = mxCreateDoubleMatrix(L, 2, mxCOMPLEX);
    mxArray *h1 = mxCreateDoubleMatrix(taps > 0 ? (size_t)taps : 0, 2, mxCOMPLEX);
    mxArray *h2 = mxCreateDoubleMatrix(taps > 0 ? (size_t)taps : 0, 2, mxCOMPLEX);
    int32_t passes = 0;
    int rc = plx_cmapolardemux(mxGetPr(prhs[0]), mxGetPi(prhs[0]), (int64_t)L, taps, mxGetScalar(prhs[2]), mxGetPr(prhs[1]), M,
                               mxGetPr(plhs[0]), mxGetPi(plhs[0]), mxGetPr(h1), mxGetPi(h1), mxGetPr(h2), mxGetPi(h2), &passes);
    if (rc) mexErrMsgTxt(plx_last_error());          /* "Ntaps should be an ODD INTEGER." (cmaadaptivefilter.c:118-119) */
    if (nlhs > 1) plhs[1] = h1; else mxDestroyArray(h1);
    if (nlhs > 2) plhs[2] = h2; else mxDestroyArray(h2);
    if (nlhs > 3) plhs[3] = mxCreateDoubleScalar((double)passes);
}

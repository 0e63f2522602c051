/* [Y,h1,h2] = cmaadaptivefilter(xx,h1,h2,taps,mu,R,sps) with the semantics of the .m TWIN, /root/reference/cmaadaptivefilter.m:1,52-72
 * (what MATLAB runs when comp_mex.m was never run): every sample updates the taps whatever sps is, there is no odd-taps
 * check, and the UPDATED taps come back in plhs[1..2] -- the inputs are left untouched, so the drivers take their
 * any(any(h1_new)) branch (DspPdmCohQpsk.m:183-186).  size(h1,1) is the number of taps, as in the .m (:53). */
#include <string.h>
#include "plx_mex_common.h"
static mxArray *complex_copy(const mxArray *a)
{
    size_t m = mxGetM(a), n = mxGetN(a);
    mxArray *c = mxCreateDoubleMatrix(m, n, mxCOMPLEX);
    memcpy(mxGetPr(c), mxGetPr(a), m * n * sizeof(double));
    if (mxGetPi(a)) memcpy(mxGetPi(c), mxGetPi(a), m * n * sizeof(double));
    return c;
}
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    (void)nlhs;
    if (plx_mex_release_request(nrhs, prhs)) return;
    if (nrhs != 7) mexErrMsgTxt("Seven inputs required.");
    plx_mex_once();
    int Mdim = (int)mxGetM(prhs[0]), ntap = (int)mxGetM(prhs[1]);
    double mu = mxGetScalar(prhs[4]);
    int L = Mdim - ntap + 1;
    plhs[0] = mxCreateDoubleMatrix(L > 0 ? L : 0, 2, mxCOMPLEX);
    plhs[1] = complex_copy(prhs[1]);
    plhs[2] = complex_copy(prhs[2]);
    int rc = plx_cmaadaptivefilter_m(mxGetPr(prhs[0]), mxGetPi(prhs[0]), Mdim, mxGetPr(plhs[1]), mxGetPi(plhs[1]),
                                     mxGetPr(plhs[2]), mxGetPi(plhs[2]), ntap, mu, mxGetPr(prhs[5]),
                                     mxGetPr(plhs[0]), mxGetPi(plhs[0]));
    if (rc) mexErrMsgTxt(plx_last_error());
}

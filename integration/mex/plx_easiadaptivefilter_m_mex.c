/* [Y,h1,h2] = easiadaptivefilter(xx,h1,h2,taps,mu,sps) with the semantics of the .m TWIN, /root/reference/easiadaptivefilter.m:1,51-84:
 * complex error matrix (abs(), complex denominators, :78-84), ALL taps of the complex h1, h2 recombined (:58-66), the
 * updated taps returned in plhs[1..2]; the inputs are left untouched.  The C file (easiadaptivefilter.c:81-90) only ever
 * touches the real parts of tap 0 -- the twins are not equivalent, and this shim is the .m one. */
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
    if (nrhs != 6) mexErrMsgTxt("Six inputs required.");
    plx_mex_once();
    int Mdim = (int)mxGetM(prhs[0]), ntap = (int)mxGetM(prhs[1]);
    double mu = mxGetScalar(prhs[4]);
    int L = Mdim - ntap + 1;
    plhs[0] = mxCreateDoubleMatrix(L > 0 ? L : 0, 2, mxCOMPLEX);
    plhs[1] = complex_copy(prhs[1]);
    plhs[2] = complex_copy(prhs[2]);
    int rc = plx_easiadaptivefilter_m(mxGetPr(prhs[0]), mxGetPi(prhs[0]), Mdim, mxGetPr(plhs[1]), mxGetPi(plhs[1]),
                                      mxGetPr(plhs[2]), mxGetPi(plhs[2]), ntap, mu, mxGetPr(plhs[0]), mxGetPi(plhs[0]));
    if (rc) mexErrMsgTxt(plx_last_error());
}

/* [Y,h1,h2] = cmaadaptivefilter(xx,h1,h2,taps,mu,R,sps) -- drop-in for /root/reference/cmaadaptivefilter.c:93-174.
 * Like the reference it updates h1, h2 IN THE CALLER'S ARRAYS and returns 0, 0 (:166-171); the drivers detect that with
 * any(any(h1_new)) (DspPdmCohQpsk.m:183-186). */
#include "plx_mex_common.h"
static double *imag_plane(const mxArray *a, size_t n)   /* the reference allocates missing imaginary planes ON THE INPUTS */
{                                                       /* (cmaadaptivefilter.c:141-155)                                 */
    double *pi = mxGetPi(a);
    if (!pi) { pi = (double *)mxCalloc(n, sizeof(double)); mxSetPi((mxArray *)a, pi); }
    return pi;
}
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    (void)nlhs;
    if (plx_mex_release_request(nrhs, prhs)) return;
    if (nrhs != 7) mexErrMsgTxt("Seven inputs required.");
    plx_mex_once();
    int Mdim = (int)mxGetM(prhs[0]), Npol = (int)mxGetN(prhs[0]);
    double Ntap = mxGetScalar(prhs[3]), mu = mxGetScalar(prhs[4]), sps = mxGetScalar(prhs[6]);
    double *xi = imag_plane(prhs[0], (size_t)Mdim * Npol);
    size_t nh = (size_t)(Ntap > 0 ? Ntap : 1) * Npol;
    double *h1i = imag_plane(prhs[1], nh), *h2i = imag_plane(prhs[2], nh);
    int L = Mdim - (int)Ntap + 1;
    plhs[0] = mxCreateDoubleMatrix(L > 0 ? L : 0, Npol, mxCOMPLEX);
    int rc = plx_cmaadaptivefilter(mxGetPr(prhs[0]), xi, Mdim, mxGetPr(prhs[1]), h1i, mxGetPr(prhs[2]), h2i,   /* updated in place */
                                   Ntap, mu, mxGetPr(prhs[5]), sps, mxGetPr(plhs[0]), mxGetPi(plhs[0]));
    if (rc) mexErrMsgTxt(plx_last_error());          /* "Ntaps should be an ODD INTEGER." etc., :118-119,132 */
    plhs[1] = mxCreateDoubleMatrix(1, 1, mxREAL);    /* 0, 0 like the reference                              */
    plhs[2] = mxCreateDoubleMatrix(1, 1, mxREAL);
}

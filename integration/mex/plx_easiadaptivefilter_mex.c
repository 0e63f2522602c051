/* [Y,h1,h2] = easiadaptivefilter(xx,h1,h2,taps,mu,sps) -- drop-in for /root/reference/easiadaptivefilter.c:95-169
 * (six inputs, sps is prhs[5]); in-place h1/h2 and 0, 0 returned, as the reference. */
#include "plx_mex_common.h"
static double *imag_plane(const mxArray *a, size_t n)
{
    double *pi = mxGetPi(a);
    if (!pi) { pi = (double *)mxCalloc(n, sizeof(double)); mxSetPi((mxArray *)a, pi); }
    return pi;
}
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    (void)nlhs;
    if (plx_mex_release_request(nrhs, prhs)) return;
    if (nrhs != 6) mexErrMsgTxt("Six inputs required.");
    plx_mex_once();
    int Mdim = (int)mxGetM(prhs[0]), Npol = (int)mxGetN(prhs[0]);
    double Ntap = mxGetScalar(prhs[3]), mu = mxGetScalar(prhs[4]), sps = mxGetScalar(prhs[5]);
    double *xi = imag_plane(prhs[0], (size_t)Mdim * Npol);
    size_t nh = (size_t)(Ntap > 0 ? Ntap : 1) * Npol;
    double *h1i = imag_plane(prhs[1], nh), *h2i = imag_plane(prhs[2], nh);
    int L = Mdim - (int)Ntap + 1;
    plhs[0] = mxCreateDoubleMatrix(L > 0 ? L : 0, Npol, mxCOMPLEX);
    int rc = plx_easiadaptivefilter(mxGetPr(prhs[0]), xi, Mdim, mxGetPr(prhs[1]), h1i, mxGetPr(prhs[2]), h2i, Ntap, mu, sps,
                                    mxGetPr(plhs[0]), mxGetPi(plhs[0]));
    if (rc) mexErrMsgTxt(plx_last_error());
    plhs[1] = mxCreateDoubleMatrix(1, 1, mxREAL);
    plhs[2] = mxCreateDoubleMatrix(1, 1, mxREAL);
}

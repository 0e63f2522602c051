/* y = fastexp(x) -- drop-in for /root/reference/fastexp.c:46-67 (build: mex -R2017b -I<repo>/include plx_fastexp_mex.c
 * -L<repo>/polmux_amd/lib -lpolmux_hip -output fastexp). */
#include "plx_mex_common.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    (void)nlhs;
    if (plx_mex_release_request(nrhs, prhs)) return;
    if (nrhs != 1) mexErrMsgTxt("One input required.");
    plx_mex_once();
    size_t m = mxGetM(prhs[0]), n = mxGetN(prhs[0]);
    plhs[0] = mxCreateDoubleMatrix(m, n, mxCOMPLEX);
    if (plx_fastexp(mxGetPr(prhs[0]), mxGetPr(plhs[0]), mxGetPi(plhs[0]), m * n))
        mexErrMsgIdAndTxt("polmux:hip", "%s", plx_last_error());
}

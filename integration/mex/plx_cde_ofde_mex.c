/* [outX,outY] = plx_cde_ofde_mex(inX,inY,fs,lambdaRef,span,D,S,fftLength,L) -- the body of CDE_OFDE.m:16-47 behind the
 * unchanged .m signature.  On a bad argument OverlapBothTrans display()s a message and returns [] (CDE_OFDE.m:63-85):
 * the shim does the same. */
#include "plx_mex_common.h"
static double *plane_or_zeros(const mxArray *a, size_t n) { double *p = mxGetPi(a); return p ? p : (double *)mxCalloc(n, sizeof(double)); }
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    (void)nlhs;
    if (plx_mex_release_request(nrhs, prhs)) return;
    if (nrhs != 9) mexErrMsgTxt("Nine inputs required.");
    plx_mex_once();
    size_t nx = mxGetNumberOfElements(prhs[0]);
    plhs[0] = mxCreateDoubleMatrix(nx, 1, mxCOMPLEX);
    plhs[1] = mxCreateDoubleMatrix(nx, 1, mxCOMPLEX);
    int rc = plx_cde_ofde(mxGetPr(prhs[0]), plane_or_zeros(prhs[0], nx), mxGetPr(prhs[1]), plane_or_zeros(prhs[1], nx), (int64_t)nx,
                          mxGetScalar(prhs[2]), mxGetScalar(prhs[3]), mxGetScalar(prhs[4]), mxGetScalar(prhs[5]), mxGetScalar(prhs[6]),
                          (int64_t)mxGetScalar(prhs[7]), (int64_t)mxGetScalar(prhs[8]), mxGetPr(plhs[0]), mxGetPi(plhs[0]),
                          mxGetPr(plhs[1]), mxGetPi(plhs[1]));
    if (rc == PLX_ERR_ARG) {                          /* display('Error: ...'); y = [] */
        mexPrintf("%s\n", plx_last_error());
        mxDestroyArray(plhs[0]); mxDestroyArray(plhs[1]);
        plhs[0] = mxCreateDoubleMatrix(0, 0, mxREAL);
        plhs[1] = mxCreateDoubleMatrix(0, 0, mxREAL);
    } else if (rc) {
        mexErrMsgIdAndTxt("polmux:hip", "%s", plx_last_error());
    }
}

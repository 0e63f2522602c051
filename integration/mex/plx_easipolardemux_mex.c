/* [y,h1,h2,passes] = plx_easipolardemux_mex(x, mu, M [, mfile_twin])
 * the whole driver loop of easipolardemux (/root/reference/DspPdmCohQpsk.m:195-244 == dsp4cohdec.m:427-478) in ONE call: taps = 1
 * (:197), up to 20*ceil(1/(L*mu)) - 1 passes (:227) with the 5e-5 test (:240) on the device.  mfile_twin ~= 0: the loop
 * around the .m twin of the filter (easiadaptivefilter.m:51-84: complex error matrix), which is what MATLAB runs when
 * comp_mex.m was never run; default 0 = the C filter (easiadaptivefilter.c:43-93: real parts of tap 0). */
#include "plx_mex_common.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (plx_mex_release_request(nrhs, prhs)) return;
    if (nrhs != 3 && nrhs != 4) mexErrMsgTxt("Three or four inputs required.");
    if (mxGetN(prhs[0]) != 2) mexErrMsgTxt("x must have two columns.");
    if (mxGetM(prhs[2]) != 2 || mxGetN(prhs[2]) != 2) mexErrMsgTxt("M must be 2 x 2.");
    plx_mex_once();
    size_t L = mxGetM(prhs[0]);
    const double *mr = mxGetPr(prhs[2]), *mi = mxGetPi(prhs[2]);
    double M[8];
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 2; c++) { M[2 * (2 * r + c)] = mr[r + 2 * c]; M[2 * (2 * r + c) + 1] = mi ? mi[r + 2 * c] : 0.0; }
    int twin = nrhs == 4 && mxGetScalar(prhs[3]) != 0.0;
    plhs[0] = mxCreateDoubleMatrix(L, 2, mxCOMPLEX);
    mxArray *h1 = mxCreateDoubleMatrix(1, 2, mxCOMPLEX), *h2 = mxCreateDoubleMatrix(1, 2, mxCOMPLEX);
    int32_t passes = 0;
    int rc = plx_easipolardemux(mxGetPr(prhs[0]), mxGetPi(prhs[0]), (int64_t)L, mxGetScalar(prhs[1]), M, twin, mxGetPr(plhs[0]),
                                mxGetPi(plhs[0]), mxGetPr(h1), mxGetPi(h1), mxGetPr(h2), mxGetPi(h2), &passes);
    if (rc) mexErrMsgTxt(plx_last_error());
    if (nlhs > 1) plhs[1] = h1; else mxDestroyArray(h1);
    if (nlhs > 2) plhs[2] = h2; else mxDestroyArray(h2);
    if (nlhs > 3) plhs[3] = mxCreateDoubleScalar((double)passes);
}

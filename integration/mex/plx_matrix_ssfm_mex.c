/* [firstdz,ncycle,ux,uy] = plx_matrix_ssfm_mex(ux,uy,betat,db1,dzmaxt,dphimaxt,gam,alphalin,nfc,Lf,nplates,manakov,fls,
 *                                              db0,theta,epsilon)
 * the new seam behind fiber.m:380-388: matrix_ssfm (fiber.m:459-554) with brf passed as its three vectors. */
#include "plx_mex_common.h"
#include <string.h>
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    (void)nlhs;
    if (plx_mex_release_request(nrhs, prhs)) return;
    if (nrhs != 16) mexErrMsgTxt("Sixteen inputs required.");
    plx_mex_once();
    plx_ssfm_desc d;
    memset(&d, 0, sizeof(d));
    d.nfft = (int64_t)mxGetM(prhs[0]); d.nfc = (int32_t)mxGetN(prhs[0]); d.dual_pol = 1; d.max_frames = 1;
    d.betat = mxGetPr(prhs[2]); d.db1 = mxGetPr(prhs[3]);
    d.dzmaxt = mxGetScalar(prhs[4]); d.dphimaxt = mxGetScalar(prhs[5]); d.gam = mxGetPr(prhs[6]);
    d.alphalin = mxGetScalar(prhs[7]); d.length = mxGetScalar(prhs[9]);
    d.nplates = (int32_t)mxGetScalar(prhs[10]); d.manakov = (int32_t)mxGetScalar(prhs[11]);
    for (int i = 0; i < 4; i++) d.fls[i] = (int32_t)mxGetPr(prhs[12])[i];
    plhs[2] = mxDuplicateArray(prhs[0]);
    plhs[3] = mxDuplicateArray(prhs[1]);
    size_t n = (size_t)d.nfft * d.nfc;
    if (!mxGetPi(plhs[2])) mxSetPi(plhs[2], (double *)mxCalloc(n, sizeof(double)));   /* real-valued input fields */
    if (!mxGetPi(plhs[3])) mxSetPi(plhs[3], (double *)mxCalloc(n, sizeof(double)));
    double fd = 0;
    int32_t nc = 0;
    if (plx_matrix_ssfm(mxGetPr(plhs[2]), mxGetPi(plhs[2]), mxGetPr(plhs[3]), mxGetPi(plhs[3]), &d, mxGetPr(prhs[13]),
                        mxGetPr(prhs[14]), mxGetPr(prhs[15]), &fd, &nc))
        mexErrMsgTxt(plx_last_error());              /* e.g. the message of fiber.m:854 for XPM + CNLSE */
    plhs[0] = mxCreateDoubleScalar(fd);
    plhs[1] = mxCreateDoubleScalar((double)nc);
}

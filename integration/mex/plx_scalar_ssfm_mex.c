/* [firstdz,ncycle,u] = plx_scalar_ssfm_mex(u,betat,dzmaxt,dphimaxt,gam,alphalin,nfc,Lf,fls)
 * the new seam behind fiber.m:384-388: scalar_ssfm (fiber.m:557-636, tolflag == 0) on a single-polarisation field. */
#include "plx_mex_common.h"
#include <string.h>
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    (void)nlhs;
    if (plx_mex_release_request(nrhs, prhs)) return;
    if (nrhs != 9) mexErrMsgTxt("Nine inputs required.");
    plx_mex_once();
    plx_ssfm_desc d;
    memset(&d, 0, sizeof(d));
    d.nfft = (int64_t)mxGetM(prhs[0]); d.nfc = (int32_t)mxGetN(prhs[0]); d.dual_pol = 0; d.max_frames = 1;
    d.betat = mxGetPr(prhs[1]);
    d.dzmaxt = mxGetScalar(prhs[2]); d.dphimaxt = mxGetScalar(prhs[3]); d.gam = mxGetPr(prhs[4]);
    d.alphalin = mxGetScalar(prhs[5]); d.length = mxGetScalar(prhs[7]); d.nplates = 1;
    for (int i = 0; i < 4; i++) d.fls[i] = (int32_t)mxGetPr(prhs[8])[i];
    plhs[2] = mxDuplicateArray(prhs[0]);
    if (!mxGetPi(plhs[2])) mxSetPi(plhs[2], (double *)mxCalloc((size_t)d.nfft * d.nfc, sizeof(double)));
    double fd = 0;
    int32_t nc = 0;
    if (plx_scalar_ssfm(mxGetPr(plhs[2]), mxGetPi(plhs[2]), &d, &fd, &nc)) mexErrMsgTxt(plx_last_error());
    plhs[0] = mxCreateDoubleScalar(fd);
    plhs[1] = mxCreateDoubleScalar((double)nc);
}

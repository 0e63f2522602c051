/* [firstdz,ncycle,u,nrej] = plx_scalar_a_ssfm_mex(u,betat,dzmaxt,dphimaxt,gam,alphalin,nfc,Lf,fls,tolflag,ltol,safety)
 * the new seam behind fiber.m:376-377 / :386-387: scalar_a_ssfm with adaptssfm (fiber.m:639-679, 938-1009; tolflag == 2,
 * x.ltol set) and the x.dphiadapt first step of scalar_ssfm (:588-611; tolflag == 1), single-polarisation fields only --
 * a dual-polarisation call never gets here: fiber.m:373-375 raises "adaptive step available in absence of polarization
 * effects" first.  ltol = trg.err, safety = trg.safety (fiber.m:145-146: x.ltol, SAFETYFCT = 0.9). */
#include "plx_mex_common.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (plx_mex_release_request(nrhs, prhs)) return;
    if (nrhs != 12) mexErrMsgTxt("Twelve inputs required.");
    if (mxGetNumberOfElements(prhs[8]) != 4) mexErrMsgTxt("fls must have four elements.");
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
    int32_t nc = 0, nrej = 0;
    if (plx_scalar_ssfm_adaptive(mxGetPr(plhs[2]), mxGetPi(plhs[2]), &d, (int)mxGetScalar(prhs[9]), mxGetScalar(prhs[10]),
                                 mxGetScalar(prhs[11]), &fd, &nc, &nrej))
        mexErrMsgTxt(plx_last_error());
    plhs[0] = mxCreateDoubleScalar(fd);
    plhs[1] = mxCreateDoubleScalar((double)nc);
    if (nlhs > 3) plhs[3] = mxCreateDoubleScalar((double)nrej);
}

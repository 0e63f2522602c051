// ssfm_gateway.hip -- the MEX-shaped gateway forms of the propagator (one frame, split planes, host memory) and the
// host-driven adaptive-step scheme of scalar fields (fiber.m:372-389, :639-679, :938-1009).
#include "ssfm_plan.h"
#include "plx_gateway.h"

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>
using namespace plxs;

// ---- gateway forms (one frame, split planes, host memory) --------------------------
// One call = one `fiber()` span of the unchanged MATLAB wrapper (fiber.m:372-389).  The plan, the device field buffers
// and the pinned staging area belong to the library (plx_gateway.h): a span on the same fibre type and grid finds its
// plan (content hash of the descriptor's scalars and tables) and allocates nothing.
static int gateway_ssfm(double *uxr, double *uxi, double *uyr, double *uyi, const plx_ssfm_desc *desc,
                        const double *db0, const double *theta, const double *epsilon, double *firstdz,
                        int32_t *ncycle)
{
    if (!desc || !uxr) PLX_FAIL(PLX_ERR_ARG, "ssfm gateway: null argument");
    plx_ssfm_desc d = *desc;
    d.max_frames = 1;
    const bool dual = d.dual_pol != 0;
    if (dual && !uyr) PLX_FAIL(PLX_ERR_ARG, "matrix_ssfm gateway: missing y field");
    if (!uxi || (dual && !uyi)) PLX_FAIL(PLX_ERR_ARG, "ssfm gateway: output imaginary planes are required");
    std::lock_guard<std::mutex> lk(plxgw::mutex());
    plxgw::count_call();
    int rc = PLX_OK;
    plx_ssfm *P = plxgw::ssfm_plan(d, &rc);
    if (!P) return rc;
    if (dual && d.fls[1]) {
        rc = plx_ssfm_set_birefringence_dev(P, db0, theta, epsilon, 1, nullptr);
        if (rc) return rc;
    }
    const size_t n = (size_t)d.nfft * d.nfc, npol = dual ? 2 : 1;
    const size_t bytes = npol * 2 * n * sizeof(double);
    double *h = (double *)plxgw::pinned(plxgw::S_IN, bytes);
    double *dx = (double *)plxgw::dev(plxgw::S_IN, bytes);
    if (!h || !dx) return PLX_ERR_HIP;
    double *dy = dual ? dx + 2 * n : nullptr, *hy = h + 2 * n;
    for (size_t i = 0; i < n; i++) {
        h[2 * i] = uxr[i]; h[2 * i + 1] = uxi[i];
        if (dual) { hy[2 * i] = uyr[i]; hy[2 * i + 1] = uyi[i]; }
    }
    PLX_HIP(hipMemcpyAsync(dx, h, bytes, hipMemcpyHostToDevice, nullptr));
    // A cached plan that once fell back to the three-sweep step (below) tries the fused step again after rearm_after calls:
    // the stall that caused the time-out is usually gone, and a plan that keeps timing out doubles its patience (16, 32, ... 4096)
    if (!P->fused && P->barrier_timeouts > 0 && ++P->calls_unfused >= P->rearm_after) plx_ssfm_barrier_timeouts(P, nullptr, 1);
    rc = plx_ssfm_propagate_dev(P, dx, dy, 1, nullptr);
    if (rc == PLX_ERR_TIMEOUT) {
        P->calls_unfused = 0;
        if (P->rearm_after < 4096) P->rearm_after *= 2;
        // fiber.m:372-389 always returns a field.  Another kernel holds part of the GPU, so the frame's workgroups could not
        // meet; the caller's input is still in the pinned staging buffer: upload it again and make the same call on the
        // barrier-free three-sweep step (the plan has switched itself), counted in plx_gateway_stats_ex.
        plxgw::count_fallback();
        PLX_HIP(hipMemcpyAsync(dx, h, bytes, hipMemcpyHostToDevice, nullptr));
        rc = plx_ssfm_propagate_dev(P, dx, dy, 1, nullptr);
    }
    if (rc) return rc;
    PLX_HIP(hipMemcpyAsync(h, dx, bytes, hipMemcpyDeviceToHost, nullptr));
    PLX_HIP(hipStreamSynchronize(nullptr));
    for (size_t i = 0; i < n; i++) {
        uxr[i] = h[2 * i]; uxi[i] = h[2 * i + 1];
        if (dual) { uyr[i] = hy[2 * i]; uyi[i] = hy[2 * i + 1]; }
    }
    return plx_ssfm_results(P, 1, firstdz, ncycle);
}

extern "C" int plx_matrix_ssfm(double *uxr, double *uxi, double *uyr, double *uyi, const plx_ssfm_desc *desc,
                               const double *db0, const double *theta, const double *epsilon,
                               double *firstdz, int32_t *ncycle)
{
    if (desc && !desc->dual_pol) PLX_FAIL(PLX_ERR_ARG, "plx_matrix_ssfm: descriptor is not dual-polarisation");
    return gateway_ssfm(uxr, uxi, uyr, uyi, desc, db0, theta, epsilon, firstdz, ncycle);
}

extern "C" int plx_scalar_ssfm(double *ur, double *ui, const plx_ssfm_desc *desc, double *firstdz, int32_t *ncycle)
{
    if (desc && desc->dual_pol) PLX_FAIL(PLX_ERR_ARG, "plx_scalar_ssfm: descriptor is dual-polarisation");
    return gateway_ssfm(ur, ui, nullptr, nullptr, desc, nullptr, nullptr, nullptr, firstdz, ncycle);
}

// ======================================================= adaptive-step scheme (scalar fields) ===
// scalar_a_ssfm / adaptssfm (fiber.m:639-679, 938-1009) and the dphiadapt first step of scalar_ssfm
// (:588-611).  The accept/reject decision needs the global max|u-uh| on the host every trial, so this
// path is host-driven: the element-wise pieces are the kernels k_nl_att / k_maxdiff / k_richardson and
// the linear operator reuses the three transform sweeps with the step length forced from the launch.
namespace {
struct Adaptive {
    plx_ssfm *P;
    SsfmArgs a;
    hipStream_t st;
    size_t n;      // nfc * N
    cplx *u, *uh, *stack;
    unsigned long long *d_max;
    unsigned long long h_max;
    double alphalin;
    int fls2, fls3;

    void lin(cplx *x, double dz)
    { // lin_step(betat*dz, x): x = ifft(fft(x).*fastexp(-betat*dz))
        SsfmArgs b = a;
        b.ux = x; b.uy = nullptr; b.force = 1; b.spm = 0; b.xpm = 0; b.f_cur = dz; b.f_leff = 0; b.f_sc = b.invN;
        const int N1 = 1 << b.p1, N2 = 1 << b.p2;
        const dim3 gcol((unsigned)(N2 / b.W), (unsigned)b.nfc), grow((unsigned)(N1 / b.R), (unsigned)b.nfc);
        launch(col_fwd_kernel(), gcol, dim3(256), P->lds_col, st, b);
        launch(row_kernel(), grow, dim3((unsigned)P->row_threads), P->lds_row, st, b);
        launch(col_inv_kernel(), gcol, dim3(256), P->lds_col, st, b);
    }
    void nl_att(cplx *x, double dz)
    { // nl_step(alphalin,gam,dz,x,...) then x = x*exp(-halfalpha*dz)
        const double leff = (alphalin == 0) ? dz : (1 - exp(-alphalin * dz)) / alphalin;
        const double att = exp(-(0.5 * alphalin) * dz);
        unsigned g = (unsigned)((P->N + 255) / 256);
        if (g > 2048) g = 2048;
        launch_nl_att(g, st, x, a.gam, P->N, a.nfc, fls2, fls3, leff, att);
    }
    // one trial of adaptssfm; returns <0 on HIP failure
    int trial(double &zdone, double &dz, double trg_err, double safety, int &nrej, int &ncycle)
    {
        const double dz1 = dz, dz2 = 0.5 * dz1, dz4 = 0.25 * dz1;
        if (hipMemcpyAsync(stack, u, n * sizeof(cplx), hipMemcpyDeviceToDevice, st) != hipSuccess) return -1;
        if (hipMemcpyAsync(uh, u, n * sizeof(cplx), hipMemcpyDeviceToDevice, st) != hipSuccess) return -1;
        nl_att(u, dz2); lin(u, dz1); nl_att(u, dz2);                                        // :972-979
        nl_att(uh, dz4); lin(uh, dz2); nl_att(uh, dz2); lin(uh, dz2); nl_att(uh, dz4);      // :983-993
        if (hipMemsetAsync(d_max, 0, sizeof(unsigned long long), st) != hipSuccess) return -1;
        unsigned g = (unsigned)((n + 255) / 256);
        if (g > 1024) g = 1024;
        launch_maxdiff(g, st, u, uh, n, d_max);
        if (hipMemcpyAsync(&h_max, d_max, sizeof(h_max), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
        if (hipStreamSynchronize(st) != hipSuccess) return -1;
        double emax;
        std::memcpy(&emax, &h_max, sizeof(double));
        const double est_err = emax / dz1;                                                  // :997
        if (est_err > trg_err) {                                                            // reject :999-1002
            dz = safety * sqrt(trg_err / est_err) * dz1;
            if (hipMemcpyAsync(u, stack, n * sizeof(cplx), hipMemcpyDeviceToDevice, st) != hipSuccess) return -1;
            nrej = nrej + 1;
        } else {                                                                            // accept :1003-1008
            launch_richardson(g, st, u, uh, n);
            zdone = zdone + dz1;
            dz = safety * sqrt(trg_err / est_err) * dz1;
            ncycle = ncycle + 1;
        }
        return 0;
    }
};

// host copy of nextstep (fiber.m:682-715) from the per-channel maxima
double host_nextstep(double dzmax, double phimax, const double *gam, const double *umax, int nfc, double alphalin, double *pmax_out)
{
    double Pmax = -INFINITY;
    for (int k = 0; k < nfc; k++) { const double gp = gam[k] * umax[k]; Pmax = gp > Pmax ? gp : Pmax; }
    if (pmax_out) *pmax_out = Pmax;
    const double leff = phimax / Pmax, dl = alphalin * leff;
    if (dl >= 1) return dzmax;
    const double step = (alphalin == 0) ? leff : -1 / alphalin * log(1 - dl);
    return step > dzmax ? dzmax : step;
}
} // namespace

extern "C" int plx_scalar_ssfm_adaptive(double *ur, double *ui, const plx_ssfm_desc *desc, int tolflag, double ltol,
                                        double safety, double *firstdz, int32_t *ncycle_out, int32_t *nrej_out)
{
    if (!ur || !ui || !desc) PLX_FAIL(PLX_ERR_ARG, "plx_scalar_ssfm_adaptive: null argument");
    if (desc->dual_pol) PLX_FAIL(PLX_ERR_REFERENCE, "adaptive step available in absence of polarization effects"); // fiber.m:374
    if (tolflag != 1 && tolflag != 2) PLX_FAIL(PLX_ERR_ARG, "plx_scalar_ssfm_adaptive: tolflag must be 1 or 2");
    plx_ssfm_desc d = *desc;
    d.max_frames = 1;
    // plan, the three field copies of adaptssfm and the staging area come from the library's gateway workspace
    std::lock_guard<std::mutex> lk(plxgw::mutex());
    plxgw::count_call();
    int rc = PLX_OK;
    plx_ssfm *P = plxgw::ssfm_plan(d, &rc);
    if (!P) return rc;
    const SsfmArgs saved = P->a;          // (the resume fields below are per call: the cached plan is handed back as it was)
    const size_t n = (size_t)d.nfft * d.nfc;
    double *h = (double *)plxgw::pinned(plxgw::S_IN, 2 * n * sizeof(double));
    cplx *fld = (cplx *)plxgw::dev(plxgw::S_IN, (3 * n + 8) * sizeof(cplx));
    if (!h || !fld) return PLX_ERR_HIP;
    for (size_t i = 0; i < n; i++) { h[2 * i] = ur[i]; h[2 * i + 1] = ui[i]; }
    Adaptive A;
    A.P = P; A.a = P->a; A.st = nullptr; A.n = n; A.u = fld; A.uh = fld + n; A.stack = fld + 2 * n;
    A.d_max = (unsigned long long *)(fld + 3 * n);
    A.a.nframes = 1; A.alphalin = d.alphalin; A.fls2 = d.fls[2]; A.fls3 = d.fls[3];
    auto cleanup = [&]() { P->a = saved; };
    if (hipMemcpy(A.u, h, n * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(P->d_ctl, 0, sizeof(FrameCtl)) != hipSuccess || hipMemset(P->d_ndone, 0, 64) != hipSuccess ||
        hipMemset(P->d_umax, 0, sizeof(unsigned long long) * d.nfc) != hipSuccess) {
        cleanup();
        PLX_FAIL(PLX_ERR_HIP, "plx_scalar_ssfm_adaptive: device allocation/upload failed");
    }
    // first step from nextstep (:585 / :666): per-channel maxima of |u|^2
    std::vector<unsigned long long> um(d.nfc);
    {
        SsfmArgs b = A.a;
        b.ux = A.u; b.uy = nullptr;
        unsigned gx = (unsigned)((P->N + 255) / 256);
        if (gx > 64) gx = 64;
        launch_umax(dim3(gx, (unsigned)d.nfc), nullptr, b);
        if (hipMemcpy(um.data(), P->d_umax, sizeof(unsigned long long) * d.nfc, hipMemcpyDeviceToHost) != hipSuccess) {
            cleanup();
            PLX_FAIL(PLX_ERR_HIP, "plx_scalar_ssfm_adaptive: readback failed");
        }
    }
    std::vector<double> umax(d.nfc), gam(d.gam, d.gam + d.nfc);
    for (int k = 0; k < d.nfc; k++) std::memcpy(&umax[k], &um[k], sizeof(double));
    double maxpow = 0;
    double dphimaxt = d.dphimaxt;
    double dz = host_nextstep(d.dzmaxt, dphimaxt, gam.data(), umax.data(), d.nfc, d.alphalin, &maxpow);
    int ncycle = 1, nrej = 0;
    const double Lf = d.length;
    if (tolflag == 2) { // scalar_a_ssfm :664-679
        *firstdz = dz;
        double zdone = 0;
        while (zdone < Lf) {
            if (zdone + dz > Lf) dz = Lf - zdone;
            if (A.trial(zdone, dz, ltol, safety, nrej, ncycle)) { cleanup(); PLX_FAIL(PLX_ERR_HIP, "plx_scalar_ssfm_adaptive: HIP failure in adaptssfm"); }
            if (dz > d.dzmaxt) dz = d.dzmaxt;
        }
        rc = (hipMemcpy(h, A.u, n * sizeof(cplx), hipMemcpyDeviceToHost) == hipSuccess) ? PLX_OK : PLX_ERR_HIP;
    } else { // dphiadapt: adaptive first step, then the constant-phase loop (:588-636)
        if (dz >= d.dzmaxt) { // :589-597
            if (d.alphalin == 0) dphimaxt = maxpow * dz;
            else dphimaxt = maxpow * (1 - exp(-d.alphalin * dz)) / d.alphalin;
        }
        const double dzini = dz;
        double zdone = 0;
        while (zdone == 0) {
            int nc = 0;
            nrej = 0;
            if (A.trial(zdone, dz, ltol, safety, nrej, nc)) { cleanup(); PLX_FAIL(PLX_ERR_HIP, "plx_scalar_ssfm_adaptive: HIP failure in adaptssfm"); }
            ncycle = nc;
        }
        if (dz > d.dzmaxt) dz = d.dzmaxt;
        dphimaxt = dphimaxt * (1 - exp(-d.alphalin * zdone)) / (1 - exp(-d.alphalin * dzini)); // :607
        P->a.resume = 1; P->a.dz0 = dz; P->a.zdone0 = zdone; P->a.ncycle0 = ncycle; P->a.dphimax = dphimaxt;
        rc = plx_ssfm_propagate_dev(P, (double *)A.u, nullptr, 1, nullptr);
        if (!rc) {
            rc = (hipMemcpy(h, A.u, n * sizeof(cplx), hipMemcpyDeviceToHost) == hipSuccess) ? PLX_OK : PLX_ERR_HIP;
            *firstdz = P->h_ctl[0].firstdz;
            ncycle = P->h_ctl[0].ncycle;
        }
    }
    cleanup();
    if (rc == PLX_ERR_HIP) plx_set_error("plx_scalar_ssfm_adaptive: download failed");
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) { ur[i] = h[2 * i]; ui[i] = h[2 * i + 1]; }
    if (ncycle_out) *ncycle_out = ncycle;
    if (nrej_out) *nrej_out = nrej;
    return PLX_OK;
}

// ssfm_ctrl.h -- the step controller of matrix_ssfm / scalar_ssfm as device code: nextstep (fiber.m:682-715), the loop
// head / tail (:512-551, :585-636) and checkstep (:718-758).  Used by k_ctrl (ssfm_small.hip, one lane per frame) and by
// the fused column sweep (ssfm_colx.hip, every workgroup of a frame on an LDS copy of the record).
#pragma once
#include "ssfm_args.h"

namespace plxs {

// --------------------------------------------------------------- step control ---
// nextstep (fiber.m:682-715), the loop head/tail of matrix_ssfm/scalar_ssfm
// (:512-551, :585-636) and checkstep (:718-758), one lane per frame.
// have_pmax: Pmax = max_k gam(k)*Umax(k) (fiber.m:694-698) is handed in by the caller (the fused sweep collects it from
// the per-tile slots of its frame barrier) instead of being formed from the umax words.
// The controller comes in two parts.  ctrl_head is what the next nonlinear step waits for: the loop tail of the step just
// finished, nextstep, the step length and Leff; ctrl_tail (the attenuation of the step and checkstep) is consumed by the row
// pass and by the NEXT column pass only.  k_ctrl runs them back to back; the fused sweep runs the head in every workgroup of
// the frame on an LDS copy of the record, and the tail off the critical path in the one workgroup that writes the record back.
// ctrl_head returns false when the frame is finished (c.done set).
template <bool AGENT, class ARGS, class REC> __device__ __forceinline__ bool ctrl_head(const ARGS &a, int f, REC &c, bool have_pmax, double pmax_in, bool count = true)
{
    if (c.started) {
        if (a.dual) c.ntot = c.ntot + c.ntrunk - c.nmem; // :529
        if (c.last) {
            c.done = 1;
            if (count) atomicAdd(a.ndone, 1);      // (fused sweep: the controller runs in every workgroup of the frame, one of them counts)
            return false;
        }
    }
    // nextstep
    double Pmax = -INFINITY;
    if (have_pmax) {
        Pmax = pmax_in;
    } else {
        for (int k = 0; k < a.nfc; k++) {
            unsigned long long *up = a.umax + f * a.nfc + k;
            const unsigned long long bits = AGENT ? ld_agent(up) : *up;
            double Umax = __longlong_as_double((long long)bits);
            double gp = a.gam[k] * Umax;
            Pmax = gp > Pmax ? gp : Pmax;
            if (AGENT) st_agent(up, 0ull); else *up = 0ull;
        }
    }
    double leffn = a.dphimax / Pmax;
    double dl = a.alphalin * leffn;
    double dz;
    if (dl >= 1) {
        dz = a.dzmax;
    } else {
        double step = (a.alphalin == 0) ? leffn : -1 / a.alphalin * log(1 - dl);
        dz = step > a.dzmax ? a.dzmax : step;
    }
    {   // (diagnostics: a replayed step sequence, the frame's own sequence logged; both off in production plans)
        const int k = c.started ? c.ncycle : 0;        // 0-based index of the step whose length this is
        if (a.dzlist && k < a.ndz) dz = a.dzlist[k];
        if (a.dzlog && count && k < a.logcap) a.dzlog[(size_t)f * a.logcap + k] = dz;
    }
    if (!c.started) {
        c.started = 1;
        if (a.resume) { // fiber.m:604-611: dz proposed by the adaptive first step, already capped at dzmax
            dz = a.dz0;
            c.firstdz = a.zdone0;
            c.zprop = a.zdone0 + dz;
            c.ncycle = a.ncycle0 + 1;
        } else {
            c.firstdz = dz;
            c.zprop = dz;
            c.ncycle = 1;
        }
    } else {
        c.zprop = c.zprop + dz;
        c.ncycle = c.ncycle + 1;
    }
    c.dz = dz;
    if (c.zprop < a.Lf) {
        c.cur = dz; c.last = 0;
    } else {
        c.cur = a.Lf - c.zprop + dz; c.last = 1; // :538, :545
    }
    c.leff = (a.alphalin == 0) ? c.cur : (1 - exp(-a.alphalin * c.cur)) / a.alphalin;
    return true;
}
template <class ARGS, class REC> __device__ __forceinline__ void ctrl_tail(const ARGS &a, REC &c)
{
    const double zc = c.last ? a.Lf : c.zprop;
    c.att = exp(-(0.5 * a.alphalin) * c.cur);
    if (a.dual) { // checkstep
        const double lcorr = a.lcorr;
        int nzc = (int)ceil(zc / lcorr);
        if (c.dz_miss == 0) {
            c.nmem = 0;
            c.ntrunk = nzc - c.ntot;
            double dzlast = c.cur - lcorr * (c.ntrunk - 1);
            c.dzb_first = c.ntrunk > 1 ? lcorr : dzlast;
            c.dzb_last = dzlast;
            c.dz_miss = lcorr - dzlast;
        } else {
            c.nmem = 1;
            c.ntrunk = nzc - c.ntot + 1;
            if (c.ntrunk == 1) {
                c.dzb_first = c.cur; c.dzb_last = c.cur;
                c.dz_miss = c.dz_miss - c.cur;
            } else {
                double dzlast = c.cur - c.dz_miss - lcorr * (c.ntrunk - 2);
                c.dzb_first = c.dz_miss; c.dzb_last = dzlast;
                c.dz_miss = lcorr - dzlast;
            }
        }
    }
}
template <bool AGENT, class ARGS> __device__ __forceinline__ void ctrl_core(const ARGS &a, int f, FrameCtl &c, bool have_pmax, double pmax_in)
{
    if (c.done) return;
    if (ctrl_head<AGENT>(a, f, c, have_pmax, pmax_in)) ctrl_tail(a, c);
}
// The fused sweep calls the controller OUT OF LINE: inlined, the libm log / exp constants are hoisted into registers for the
// whole kernel and push the 16-point register blocks into scratch.  k: the plan's step-control constants, rec: the
// workgroup's copy of the frame's record, both in LDS (LDS pointers by type: through a generic pointer every field would be
// a flat access, and scalar arguments would be re-read from the kernel argument segment at every call).
struct CtrlK {
    double dphimax, alphalin, dzmax, dz0, zdone0, Lf, lcorr;
    int dual, resume, ncycle0, nfc;
    int *ndone;
    unsigned long long *umax;
    const double *gam;
    const double *dzlist;
    double *dzlog;
    int ndz, logcap;
};
static_assert(sizeof(CtrlK) <= 128, "k_colx16 reserves 128 bytes of LDS for the constants");
// Returns Leff of the next step, or -1 when the frame has reached the fibre end.
template <bool AGENT> __device__ __forceinline__ void ctrl_step(const SsfmArgs &a, int f, bool have_pmax = false, double pmax_in = 0.0)
{
    FrameCtl c = a.ctl[f];
    if (c.done) return;
    ctrl_core<AGENT>(a, f, c, have_pmax, pmax_in);
    a.ctl[f] = c;
}

} // namespace plxs

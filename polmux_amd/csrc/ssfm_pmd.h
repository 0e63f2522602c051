// ssfm_pmd.h -- the waveplate loop of matrix_step (fiber.m:907-933) as device code shared by the row passes.
#pragma once
#include "ssfm_args.h"

namespace plxs {

// The waveplate loop of matrix_step (fiber.m:907-933) at one frequency.  Per trunk the reference applies
//   u <- R diag(e^{-i(a+b)}, e^{-i(a-b)}) R^H u,   a = betat*dzb(k),  b = (db1 + db0(n))/2 * dzb(k)/lcorr.
// With diag(e^{-ib}, e^{ib}) = cos b I - i sin b sigma3 this is  e^{-ia} (cos b u - i sin b S u),  S = R sigma3 R^H
// (Hermitian, traceless, frequency independent: S11 real, S12 complex -- formed on the host per waveplate), and the
// scalar factors e^{-ia} of all trunks multiply to e^{-i betat * dz}: one exponential per trunk instead of two, 20
// multiply-adds instead of 40.  BRF_STRIDE doubles per plate: S11, Re S12, Im S12, db0 (turns).
__device__ __forceinline__ void pmd_trunks(cplx &x, cplx &y, double btf, double d1f, const double *brf, int nplates, int n0, int ntrunk,
                                           double dzb_first, double dzb_last, double lcorr, double dz_total)
{
    const double rl = 1.0 / lcorr;          // x / lcorr as x*rl with one Newton correction (rounds like the quotient; see div3)
    for (int k = 1; k <= ntrunk; k++) {
        int plate = n0 + k - 1; // the reference indexes brf.theta(n) unchecked; stay in bounds
        plate = plate < 0 ? 0 : (plate >= nplates ? nplates - 1 : plate);
        const double *m = brf + (size_t)plate * BRF_STRIDE;
        const double s11 = m[0];
        const cplx s12 = make_double2(m[1], m[2]);
        const double dzk = (k == 1) ? dzb_first : (k == ntrunk ? dzb_last : lcorr);
        const double num = 0.5 * (d1f + m[3]) * dzk, q0 = num * rl;
        const double deltabeta = fma(fma(-lcorr, q0, num), rl, q0);            // = num / lcorr, :925 (turns)
        const cplx e = cexp_neg_turns(deltabeta);                               // (cos b, -sin b)
        const cplx sx = cadd(cscale(x, s11), cmul(s12, y));                     // S u
        const cplx sy = csub(cmulc(x, s12), cscale(y, s11));
        // cos b u - i sin b S u, with e.y = -sin b:  -i sin b (p + i q) = e.y (-q) ... written out per component
        x = make_double2(e.x * x.x - e.y * sx.y, e.x * x.y + e.y * sx.x);
        y = make_double2(e.x * y.x - e.y * sy.y, e.x * y.y + e.y * sy.x);
    }
    const cplx h = cexp_neg_turns(btf * dz_total);                              // prod_k e^{-i betat dzb(k)}, :924,:927-928
    x = cmul(h, x);
    y = cmul(h, y);
}

// The same loop with the trunk phasors read from the tables of k_pmd_tab: e = e1[trunk][row] * e2[trunk][column].
__device__ __forceinline__ void pmd_trunks_tab(cplx &x, cplx &y, double btf, const cplx *e1, int s1, const cplx *e2, int s2, const double *brf,
                                               int nplates, int n0, int ntrunk, double dz_total)
{
    for (int k = 1; k <= ntrunk; k++) {
        int plate = n0 + k - 1;
        plate = plate < 0 ? 0 : (plate >= nplates ? nplates - 1 : plate);
        const double *m = brf + (size_t)plate * BRF_STRIDE;
        const double s11 = m[0];
        const cplx s12 = make_double2(m[1], m[2]);
        const cplx e = cmul(e1[(size_t)(k - 1) * s1], e2[(size_t)(k - 1) * s2]);     // (cos b, -sin b)
        const cplx sx = cadd(cscale(x, s11), cmul(s12, y));
        const cplx sy = csub(cmulc(x, s12), cscale(y, s11));
        x = make_double2(e.x * x.x - e.y * sx.y, e.x * x.y + e.y * sx.x);
        y = make_double2(e.x * y.x - e.y * sy.y, e.x * y.y + e.y * sy.x);
    }
    const cplx h = cexp_neg_turns(btf * dz_total);
    x = cmul(h, x);
    y = cmul(h, y);
}

// The multiplier of a row pass that holds BOTH polarisations of a row in one wave (k_row4k<true>, k_rowreg<., true>): lane i
// (X) and lane i + 32 (Y) hold the same sixteen bins; they trade halves (half_trade) so that each holds ux and uy of eight bins
// -- x[k] = ux, x[k + 8] = uy of bin ib + k -- apply the waveplate trunks of matrix_step (fiber.m:907-933; phasor tables of
// k_pmd_tab, or one exponential per bin and trunk) or inverse_pmd's matrix tables (inverse_pmd.m:130-141), and trade back.
// LOGM: log2 of the row length (the stride of the column phasors); btv: betat (turns) of the lane's eight bins; row: the row's
// index in the frame (the row phasors); ib: the first of the lane's eight bins within the row.
// TABONLY: the caller's plan has phasor tables (a linear db1) and no matrix tables are in play: only that trunk form is compiled
// in (the general one, an exponential per bin and trunk inlined eight times, is what holds ~90 registers more), and the phases
// are asked for here, behind the trunk loop (btv unused).
template <int LOGM, bool TABONLY = false> __device__ __forceinline__ void pair_multiplier(const SsfmArgs &a, cplx *x, const double *btv, const cplx *ct, const FrameCtl *ctl,
                                                                    int f, int c, int row, size_t rowbase, int ib)
{
    const size_t N = (size_t)1 << (a.p1 + a.p2);
    if (!TABONLY && a.umat) {
        // Uinv = conj(Hgvd) [conj(U11) -U12; conj(U12) U11] applied to [x; y]  (inverse_pmd.m:130-141; k_row's form, bin by bin)
        const cplx *um = a.umat + 3 * ((size_t)f * N + rowbase + ib);
#pragma unroll
        for (int k = 0; k < 8; k++) half_trade(x[k], x[k + 8]);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const cplx u11 = um[3 * k], u12 = um[3 * k + 1], hg = um[3 * k + 2], p = x[k], q = x[k + 8];
            x[k] = cmulc(csub(cmulc(p, u11), cmul(u12, q)), hg);
            x[k + 8] = cmulc(cadd(cmulc(p, u12), cmul(u11, q)), hg);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) half_trade(x[k], x[k + 8]);
        return;
    }
    const double cur = a.force ? a.f_cur : ctl->cur;
    const double *brf = a.brf + (a.brf_per_frame ? (size_t)f * a.nplates * BRF_STRIDE : 0);
    const int ntrunk = ctl->ntrunk, n0 = ctl->ntot - ctl->nmem;
#pragma unroll
    for (int k = 0; k < 8; k++) half_trade(x[k], x[k + 8]);
    if (TABONLY || (a.e1tab && ntrunk <= a.tmax)) {
        const int N1 = 1 << a.p1;
        const cplx *e1 = a.e1tab + (size_t)f * a.tmax * N1 + row, *e2 = a.e2tab + ((size_t)f * a.tmax << LOGM) + ib;
        // (pmd_trunks_tab with the trunk loop outside the bins: a trunk's plate and row phasor are fetched once)
        for (int t = 0; t < ntrunk; t++) {
            int plate = n0 + t;
            plate = plate < 0 ? 0 : (plate >= a.nplates ? a.nplates - 1 : plate);
            const double *m = brf + (size_t)plate * BRF_STRIDE;
            const double s11 = m[0];
            const cplx s12 = make_double2(m[1], m[2]);
            const cplx e1v = e1[(size_t)t * N1];
            const cplx *e2t = e2 + ((size_t)t << LOGM);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const cplx e = cmul(e1v, e2t[k]);
                const cplx u = x[k], v = x[k + 8];
                const cplx sx = cadd(cscale(u, s11), cmul(s12, v));
                const cplx sy = csub(cmulc(u, s12), cscale(v, s11));
                x[k] = make_double2(e.x * u.x - e.y * sx.y, e.x * u.y + e.y * sx.x);
                x[k + 8] = make_double2(e.x * v.x - e.y * sy.y, e.x * v.y + e.y * sy.x);
                if (TABONLY && (k & 1)) sched_fence();
            }
        }
        double bl[8];
        if (TABONLY) {
            int o = ib;
            pin(o);
            const double *bt = a.betat_p + (size_t)c * N + rowbase + o;
#pragma unroll
            for (int k = 0; k < 8; k++) bl[k] = bt[k];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const cplx h = cexp_neg_turns_tab((TABONLY ? bl[k] : btv[k]) * cur, ct);
            x[k] = cmul(h, x[k]);
            x[k + 8] = cmul(h, x[k + 8]);
        }
    } else if (!TABONLY) {
        const double *d1 = a.db1_p + (size_t)c * N + rowbase + ib;
        const double dzb_first = ctl->dzb_first, dzb_last = ctl->dzb_last;
        for (int k = 0; k < 8; k++) pmd_trunks(x[k], x[k + 8], btv[k], d1[k], brf, a.nplates, n0, ntrunk, dzb_first, dzb_last, a.lcorr, cur);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) half_trade(x[k], x[k + 8]);
}


} // namespace plxs

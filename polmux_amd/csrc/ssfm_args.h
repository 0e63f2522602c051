// ssfm_args.h -- the argument block and per-frame step state shared by every kernel of the split-step Fourier propagator
// (fiber.m:459-935), and the small device helpers they have in common.  Included by the kernel translation units
// (ssfm_*.hip) and by the plan (ssfm_plan.hip); nothing here is part of the C ABI.
#pragma once
#include "plx_fft.h"
#include "plx_internal.h"

#define BRF_STRIDE 4      // doubles per waveplate in SsfmArgs::brf: S11, Re S12, Im S12 (S = R sigma3 R^H), db0 in turns (ssfm_pmd.h)

// a wave waits for its own LDS operations (exchanges that stay inside one wave need no workgroup barrier)
#ifdef PLX_EMU
#define ROWR_SYNC() __syncthreads()
#else
#define ROWR_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

namespace plxs {

struct FrameCtl {
    double zprop, dz;       // running distance, last nextstep() result      fiber.m:512-535
    double cur;             // length of the step being executed (dz or last_step :538)
    double leff, att;       // effective length of cur (:787-791), exp(-alpha/2*cur) (:531)
    double dz_miss;         // checkstep state (:741-757)
    double firstdz;
    double dzb_first, dzb_last;
    int ntot, nmem, ntrunk; // waveplate bookkeeping (:524-529)
    int started, last, ncycle;
    int pad_[7];
    int done;               // LAST word of the record: k_colx16 lands its copy by LDS-DMA and spins on this word (sentinel)
};
static_assert(sizeof(FrameCtl) == 128 && offsetof(FrameCtl, done) == 124, "done is the last word of the last 16-byte piece");

struct SsfmArgs {
    cplx *ux, *uy;
    const double *betat_p, *db1_p; // [nfc][N], bit-reversed/transposed order
    const cplx *tpass;             // [N] inter-pass twiddles W_N^(n2*k1), same order
    const cplx *hmul;              // [N] general spectral multiplier replacing exp(-i betat dz) (filter passes), same order
    const cplx *umat;              // [F][N][3] per frequency: SU(2) row (U11, U12) and scalar Hgvd; applies (Hgvd U)^H (inverse_pmd.m:130-141)
    const cplx *tw1, *tw2;         // half tables W_N1^k, W_N2^k
    const cplx *ctab;              // [PLX_CTAB] (cos, -sin)(2 pi k / 64): cexp_neg_turns_tab (k_row4k)
    const cplx *tw2c, *twmid;      // k_rowreg: compact table of W_N2 (TwCompact) and the middle level's lane twiddles [7][16]
    const double *gam;             // [nfc] effective gamma (x8/9 when Manakov, :499-501)
    const double *brf;             // [sets][nplates][9]: R11 R12 R21 R22 (re,im) db0
    // PMD plans whose db1 is LINEAR in the signed frequency index m (fiber.m:358: db1 = dgdrms*omega): the trunk phase
    // deltabeta(m) = A m + B (fiber.m:925) factors over the four-step split m = k1 + N1 m2 into a row and a column phasor,
    // e1[frame][trunk][N1] = exp(-i 2 pi (A k1 + B)) and e2[frame][trunk][N2] = exp(-i 2 pi A N1 m2), formed once per frame
    // and step by k_pmd_tab; the row pass then needs one complex product per bin and trunk instead of an exponential.
    cplx *e1tab, *e2tab;
    double d1slope;                // D: db1 in turns per unit of m
    int tmax;                      // trunks the tables hold per frame
    double *psum;                  // [F][N] row-sum of channel powers (scalar XPM, :795)
    FrameCtl *ctl;
    unsigned long long *umax;      // [F][nfc] bit pattern of max |u|^2 (>= 0)
    int *ndone;                    // [0] frames that have reached the fibre end, [1] sticky abort word (a frame barrier timed out)
    const int *active;             // [nframes] frames still propagating, in frame order (k_compact); nullptr: all of them
    int *nactive;                  // [0] length of the list, [1] running sum of it over the steps (utilisation accounting)
    long long spin_ticks;          // frame-barrier timeout in ticks of plx_clock() (10 ns)
    unsigned long long *slots;     // [2][nframes][tiles per frame] per-tile max |u|^2 by launch parity (k_colx16), ~0 = not arrived
    int row_rev;                   // k_row256r takes the listed frames in DESCENDING order (the column sweep takes them ascending: each kernel starts on the frames the other finished with, which are still in the Infinity Cache)
    int store_late;                // fused sweep: a tile's stores are issued AFTER the next tile has landed (multi-team launches: see k_colx16)
    int safe_land;                 // PLX_SSFM_SAFE_LANDING=1: the staged tile is also waited for with s_waitcnt vmcnt(0) (checks the sentinel landing)
    int round;                     // launch index of the fused sweep within this propagate call
    int *grab;                     // [2] frames claimed beyond the first of every team, by launch parity (k_colx16)
    unsigned long long *mbox;      // [teams][mbox_stride] the team's frame of iteration k, posted by its first workgroup: (launch, k, frame)
    int mbox_stride;               // entries per team: one per iteration a team can reach in a launch (no reuse: a team's workgroups
                                   // run through finished frames of a stale list without meeting, and its first may be far ahead)
    int p1, p2, nfc, dual, W, logW, T, logT, R, logR; // column tile: N1 rows x T complex (T = W*npol)
    int spm, xpm, manakov, pmd, nplates, brf_per_frame;
    int nframes; // frames of the current propagate call (kernels return at once when all are done)
    // host-driven sub-steps of the adaptive scheme (adaptssfm, fiber.m:938-1009): step length, effective
    // length and attenuation come from the launch arguments instead of the per-frame controller
    int force;
    double f_cur, f_leff, f_sc;
    // diagnostics (plx_ssfm_set_step_sequence / plx_ssfm_log_steps): nextstep's result of step k (0-based) replaced by
    // dzlist[k] while k < ndz -- parity tests replay the ORACLE's step sequence on noise-loaded fields, where the step
    // rule amplifies rounding differences ("Conditioning of the step rule") -- and every frame's own sequence
    // written to dzlog[frame][k], k < logcap
    const double *dzlist;
    int ndz, logcap;
    double *dzlog;
    // resume the constant-phase loop after an adaptive first step (tolflag == 1, fiber.m:588-611)
    int resume, ncycle0;
    double dz0, zdone0;
    double alphalin, Lf, dzmax, dphimax, lcorr, invN;
};

__device__ __forceinline__ double wave_max(double v)
{
    for (int m = 32; m >= 1; m >>= 1) {
        double o = __shfl_xor(v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}

// Every kernel of the step loop returns at once when all frames of the call are done, or after a frame barrier of the
// fused sweep has timed out (sticky: nothing is stored or advanced any more, the host reports the error).
template <class ARGS> __device__ __forceinline__ bool all_done_or_aborted(const ARGS &a) { return a.ndone[0] >= a.nframes || a.ndone[1] != 0; }

// Frames leave the step loop one by one (data-dependent trip count, fiber.m:518): the launches of a step cover the
// frames of the ACTIVE list only.  slot -> frame; false when the slot lies beyond the list.
__device__ __forceinline__ bool slot_frame(const SsfmArgs &a, int slot, int &f)
{
    if (!a.active) { f = slot; return slot < a.nframes; }
    if (slot >= a.nactive[0]) return false;
    f = a.active[slot];
    return true;
}

// The row pass takes the listed frames in DESCENDING order where the column sweep takes them ascending (SsfmArgs::row_rev): each
// kernel then starts on the frames the other has just finished with, which are still in the 256 MiB Infinity Cache (128 frames
// of 2^16 samples, 8 of 2^20).  false: the slot lies beyond the list (the host's grid may be longer than the list).
__device__ __forceinline__ bool row_slot(const SsfmArgs &a, int &slot)
{
    if (!a.row_rev) return true;
    const int n = a.active ? a.nactive[0] : a.nframes;
    if (slot >= n) return false;
    slot = n - 1 - slot;
    return true;
}

// exp(i a) for the Kerr step.  The step controller bounds |a| by dphimax (fiber.m:699), a few
// mrad, so the Taylor branch (|a| < 2^-4, truncation < 1e-25, ~1 ulp) is the one that runs;
// larger arguments ('--s-' exact single step, fiber.m:172-174) take the full-range sincos.
__device__ __forceinline__ void sincos_small(double a, double *s, double *c)
{
    if (fabs(a) < 0.0625) {
        const double z = a * a;
        double ps = fma(z, -1.0 / 39916800.0, 1.0 / 362880.0);
        ps = fma(z, ps, -1.0 / 5040.0);
        ps = fma(z, ps, 1.0 / 120.0);
        ps = fma(z, ps, -1.0 / 6.0);
        *s = fma(a * z, ps, a);
        double pc = fma(z, 1.0 / 479001600.0, -1.0 / 3628800.0);
        pc = fma(z, pc, 1.0 / 40320.0);
        pc = fma(z, pc, -1.0 / 720.0);
        pc = fma(z, pc, 1.0 / 24.0);
        pc = fma(z, pc, -0.5);
        *c = fma(z, pc, 1.0);
    } else {
        sincos(a, s, c);
    }
}

// branch-free Taylor form (callers guarantee |a| < 2^-4)
// x / 3 (fiber.m:844 divides) without the ~20-instruction IEEE division sequence: one Newton correction of x * (1/3)
// (Markstein: q = x*c, r = x - 3q exactly in an fma, q + r*c rounds like the quotient)
__device__ __forceinline__ double div3(double x)
{
    const double c = 1.0 / 3.0;
    const double q = x * c;
    return fma(fma(-3.0, q, x), c, q);
}
__device__ __forceinline__ void sincos_taylor(double a, double *s, double *c)
{   // |a| < 2^-4: the first neglected terms, a^11/11! and a^10/10!, are below 3e-20 and 3e-19 relative
    const double z = a * a;
    double ps = fma(z, 1.0 / 362880.0, -1.0 / 5040.0);
    ps = fma(z, ps, 1.0 / 120.0);
    ps = fma(z, ps, -1.0 / 6.0);
    *s = fma(a * z, ps, a);
    double pc = fma(z, 1.0 / 40320.0, -1.0 / 720.0);
    pc = fma(z, pc, 1.0 / 24.0);
    pc = fma(z, pc, -0.5);
    *c = fma(z, pc, 1.0);
}

// block-wide max -> one atomicMax.  red: LDS scratch of >= 16 doubles.
__device__ __forceinline__ void block_atomic_max(double v, double *red, unsigned long long *dst, int tid, int nthr)
{
    v = wave_max(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        double m = red[0];
        for (int w = 1; w < (nthr + 63) / 64; w++) m = red[w] > m ? red[w] : m;
        atomicMax(dst, (unsigned long long)__double_as_longlong(m));
    }
}

} // namespace plxs

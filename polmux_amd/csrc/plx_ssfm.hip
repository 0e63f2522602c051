// plx_ssfm.hip -- split-step Fourier propagator of fiber.m on gfx950.
//
// Reference: /root/reference/fiber.m:459-935 (matrix_ssfm, scalar_ssfm, nextstep,
// checkstep, lin_step, nl_step, matrix_nl_step, matrix_step).
//
// MI355X design (not a translation of the MATLAB): the field stays in HBM, a step is two or three
// sweeps over it, organised around a four-step FFT  N = N1 x N2  whose forward half is
// decimation-in-frequency and whose inverse half is decimation-in-time, so the spectrum is only
// ever held in (bit-reversed, transposed) order and no reorder pass exists:
//
//   k_col_fwd  load N1 x 16 column tile -> [Kerr step fused on load] -> N1-point
//              DIF in LDS -> store in place
//   k_row      load rows -> x inter-pass twiddle -> N2-point DIF in LDS ->
//              x exp(-i beta dz) / PMD waveplates (both polarisations of one
//              frequency co-resident) -> N2-point DIT -> x conj twiddle -> store
//   k_row256r  the same for 256-point rows of dual-polarisation plans with every radix level in
//              registers: one wave = 2 rows x 2 polarisations, two LDS exchanges, no workgroup
//              barrier, the multiplier shared between the wave's halves (k_row4k: 4096-point rows)
//   k_col_inv  load column tile -> N1-point DIT -> x exp(-alpha dz/2)/N ->
//              wave-shuffle max of |ux|^2+|uy|^2 -> one atomicMax per workgroup
//   k_ctrl     one lane per frame: nextstep + checkstep + last-step rule; the
//              data-dependent step loop never round-trips to the host
//   k_colx16   dual-polarisation plans with 256-row tiles: k_col_inv of step s, the step
//              controller and k_col_fwd of step s+1 in ONE launch on a register-resident
//              tile (the step is then two sweeps), the tiles of a frame meeting at a barrier
//
// Twiddles of the in-LDS transforms are staged in LDS; frames of a batch carry
// their own step state, so a batch of Monte-Carlo realisations advances in
// lock-step launches while every frame keeps the reference's own step sequence.
// (Variants that were measured and lost -- persistent prefetching sweeps, register-blocked rows,
// an LDS-resident fused sweep -- live in the history and in profiles/r01_notes.md, not here.)
#include "../../include/polmux_hip.h"
#include "plx_fft.h"
#include <type_traits>
#include "plx_internal.h"
#include "plx_gateway.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace {

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

// ---------------------------------------------------------------- initial max ---
// Umax of nextstep (fiber.m:693-697) for the field as handed to fiber().
__global__ __launch_bounds__(256) void k_umax(SsfmArgs a)
{
    PLX_DYN_LDS(lds);
    double *red = (double *)lds;
    const int fc = blockIdx.y;
    const size_t N = (size_t)1 << (a.p1 + a.p2);
    const cplx *x = a.ux + (size_t)fc * N;
    const cplx *y = a.dual ? a.uy + (size_t)fc * N : nullptr;
    double m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
        cplx v = x[i];
        double p = v.x * v.x + v.y * v.y;
        if (y) {
            cplx w = y[i];
            p = p + w.x * w.x;
            p = p + w.y * w.y;
        }
        m = p > m ? p : m;
    }
    block_atomic_max(m, red, a.umax + fc, threadIdx.x, blockDim.x);
}

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
__device__ __noinline__ double ctrl_head_call(const PLX_LDS_QUAL CtrlK *k, PLX_LDS_QUAL FrameCtl *rec, int count, double pmax, int f)
{
    if (!ctrl_head<true>(*k, f, *rec, true, pmax, count != 0)) return -1.0;
    return rec->leff;
}
__device__ __noinline__ void ctrl_tail_call(const PLX_LDS_QUAL CtrlK *k, PLX_LDS_QUAL FrameCtl *rec)
{
    ctrl_tail(*k, *rec);
}
template <bool AGENT> __device__ __forceinline__ void ctrl_step(const SsfmArgs &a, int f, bool have_pmax = false, double pmax_in = 0.0)
{
    FrameCtl c = a.ctl[f];
    if (c.done) return;
    ctrl_core<AGENT>(a, f, c, have_pmax, pmax_in);
    a.ctl[f] = c;
}

__global__ void k_ctrl(SsfmArgs a, int nframes)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    ctrl_step<false>(a, f);
}

// The active list: frames that have not reached the fibre end, in frame order (one workgroup; a block scan over
// contiguous chunks of frames).  Runs once per step, in front of the step's sweeps.
// `serves`: the number of steps this list will be used for (utilisation accounting).
#define COMPACT_THREADS 256
__global__ __launch_bounds__(COMPACT_THREADS) void k_compact(const FrameCtl *ctl, int nframes, int *active, int *nactive, int serves)
{
    PLX_DYN_LDS(lds);
    int *cnt = (int *)lds;                         // [COMPACT_THREADS]
    const int tid = threadIdx.x, per = (nframes + COMPACT_THREADS - 1) / COMPACT_THREADS;
    const int f0 = tid * per, f1 = min(nframes, f0 + per);
    int n = 0;
    for (int f = f0; f < f1; f++) n += ctl[f].done ? 0 : 1;
    cnt[tid] = n;
    __syncthreads();
    for (int d = 1; d < COMPACT_THREADS; d <<= 1) { // inclusive scan
        const int v = tid >= d ? cnt[tid - d] : 0;
        __syncthreads();
        cnt[tid] += v;
        __syncthreads();
    }
    int o = cnt[tid] - n;
    for (int f = f0; f < f1; f++)
        if (!ctl[f].done) active[o++] = f;
    if (tid == COMPACT_THREADS - 1) { nactive[0] = cnt[tid]; nactive[1] += cnt[tid] * serves; }
}

// -------------------------------------------- adaptive scheme: element-wise pieces ---
// nl_step (fiber.m:776-804) followed by the attenuation of the half/quarter step (:973,:979,...), scalar fields.
__global__ __launch_bounds__(256) void k_nl_att(cplx *u, const double *gam, size_t N, int nfc, int spm, int xpm,
                                                double leff, double att)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
        double tot = 0;
        if (xpm)
            for (int k = 0; k < nfc; k++) {
                const cplx v = u[(size_t)k * N + i];
                tot += v.x * v.x + v.y * v.y;
            }
        for (int k = 0; k < nfc; k++) {
            cplx v = u[(size_t)k * N + i];
            if (spm || xpm) {
                double pw = v.x * v.x + v.y * v.y;
                if (xpm) pw = spm ? 2 * tot - pw : 2 * (tot - pw);
                v = cmul(v, cexpi(-gam[k] * pw * leff));
            }
            u[(size_t)k * N + i] = cscale(v, att);
        }
    }
}

// est_err numerator max|u-uh| (:997) -> atomicMax on the bit pattern
__global__ __launch_bounds__(256) void k_maxdiff(const cplx *u, const cplx *uh, size_t n, unsigned long long *out)
{
    PLX_DYN_LDS(lds);
    double *red = (double *)lds;
    double m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double dr = u[i].x - uh[i].x, di = u[i].y - uh[i].y;
        const double e = sqrt(dr * dr + di * di);
        m = e > m ? e : m;
    }
    block_atomic_max(m, red, out, threadIdx.x, blockDim.x);
}

// Richardson extrapolation u = 4/3*uh - 1/3*u (:1004)
__global__ __launch_bounds__(256) void k_richardson(cplx *u, const cplx *uh, size_t n)
{
    const double c43 = 4.0 / 3, c13 = 1.0 / 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        u[i] = make_double2(c43 * uh[i].x - c13 * u[i].x, c43 * uh[i].y - c13 * u[i].y);
}

// ------------------------------------------------- scalar XPM row sum (:795) ---
__global__ __launch_bounds__(256) void k_rowsum(SsfmArgs a)
{
    const int f = blockIdx.y;
    if (a.ctl[f].done) return;
    const size_t N = (size_t)1 << (a.p1 + a.p2);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
        double sum = 0;
        for (int k = 0; k < a.nfc; k++) {
            cplx v = a.ux[((size_t)f * a.nfc + k) * N + i];
            sum += v.x * v.x + v.y * v.y;
        }
        a.psum[(size_t)f * N + i] = sum;
    }
}

// ------------------------------------------------------------ pass 1: columns ---
// Kerr step (matrix_nl_step :832-852 / nl_step :792-804) fused into the load of
// the forward column transform.  Tile = N1 rows x T complex (dual: W columns of ux |
// W of uy; scalar: W columns): T*16 B contiguous bytes of LDS per row, so the T
// interleaved transforms are read conflict-free.  Global loads are issued in
// batches of COL_CH per thread before any arithmetic, to keep >= 64 KiB in flight
// per CU.
#define COL_CH 4
#define COL_THREADS_MAX 1024
__global__ __launch_bounds__(COL_THREADS_MAX) void k_col_fwd(SsfmArgs a)
{
    PLX_DYN_LDS(lds);
    if (all_done_or_aborted(a)) return;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int slot = blockIdx.y / a.nfc, c = blockIdx.y - slot * a.nfc;
    int f;
    if (!slot_frame(a, slot, f)) return;
    const int fc = f * a.nfc + c;
    const FrameCtl *ctl = a.ctl + f;
    if (ctl->done) return;
    const int N1 = 1 << a.p1, N2 = 1 << a.p2, T = a.T, W = a.W;
    cplx *s = (cplx *)lds;
    cplx *tw = s + ((size_t)N1 << a.logT);
    lds_load_twiddles(tw, a.tw1, N1 >> 1, tid, nthr);
    const size_t base = (size_t)fc << (a.p1 + a.p2);
    const int col0 = blockIdx.x * W;
    const double leff = a.force ? a.f_leff : ctl->leff;
    const double gam = a.gam[c], gamleff = gam * leff;
    const int nel = N1 << a.logW;
    if (a.dual) {
        for (int e0 = tid; e0 < nel; e0 += nthr * COL_CH) {
            cplx xv[COL_CH], yv[COL_CH];
#pragma unroll
            for (int k = 0; k < COL_CH; k++) {
                const int e = min(e0 + k * nthr, nel - 1);
                const size_t g = base + (size_t)(e >> a.logW) * N2 + col0 + (e & (W - 1));
                xv[k] = a.ux[g]; yv[k] = a.uy[g];
            }
#pragma unroll
            for (int k = 0; k < COL_CH; k++) { pin(xv[k]); pin(yv[k]); }
#pragma unroll
            for (int k = 0; k < COL_CH; k++) {
                const int e = e0 + k * nthr;
                if (e >= nel) continue;
                cplx x = xv[k], y = yv[k];
                if (a.spm) {
                    const double P = x.x * x.x + x.y * x.y + y.x * y.x + y.y * y.y; // :834-835
                    double sn, cs;
                    sincos_small(-gamleff * P, &sn, &cs);                             // :837
                    const cplx nl = make_double2(cs, sn);
                    x = cmul(x, nl);
                    y = cmul(y, nl);
                    if (!a.manakov) { // CNLSE rotation :842-850
                        const double s3 = 2 * (x.x * y.y - x.y * y.x);
                        double sp, cp;
                        sincos_small(div3(gamleff * s3), &sp, &cp);
                        const cplx xx = make_double2(cp * x.x + sp * y.x, cp * x.y + sp * y.y);
                        const cplx yy = make_double2(cp * y.x - sp * x.x, cp * y.y - sp * x.y);
                        x = xx; y = yy;
                    }
                }
                const int row = e >> a.logW, col = e & (W - 1);
                s[(row << a.logT) + col] = x;
                s[(row << a.logT) + W + col] = y;
            }
        }
    } else {
        const bool active = a.spm || a.xpm; // :800-802
        for (int e0 = tid; e0 < nel; e0 += nthr * COL_CH) {
            cplx xv[COL_CH];
            double pv[COL_CH];
#pragma unroll
            for (int k = 0; k < COL_CH; k++) {
                const int e = min(e0 + k * nthr, nel - 1);
                const size_t off = (size_t)(e >> a.logW) * N2 + col0 + (e & (W - 1));
                xv[k] = a.ux[base + off];
                pv[k] = a.xpm ? a.psum[((size_t)f << (a.p1 + a.p2)) + off] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < COL_CH; k++) { pin(xv[k]); pin(pv[k]); }
#pragma unroll
            for (int k = 0; k < COL_CH; k++) {
                const int e = e0 + k * nthr;
                if (e >= nel) continue;
                cplx x = xv[k];
                if (active) {
                    double pw = x.x * x.x + x.y * x.y; // :792
                    if (a.xpm) pw = a.spm ? 2 * pv[k] - pw : 2 * (pv[k] - pw); // :795,797
                    double sn, cs;
                    sincos_small(-gam * pw * leff, &sn, &cs); // :804
                    x = cmul(x, make_double2(cs, sn));
                }
                s[((e >> a.logW) << a.logT) + (e & (W - 1))] = x;
            }
        }
    }
    __syncthreads();
    lds_fft_dif(s, a.p1, T, 1, a.logT, tw, tid, nthr, true);
    for (int e = tid; e < nel; e += nthr) {
        const int row = e >> a.logW, col = e & (W - 1);
        const size_t g = base + (size_t)row * N2 + col0 + col;
        a.ux[g] = s[(row << a.logT) + col];
        if (a.dual) a.uy[g] = s[(row << a.logT) + W + col];
    }
}

// The waveplate loop of matrix_step (fiber.m:907-933) at one frequency.  Per trunk the reference applies
//   u <- R diag(e^{-i(a+b)}, e^{-i(a-b)}) R^H u,   a = betat*dzb(k),  b = (db1 + db0(n))/2 * dzb(k)/lcorr.
// With diag(e^{-ib}, e^{ib}) = cos b I - i sin b sigma3 this is  e^{-ia} (cos b u - i sin b S u),  S = R sigma3 R^H
// (Hermitian, traceless, frequency independent: S11 real, S12 complex -- formed on the host per waveplate), and the
// scalar factors e^{-ia} of all trunks multiply to e^{-i betat * dz}: one exponential per trunk instead of two, 20
// multiply-adds instead of 40.  BRF_STRIDE doubles per plate: S11, Re S12, Im S12, db0 (turns).
#define BRF_STRIDE 4
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

// Row and column phasors of every trunk of the step each listed frame is about to make (see SsfmArgs::e1tab).  One workgroup
// per frame; runs between the step controller (k_ctrl / the fused sweep, which leave ntrunk, dzb_first, dzb_last in the
// frame's record) and the row pass.
__global__ __launch_bounds__(256) void k_pmd_tab(SsfmArgs a)
{
    if (all_done_or_aborted(a)) return;
    int f;
    if (!slot_frame(a, blockIdx.x, f)) return;
    const FrameCtl *ctl = a.ctl + f;
    if (ctl->done) return;
    const int N1 = 1 << a.p1, N2 = 1 << a.p2;
    const int ntrunk = ctl->ntrunk, n0 = ctl->ntot - ctl->nmem;
    if (ntrunk > a.tmax) return;                  // (the row pass then takes the general loop for this frame)
    const double *brf = a.brf + (a.brf_per_frame ? (size_t)f * a.nplates * BRF_STRIDE : 0);
    const double lcorr = a.lcorr, rl = 1.0 / lcorr;
    cplx *e1 = a.e1tab + (size_t)f * a.tmax * N1, *e2 = a.e2tab + (size_t)f * a.tmax * N2;
    for (int k = 1; k <= ntrunk; k++) {
        int plate = n0 + k - 1;
        plate = plate < 0 ? 0 : (plate >= a.nplates ? a.nplates - 1 : plate);
        const double db0 = brf[(size_t)plate * BRF_STRIDE + 3];
        const double dzk = (k == 1) ? ctl->dzb_first : (k == ntrunk ? ctl->dzb_last : lcorr);
        // A = 0.5 D dzk / lcorr, B = 0.5 db0 dzk / lcorr (the quotient formed as in pmd_trunks)
        const double na = 0.5 * a.d1slope * dzk, qa = na * rl, A = fma(fma(-lcorr, qa, na), rl, qa);
        const double nb = 0.5 * db0 * dzk, qb = nb * rl, B = fma(fma(-lcorr, qb, nb), rl, qb);
        for (int j = threadIdx.x; j < N1; j += blockDim.x)
            e1[(size_t)(k - 1) * N1 + j] = cexp_neg_turns(fma(A, (double)plx_bitrev((unsigned)j, a.p1), B));
        for (int i = threadIdx.x; i < N2; i += blockDim.x) {
            const int k2 = (int)plx_bitrev((unsigned)i, a.p2), m2 = k2 >= (N2 >> 1) ? k2 - N2 : k2;
            e2[(size_t)(k - 1) * N2 + i] = cexp_neg_turns(A * (double)((long long)N1 * m2));
        }
    }
}

// --------------------------------------------------------------- pass 2: rows ---
// Second half of the forward transform, the linear operator of the step
// (lin_step :771-773 / matrix_step :907-933) and the first half of the inverse
// transform, all on one LDS-resident row set (padded layout, see plx_fft.h).
#define ROW_THREADS 128
#define ROW_CH 4
__global__ __launch_bounds__(1024) void k_row(SsfmArgs a)
{
    PLX_DYN_LDS(lds);
    if (all_done_or_aborted(a)) return;
    const int tid = threadIdx.x, nthr = blockDim.x;
    int slot = blockIdx.y / a.nfc;
    const int c = blockIdx.y - slot * a.nfc;
    if (!row_slot(a, slot)) return;
    int f;
    if (!slot_frame(a, slot, f)) return;
    const int fc = f * a.nfc + c;
    const FrameCtl *ctl = a.ctl + f;
    if (ctl->done) return;
    const int N2 = 1 << a.p2, R = a.R, TSp = row_pitch(N2);
    cplx *s = (cplx *)lds;                       // [npol*R][TSp]
    cplx *tw = s + (size_t)(a.dual ? 2 : 1) * R * TSp;
    // the half table of W_N2: with at most one entry per thread (the usual shape) the load is issued HERE and lands in LDS
    // after the tile's loads have been issued -- one memory round trip for both instead of two in a row
    const int twn = N2 >> 1;
    const bool tw_one = twn <= nthr;
    cplx twv = make_double2(0, 0);
    if (tw_one) { if (tid < twn) twv = a.tw2[tid]; }
    else lds_load_twiddles(tw, a.tw2, twn, tid, nthr);
    const size_t N = (size_t)1 << (a.p1 + a.p2);
    const int j0 = blockIdx.x * R;
    const int nel = R << a.p2;
    // the field (rows N2 apart: the R rows of this workgroup are contiguous).  gofs(e): where element e = (row, point) lives.
    cplx *const fx = a.ux;
    cplx *const fy = a.dual ? a.uy : a.ux;
    const size_t base = (size_t)fc * N + (size_t)j0 * N2;
    auto gofs = [&](int e) -> size_t { return base + e; };
    const size_t rowbase = (size_t)j0 * N2;      // (position of the workgroup's rows in the per-frequency tables)
    // the inter-pass twiddles of a thread's points: with exactly ROW_CH points per thread (the usual shape) they stay
    // in registers for the conjugate multiply on the way out
    const bool keep_tw = nel == nthr * ROW_CH;
    cplx tkeep[ROW_CH];
    // ... and so does beta of its ROW_CH bins when the multiplier is the step's plain exp(-i beta dz): the (L2-resident)
    // table reads travel with the tile loads instead of sitting, one after the other, between the two transforms
    const double *bt = a.betat_p + (size_t)c * N + rowbase;
    const bool fastmul = keep_tw && a.dual && !a.pmd && !a.umat && !a.hmul;
    double btk[ROW_CH];
    if (fastmul) {
#pragma unroll
        for (int k = 0; k < ROW_CH; k++) btk[k] = bt[row_lane_point(tid + k * nthr, N2)];
    }
    for (int e0 = tid; e0 < nel; e0 += nthr * ROW_CH) {
        cplx tv[ROW_CH], xv[ROW_CH], yv[ROW_CH];
#pragma unroll
        for (int k = 0; k < ROW_CH; k++) {
            const int e = row_lane_point(min(e0 + k * nthr, nel - 1), N2);   // (the lane's point: rotated within blocks of 16, see plx_fft.h)
            tv[k] = a.tpass[rowbase + e];
            xv[k] = fx[gofs(e)];
            yv[k] = fy[gofs(e)];
        }
#pragma unroll
        for (int k = 0; k < ROW_CH; k++) { pin(tv[k]); pin(xv[k]); pin(yv[k]); }
#pragma unroll
        for (int k = 0; k < ROW_CH; k++) {
            const int el = e0 + k * nthr, e = row_lane_point(el, N2);
            tkeep[k] = tv[k];
            if (el < nel) {
                const int r = e >> a.p2, i = e & (N2 - 1);
                s[r * TSp + row_phys(i)] = cmul(xv[k], tv[k]);
                if (a.dual) s[(R + r) * TSp + row_phys(i)] = cmul(yv[k], tv[k]);
            }
        }
    }
    if (tw_one && tid < twn) tw[tid] = twv;
    __syncthreads();
    row_fft_dif(s, a.p2, a.logR + (a.dual ? 1 : 0), tw, tid, nthr);
    const double cur = a.force ? a.f_cur : ctl->cur;
    if (!a.dual) {
        for (int el = tid; el < nel; el += nthr) {
            const int e = row_lane_point(el, N2); // Hf = fastexp(-betat*dz) :771
            const int o = (e >> a.p2) * TSp + row_phys(e & (N2 - 1));
            s[o] = cmul(s[o], a.hmul ? a.hmul[rowbase + e] : cexp_neg_turns(bt[e] * cur));
        }
    } else if (!a.pmd) {
        // zero birefringence, one trunk (fiber.m:291-297): matR = I, deltabeta = 0
        // (one loop per kind of multiplier: the step's exp(-i beta dz) loop stays a single basic block)
        if (a.umat) { // Uinv = conj(Hgvd) [conj(U11) -U12; conj(U12) U11] applied to [x; y]  (inverse_pmd.m:130-141)
            for (int el = tid; el < nel; el += nthr) {
            const int e = row_lane_point(el, N2);
                const int o = (e >> a.p2) * TSp + row_phys(e & (N2 - 1));
                const cplx *um = a.umat + 3 * ((size_t)f * N + rowbase + e);
                const cplx u11 = um[0], u12 = um[1], hg = um[2], x = s[o], y = s[o + R * TSp];
                s[o] = cmulc(csub(cmulc(x, u11), cmul(u12, y)), hg);
                s[o + R * TSp] = cmulc(cadd(cmulc(x, u12), cmul(u11, y)), hg);
            }
        } else if (a.hmul) {
            for (int el = tid; el < nel; el += nthr) {
            const int e = row_lane_point(el, N2);
                const int o = (e >> a.p2) * TSp + row_phys(e & (N2 - 1));
                const cplx h = a.hmul[rowbase + e];
                s[o] = cmul(h, s[o]);
                s[o + R * TSp] = cmul(h, s[o + R * TSp]);
            }
        } else if (fastmul) {
#pragma unroll
            for (int k = 0; k < ROW_CH; k++) {
                const int e = row_lane_point(tid + k * nthr, N2);
                const int o = (e >> a.p2) * TSp + row_phys(e & (N2 - 1));
                const cplx h = cexp_neg_turns(btk[k] * cur);
                s[o] = cmul(h, s[o]);
                s[o + R * TSp] = cmul(h, s[o + R * TSp]);
            }
        } else {
            for (int el = tid; el < nel; el += nthr) {
            const int e = row_lane_point(el, N2);
                const int o = (e >> a.p2) * TSp + row_phys(e & (N2 - 1));
                const cplx h = cexp_neg_turns(bt[e] * cur);
                s[o] = cmul(h, s[o]);
                s[o + R * TSp] = cmul(h, s[o + R * TSp]);
            }
        }
    } else {
        const double *d1 = a.db1_p + (size_t)c * N + rowbase;
        const double *brf = a.brf + (a.brf_per_frame ? (size_t)f * a.nplates * BRF_STRIDE : 0);
        const int ntrunk = ctl->ntrunk, n0 = ctl->ntot - ctl->nmem; // plate of piece k: n0+k (1-based) :908
        const double dzb_first = ctl->dzb_first, dzb_last = ctl->dzb_last, lcorr = a.lcorr;
        if (a.e1tab && ntrunk <= a.tmax) {       // trunk phasors from the tables of k_pmd_tab (db1 linear in the frequency index)
            const int N1 = 1 << a.p1;
            const cplx *e1 = a.e1tab + (size_t)f * a.tmax * N1 + j0, *e2 = a.e2tab + (size_t)f * a.tmax * N2;
            for (int el = tid; el < nel; el += nthr) {
                const int e = row_lane_point(el, N2);
                const int r = e >> a.p2, i = e & (N2 - 1);
                const int o = r * TSp + row_phys(i);
                cplx x = s[o], y = s[o + R * TSp];
                pmd_trunks_tab(x, y, bt[e], e1 + r, N1, e2 + i, N2, brf, a.nplates, n0, ntrunk, cur);
                s[o] = x;
                s[o + R * TSp] = y;
            }
        } else
        for (int el = tid; el < nel; el += nthr) {
            const int e = row_lane_point(el, N2);
            const int o = (e >> a.p2) * TSp + row_phys(e & (N2 - 1));
            cplx x = s[o], y = s[o + R * TSp];
            pmd_trunks(x, y, bt[e], d1[e], brf, a.nplates, n0, ntrunk, dzb_first, dzb_last, lcorr, cur);
            s[o] = x;
            s[o + R * TSp] = y;
        }
    }
    __syncthreads();
    row_fft_dit(s, a.p2, a.logR + (a.dual ? 1 : 0), tw, tid, nthr);
    if (keep_tw) {
#pragma unroll
        for (int k = 0; k < ROW_CH; k++) {
            const int e = row_lane_point(tid + k * nthr, N2);
            const int o = (e >> a.p2) * TSp + row_phys(e & (N2 - 1));
            fx[gofs(e)] = cmulc(s[o], tkeep[k]);
            if (a.dual) fy[gofs(e)] = cmulc(s[o + R * TSp], tkeep[k]);
        }
        return;
    }
    for (int el = tid; el < nel; el += nthr) {
            const int e = row_lane_point(el, N2);
        const int o = (e >> a.p2) * TSp + row_phys(e & (N2 - 1));
        const cplx t = a.tpass[rowbase + e];
        fx[gofs(e)] = cmulc(s[o], t);
        if (a.dual) fy[gofs(e)] = cmulc(s[o + R * TSp], t);
    }
}

// ------------------------------------------------- pass 2 for 256-point rows, register form ---
// The headline shape (dual polarisation, no PMD, 256-point rows, the step's exp(-i beta dz)) with every radix level in
// registers, as k_row4k does for 4096-point rows: ONE WAVE = one tile of 2 rows x 2 polarisations, 16 lanes per row, lane j
// of a row holds its points j + 16 k.  Forward: lvl2_dif256 -> one exchange through the padded LDS row -> r16_dif; the
// multiplier on the 16 bins the lane then holds (bit-reversed order, where the tables are); the inverse mirrors it.  Two LDS
// exchanges per transform pair end (4 x 16 ds_write_b128 + 4 x 16 ds_read_b128 per lane and tile) where k_row's eleven passes make 64 + 96 per
// lane on half as many points, no workgroup barrier at all (a one-wave workgroup only waits for its own LDS operations), and
// the butterflies are k_row's, stage by stage.  Inter-pass twiddles as in k_row4k: tpass[j + 16 k] = tpass[j] * tpass[16 k],
// a lane reads one entry and the row's sixteen lanes share sixteen (bk).
#define ROWR_THREADS 64
#define ROWR_LDS ((4 * 272 + 128 + 48) * sizeof(cplx))        // 20224 B: eight one-wave workgroups per CU
#define ROWR_LDS_SC ((4 * 272 + 128 + 80) * sizeof(cplx))     // scalar plans: four rows' bk entries (seven workgroups per CU)
#ifdef PLX_EMU
#define ROWR_SYNC() __syncthreads()
#else
#define ROWR_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
// PMD: the waveplate trunks of matrix_step (fiber.m:907-933) need both polarisations of a bin in one lane: the halves of the
// wave trade (half_trade) so that every lane holds ux and uy of eight bins, run pmd_trunks / pmd_trunks_tab on them -- the
// arithmetic of k_row's PMD branch, bin by bin -- and trade back.
// SC (scalar plans, 2^16-sample frames of scalar_ssfm): the wave's four lane groups are four rows of the one field.
// SPLIT (the PMD form with phasor tables, i.e. a linear db1: the plan knows): three waves per SIMD -- the two exchanges through
// LDS in real / imaginary halves (11.5 KiB per one-wave workgroup instead of 20), the phases asked for behind the trunk loop,
// the column phasors fetched two bins at a time, and ONLY the table form of the trunk loop in the kernel (the general form, one
// exponential per bin and trunk, stays in k_row256r<true>): 168 registers.
template <bool PMD, bool SC = false, bool SPLIT = false> __global__ __launch_bounds__(ROWR_THREADS, SPLIT ? 3 : 2) void k_row256r(SsfmArgs a)
{
    static_assert(!(PMD && SC), "PMD needs two polarisations");
    static_assert(!SPLIT || PMD, "the split form is the PMD kernel's");
    PLX_DYN_LDS(lds);
    if (all_done_or_aborted(a)) return;
    const int tid = threadIdx.x;
    int slot = blockIdx.y / a.nfc;
    const int c = blockIdx.y - slot * a.nfc;       // (channels of a frame: 'sepfields' WDM)
    if (!row_slot(a, slot)) return;
    int f;
    if (!slot_frame(a, slot, f)) return;
    const int fc = f * a.nfc + c;
    const FrameCtl *ctl = a.ctl + f;
    if (ctl->done) return;
    cplx *const s = (cplx *)lds + (tid >> 4) * 272;      // this lane group's padded row: physical(p) = p + (p >> 4)
    double *const sd = (double *)lds + (tid >> 4) * 272; // SPLIT: the same row, one component at a time
    cplx *const tw = SPLIT ? (cplx *)((double *)lds + 4 * 272) : (cplx *)lds + 4 * 272;              // W_256^k, k < 128
    cplx *const bk = tw + 128 + 17 * (SC ? tid >> 4 : (tid >> 4) & 1);   // tpass[row][16 k], k < 16 (the rows' entries on different banks)
    const Tw256half w8{tw};
    const int j = tid & 15, r = SC ? tid >> 4 : (tid >> 4) & 1;
    const size_t N = (size_t)1 << 16;
    const size_t rowbase = ((size_t)blockIdx.x * (SC ? 4 : 2) + r) << 8;
    cplx *const u = (!SC && tid >= 32 ? a.uy : a.ux) + (size_t)fc * N + rowbase;
    const cplx *const tp = a.tpass + rowbase;
    cplx x[16];
    // the second register stage's twiddles W_256^{4j}, W^{8j}, W^{12j}: from the table in memory into registers, once -- out of
    // LDS the sixteen lanes of a transform would fetch them from the same banks (strides of 4, 8 and 12 entries)
    cplx v1, v2, v3;
    {
        const cplx ta = tp[j];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = u[j + 16 * k];
        if (!PMD) { v1 = a.tw2[4 * j]; v2 = a.tw2[8 * j]; v3 = tw3(a.tw2, 12 * j, 128); }     // (PMD: the trunk loop needs the registers, LDS serves)
        {
            const cplx t0 = a.tw2[tid], t1 = a.tw2[tid + 64], t3 = tp[16 * j];
            tw[tid] = t0; tw[tid + 64] = t1;
            if (SC || tid < 32) bk[j] = t3;
        }
        ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = cmul(x[k], cmul(ta, bk[k]));
    }
    sched_fence();
    if (PMD) lvl2_dif<16>(x, j, w8); else lvl2_dif<16>(x, j, w8, v1, v2, v3);
    sched_fence();
    if (SPLIT) {                           // row_phys(j + 16 k) = j + 17 k, row_phys(16 j + k) = 17 j + k
#pragma unroll
        for (int k = 0; k < 16; k++) sd[j + 17 * k] = x[k].x;
        ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k].x = sd[17 * j + k];
        ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) sd[j + 17 * k] = x[k].y;
        ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k].y = sd[17 * j + k];
        ROWR_SYNC();
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) s[row_phys(j + 16 * k)] = x[k];
        ROWR_SYNC();
    }
    // the multiplier is the same for the two polarisations of a bin, which sit in lanes i and i + 32: the lower half of the
    // wave forms it for the lane's bins 0-7, the upper half for bins 8-15, and they swap (half_share)
    double btv[SC ? 16 : 8];
    if (!SPLIT) {
        const double *bt = a.betat_p + (size_t)c * N + rowbase + 16 * j + (!SC && tid >= 32 ? 8 : 0);
#pragma unroll
        for (int k = 0; k < (SC ? 16 : 8); k++) btv[k] = bt[k];
    }
    if (!SPLIT) {
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[17 * j + k];           // row_phys(16 j + k)
    }
    r16_dif(x);
    sched_fence();
    if (PMD) {
        const double cur = ctl->cur;
        const int ib = 16 * j + (tid >= 32 ? 8 : 0);            // the first of this lane's eight bins within the row
        const size_t rowf = (size_t)blockIdx.x * 2 + r;
        const double *brf = a.brf + (a.brf_per_frame ? (size_t)f * a.nplates * BRF_STRIDE : 0);
        const int ntrunk = ctl->ntrunk, n0 = ctl->ntot - ctl->nmem;
#pragma unroll
        for (int k = 0; k < 8; k++) half_trade(x[k], x[k + 8]);   // x[k] = ux, x[k + 8] = uy of bin ib + k
        if (SPLIT || (a.e1tab && ntrunk <= a.tmax)) {       // (SPLIT: launched for plans with tables only; ntrunk <= tmax by the plan's bound on dz)
            const cplx *e1 = a.e1tab + (size_t)f * a.tmax * 256 + rowf, *e2 = a.e2tab + (size_t)f * a.tmax * 256 + ib;
            // (pmd_trunks_tab with the trunk loop outside the bins: a trunk's plate and row phasor are fetched once)
            for (int t = 0; t < ntrunk; t++) {
                int plate = n0 + t;
                plate = plate < 0 ? 0 : (plate >= a.nplates ? a.nplates - 1 : plate);
                const double *m = brf + (size_t)plate * BRF_STRIDE;
                const double s11 = m[0];
                const cplx s12 = make_double2(m[1], m[2]);
                const cplx e1v = e1[(size_t)t * 256];
                const cplx *e2t = e2 + (size_t)t * 256;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const cplx e = cmul(e1v, e2t[k]);
                    const cplx u = x[k], v = x[k + 8];
                    const cplx sx = cadd(cscale(u, s11), cmul(s12, v));
                    const cplx sy = csub(cmulc(u, s12), cscale(v, s11));
                    x[k] = make_double2(e.x * u.x - e.y * sx.y, e.x * u.y + e.y * sx.x);
                    x[k + 8] = make_double2(e.x * v.x - e.y * sy.y, e.x * v.y + e.y * sy.x);
                    if (SPLIT && (k & 1)) sched_fence();
                }
            }
            if (SPLIT) {
                int o = ib;
                pin(o);
                const double *bt = a.betat_p + (size_t)c * N + rowbase + o;
#pragma unroll
                for (int k = 0; k < 8; k++) btv[k] = bt[k];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const cplx h = cexp_neg_turns(btv[k] * cur);
                x[k] = cmul(h, x[k]);
                x[k + 8] = cmul(h, x[k + 8]);
                sched_fence();
            }
        } else if (!SPLIT) {
            const double *d1 = a.db1_p + (size_t)c * N + rowbase + ib;
            const double dzb_first = ctl->dzb_first, dzb_last = ctl->dzb_last;
            for (int k = 0; k < 8; k++) pmd_trunks(x[k], x[k + 8], btv[k], d1[k], brf, a.nplates, n0, ntrunk, dzb_first, dzb_last, a.lcorr, cur);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) half_trade(x[k], x[k + 8]);
    } else if (SC) {
        const double cur = ctl->cur;
#pragma unroll
        for (int k = 0; k < (SC ? 16 : 8); k++) {
            x[k] = cmul(cexp_neg_turns(btv[k] * cur), x[k]);
            sched_fence();
        }
    } else {
        const double cur = ctl->cur;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const cplx h = cexp_neg_turns(btv[k] * cur);
            cplx ha, hb;
            half_share(h, ha, hb);
            x[k] = cmul(ha, x[k]);
            x[k + 8] = cmul(hb, x[k + 8]);
            sched_fence();
        }
    }
    sched_fence();
    r16_dit(x);
    sched_fence();
    if (!SPLIT) {
#pragma unroll
        for (int k = 0; k < 16; k++) s[17 * j + k] = x[k];
        ROWR_SYNC();
    }
    const cplx tb = tp[j];
    if (SPLIT) {
#pragma unroll
        for (int k = 0; k < 16; k++) sd[17 * j + k] = x[k].x;
        ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k].x = sd[j + 17 * k];
        ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) sd[17 * j + k] = x[k].y;
        ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k].y = sd[j + 17 * k];
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[row_phys(j + 16 * k)];
    }
    sched_fence();
    if (PMD) lvl2_dit<16>(x, j, w8); else lvl2_dit<16>(x, j, w8, v1, v2, v3);
    sched_fence();
#pragma unroll
    for (int k = 0; k < 16; k++) u[j + 16 * k] = cmulc(x[k], cmul(tb, bk[k]));
}

// ------------------------------------------------- pass 2 for 4096-point rows (2^20-sample frames) ---
// One workgroup = one row of ONE polarisation (no PMD: the polarisations only share the multiplier), every radix level
// in registers: 4096 = 16 x 16 x 16, thread j holds points j + 256 k, three register levels per direction (lvl2_dif<256>
// on W_4096, lvl2_dif<16> on W_256, r16_dif; the inverse mirrors them) with ONE exchange through a padded LDS row between
// consecutive levels (a thread writes a level's result back where it read its input, so one barrier per exchange) -- four
// exchanges per row where the LDS-resident k_row makes eleven barrier-separated passes.  The
// spectrum is left in the bit-reversed order of the in-place transform, where the multiplier tables already are.
// Twiddles: the compact table of W_4096 (8 KiB); 78 KiB of LDS per workgroup: two per CU.
// Inter-pass twiddles: the row's table is a geometric sequence, tpass[i] = w^i (w = W_N^k1 of the row), so
// tpass[tid + 256 k] = tpass[tid] * tpass[256 k]: a thread reads ONE entry and the workgroup shares sixteen (bk, in LDS)
// instead of 16 entries per thread at either end of the kernel -- 128 KiB less through the L2 per 64-KiB row, for two more
// complex products per point.
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

// The two MIDDLE exchanges (level 2 <-> level 3) stay inside a block of 256 points = 16 consecutive threads = one wave: they
// wait for the wave's own LDS operations only (ROWR_SYNC), the workgroup meets at the two outer exchanges.  Level 2's twiddles
// W_256^e come from a copy of their own, t8[e + (e >> 2)] (e < 128): its lanes ask for e = 4 j2, 8 j2, 12 j2 (+ 16 r1 ...),
// which in the compact W_4096 table are strides of 16, 32 and 48 entries -- every lane of a 16-lane group on the same banks
// (41 % of the kernel's LDS cycles were bank conflicts, profiles/r03_pmc_2pow20.txt); the values are the same table entries.
struct Tw256pad {
    const cplx *t;
    __device__ __forceinline__ cplx operator()(int e) const
    {
        const int i = e & 127;
        const cplx w = t[i + (i >> 2)];
        const bool neg = e >= 128;
        return make_double2(neg ? -w.x : w.x, neg ? -w.y : w.y);
    }
};
// PAIR (PMD plans, and the matrix tables of inverse_pmd): the waveplate trunks of matrix_step (fiber.m:907-933) need both
// polarisations of a bin in one lane.  One workgroup of 512 threads then holds the row of BOTH polarisations (two padded rows,
// 148 KiB: one workgroup per CU, the same eight waves): lanes 0-31 of a wave are 32 threads of the X row, lanes 32-63 the
// same 32 threads of the Y row, so that the holders of a bin's two polarisations are lanes i and i + 32 of one wave and trade
// halves (half_trade: v_permlane32_swap, no LDS, no barrier) around the multiplier exactly as k_row256r<PMD> does.  Everything
// else -- the three register levels, the four exchanges, the blocks of sixteen threads that stay inside a wave -- is the
// one-polarisation kernel with the thread's index within its row (tj) in the place of tid.
// SPLIT (the one-polarisation form): the four exchanges in real / imaginary halves, as in k_rowreg -- a 34 KiB padded row, three
// workgroups per CU (twelve waves) at <= 168 registers; the two outer exchanges then meet at three workgroup barriers each.
template <bool PAIR, bool SPLIT = false> __global__ __launch_bounds__(PAIR ? 512 : 256, PAIR ? 1 : (SPLIT ? 3 : 2)) void k_row4k(SsfmArgs a)
{
    static_assert(!(PAIR && SPLIT), "the split exchange belongs to the one-polarisation form");
    PLX_DYN_LDS(lds);
    if (all_done_or_aborted(a)) return;
    const int tid = PAIR ? (int)((threadIdx.x >> 6) * 32u + (threadIdx.x & 31u)) : (int)threadIdx.x;   // the thread's index within its row
    const int ts = threadIdx.x;                  // (staging of the shared tables: the workgroup's first 256 threads)
    // Workgroup -> (row, frame-channel, polarisation).  The users of a row's tables (betat: 32 KiB, tpass: 64 KiB per row, the
    // same for every frame and both polarisations) are dealt to ONE XCD -- workgroups 8 apart under the round-robin dealing --
    // and next to each other in time: id = 8 K g + 8 k + c with row = 8 g + c and k = 2 (frame-channel) + polarisation < K,
    // so the tables come out of that XCD's L2 for all but the first of a row's K workgroups (they were re-read from HBM a
    // quarter of the time under the (row, frame, polarisation) grid: 75.5 B per sample, profiles/r03_traffic.json).
    // Measured (profiles/r03_row4k_map_ab.txt): FETCH_SIZE 3.57e5 -> 2.68e5 KB per 16-frame launch, 319 -> 314 us; no change at 64
    // frames; but 8 frames (config[4]'s ladder) run 3 - 5 % SLOWER that way, so batches under 16 frames keep the plain order
    // (row fastest, then frame-channel, then polarisation).
    // (PAIR: k = the frame-channel, both polarisations in the workgroup)
    const int K = (int)(gridDim.x >> a.p1), lg = a.p1 < 3 ? a.p1 : 3, G = 1 << lg;     // (G = 8 rows to a group; fewer rows: all of them)
    // (a scalar plan -- no second field -- launches one workgroup per row and frame-channel as well)
    const bool unit_fc = PAIR || a.uy == nullptr;
    int brow, by, bpol;
    if (K >= (unit_fc ? 16 : 32)) {
        const int g8 = (int)blockIdx.x / (G * K), rem = (int)blockIdx.x - g8 * G * K, bk2 = rem >> lg;
        brow = g8 * G + (rem & (G - 1)); by = unit_fc ? bk2 : bk2 >> 1; bpol = unit_fc ? 0 : bk2 & 1;
    } else if (unit_fc) {
        const int N1 = 1 << a.p1;
        brow = (int)blockIdx.x & (N1 - 1); by = (int)blockIdx.x >> a.p1; bpol = 0;
    } else {
        const int N1 = 1 << a.p1, q = (int)blockIdx.x >> a.p1, FCn = K >> 1;
        brow = (int)blockIdx.x & (N1 - 1); bpol = q / FCn; by = q - bpol * FCn;
    }
    if (PAIR) bpol = (int)((threadIdx.x >> 5) & 1u);
    int slot = by / a.nfc;
    const int c = by - slot * a.nfc;
    if (!row_slot(a, slot)) return;
    int f;
    if (!slot_frame(a, slot, f)) return;
    const int fc = f * a.nfc + c;
    const FrameCtl *ctl = a.ctl + f;
    if (ctl->done) return;
    cplx *s = (cplx *)lds + (PAIR ? bpol * 4352 : 0);    // [4352] padded row: physical(p) = p + (p >> 4)
    double *const sd = (double *)lds;            // SPLIT: the padded row, one component at a time
    cplx *tw = SPLIT ? (cplx *)((double *)lds + 4352) : (cplx *)lds + (PAIR ? 2 : 1) * 4352;      // W_4096^{4k}, k < 512, then W_4096^0..3
    cplx *bk = tw + 516;                         // tpass[256 k], k < 16
    cplx *t8 = bk + 16;                          // W_256^e at e + (e >> 2), e < 128
    cplx *ct = t8 + 160;                         // the unit circle in 64 steps (cexp_neg_turns_tab)
    const size_t N = (size_t)1 << (a.p1 + a.p2);
    const size_t rowbase = (size_t)brow << 12;
    cplx *const u = (bpol ? a.uy : a.ux) + (size_t)fc * N + rowbase;
    const cplx *const tp = a.tpass + rowbase;

    const Tw4096 w1{tw};
    const Tw256pad w2{t8};
    const int b = tid >> 4, j2 = tid & 15;       // level 2: block b of 256 points, point j2 + 16 k of it
    cplx x[16];
    // SPLIT: one exchange in two halves (x[k] to slot wi(k), the thread's next sixteen values from slot ri(k); the real parts land
    // in x[k].x while x[k].y still holds the old imaginary parts).  The padded slots in closed form: row_phys(tid + 256 k) =
    // tid + (tid >> 4) + 272 k, row_phys(256 b + j2 + 16 k) = 272 b + j2 + 17 k, row_phys(16 tid + k) = 17 tid + k.
    // outer: the partners are the whole workgroup (barriers); else lanes of this wave.  No barrier behind the last read: a slot a
    // thread reads in one exchange is written next by that thread itself, or after a later barrier.
    auto exchange_split = [&](auto wi, auto ri, bool outer) {
#pragma unroll
        for (int k = 0; k < 16; k++) sd[wi(k)] = x[k].x;
        if (outer) __syncthreads(); else ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k].x = sd[ri(k)];
        if (outer) __syncthreads(); else ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) sd[wi(k)] = x[k].y;
        if (outer) __syncthreads(); else ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k].y = sd[ri(k)];
        if (!outer) ROWR_SYNC();
    };
    const int obase = tid + (tid >> 4), cbase = 272 * b + j2, tbase = 17 * tid;
    const auto outerp = [&](int k) { return obase + 272 * k; };
    const auto chunkp = [&](int k) { return cbase + 17 * k; };
    const auto own16p = [&](int k) { return tbase + k; };
    {
        cplx ta = tp[tid];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = u[tid + 256 * k];
        // (the row is asked for first: the tables, a few KiB out of the L2, arrive behind it under the same wait)
        if (!PAIR || ts < 256) {
            const cplx t0 = a.tw2[ts], t1 = a.tw2[ts + 256], t2 = a.tw2[512 + (ts & 3)], t3 = tp[256 * (ts & 15)], t4 = a.tw2[4 * (ts & 127)];
            tw[ts] = t0; tw[ts + 256] = t1;
            if (ts < 4) tw[512 + ts] = t2;
            if (ts < 16) bk[ts] = t3;
            if (ts < 128) t8[ts + (ts >> 2)] = t4;
            if (ts < PLX_CTAB) ct[ts] = a.ctab[ts];
        }
#pragma unroll
        for (int k = 0; k < 16; k++) pin(x[k]);
        pin(ta);
        __syncthreads();                         // twiddles and bk staged
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = cmul(x[k], cmul(ta, bk[k]));
    }
    lvl2_dif<256>(x, tid, w1);
    if (SPLIT) exchange_split(outerp, chunkp, true);
    else {
#pragma unroll
        for (int k = 0; k < 16; k++) s[row_phys(tid + 256 * k)] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[row_phys(256 * b + j2 + 16 * k)];
    }
    lvl2_dif<16>(x, j2, w2);                     // (written back where this thread read it: no barrier in between)
    if (SPLIT) exchange_split(chunkp, own16p, false);
    else {
#pragma unroll
        for (int k = 0; k < 16; k++) s[row_phys(256 * b + j2 + 16 * k)] = x[k];
        ROWR_SYNC();                             // (the block's sixteen threads are lanes of one wave)
    }
    // the step's multiplier at the 16 bins this thread holds (lin_step :771-773 / matrix_step with matR = I): the phases are
    // asked for HERE, one exchange and one register level ahead of their use (16 more registers fit beside r16_dif)
    // (PAIR: the eight bins whose two polarisations the lane holds after the trade -- the lower half of the wave the thread's
    //  bins 0-7, the upper half bins 8-15)
    const int ib = 16 * tid + (PAIR && bpol ? 8 : 0);
    double btv[PAIR ? 8 : 16];
    if (!SPLIT && !a.hmul && !(PAIR && a.umat)) {      // (SPLIT: asked for in two halves at the multiplier, see k_rowreg)
        const double *bt = a.betat_p + (size_t)c * N + rowbase + ib;
#pragma unroll
        for (int k = 0; k < (PAIR ? 8 : 16); k++) btv[k] = bt[k];
    }
    if (!SPLIT) {
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[17 * tid + k];         // row_phys(16 tid + k)
    }
    r16_dif(x);
    if (PAIR && !a.hmul) {
        pair_multiplier<12>(a, x, btv, ct, ctl, f, c, brow, rowbase, ib);
    } else {
        if (a.hmul) {
            int o16 = 16 * tid;
            pin(o16);
            const cplx *h = a.hmul + rowbase + o16;
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = cmul(h[k], x[k]);
        } else {
            const double cur = a.force ? a.f_cur : ctl->cur;
            if (SPLIT) {
#pragma unroll
                for (int h = 0; h < 16; h += 8) {
                    int o = ib + h;
                    pin(o);
                    const double *bt = a.betat_p + (size_t)c * N + rowbase + o;
                    double bh[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) bh[k] = bt[k];
#pragma unroll
                    for (int k = 0; k < 8; k++) x[h + k] = cmul(cexp_neg_turns_tab(bh[k] * cur, ct), x[h + k]);
                }
            } else
#pragma unroll
            for (int k = 0; k < (PAIR ? 8 : 16); k++) x[k] = cmul(cexp_neg_turns_tab(btv[k] * cur, ct), x[k]);    // (PAIR comes here with hmul only)
        }
    }
    r16_dit(x);
    if (SPLIT) exchange_split(own16p, chunkp, false);
    else {
#pragma unroll
        for (int k = 0; k < 16; k++) s[17 * tid + k] = x[k];
        ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[row_phys(256 * b + j2 + 16 * k)];
    }
    lvl2_dit<16>(x, j2, w2);
    if (!SPLIT) {
#pragma unroll
        for (int k = 0; k < 16; k++) s[row_phys(256 * b + j2 + 16 * k)] = x[k];
        __syncthreads();
    }
    int jo = tid;
    pin(jo);
    // (SPLIT: the row's pointers are formed again here instead of being held since the top of the kernel)
    int rq = brow, pq = bpol, fq = fc;
    if (SPLIT) { pin(rq); pin(pq); pin(fq); }
    const size_t rowbase2 = SPLIT ? (size_t)rq << 12 : rowbase;
    cplx *const u2 = SPLIT ? (pq ? a.uy : a.ux) + ((size_t)fq << (a.p1 + 12)) + rowbase2 : u;
    const cplx tb = (SPLIT ? a.tpass + rowbase2 : tp)[jo];                      // (asked for ahead of the last register level)
    if (SPLIT) exchange_split(chunkp, outerp, true);
    else {
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[row_phys(tid + 256 * k)];
    }
    lvl2_dit<256>(x, SPLIT ? jo : tid, w1);
#pragma unroll
    for (int k = 0; k < 16; k++) u2[jo + 256 * k] = cmulc(x[k], cmul(tb, bk[k]));
}

// ------------------------------------------------- pass 2 for rows of 512, 1024 and 2048 points, register form ---
// Frames of 2^17 ... 2^19 samples on the 256-row split (2^18 = 4096 symbols x 64 samples is what Run_my_PDM_QPSK.m:21-24 ships
// with): dual polarisation, no PMD.  M = 16 x R x 16 points (R = 2, 4, 8), S = M / 16 threads per row and polarisation, thread
// t holds points t + S k; three register levels per direction -- lvl2_dif<S> on W_M, R-point butterflies at stride 16
// (lvlmid_dif), r16_dif -- with one exchange through the padded LDS row between consecutive levels, as in k_row4k; the
// multiplier is applied on the sixteen bins a thread then holds (bit-reversed order, where the tables are) and the inverse
// mirrors the three levels.  A workgroup of 256 threads takes 256 / S row-polarisations: 512-point rows: 4 rows x 2 (a wave =
// the two polarisations of a row), 1024: 2 rows x 2 (a wave = one polarisation of a row: no workgroup barrier at all), 2048:
// one row x 2 (two waves per polarisation: one barrier per outer exchange).  The LDS-resident k_row makes 9 - 11
// barrier-separated passes over the same rows (0.48 / 0.41 / 0.29 of 8 TB/s at 2^17 / 2^18 / 2^19 samples).
// Twiddles: the compact table of W_M (TwCompact); the middle level's lane twiddles from a 7 x 16 table of the plan (twmid).
// PAIR (PMD plans, inverse_pmd's matrix tables): the same workgroup with the threads dealt so that lanes i and i + 32 of every
// wave hold the same thread index of the X and the Y row (a row-polarisation is then half of 1, 2 or 4 waves: the outer
// exchanges of 1024- and 2048-point rows meet at a workgroup barrier) and pair_multiplier in the place of the scalar phase.
#define ROWG_THREADS 256
#define ROWG_NTW(M) ((M) <= 1024 ? (M) / 2 : (M) / 8 + 4)
// SC (scalar plans: scalar_ssfm, and the electrical filter of the front end): every row-polarisation is a row of the one field.
// SPLIT (rows of 512 / 1024 points, whose row-polarisations are lanes of one wave): the exchanges go through LDS in two halves,
// real parts then imaginary parts, so a row-polarisation's padded row is 8.5 KiB instead of 17 and THREE workgroups share a CU
// (twelve waves instead of eight: the kernel then has to fit 168 registers).
template <int LOGM, bool PAIR, bool SC = false, bool SPLIT = false> __global__ __launch_bounds__(ROWG_THREADS, SPLIT ? 3 : 2) void k_rowreg(SsfmArgs a)
{
    static_assert(!(PAIR && SC), "a scalar plan has no second polarisation to pair with");
    // (PAIR && SPLIT: the PMD form for plans with phasor tables -- pair_multiplier<., true> -- which then fits three waves per SIMD too)
    // (twiddles of the outer level: the half table W_M^k where it fits beside two workgroups' rows -- 512 and 1024 points: 4 / 8
    //  KiB -- and the compact table, one more complex product per twiddle, for 2048 points)
    constexpr bool HALF_TW = LOGM <= 10;
    constexpr int M = 1 << LOGM, S = M / 16, R = M / 256, RP = ROWG_THREADS / S, PITCH = M + M / 16, NTW = ROWG_NTW(M);
    PLX_DYN_LDS(lds);
    if (all_done_or_aborted(a)) return;
    const int tid = threadIdx.x;
    int slot = blockIdx.y / a.nfc;
    const int c = blockIdx.y - slot * a.nfc;       // (channels of a frame: 'sepfields' WDM)
    if (!row_slot(a, slot)) return;
    int f;
    if (!slot_frame(a, slot, f)) return;
    const int fc = f * a.nfc + c;
    const FrameCtl *ctl = a.ctl + f;
    if (ctl->done) return;
    // row within the workgroup, polarisation, thread within the row-polarisation
    constexpr int WPR = S / 32 > 0 ? S / 32 : 1;         // PAIR: waves per row (both polarisations)
    const int rl = PAIR ? (tid >> 6) / WPR : (SC ? tid / S : (tid / S) >> 1), pol = PAIR ? (tid >> 5) & 1 : (SC ? 0 : (tid / S) & 1);
    const int t = PAIR ? ((tid >> 6) % WPR) * 32 + (tid & 31) : tid % S;
    const int rp = SC ? rl : 2 * rl + pol;
    constexpr int ROWS = SC ? RP : RP / 2;               // rows per workgroup
    constexpr bool WAVE_LOCAL = PAIR ? S <= 32 : S <= 64;    // a row-polarisation's threads are lanes of one wave
    cplx *const s = (cplx *)lds + rp * PITCH;            // this row-polarisation's padded row: physical(p) = p + (p >> 4)
    double *const sd = (double *)lds + rp * PITCH;       // SPLIT: the same row, one component at a time
    cplx *const tw = SPLIT ? (cplx *)((double *)lds + RP * PITCH) : (cplx *)lds + RP * PITCH;           // W_M (half or compact table)
    cplx *const tm = tw + NTW;                           // [7][16]: the middle level's twiddles, lane-fastest
    cplx *const ct = tm + 7 * 16;                        // the unit circle in 64 steps (cexp_neg_turns_tab)
    cplx *const bk = ct + PLX_CTAB + 17 * rl;            // tpass[row][S k], k < 16 (the rows' entries on different banks)
    const size_t N = (size_t)M << a.p1;
    const size_t rowbase = ((size_t)blockIdx.x * ROWS + rl) << LOGM;
    cplx *const u = (pol ? a.uy : a.ux) + (size_t)fc * N + rowbase;
    const cplx *const tp = a.tpass + rowbase;
    typename std::conditional<HALF_TW, TwHalf<M>, TwCompact<M>>::type wm{tw};
    const int b = t >> 4, j2 = t & 15;                   // middle level: chunk b of 256 points, point j2 + 16 kk of it
    cplx x[16];
    // SPLIT: one exchange in two halves -- x[k] goes to slot wi(k), the thread's next sixteen values come from slot ri(k)
    // (block: the partners of an OUTER exchange of a 2048-point row are two waves -- workgroup barriers, and none behind the last
    //  read: a slot a thread reads in one exchange is written next by that thread itself, or behind a later barrier)
    auto exchange_split = [&](auto wi, auto ri, bool block = false) {
        // (the real parts travel first and land in x[k].x while x[k].y still holds the OLD imaginary parts: no spare registers)
#pragma unroll
        for (int k = 0; k < 16; k++) sd[wi(k)] = x[k].x;
        if (block) __syncthreads(); else ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k].x = sd[ri(k)];
        if (block) __syncthreads(); else ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) sd[wi(k)] = x[k].y;
        if (block) __syncthreads(); else ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k].y = sd[ri(k)];
        if (!block) ROWR_SYNC();
    };
    // (the padded slots in closed form -- S and 256 are multiples of 16, j2 < 16 -- so that every access is ONE base register + an
    //  immediate: row_phys(t + S k) = t + (t >> 4) + (S + S / 16) k, row_phys(256 b + j2 + 16 k) = 272 b + j2 + 17 k)
    const int obase = t + (t >> 4), cbase = 272 * b + j2, tbase = 17 * t;
    const auto outer = [&](int k) { return obase + (S + S / 16) * k; };
    const auto chunk = [&](int k) { return cbase + 17 * k; };
    const auto own16 = [&](int k) { return tbase + k; };                     // row_phys(16 t + k)
    {
        const cplx ta = tp[t];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = u[t + S * k];
        for (int i = tid; i < NTW; i += ROWG_THREADS) tw[i] = HALF_TW ? a.tw2[i] : a.tw2c[i];
        if (tid < 7 * 16) tm[tid] = a.twmid[tid];
        if (tid < PLX_CTAB) ct[tid] = a.ctab[tid];
        if (pol == 0 && t < 16) bk[t] = tp[S * t];
        __syncthreads();                                 // tables staged
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = cmul(x[k], cmul(ta, bk[k]));
    }
    lvl2_dif<S>(x, t, wm);
    if (SPLIT) exchange_split(outer, chunk, !WAVE_LOCAL);
    else {
#pragma unroll
        for (int k = 0; k < 16; k++) s[row_phys(t + S * k)] = x[k];
        if (!WAVE_LOCAL) __syncthreads(); else ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[row_phys(256 * b + j2 + 16 * k)];
    }
    cplx wl[7];
#pragma unroll
    for (int q = 0; q < (R == 8 ? 7 : (R == 4 ? 3 : 1)); q++) wl[q] = tm[16 * q + j2];     // (the plan lists this R's twiddles first)
    lvlmid_dif<R>(x, wl);                                // (written back where this thread read it: no barrier in between)
    if (SPLIT) exchange_split(chunk, own16);
    else {
#pragma unroll
        for (int k = 0; k < 16; k++) s[row_phys(256 * b + j2 + 16 * k)] = x[k];
        ROWR_SYNC();                                     // (a chunk's sixteen threads are lanes of one wave)
    }
    // SHARE (512-point rows: a wave = the two polarisations of one row, as in k_row256r): the multiplier is the same for the two
    // polarisations of a bin, which sit in lanes i and i + 32 -- the lower half of the wave forms it for the thread's bins 0-7,
    // the upper half for bins 8-15, and they swap (half_share).  (For 1024-point rows the same sharing needs the PAIR dealing of
    // the threads and with it a workgroup barrier at the outer exchanges: measured 939.3 / 941.1 us against 938.4 / 949.9 per
    // 256-frame launch -- nothing, so those rows keep one wave per row-polarisation.)
    constexpr bool SHARE = !PAIR && !SC && LOGM == 9, HALF_BINS = PAIR || SHARE;
    const int ib = 16 * t + (HALF_BINS && pol ? 8 : 0);  // (PAIR: the eight bins whose two polarisations the lane holds after the trade)
    double btv[HALF_BINS ? 8 : 16];
    // (SPLIT without sharing: the sixteen phases are asked for in two halves AT the multiplier -- sixteen registers less across r16_dif,
    //  what the 168-register form of the 1024-point rows is short of)
    constexpr bool LATE_BT = SPLIT && !HALF_BINS;
    if (!LATE_BT && !(PAIR && SPLIT) && !a.hmul && !(PAIR && a.umat)) {
        const double *bt = a.betat_p + (size_t)c * N + rowbase + ib;
#pragma unroll
        for (int k = 0; k < (HALF_BINS ? 8 : 16); k++) btv[k] = bt[k];
    }
    if (!SPLIT) {
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[17 * t + k];           // row_phys(16 t + k)
    }
    r16_dif(x);
    if (PAIR && !a.hmul) {
        pair_multiplier<LOGM, SPLIT>(a, x, btv, ct, ctl, f, c, (int)blockIdx.x * ROWS + rl, rowbase, ib);
    } else if (a.hmul) {
        int o16 = 16 * t;
        pin(o16);
        const cplx *h = a.hmul + rowbase + o16;
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = cmul(h[k], x[k]);
    } else {
        const double cur = a.force ? a.f_cur : ctl->cur;
        if (SHARE) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const cplx h = cexp_neg_turns_tab(btv[k] * cur, ct);
                cplx ha, hb;
                half_share(h, ha, hb);
                x[k] = cmul(ha, x[k]);
                x[k + 8] = cmul(hb, x[k + 8]);
            }
        } else if (LATE_BT) {
#pragma unroll
            for (int h = 0; h < 16; h += 8) {
                int o = ib + h;
                pin(o);
                const double *bt = a.betat_p + (size_t)c * N + rowbase + o;
                double bh[8];
#pragma unroll
                for (int k = 0; k < 8; k++) bh[k] = bt[k];
#pragma unroll
                for (int k = 0; k < 8; k++) x[h + k] = cmul(cexp_neg_turns_tab(bh[k] * cur, ct), x[h + k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < (HALF_BINS ? 8 : 16); k++) x[k] = cmul(cexp_neg_turns_tab(btv[k] * cur, ct), x[k]);      // (PAIR comes here with hmul only)
        }
    }
    r16_dit(x);
    if (SPLIT) exchange_split(own16, chunk);
    else {
#pragma unroll
        for (int k = 0; k < 16; k++) s[17 * t + k] = x[k];
        ROWR_SYNC();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[row_phys(256 * b + j2 + 16 * k)];
    }
    {
        // (the lane's twiddles are fetched again rather than held across the multiplier: up to 28 registers)
        int jq = j2;
        pin(jq);
#pragma unroll
        for (int q = 0; q < (R == 8 ? 7 : (R == 4 ? 3 : 1)); q++) wl[q] = tm[16 * q + jq];
    }
    lvlmid_dit<R>(x, wl);
    if (!SPLIT) {
#pragma unroll
        for (int k = 0; k < 16; k++) s[row_phys(256 * b + j2 + 16 * k)] = x[k];
        if (!WAVE_LOCAL) __syncthreads(); else ROWR_SYNC();
    }
    int jo = t;
    pin(jo);
    // (SPLIT: the row's two pointers are formed again here instead of being held -- or spilled -- since the top of the kernel)
    int rq = rl, pq = pol, fq = fc;
    if (SPLIT) { pin(rq); pin(pq); pin(fq); }
    const size_t rowbase2 = SPLIT ? ((size_t)blockIdx.x * ROWS + rq) << LOGM : rowbase;
    cplx *const u2 = SPLIT ? (pq ? a.uy : a.ux) + ((size_t)fq << (LOGM + a.p1)) + rowbase2 : u;
    const cplx tb = (SPLIT ? a.tpass + rowbase2 : tp)[jo];                              // (asked for ahead of the last register level)
    if (SPLIT) exchange_split(chunk, outer, !WAVE_LOCAL);
    else {
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = s[row_phys(t + S * k)];
    }
    lvl2_dit<S>(x, (PAIR || SPLIT) ? jo : t, wm);                   // (PAIR, jo: the lane's three second-stage twiddles are formed again, not held across the trunk loop)
#pragma unroll
    for (int k = 0; k < 16; k++) u2[jo + S * k] = cmulc(x[k], cmul(tb, bk[k]));
}
#define ROWG_LDS(M) ((size_t)((ROWG_THREADS / ((M) / 16)) * ((M) + (M) / 16) + ROWG_NTW(M) + 7 * 16 + PLX_CTAB + 17 * (ROWG_THREADS / ((M) / 16))) * sizeof(cplx))
#define ROWG_LDS_SPLIT(M) (ROWG_LDS(M) - (size_t)((ROWG_THREADS / ((M) / 16)) * ((M) + (M) / 16)) * sizeof(double))

// ------------------------------------------------- pass 2 for rows of 32, 64 and 128 points, register form ---
// Frames of 2^13 ... 2^15 samples on the 256-row split (the sizes of the reference's own examples: ex19 / ex20 run 256 x 64 =
// 2^14): dual polarisation without PMD, and scalar plans.  M = R x 16 points (R = 2, 4, 8): R threads per row and polarisation
// with 16 points each, ONE WAVE = 64 / R row-polarisations, no workgroup barrier.  Thread j holds points j + R k; for each
// i = j + R par (par < 16 / R) the R points i + 16 q sit in R of its registers -- one radix-R set (radset_dif: the middle
// level of k_rowreg, on a row that is a single block) -- then one exchange through the wave's padded rows (in real /
// imaginary halves: 8.5 KiB per wave) hands every thread sixteen contiguous points for r16_dif; multiplier on the thread's
// sixteen bins; the inverse mirrors it.  The inter-pass twiddles are read from tpass directly (16 per thread and direction).
template <int LOGM, bool SC> __global__ __launch_bounds__(64, 3) void k_rowsm(SsfmArgs a)
{
    constexpr int M = 1 << LOGM, R = M / 16, NS = 16 / R, RPW = 64 / R, NQ = R == 8 ? 7 : (R == 4 ? 3 : 1);
    PLX_DYN_LDS(lds);
    if (all_done_or_aborted(a)) return;
    const int tid = threadIdx.x;
    int slot = blockIdx.y / a.nfc;
    const int c = blockIdx.y - slot * a.nfc;       // (channels of a frame: 'sepfields' WDM)
    if (!row_slot(a, slot)) return;
    int f;
    if (!slot_frame(a, slot, f)) return;
    const int fc = f * a.nfc + c;
    const FrameCtl *ctl = a.ctl + f;
    if (ctl->done) return;
    const int g = tid / R, j = tid - g * R;              // row-polarisation of the wave, thread within it
    const int rp = (int)blockIdx.x * RPW + g, row = SC ? rp : rp >> 1, pol = SC ? 0 : rp & 1;
    double *const sd = (double *)lds + g * (17 * R);     // this row-polarisation's padded row (one component at a time): physical(p) = p + (p >> 4)
    cplx *const tm = (cplx *)((double *)lds + 64 * 17);  // [7][16]: the radix-R level's twiddles by i (the plan lists this R's first)
    cplx *const ct = tm + 7 * 16;                        // the unit circle in 64 steps (cexp_neg_turns_tab)
    const size_t N = (size_t)M << a.p1;
    const size_t rowbase = (size_t)row << LOGM;
    cplx *const u = (pol ? a.uy : a.ux) + (size_t)fc * N + rowbase;
    const cplx *const tp = a.tpass + rowbase;
    cplx x[16];                                          // x[R par + q] = point i + 16 q, i = j + R par
#pragma unroll
    for (int k = 0; k < 16; k++) x[R * (k % NS) + k / NS] = u[j + R * k];
    tm[tid] = a.twmid[tid];
    if (tid < 7 * 16 - 64) tm[64 + tid] = a.twmid[64 + tid];
    ct[tid] = a.ctab[tid];
    {
        cplx tv[16];
#pragma unroll
        for (int k = 0; k < 16; k++) tv[k] = tp[j + R * k];
#pragma unroll
        for (int k = 0; k < 16; k++) x[R * (k % NS) + k / NS] = cmul(x[R * (k % NS) + k / NS], tv[k]);
    }
    ROWR_SYNC();                                         // tables staged (one wave)
#pragma unroll
    for (int par = 0; par < NS; par++) {
        cplx w[7];
#pragma unroll
        for (int q = 0; q < NQ; q++) w[q] = tm[16 * q + j + R * par];
        radset_dif<R>(x + R * par, w);
    }
    // exchange: register (par, q) = point i + 16 q goes to slot i + 17 q; the thread then takes the sixteen points of block j
#pragma unroll
    for (int k = 0; k < 16; k++) sd[j + R * (k / R) + 17 * (k % R)] = x[k].x;
    ROWR_SYNC();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k].x = sd[17 * j + k];
    ROWR_SYNC();
#pragma unroll
    for (int k = 0; k < 16; k++) sd[j + R * (k / R) + 17 * (k % R)] = x[k].y;
    ROWR_SYNC();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k].y = sd[17 * j + k];
    ROWR_SYNC();
    r16_dif(x);
    if (a.hmul) {
        const cplx *h = a.hmul + rowbase + 16 * j;
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = cmul(h[k], x[k]);
    } else {
        const double cur = a.force ? a.f_cur : ctl->cur;
#pragma unroll
        for (int h = 0; h < 16; h += 8) {
            int o = 16 * j + h;
            pin(o);
            const double *bt = a.betat_p + (size_t)c * N + rowbase + o;
            double bh[8];
#pragma unroll
            for (int k = 0; k < 8; k++) bh[k] = bt[k];
#pragma unroll
            for (int k = 0; k < 8; k++) x[h + k] = cmul(cexp_neg_turns_tab(bh[k] * cur, ct), x[h + k]);
        }
    }
    r16_dit(x);
#pragma unroll
    for (int k = 0; k < 16; k++) sd[17 * j + k] = x[k].x;
    ROWR_SYNC();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k].x = sd[j + R * (k / R) + 17 * (k % R)];
    ROWR_SYNC();
#pragma unroll
    for (int k = 0; k < 16; k++) sd[17 * j + k] = x[k].y;
    ROWR_SYNC();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k].y = sd[j + R * (k / R) + 17 * (k % R)];
#pragma unroll
    for (int par = 0; par < NS; par++) {
        cplx w[7];
#pragma unroll
        for (int q = 0; q < NQ; q++) w[q] = tm[16 * q + j + R * par];
        radset_dit<R>(x + R * par, w);
    }
    {
        int jo = j;
        pin(jo);
        cplx tv[16];
#pragma unroll
        for (int k = 0; k < 16; k++) tv[k] = tp[jo + R * k];
#pragma unroll
        for (int k = 0; k < 16; k++) u[jo + R * k] = cmulc(x[R * (k % NS) + k / NS], tv[k]);
    }
}
#define ROWSM_LDS ((size_t)64 * 17 * sizeof(double) + (7 * 16 + PLX_CTAB) * sizeof(cplx))

// ------------------------------------------------------ pass 3: inverse columns ---
// Completes ifft (1/N), applies the attenuation of the step (:531-532) and feeds
// nextstep's global maximum (:694-696) -- no extra pass over the field.
__global__ __launch_bounds__(COL_THREADS_MAX) void k_col_inv(SsfmArgs a)
{
    PLX_DYN_LDS(lds);
    if (all_done_or_aborted(a)) return;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int slot = blockIdx.y / a.nfc;
    int f;
    if (!slot_frame(a, slot, f)) return;
    const int fc = f * a.nfc + (blockIdx.y - slot * a.nfc);
    const FrameCtl *ctl = a.ctl + f;
    if (ctl->done) return;
    const int N1 = 1 << a.p1, N2 = 1 << a.p2, T = a.T, W = a.W;
    cplx *s = (cplx *)lds;
    cplx *tw = s + ((size_t)N1 << a.logT);
    double *red = (double *)(tw + (N1 >> 1));
    lds_load_twiddles(tw, a.tw1, N1 >> 1, tid, nthr);
    const size_t base = (size_t)fc << (a.p1 + a.p2);
    const int col0 = blockIdx.x * W;
    const int nel = N1 << a.logW;
    const cplx *uyp = a.dual ? a.uy : a.ux; // scalar plans read ux twice (second copy unused)
    for (int e0 = tid; e0 < nel; e0 += nthr * COL_CH) {
        cplx xv[COL_CH], yv[COL_CH];
#pragma unroll
        for (int k = 0; k < COL_CH; k++) {
            const int e = min(e0 + k * nthr, nel - 1); // clamped duplicate loads keep the batch branch-free
            const size_t g = base + (size_t)(e >> a.logW) * N2 + col0 + (e & (W - 1));
            xv[k] = a.ux[g];
            yv[k] = uyp[g];
        }
#pragma unroll
        for (int k = 0; k < COL_CH; k++) { pin(xv[k]); pin(yv[k]); }
#pragma unroll
        for (int k = 0; k < COL_CH; k++) {
            const int e = e0 + k * nthr;
            if (e < nel) {
                const int o = ((e >> a.logW) << a.logT) + (e & (W - 1));
                s[o] = xv[k];
                if (a.dual) s[o + W] = yv[k];
            }
        }
    }
    __syncthreads();
    lds_fft_dit(s, a.p1, T, 1, a.logT, tw, tid, nthr, true);
    const double sc = a.force ? a.f_sc : ctl->att * a.invN;
    double m = 0;
    for (int e = tid; e < nel; e += nthr) {
        const int row = e >> a.logW, col = e & (W - 1);
        const size_t g = base + (size_t)row * N2 + col0 + col;
        cplx x = cscale(s[(row << a.logT) + col], sc);
        double p = x.x * x.x + x.y * x.y;
        a.ux[g] = x;
        if (a.dual) {
            cplx y = cscale(s[(row << a.logT) + W + col], sc);
            p = p + y.x * y.x;
            p = p + y.y * y.y;
            a.uy[g] = y;
        }
        m = p > m ? p : m;
    }
    block_atomic_max(m, red, a.umax + fc, tid, nthr);
}

// The rare full-range Kerr step of k_colx16 ('--s-' exact single step: |gamma Leff P| not small), on the tile parked in the
// exchange buffer, one lane per polarisation pair.  Out of line for the same reason as ctrl_head_call: the argument
// reduction constants of sincos must not live in the registers of the hot loop.
__device__ __noinline__ void kerr_full_range(int j, int t, double gamleff, int manakov)
{
    PLX_DYN_LDS(lds);
    cplx *s = (cplx *)lds;
    for (int k = 0; k < 16; k++) {
        cplx X = s[((j + 16 * k) << 4) + t], Y = s[((j + 16 * k) << 4) + t + 8];
        const double P = X.x * X.x + X.y * X.y + Y.x * Y.x + Y.y * Y.y;
        const cplx nl = cexpi(-gamleff * P);
        X = cmul(X, nl);
        Y = cmul(Y, nl);
        if (!manakov) {
            const double s3 = 2 * (X.x * Y.y - X.y * Y.x);
            double sp, cp;
            sincos(gamleff * s3 / 3, &sp, &cp);
            const cplx xx = make_double2(cp * X.x + sp * Y.x, cp * X.y + sp * Y.y);
            const cplx yy = make_double2(cp * Y.x - sp * X.x, cp * Y.y - sp * X.y);
            X = xx; Y = yy;
        }
        s[((j + 16 * k) << 4) + t] = X;
        s[((j + 16 * k) << 4) + t + 8] = Y;
    }
}

// ... and of the scalar form of the sweep: nl_step (:792-804, SPM only) on the lane's column of the parked tile.
__device__ __noinline__ void kerr_full_range_scalar(int j, int t, double gam, double leff)
{
    PLX_DYN_LDS(lds);
    cplx *s = (cplx *)lds;
    for (int k = 0; k < 16; k++) {
        const cplx X = s[((j + 16 * k) << 4) + t];
        const double pw = X.x * X.x + X.y * X.y;
        s[((j + 16 * k) << 4) + t] = cmul(X, cexpi(-gam * pw * leff));
    }
}

// ----------------------------------------------------------------------------------------------
// k_colx16: the fused column sweep for the 256 x (8+8) tile with both column transforms held in
// REGISTERS (16 points per thread, r16_* + lvl2_*256): per tile one LDS exchange per transform instead
// of four read+write passes, the Kerr step on registers (the other polarisation of a sample sits in
// lane t^8: one DPP row rotation).  The kernel sits at the VGPR cap (16 FP64 complex points per lane), so
// the NEXT tile is staged by LDS-DMA (global_load_lds_dwordx4: no register destination) straight into the
// exchange buffer as soon as the current tile has left it, and is in flight during the last register
// transform and the stores of the current tile.
// Thread = (j = tid>>4, t = tid&15): t < 8 -> column t of ux, t >= 8 -> column t-8 of uy.
// LDS image of a tile: s[row][16] (256 B per row: 8 columns of ux | 8 of uy); one LDS-DMA instruction of a wave
// fills 4 consecutive rows (64 lanes x 16 B = 1 KiB, lane-linear), wave w owns rows 64w .. 64w+63 -- exactly the
// rows its own threads read first, so the landing needs the wave's own vmcnt wait and no workgroup barrier.
//
// Landing WITHOUT a vmcnt wait.  A gfx9-family wave has one counter for its loads and its stores, and they retire out of
// order with respect to each other: waiting for the staged tile through vmcnt means waiting for the acknowledgement of
// every store of the previous tile as well (~1 us at the top of every tile, the most variable microsecond of the loop, and
// what varies shows up again as waiting at the frame barrier).  So the copies are issued from inline assembly (the
// compiler's wait-count pass does not see them and inserts no wait of its own in front of the LDS reads), the frame record
// goes LAST into a copy whose `done` word holds a sentinel, and the wave spins on that word in LDS: loads return in
// issue order, so the record's arrival implies the tile's.  The workgroup barriers of the tile loop are bare s_barrier +
// lgkmcnt waits for the same reason (__syncthreads carries a release fence = a vmcnt wait while stores are in flight).
#define PLX_REC_SENTINEL 0x7fffffff
#define COLX_NFC 64            // channels whose gam the fused sweep keeps in LDS
#ifdef PLX_EMU
__device__ __forceinline__ void glds16(const cplx *src, cplx *lds_wave_base, int lane) { lds_wave_base[lane] = *src; }
// rows row .. row+63 of a tile: 16 copies of 4 rows each; src: this lane's first element, stride: elements between copies
__device__ __forceinline__ void glds_rows(const cplx *src, size_t stride, cplx *lds_wave_base, int lane)
{
    for (int i = 0; i < 16; i++) glds16(src + i * stride, lds_wave_base + 64 * i, lane);
}
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
__device__ __forceinline__ int lds_peek(const int *p) { return *(const volatile int *)p; }
__device__ __forceinline__ void lds_settle() {}
__device__ __forceinline__ void emu_lockstep() { __syncthreads(); }   // (the emulator's lanes are free-running threads: a wave's lanes meet here)
#else
__device__ __forceinline__ void glds16(const cplx *src, cplx *lds_wave_base, int)
{
    const unsigned lb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)lds_wave_base);   // (wave-uniform by construction)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lb) : "memory", "m0");
}
__device__ __forceinline__ void glds_rows(const cplx *src, size_t stride, cplx *lds_wave_base, int)
{
    unsigned lb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    unsigned long long p = (unsigned long long)src;
    const unsigned long long st = (unsigned long long)stride * sizeof(cplx);
#define PLX_GLDS_STEP "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tv_lshl_add_u64 %0, %0, 0, %2\n\ts_add_u32 %1, %1, 0x400\n\t"
    asm volatile(PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP
                 PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP
                 : "+v"(p), "+s"(lb) : "s"(st) : "memory", "m0", "scc");
#undef PLX_GLDS_STEP
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int lds_peek(const int *p)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) void *)p) : "memory");
    return v;
}
__device__ __forceinline__ void lds_settle() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void emu_lockstep() {}
#endif


// D = false: the same sweep for SCALAR plans (scalar_ssfm, fiber.m:557-636, without XPM -- its row sums across channels would
// need the other channels' tiles): a tile is sixteen columns of the one field, lane t its column t; the frame maximum is
// max |u|^2 (nextstep :694-698 with ~isy), the Kerr step nl_step's u .*= fastexp(-gam |u|^2 leff) (:792-804) on the lane's own
// sixteen points -- no lane pairs, no swapped halves.
template <bool D> __global__ __launch_bounds__(256, 2) void k_colx16(SsfmArgs a, int tiles_x, int tiles_pf)
{
    constexpr int CW = D ? 8 : 16;         // columns of one polarisation in a tile
    PLX_DYN_LDS(lds);
    if (all_done_or_aborted(a)) return;
    const int tid = threadIdx.x, t = tid & 15, j = tid >> 4;
    const int N2 = 1 << a.p2;
    const int LOGN = a.p1 + a.p2;
    cplx *s = (cplx *)lds;                 // [256][16] exchange buffer
    cplx *tw = s + 4096;                   // W_256^k, k < 128
    double *red = (double *)(tw + 128);
    FrameCtl *lctl = (FrameCtl *)(red + 32);
    lds_load_twiddles(tw, a.tw1, 128, tid, 256);
    cplx *const fld = (D && t >= 8) ? a.uy : a.ux;
    const int round = a.round;
    const int colt = D ? t & 7 : t;
    const bool isx = !D || t < 8;
    const int lane = tid & 63, row0 = (tid >> 6) * 64;     // this wave stages rows row0 .. row0+63
    // Teams.  The grid is a whole number of TEAMS of tiles_pf workgroups; a team takes a frame at a time, workgroup ti of the
    // team its tile ti (channel c, column block bx: the same every time round).  A team's first frame is slot `team` of the
    // active list; the further ones are claimed from a counter, one at a time, so a team that runs late -- its workgroups
    // found no room beside another kernel's waves, or its CUs are slow -- leaves its share to the others instead of holding
    // the launch up.  The team's first workgroup claims the frame of iteration k+2 while the team sits at the barrier of
    // iteration k (its second wave does: it has nothing else to do there) and posts it in the team's mailbox; everybody
    // picks the frame of iteration k+1 up with the polls of barrier k, where it has been lying for a whole iteration.
    // The mailbox is a log, one entry per iteration and no reuse within a launch: through finished frames of a stale list the
    // team's workgroups run without meeting, and its first may be any number of iterations ahead of its slowest.
    const int nact = a.nactive[0];
    const int NT = gridDim.x / tiles_pf, team = blockIdx.x / tiles_pf, ti = blockIdx.x - team * tiles_pf;
    const int c = ti / tiles_x, bx = ti - c * tiles_x;
    const unsigned long long rtag = (unsigned long long)(((unsigned)round + 1u) & 0xfffffu) << 22;
    unsigned long long *const mbox = a.mbox + (size_t)a.mbox_stride * team;
    if (blockIdx.x == 0 && tid == 0) a.grab[(round & 1) ^ 1] = 0;        // (the other parity's counter: for the next launch)
    auto post = [&](int k) {               // the team's first workgroup, thread 64: claim the frame of iteration k and post it
        const int sl = NT + atomicAdd(a.grab + (round & 1), 1);
        const int fr = sl < nact ? a.active[sl] : -1;
        if (k < a.mbox_stride) st_agent(mbox + k, ((rtag | (unsigned long long)(k + 1)) << 22) | (unsigned long long)(fr + 1));
    };
    auto posted = [&](int k, unsigned long long v) -> bool { return (v >> 22) == (rtag | (unsigned long long)(k + 1)); };
    // stage(f, par): start the asynchronous copy of this workgroup's tile of frame f into s, and of the frame's step-control
    // record into this wave's own copy (7 lanes x 16 B): everything the next iteration needs arrives without a load between
    // its loop top and its first transform.  The list may be a few steps old (small batches rebuild it once per chunk of
    // steps), so the record's `done` is still checked at the loop top; it cannot change before THIS workgroup has met the
    // frame's barrier.
    // (record copies: [iteration parity][wave] -- the copy of the tile in hand is still needed while the next one lands)
    int it = 0;
    auto stage = [&](int f, int par) {
        const int fc = f * a.nfc + c;
        // (lane & 15 == t: the lane stages a piece of the same column of the same polarisation it later works on)
        int lq = lane >> 4;
        pin(lq);                           // (addresses are formed where they are used: hoisted out of the tile loop they end up in scratch)
        const cplx *src = fld + ((size_t)fc << LOGN) + (size_t)bx * CW + colt + (size_t)(row0 + lq) * N2;
        FrameCtl *const rec = lctl + 4 * par + (tid >> 6);
        if (lane == (int)(offsetof(FrameCtl, done) / 16)) rec->done = PLX_REC_SENTINEL;   // (the lane whose piece of the record holds the word)
        lds_settle();                      // (the sentinel is in place before the copy that replaces it can land)
        glds_rows(src, (size_t)4 * N2, s + (size_t)row0 * 16, lane);
        static_assert(sizeof(FrameCtl) % 16 == 0, "the record travels as 16-byte pieces");
        int ln = lane;
        pin(ln);                           // (the address is formed here: kept across the tile loop it would sit in scratch)
        if (ln < (int)(sizeof(FrameCtl) / 16)) glds16((const cplx *)(a.ctl + f) + ln, (cplx *)rec, ln);
    };
    // [stamps:init]
    double *const gaml = (double *)((char *)(lctl + 8) + 128);    // gam[channel] (at most COLX_NFC channels: checked by the plan)
    for (int k = tid; k < a.nfc; k += 256) gaml[k] = a.gam[k];
    // The lanes of the second polarisation (t >= 8) take the odd twiddles of the two m = 256 stages from a NEGATED copy of the
    // table (lvl2_dit256s / lvl2_dif256s): between the inverse and the forward transform their registers hold the halves of
    // the tile's points swapped, y[k] = point j + 16 (k ^ 8), so that the register pair (k, k + 8) of the lane pair (t, t ^ 8)
    // is the two polarisations of ONE sample -- for k < 8 the X lane's sample k and the Y lane's sample k + 8 -- and the
    // frame maximum and the Kerr step need no per-lane selects (130 v_cndmask per tile before).  Exact: negations only.
    cplx *const twn = (cplx *)(gaml + COLX_NFC);
    if (D && tid < 128) { const cplx w = a.tw1[tid]; twn[tid] = make_double2(-w.x, -w.y); }
    // (the per-lane table pointer and the LDS distance between the halves of a column, 2048 elements for the lanes that swap
    //  them, are re-derived from t where they are used: held across the tile loop they cost the two registers that spill)
#define COLX_TWA(tp) ((D && (tp) >= 8) ? (const cplx *)twn : (const cplx *)tw)
#define COLX_HSW(tp) ((D && (tp) >= 8) ? 2048 : 0)
    int f = team < nact ? a.active[team] : -1;
    if (f < 0) return;                     // (more teams than frames)
    stage(f, 0);
    if (ti == 0 && tid == 64) post(1);     // (the one claim nobody's wait hides: once per launch)
    // The workgroup of a frame's first tile owns the frame's record: it finishes the controller (ctrl_tail) and writes the
    // record back LATER, while it waits at the barrier of its next tile (red[8]: the frame owed, or -1).
    CtrlK *const kk = (CtrlK *)(lctl + 8);
    if (tid == 0) {
        red[8] = -1.0;
        kk->dphimax = a.dphimax; kk->alphalin = a.alphalin; kk->dzmax = a.dzmax; kk->dz0 = a.dz0; kk->zdone0 = a.zdone0; kk->Lf = a.Lf; kk->lcorr = a.lcorr;
        kk->dual = a.dual ? 1 : 0; kk->resume = a.resume ? 1 : 0; kk->ncycle0 = a.ncycle0; kk->nfc = 0; kk->ndone = a.ndone; kk->umax = nullptr; kk->gam = nullptr;
        kk->dzlist = a.dzlist; kk->dzlog = a.dzlog; kk->ndz = a.ndz; kk->logcap = a.logcap;
    }
    auto settle = [&](int par) {           // tid 0 only; par: the parity the owed record was staged in
        const int pf = (int)red[8];
        if (pf < 0) return;
        FrameCtl *pr = lctl + 4 * par;
        if (!pr->done) ctrl_tail_call((const PLX_LDS_QUAL CtrlK *)kk, (PLX_LDS_QUAL FrameCtl *)pr);
        a.ctl[pf] = *pr;
        red[8] = -1.0;
    };
    __syncthreads();                       // twiddles staged
    // (red[10 + (it & 1)]: the team's next frame, by iteration parity -- the waves of a workgroup read it at their own pace at the
    //  end of an iteration, and the next iteration may write its successor with no workgroup barrier in between)
    for (;; it++) {
        FrameCtl *const wrec = lctl + 4 * (it & 1) + (tid >> 6);
        const int fc = f * a.nfc + c;
        // [phase 0] loop top
        if (a.safe_land) drain_vmem();     // (checking mode: the ordinary wait as well -- results must not depend on it)
        while (lds_peek(&wrec->done) == PLX_REC_SENTINEL) nap();   // this wave's rows of the tile and its copy of the record are in LDS
        emu_lockstep();
        // [phase 1] wait for the staged tile
        if (wrec->done) {                  // a listed frame that has finished meanwhile (the same answer in every wave)
            if (tid == 0) {
                settle((it & 1) ^ 1);
                unsigned long long v;
                unsigned spins = 0;
                bool dead = false;
                const long long t0 = plx_clock();
                while (!posted(it + 1, v = ld_agent(mbox + (it + 1)))) {
                    nap();
                    if ((++spins & 255u) == 0 && (ld_agent((const unsigned *)a.ndone + 1) != 0 || plx_clock() - t0 > a.spin_ticks)) {
                        st_agent((unsigned *)a.ndone + 1, 1u);      // (the same sticky abort as the frame barrier's)
                        dead = true;
                        break;
                    }
                }
                red[10 + (it & 1)] = dead ? -2.0 : (double)((int)(v & 0x3fffffull) - 1);
            } else if (ti == 0 && tid == 64) {
                post(it + 2);
            }
            lds_barrier();
            f = (int)red[10 + (it & 1)];
            if (f == -2) return;           // (timed out: uniform over the workgroup)
            if (f < 0) { it++; break; }
            stage(f, (it & 1) ^ 1);        // (s is free here: every path below ends past its last read of s, and so far
            continue;                      //  each wave has only touched its own rows)
        }
        const size_t cbase = ((size_t)fc << LOGN) + (size_t)bx * CW + colt;   // in the caller's arrays
        const bool started = wrec->started != 0;
        {
            cplx x[16];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = s[((16 * j + k) << 4) + t];
            if (started) {                 // rows 16j .. 16j+15 of the (bit-reversed) column spectrum
                r16_dit(x);
#pragma unroll
                for (int k = 0; k < 16; k++) s[((16 * j + k) << 4) + t] = x[k];
            }
        }
        lds_barrier();
        // [phase 2] r16_dit + exchange write + workgroup barrier
        cplx y[16];                        // point j + 16k (lanes t >= 8: j + 16 (k ^ 8) from here to the forward transform)
        {
            // (a frame that has not started yet is read with the halves swapped; a started one gets them swapped by the
            //  negated twiddles of the inverse transform's last stage)
            int tp = t;
            pin(tp);
            const int ysw = started ? 0 : COLX_HSW(tp);
#pragma unroll
            for (int k = 0; k < 8; k++) y[k] = s[((j + 16 * k) << 4) + t + ysw];
#pragma unroll
            for (int k = 8; k < 16; k++) y[k] = s[((j + 16 * k) << 4) + t - ysw];
        }
        double sc = 1.0;
        if (started) {                     // finish step s: ifft (1/N), attenuation (:531-532)
            lvl2_dit256s(y, j, tw, COLX_TWA(t));
            sc = wrec->att * a.invN;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) y[k] = cscale(y[k], sc);
        double m = 0;
        if (!D) {
#pragma unroll
            for (int k = 0; k < 16; k++) m = fmax(fma(y[k].y, y[k].y, y[k].x * y[k].x), m);
        } else
#pragma unroll
        for (int k = 0; k < 8; k++) {
            // |ux|^2 + |uy|^2 of ONE sample: this lane's y[k] and the partner lane's y[k + 8] (lane t ^ 8).  The X lane forms
            // it for the samples k < 8, the Y lane for k + 8: the pair covers the column, each sum once (the same two powers
            // added as before: a + b == b + a, the maximum is the same to the bit)
            const double po = fma(y[k].y, y[k].y, y[k].x * y[k].x);
            const double pq = fma(y[k + 8].y, y[k + 8].y, y[k + 8].x * y[k + 8].x);
            const double p = po + lane_xchg<8>(pq);
            m = fmax(p, m);
        }
        m = wave_max(m);
        if ((tid & 63) == 0) red[tid >> 6] = m;
        lds_barrier();
        // [phase 3] exchange read + lvl2_dit + scale + max
        // Frame barrier (dz of the next step needs the frame-wide maximum, fiber.m:694-698): an all-gather.  Every workgroup
        // stores its tile maximum into its own slot, then its first wave polls the slots of the whole frame and runs the
        // step controller itself on its copy of the record (the same inputs, the same instructions: the same step in every
        // workgroup, to the bit) -- one store-to-load trip across the chip instead of two (members -> leader -> members).
        // One launch = one round, so kernel boundaries order the rounds and the protocol needs no read-modify-write: the slots
        // of the two launch parities alternate, and a workgroup empties its slot of the OTHER parity (last read one launch
        // ago) for the next round.  The workgroup of the frame's first tile writes the record back (k_row reads it) and
        // counts the finished frame.
        if (tid < 64) {
            const unsigned par = (unsigned)round & 1u;
            unsigned long long *slots = a.slots + ((size_t)par * a.nframes + f) * tiles_pf;
            double mm = red[0];
            for (int w = 1; w < 4; w++) mm = red[w] > mm ? red[w] : mm;
            const unsigned long long mine = (unsigned long long)__double_as_longlong(mm);
            if (tid == 0) {
                st_agent(slots + ti, mine);
                st_agent(a.slots + ((size_t)(par ^ 1u) * a.nframes + f) * tiles_pf + ti, ~0ull);
                settle((it & 1) ^ 1);      // (the wait below hides it)
            }
            double pm;
            unsigned long long mv = 0;      // (lane 0: the team's mailbox entry of the next iteration, read with the polls)
            unsigned spins = 0;
            bool dead = false;
            const long long t0 = plx_clock();
            for (;;) {
                bool all = true;
                pm = -INFINITY;
                if (tid == 0) { mv = ld_agent(mbox + (it + 1)); all = posted(it + 1, mv); }
                int i0 = tid;
                pin(i0);
                for (int i = i0; i < tiles_pf; i += 64) {
                    const unsigned long long b = (i == ti) ? mine : ld_agent(slots + i);
                    if (b == ~0ull) all = false;
                    else { const double gp = gaml[i / tiles_x] * __longlong_as_double((long long)b); pm = gp > pm ? gp : pm; }
                }
    // [stamps:poll]
                if (__all(all)) break;
                nap();
                if ((++spins & 255u) == 0) {           // (wave-uniform: every lane evaluates the same test)
                    const int late = ld_agent((const unsigned *)a.ndone + 1) != 0 || plx_clock() - t0 > a.spin_ticks;
                    if (__any(late)) { dead = true; break; }    // the frame's partners are not co-resident: abort, store nothing
                }
            }
            pm = wave_max(pm);
            // [phase 8] (dev) slot store -> every slot of the frame seen
            if (tid == 0) {
                red[10 + (it & 1)] = (double)((int)(mv & 0x3fffffull) - 1);
                if (dead) {
                    st_agent((unsigned *)a.ndone + 1, 1u);
                    red[19] = 1.0;
                } else {
                    const double pv = ctrl_head_call((const PLX_LDS_QUAL CtrlK *)kk, (PLX_LDS_QUAL FrameCtl *)wrec, ti == 0, pm, f);
                    if (ti == 0) red[8] = (double)f;
                    red[16] = pv; red[17] = pv < 0 ? 1.0 : 0.0; red[18] = mm; red[19] = 0.0;
                }
            }
        } else if (ti == 0 && tid == 64) {
            post(it + 2);                  // (this wave only waits for the first one here)
        }
        lds_barrier();
        // [phase 4] frame barrier
        if (red[19] != 0.0) return;        // barrier timed out (uniform over the workgroup): no store, no control update
        const double leff = red[16];
        const bool finished = red[17] != 0.0;
        if (finished) {                    // the frame has reached the fibre end: write the field out
            const int nf = (int)red[10 + (it & 1)];    // (the team's next frame, or -1: none left)
            if (nf >= 0) stage(nf, (it & 1) ^ 1);  // (every thread is past its reads of s: the barrier above)
            {
                int tq = t, jq = j;        // (opaque here: the sixteen row offsets of this once-per-frame store are not loop invariants
                pin(tq);                   //  worth thirty-two registers of the tile loop)
                pin(jq);
                const int rsw = (D && tq >= 8) ? 128 : 0;  // (the second polarisation's lanes hold the halves swapped)
#pragma unroll
                for (int k = 0; k < 8; k++) fld[cbase + (size_t)(jq + 16 * k + rsw) * N2] = y[k];
#pragma unroll
                for (int k = 8; k < 16; k++) fld[cbase + (size_t)(jq + 16 * k - rsw) * N2] = y[k];
            }
        } else {
            if (!D) {
                if (a.spm) {               // nl_step (:792-804, SPM only) on the lane's own sixteen points
                    const double gam = gaml[c];
                    if (fabs(gam * leff) * red[18] < 0.0625) {
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const double pw = y[k].x * y[k].x + y[k].y * y[k].y;
                            double sn, cs;
                            sincos_taylor(-gam * pw * leff, &sn, &cs);
                            y[k] = cmul(y[k], make_double2(cs, sn));
                        }
                    } else {               // ('--s-': the exact single step, radians of phase) through the exchange buffer
#pragma unroll
                        for (int k = 0; k < 16; k++) s[((j + 16 * k) << 4) + t] = y[k];
                        lds_barrier();
                        kerr_full_range_scalar(j, t, gam, leff);
                        lds_barrier();
#pragma unroll
                        for (int k = 0; k < 16; k++) y[k] = s[((j + 16 * k) << 4) + t];
                        lds_barrier();
                    }
                }
            } else if (a.spm) {            // Kerr step of step s+1 (:832-852) on registers
                const double gamleff = gaml[c] * leff;
                // |gamleff*P| <= gamleff * (tile maximum): a few mrad under the step controller, so the
                // Taylor form applies to the whole tile; otherwise ('--s-' exact single step) the rare
                // full-range path goes through LDS, one thread per polarisation pair.
                int tq = t;
                pin(tq);                   // (formed here, per tile: hoisted out of the tile loop the two constants would cost four registers for good)
                const double sgn = tq < 8 ? 1.0 : -1.0, sgn2 = tq < 8 ? 2.0 : -2.0;
                if (fabs(gamleff) * red[18] < 0.0625) {
                    // (one loop per equation: a uniform branch inside the unrolled body would cut it into 16 basic blocks)
                    // A sample's two polarisations sit in the lane pair (t, t^8), and the Kerr rotation is the same arithmetic
                    // on both: the pair shares the work by SAMPLES instead of doing all of it twice -- the X lane takes sample
                    // k (its own ux, the partner's uy), the Y lane sample k+8 (its own uy, the partner's ux); each forms both
                    // outputs of its sample and hands the partner's back.  own / oth = this lane's and the other polarisation.
                    auto kerr16 = [&](auto cn) {
                        constexpr bool CNLSE = decltype(cn)::value;
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            // (y[k] is this lane's polarisation of ITS sample, y[k + 8] the partner's sample: see twn above)
                            const cplx own = y[k], snd = y[k + 8];
                            const cplx oth = make_double2(lane_xchg<8>(snd.x), lane_xchg<8>(snd.y));
                            const double P = fma(own.y, own.y, own.x * own.x) + fma(oth.y, oth.y, oth.x * oth.x);
                            double sn, cs;
                            sincos_taylor(-gamleff * P, &sn, &cs);
                            const cplx nl = make_double2(cs, sn);
                            cplx A = cmul(own, nl), B = cmul(oth, nl);
                            if (CNLSE) {
                                // s3 = 2 (Re ux Im uy - Im ux Re uy) (:841-851): on the Y lane own/oth are swapped, the two
                                // products swap and the difference changes sign exactly
                                const double s3 = sgn2 * __dsub_rn(__dmul_rn(A.x, B.y), __dmul_rn(A.y, B.x));
                                double sp, cp;
                                sincos_taylor(div3(gamleff * s3), &sp, &cp);
                                const double sg = sgn * sp, ng = -sg;   // ux' = cp ux + sp uy,  uy' = cp uy - sp ux
                                const cplx A2 = make_double2(cp * A.x + sg * B.x, cp * A.y + sg * B.y);
                                B = make_double2(cp * B.x + ng * A.x, cp * B.y + ng * A.y);
                                A = A2;
                            }
                            const cplx back = make_double2(lane_xchg<8>(B.x), lane_xchg<8>(B.y));
                            y[k] = A;
                            y[k + 8] = back;
                            sched_fence();         // (one sample pair at a time: a lone wave's FP64 rate does not depend on
                        }                          //  interleaving, and the pairs' operands need not all be selected up front)
                    };
                    if (a.manakov) kerr16(std::false_type{}); else kerr16(std::true_type{});
                } else {
                    int tp = t;            // (the tile goes through the exchange buffer in its natural layout)
                    pin(tp);
                    const int ksw = COLX_HSW(tp);
#pragma unroll
                    for (int k = 0; k < 8; k++) s[((j + 16 * k) << 4) + t + ksw] = y[k];
#pragma unroll
                    for (int k = 8; k < 16; k++) s[((j + 16 * k) << 4) + t - ksw] = y[k];
                    lds_barrier();
                    if (isx) kerr_full_range(j, t, gamleff, a.manakov);
                    lds_barrier();
#pragma unroll
                    for (int k = 0; k < 8; k++) y[k] = s[((j + 16 * k) << 4) + t + ksw];
#pragma unroll
                    for (int k = 8; k < 16; k++) y[k] = s[((j + 16 * k) << 4) + t - ksw];
                    lds_barrier();
                }
            }
            // [phase 5] Kerr step
            lvl2_dif256s(y, j, tw, COLX_TWA(t));
#pragma unroll
            for (int k = 0; k < 16; k++) s[((j + 16 * k) << 4) + t] = y[k];
            lds_barrier();
            cplx x[16];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = s[((16 * j + k) << 4) + t];
            lds_barrier();               // the exchange buffer is free: the next tile may land in it ...
            // [phase 6] lvl2_dif + exchange
            {
                const int nf = (int)red[10 + (it & 1)];
                if (nf >= 0) stage(nf, (it & 1) ^ 1);
            }
            r16_dif(x);                    // ... during the last register transform and the stores of this one
            // store_late (what ships: ON for multi-team launches such as C1, OFF where a frame is the whole grid): the tile's stores
            // are held back until this wave's rows of the NEXT tile are in LDS -- requests first, stores while the next tile
            // computes.  Multi-team: the landing no longer shares the workgroup's memory queue with 64 KiB of stores, k_colx16
            // beside the receiver 1170 -> 1138 us (+3...4 %).  One team (2^20-sample frames): the whole grid would wait for the
            // last landing before ANY store is issued, 359 -> 395 us per 16 frames, so the plan leaves it off there
            // (profiles/r03_store_late_ab.txt).
            if (a.store_late) {
                const int nf = (int)red[10 + (it & 1)];
                if (nf >= 0) {
                    FrameCtl *const nrec = lctl + 4 * ((it & 1) ^ 1) + (tid >> 6);
                    while (lds_peek(&nrec->done) == PLX_REC_SENTINEL) nap();
                }
            }
#pragma unroll
            for (int k = 0; k < 16; k++) fld[cbase + (size_t)(16 * j + k) * N2] = x[k];
            // [phase 7] staging issue + r16_dif + stores issued
        }
        f = (int)red[10 + (it & 1)];
    // [stamps:iter]
        if (f < 0) { it++; break; }
    }
    if (tid == 0) settle((it & 1) ^ 1);
    // [stamps:exit]
}


} // namespace

// ================================================================= host side ===
struct plx_ssfm {
    plx_ssfm_desc d;
    int p, p1, p2;
    size_t N;
    SsfmArgs a;
    double *d_betat = nullptr, *d_db1 = nullptr, *d_gam = nullptr, *d_brf = nullptr, *d_psum = nullptr;
    cplx *d_tpass = nullptr, *d_tw1 = nullptr, *d_tw2 = nullptr, *d_ctab = nullptr;
    FrameCtl *d_ctl = nullptr;
    unsigned long long *d_umax = nullptr;
    int *d_ndone = nullptr;   // [0] frames done, [1] abort word, [2] frames in the active list, [3] its running sum over the steps
    int *h_ndone = nullptr;   // pinned copy of the four words
    int *d_active = nullptr;  // [max_frames] active list (k_compact)
    hipEvent_t ev = nullptr;  // completion of the last read-back of d_ndone
    std::vector<FrameCtl> h_ctl;
    int brf_sets = 0;
    size_t lds_col = 0, lds_row = 0;
    cplx *d_e1 = nullptr, *d_e2 = nullptr;   // per-frame, per-trunk row / column phasors of PMD plans with a linear db1 (k_pmd_tab)
    unsigned long long *d_slots = nullptr;   // slot barrier of the fused column sweep: [launch parity][frame][tile]
    unsigned long long *d_mbox = nullptr;    // [teams][frames + 4] mailboxes of the fused column sweep's teams, then the two claim counters
    size_t mbox_bytes = 0;
    int fused = 0, fused_grid = 0, tiles_pf = 0;
    uint32_t flags = 0;                      // plx_ssfm_create_ex
    int barrier_timeouts = 0;                // propagate calls of this plan that ended in a frame-barrier time-out (it then takes the three-sweep step for good)
    double *d_dzlist = nullptr, *d_dzlog = nullptr;   // diagnostics: replayed / logged step sequences
    int dzlist_cap = 0;
    int col_threads = 512;         // workgroup size of k_col_fwd / k_col_inv
    int row_threads = ROW_THREADS; // workgroup size of k_row
    int rowr = 0;                  // k_row256r serves the step's row pass
    int row4k_split = 0;           // k_row4k<false, true>: the same for 4096-point rows
    int rowsm = 0;                 // k_rowsm<p2> serves it (rows of 32 / 64 / 128 points; dual polarisation without PMD, scalar)
    int row256_split = 0;          // k_row256r<true, false, true>: the PMD form with tables at three waves per SIMD
    int rowg_pair_split = 0;       // ... and the PMD form with phasor tables as well (k_rowreg<., true, false, true>)
    int rowg_split = 0;            // ... with the exchanges split into real and imaginary halves (three workgroups per CU)
    int rowreg = 0;                // k_rowreg<p2> serves it (dual polarisation, no PMD, rows of 512 / 1024 / 2048 points)
    cplx *d_tw2c = nullptr, *d_twmid = nullptr;
    int row_split = 0, rs_threads = 0; // long rows without PMD: one polarisation per workgroup (scalar row pass twice)
    size_t rs_lds = 0;
    int tw_compact = 0;            // 4096-point rows: compact twiddle table in d_tw2, register-blocked row pass k_row4k
    int row_pair4k = 0;            // ... of a PMD-type plan: both polarisations of a row in one workgroup (k_row4k<true>)
    size_t rs_lds_pair = 0;
    double *h_brf[2] = {nullptr, nullptr}; // pinned staging of the waveplate tables
    hipEvent_t brf_ev[2] = {nullptr, nullptr};
    int brf_slot = 0;
    int64_t row_launches = 0, sample_steps = 0;
    int64_t slots_launched = 0, slots_listed = 0, frame_steps = 0;   // utilisation accounting of the last propagate
    // optional per-kernel timing of the step loop (plx_ssfm_profile): one event between consecutive launches
    int profile = 0;
    // The intervals are read LATER -- while the next call's first launches run, or when the times are asked for: some 160
    // hipEventElapsedTime calls per propagate would otherwise sit between two calls with the GPU idle (~2 ms per 110 ms).
    struct ProfRun { std::vector<hipEvent_t> ev; std::vector<int> cls, step; int maxnc = 0; bool fused = false; };
    std::vector<hipEvent_t> evfree;          // events not in use
    std::vector<ProfRun> prof_pending;       // finished step loops whose intervals have not been read yet
    double k_ms[4] = {0, 0, 0, 0};           // accumulated since the last plx_ssfm_kernel_times
    int64_t k_launches[4] = {0, 0, 0, 0};
};

static const double kInv2Pi = 0.15915494309189533577;

static int ilog2(int64_t v)
{
    int l = 0;
    while (((int64_t)1 << l) < v) l++;
    return l;
}

static void free_plan(plx_ssfm *P)
{
    if (!P) return;
    hipFree(P->d_betat); hipFree(P->d_db1); hipFree(P->d_gam); hipFree(P->d_brf); hipFree(P->d_psum);
    hipFree(P->d_tpass); hipFree(P->d_tw1); hipFree(P->d_tw2); hipFree(P->d_ctab); hipFree(P->d_ctl); hipFree(P->d_umax);
    hipFree(P->d_dzlist); hipFree(P->d_dzlog); hipFree(P->d_tw2c); hipFree(P->d_twmid);
    hipFree(P->d_ndone); hipFree(P->d_slots); hipFree(P->d_mbox); hipFree(P->d_active); hipFree(P->d_e1); hipFree(P->d_e2);
    if (P->h_ndone) hipHostFree(P->h_ndone);
    if (P->ev) hipEventDestroy(P->ev);
    for (hipEvent_t e : P->evfree) hipEventDestroy(e);
    for (auto &r : P->prof_pending) for (hipEvent_t e : r.ev) hipEventDestroy(e);
    for (int k = 0; k < 2; k++) {
        if (P->h_brf[k]) hipHostFree(P->h_brf[k]);
        if (P->brf_ev[k]) hipEventDestroy(P->brf_ev[k]);
    }
    delete P;
}

static void half_table(std::vector<cplx> &t, int M)
{
    t.resize(M / 2 > 0 ? M / 2 : 1);
    for (int k = 0; k < M / 2; k++) {
        long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)M;
        t[k] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
}

#ifndef PLX_EMU
template <class K> static hipError_t allow_lds(K kern, size_t bytes)
{
    return hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
// workgroups of `kern` the runtime admits per CU at this block size and dynamic LDS (0 on failure)
template <class K> static int blocks_per_cu(K kern, int threads, size_t lds)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, threads, lds) != hipSuccess) return 0;
    return nb;
}
#else
template <class K> static hipError_t allow_lds(K, size_t) { return hipSuccess; }
template <class K> static int blocks_per_cu(K, int, size_t) { return 2; }
#endif

// Plan-time overrides, read ONCE in plx_ssfm_create: the geometry switches the tests use to reach other splits and kernels
// (P1, LOGW, COL_THREADS, NO_ROW_SPLIT, SHORT_ROWS, ROWR, NO_PMD_TAB), A/B switches of shipped choices (STORE_LATE, ROW_REV,
// SAFE_LANDING) and the barrier time-out.  PLX_SSFM_NO_FUSE=1 = plx_ssfm_create_ex(..., PLX_SSFM_SHARE_DEVICE) for a whole
// process (barrier-free three-sweep step, e.g. when several processes share a GPU).  The switches of experiments that were
// not adopted (working copy, frame groups, grid sizing: profiles/r03_notes.md) are gone with their code.
namespace {
struct Tune {
    int short_rows = 0, no_fuse = 0, p1 = -1, logW = -1, col_threads = -1, no_row_split = 0, safe_landing = 0, no_pmd_tab = 0, rowr = 0, store_late = -1, row_rev = 0, rowg_split = 1, row4k_split = 1, rowsm = 1, row256_split = 1;
    double barrier_timeout_ms = 500.0;
    static int geti(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
    Tune()
    {
        no_fuse = geti("PLX_SSFM_NO_FUSE", 0);
        short_rows = geti("PLX_SSFM_SHORT_ROWS", 0);   // (A/B, tests) 2^20-sample frames on the 512 x 2048 split instead of 256 x 4096
        p1 = geti("PLX_SSFM_P1", -1);
        logW = geti("PLX_SSFM_LOGW", -1);
        col_threads = geti("PLX_SSFM_COL_THREADS", -1);
        no_row_split = geti("PLX_SSFM_NO_ROW_SPLIT", 0);
        safe_landing = geti("PLX_SSFM_SAFE_LANDING", 0);
        store_late = geti("PLX_SSFM_STORE_LATE", -1);  // fused sweep: stores after the next tile's landing; -1: where a launch has more than one team
        row_rev = geti("PLX_SSFM_ROW_REV", 1);         // 0: the row pass takes the listed frames in ascending order as well (A/B)
        row256_split = geti("PLX_SSFM_ROW256_SPLIT", 1);  // 0: k_row256r<PMD> with both trunk forms and whole-sample exchanges also for plans with phasor tables (A/B, tests)
        rowsm = geti("PLX_SSFM_ROWSM", 1);              // 0: k_row for rows of 32 / 64 / 128 points everywhere; 2: k_rowsm wherever it applies (tests)
        row4k_split = geti("PLX_SSFM_ROW4K_SPLIT", 1);  // 0: k_row4k's whole-sample exchanges (two workgroups per CU; A/B, tests)
        rowg_split = geti("PLX_SSFM_ROWG_SPLIT", 1);   // 0: k_rowreg's whole-sample exchanges (two workgroups per CU; A/B, tests)
        rowr = geti("PLX_SSFM_ROWR", 1);               // 0: the LDS-resident k_row also where the register forms k_row256r / k_rowreg apply (A/B, tests)
        no_pmd_tab = geti("PLX_SSFM_NO_PMD_TAB", 0);   // PMD plans: one exponential per bin and trunk instead of the phasor tables (A/B, tests)
        if (const char *e = getenv("PLX_SSFM_BARRIER_TIMEOUT_MS")) barrier_timeout_ms = atof(e);
    }
};
bool pow2_in(int v, int lo, int hi) { return v >= lo && v <= hi && (v & (v - 1)) == 0; }
} // namespace

extern "C" int plx_ssfm_create(plx_ssfm **out, const plx_ssfm_desc *desc) { return plx_ssfm_create_ex(out, desc, 0u); }

extern "C" int plx_ssfm_create_ex(plx_ssfm **out, const plx_ssfm_desc *desc, uint32_t flags)
{
    if (!out || !desc) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create: null argument");
    if (flags & ~(uint32_t)PLX_SSFM_SHARE_DEVICE) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create_ex: unknown flag");
    *out = nullptr;
    const int64_t N = desc->nfft;
    const int p = ilog2(N);
    if (N < 256 || ((int64_t)1 << p) != N || p > 20)
        PLX_FAIL(PLX_ERR_UNSUPPORTED, "plx_ssfm_create: nfft must be a power of two in [256, 2^20]");
    if (desc->nfc < 1 || desc->max_frames < 1) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create: nfc and max_frames must be >= 1");
    if (!desc->gam || !desc->betat) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create: gam and betat are required");
    if (desc->dual_pol && desc->fls[3] && desc->nfc > 1)
        PLX_FAIL(PLX_ERR_REFERENCE, "The CNLSE with separate fields is not yet implemented"); // fiber.m:854
    if (desc->dual_pol && desc->fls[3] && desc->nfc == 1) { /* xpm flag is forced to 0 for one field, :224 */ }
    if (desc->nplates < 1) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create: nplates must be >= 1");
    Tune tune;
    if (flags & PLX_SSFM_SHARE_DEVICE) tune.no_fuse = 1;      // the barrier-free three-sweep step: no co-residency requirement

    plx_ssfm *P = new plx_ssfm();
    P->d = *desc;
    P->flags = flags;
    P->p = p;
    // Four-step split N = N1 x N2.  The column tile is N1 rows x T complex (T = W columns per
    // polarisation x npol) and is kept at <= 64 KiB so two workgroups share a CU; a wider, shorter
    // tile means longer contiguous row segments in HBM (W*16 B per polarisation).  The row pass
    // holds npol x N2 complex (+ twiddles) in LDS, which bounds N2 at 2048 for dual-pol frames.
    int logW;
    {
        const int npol = desc->dual_pol ? 2 : 1;
        logW = desc->dual_pol ? 3 : 4;                       // 8 (dual) / 16 (scalar) columns per tile (measured best)
        int p1 = 12 - (logW + (npol == 2 ? 1 : 0));          // N1 * T = 4096 complex = 64 KiB
        // 2^20-sample dual-polarisation frames -- one field or several 'sepfields' channels, fused or three-sweep step, with or
        // without PMD -- keep the 256-row column tile and take 4096-point rows instead (k_row4k, compact twiddle table: one
        // polarisation per row workgroup and two workgroups per CU, or with PMD both polarisations in one workgroup of twice
        // the size); everything else stops at 2048-point rows and gets taller column tiles
        // (scalar plans take the same split: k_row4k on the rows of the one field)
        const bool long_rows = !tune.no_row_split && !tune.short_rows;
        const int p2max = long_rows ? 12 : 11;
        if (p - p1 > p2max) p1 = p - p2max;                  // large frames: taller tiles instead
        if (p1 > p - 4) p1 = p - 4;                          // keep N2 >= 16
        if (p1 < 2) p1 = 2;
        while ((((size_t)1 << p1) << (logW + (npol == 2 ? 1 : 0))) * sizeof(cplx) > 128 * 1024 && logW > 3) logW--;
        if (tune.p1 >= 2 && tune.p1 <= p - 4) p1 = tune.p1;
        if (tune.logW >= 2 && tune.logW <= 6) logW = tune.logW;
        while (((int64_t)1 << (p - p1)) < ((int64_t)1 << logW)) logW--;  // tile not wider than a row
        P->p1 = p1;
    }
    P->p2 = p - P->p1;
    P->N = (size_t)N;
    const int N1 = 1 << P->p1, N2 = 1 << P->p2;
    const int nfc = desc->nfc, F = desc->max_frames;
    SsfmArgs &a = P->a;
    std::memset(&a, 0, sizeof(a));
    a.p1 = P->p1; a.p2 = P->p2; a.nfc = nfc; a.dual = desc->dual_pol ? 1 : 0;
    a.logW = logW; a.W = 1 << a.logW;
    a.logT = a.logW + (a.dual ? 1 : 0); a.T = 1 << a.logT;
    // rows per workgroup in the row pass: >= one 16-point register block per thread
    {
        int R = 1, npol = a.dual ? 2 : 1;
        while (R * npol * (N2 / 16) < ROW_THREADS / 2 && R * 2 <= N1) R *= 2;   // measured: 2 rows x 2 pols at N2 = 256
        a.R = R; a.logR = ilog2(R);
        // row pass: ~8 points per thread (128 threads for 2 rows x 2 polarisations x 256 points); long rows leave room
        // for only one or two workgroups per CU, so those get proportionally more waves (up to 1024 threads)
        int rowthr = ROW_THREADS;
        const int64_t pts = (int64_t)npol * R * N2;
        while (rowthr < 1024 && (int64_t)rowthr * 8 < pts) rowthr *= 2;
        P->row_threads = rowthr;
    }
    // Long rows leave room for a single dual-polarisation workgroup per CU.  Without PMD the two polarisations only
    // share the multiplier, so each gets its own workgroup (the scalar form of the row pass, R = 1): half the LDS,
    // 2-3 workgroups per CU.
    P->tw_compact = P->p2 >= 12 ? 1 : 0;
    if (a.dual && (N2 >= 2048 && !tune.no_row_split)) {   // measured: 2^20 frames 74 -> 66 ms; at N2 = 1024 it loses (47 -> 52)
        P->row_split = 1;
        P->rs_threads = N2 / 8 < ROW_THREADS ? ROW_THREADS : (N2 / 8 > 1024 ? 1024 : N2 / 8);
        P->rs_lds = ((size_t)(N2 + N2 / 16) + (P->tw_compact ? N2 / 8 + 4 + 16 + 160 + PLX_CTAB : N2 / 2)) * sizeof(cplx);   // (+16: k_row4k's bk, +160: its padded W_256 table, + the unit-circle table)
    }
    if (P->tw_compact && !a.dual && !tune.no_row_split) P->rs_lds = ((size_t)(N2 + N2 / 16) + N2 / 8 + 4 + 16 + 160 + PLX_CTAB) * sizeof(cplx);
    if (P->tw_compact && !P->row_split && a.dual) {
        free_plan(P);
        PLX_FAIL(PLX_ERR_UNSUPPORTED, "plx_ssfm_create: 4096-point rows need the one-polarisation row pass (PLX_SSFM_NO_ROW_SPLIT is set)");
    }
    if (P->tw_compact) {
        P->row_pair4k = desc->fls[1] ? 1 : 0;
        P->rs_lds_pair = P->rs_lds + (size_t)(N2 + N2 / 16) * sizeof(cplx);
    }
    a.spm = desc->fls[2]; a.xpm = desc->fls[3]; a.manakov = desc->manakov ? 1 : 0; a.pmd = desc->fls[1] ? 1 : 0;
    a.nplates = desc->nplates;
    a.alphalin = desc->alphalin; a.Lf = desc->length; a.dzmax = desc->dzmaxt; a.dphimax = desc->dphimaxt;
    a.lcorr = desc->length / desc->nplates; // fiber.m:507
    a.invN = 1.0 / (double)N;
    a.spin_ticks = (long long)(tune.barrier_timeout_ms * 1e5);
    a.safe_land = tune.safe_landing;
    a.row_rev = tune.row_rev;
    // (the mailbox entries of k_colx16 pack frame + 1 and iteration + 1 into 22-bit fields)
    if ((int64_t)desc->max_frames + 4 >= ((int64_t)1 << 22)) { free_plan(P); PLX_FAIL(PLX_ERR_UNSUPPORTED, "plx_ssfm_create: max_frames must be below 2^22 - 4"); }

    // ---- tables: spectral multipliers in the order the row pass sees them ----
    std::vector<double> bt((size_t)nfc * N), d1;
    std::vector<cplx> tp((size_t)N);
    const bool have_db1 = desc->db1 != nullptr && a.dual;
    if (have_db1) d1.resize((size_t)nfc * N);
    for (int j = 0; j < N1; j++) {
        const unsigned k1 = plx_bitrev((unsigned)j, P->p1);
        for (int i = 0; i < N2; i++) {
            const unsigned k2 = plx_bitrev((unsigned)i, P->p2);
            const size_t k = (size_t)k1 + (size_t)N1 * k2, pos = (size_t)j * N2 + i;
            for (int c = 0; c < nfc; c++) {
                // phases are kept in TURNS (rad / 2 pi) for the exact range reduction of cexp_neg_turns
                bt[(size_t)c * N + pos] = desc->betat[(size_t)c * N + k] * kInv2Pi;
                if (have_db1) d1[(size_t)c * N + pos] = desc->db1[(size_t)c * N + k] * kInv2Pi;
            }
            const uint64_t e = ((uint64_t)i * k1) & (uint64_t)(N - 1); // n2*k1 mod N
            long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)e / (long double)N;
            tp[pos] = make_double2((double)cosl(ang), (double)sinl(ang));
        }
    }
    std::vector<cplx> t1, t2;
    half_table(t1, N1);
    if (P->tw_compact) {   // W_N2^{4k}, k < N2/8, then W_N2^0..3 (plx_fft.h, Tw4096)
        t2.resize(N2 / 8 + 4);
        for (int k = 0; k < N2 / 8 + 4; k++) {
            const int e = k < N2 / 8 ? 4 * k : k - N2 / 8;
            long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)e / (long double)N2;
            t2[k] = make_double2((double)cosl(ang), (double)sinl(ang));
        }
    } else {
        half_table(t2, N2);
    }
    std::vector<double> gam(nfc);
    for (int c = 0; c < nfc; c++) gam[c] = (a.dual && a.manakov) ? desc->gam[c] * 8 / 9 : desc->gam[c]; // :499-501

#define UP(dst, vec, T)                                                                             \
    do {                                                                                            \
        if (hipMalloc((void **)&(dst), (vec).size() * sizeof(T)) != hipSuccess ||                   \
            hipMemcpy((dst), (vec).data(), (vec).size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { \
            free_plan(P);                                                                           \
            PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_create: device allocation/upload failed");              \
        }                                                                                           \
    } while (0)
    UP(P->d_betat, bt, double);
    if (have_db1) UP(P->d_db1, d1, double);
    UP(P->d_tpass, tp, cplx);
    UP(P->d_tw1, t1, cplx);
    UP(P->d_tw2, t2, cplx);
    {
        std::vector<cplx> ctv(PLX_CTAB);
        for (int k = 0; k < PLX_CTAB; k++) {
            const long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)PLX_CTAB;
            ctv[k] = make_double2((double)cosl(ang), (double)-sinl(ang));
        }
        UP(P->d_ctab, ctv, cplx);
    }
    // k_rowsm: register-form row pass for rows of 32 / 64 / 128 points (one wave = 64 / R row-polarisations)
    // (measured, fraction of 8 TB/s: scalar plans 0.55 / 0.67 / 0.68 at 32 / 64 / 128 points against k_row's 0.49 / 0.50 / 0.45;
    //  dual-polarisation plans 0.64 / 0.69 / 0.70 against 0.70 / 0.70 / 0.60 -- k_row's wider workgroups win the short dual rows,
    //  so those take it at 128 points only; PLX_SSFM_ROWSM=2 forces it wherever it applies: tests)
    const int rowsm_min = a.dual ? (tune.rowsm == 2 ? 5 : 7) : 5;
    if (tune.rowr && tune.rowsm && !desc->fls[1] && P->p2 >= rowsm_min && P->p2 <= 7 && (N1 * (a.dual ? 2 : 1)) % (64 / (N2 / 16)) == 0) {
        const long double tau = -2.0L * 3.14159265358979323846264338327950288L;
        std::vector<cplx> tm(7 * 16, make_double2(1.0, 0.0));
        const int R = N2 / 16;
        auto put = [&](int q, int j, int e, int m) { tm[16 * q + j] = make_double2((double)cosl(tau * e / m), (double)sinl(tau * e / m)); };
        for (int j = 0; j < 16; j++) {
            if (R == 2) put(0, j, j, 32);
            const int q0 = R == 8 ? 4 : 0;
            if (R >= 4) for (int q = 0; q < 3; q++) put(q0 + q, j, (q + 1) * j, 64);
            if (R == 8) for (int q = 0; q < 4; q++) put(q, j, j + 16 * q, 128);
        }
        UP(P->d_twmid, tm, cplx);
        hipError_t e = hipSuccess;
        if (a.dual) e = P->p2 == 5 ? allow_lds(k_rowsm<5, false>, ROWSM_LDS) : P->p2 == 6 ? allow_lds(k_rowsm<6, false>, ROWSM_LDS) : allow_lds(k_rowsm<7, false>, ROWSM_LDS);
        else e = P->p2 == 5 ? allow_lds(k_rowsm<5, true>, ROWSM_LDS) : P->p2 == 6 ? allow_lds(k_rowsm<6, true>, ROWSM_LDS) : allow_lds(k_rowsm<7, true>, ROWSM_LDS);
        if (e == hipSuccess) P->rowsm = 1;
    }
    // k_rowreg: register-form row pass for dual-polarisation plans without PMD whose rows have 512, 1024 or 2048 points
    if (tune.rowr && P->p2 >= 9 && P->p2 <= 11 && N1 >= (ROWG_THREADS / (N2 / 16)) / (a.dual ? 2 : 1)) {
        const long double tau = -2.0L * 3.14159265358979323846264338327950288L;
        std::vector<cplx> tc(N2 / 8 + 4), tm(7 * 16, make_double2(1.0, 0.0));
        for (int k = 0; k < N2 / 8 + 4; k++) {
            const int e = k < N2 / 8 ? 4 * k : k - N2 / 8;
            tc[k] = make_double2((double)cosl(tau * e / N2), (double)sinl(tau * e / N2));
        }
        const int R = N2 / 256;
        auto put = [&](int q, int j, int e, int m) { tm[16 * q + j] = make_double2((double)cosl(tau * e / m), (double)sinl(tau * e / m)); };
        for (int j = 0; j < 16; j++) {
            if (R == 2) put(0, j, j, 32);
            const int q0 = R == 8 ? 4 : 0;
            if (R >= 4) for (int q = 0; q < 3; q++) put(q0 + q, j, (q + 1) * j, 64);
            if (R == 8) for (int q = 0; q < 4; q++) put(q, j, j + 16 * q, 128);
        }
        UP(P->d_tw2c, tc, cplx);
        UP(P->d_twmid, tm, cplx);
        hipError_t e = P->p2 == 9 ? allow_lds(k_rowreg<9, false>, ROWG_LDS(512)) : P->p2 == 10 ? allow_lds(k_rowreg<10, false>, ROWG_LDS(1024)) : allow_lds(k_rowreg<11, false>, ROWG_LDS(2048));
        if (e == hipSuccess) e = P->p2 == 9 ? allow_lds(k_rowreg<9, true>, ROWG_LDS(512)) : P->p2 == 10 ? allow_lds(k_rowreg<10, true>, ROWG_LDS(1024)) : allow_lds(k_rowreg<11, true>, ROWG_LDS(2048));
        if (e == hipSuccess) e = P->p2 == 9 ? allow_lds(k_rowreg<9, false, true>, ROWG_LDS(512)) : P->p2 == 10 ? allow_lds(k_rowreg<10, false, true>, ROWG_LDS(1024)) : allow_lds(k_rowreg<11, false, true>, ROWG_LDS(2048));
        if (e == hipSuccess) P->rowreg = 1;
        // rows of 512 / 1024 points without PMD: the exchanges in real / imaginary halves, three workgroups per CU
        // (PLX_SSFM_ROWG_SPLIT=0: the whole-sample exchange, A/B and tests)
        if (P->rowreg && tune.rowg_split) {
            const hipError_t e2 = a.dual ? (P->p2 == 9 ? allow_lds(k_rowreg<9, false, false, true>, ROWG_LDS_SPLIT(512)) : P->p2 == 10 ? allow_lds(k_rowreg<10, false, false, true>, ROWG_LDS_SPLIT(1024))
                                                                                                                                        : allow_lds(k_rowreg<11, false, false, true>, ROWG_LDS_SPLIT(2048)))
                                         : (P->p2 == 9 ? allow_lds(k_rowreg<9, false, true, true>, ROWG_LDS_SPLIT(512)) : P->p2 == 10 ? allow_lds(k_rowreg<10, false, true, true>, ROWG_LDS_SPLIT(1024))
                                                                                                                                       : allow_lds(k_rowreg<11, false, true, true>, ROWG_LDS_SPLIT(2048)));
            if (e2 == hipSuccess) P->rowg_split = 1;
            if (P->rowg_split && a.dual &&
                (P->p2 == 9 ? allow_lds(k_rowreg<9, true, false, true>, ROWG_LDS_SPLIT(512)) : P->p2 == 10 ? allow_lds(k_rowreg<10, true, false, true>, ROWG_LDS_SPLIT(1024))
                                                                                                             : allow_lds(k_rowreg<11, true, false, true>, ROWG_LDS_SPLIT(2048))) == hipSuccess)
                P->rowg_pair_split = 1;
        }
    }
    UP(P->d_gam, gam, double);
#undef UP
    bool ok = hipMalloc((void **)&P->d_ctl, sizeof(FrameCtl) * F) == hipSuccess &&
              hipMalloc((void **)&P->d_umax, sizeof(unsigned long long) * F * nfc) == hipSuccess &&
              hipMalloc((void **)&P->d_ndone, 64) == hipSuccess &&
              hipMalloc((void **)&P->d_active, sizeof(int) * (size_t)F) == hipSuccess &&
              hipHostMalloc((void **)&P->h_ndone, 64, hipHostMallocDefault) == hipSuccess &&
              hipEventCreateWithFlags(&P->ev, hipEventDisableTiming) == hipSuccess;
    if (ok && !a.dual && a.xpm) ok = hipMalloc((void **)&P->d_psum, sizeof(double) * (size_t)F * N) == hipSuccess;
    if (!ok) { free_plan(P); PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_create: device allocation failed"); }
    a.betat_p = P->d_betat; a.db1_p = P->d_db1; a.tpass = P->d_tpass; a.tw1 = P->d_tw1; a.tw2 = P->d_tw2; a.ctab = P->d_ctab; a.tw2c = P->d_tw2c; a.twmid = P->d_twmid;
    a.gam = P->d_gam; a.ctl = P->d_ctl; a.umax = P->d_umax; a.ndone = P->d_ndone; a.psum = P->d_psum;
    P->h_ctl.resize(F);

    P->lds_col = (((size_t)N1 << a.logT) + N1 / 2) * sizeof(cplx) + 32 * sizeof(double) + 8 * sizeof(FrameCtl) + 128 + COLX_NFC * sizeof(double) + 128 * sizeof(cplx);   // (128: CtrlK; 128 cplx: k_colx16's negated W_256 table)
    // [stamps:lds]
    P->lds_row = ((size_t)(a.dual ? 2 : 1) * a.R * (N2 + N2 / 16) + N2 / 2) * sizeof(cplx);
    P->col_threads = P->lds_col > 80 * 1024 ? 1024 : 512;   // measured: 512-thread column workgroups (2 per CU, 16 waves) beat
                                                            // 256 by 3-12 %; tall tiles of large frames: one workgroup per CU, 16 waves
    if (tune.col_threads == 128 || tune.col_threads == 256 || tune.col_threads == 512 || tune.col_threads == 1024) P->col_threads = tune.col_threads;
    if (allow_lds(k_colx16<true>, P->lds_col) != hipSuccess || allow_lds(k_colx16<false>, P->lds_col) != hipSuccess || allow_lds(k_col_fwd, P->lds_col) != hipSuccess ||
        allow_lds(k_col_inv, P->lds_col) != hipSuccess ||
        (!P->tw_compact && allow_lds(k_row, P->lds_row > P->rs_lds ? P->lds_row : P->rs_lds) != hipSuccess) ||
        (P->tw_compact && (allow_lds(k_row4k<false>, P->rs_lds) != hipSuccess || allow_lds(k_row4k<true>, P->rs_lds_pair) != hipSuccess ||
                           allow_lds(k_row4k<false, true>, P->rs_lds) != hipSuccess))) {
        free_plan(P);
        PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_create: cannot reserve LDS for the transform kernels");
    }
    P->row4k_split = (P->tw_compact && tune.row4k_split) ? 1 : 0;
    if (tune.rowr && a.dual && a.p1 == 8 && a.p2 == 8 && !P->row_split &&
        (a.pmd ? allow_lds(k_row256r<true>, ROWR_LDS) : allow_lds(k_row256r<false>, ROWR_LDS)) == hipSuccess) P->rowr = 1;
    if (tune.rowr && !a.dual && a.p1 == 8 && a.p2 == 8 && allow_lds(k_row256r<false, true>, ROWR_LDS_SC) == hipSuccess) P->rowr = 1;
    if (P->rowr && a.dual && a.pmd && tune.row256_split && allow_lds(k_row256r<true, false, true>, ROWR_LDS) == hipSuccess) P->row256_split = 1;
    // Fused column sweep (k_colx16): the inverse column pass of step s, the step controller and the forward column
    // pass of step s+1 in ONE launch on a register/LDS-resident tile (2 sweeps over HBM per step instead of 3), for
    // dual-polarisation plans with 256 x (8+8) column tiles.  The tiles of a frame meet at a barrier inside the
    // launch, so all of them must be resident together: the grid is sized from the runtime's own occupancy answer
    // for this kernel (block size and dynamic LDS as launched), a multiple of the tiles per frame; a plan whose
    // frame does not fit the chip that way takes the barrier-free three-sweep step.
    // (scalar plans: the same sweep on sixteen columns of the one field -- not with XPM, whose Kerr step needs the other
    //  channels' powers at the same sample, i.e. other workgroups' tiles)
    if (((a.dual && a.W == 8) || (!a.dual && a.W == 16 && !desc->fls[3])) && !tune.no_fuse && a.p1 == 8 && nfc <= COLX_NFC) {
        int ncu = 256;
        {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
                ncu = v;
        }
        const int tiles_pf = nfc * (N2 / a.W);
        const int per_cu = a.dual ? blocks_per_cu(k_colx16<true>, 256, P->lds_col) : blocks_per_cu(k_colx16<false>, 256, P->lds_col);
        const int cap = ncu * per_cu;
        if (tiles_pf <= cap) {
            P->fused = 1;
            P->tiles_pf = tiles_pf;
            P->fused_grid = (cap / tiles_pf) * tiles_pf;
            a.store_late = tune.store_late >= 0 ? (tune.store_late ? 1 : 0) : (P->fused_grid > tiles_pf ? 1 : 0);
            const int mstride = F + 4;       // (iterations of a team in a launch <= frames listed; its first workgroup posts two ahead)
            P->mbox_bytes = sizeof(unsigned long long) * ((size_t)mstride * (P->fused_grid / tiles_pf) + 1);
            if (hipMalloc((void **)&P->d_slots, sizeof(unsigned long long) * 2 * (size_t)F * tiles_pf) != hipSuccess ||
                hipMalloc((void **)&P->d_mbox, P->mbox_bytes) != hipSuccess) {
                free_plan(P);
                PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_create: device allocation failed");
            }
            a.slots = P->d_slots;
            a.mbox = P->d_mbox;
            a.mbox_stride = mstride;
            a.grab = (int *)(P->d_mbox + (size_t)mstride * (P->fused_grid / tiles_pf));
        }
    }
    // PMD plans: is db1 linear in the signed frequency index (fiber.m:358)?  Then the trunk phases factor into row x column
    // phasors (SsfmArgs::e1tab) and the row pass does one complex product per bin and trunk instead of an exponential.
    if (a.pmd && have_db1 && !tune.no_pmd_tab) {
        const double D = desc->db1[1] * kInv2Pi;       // turns per unit of m (k = 1 <-> m = 1)
        bool lin = D != 0.0 && N >= 4;
        for (int c = 0; c < nfc && lin; c++)
            for (int64_t k = 0; k < N && lin; k++) {
                const double m = (double)(k < N / 2 ? k : k - N);
                if (fabs(desc->db1[(size_t)c * N + k] * kInv2Pi - D * m) > 4e-15 * fabs(D) * (double)N) lin = false;
            }
        if (lin) {
            const int tmax = (int)ceil(desc->dzmaxt / a.lcorr) + 2;
            const size_t n1 = (size_t)F * tmax * N1, n2 = (size_t)F * tmax * N2;
            if (tmax <= 64 && (n1 + n2) * sizeof(cplx) <= ((size_t)4 << 30)) {       // (bounded: 27 trunks at 100 plates and dzmax = L / 4)
                if (hipMalloc((void **)&P->d_e1, n1 * sizeof(cplx)) != hipSuccess || hipMalloc((void **)&P->d_e2, n2 * sizeof(cplx)) != hipSuccess) {
                    free_plan(P);
                    PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_create: device allocation failed (trunk phasor tables)");
                }
                a.e1tab = P->d_e1; a.e2tab = P->d_e2; a.d1slope = D; a.tmax = tmax;
            }
        }
    }
    if (a.pmd && !a.dual) { free_plan(P); PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create: PMD needs a dual-polarisation plan"); }
    if (!a.pmd) { // fiber.m:291-297: birefringence off
        double z = 0;
        int rc = plx_ssfm_set_birefringence(P, &z, &z, &z, 1);
        if (rc) { free_plan(P); return rc; }
    }
    *out = P;
    return PLX_OK;
}

extern "C" int plx_ssfm_destroy(plx_ssfm *P)
{
    free_plan(P);
    return PLX_OK;
}

// Waveplate tables.  Device storage is sized once for max_frames sets; uploads go through two pinned staging
// slots and are stream-ordered (the copy lands after whatever propagate call is still reading the old table on
// that stream and before the next one), so a Monte-Carlo loop can draw fresh birefringence for batch i+1 while
// batch i is still in flight elsewhere on the GPU -- no device-wide synchronisation.
static int set_brf(plx_ssfm *P, const double *db0, const double *theta, const double *epsilon, int nsets, hipStream_t st,
                   bool wait)
{
    if (!P || !db0 || !theta || !epsilon) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_set_birefringence: null argument");
    if (nsets < 1 || (nsets != 1 && nsets > P->d.max_frames)) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_set_birefringence: more sets than frames");
    const int np = P->d.nplates;
    const size_t cap = (size_t)P->d.max_frames * np * BRF_STRIDE, cnt = (size_t)nsets * np * BRF_STRIDE;
    if (!P->d_brf) PLX_HIP(hipMalloc((void **)&P->d_brf, cap * sizeof(double)));
    const int slot = P->brf_slot;
    P->brf_slot ^= 1;
    if (!P->h_brf[slot]) {
        PLX_HIP(hipHostMalloc((void **)&P->h_brf[slot], cap * sizeof(double), hipHostMallocDefault));
        PLX_HIP(hipEventCreateWithFlags(&P->brf_ev[slot], hipEventDisableTiming));
    } else {
        PLX_HIP(hipEventSynchronize(P->brf_ev[slot]));   // the upload that last used this slot has executed
    }
    double *t = P->h_brf[slot];
    for (int sidx = 0; sidx < nsets; sidx++)
        for (int n = 0; n < np; n++) {
            const size_t i = (size_t)sidx * np + n;
            // matR = matRth*matRepsilon, fiber.m:910-912; the kernel needs S = matR * sigma3 * matR' (see pmd_trunks)
            const double ct = cos(theta[i]), sn = sin(theta[i]), ce = cos(epsilon[i]), se = sin(epsilon[i]);
            const double r11x = ct * ce, r11y = -sn * se, r12x = -sn * ce, r12y = ct * se;
            const double r21x = sn * ce, r21y = ct * se, r22x = ct * ce, r22y = sn * se;
            double *m = &t[i * BRF_STRIDE];
            m[0] = (r11x * r11x + r11y * r11y) - (r12x * r12x + r12y * r12y);                       // S11 = |R11|^2 - |R12|^2
            m[1] = (r11x * r21x + r11y * r21y) - (r12x * r22x + r12y * r22y);                       // S12 = R11 conj(R21) - R12 conj(R22)
            m[2] = (r11y * r21x - r11x * r21y) - (r12y * r22x - r12x * r22y);
            m[3] = db0[i] * kInv2Pi; // turns, like betat_p / db1_p
        }
    PLX_HIP(hipMemcpyAsync(P->d_brf, t, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    PLX_HIP(hipEventRecord(P->brf_ev[slot], st));
    if (wait) PLX_HIP(hipStreamSynchronize(st));
    P->a.brf = P->d_brf;
    P->a.brf_per_frame = nsets > 1 ? 1 : 0;
    P->brf_sets = nsets;
    return PLX_OK;
}

extern "C" int plx_ssfm_set_birefringence(plx_ssfm *P, const double *db0, const double *theta,
                                          const double *epsilon, int nsets)
{
    return set_brf(P, db0, theta, epsilon, nsets, nullptr, true);
}

extern "C" int plx_ssfm_set_birefringence_dev(plx_ssfm *P, const double *db0, const double *theta,
                                              const double *epsilon, int nsets, void *stream)
{
    return set_brf(P, db0, theta, epsilon, nsets, (hipStream_t)stream, false);
}

// The row pass of one step / filter pass: one launch over both polarisations, or -- long rows without PMD -- the
// one-polarisation form twice (the polarisations only share the multiplier there).
static void launch_row(plx_ssfm *P, const SsfmArgs &a, unsigned FC, hipStream_t st)
{
    const int N1 = 1 << a.p1;
    if (P->tw_compact && !a.dual) {              // scalar plan, 4096-point rows: one workgroup per row and frame-channel
        if (P->row4k_split) PLX_LAUNCH((k_row4k<false, true>), dim3((unsigned)N1 * FC), dim3(256), P->rs_lds - 4352 * sizeof(double), st, a);
        else PLX_LAUNCH(k_row4k<false>, dim3((unsigned)N1 * FC), dim3(256), P->rs_lds, st, a);
        return;
    }
    if (P->tw_compact && a.dual && (a.pmd || a.umat)) {      // the multiplier couples the polarisations: both rows in one workgroup
        PLX_LAUNCH(k_row4k<true>, dim3((unsigned)N1 * FC), dim3(512), P->rs_lds_pair, st, a);
        return;
    }
    if (P->rowsm && !a.pmd && !a.umat) {
        const dim3 g((unsigned)(N1 * (a.dual ? 2 : 1) / (64 / ((1 << a.p2) / 16))), FC), bs(64);
        if (a.dual) {
            if (a.p2 == 5) PLX_LAUNCH((k_rowsm<5, false>), g, bs, ROWSM_LDS, st, a);
            else if (a.p2 == 6) PLX_LAUNCH((k_rowsm<6, false>), g, bs, ROWSM_LDS, st, a);
            else PLX_LAUNCH((k_rowsm<7, false>), g, bs, ROWSM_LDS, st, a);
        } else {
            if (a.p2 == 5) PLX_LAUNCH((k_rowsm<5, true>), g, bs, ROWSM_LDS, st, a);
            else if (a.p2 == 6) PLX_LAUNCH((k_rowsm<6, true>), g, bs, ROWSM_LDS, st, a);
            else PLX_LAUNCH((k_rowsm<7, true>), g, bs, ROWSM_LDS, st, a);
        }
        return;
    }
    if (P->rowreg && !a.dual) {                  // scalar plan: every row-polarisation of the workgroup is a row
        const dim3 g((unsigned)(N1 / (ROWG_THREADS / ((1 << a.p2) / 16))), FC), bs(ROWG_THREADS);
        if (P->rowg_split && a.p2 == 9) PLX_LAUNCH((k_rowreg<9, false, true, true>), g, bs, ROWG_LDS_SPLIT(512), st, a);
        else if (P->rowg_split && a.p2 == 10) PLX_LAUNCH((k_rowreg<10, false, true, true>), g, bs, ROWG_LDS_SPLIT(1024), st, a);
        else if (P->rowg_split && a.p2 == 11) PLX_LAUNCH((k_rowreg<11, false, true, true>), g, bs, ROWG_LDS_SPLIT(2048), st, a);
        else if (a.p2 == 9) PLX_LAUNCH((k_rowreg<9, false, true>), g, bs, ROWG_LDS(512), st, a);
        else if (a.p2 == 10) PLX_LAUNCH((k_rowreg<10, false, true>), g, bs, ROWG_LDS(1024), st, a);
        else PLX_LAUNCH((k_rowreg<11, false, true>), g, bs, ROWG_LDS(2048), st, a);
        return;
    }
    if (P->rowreg && a.dual) {
        const unsigned gx = (unsigned)(N1 / ((ROWG_THREADS / ((1 << a.p2) / 16)) / 2));
        const dim3 g(gx, FC), bs(ROWG_THREADS);
        if (a.pmd && !a.umat && a.e1tab && P->rowg_split && P->rowg_pair_split) {      // ... with phasor tables: the three-waves-per-SIMD form
            if (a.p2 == 9) PLX_LAUNCH((k_rowreg<9, true, false, true>), g, bs, ROWG_LDS_SPLIT(512), st, a);
            else if (a.p2 == 10) PLX_LAUNCH((k_rowreg<10, true, false, true>), g, bs, ROWG_LDS_SPLIT(1024), st, a);
            else PLX_LAUNCH((k_rowreg<11, true, false, true>), g, bs, ROWG_LDS_SPLIT(2048), st, a);
        } else if (a.pmd || a.umat) {            // the multiplier couples the polarisations: lanes i and i + 32 hold X and Y
            if (a.p2 == 9) PLX_LAUNCH((k_rowreg<9, true>), g, bs, ROWG_LDS(512), st, a);
            else if (a.p2 == 10) PLX_LAUNCH((k_rowreg<10, true>), g, bs, ROWG_LDS(1024), st, a);
            else PLX_LAUNCH((k_rowreg<11, true>), g, bs, ROWG_LDS(2048), st, a);
        } else if (P->rowg_split) {
            if (a.p2 == 9) PLX_LAUNCH((k_rowreg<9, false, false, true>), g, bs, ROWG_LDS_SPLIT(512), st, a);
            else if (a.p2 == 10) PLX_LAUNCH((k_rowreg<10, false, false, true>), g, bs, ROWG_LDS_SPLIT(1024), st, a);
            else PLX_LAUNCH((k_rowreg<11, false, false, true>), g, bs, ROWG_LDS_SPLIT(2048), st, a);
        } else {
            if (a.p2 == 9) PLX_LAUNCH((k_rowreg<9, false>), g, bs, ROWG_LDS(512), st, a);
            else if (a.p2 == 10) PLX_LAUNCH((k_rowreg<10, false>), g, bs, ROWG_LDS(1024), st, a);
            else PLX_LAUNCH((k_rowreg<11, false>), g, bs, ROWG_LDS(2048), st, a);
        }
        return;
    }
    if (P->row_split && a.dual && !a.pmd) {
        SsfmArgs b = a;
        b.dual = 0; b.R = 1; b.logR = 0;
        const dim3 gs((unsigned)N1, FC), bs((unsigned)P->rs_threads);
        if (P->tw_compact) {                     // (both polarisations in one launch: one tail instead of two)
            if (P->row4k_split) PLX_LAUNCH((k_row4k<false, true>), dim3(gs.x * gs.y * 2u), dim3(256), P->rs_lds - 4352 * sizeof(double), st, b);
            else PLX_LAUNCH(k_row4k<false>, dim3(gs.x * gs.y * 2u), dim3(256), P->rs_lds, st, b);   // (rows x frame-channels x polarisations: decoded in the kernel)
            return;
        }
        for (int pol = 0; pol < 2; pol++) {
            if (pol) b.ux = a.uy;
            PLX_LAUNCH(k_row, gs, bs, P->rs_lds, st, b);
        }
        return;
    }
    if (P->rowr && !a.dual && !a.force && !a.hmul) {
        PLX_LAUNCH((k_row256r<false, true>), dim3(64u, FC), dim3(ROWR_THREADS), ROWR_LDS_SC, st, a);
        return;
    }
    if (P->rowr && a.dual && !a.force && !a.hmul && !a.umat) {
        if (a.pmd && P->row256_split && a.e1tab) PLX_LAUNCH((k_row256r<true, false, true>), dim3(128u, FC), dim3(ROWR_THREADS), ROWR_LDS - 4 * 272 * sizeof(double), st, a);
        else if (a.pmd) PLX_LAUNCH(k_row256r<true>, dim3(128u, FC), dim3(ROWR_THREADS), ROWR_LDS, st, a);
        else PLX_LAUNCH(k_row256r<false>, dim3(128u, FC), dim3(ROWR_THREADS), ROWR_LDS, st, a);
        return;
    }
    PLX_LAUNCH(k_row, dim3((unsigned)(N1 / a.R), FC), dim3((unsigned)P->row_threads), P->lds_row, st, a);
}

// Read the event intervals of the step loops that have finished (see plx_ssfm::ProfRun) into k_ms / k_launches.
static int resolve_profiles(plx_ssfm *P)
{
    for (auto &r : P->prof_pending) {
        // ACTIVE launches only: the chunked loop also issues launches after every frame has finished (they return at
        // once).  Step s of the slowest frame is its (s+1)-th; the fused column sweep needs one more round to finish
        // the last step and write the field out.
        for (size_t i = 0; i + 1 < r.ev.size(); i++) {
            const int cls = r.cls[i], step = r.step[i];
            const bool active = (r.fused && cls == 0) ? step <= r.maxnc : step < r.maxnc;
            if (!active) continue;
            float ms = 0;
            PLX_HIP(hipEventElapsedTime(&ms, r.ev[i], r.ev[i + 1]));
            P->k_ms[cls] += ms;
            P->k_launches[cls]++;
        }
        for (hipEvent_t e : r.ev) P->evfree.push_back(e);
    }
    P->prof_pending.clear();
    return PLX_OK;
}

// The frames of one call through the whole step loop (fiber.m:518-552).
static int propagate_frames(plx_ssfm *P, cplx *d_ux, cplx *d_uy, int nframes, hipStream_t st)
{
    SsfmArgs a = P->a;
    const int nfc = a.nfc, N1 = 1 << a.p1, N2 = 1 << a.p2;
    a.ux = d_ux;
    a.uy = d_uy;
    a.nframes = nframes;
    a.active = P->d_active;
    a.nactive = P->d_ndone + 2;
    unsigned FC = (unsigned)nframes * nfc;      // frame-channels launched: shrinks with the host's (lagging) view of the active list
    PLX_HIP(hipMemsetAsync(a.ctl, 0, sizeof(FrameCtl) * nframes, st));
    PLX_HIP(hipMemsetAsync(a.umax, 0, sizeof(unsigned long long) * FC, st));
    PLX_HIP(hipMemsetAsync(P->d_ndone, 0, 64, st));
    const bool fused = P->fused != 0;
    if (fused) { // the first fused launch also forms nextstep's initial maximum
        PLX_HIP(hipMemsetAsync(P->d_slots, 0xFF, sizeof(unsigned long long) * 2 * (size_t)nframes * P->tiles_pf, st));   // ~0 = "not arrived"; [parity][frame][tile]
        PLX_HIP(hipMemsetAsync(P->d_mbox, 0, P->mbox_bytes, st));                                                      // no entry posted, nothing claimed
    } else {
        unsigned gx = (unsigned)((P->N + 255) / 256);
        if (gx > 64) gx = 64;
        PLX_LAUNCH(k_umax, dim3(gx, FC), dim3(256), 16 * sizeof(double), st, a);
    }
    const dim3 blk(256);
    const dim3 gctl((unsigned)((nframes + 63) / 64)), bctl(64);
    const dim3 bcol((unsigned)P->col_threads);
    // Data-dependent trip count (fiber.m:518): steps are enqueued in chunks; the completed-frame counter and the
    // abort word of chunk k are read back while chunk k+1 executes.
    // The sweeps of a step cover the frames of the active list (k_compact, once per step); their grids follow the
    // host's last read-back of its length, an upper bound (frames only ever leave), workgroups beyond the list exit.
    // (the list is rebuilt before every step for batches of 64 frames and more, once per chunk for small ones, whose
    // steps are launch-bound: the sweeps skip a listed frame that has finished meanwhile)
    const bool compact_every_step = nframes >= 64;
    int chunk = 4, steps = 0;
    const int kMaxSteps = 1 << 19;      // (far beyond any physical span; also below the period of the mailbox tags of k_colx16)
    bool pending = false, aborted = false;
    // profiling: an event in front of every launch of the loop (and one after the last); intervals are attributed to
    // the kernel class that follows the event.  Classes: 0 k_colx16 / k_col_fwd, 1 k_row, 2 k_col_inv, 3 control.
    // (the events of a call that ends early -- a HIP failure, a frame-barrier time-out -- go back to the plan's free list)
    struct ProfGuard {
        plx_ssfm *P;
        plx_ssfm::ProfRun run;
        ~ProfGuard() { for (hipEvent_t e : run.ev) P->evfree.push_back(e); }
    } guard{P, {}};
    plx_ssfm::ProfRun &run = guard.run;
    auto mark = [&](int cls, int step) -> int {
        if (!P->profile) return PLX_OK;
        hipEvent_t e;
        if (!P->evfree.empty()) { e = P->evfree.back(); P->evfree.pop_back(); }
        else PLX_HIP(hipEventCreate(&e));
        run.ev.push_back(e); run.cls.push_back(cls); run.step.push_back(step);   // (owned by the guard from here on)
        PLX_HIP(hipEventRecord(e, st));
        return PLX_OK;
    };
#define PLX_MARK(cls, step) do { int rc_ = mark((cls), (step)); if (rc_) return rc_; } while (0)
    for (;;) {
        for (int sidx = 0; sidx < chunk; sidx++) {
            const dim3 gcol((unsigned)(N2 / a.W), FC);
            P->slots_launched += FC / nfc;
            if (fused) {
                const int tcx = (int)gcol.x, tct = (int)(gcol.x * FC);
                const dim3 gx((unsigned)(tct < P->fused_grid ? tct : P->fused_grid));
                if (compact_every_step || sidx == 0) {
                    PLX_MARK(3, steps + sidx);
                    PLX_LAUNCH(k_compact, dim3(1), dim3(COMPACT_THREADS), COMPACT_THREADS * sizeof(int), st, (const FrameCtl *)a.ctl, nframes, P->d_active, P->d_ndone + 2,
                               compact_every_step ? 1 : chunk);
                }
#ifdef PLX_EMU
                // the emulator must keep one frame's workgroups alive together (PLX_EMU_STARVE: a test starves the barrier)
                emu::g_concurrency = getenv("PLX_EMU_STARVE") ? 1 : P->tiles_pf;
#endif
                a.round = steps + sidx;
                PLX_MARK(0, steps + sidx);
                if (a.dual) PLX_LAUNCH(k_colx16<true>, gx, blk, P->lds_col, st, a, tcx, P->tiles_pf);
                else PLX_LAUNCH(k_colx16<false>, gx, blk, P->lds_col, st, a, tcx, P->tiles_pf);
#ifdef PLX_EMU
                emu::g_concurrency = 1;
#endif
                if (a.e1tab) {             // PMD: the step's trunk phasors, once per frame (between the controller and the row pass)
                    PLX_MARK(3, steps + sidx);
                    PLX_LAUNCH(k_pmd_tab, dim3(FC / nfc), blk, 0, st, a);
                }
                PLX_MARK(1, steps + sidx);
                launch_row(P, a, FC, st);
                P->row_launches++;
                continue;
            }
            PLX_MARK(3, steps + sidx);
            PLX_LAUNCH(k_ctrl, gctl, bctl, 0, st, a, nframes);
            if (compact_every_step || sidx == 0)
                PLX_LAUNCH(k_compact, dim3(1), dim3(COMPACT_THREADS), COMPACT_THREADS * sizeof(int), st, (const FrameCtl *)a.ctl, nframes, P->d_active, P->d_ndone + 2,
                               compact_every_step ? 1 : chunk);
            if (!a.dual && a.xpm) {
                unsigned gx = (unsigned)((P->N + 255) / 256);
                if (gx > 256) gx = 256;
                PLX_LAUNCH(k_rowsum, dim3(gx, (unsigned)nframes), blk, 0, st, a);
            }
            if (a.e1tab) PLX_LAUNCH(k_pmd_tab, dim3(FC / nfc), blk, 0, st, a);
            PLX_MARK(0, steps + sidx);
            PLX_LAUNCH(k_col_fwd, gcol, bcol, P->lds_col, st, a);
            PLX_MARK(1, steps + sidx);
            launch_row(P, a, FC, st);
            PLX_MARK(2, steps + sidx);
            PLX_LAUNCH(k_col_inv, gcol, bcol, P->lds_col, st, a);
            P->row_launches++;
        }
        PLX_MARK(3, steps + chunk);     // closes the last interval of the chunk (the read-back below lands in class 3)
        if (steps == 0 && !P->prof_pending.empty()) {      // (the GPU has this call's first chunk to work on meanwhile)
            const int rc_ = resolve_profiles(P);
            if (rc_) return rc_;
        }
        steps += chunk;
        if (pending) {
            PLX_HIP(hipEventSynchronize(P->ev));
            if (P->h_ndone[1]) { aborted = true; break; }
            if (P->h_ndone[0] >= nframes) break;
            const unsigned live = (unsigned)(nframes - P->h_ndone[0]) * nfc;   // as of the previous chunk: an upper bound
            if (live < FC) FC = live;
        }
        PLX_HIP(hipMemcpyAsync(P->h_ndone, P->d_ndone, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
        PLX_HIP(hipEventRecord(P->ev, st));
        pending = true;
        if (chunk > 8) chunk = 4;                                     // (after a predicted first chunk)
        else if (chunk < (compact_every_step ? 8 : 16)) chunk *= 2;   // (small batches are launch-bound: longer chunks keep the queue fed)
        if (steps > kMaxSteps) PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_propagate_dev: step loop did not terminate");
    }
    PLX_HIP(hipMemcpyAsync(P->h_ctl.data(), a.ctl, sizeof(FrameCtl) * nframes, hipMemcpyDeviceToHost, st));
    PLX_HIP(hipMemcpyAsync(P->h_ndone, P->d_ndone, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    PLX_HIP(hipStreamSynchronize(st));
    P->slots_listed += P->h_ndone[3];
    if (aborted || P->h_ndone[1]) {
        // The field of this call is lost (it has been propagated in place up to the time-out).  The plan itself stays usable:
        // from now on it takes the barrier-free three-sweep step, which needs no co-residency.  The gateway tier, whose
        // pristine input is still in its pinned staging buffer, repeats the call that way at once (gateway_ssfm).
        P->barrier_timeouts++;
        P->fused = 0;
        PLX_FAIL(PLX_ERR_TIMEOUT, "plx_ssfm_propagate_dev: frame barrier timed out (the workgroups of a frame were not co-resident: "
                                  "another kernel holds the GPU); nothing was stored after the time-out and the field of this call is "
                                  "INVALID -- the plan now takes the barrier-free three-sweep step: restore the field and call again, "
                                  "or create such plans with plx_ssfm_create_ex(..., PLX_SSFM_SHARE_DEVICE)");
    }
    int maxnc = 0;
    for (int f = 0; f < nframes; f++) {
        P->frame_steps += P->h_ctl[f].ncycle + (fused ? 1 : 0);   // (the fused sweep's last round writes the field out)
        if (!P->h_ctl[f].done) PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_propagate_dev: a frame did not reach the fibre end");
        P->sample_steps += (int64_t)P->h_ctl[f].ncycle * (int64_t)P->N * nfc;
        if (P->h_ctl[f].ncycle > maxnc) maxnc = P->h_ctl[f].ncycle;
    }
    if (!run.ev.empty()) {
        run.maxnc = maxnc; run.fused = fused;
        P->prof_pending.push_back(std::move(run));
        run.ev.clear();                 // (moved out: nothing left for the guard to return)
    }
#undef PLX_MARK
    return PLX_OK;
}

extern "C" int plx_ssfm_propagate_dev(plx_ssfm *P, double *d_ux, double *d_uy, int nframes, void *stream)
{
    if (!P || !d_ux) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_propagate_dev: null argument");
    if (nframes < 1 || nframes > P->d.max_frames) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_propagate_dev: nframes out of range");
    if (P->a.dual && !d_uy) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_propagate_dev: dual-polarisation plan needs d_uy");
    if (P->a.brf_per_frame && P->brf_sets < nframes)
        PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_propagate_dev: fewer birefringence sets than frames");
    P->slots_launched = 0; P->slots_listed = 0; P->row_launches = 0; P->sample_steps = 0; P->frame_steps = 0;
    const int rc = propagate_frames(P, (cplx *)d_ux, (cplx *)d_uy, nframes, (hipStream_t)stream);
    if (rc) return rc;
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_ssfm_set_step_sequence(plx_ssfm *P, const double *dz, int nsteps)
{
    if (!P || nsteps < 0 || (nsteps > 0 && !dz)) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_set_step_sequence: bad argument");
    if (nsteps > P->dzlist_cap) {
        if (P->d_dzlist) { (void)hipFree(P->d_dzlist); P->d_dzlist = nullptr; P->dzlist_cap = 0; }
        PLX_HIP(hipMalloc((void **)&P->d_dzlist, sizeof(double) * (size_t)nsteps));
        P->dzlist_cap = nsteps;
    }
    if (nsteps > 0) PLX_HIP(hipMemcpy(P->d_dzlist, dz, sizeof(double) * (size_t)nsteps, hipMemcpyHostToDevice));
    P->a.dzlist = nsteps > 0 ? P->d_dzlist : nullptr;
    P->a.ndz = nsteps;
    return PLX_OK;
}

extern "C" int plx_ssfm_log_steps(plx_ssfm *P, int max_steps)
{
    if (!P || max_steps < 0) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_log_steps: bad argument");
    if (P->d_dzlog) { (void)hipFree(P->d_dzlog); P->d_dzlog = nullptr; }
    P->a.dzlog = nullptr; P->a.logcap = 0;
    if (max_steps > 0) {
        PLX_HIP(hipMalloc((void **)&P->d_dzlog, sizeof(double) * (size_t)max_steps * P->d.max_frames));
        P->a.dzlog = P->d_dzlog; P->a.logcap = max_steps;
    }
    return PLX_OK;
}

extern "C" int plx_ssfm_step_sequence(plx_ssfm *P, int frame, double *dz, int max_steps)
{
    if (!P || !dz || frame < 0 || frame >= P->d.max_frames || max_steps < 0) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_step_sequence: bad argument");
    if (!P->d_dzlog) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_step_sequence: no log (plx_ssfm_log_steps)");
    const int n = max_steps < P->a.logcap ? max_steps : P->a.logcap;
    PLX_HIP(hipMemcpy(dz, P->d_dzlog + (size_t)frame * P->a.logcap, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return PLX_OK;
}

extern "C" int plx_ssfm_utilisation(plx_ssfm *P, int64_t *frame_steps, int64_t *slots_listed, int64_t *slots_launched)
{
    if (!P) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_utilisation: null plan");
    if (frame_steps) *frame_steps = P->frame_steps;
    if (slots_listed) *slots_listed = P->slots_listed;
    if (slots_launched) *slots_launched = P->slots_launched;
    return PLX_OK;
}

extern "C" int plx_ssfm_info(plx_ssfm *P, int32_t *info)
{
    if (!P || !info) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_info: null argument");
    info[0] = P->fused; info[1] = P->p1; info[2] = P->p2; info[3] = P->fused_grid; info[4] = P->tiles_pf;
    info[5] = P->col_threads; info[6] = (P->rowr || P->rowsm) ? ROWR_THREADS : P->rowreg ? ROWG_THREADS : (P->tw_compact ? (P->row_pair4k ? 512 : 256) : (P->row_split ? P->rs_threads : P->row_threads)); info[7] = (P->rowreg || P->rowsm) ? 2 : (P->row_pair4k ? 0 : ((P->tw_compact && !P->a.dual) ? 1 : P->row_split));
    return PLX_OK;
}

extern "C" int plx_ssfm_profile(plx_ssfm *P, int enable)
{
    if (!P) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_profile: null plan");
    P->profile = enable ? 1 : 0;
    return PLX_OK;
}

extern "C" int plx_ssfm_kernel_times(plx_ssfm *P, double *ms, int64_t *launches)
{
    if (!P || !ms || !launches) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_kernel_times: null argument");
    const int rc = resolve_profiles(P);
    if (rc) return rc;
    for (int k = 0; k < 4; k++) { ms[k] = P->k_ms[k]; launches[k] = P->k_launches[k]; P->k_ms[k] = 0; P->k_launches[k] = 0; }
    return PLX_OK;
}

// ---- the plan's FFT engine as a spectral filter (library-internal, plx_internal.h) ----
int plx_ssfm_filter_table(plx_ssfm *P, const double *h_re, const double *h_im, cplx **d_out)
{
    if (!P || !h_re || !d_out) PLX_FAIL(PLX_ERR_ARG, "filter table: null argument");
    const int N1 = 1 << P->p1, N2 = 1 << P->p2;
    std::vector<cplx> h(P->N);
    for (int j = 0; j < N1; j++) {
        const unsigned k1 = plx_bitrev((unsigned)j, P->p1);
        for (int i = 0; i < N2; i++) {
            const size_t k = (size_t)k1 + (size_t)N1 * plx_bitrev((unsigned)i, P->p2);
            h[(size_t)j * N2 + i] = make_double2(h_re[k], h_im ? h_im[k] : 0.0);
        }
    }
    cplx *d = nullptr;
    if (hipMalloc((void **)&d, P->N * sizeof(cplx)) != hipSuccess ||
        hipMemcpy(d, h.data(), P->N * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess) {
        if (d) (void)hipFree(d);
        PLX_FAIL(PLX_ERR_HIP, "filter table: device allocation/upload failed");
    }
    *d_out = d;
    return PLX_OK;
}

void plx_ssfm_geometry(const plx_ssfm *P, int *p1, int *p2) { *p1 = P->p1; *p2 = P->p2; }

int plx_ssfm_filter_dev(plx_ssfm *P, cplx *d_ux, cplx *d_uy, const cplx *d_hmul, int nframes, void *stream, const cplx *d_umat)
{
    if (!P || !d_ux || (!d_hmul && !d_umat)) PLX_FAIL(PLX_ERR_ARG, "filter: null argument");
    if (d_umat && (!P->a.dual || P->a.nfc != 1)) PLX_FAIL(PLX_ERR_ARG, "filter: matrix tables need a dual-polarisation single-field plan");
    if (nframes < 1 || nframes > P->d.max_frames) PLX_FAIL(PLX_ERR_ARG, "filter: nframes outside [1, max_frames]");
    if (P->a.dual && !d_uy) PLX_FAIL(PLX_ERR_ARG, "filter: dual-polarisation plan needs d_uy");
    hipStream_t st = (hipStream_t)stream;
    SsfmArgs b = P->a;
    b.ux = d_ux; b.uy = d_uy; b.nframes = nframes; b.hmul = d_hmul; b.umat = d_umat;
    b.force = 1; b.spm = 0; b.xpm = 0; b.pmd = 0; b.f_cur = 0; b.f_leff = 0; b.f_sc = b.invN;
    const int N1 = 1 << b.p1, N2 = 1 << b.p2;
    const unsigned FC = (unsigned)nframes * b.nfc;
    PLX_HIP(hipMemsetAsync(P->d_ctl, 0, sizeof(FrameCtl) * nframes, st));   // no frame is "done"
    PLX_HIP(hipMemsetAsync(P->d_ndone, 0, 64, st));
    const dim3 gcol((unsigned)(N2 / b.W), FC), grow((unsigned)(N1 / b.R), FC);
    PLX_LAUNCH(k_col_fwd, gcol, dim3((unsigned)P->col_threads), P->lds_col, st, b);
    if (d_umat && !P->tw_compact && !P->rowreg) PLX_LAUNCH(k_row, grow, dim3((unsigned)P->row_threads), P->lds_row, st, b);   // matrix tables couple the polarisations
    else launch_row(P, b, FC, st);
    PLX_LAUNCH(k_col_inv, gcol, dim3((unsigned)P->col_threads), P->lds_col, st, b);
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_ssfm_results(plx_ssfm *P, int nframes, double *firstdz, int32_t *ncycle)
{
    if (!P || nframes < 1 || nframes > P->d.max_frames) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_results: bad argument");
    for (int f = 0; f < nframes; f++) {
        if (firstdz) firstdz[f] = P->h_ctl[f].firstdz;
        if (ncycle) ncycle[f] = P->h_ctl[f].ncycle;
    }
    return PLX_OK;
}

extern "C" int plx_ssfm_stats(plx_ssfm *P, int64_t *row_pass_launches, int64_t *sample_steps)
{
    if (!P) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_stats: null plan");
    if (row_pass_launches) *row_pass_launches = P->row_launches;
    if (sample_steps) *sample_steps = P->sample_steps;
    return PLX_OK;
}

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
        rc = set_brf(P, db0, theta, epsilon, 1, nullptr, false);
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
    rc = plx_ssfm_propagate_dev(P, dx, dy, 1, nullptr);
    if (rc == PLX_ERR_TIMEOUT) {
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
        PLX_LAUNCH(k_col_fwd, gcol, dim3(256), P->lds_col, st, b);
        PLX_LAUNCH(k_row, grow, dim3((unsigned)P->row_threads), P->lds_row, st, b);
        PLX_LAUNCH(k_col_inv, gcol, dim3(256), P->lds_col, st, b);
    }
    void nl_att(cplx *x, double dz)
    { // nl_step(alphalin,gam,dz,x,...) then x = x*exp(-halfalpha*dz)
        const double leff = (alphalin == 0) ? dz : (1 - exp(-alphalin * dz)) / alphalin;
        const double att = exp(-(0.5 * alphalin) * dz);
        unsigned g = (unsigned)((P->N + 255) / 256);
        if (g > 2048) g = 2048;
        PLX_LAUNCH(k_nl_att, dim3(g), dim3(256), 0, st, x, (const double *)a.gam, P->N, a.nfc, fls2, fls3, leff, att);
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
        PLX_LAUNCH(k_maxdiff, dim3(g), dim3(256), 16 * sizeof(double), st, (const cplx *)u, (const cplx *)uh, n, d_max);
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
            PLX_LAUNCH(k_richardson, dim3(g), dim3(256), 0, st, u, (const cplx *)uh, n);
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
        PLX_LAUNCH(k_umax, dim3(gx, (unsigned)d.nfc), dim3(256), 16 * sizeof(double), nullptr, b);
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


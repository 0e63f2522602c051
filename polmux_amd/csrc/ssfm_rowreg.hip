// ssfm_rowreg.hip -- the register-form row pass for rows of 512, 1024 and 2048 points (frames of 2^17 ... 2^19 samples).
#include "ssfm_pmd.h"
#include "ssfm_kernels.h"
using namespace plxs;

namespace {

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
} // namespace

namespace plxs {
namespace {
template <int LOGM> sweep_kernel_t rowreg_pick(bool pair, bool scalar, bool split)
{
    if (pair) return scalar ? nullptr : (split ? (sweep_kernel_t)k_rowreg<LOGM, true, false, true> : (sweep_kernel_t)k_rowreg<LOGM, true>);
    if (scalar) return split ? (sweep_kernel_t)k_rowreg<LOGM, false, true, true> : (sweep_kernel_t)k_rowreg<LOGM, false, true>;
    return split ? (sweep_kernel_t)k_rowreg<LOGM, false, false, true> : (sweep_kernel_t)k_rowreg<LOGM, false>;
}
} // namespace
sweep_kernel_t rowreg_kernel(int logm, bool pair, bool scalar, bool split)
{
    return logm == 9 ? rowreg_pick<9>(pair, scalar, split) : logm == 10 ? rowreg_pick<10>(pair, scalar, split) : logm == 11 ? rowreg_pick<11>(pair, scalar, split) : nullptr;
}
} // namespace plxs

// ssfm_rowsm.hip -- the register-form row pass for rows of 32, 64 and 128 points (frames of 2^13 ... 2^15 samples).
#include "ssfm_kernels.h"
using namespace plxs;

namespace {

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
} // namespace

namespace plxs {
sweep_kernel_t rowsm_kernel(int logm, bool scalar)
{
    if (scalar) return logm == 5 ? (sweep_kernel_t)k_rowsm<5, true> : logm == 6 ? (sweep_kernel_t)k_rowsm<6, true> : logm == 7 ? (sweep_kernel_t)k_rowsm<7, true> : nullptr;
    return logm == 5 ? (sweep_kernel_t)k_rowsm<5, false> : logm == 6 ? (sweep_kernel_t)k_rowsm<6, false> : logm == 7 ? (sweep_kernel_t)k_rowsm<7, false> : nullptr;
}
} // namespace plxs

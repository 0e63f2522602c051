// ssfm_row256.hip -- the register-form row pass for 256-point rows (the 2^16-sample frame of BASELINE config[1]).
#include "ssfm_pmd.h"
#include "ssfm_kernels.h"
using namespace plxs;

namespace {

// ------------------------------------------------- pass 2 for 256-point rows, register form ---
// The headline shape (dual polarisation, no PMD, 256-point rows, the step's exp(-i beta dz)) with every radix level in
// registers, as k_row4k does for 4096-point rows: ONE WAVE = one tile of 2 rows x 2 polarisations, 16 lanes per row, lane j
// of a row holds its points j + 16 k.  Forward: lvl2_dif256 -> one exchange through the padded LDS row -> r16_dif; the
// multiplier on the 16 bins the lane then holds (bit-reversed order, where the tables are); the inverse mirrors it.  Two LDS
// exchanges per transform pair end (4 x 16 ds_write_b128 + 4 x 16 ds_read_b128 per lane and tile) where k_row's eleven passes make 64 + 96 per
// lane on half as many points, no workgroup barrier at all (a one-wave workgroup only waits for its own LDS operations), and
// the butterflies are k_row's, stage by stage.  Inter-pass twiddles as in k_row4k: tpass[j + 16 k] = tpass[j] * tpass[16 k],
// a lane reads one entry and the row's sixteen lanes share sixteen (bk).
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

} // namespace

namespace plxs {
// dual-polarisation plans: <PMD> with whole-sample exchanges, or <PMD = true, SPLIT> for plans with trunk phasor tables;
// scalar plans: <false, SC>
sweep_kernel_t row256_kernel(bool pmd, bool scalar, bool split)
{
    if (scalar) return (pmd || split) ? nullptr : (sweep_kernel_t)k_row256r<false, true>;
    if (split) return pmd ? (sweep_kernel_t)k_row256r<true, false, true> : nullptr;
    return pmd ? (sweep_kernel_t)k_row256r<true> : (sweep_kernel_t)k_row256r<false>;
}
} // namespace plxs

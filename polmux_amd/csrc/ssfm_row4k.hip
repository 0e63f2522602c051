// ssfm_row4k.hip -- the register-form row pass for 4096-point rows (2^20-sample frames, BASELINE config[4]).
#include "ssfm_pmd.h"
#include "ssfm_kernels.h"
using namespace plxs;

namespace {

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

} // namespace

namespace plxs {
sweep_kernel_t row4k_kernel(bool pair, bool split)
{
    if (pair) return split ? nullptr : (sweep_kernel_t)k_row4k<true>;
    return split ? (sweep_kernel_t)k_row4k<false, true> : (sweep_kernel_t)k_row4k<false>;
}
} // namespace plxs

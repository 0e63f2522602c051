// ssfm_row.hip -- the LDS-resident row pass k_row (any split, filter passes, matrix tables).  fiber.m:762-773, :877-935.
#include "ssfm_pmd.h"
#include "ssfm_kernels.h"
using namespace plxs;

namespace {

// --------------------------------------------------------------- pass 2: rows ---
// Second half of the forward transform, the linear operator of the step
// (lin_step :771-773 / matrix_step :907-933) and the first half of the inverse
// transform, all on one LDS-resident row set (padded layout, see plx_fft.h).
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

} // namespace

namespace plxs {
sweep_kernel_t row_kernel() { return k_row; }
} // namespace plxs

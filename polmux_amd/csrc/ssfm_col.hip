// ssfm_col.hip -- the plain column sweeps of the three-sweep step: k_col_fwd (Kerr step fused on load + N1-point DIF) and
// k_col_inv (N1-point DIT + attenuation + nextstep's maximum).  fiber.m:776-874, :531-532, :694-696.
#include "ssfm_kernels.h"
using namespace plxs;

namespace {

#define COL_CH 4
// ------------------------------------------------------------ pass 1: columns ---
// Kerr step (matrix_nl_step :832-852 / nl_step :792-804) fused into the load of
// the forward column transform.  Tile = N1 rows x T complex (dual: W columns of ux |
// W of uy; scalar: W columns): T*16 B contiguous bytes of LDS per row, so the T
// interleaved transforms are read conflict-free.  Global loads are issued in
// batches of COL_CH per thread before any arithmetic, to keep >= 64 KiB in flight
// per CU.
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

} // namespace

namespace plxs {
sweep_kernel_t col_fwd_kernel() { return k_col_fwd; }
sweep_kernel_t col_inv_kernel() { return k_col_inv; }
} // namespace plxs

// ssfm_small.hip -- the small kernels of the split-step Fourier propagator: nextstep's initial maximum, the one-lane-per-frame
// step controller, the active list, the trunk phasor tables of PMD plans, scalar XPM's row sums and the element-wise pieces
// of the adaptive-step scheme (fiber.m:682-758, :795, :925, :938-1009).
#include "ssfm_ctrl.h"
#include "ssfm_kernels.h"
using namespace plxs;

namespace {

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

__global__ __launch_bounds__(64) void k_ctrl(SsfmArgs a, int nframes)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    ctrl_step<false>(a, f);
}

// The active list: frames that have not reached the fibre end, in frame order (one workgroup; a block scan over
// contiguous chunks of frames).  Runs once per step, in front of the step's sweeps.
// `serves`: the number of steps this list will be used for (utilisation accounting).
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

} // namespace

namespace plxs {
void launch_umax(dim3 grid, hipStream_t st, const SsfmArgs &a) { PLX_LAUNCH(k_umax, grid, dim3(256), 16 * sizeof(double), st, a); }
void launch_ctrl(int nframes, hipStream_t st, const SsfmArgs &a) { PLX_LAUNCH(k_ctrl, dim3((unsigned)((nframes + 63) / 64)), dim3(64), 0, st, a, nframes); }
void launch_compact(const FrameCtl *ctl, int nframes, int *active, int *nactive, int serves, hipStream_t st)
{
    PLX_LAUNCH(k_compact, dim3(1), dim3(COMPACT_THREADS), COMPACT_THREADS * sizeof(int), st, ctl, nframes, active, nactive, serves);
}
void launch_rowsum(dim3 grid, hipStream_t st, const SsfmArgs &a) { PLX_LAUNCH(k_rowsum, grid, dim3(256), 0, st, a); }
void launch_pmd_tab(unsigned frames, hipStream_t st, const SsfmArgs &a) { PLX_LAUNCH(k_pmd_tab, dim3(frames), dim3(256), 0, st, a); }
void launch_nl_att(unsigned grid, hipStream_t st, cplx *u, const double *gam, size_t N, int nfc, int spm, int xpm, double leff, double att)
{
    PLX_LAUNCH(k_nl_att, dim3(grid), dim3(256), 0, st, u, gam, N, nfc, spm, xpm, leff, att);
}
void launch_maxdiff(unsigned grid, hipStream_t st, const cplx *u, const cplx *uh, size_t n, unsigned long long *out)
{
    PLX_LAUNCH(k_maxdiff, dim3(grid), dim3(256), 16 * sizeof(double), st, u, uh, n, out);
}
void launch_richardson(unsigned grid, hipStream_t st, cplx *u, const cplx *uh, size_t n) { PLX_LAUNCH(k_richardson, dim3(grid), dim3(256), 0, st, u, uh, n); }
} // namespace plxs

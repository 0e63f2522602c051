// dev experiment (round 4): where do the ~440 clocks per symbol of the CMA recurrence go?
// A standalone replica of k_cma16's per-symbol body (polmux_amd/csrc/plx_rx.hip: 16 lanes per frame = output row r x tap t,
// taps in registers, two 8-lane DPP sums, k = mu (R - |y|^2), four fused multiply-add pairs for the tap update) with
// compile-time ABLATIONS that take one class of work out at a time, run by ONE wave (4 frames) and by one wave per SIMD on
// every CU.  Reports ns and shader clocks per symbol.
//   FULL      the recurrence as shipped
//   NORED     no 8-lane reduction (a lane's own partial stands in for y): products + k + update, same loop-carried chain
//   NOCHAIN   the update is computed but NOT carried to the next symbol (taps stay): the in-order instruction stream
//             without its dependency -- what the issue rate alone costs
//   REDONLY   only the two reductions, chained through one multiply-add (the DPP moves' own latency)
//   LDSRED    the reduction through LDS swizzles (ds_swizzle_b32) instead of DPP moves
//   LANES8    8 lanes per frame, 2 taps per lane: two reduction levels instead of three, twice the products and updates
// build: hipcc --offload-arch=gfx950 -O3 cma_chain.hip -o cma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double2 cplx;
template <int X> __device__ __forceinline__ double lane_xchg(double v)
{
    constexpr int ctrl = X == 1 ? 0xB1 : X == 2 ? 0x4E : X == 7 ? 0x141 : X == 8 ? 0x128 : 0x140;
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void sum8x2(double &a, double &b)
{
    double a1 = lane_xchg<1>(a), b1 = lane_xchg<1>(b);
    a += a1; b += b1;
    a1 = lane_xchg<2>(a); b1 = lane_xchg<2>(b);
    a += a1; b += b1;
    a1 = lane_xchg<7>(a); b1 = lane_xchg<7>(b);
    a += a1; b += b1;
}
__device__ __forceinline__ void sum4x2(double &a, double &b)     // (8 lanes per frame: rows of 4 tap lanes)
{
    double a1 = lane_xchg<1>(a), b1 = lane_xchg<1>(b);
    a += a1; b += b1;
    a1 = lane_xchg<2>(a); b1 = lane_xchg<2>(b);
    a += a1; b += b1;
}
// ds_swizzle butterflies (no LDS memory is touched: the crossbar only): xor 1, 2, 4 within groups of 8 lanes
template <int M> __device__ __forceinline__ double swz(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, (M << 10) | 0x1F);
    hi = __builtin_amdgcn_ds_swizzle(hi, (M << 10) | 0x1F);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void sum8x2_lds(double &a, double &b)
{
    double a1 = swz<1>(a), b1 = swz<1>(b);
    a += a1; b += b1;
    a1 = swz<2>(a); b1 = swz<2>(b);
    a += a1; b += b1;
    a1 = swz<4>(a); b1 = swz<4>(b);
    a += a1; b += b1;
}

enum { FULL = 0, NORED = 1, NOCHAIN = 2, REDONLY = 3, LDSRED = 4, LANES8 = 5 };

template <int V> __global__ __launch_bounds__(64) void k_chain(const cplx *x, cplx *y, double *hout, long long *clk, int L, int passes, double mu)
{
    const int lane = threadIdx.x, l16 = lane & 15, r = l16 >> 3, t = l16 & 7;
    const int f = blockIdx.x * 4 + (lane >> 4);
    const cplx *x1 = x + (size_t)f * 2 * L, *x2 = x1 + L;
    cplx ha = make_double2(t == 3 ? (r == 0 ? 1.0 : 0.0) : 0.0, 0.0), hb = make_double2(t == 3 ? (r == 1 ? 1.0 : 0.0) : 0.0, 0.0);
    cplx hc = make_double2(0, 0), hd = hc;      // (LANES8: the lane's second tap)
    double acc = 0;
    __syncthreads();
    const long long w0 = wall_clock64(), t0 = clock64();
    for (int p = 0; p < passes; p++) {
        for (int i0 = 0; i0 < L; i0 += 8) {
            cplx ca[8], cb[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int idx = i0 + u + t - 3;
                if (idx < 0) idx += L; else if (idx >= L) idx -= L;
                ca[u] = x1[idx]; cb[u] = x2[idx];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const cplx xa = ca[u], xb = cb[u];
                double yr, yi;
                if (V == REDONLY) { yr = ha.x + xa.x; yi = ha.y + xa.y; }
                else {
                    yr = (xa.x * ha.x - xa.y * ha.y) + (xb.x * hb.x - xb.y * hb.y);
                    yi = (xa.x * ha.y + xa.y * ha.x) + (xb.x * hb.y + xb.y * hb.x);
                    if (V == LANES8) {      // second tap of the lane (its samples: the neighbouring lane's, here the same registers shifted)
                        yr += (xb.x * hc.x - xb.y * hc.y) + (xa.x * hd.x - xa.y * hd.y);
                        yi += (xb.x * hc.y + xb.y * hc.x) + (xa.x * hd.y + xa.y * hd.x);
                    }
                }
                if (V == LDSRED) sum8x2_lds(yr, yi);
                else if (V == LANES8) sum4x2(yr, yi);
                else if (V != NORED) sum8x2(yr, yi);
                if (V == REDONLY) { ha.x = fma(yr, 1e-9, ha.x); ha.y = fma(yi, 1e-9, ha.y); continue; }
                const double k = mu * (1.0 - yr * yr - yi * yi);
                const double kr = k * yr, ki = k * yi;
                if (V == NOCHAIN) {          // the same arithmetic, its results not fed back (summed up so that they stay alive)
                    acc += fma(ki, xa.y, fma(kr, xa.x, ha.x)) + fma(-kr, xa.y, fma(ki, xa.x, ha.y)) + fma(ki, xb.y, fma(kr, xb.x, hb.x)) +
                           fma(-kr, xb.y, fma(ki, xb.x, hb.y));
                } else {
                    ha.x = fma(ki, xa.y, fma(kr, xa.x, ha.x)); ha.y = fma(-kr, xa.y, fma(ki, xa.x, ha.y));
                    hb.x = fma(ki, xb.y, fma(kr, xb.x, hb.x)); hb.y = fma(-kr, xb.y, fma(ki, xb.x, hb.y));
                    if (V == LANES8) {
                        hc.x = fma(ki, xb.y, fma(kr, xb.x, hc.x)); hc.y = fma(-kr, xb.y, fma(ki, xb.x, hc.y));
                        hd.x = fma(ki, xa.y, fma(kr, xa.x, hd.x)); hd.y = fma(-kr, xa.y, fma(ki, xa.x, hd.y));
                    }
                }
                if (t == 0 && u == 7) y[(size_t)f * L + i0] = make_double2(yr, yi);
            }
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    hout[blockIdx.x * 64 + lane] = ha.x + ha.y + hb.x + hb.y + hc.x + hd.y + acc;
    if (lane == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int V> static void run(const char *name, int nblocks, const cplx *dx, cplx *dy, double *dh, long long *dclk, int L, int passes)
{
    hipLaunchKernelGGL(k_chain<V>, dim3(nblocks), dim3(64), 0, 0, dx, dy, dh, dclk, L, passes, 1.0 / 6000);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_chain<V>, dim3(nblocks), dim3(64), 0, 0, dx, dy, dh, dclk, L, passes, 1.0 / 6000);
    hipDeviceSynchronize();
    std::vector<long long> c(2 * nblocks);
    hipMemcpy(c.data(), dclk, sizeof(long long) * 2 * nblocks, hipMemcpyDeviceToHost);
    double clk = 0, wall = 0;
    for (int b = 0; b < nblocks; b++) { clk += c[2 * b]; wall += c[2 * b + 1]; }
    const double sym = (double)L * passes;
    printf("%-8s %5d wave(s): %7.1f ns per symbol, %6.1f shader clocks per symbol\n", name, nblocks, wall / nblocks * 10.0 / sym, clk / nblocks / sym);
}

int main()
{
    const int L = 1024, passes = 40, maxb = 1024;
    std::vector<cplx> hx((size_t)maxb * 4 * 2 * L);
    srand(1);
    for (auto &v : hx) { const int q = rand() & 3; v = make_double2((q & 1 ? 1 : -1) * 0.7071 + 0.3 * (rand() / (double)RAND_MAX - 0.5), (q & 2 ? 1 : -1) * 0.7071 + 0.3 * (rand() / (double)RAND_MAX - 0.5)); }
    cplx *dx, *dy; double *dh; long long *dclk;
    hipMalloc(&dx, hx.size() * sizeof(cplx)); hipMalloc(&dy, (size_t)maxb * 4 * L * sizeof(cplx)); hipMalloc(&dh, maxb * 64 * sizeof(double)); hipMalloc(&dclk, 2 * maxb * sizeof(long long));
    hipMemcpy(dx, hx.data(), hx.size() * sizeof(cplx), hipMemcpyHostToDevice);
    for (int nb : {1, 1024}) {
        run<FULL>("FULL", nb, dx, dy, dh, dclk, L, passes);
        run<NORED>("NORED", nb, dx, dy, dh, dclk, L, passes);
        run<NOCHAIN>("NOCHAIN", nb, dx, dy, dh, dclk, L, passes);
        run<REDONLY>("REDONLY", nb, dx, dy, dh, dclk, L, passes);
        run<LDSRED>("LDSRED", nb, dx, dy, dh, dclk, L, passes);
        run<LANES8>("LANES8", nb, dx, dy, dh, dclk, L, passes);
    }
    return 0;
}

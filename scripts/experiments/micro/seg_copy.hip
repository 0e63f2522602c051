// dev microbenchmark: in-place tile copy through LDS with the column-sweep access pattern.
// A tile = ROWS rows x SEG bytes, rows PITCH bytes apart.  Compares 128 B segments (separate X / Y arrays,
// W = 8 complex128) with 256 B / 512 B segments (what a polarisation-interleaved layout would give).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double2 cplx;
template <int NARR>
__global__ __launch_bounds__(512) void k_tile(cplx *a0, cplx *a1, int rows, int segc /*cplx per segment*/, size_t pitchc, size_t framec)
{
    extern __shared__ cplx lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const size_t base = (size_t)blockIdx.y * framec + (size_t)blockIdx.x * segc;
    const int nel = rows * segc;
    for (int e0 = tid; e0 < nel; e0 += nthr * 4) {
        cplx v[4][NARR];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int e = e0 + k * nthr; if (e >= nel) e = nel - 1;
            const size_t g = base + (size_t)(e / segc) * pitchc + (e % segc);
            v[k][0] = a0[g];
            if (NARR == 2) v[k][1] = a1[g];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int e = e0 + k * nthr;
            if (e < nel) { lds[e * NARR] = v[k][0]; if (NARR == 2) lds[e * NARR + 1] = v[k][1]; }
        }
    }
    __syncthreads();
    for (int e = tid; e < nel; e += nthr) {
        const size_t g = base + (size_t)(e / segc) * pitchc + (e % segc);
        cplx x = lds[e * NARR]; x.x += 1.0;
        a0[g] = x;
        if (NARR == 2) { cplx y = lds[e * NARR + 1]; y.y += 1.0; a1[g] = y; }
    }
}
__global__ __launch_bounds__(256) void k_stream(cplx *a, size_t n)
{   // plain grid-stride in-place read-modify-write, no LDS: the streaming ceiling of the part for this traffic mix
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { cplx v = a[i]; v.x += 1.0; a[i] = v; }
}
__global__ __launch_bounds__(256) void k_stream4(cplx *a, size_t n)
{   // the same with 4 independent loads in flight per thread
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        cplx v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        v0.x += 1.0; v1.x += 1.0; v2.x += 1.0; v3.x += 1.0;
        a[i] = v0; a[i + stride] = v1; a[i + 2 * stride] = v2; a[i + 3 * stride] = v3;
    }
}
int main()
{
    const int F = 256, N1 = 256, N2 = 256;          // frames of N1 x N2 dual-pol samples
    const size_t NS = (size_t)N1 * N2;
    cplx *x, *y;
    hipMalloc(&x, F * NS * sizeof(cplx) * 2);        // also used as one interleaved array of 2*NS per frame
    y = x + F * NS;
    hipMemset(x, 0, F * NS * sizeof(cplx) * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { const char *name; int narr, segc; size_t pitchc, framec; int tiles; int nthr = 512; } cfgs[] = {
        {"separate X,Y arrays, 64 B segments (W=4, tile 32 KiB: 4 WG/CU), 128 thr", 2, 4, (size_t)N2, NS, N2 / 4, 128},
        {"separate X,Y arrays, 64 B segments (W=4, tile 32 KiB: 4 WG/CU), 256 thr", 2, 4, (size_t)N2, NS, N2 / 4, 256},
        {"separate X,Y arrays, 128 B segments (W=8), 256 thr", 2, 8, (size_t)N2, NS, N2 / 8, 256},
        {"separate X,Y arrays, 128 B segments (W=8)", 2, 8, (size_t)N2, NS, N2 / 8},
        {"separate X,Y arrays, 256 B segments (W=16, tile 128 KiB: 1 WG/CU)", 2, 16, (size_t)N2, NS, N2 / 16},
        {"interleaved pols, 256 B segments (8 samples x 2 pols)", 1, 16, (size_t)2 * N2, 2 * NS, 2 * N2 / 16},
        {"interleaved pols, 512 B segments (tile 128 KiB: 1 WG/CU)", 1, 32, (size_t)2 * N2, 2 * NS, 2 * N2 / 32},
    };
    for (auto &c : cfgs) {
        const size_t ldsb = (size_t)N1 * c.segc * c.narr * sizeof(cplx);
        if (c.narr == 2) hipFuncSetAttribute((const void *)k_tile<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        else hipFuncSetAttribute((const void *)k_tile<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        float best = 1e9;
        for (int it = 0; it < 6; it++) {
            hipEventRecord(e0, 0);
            if (c.narr == 2) hipLaunchKernelGGL(k_tile<2>, dim3(c.tiles, F), dim3(c.nthr), ldsb, 0, x, y, N1, c.segc, c.pitchc, c.framec);
            else hipLaunchKernelGGL(k_tile<1>, dim3(c.tiles, F), dim3(c.nthr), ldsb, 0, x, x, N1, c.segc, c.pitchc, c.framec);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
        }
        const double bytes = 2.0 * F * NS * 2 * sizeof(cplx);
        printf("%-70s %7.1f us  %.2f TB/s (LDS %zu KiB)\n", c.name, best * 1e3, bytes / (best * 1e-3) / 1e12, ldsb >> 10);
    }
    for (int v = 0; v < 2; v++) for (int g : {2048, 8192, 32768}) {
        float best = 1e9; const size_t n = (size_t)F * NS * 2;
        for (int it = 0; it < 6; it++) {
            hipEventRecord(e0, 0);
            if (v == 0) hipLaunchKernelGGL(k_stream, dim3(g), dim3(256), 0, 0, x, n); else hipLaunchKernelGGL(k_stream4, dim3(g), dim3(256), 0, 0, x, n);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
        }
        printf("streaming in-place RMW (%s, grid %d): %7.1f us  %.2f TB/s\n", v ? "4 loads in flight" : "1 load", g, best * 1e3, 2.0 * n * sizeof(cplx) / (best * 1e-3) / 1e12);
    }
    return 0;
}

// dev microbenchmark: the FP64 VALU rate the part sustains (v_fma_f64, independent chains), the denominator of roofline.fp64.
// Spec arithmetic: 256 CUs x 4 SIMDs x 16 lanes x 2 flop per FMA x 2.4 GHz = 78.6 TFLOP/s (AMD's vector FP64 figure for MI355X).
//   hipcc --offload-arch=gfx950 -O3 fp64_peak.hip -o fp64_peak && ./fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CH> __global__ __launch_bounds__(256) void k_fma(double *out, double a, double b, int iters)
{
    double x[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) x[k] = threadIdx.x * 1e-3 + k;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < CH; k++) x[k] = fma(x[k], a, b);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < CH; k++) s += x[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH> static void run(double *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_fma<CH><<<blocks, 256>>>(d, 0.999999, 1e-9, 16);
    hipEventRecord(e0);
    k_fma<CH><<<blocks, 256>>>(d, 0.999999, 1e-9, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * CH * (double)iters * 256.0 * blocks;
    printf("chains %2d blocks %5d (%d waves per SIMD): %.2f ms  %.1f TFLOP/s FP64\n", CH, blocks, blocks / 256, ms, flops / ms * 1e-9);
}
int main(int argc, char **argv)
{
    double *d;
    hipMalloc(&d, sizeof(double) * 256 * 8192);
    if (argc > 1) {       // sustained load for a clock / power reading: <seconds-ish> repetitions of the densest shape
        for (int r = 0; r < atoi(argv[1]); r++) run<8>(d, 4096, 1 << 19);
        hipFree(d);
        return 0;
    }
    for (int blocks : {256, 512, 1024, 2048, 4096}) { run<1>(d, blocks, 1 << 16); run<4>(d, blocks, 1 << 15); run<8>(d, blocks, 1 << 14); }
    hipFree(d);
    return 0;
}

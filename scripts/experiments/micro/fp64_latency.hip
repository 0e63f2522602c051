// dev experiment: what does a DEPENDENT FP64 VALU operation cost a gfx950 wave, and how many independent chains hide it?
// K independent chains of v_fma_f64 / v_add_f64 / v_mul_f64, interleaved round-robin in the instruction stream (inline asm, so
// the compiler cannot reorder), W waves per SIMD.  Prints shader clocks per instruction per wave.
// build: hipcc --offload-arch=gfx950 -O2 fp64_latency.hip -o fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define OPFMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c))
#define OPADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c))
#define OPMUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(m))

template <int K, int OP> __global__ void chains(double *out, long long *clk, int iters)
{
    double x[8];
    for (int k = 0; k < 8; k++) x[k] = 1.0 + 1e-9 * (threadIdx.x + k);
    double m = 1.0000000001, c = 1e-12;
    __syncthreads();
    long long w0 = wall_clock64(); long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / K; r++) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (OP == 0) OPFMA(x[k]);
                if (OP == 1) OPADD(x[k]);
                if (OP == 2) OPMUL(x[k]);
            }
        }
    }
    long long t1 = clock64(); long long w1 = wall_clock64();
    double s = 0;
    for (int k = 0; k < 8; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int K, int OP> static void run(int waves_per_simd, const char *name)
{
    const int nb = 256 * 2, iters = 2000;
    double *out; long long *clk, h[2 * nb];
    hipMalloc(&out, sizeof(double) * nb * 1024); hipMalloc(&clk, sizeof(h));
    int threads = 256 * waves_per_simd;                                                    // 4 SIMDs x W waves of 64
    chains<K, OP><<<nb, threads>>>(out, clk, iters);
    hipDeviceSynchronize();
    chains<K, OP><<<nb, threads>>>(out, clk, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0, wavg = 0;
    for (int b = 0; b < nb; b++) { avg += (double)h[2 * b]; wavg += (double)h[2 * b + 1]; }
    avg /= nb; wavg /= nb;
    double per = avg / ((double)iters * (64 / K) * K);
    double ns = wavg * 10.0 / ((double)iters * (64 / K) * K);                              // wall clock: 100 MHz
    printf("%s K=%d waves/SIMD=%d: %.2f clock64 ticks, %.3f ns per instruction per wave (%.3f ns per SIMD instruction slot)\n", name, K, waves_per_simd, per, ns,
           ns / waves_per_simd);
    hipFree(out); hipFree(clk);
}

int main()
{
    // clock64() = s_memtime: counts at a fixed rate (100 MHz on gfx94x/95x?) -- calibrate against wall_clock64 and the known issue cost
    int dev_clk = 0; hipDeviceGetAttribute(&dev_clk, hipDeviceAttributeClockRate, 0);
    int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
    printf("device clock %d kHz, wall clock %d kHz\n", dev_clk, wc);
    for (int w = 1; w <= 2; w++) {
        run<1, 0>(w, "fma"); run<2, 0>(w, "fma"); run<4, 0>(w, "fma"); run<8, 0>(w, "fma");
        run<1, 1>(w, "add"); run<2, 1>(w, "add"); run<4, 1>(w, "add");
        run<1, 2>(w, "mul"); run<2, 2>(w, "mul"); run<4, 2>(w, "mul");
    }
    return 0;
}

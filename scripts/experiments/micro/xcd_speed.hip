// dev microbenchmark: is one XCD of this box slower?  512 workgroups (two per CU), each streams tiles of the column-sweep
// shape (256 rows x 256 B, rows 4 KiB apart) through registers -- (a) read-modify-write in place, (b) arithmetic only -- and
// records its own wall-clock duration and HW_REG_XCC_ID.  Prints mean / min / max duration per XCD.
//   hipcc --offload-arch=gfx950 -O3 xcd_speed.hip -o xcd_speed && ./xcd_speed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double2 cplx;
__global__ __launch_bounds__(256, 2) void k_probe(cplx *buf, long long *dur, int *xcc, int tiles, int mode, size_t frame_stride)
{
    const int tid = threadIdx.x, j = tid >> 4, t = tid & 15;
    const long long t0 = wall_clock64();
    cplx acc = make_double2(1.0, 0.0);
    for (int it = 0; it < tiles; it++) {
        cplx *base = buf + (size_t)it * frame_stride + (size_t)blockIdx.x * 16 + t;     // tile = columns 16 b .. 16 b + 15 of a 256 x 8192 frame
        cplx x[16];
        if (mode != 1) {
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = base[(size_t)(j + 16 * k) * 8192];
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = make_double2(acc.x + k, acc.y - k);
        }
        if (mode != 2) {
            for (int r = 0; r < 24; r++) {
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = make_double2(fma(x[k].x, 0.999, -x[k].y * 0.001), fma(x[k].y, 0.999, x[k].x * 0.001));
            }
        }
        if (mode != 1) {
#pragma unroll
            for (int k = 0; k < 16; k++) base[(size_t)(j + 16 * k) * 8192] = x[k];
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) { acc.x += x[k].x; acc.y += x[k].y; }
        }
    }
    if (mode == 1 && acc.x == 12345.678) buf[0] = acc;
    if (tid == 0) { dur[blockIdx.x] = wall_clock64() - t0; xcc[blockIdx.x] = (int)__builtin_amdgcn_s_getreg(6164) & 15; }
}
int main(int argc, char **argv)
{
    const int tiles = 64;
    const int sustain = argc > 1 ? atoi(argv[1]) : 0;      // repetitions of the read + write mode only (clock / power reading)
    const size_t frame = (size_t)256 * 8192;              // complex per frame: 32 MiB
    cplx *buf; long long *dur; int *xcc;
    hipMalloc(&buf, frame * tiles * sizeof(cplx)); hipMemset(buf, 0, frame * tiles * sizeof(cplx));
    hipMalloc(&dur, 512 * sizeof(long long)); hipMalloc(&xcc, 512 * sizeof(int));
    const char *names[3] = {"read + arithmetic + write", "arithmetic only", "read + write only"};
    for (int mode = sustain ? 2 : 0; mode < 3; mode++) {
        for (int rep = 0; rep < (sustain ? sustain : 3); rep++) {
            k_probe<<<512, 256>>>(buf, dur, xcc, tiles, mode, frame);
            hipDeviceSynchronize();
        }
        std::vector<long long> d(512); std::vector<int> x(512);
        hipMemcpy(d.data(), dur, 512 * sizeof(long long), hipMemcpyDeviceToHost);
        hipMemcpy(x.data(), xcc, 512 * sizeof(int), hipMemcpyDeviceToHost);
        printf("%s (64 tiles per workgroup), us per tile by XCD:", names[mode]);
        for (int c = 0; c < 8; c++) {
            double s = 0, mn = 1e30, mx = 0; int n = 0;
            for (int i = 0; i < 512; i++) if (x[i] == c) { const double v = d[i] * 0.01 / tiles; s += v; n++; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
            printf("  [%d] %.2f (%.2f-%.2f, n %d)", c, n ? s / n : 0.0, mn, mx, n);
        }
        printf("\n");
    }
    return 0;
}

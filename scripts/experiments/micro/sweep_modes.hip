// dev microbenchmark (round 3): what the part streams for the SSFM step's two sweeps, by
//   * in place  vs  out of place (ping-pong between two buffers),
//   * row pitch under the column tiles (4 KiB at C1, 64 KiB at 2^20-sample frames, padded pitches),
//   * footprint (does a batch that fits the 256 MiB Infinity Cache sweep faster than one that does not).
// The access shapes are those of k_colx16 (256 rows x (8 ux + 8 uy) columns per workgroup, 16 points per lane, one wave
// instruction = 4 rows x 2 x 128 B) and of the row pass (whole contiguous rows, 4 KiB per wave-quad instruction).
// No arithmetic beyond one add per point: this is the memory side alone.
// build: hipcc --offload-arch=gfx950 -O3 -o sweep_modes sweep_modes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double2 cplx;

// column tile: thread (j = tid >> 4, t = tid & 15); t < 8: column t of X, else column t-8 of Y; rows j + 16 k
__global__ __launch_bounds__(256, 2) void k_col(const cplx *sx, const cplx *sy, cplx *dx, cplx *dy, size_t pitch, size_t frame)
{
    const int tid = threadIdx.x, t = tid & 15, j = tid >> 4;
    const cplx *s = (t < 8 ? sx : sy) + (size_t)blockIdx.y * frame + (size_t)blockIdx.x * 8 + (t & 7);
    cplx *d = (t < 8 ? dx : dy) + (size_t)blockIdx.y * frame + (size_t)blockIdx.x * 8 + (t & 7);
    cplx v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = s[(size_t)(j + 16 * k) * pitch];
#pragma unroll
    for (int k = 0; k < 16; k++) { v[k].x += 1.0; d[(size_t)(j + 16 * k) * pitch] = v[k]; }
}
// row pass: one workgroup = ROWS rows of both polarisations, n2 points each (n2 a multiple of 256)
__global__ __launch_bounds__(256, 2) void k_rows(const cplx *sx, const cplx *sy, cplx *dx, cplx *dy, size_t pitch, size_t frame, int n2, int rows)
{
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.y * frame + (size_t)blockIdx.x * rows * pitch;
    const int per = n2 / 256 * rows;        // points per thread and polarisation (<= 16 here)
    cplx vx[16], vy[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k < per) {
            const int e = tid + 256 * k, r = e / n2, i = e - r * n2;
            vx[k] = sx[base + (size_t)r * pitch + i];
            vy[k] = sy[base + (size_t)r * pitch + i];
        }
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k < per) {
            const int e = tid + 256 * k, r = e / n2, i = e - r * n2;
            vx[k].x += 1.0; vy[k].y += 1.0;
            dx[base + (size_t)r * pitch + i] = vx[k];
            dy[base + (size_t)r * pitch + i] = vy[k];
        }
}
__global__ __launch_bounds__(256) void k_copy(const double4 *s, double4 *d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

static float timed(hipEvent_t e0, hipEvent_t e1) { float ms; hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms; }

int main(int argc, char **argv)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t cap = (size_t)5 << 30;           // two buffers of X|Y each
    cplx *A, *B;
    if (hipMalloc(&A, cap) != hipSuccess || hipMalloc(&B, cap) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(A, 0, cap); hipMemset(B, 0, cap);
    struct G { const char *name; int n1, n2; int pad; int F; };
    std::vector<G> gs = {
        {"C1  256x256   pitch 4 KiB          ", 256, 256, 0, 1024},
        {"C1  256x256   pitch 4 KiB + 128 B  ", 256, 256, 8, 1024},
        {"C4  256x4096  pitch 64 KiB         ", 256, 4096, 0, 64},
        {"C4  256x4096  pitch 64 KiB + 128 B ", 256, 4096, 8, 64},
        {"C4  256x4096  pitch 64 KiB + 256 B ", 256, 4096, 16, 64},
        {"C4  256x4096  pitch 64 KiB + 512 B ", 256, 4096, 32, 64},
        {"C4  256x4096  pitch 64 KiB + 1 KiB ", 256, 4096, 64, 64},
        {"C4  256x4096  pitch 64 KiB + 4 KiB ", 256, 4096, 256, 64},
        {"C4  256x4096  pitch 64 KiB + 4.125K", 256, 4096, 264, 64},
        {"C4 16 frames  pitch 64 KiB         ", 256, 4096, 0, 16},
        {"C4 16 frames  pitch 64 KiB + 128 B ", 256, 4096, 8, 16},
        {"C4 16 frames  pitch 64 KiB + 4.125K", 256, 4096, 264, 16},
        {"C1  F=512                          ", 256, 256, 0, 512},
        {"C1  F=256                          ", 256, 256, 0, 256},
        {"C1  F=128  (256 MiB)               ", 256, 256, 0, 128},
        {"C1  F=96   (192 MiB)               ", 256, 256, 0, 96},
        {"C1  F=64   (128 MiB)               ", 256, 256, 0, 64},
        {"C1  F=32   (64 MiB)                ", 256, 256, 0, 32},
        {"C1  F=16   (32 MiB = the L2s)      ", 256, 256, 0, 16},
    };
    printf("%-38s | %-27s | %-27s | %-27s\n", "geometry (us per sweep, TB/s of 64 B/sample)", "column sweep in / out of place", "row sweep in / out of place", "pair in place / ping-pong");
    for (auto &g : gs) {
        const size_t pitch = (size_t)g.n2 + g.pad, frame = pitch * g.n1, tot = frame * g.F;
        if (2 * tot * sizeof(cplx) > cap) { printf("%s: too large\n", g.name); continue; }
        cplx *ax = A, *ay = A + tot, *bx = B, *by = B + tot;
        const double bytes = 64.0 * g.n1 * g.n2 * g.F;
        const dim3 gc(g.n2 / 8, g.F), gr(g.n1 / (g.n2 >= 4096 ? 1 : 8), g.F);
        const int rows = g.n2 >= 4096 ? 1 : 8;
        auto col = [&](bool oop) { hipLaunchKernelGGL(k_col, gc, dim3(256), 0, 0, ax, ay, oop ? bx : ax, oop ? by : ay, pitch, frame); };
        auto row = [&](bool oop, bool back) {
            const cplx *sx = back ? bx : ax, *sy = back ? by : ay;
            cplx *dx = oop ? (back ? ax : bx) : (cplx *)sx, *dy = oop ? (back ? ay : by) : (cplx *)sy;
            hipLaunchKernelGGL(k_rows, gr, dim3(256), 0, 0, sx, sy, dx, dy, pitch, frame, g.n2, rows);
        };
        float t[6];
        for (int m = 0; m < 6; m++) {
            float best = 1e9;
            for (int it = 0; it < 5; it++) {
                const int reps = 6;
                hipEventRecord(e0, 0);
                for (int r = 0; r < reps; r++) {
                    if (m == 0) col(false);
                    if (m == 1) col(true);
                    if (m == 2) row(false, false);
                    if (m == 3) row(true, false);
                    if (m == 4) { col(false); row(false, false); }
                    if (m == 5) { col(true); row(true, true); }     // A -> B, B -> A
                }
                hipEventRecord(e1, 0);
                const float ms = timed(e0, e1) / reps;
                if (it && ms < best) best = ms;
            }
            t[m] = best;
        }
        auto tb = [&](float ms, double b) { return b / (ms * 1e-3) / 1e12; };
        printf("%-38s | %7.1f %5.2f  %7.1f %5.2f | %7.1f %5.2f  %7.1f %5.2f | %7.1f %5.2f  %7.1f %5.2f\n", g.name, t[0] * 1e3, tb(t[0], bytes), t[1] * 1e3, tb(t[1], bytes),
               t[2] * 1e3, tb(t[2], bytes), t[3] * 1e3, tb(t[3], bytes), t[4] * 1e3, tb(t[4], 2 * bytes), t[5] * 1e3, tb(t[5], 2 * bytes));
        fflush(stdout);
    }
    {   // the part's copy rate, for scale (2 GiB -> 2 GiB)
        const size_t n = ((size_t)2 << 30) / sizeof(double4);
        float best = 1e9;
        for (int it = 0; it < 5; it++) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const double4 *)A, (double4 *)B, n);
            hipEventRecord(e1, 0);
            const float ms = timed(e0, e1);
            if (it && ms < best) best = ms;
        }
        printf("copy 2 GiB -> 2 GiB (32 B per lane): %.1f us, %.2f TB/s (read + written)\n", best * 1e3, 2.0 * n * sizeof(double4) / (best * 1e-3) / 1e12);
    }
    return 0;
}

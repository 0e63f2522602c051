// dev microbenchmark (round 3): can the column sweep be walked in STRIPS -- one wave = 256 rows x (2 ux + 2 uy) columns,
// 32-byte segments per row and polarisation -- without losing memory-side rate against k_colx16's tile shape (256 rows x
// (8 + 8) columns per workgroup, 128-byte segments)?  The four waves of a workgroup cover the same 128-byte lines at about
// the same time: whether the L2 merges their quarter-line reads and writes is what this measures (in place, no arithmetic).
//   L/S = 0: tile shape (thread (j = tid >> 4, t = tid & 15), rows j + 16 k)      -- as sweep_modes.hip's k_col
//   L/S = 1: strip shape (wave w: columns 2w, 2w+1; lane = q + 4 jj, q = 2 pol + c, rows jj + 16 k)
// build: hipcc --offload-arch=gfx950 -O3 -o strip_modes strip_modes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double2 cplx;

template <int LS, int SS> __global__ __launch_bounds__(256, 2) void k_col(cplx *x, cplx *y, size_t pitch, size_t frame)
{
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.y * frame + (size_t)blockIdx.x * 8;
    const int t = tid & 15, j = tid >> 4;
    const int w = tid >> 6, lane = tid & 63, q = lane & 3, jj = lane >> 2;
    cplx *const pt = (t < 8 ? x : y) + base + (t & 7) + (size_t)j * pitch;
    cplx *const ps = (q < 2 ? x : y) + base + 2 * w + (q & 1) + (size_t)jj * pitch;
    cplx *const pl = LS ? ps : pt, *const pst = SS ? ps : pt;
    cplx v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = pl[(size_t)(16 * k) * pitch];
#pragma unroll
    for (int k = 0; k < 16; k++) { v[k].x += 1.0; pst[(size_t)(16 * k) * pitch] = v[k]; }
}
// one-wave workgroups: the four strips of a line group are workgroups b, b + 8, b + 16, b + 24 of a block of 32 (the same XCD
// under the round-robin dealing of workgroups to XCDs)
__global__ __launch_bounds__(64, 2) void k_col_w(cplx *x, cplx *y, size_t pitch, size_t frame)
{
    const int lane = threadIdx.x, q = lane & 3, jj = lane >> 2;
    const int b = blockIdx.x, grp = b >> 5, i = b & 31, w = i >> 3, lg = (grp << 3) + (i & 7);
    const size_t base = (size_t)blockIdx.y * frame + (size_t)lg * 8;
    cplx *const ps = (q < 2 ? x : y) + base + 2 * w + (q & 1) + (size_t)jj * pitch;
    cplx v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = ps[(size_t)(16 * k) * pitch];
#pragma unroll
    for (int k = 0; k < 16; k++) { v[k].x += 1.0; ps[(size_t)(16 * k) * pitch] = v[k]; }
}

static float timed(hipEvent_t e0, hipEvent_t e1) { float ms; hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms; }

int main()
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t cap = (size_t)5 << 30;
    cplx *A;
    if (hipMalloc(&A, cap) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(A, 0, cap);
    struct G { const char *name; int n1, n2, pad, F; };
    std::vector<G> gs = {
        {"C1  256x256   F=1024", 256, 256, 0, 1024},
        {"C4  256x4096  F=64  ", 256, 4096, 0, 64},
        {"C4  256x4096  F=16  ", 256, 4096, 0, 16},
    };
    printf("%-22s | us per sweep, TB/s of 64 B per sample: tile/tile  strip-load/tile-store  tile-load/strip-store  strip/strip  one-wave strips\n", "geometry (in place)");
    for (auto &g : gs) {
        const size_t pitch = (size_t)g.n2 + g.pad, frame = pitch * g.n1, tot = frame * g.F;
        cplx *ax = A, *ay = A + tot;
        const double bytes = 64.0 * g.n1 * g.n2 * g.F;
        const dim3 gc(g.n2 / 8, g.F), gw(g.n2 / 2, g.F);
        float t[5];
        for (int m = 0; m < 5; m++) {
            float best = 1e9;
            for (int it = 0; it < 5; it++) {
                const int reps = 6;
                hipEventRecord(e0, 0);
                for (int r = 0; r < reps; r++) {
                    if (m == 0) hipLaunchKernelGGL((k_col<0, 0>), gc, dim3(256), 0, 0, ax, ay, pitch, frame);
                    if (m == 1) hipLaunchKernelGGL((k_col<1, 0>), gc, dim3(256), 0, 0, ax, ay, pitch, frame);
                    if (m == 2) hipLaunchKernelGGL((k_col<0, 1>), gc, dim3(256), 0, 0, ax, ay, pitch, frame);
                    if (m == 3) hipLaunchKernelGGL((k_col<1, 1>), gc, dim3(256), 0, 0, ax, ay, pitch, frame);
                    if (m == 4) hipLaunchKernelGGL(k_col_w, gw, dim3(64), 0, 0, ax, ay, pitch, frame);
                }
                hipEventRecord(e1, 0);
                const float ms = timed(e0, e1) / reps;
                if (it && ms < best) best = ms;
            }
            t[m] = best;
        }
        printf("%-22s |", g.name);
        for (int m = 0; m < 5; m++) printf("  %7.1f %5.2f", t[m] * 1e3, bytes / (t[m] * 1e-3) / 1e12);
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

// hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny host-side emulation of the HIP execution model (grid / workgroup /
// 64-lane wave, LDS, barriers, wave shuffles, atomics, streams as in-order
// queues) so that the *unmodified* kernel sources under polmux_amd/csrc can be
// compiled with g++ (-DPLX_EMU) and run under ASan/UBSan on the CPU
// (GPU AddressSanitizer is not available on the GPU pool).  It is NOT a CPU
// fallback: polmux_amd never loads the emulated library; only
// tests/test_emu_*.py do, to catch indexing / synchronisation bugs before a
// kernel ever reaches a GPU.  One OS thread per work-item, one workgroup at a
// time.
#pragma once
#include <atomic>
#include <barrier>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
struct int2 { int x, y; };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __restrict__

typedef int hipError_t;
typedef void *hipStream_t;
struct emu_event { double t; };
typedef emu_event *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0 };

namespace emu {
struct BlockCtx {
    dim3 grid, block;
    unsigned nthreads = 0;
    std::unique_ptr<std::barrier<>> bar;
    std::vector<std::unique_ptr<std::barrier<>>> wave_bar;
    std::vector<uint64_t> xchg; // per-thread 8-byte exchange slot (2 slots for 16 B)
    std::vector<char> lds;
};
extern thread_local BlockCtx *t_ctx;   // the workgroup this OS thread belongs to
extern thread_local dim3 t_threadIdx, t_blockIdx;
extern thread_local unsigned t_linear;
// number of workgroups executed concurrently (default 1).  Kernels whose workgroups wait for each
// other inside a launch (frame barrier of the fused SSFM column sweep) need their partners alive.
extern int g_concurrency;
bool starve_barriers();   // PLX_EMU_STARVE is set: a test keeps a frame's workgroups from being alive together (the barrier must time out)
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
inline char *dyn_lds() { return t_ctx->lds.data(); }
} // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_ctx->block)
#define gridDim (emu::t_ctx->grid)

static inline void __syncthreads() { emu::t_ctx->bar->arrive_and_wait(); }

template <class T> static inline T emu_shfl_src(T v, int src_lane)
{
    static_assert(sizeof(T) <= 8, "shuffle of <= 8 bytes");
    unsigned lin = emu::t_linear, wave = lin / 64, lane = lin % 64;
    uint64_t bits = 0;
    std::memcpy(&bits, &v, sizeof(T));
    emu::BlockCtx &cx = *emu::t_ctx;
    cx.xchg[lin] = bits;
    cx.wave_bar[wave]->arrive_and_wait();
    unsigned src = wave * 64 + (unsigned)(src_lane & 63);
    uint64_t got = src < cx.nthreads ? cx.xchg[src] : bits;
    cx.wave_bar[wave]->arrive_and_wait();
    T r;
    std::memcpy(&r, &got, sizeof(T));
    (void)lane;
    return r;
}
// up to four doubles exchanged in ONE rendezvous of the wave: out[i] = value i of lane src[i] (the permlane-swap idioms of
// plx_common.h move a complex pair at a time; four separate shuffles would be eight barrier rounds of 64 host threads)
static inline void emu_shfl4(const double *v, const int *src, double *out, int n)
{
    unsigned lin = emu::t_linear, wave = lin / 64;
    emu::BlockCtx &cx = *emu::t_ctx;
    for (int i = 0; i < n; i++) std::memcpy(&cx.xchg[(size_t)i * cx.nthreads + lin], &v[i], 8);
    cx.wave_bar[wave]->arrive_and_wait();
    for (int i = 0; i < n; i++) {
        unsigned s_ = wave * 64 + (unsigned)(src[i] & 63);
        if (s_ < cx.nthreads) std::memcpy(&out[i], &cx.xchg[(size_t)i * cx.nthreads + s_], 8); else out[i] = v[i];
    }
    cx.wave_bar[wave]->arrive_and_wait();
}
template <class T> static inline T __shfl_xor(T v, int mask, int = 64) { return emu_shfl_src(v, (int)(emu::t_linear % 64) ^ mask); }
template <class T> static inline T __shfl_down(T v, int d, int = 64)
{
    int lane = (int)(emu::t_linear % 64);
    return emu_shfl_src(v, lane + d < 64 ? lane + d : lane);
}
template <class T> static inline T __shfl(T v, int src, int = 64) { return emu_shfl_src(v, src); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline int __any(int pred)
{
    int v = pred ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v |= __shfl_xor(v, m, 64);
    return v;
}
static inline int __all(int pred)
{
    int v = pred ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v &= __shfl_xor(v, m, 64);
    return v;
}

static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v)
{
    auto *a = reinterpret_cast<std::atomic<unsigned long long> *>(p);
    unsigned long long old = a->load();
    while (old < v && !a->compare_exchange_weak(old, v)) {}
    return old;
}
static inline int atomicAdd(int *p, int v) { return reinterpret_cast<std::atomic<int> *>(p)->fetch_add(v); }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return reinterpret_cast<std::atomic<unsigned> *>(p)->fetch_add(v); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v)
{
    return reinterpret_cast<std::atomic<unsigned long long> *>(p)->fetch_add(v);
}
static inline unsigned emu_atomic_load_u32(const unsigned *p) { return reinterpret_cast<const std::atomic<unsigned> *>(p)->load(); }
static inline void emu_atomic_store_u32(unsigned *p, unsigned v) { reinterpret_cast<std::atomic<unsigned> *>(p)->store(v); }
static inline unsigned long long emu_atomic_load_u64(const unsigned long long *p) { return reinterpret_cast<const std::atomic<unsigned long long> *>(p)->load(); }
static inline void emu_atomic_store_u64(unsigned long long *p, unsigned long long v) { reinterpret_cast<std::atomic<unsigned long long> *>(p)->store(v); }
// sincos(): glibc's (declared by <cmath> under _GNU_SOURCE)
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline long long __double_as_longlong(double d) { long long r; std::memcpy(&r, &d, 8); return r; }
static inline double __longlong_as_double(long long l) { double r; std::memcpy(&r, &l, 8); return r; }
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }   // (never contracted into an fma)
static inline double __dsub_rn(double a, double b) { volatile double r = a - b; return r; }

// ---- runtime subset -------------------------------------------------------------
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void *p) { return hipFree(p); }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
// (PLX_EMU_CUS: the emulated device's CU count -- a test sets 1 so that the persistent kernels loop over several tiles)
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int)
{
    const char *e = std::getenv("PLX_EMU_CUS");
    *v = e ? std::atoi(e) : 256;
    return hipSuccess;
}
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event{0}; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
double emu_now_ms();
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = emu_now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }

#define PLX_EMU_LAUNCH(kern, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (shmem), [=]() { kern(__VA_ARGS__); })

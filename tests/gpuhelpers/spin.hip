// TEST INFRASTRUCTURE ONLY (tests/test_gpu_configs.py): a kernel that holds part of the GPU for a bounded time, standing in
// for "another process / another long-running kernel on the device".  Each workgroup claims `lds_bytes` of LDS and spins on
// the constant-rate wall clock until its deadline -- an exit condition every wave reaches whatever else happens.
#include <hip/hip_runtime.h>

__global__ void k_spin(long long ticks, int *sink)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const long long t0 = (long long)wall_clock64();
    int acc = 0;
    while ((long long)wall_clock64() - t0 < ticks) {
        acc += lds[(threadIdx.x * 64) & 1023];
        __builtin_amdgcn_s_sleep(32);
    }
    if (acc == 0x7fffffff) *sink = acc;      // (keeps the LDS read alive)
}

extern "C" int plx_test_spin(int workgroups, int threads, int lds_bytes, double seconds, void *stream)
{
    static int *sink = nullptr;
    if (!sink && hipMalloc((void **)&sink, sizeof(int)) != hipSuccess) return -1;
    if (seconds > 5.0) seconds = 5.0;        // bounded by construction
    if (hipFuncSetAttribute((const void *)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -2;
    hipLaunchKernelGGL(k_spin, dim3((unsigned)workgroups), dim3((unsigned)threads), (size_t)lds_bytes, (hipStream_t)stream,
                       (long long)(seconds * 1e8), sink);           // wall_clock64: 100 MHz on gfx950
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// the host waits for the spinner (a helper process keeps the device until its kernel has finished)
extern "C" int plx_test_spin_wait(void) { return hipDeviceSynchronize() == hipSuccess ? 0 : -1; }

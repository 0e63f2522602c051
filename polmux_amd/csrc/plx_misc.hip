// plx_misc.hip -- error plumbing, device selection, fastexp and small helpers.
#include "../../include/polmux_hip.h"
#include "plx_common.h"

#include <vector>

static thread_local std::string g_err;
void plx_set_error(const std::string &msg) { g_err = msg; }

extern "C" const char *plx_last_error(void) { return g_err.c_str(); }
extern "C" int plx_abi_version(void) { return 1000; }

extern "C" int plx_device_count(int *count)
{
    if (!count) PLX_FAIL(PLX_ERR_ARG, "plx_device_count: null argument");
    PLX_HIP(hipGetDeviceCount(count));
    return PLX_OK;
}

extern "C" int plx_set_device(int device)
{
    PLX_HIP(hipSetDevice(device));
    return PLX_OK;
}

namespace {
// fastexp.c:37-44: y = cos(x) + i sin(x).  8 B in, 16 B out per element: a pure
// streaming kernel (sincos is the only arithmetic), grid-stride over the array.
__global__ __launch_bounds__(256) void k_fastexp(const double *__restrict__ x, cplx *__restrict__ y, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = cexpi(x[i]);
}

__global__ __launch_bounds__(256) void k_pick(const cplx *__restrict__ in, cplx *__restrict__ out, int64_t n_in,
                                              int64_t n_out, int64_t offset, int64_t stride, double scale,
                                              int64_t out_pitch)
{
    const int sig = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * blockDim.x)
        out[(size_t)sig * out_pitch + i] = cscale(in[(size_t)sig * n_in + offset + i * stride], scale);
}
} // namespace

static unsigned grid_for(size_t n)
{
    size_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

extern "C" int plx_fastexp_dev(const double *d_x, double *d_y, size_t count, void *stream)
{
    if (count == 0) return PLX_OK;
    if (!d_x || !d_y) PLX_FAIL(PLX_ERR_ARG, "plx_fastexp_dev: null argument");
    PLX_LAUNCH(k_fastexp, dim3(grid_for(count)), dim3(256), 0, stream, d_x, (cplx *)d_y, count);
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_fastexp(const double *x, double *yr, double *yi, size_t count)
{
    if (count == 0) return PLX_OK;
    if (!x || !yr || !yi) PLX_FAIL(PLX_ERR_ARG, "plx_fastexp: null argument");
    double *dx = nullptr, *dy = nullptr;
    PLX_HIP(hipMalloc((void **)&dx, count * sizeof(double)));
    if (hipMalloc((void **)&dy, 2 * count * sizeof(double)) != hipSuccess) {
        hipFree(dx);
        PLX_FAIL(PLX_ERR_HIP, "plx_fastexp: device allocation failed");
    }
    std::vector<double> h(2 * count);
    int rc = PLX_OK;
    if (hipMemcpy(dx, x, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = PLX_ERR_HIP;
    if (!rc) rc = plx_fastexp_dev(dx, dy, count, nullptr);
    if (!rc && hipMemcpy(h.data(), dy, 2 * count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = PLX_ERR_HIP;
    hipFree(dx);
    hipFree(dy);
    if (rc) {
        if (rc == PLX_ERR_HIP) plx_set_error("plx_fastexp: HIP transfer failed");
        return rc;
    }
    for (size_t i = 0; i < count; i++) { yr[i] = h[2 * i]; yi[i] = h[2 * i + 1]; }
    return PLX_OK;
}

extern "C" int plx_pick_dev(const double *d_in, double *d_out, int64_t n_in, int64_t n_out, int64_t offset,
                            int64_t stride, double scale, int nsig, int64_t out_pitch, void *stream)
{
    if (out_pitch == 0) out_pitch = n_out;
    if (out_pitch < n_out) PLX_FAIL(PLX_ERR_ARG, "plx_pick_dev: out_pitch smaller than n_out");
    if (!d_in || !d_out) PLX_FAIL(PLX_ERR_ARG, "plx_pick_dev: null argument");
    if (n_out < 1 || nsig < 1 || stride < 1 || offset < 0 || offset + (n_out - 1) * stride >= n_in)
        PLX_FAIL(PLX_ERR_ARG, "plx_pick_dev: selection out of range");
    PLX_LAUNCH(k_pick, dim3(grid_for((size_t)n_out), (unsigned)nsig), dim3(256), 0, stream, (const cplx *)d_in,
               (cplx *)d_out, n_in, n_out, offset, stride, scale, out_pitch);
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

// plx_misc.hip -- error plumbing, device selection, fastexp and small helpers.
#include "../../include/polmux_hip.h"
#include "plx_common.h"
#include "plx_gateway.h"

#include <cstring>

#include <vector>

static thread_local std::string g_err;
void plx_set_error(const std::string &msg) { g_err = msg; }

extern "C" const char *plx_last_error(void) { return g_err.c_str(); }
extern "C" int plx_abi_version(void) { return 1002; }

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
// ---- ampliflat.m:78-148 : flat gain + ASE (the step between spans) -------------------
// Philox-4x32-10 (Salmon et al., SC'11), counter = (sample, column, polarisation, 0), key = (seed, frame
// key): one call -> one complex normal sample by Box-Muller.  Counter-based, so the noise of a realisation
// depends only on its key, never on how frames are batched or sharded over GPUs.
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t *out)
{
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct AmpArgs {
    cplx *ux, *uy;          // [frame][nfc][nfft]
    const cplx *noise;      // optional injected unit noise [frame][2*nfc][nfft] ([X cols | Y cols], ampliflat.m:123-129)
    const double *sigma;    // [nfc] sqrt(mW), ampliflat.m:91-102 (device copy, only when nfc > 32)
    double sig_v[32];       // the same by value: no allocation, no synchronisation on the usual path
    int has_sigma;
    const int64_t *keys;    // optional per-frame RNG keys
    int64_t nfft;
    int nfc, asex, asey;
    double sqrt_gain;
    uint64_t seed;
};

__global__ __launch_bounds__(256) void k_ampliflat(AmpArgs a)
{
    const int f = blockIdx.z, c = blockIdx.y;
    const size_t base = ((size_t)f * a.nfc + c) * (size_t)a.nfft;
    const double sg = a.sqrt_gain, sig = !a.has_sigma ? 0.0 : (a.sigma ? a.sigma[c] : a.sig_v[c]);
    const uint64_t key = a.keys ? (uint64_t)a.keys[f] : (uint64_t)f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.nfft; i += (int64_t)gridDim.x * blockDim.x) {
        for (int pol = 0; pol < 2; pol++) {
            cplx *u = pol ? a.uy : a.ux;
            if (!u) continue;
            cplx v = cscale(u[base + i], sg);                       // FIELD*sqrt(gain) :78-82
            const bool on = sig != 0.0 && (pol ? a.asey : a.asex);
            if (on) {
                cplx n;
                if (a.noise) {
                    n = a.noise[((size_t)f * 2 * a.nfc + (size_t)pol * a.nfc + c) * (size_t)a.nfft + i];
                } else {
                    uint32_t r[4];
                    philox4x32((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)c, (uint32_t)pol,
                               (uint32_t)(a.seed ^ key), (uint32_t)((a.seed >> 32) ^ (key * 0x9E3779B97F4A7C15ull >> 32)), r);
                    const double u1 = ((double)(((uint64_t)r[0] << 21) ^ (r[1] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
                    const double u2 = ((double)(((uint64_t)r[2] << 21) ^ (r[3] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
                    const double rad = sqrt(-2.0 * log(u1));
                    double sn, cs;
                    sincos(6.28318530717958647692 * u2, &sn, &cs);
                    n = make_double2(rad * cs, rad * sn);               // randn + i*randn :131-136
                }
                v = make_double2(v.x + sig * n.x, v.y + sig * n.y);
            }
            u[base + i] = v;
        }
    }
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
    // device buffers and pinned staging are the library's (plx_gateway.h): a call allocates nothing once they have grown
    std::lock_guard<std::mutex> lk(plxgw::mutex());
    plxgw::count_call();
    double *h = (double *)plxgw::pinned(plxgw::S_IN, 2 * count * sizeof(double));
    double *dx = (double *)plxgw::dev(plxgw::S_IN, count * sizeof(double)), *dy = (double *)plxgw::dev(plxgw::S_OUT, 2 * count * sizeof(double));
    if (!h || !dx || !dy) return PLX_ERR_HIP;
    std::memcpy(h, x, count * sizeof(double));
    if (hipMemcpyAsync(dx, h, count * sizeof(double), hipMemcpyHostToDevice, nullptr) != hipSuccess) PLX_FAIL(PLX_ERR_HIP, "plx_fastexp: HIP transfer failed");
    int rc = plx_fastexp_dev(dx, dy, count, nullptr);
    if (rc) return rc;
    if (hipMemcpyAsync(h, dy, 2 * count * sizeof(double), hipMemcpyDeviceToHost, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)
        PLX_FAIL(PLX_ERR_HIP, "plx_fastexp: HIP transfer failed");
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

extern "C" int plx_ampliflat_dev(double *d_ux, double *d_uy, int64_t nfft, int32_t nfc, int nframes, double gain_lin,
                                 const double *sigma, const double *d_noise, uint64_t seed, const int64_t *d_keys,
                                 int32_t asex, int32_t asey, void *stream)
{
    if (!d_ux || nfft < 1 || nfc < 1 || nframes < 1) PLX_FAIL(PLX_ERR_ARG, "plx_ampliflat_dev: bad argument");
    if (!(gain_lin >= 0)) PLX_FAIL(PLX_ERR_ARG, "plx_ampliflat_dev: gain must be >= 0");
    double *d_sigma = nullptr;
    AmpArgs a;
    a.ux = (cplx *)d_ux; a.uy = (cplx *)d_uy; a.noise = (const cplx *)d_noise; a.keys = d_keys; a.nfft = nfft;
    a.nfc = nfc; a.asex = asex; a.asey = asey; a.sqrt_gain = sqrt(gain_lin); a.seed = seed; a.sigma = nullptr;
    a.has_sigma = sigma ? 1 : 0;
    for (int c = 0; c < 32; c++) a.sig_v[c] = (sigma && c < nfc) ? sigma[c] : 0.0;
    if (sigma && nfc > 32) {
        PLX_HIP(hipMalloc((void **)&d_sigma, sizeof(double) * nfc));
        if (hipMemcpyAsync(d_sigma, sigma, sizeof(double) * nfc, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
            hipFree(d_sigma);
            PLX_FAIL(PLX_ERR_HIP, "plx_ampliflat_dev: upload failed");
        }
        a.sigma = d_sigma;
    }
    unsigned gx = (unsigned)((nfft + 255) / 256);
    if (gx > 256) gx = 256;
    PLX_LAUNCH(k_ampliflat, dim3(gx, (unsigned)nfc, (unsigned)nframes), dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (d_sigma) { hipStreamSynchronize((hipStream_t)stream); hipFree(d_sigma); }
    if (e != hipSuccess) PLX_FAIL(PLX_ERR_HIP, "plx_ampliflat_dev: launch failed");
    return PLX_OK;
}

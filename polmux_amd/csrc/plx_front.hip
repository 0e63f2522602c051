// plx_front.hip -- coherent front end between the fibre and the CD equaliser on gfx950.
//
// Reference: /root/reference/receiver_cohmix.m:165-307 (optical filter + post-compensation, LO mixing in two
// 90-degree hybrids, balanced/normal photodetection, electrical low-pass) and RxPdmCohQpsk.m:36-72 (ADC
// quantisation, timing shift, decimation to 1 or 2 samples per symbol, I/Q recombination).
//
// MI355X design: the two filters run on the SSFM plan's batched four-step FFT engine (three in-place HBM
// sweeps each, plx_ssfm_filter_dev); the photocurrents of one polarisation are carried as ONE complex
// signal I + jQ, so the electrical filter  real(ifft(fft(I).*H)), real(ifft(fft(Q).*H))  (two real
// transforms per polarisation, :300-304) is a single complex pass with the Hermitian part of H,
//   He(f) = (H(f) + conj(H(-f))) / 2,
// whose impulse response is real.  Mixing is one element-wise sweep; the ADC maximum is one read sweep;
// quantisation, the circular timing shift and the decimating FIR are fused into the final gather, which
// touches only the 17 taps around every r-th sample.
#include "plx_internal.h"
#include "plx_gateway.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct FrontArgs {
    cplx *ux, *uy;
    const cplx *elo;      // [n] or null
    double elo_s;
    int64_t n;
    int balanced, dual;
};

// One hybrid + photodiodes, literally as receiver_cohmix.m:254-274: the four mixer outputs
//   j*s + j*Elo,  s - Elo,  j*s - Elo,  -s + j*Elo
// are squared (real(E.*conj(E))) and subtracted pairwise (balanced) or fields 1 and 3 are kept (normal).
__device__ __forceinline__ cplx hybrid(cplx s, cplx lo, int balanced)
{
    const cplx js = make_double2(-s.y, s.x), jlo = make_double2(-lo.y, lo.x);
    const cplx e1 = cadd(js, jlo), e2 = csub(s, lo), e3 = csub(js, lo), e4 = make_double2(-s.x + jlo.x, -s.y + jlo.y);
    const double i1 = e1.x * e1.x + e1.y * e1.y, i2 = e2.x * e2.x + e2.y * e2.y;
    const double i3 = e3.x * e3.x + e3.y * e3.y, i4 = e4.x * e4.x + e4.y * e4.y;
    return balanced ? make_double2(i1 - i2, i3 - i4) : make_double2(i1, i3);
}

__global__ __launch_bounds__(256) void k_mix(FrontArgs a)
{
    const size_t base = (size_t)blockIdx.y * a.n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx lo = a.elo ? a.elo[i] : make_double2(a.elo_s, 0.0);
        a.ux[base + i] = hybrid(a.ux[base + i], lo, a.balanced);
        if (a.dual) a.uy[base + i] = hybrid(a.uy[base + i], lo, a.balanced);
    }
}

// M = max(max(abs(Irx))) over the 2 or 4 current columns of a frame, RxPdmCohQpsk.m:37
__global__ __launch_bounds__(256) void k_absmax(FrontArgs a, unsigned long long *dmax)
{
    PLX_DYN_LDS(lds);
    double *red = (double *)lds;
    const size_t base = (size_t)blockIdx.y * a.n;
    double m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx x = a.ux[base + i];
        m = fmax(m, fmax(fabs(x.x), fabs(x.y)));
        if (a.dual) {
            const cplx y = a.uy[base + i];
            m = fmax(m, fmax(fabs(y.x), fabs(y.y)));
        }
    }
    for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) m = fmax(m, red[w]);
        atomicMax(dmax + blockIdx.y, (unsigned long long)__double_as_longlong(m));
    }
}

struct DecArgs {
    const cplx *ux, *uy;
    cplx *out;            // [F][npol][nout]
    const unsigned long long *dmax;
    int64_t n, nout;
    int r, ntaps, gd;     // decimation rate, FIR length, group delay (ntaps-1)/2
    int adcbits;
    int64_t shift[2];     // fastshift amount per polarisation (RxPdmCohQpsk.m:42-44)
    double fir[64];
};

// ADC emulation, RxPdmCohQpsk.m:36-40, same operation order:
//   round((I + M)/2 ./ M * 2^bits) * 2 .* M / 2^bits - M
__device__ __forceinline__ double adc(double v, double M, double lv)
{
    return round((v + M) / 2 / M * lv) * 2 * M / lv - M;
}

// Fused ADC + fastshift + decimating FIR.  Definition (decimate(x,r,16,'fir') is MathWorks code that is not
// part of the reference, see DESIGN.md "front end"): the shifted sequence xs[i] = x[(i - shift) mod n] is
// extended by odd reflection about both end points, filtered causally with the ntaps-tap FIR (oldest sample
// accumulated first) and the output is taken at 0-based positions gd + m*r, m = 0 .. ceil(n/r)-1.
__global__ __launch_bounds__(256) void k_decimate(DecArgs a)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.nout) return;
    const int f = blockIdx.y, pol = blockIdx.z, npol = gridDim.z;
    const cplx *x = (pol ? a.uy : a.ux) + (size_t)f * a.n;
    const double M = __longlong_as_double((long long)a.dmax[f]);
    const double lv = (double)(1u << a.adcbits);
    const bool q = a.adcbits > 0 && M > 0;
    const int64_t n = a.n, sh = a.shift[pol];
    auto xs = [&](int64_t i) -> cplx {       // quantised, circularly shifted sample i, 0 <= i < n
        int64_t src = (i - sh) % n;
        if (src < 0) src += n;
        cplx v = x[src];
        if (q) { v.x = adc(v.x, M, lv); v.y = adc(v.y, M, lv); }
        return v;
    };
    const cplx first = xs(0), last = xs(n - 1);
    double accr = 0, acci = 0;
    const int64_t p = a.gd + m * a.r;
    for (int k = a.ntaps - 1; k >= 0; k--) {
        const int64_t i = p - k;
        cplx v;
        if (i < 0) { const cplx w = xs(-i < n ? -i : n - 1); v = make_double2(2 * first.x - w.x, 2 * first.y - w.y); }
        else if (i >= n) { const int64_t jx = 2 * (n - 1) - i; const cplx w = xs(jx > 0 ? jx : 0); v = make_double2(2 * last.x - w.x, 2 * last.y - w.y); }
        else v = xs(i);
        accr = accr + a.fir[k] * v.x;
        acci = acci + a.fir[k] * v.y;
    }
    a.out[((size_t)f * npol + pol) * a.nout + m] = make_double2(accr, acci);   // complex(I, Q), RxPdmCohQpsk.m:66-72
}

} // namespace

struct plx_front {
    plx_front_desc d;
    plx_ssfm *fft = nullptr;
    cplx *d_hopt = nullptr, *d_hel = nullptr, *d_elo = nullptr;
    unsigned long long *d_max = nullptr;
    std::vector<double> fir;
    int64_t nout = 0;
};

static void front_free(plx_front *P)
{
    if (!P) return;
    if (P->fft) plx_ssfm_destroy(P->fft);
    if (P->d_hopt) (void)hipFree(P->d_hopt);
    if (P->d_hel) (void)hipFree(P->d_hel);
    if (P->d_elo) (void)hipFree(P->d_elo);
    if (P->d_max) (void)hipFree(P->d_max);
    delete P;
}

extern "C" int plx_front_create(plx_front **out, const plx_front_desc *desc)
{
    if (!out || !desc) PLX_FAIL(PLX_ERR_ARG, "plx_front_create: null argument");
    *out = nullptr;
    if (!desc->hopt_re || !desc->hel_re) PLX_FAIL(PLX_ERR_ARG, "plx_front_create: filter tables are required");
    if (desc->decim < 1) PLX_FAIL(PLX_ERR_ARG, "plx_front_create: decimation rate must be >= 1");
    if (desc->decim > 1 && (!desc->fir || desc->ntaps < 1 || desc->ntaps > 63 || (desc->ntaps & 1) == 0))
        PLX_FAIL(PLX_ERR_ARG, "plx_front_create: the decimation FIR needs an odd number of taps in [1, 63]");
    if (desc->adcbits < 0 || desc->adcbits > 30) PLX_FAIL(PLX_ERR_ARG, "plx_front_create: adcbits outside [0, 30]");
    if (desc->max_frames < 1) PLX_FAIL(PLX_ERR_ARG, "plx_front_create: max_frames must be >= 1");
    const int64_t N = desc->nfft;
    plx_front *P = new plx_front();
    P->d = *desc;
    // the FFT engine: an SSFM plan with every physical effect off (its tables are not read by filter passes)
    std::vector<double> zeros((size_t)N, 0.0);
    double gam0 = 0.0;
    plx_ssfm_desc sd;
    std::memset(&sd, 0, sizeof(sd));
    sd.nfft = N; sd.nfc = 1; sd.dual_pol = desc->dual_pol ? 1 : 0; sd.max_frames = desc->max_frames;
    sd.dzmaxt = 1; sd.dphimaxt = 1; sd.length = 1; sd.nplates = 1; sd.gam = &gam0; sd.betat = zeros.data();
    int rc = plx_ssfm_create(&P->fft, &sd);
    if (rc != PLX_OK) { front_free(P); return rc; }
    // He = Hermitian part of the electrical filter (see file header); index of -f on the fft grid is (N-k) mod N
    std::vector<double> her((size_t)N), hei((size_t)N);
    for (int64_t k = 0; k < N; k++) {
        const int64_t km = (N - k) % N;
        const double ar = desc->hel_re[k], ai = desc->hel_im ? desc->hel_im[k] : 0.0;
        const double br = desc->hel_re[km], bi = desc->hel_im ? desc->hel_im[km] : 0.0;
        her[k] = 0.5 * (ar + br);
        hei[k] = 0.5 * (ai - bi);
    }
    rc = plx_ssfm_filter_table(P->fft, desc->hopt_re, desc->hopt_im, &P->d_hopt);
    if (rc == PLX_OK) rc = plx_ssfm_filter_table(P->fft, her.data(), hei.data(), &P->d_hel);
    if (rc != PLX_OK) { front_free(P); return rc; }
    if (desc->elo_re) {
        std::vector<cplx> lo((size_t)N);
        for (int64_t k = 0; k < N; k++) lo[k] = make_double2(desc->elo_re[k], desc->elo_im ? desc->elo_im[k] : 0.0);
        if (hipMalloc((void **)&P->d_elo, N * sizeof(cplx)) != hipSuccess ||
            hipMemcpy(P->d_elo, lo.data(), N * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess) {
            front_free(P);
            PLX_FAIL(PLX_ERR_HIP, "plx_front_create: device allocation/upload failed");
        }
    }
    if (hipMalloc((void **)&P->d_max, sizeof(unsigned long long) * desc->max_frames) != hipSuccess) {
        front_free(P);
        PLX_FAIL(PLX_ERR_HIP, "plx_front_create: device allocation failed");
    }
    if (desc->decim > 1) P->fir.assign(desc->fir, desc->fir + desc->ntaps);
    else P->fir.assign(1, 1.0);
    P->nout = (N + desc->decim - 1) / desc->decim;
    P->d.fir = nullptr; P->d.hopt_re = P->d.hopt_im = P->d.hel_re = P->d.hel_im = P->d.elo_re = P->d.elo_im = nullptr;
    *out = P;
    return PLX_OK;
}

extern "C" int plx_front_destroy(plx_front *P)
{
    front_free(P);
    return PLX_OK;
}

extern "C" int64_t plx_front_out_len(const plx_front *P) { return P ? P->nout : 0; }

extern "C" int plx_front_run_dev(plx_front *P, double *d_ux, double *d_uy, int nframes, const int64_t *shift,
                                 double *d_out, void *stream)
{
    if (!P || !d_ux || !d_out) PLX_FAIL(PLX_ERR_ARG, "plx_front_run_dev: null argument");
    if (nframes < 1 || nframes > P->d.max_frames) PLX_FAIL(PLX_ERR_ARG, "plx_front_run_dev: nframes outside [1, max_frames]");
    const int dual = P->d.dual_pol ? 1 : 0;
    if (dual && !d_uy) PLX_FAIL(PLX_ERR_ARG, "plx_front_run_dev: dual-polarisation plan needs d_uy");
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = P->d.nfft;
    int rc = plx_ssfm_filter_dev(P->fft, (cplx *)d_ux, (cplx *)d_uy, P->d_hopt, nframes, stream);   // :183, :232-233
    if (rc != PLX_OK) return rc;
    FrontArgs a;
    a.ux = (cplx *)d_ux; a.uy = (cplx *)d_uy; a.elo = P->d_elo; a.elo_s = P->d.elo_scalar; a.n = N;
    a.balanced = P->d.balanced; a.dual = dual;
    unsigned gx = (unsigned)((N + 255) / 256);
    if (gx > 256) gx = 256;
    PLX_LAUNCH(k_mix, dim3(gx, (unsigned)nframes), dim3(256), 0, st, a);
    rc = plx_ssfm_filter_dev(P->fft, (cplx *)d_ux, (cplx *)d_uy, P->d_hel, nframes, stream);        // :300-304
    if (rc != PLX_OK) return rc;
    PLX_HIP(hipMemsetAsync(P->d_max, 0, sizeof(unsigned long long) * nframes, st));
    if (P->d.adcbits > 0) {
        if (gx > 64) gx = 64;
        PLX_LAUNCH(k_absmax, dim3(gx, (unsigned)nframes), dim3(256), 4 * sizeof(double), st, a, P->d_max);
    }
    DecArgs g;
    g.ux = (const cplx *)d_ux; g.uy = (const cplx *)d_uy; g.out = (cplx *)d_out; g.dmax = P->d_max;
    g.n = N; g.nout = P->nout; g.r = P->d.decim; g.ntaps = (int)P->fir.size(); g.gd = (g.ntaps - 1) / 2;
    g.adcbits = P->d.adcbits;
    g.shift[0] = shift ? shift[0] : 0;
    g.shift[1] = shift ? shift[dual ? 1 : 0] : 0;
    for (int k = 0; k < 64; k++) g.fir[k] = k < g.ntaps ? P->fir[k] : 0.0;
    PLX_LAUNCH(k_decimate, dim3((unsigned)((P->nout + 255) / 256), (unsigned)nframes, (unsigned)(dual + 1)), dim3(256), 0, st, g);
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

// ---------------------------------------------------------------- generic spectral filter ---
// y = ifft(fft(x) .* H) on [nsig][nfft] complex128 signals (every row filtered by the same H): the DSP-side
// dispersion-compensating FIR of RxPdmCohQpsk.m:74-84 / dsp4cohdec.m:163-173 (H from DispCompFilter, host) and any
// other fixed frequency response a caller wants applied on the device.
struct plx_filter {
    plx_ssfm *fft = nullptr;
    cplx *d_h = nullptr;
    int max_sig = 0;
};

extern "C" int plx_filter_create(plx_filter **out, int64_t nfft, int max_signals, const double *h_re, const double *h_im)
{
    if (!out || !h_re) PLX_FAIL(PLX_ERR_ARG, "plx_filter_create: null argument");
    *out = nullptr;
    if (max_signals < 1) PLX_FAIL(PLX_ERR_ARG, "plx_filter_create: max_signals must be >= 1");
    plx_filter *P = new plx_filter();
    P->max_sig = max_signals;
    std::vector<double> zeros((size_t)(nfft > 0 ? nfft : 1), 0.0);
    double gam0 = 0.0;
    plx_ssfm_desc sd;
    std::memset(&sd, 0, sizeof(sd));
    sd.nfft = nfft; sd.nfc = 1; sd.dual_pol = 0; sd.max_frames = max_signals;
    sd.dzmaxt = 1; sd.dphimaxt = 1; sd.length = 1; sd.nplates = 1; sd.gam = &gam0; sd.betat = zeros.data();
    int rc = plx_ssfm_create(&P->fft, &sd);
    if (rc == PLX_OK) rc = plx_ssfm_filter_table(P->fft, h_re, h_im, &P->d_h);
    if (rc != PLX_OK) {
        if (P->fft) plx_ssfm_destroy(P->fft);
        delete P;
        return rc;
    }
    *out = P;
    return PLX_OK;
}

extern "C" int plx_filter_destroy(plx_filter *P)
{
    if (!P) return PLX_OK;
    if (P->fft) plx_ssfm_destroy(P->fft);
    if (P->d_h) (void)hipFree(P->d_h);
    delete P;
    return PLX_OK;
}

extern "C" int plx_filter_apply_dev(plx_filter *P, double *d_x, int nsignals, void *stream)
{
    if (!P || !d_x) PLX_FAIL(PLX_ERR_ARG, "plx_filter_apply_dev: null argument");
    if (nsignals < 1 || nsignals > P->max_sig) PLX_FAIL(PLX_ERR_ARG, "plx_filter_apply_dev: nsignals outside [1, max_signals]");
    return plx_ssfm_filter_dev(P->fft, (cplx *)d_x, nullptr, P->d_h, nsignals, stream);
}

// gateway tier: one RxPdmCohQpsk front end on host arrays with MATLAB's separate planes (one frame)
extern "C" int plx_rx_front(const double *xr, const double *xi, const double *yr, const double *yi, const plx_front_desc *desc,
                            const int64_t *shift, double *outr, double *outi, double *cur_r, double *cur_i)
{
    if (!xr || !desc || !outr || !outi) PLX_FAIL(PLX_ERR_ARG, "plx_rx_front: null argument");
    plx_front_desc d = *desc;
    d.max_frames = 1;
    const int dual = d.dual_pol ? 1 : 0;
    if (dual && !yr) PLX_FAIL(PLX_ERR_ARG, "plx_rx_front: dual-polarisation descriptor needs the y field");
    // plan (content hash of the descriptor and its tables), device buffers, pinned staging: the library's (plx_gateway.h)
    std::lock_guard<std::mutex> lk(plxgw::mutex());
    plxgw::count_call();
    int rc = PLX_OK;
    plx_front *P = plxgw::front_plan(d, &rc);
    if (!P) return rc;
    const size_t N = (size_t)d.nfft, nout = (size_t)plx_front_out_len(P);
    const size_t nh = N * (dual + 1), no = nout * (dual + 1);
    cplx *h = (cplx *)plxgw::pinned(plxgw::S_IN, nh * sizeof(cplx)), *o = (cplx *)plxgw::pinned(plxgw::S_OUT, no * sizeof(cplx));
    cplx *d_u = (cplx *)plxgw::dev(plxgw::S_IN, nh * sizeof(cplx)), *d_o = (cplx *)plxgw::dev(plxgw::S_OUT, no * sizeof(cplx));
    if (!h || !o || !d_u || !d_o) return PLX_ERR_HIP;
    for (size_t i = 0; i < N; i++) {
        h[i] = make_double2(xr[i], xi ? xi[i] : 0.0);
        if (dual) h[N + i] = make_double2(yr[i], yi ? yi[i] : 0.0);
    }
    PLX_HIP(hipMemcpyAsync(d_u, h, nh * sizeof(cplx), hipMemcpyHostToDevice, nullptr));
    rc = plx_front_run_dev(P, (double *)d_u, dual ? (double *)(d_u + N) : nullptr, 1, shift, (double *)d_o, nullptr);
    if (rc != PLX_OK) return rc;
    if (hipMemcpyAsync(o, d_o, no * sizeof(cplx), hipMemcpyDeviceToHost, nullptr) != hipSuccess ||
        hipMemcpyAsync(h, d_u, nh * sizeof(cplx), hipMemcpyDeviceToHost, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)
        PLX_FAIL(PLX_ERR_HIP, "plx_rx_front: HIP allocation or transfer failed");
    for (size_t i = 0; i < no; i++) { outr[i] = o[i].x; outi[i] = o[i].y; }          // [nout x (1 + dual)] column-major
    if (cur_r && cur_i)                                                                      // photocurrents [nfft x 2(1 + dual)]: I, Q per pol
        for (int p = 0; p <= dual; p++)
            for (size_t i = 0; i < N; i++) {
                cur_r[(2 * p) * N + i] = h[p * N + i].x; cur_i[(2 * p) * N + i] = 0.0;
                cur_r[(2 * p + 1) * N + i] = h[p * N + i].y; cur_i[(2 * p + 1) * N + i] = 0.0;
            }
    return PLX_OK;
}

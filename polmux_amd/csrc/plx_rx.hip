// plx_rx.hip -- receiver side of the hot path on gfx950:
//   CDE_OFDE.m overlap-save CD equaliser, the CMA / EASI 2x2 butterfly of
//   cmaadaptivefilter.c / easiadaptivefilter.c with its driver loops
//   (DspPdmCohQpsk.m:142-244), the DspPdmCohQpsk.m body (decimate, NLRotation,
//   normalise, carrier recovery :12-79, vitvit :97-123) and samp2pat decisions.
//
// MI355X mapping:
//   * overlap-save: one workgroup = 8 blocks of one signal; the 256-point block
//     transform, the filter taps H (bit-reversed, pre-scaled by 1/N) and the twiddles
//     all live in LDS; each input sample is read twice (50 % overlap), each output
//     written once.
//   * CMA: a strictly serial recurrence in the sample index, so parallelism comes
//     from (frames) x (taps): a group of 8/16/32 lanes owns one frame, lane = tap
//     index, each lane keeps its four complex taps in registers; the four dot
//     products are reduced across the group with wave shuffles, the Godard error is
//     formed redundantly by every lane, and the tap update is lane-local.  The pass
//     loop and its 5e-5 convergence test run on the device.
//   * carrier recovery: one workgroup per (frame, polarisation); the circular boxcar
//     of vitvit is evaluated as the direct circular moving sum it is (the reference
//     goes through fft/ifft of length L; same linear operator), cumsum/unwrap are
//     block scans.
#include "../../include/polmux_hip.h"
#include "plx_fft.h"
#include "plx_gateway.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

// ===================================================================== CDE ======
struct CdeArgs {
    const cplx *x;
    cplx *y;
    const cplx *Hp;  // [N] H at bit-reversed bins, fftshift undone, times 1/N
    const cplx *tw;  // half table W_N^k
    int64_t nx;
    int logN, L, B2, G, logG, nblocks, h_in_lds;
};

// OverlapBothTrans, CDE_OFDE.m:88-124
__global__ __launch_bounds__(256) void k_cde(CdeArgs a)
{
    PLX_DYN_LDS(lds);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int N = 1 << a.logN;
    cplx *s = (cplx *)lds;            // [G][N]
    cplx *tw = s + (size_t)a.G * N;   // [N/2]
    cplx *Hl = tw + (N >> 1);         // [N] when staged
    lds_load_twiddles(tw, a.tw, N >> 1, tid, nthr);
    if (a.h_in_lds)
        for (int k = tid; k < N; k += nthr) Hl[k] = a.Hp[k];
    const cplx *H = a.h_in_lds ? Hl : a.Hp;
    const size_t sig = blockIdx.y;
    const cplx *x = a.x + sig * (size_t)a.nx;
    cplx *y = a.y + sig * (size_t)a.nx;
    const int b0 = blockIdx.x * a.G;
    const int nel = a.G << a.logN;
    for (int e = tid; e < nel; e += nthr) {
        const int g = e >> a.logN, n = e & (N - 1);
        const int64_t src = (int64_t)(b0 + g) * a.L - a.B2 + n; // zero extension :92-102
        cplx v = make_double2(0.0, 0.0);
        if (b0 + g < a.nblocks && src >= 0 && src < a.nx) v = x[src];
        s[e] = v;
    }
    __syncthreads();
    lds_fft_dif(s, a.logN, 1, N, a.logG, tw, tid, nthr, false);
    for (int e = tid; e < nel; e += nthr) s[e] = cmul(s[e], H[e & (N - 1)]); // Yc = Xc.*H :110
    __syncthreads();
    lds_fft_dit(s, a.logN, 1, N, a.logG, tw, tid, nthr, false);
    const int nout = a.G * a.L;
    for (int e = tid; e < nout; e += nthr) {
        const int g = e / a.L, j = e - g * a.L;
        const int64_t dst = (int64_t)(b0 + g) * a.L + j;
        if (b0 + g < a.nblocks && dst < a.nx) y[dst] = s[(g << a.logN) + a.B2 + j]; // :115,119
    }
}

// ============================================================ CMA / EASI ======
struct DemuxArgs {
    const cplx *x;   // [frame][2][L]
    cplx *y;         // [frame][2][L]
    const cplx *M;   // [frame][4] or [4] (m_stride 0)
    cplx *h;         // optional [frame][2][2*taps]: h1(:,1) h1(:,2) | h2(:,1) h2(:,2)
    int *passes;     // optional
    int64_t L;
    int nframes, taps, halftaps, G, logG, m_stride, max_passes, single_pass, dontskip, skipk;
    double mu, R1, R2;
};

template <class T> __device__ __forceinline__ T group_sum(T v, int G)
{
    for (int m = 1; m < G; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double group_max(double v, int G)
{
    for (int m = 1; m < G; m <<= 1) {
        double o = __shfl_xor(v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}

// cmafilter (cmaadaptivefilter.c:57-91) inside the pass loop of cmapolardemux
// (DspPdmCohQpsk.m:161-191).  Group of G lanes per frame, lane t owns tap t.
__global__ __launch_bounds__(256) void k_cma(DemuxArgs a)
{
    const int lane_in_grp = threadIdx.x & (a.G - 1);
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> a.logG;
    const bool frame_ok = grp < a.nframes;
    const int f = frame_ok ? grp : a.nframes - 1; // idle groups shadow the last frame, never store
    const int t = lane_in_grp;
    const bool tap_ok = t < a.taps;
    const int64_t L = a.L;
    const cplx *x1 = a.x + (size_t)f * 2 * L, *x2 = x1 + L;
    cplx *y1 = a.y + (size_t)f * 2 * L, *y2 = y1 + L;
    // taps: h11 = h1(t,1), h12 = h1(t,2), h21 = h2(t,1), h22 = h2(t,2)
    cplx h11 = make_double2(0, 0), h12 = h11, h21 = h11, h22 = h11;
    if (a.single_pass) {
        if (tap_ok) { // taps handed in by the caller: [2][2*taps]
            const cplx *hh = a.h + (size_t)f * 4 * a.taps;
            h11 = hh[t]; h12 = hh[a.taps + t]; h21 = hh[2 * a.taps + t]; h22 = hh[3 * a.taps + t];
        }
    } else if (t == a.halftaps) { // hzero(halftaps+1,:,:) = M  :160
        const cplx *M = a.M + (size_t)f * a.m_stride;
        h11 = M[0]; h12 = M[1]; h21 = M[2]; h22 = M[3];
    }
    const int64_t nout = a.single_pass ? L - a.taps + 1 : L; // gateway: dimY = Mdim-Ntap+1
    int c = 1, npass = 0;
    // Groups of one wave may need different pass counts; the wave keeps looping until its
    // slowest group is done and finished groups run predicated (no stores, no updates).
    bool active = c < a.max_passes; // while ~convergence && (c < repetitions)  :176
    while (__any(active)) {
        const cplx o11 = h11, o12 = h12, o21 = h21, o22 = h22;
        for (int64_t i = 0; i < nout; i++) {
            int64_t idx = a.single_pass ? i + t : i + t - a.halftaps; // cyclic extension :161-165
            if (!a.single_pass) { if (idx < 0) idx += L; else if (idx >= L) idx -= L; }
            cplx xa = make_double2(0, 0), xb = xa;
            if (tap_ok) { xa = x1[idx]; xb = x2[idx]; }
            // y_r = sum_t x1[i+t] h_r1[t] + x2[i+t] h_r2[t]   (cmaadaptivefilter.c:71-81)
            cplx p1 = cadd(cmul(xa, h11), cmul(xb, h12));
            cplx p2 = cadd(cmul(xa, h21), cmul(xb, h22));
            const double y1r = group_sum(p1.x, a.G), y1i = group_sum(p1.y, a.G);
            const double y2r = group_sum(p2.x, a.G), y2i = group_sum(p2.y, a.G);
            if (active && frame_ok && t == 0) { y1[i] = make_double2(y1r, y1i); y2[i] = make_double2(y2r, y2i); }
            if (active && (a.dontskip || ((int)(i & 1) == a.skipk))) { // :85
                // updatecoeff :41-54: h += k*y*conj(x), k = mu*(R-|y|^2)
                const double k1 = a.mu * (a.R1 - y1r * y1r - y1i * y1i);
                const double k2 = a.mu * (a.R2 - y2r * y2r - y2i * y2i);
                h11.x += k1 * (y1r * xa.x + y1i * xa.y); h11.y += k1 * (y1i * xa.x - y1r * xa.y);
                h12.x += k1 * (y1r * xb.x + y1i * xb.y); h12.y += k1 * (y1i * xb.x - y1r * xb.y);
                h21.x += k2 * (y2r * xa.x + y2i * xa.y); h21.y += k2 * (y2i * xa.x - y2r * xa.y);
                h22.x += k2 * (y2r * xb.x + y2i * xb.y); h22.y += k2 * (y2i * xb.x - y2r * xb.y);
            }
        }
        // max(max(abs([h1_old-h1 h2_old-h2]))) < 5e-5  :187
        double d = hypot(o11.x - h11.x, o11.y - h11.y);
        double e = hypot(o12.x - h12.x, o12.y - h12.y);
        d = e > d ? e : d;
        e = hypot(o21.x - h21.x, o21.y - h21.y);
        d = e > d ? e : d;
        e = hypot(o22.x - h22.x, o22.y - h22.y);
        d = e > d ? e : d;
        d = group_max(d, a.G);
        if (active) {
            npass++;
            c++;
            if (a.single_pass || d < 5e-5 || !(c < a.max_passes)) active = false;
        }
    }
    if (frame_ok && tap_ok && a.h) {
        cplx *hh = a.h + (size_t)f * 4 * a.taps;
        hh[t] = h11; hh[a.taps + t] = h12; hh[2 * a.taps + t] = h21; hh[3 * a.taps + t] = h22;
    }
    if (frame_ok && t == 0 && a.passes) a.passes[f] = npass;
}

__device__ __forceinline__ double sum8(double v)
{
    v += lane_xchg<1>(v);
    v += lane_xchg<2>(v);
    v += lane_xchg<7>(v);
    return v;
}
// the same reduction for two values at once, level by level: the two dependent chains (move, move, add) interleave,
// which is what hides their latency on the one wave per SIMD this kernel runs with
__device__ __forceinline__ void sum8x2(double &a, double &b)
{
    double a1 = lane_xchg<1>(a), b1 = lane_xchg<1>(b);
    a += a1; b += b1;
    a1 = lane_xchg<2>(a); b1 = lane_xchg<2>(b);
    a += a1; b += b1;
    a1 = lane_xchg<7>(a); b1 = lane_xchg<7>(b);
    a += a1; b += b1;
}
__device__ __forceinline__ double max16(double v)
{
    double o = lane_xchg<1>(v); v = o > v ? o : v;
    o = lane_xchg<2>(v); v = o > v ? o : v;
    o = lane_xchg<7>(v); v = o > v ? o : v;
    o = lane_xchg<15>(v); v = o > v ? o : v;
    return v;
}

// Fast path of the cmapolardemux driver for taps <= 8 (the reference's 7): 16 lanes per frame,
// lane = (output row r, tap t); each lane keeps h_r(t,1), h_r(t,2) in registers, the dot
// products are DPP-reduced over the 8 tap lanes, the input samples of the next chunk are
// prefetched while the current chunk runs (the recurrence itself is latency-bound).
#define CMA_U 8
// Workgroup = 4 waves = 16 frames.  The waves are independent (no workgroup barrier); they travel together so that a long
// Monte-Carlo demultiplexing pass running BESIDE the fibre of the next batch sits on a quarter of the CUs, one wave on every
// SIMD of each, instead of taking one SIMD's registers on every CU of the chip: a 210-VGPR wave leaves room for one
// 254-VGPR wave of the fused column sweep on its SIMD, so a CU with a CMA wave anywhere holds ONE column workgroup, not two.
#ifdef PLX_EMU
#define CMA16_THREADS 64      // (the host emulator runs one OS thread per lane: one wave per workgroup keeps its tests short)
#else
#define CMA16_THREADS 256
#endif
__global__ __launch_bounds__(CMA16_THREADS) void k_cma16(DemuxArgs a)
{
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int l16 = threadIdx.x & 15, r = l16 >> 3, t = l16 & 7;
    const int grp = gtid >> 4;
    const bool frame_ok = grp < a.nframes;
    const int f = frame_ok ? grp : a.nframes - 1;
    const bool tap_ok = t < a.taps;
    const int64_t L = a.L;
    const cplx *x1 = a.x + (size_t)f * 2 * L, *x2 = x1 + L;
    cplx *yo = a.y + (size_t)f * 2 * L + (size_t)r * L;
    cplx ha = make_double2(0, 0), hb = ha; // h_r(t,1), h_r(t,2)
    // single_pass: ONE call of the MEX gateway (cmaadaptivefilter.c:57-91 as called by the unchanged driver): the taps come
    // from the caller, output i uses samples i .. i+taps-1 (no cyclic extension), Mdim - Ntap + 1 outputs, one pass
    const bool single = a.single_pass != 0;
    if (single) {
        if (tap_ok) {
            const cplx *hh = a.h + (size_t)f * 4 * a.taps + (size_t)r * 2 * a.taps;
            ha = hh[t]; hb = hh[a.taps + t];
        }
    } else if (t == a.halftaps) {
        const cplx *M = a.M + (size_t)f * a.m_stride;
        ha = M[2 * r]; hb = M[2 * r + 1];
    }
    const double Rr = r ? a.R2 : a.R1, mu = a.mu;
    const int Li = (int)L;                                  // (32-bit indices: L < 2^31, checked by the host)
    const int off = single ? t : t - a.halftaps;
    const int nout = (int)(single ? L - a.taps + 1 : L);
    int c = 1, npass = 0;
    bool active = c < a.max_passes;
    const int nchunks = nout / CMA_U, tail0 = nchunks * CMA_U;
    // The wave is ISSUE-bound: in order, ~5 clocks per instruction whatever their dependencies (profiles/r04_notes.md: the
    // recurrence with its loop-carried dependency cut runs no faster), so what counts is the instruction count per symbol --
    // 28 FP64 operations + 12 DPP moves are the recurrence itself; everything around them is kept off the symbol loop:
    //  * the samples of the NEXT chunk are loaded unconditionally by every lane (a lane beyond the last tap keeps zero taps and
    //    a zero step size instead of a predicate per load: eight exec-masked blocks of 28 instructions per chunk before), from
    //    a position kept modulo L by one compare per sample instead of 64-bit wrap arithmetic;
    //  * the samples of an INTERIOR chunk k (0 < k < nchunks - 2: no lane's window leaves [0, L)) sit at one per-lane pointer
    //    plus a wave-uniform offset and eight immediates; only the chunks at the two ends of a pass wrap around;
    //  * two chunks per trip, the second computing from the registers the first one's prefetch landed in (no register copies).
    const double mul_lane = tap_ok ? mu : 0.0;
    const cplx *const q1 = x1 + off, *const q2 = x2 + off;  // (only dereferenced where 0 <= i + off < L)
    auto load8 = [&](int k, cplx *da, cplx *db) {           // the samples of chunk k: symbols 8 k .. 8 k + 7
        const int i0 = k * CMA_U;
        if (k > 0 && k < nchunks - 2) {
#pragma unroll
            for (int u = 0; u < CMA_U; u++) { da[u] = q1[i0 + u]; db[u] = q2[i0 + u]; }
        } else {                                            // cyclic extension (:161-165): positions modulo L
            int base = i0 + off;
            base = base < 0 ? base + Li : (base >= Li ? base - Li : base);
#pragma unroll
            for (int u = 0; u < CMA_U; u++) {
                int idx = base + u;
                idx = idx >= Li ? idx - Li : idx;
                da[u] = x1[idx]; db[u] = x2[idx];
            }
        }
    };
    while (__any(active)) {
        const cplx oa = ha, ob = hb;
        const double mua = active ? mul_lane : 0.0;
        // one chunk of CMA_U symbols from the samples in (sa, sb); its outputs go to yo[i0 ...]
        auto chunk = [&](const cplx *sa, const cplx *sb, int i0) {
            cplx yk[CMA_U];                // the chunk's outputs: stored once, outside the dependent chain
#pragma unroll
            for (int u = 0; u < CMA_U; u++) {
                const cplx xa = sa[u], xb = sb[u];
                double yr = (xa.x * ha.x - xa.y * ha.y) + (xb.x * hb.x - xb.y * hb.y);
                double yi = (xa.x * ha.y + xa.y * ha.x) + (xb.x * hb.y + xb.y * hb.x);
                sum8x2(yr, yi);
                yk[u] = make_double2(yr, yi);
                // (a finished frame keeps iterating with mu = 0: its taps stay as they are, no branch in the chain)
                const double k = mua * (Rr - yr * yr - yi * yi);
                const double kr = k * yr, ki = k * yi;
                // h += k y conj(x) as two dependent fused multiply-adds per component (the shortest chain back to the next
                // symbol's products)
                ha.x = fma(ki, xa.y, fma(kr, xa.x, ha.x)); ha.y = fma(-kr, xa.y, fma(ki, xa.x, ha.y));
                hb.x = fma(ki, xb.y, fma(kr, xb.x, hb.x)); hb.y = fma(-kr, xb.y, fma(ki, xb.x, hb.y));
            }
            if (active && frame_ok && t == 0) {
#pragma unroll
                for (int u = 0; u < CMA_U; u++) yo[i0 + u] = yk[u];
            }
        };
        cplx ca[CMA_U], cb[CMA_U], na[CMA_U], nb[CMA_U];
        if (nchunks > 0) load8(0, ca, cb);
        int ch = 0;
        for (; ch + 1 < nchunks; ch += 2) {
            load8(ch + 1, na, nb);                           // chunk ch+1 lands while chunk ch computes
            chunk(ca, cb, ch * CMA_U);
            if (ch + 2 < nchunks) load8(ch + 2, ca, cb);     // chunk ch+2 lands while chunk ch+1 computes
            chunk(na, nb, (ch + 1) * CMA_U);
        }
        if (ch < nchunks) chunk(ca, cb, ch * CMA_U);         // (an odd number of chunks: the last one is in ca, cb)
        for (int i = tail0; i < nout; i++) { // the outputs beyond the last whole chunk
            int idx = i + off;
            if (idx < 0) idx += Li; else if (idx >= Li) idx -= Li;
            const cplx xa = x1[idx], xb = x2[idx];
            double yr = (xa.x * ha.x - xa.y * ha.y) + (xb.x * hb.x - xb.y * hb.y);
            double yi = (xa.x * ha.y + xa.y * ha.x) + (xb.x * hb.y + xb.y * hb.x);
            sum8x2(yr, yi);
            if (active && frame_ok && t == 0) yo[i] = make_double2(yr, yi);
            const double k = mua * (Rr - yr * yr - yi * yi);
            const double kr = k * yr, ki = k * yi;
            ha.x = fma(ki, xa.y, fma(kr, xa.x, ha.x)); ha.y = fma(-kr, xa.y, fma(ki, xa.x, ha.y));
            hb.x = fma(ki, xb.y, fma(kr, xb.x, hb.x)); hb.y = fma(-kr, xb.y, fma(ki, xb.x, hb.y));
        }
        double d = hypot(oa.x - ha.x, oa.y - ha.y);
        const double e = hypot(ob.x - hb.x, ob.y - hb.y);
        d = max16(e > d ? e : d);
        if (active) {
            npass++;
            c++;
            if (single || d < 5e-5 || !(c < a.max_passes)) active = false;
        }
    }
    if (frame_ok && tap_ok && a.h) { // [h1(:,1) h1(:,2) | h2(:,1) h2(:,2)]
        cplx *hh = a.h + (size_t)f * 4 * a.taps + (size_t)r * 2 * a.taps;
        hh[t] = ha; hh[a.taps + t] = hb;
    }
    if (frame_ok && l16 == 0 && a.passes) a.passes[f] = npass;
}

// easifilter (easiadaptivefilter.c:52-93) inside easipolardemux (DspPdmCohQpsk.m:195-244).
// taps == 1 in the driver (:197); the gateway form allows any odd/even taps: the update only
// ever touches the real parts of tap 0 (:83-90), the output uses all taps.  One lane per frame.
__global__ __launch_bounds__(64) void k_easi(DemuxArgs a)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.nframes) return;
    const int64_t L = a.L;
    const int taps = a.taps;
    const cplx *x1 = a.x + (size_t)f * 2 * L, *x2 = x1 + L;
    cplx *y1 = a.y + (size_t)f * 2 * L, *y2 = y1 + L;
    cplx *hh = a.h + (size_t)f * 4 * taps; // working taps live in global memory (tap 0 in registers)
    if (!a.single_pass) {
        const cplx *M = a.M + (size_t)f * a.m_stride;
        hh[0] = M[0]; hh[1] = M[1]; hh[2] = M[2]; hh[3] = M[3];
    }
    // the only mutable state: real parts of h1(1,1) h1(1,2)... note the reference indexes
    // h1r[0], h1r[1]: with taps==1 these are h1(1,1), h1(1,2); with taps>1 they are h1(1,1), h1(2,1).
    const int64_t nout = a.single_pass ? L - taps + 1 : L;
    int c = 1, npass = 0;
    bool conv = false;
    const double mu = a.mu;
    while (!conv && c < a.max_passes) {
        const double o0 = hh[0].x, o1 = hh[1].x, o2 = hh[2 * taps].x, o3 = hh[2 * taps + 1].x;
        double a0 = o0, a1 = o1, b0 = o2, b1 = o3; // h1r[0], h1r[1], h2r[0], h2r[1]
        for (int64_t i = 0; i < nout; i++) {
            double y1r = 0, y1i = 0, y2r = 0, y2i = 0;
            for (int p = 0; p < 2; p++) {
                const cplx *xp = p ? x2 : x1;
                double s1r = 0, s1i = 0, s2r = 0, s2i = 0;
                for (int t = 0; t < taps; t++) {
                    const cplx xv = xp[i + t];
                    cplx g1 = hh[p * taps + t], g2 = hh[2 * taps + p * taps + t];
                    const int flat = p * taps + t; // position inside h1r / h2r
                    if (flat == 0) { g1.x = a0; g2.x = b0; }
                    if (flat == 1) { g1.x = a1; g2.x = b1; }
                    s1r += xv.x * g1.x - xv.y * g1.y; s1i += xv.y * g1.x + xv.x * g1.y;
                    s2r += xv.x * g2.x - xv.y * g2.y; s2i += xv.y * g2.x + xv.x * g2.y;
                }
                y1r += s1r; y1i += s1i; y2r += s2r; y2i += s2i;
            }
            y1[i] = make_double2(y1r, y1i);
            y2[i] = make_double2(y2r, y2i);
            if (a.dontskip || ((int)(i & 1) == a.skipk)) {
                const double A = y1r, B = y2r; // only the real parts enter :81
                const double den1 = 1 + mu * (A * A + B * B);
                const double den2 = 1 + mu * (A * fabs(A) + B * fabs(B));
                const double E0 = (A * A - 1) / den1;
                const double E1 = (A * B) / den1 + (A * B * (A * A - B * B)) / den2;
                const double E2 = (A * B) / den1 + (A * B * (B * B - A * A)) / den2;
                const double E3 = (B * B - 1) / den1;
                const double n11 = (1 - mu * E0) * a0 + (-mu * E1) * b0;
                const double n12 = (1 - mu * E0) * a1 + (-mu * E1) * b1;
                const double n21 = (-mu * E2) * a0 + (1 - mu * E3) * b0;
                const double n22 = (-mu * E2) * a1 + (1 - mu * E3) * b1;
                a0 = n11; a1 = n12; b0 = n21; b1 = n22;
            }
        }
        hh[0].x = a0; hh[1].x = a1; hh[2 * taps].x = b0; hh[2 * taps + 1].x = b1;
        npass++;
        if (a.single_pass) break;
        double d = fabs(o0 - a0), e = fabs(o1 - a1);
        d = e > d ? e : d; e = fabs(o2 - b0); d = e > d ? e : d; e = fabs(o3 - b1); d = e > d ? e : d;
        if (d < 5e-5) conv = true;
        c++;
    }
    if (a.passes) a.passes[f] = npass;
}

// The .m twin of easifilter (easiadaptivefilter.m:51-84), what the drivers run when no MEX is compiled: the error
// matrix is formed from the COMPLEX outputs a, b (abs(), complex denominators) and ALL taps of the complex h1, h2 are
// recombined (:58-66) -- not the real-parts-of-tap-0 update of the C file.  One lane per frame, taps in global memory
// (hh [2][2*taps]: h1(:,1) h1(:,2) | h2(:,1) h2(:,2)); the driver's taps == 1 keeps them in registers in effect.
__device__ __forceinline__ cplx cdiv(cplx a, cplx b)
{
    const double d = b.x * b.x + b.y * b.y;
    return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}
__global__ __launch_bounds__(64) void k_easi_m(DemuxArgs a)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.nframes) return;
    const int64_t L = a.L;
    const int taps = a.taps;
    const cplx *x1 = a.x + (size_t)f * 2 * L, *x2 = x1 + L;
    cplx *y1 = a.y + (size_t)f * 2 * L, *y2 = y1 + L;
    cplx *hh = a.h + (size_t)f * 4 * taps;
    if (!a.single_pass) {
        const cplx *M = a.M + (size_t)f * a.m_stride;
        hh[0] = M[0]; hh[1] = M[1]; hh[2] = M[2]; hh[3] = M[3];
    }
    const int64_t nout = a.single_pass ? L - taps + 1 : L;
    const double mu = a.mu;
    int c = 1, npass = 0;
    bool conv = false;
    while (!conv && c < a.max_passes) {
        cplx o[4];
        if (taps == 1) { o[0] = hh[0]; o[1] = hh[1]; o[2] = hh[2]; o[3] = hh[3]; }
        for (int64_t i = 0; i < nout; i++) {
            cplx A = make_double2(0, 0), B = A;      // sum(sum(xx(nindex,:).*h)): column sums, then their sum (:53-54)
            for (int p = 0; p < 2; p++) {
                const cplx *xp = p ? x2 : x1;
                cplx s1 = make_double2(0, 0), s2 = s1;
                for (int t = 0; t < taps; t++) {
                    const cplx xv = xp[i + t];
                    s1 = cadd(s1, cmul(xv, hh[p * taps + t]));
                    s2 = cadd(s2, cmul(xv, hh[2 * taps + p * taps + t]));
                }
                A = cadd(A, s1); B = cadd(B, s2);
            }
            y1[i] = A; y2[i] = B;
            const double aa = hypot(A.x, A.y), ab = hypot(B.x, B.y);
            const double den1 = 1 + mu * (aa * aa + ab * ab);                                    // errorfun :78-84
            const cplx den2 = make_double2(1 + mu * (A.x * aa + B.x * ab), mu * (A.y * aa + B.y * ab));
            const cplx pr = cmul(A, B);
            const double E11 = (aa * aa - 1) / den1, E22 = (ab * ab - 1) / den1;
            const cplx q = make_double2(pr.x / den1, pr.y / den1);
            const cplx E12 = cadd(q, cdiv(cscale(pr, aa * aa - ab * ab), den2));
            const cplx E21 = cadd(q, cdiv(cscale(pr, ab * ab - aa * aa), den2));
            const cplx m12 = cscale(E12, -mu), m21 = cscale(E21, -mu);
            const double d1 = 1 - mu * E11, d2 = 1 - mu * E22;
            for (int t = 0; t < taps; t++)
                for (int col = 0; col < 2; col++) {                                              // :58-66
                    const cplx g1 = hh[col * taps + t], g2 = hh[2 * taps + col * taps + t];
                    hh[col * taps + t] = cadd(cscale(g1, d1), cmul(m12, g2));
                    hh[2 * taps + col * taps + t] = cadd(cmul(m21, g1), cscale(g2, d2));
                }
        }
        npass++;
        if (a.single_pass) break;
        double d = 0;                                                                            // :236 (taps == 1 in the driver)
        for (int k = 0; k < 4; k++) { const double e = hypot(o[k].x - hh[k].x, o[k].y - hh[k].y); d = e > d ? e : d; }
        if (d < 5e-5) conv = true;
        c++;
    }
    if (a.passes) a.passes[f] = npass;
}

// ================================================== DspPdmCohQpsk front part ======
struct PreArgs {
    const cplx *in; // [frame][ncol][Lin]
    cplx *out;      // [frame][ncol][L]
    int64_t Lin, L;
    int ncol, stride, applynlr;
    double nlralpha, inv_peak;
};

// :12-23: 1:2:end, NLRotation (:87-94), /peak.  One workgroup per frame.
__global__ __launch_bounds__(256) void k_dsp_pre(PreArgs a)
{
    PLX_DYN_LDS(lds);
    double *red = (double *)lds;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const size_t f = blockIdx.x;
    const cplx *in = a.in + f * a.ncol * (size_t)a.Lin;
    cplx *out = a.out + f * a.ncol * (size_t)a.L;
    double mean = 0;
    if (a.applynlr) {
        double acc = 0;
        for (int64_t i = tid; i < a.L; i += nthr) {
            double q = 0;
            for (int c = 0; c < a.ncol; c++) {
                const cplx v = in[(size_t)c * a.Lin + i * a.stride];
                const double m = hypot(v.x, v.y);
                q += m * m;
            }
            acc += q;
        }
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        for (int w = 0; w < (nthr + 63) / 64; w++) mean += red[w];
        mean /= (double)a.L;
    }
    for (int64_t i = tid; i < a.L; i += nthr) {
        double q = 0;
        if (a.applynlr)
            for (int c = 0; c < a.ncol; c++) {
                const cplx v = in[(size_t)c * a.Lin + i * a.stride];
                const double m = hypot(v.x, v.y);
                q += m * m;
            }
        for (int c = 0; c < a.ncol; c++) {
            cplx v = in[(size_t)c * a.Lin + i * a.stride];
            if (a.applynlr) {
                const double ph = atan2(v.y, v.x) + a.nlralpha * (q - mean);
                const double am = hypot(v.x, v.y);
                double sp, cp;
                sincos(ph, &sp, &cp);
                v = make_double2(am * cp, am * sp);
            }
            out[(size_t)c * a.L + i] = make_double2(v.x * a.inv_peak, v.y * a.inv_peak);
        }
    }
}

// rotpolar (:126-139) from the Kikuchi ratio r = mean(x1./x2): either applies y = x*M
// ('singlepol', :29-31) or only writes M.' as the initial centre taps (:151-153).
__global__ __launch_bounds__(256) void k_rotpolar(const cplx *x, cplx *y, cplx *Mout, int64_t L, int apply)
{
    PLX_DYN_LDS(lds);
    double *red = (double *)lds;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const size_t f = blockIdx.x;
    const cplx *x1 = x + f * 2 * (size_t)L, *x2 = x1 + L;
    double sr = 0, si = 0;
    for (int64_t i = tid; i < L; i += nthr) {
        const cplx a = x1[i], b = x2[i];
        const double den = b.x * b.x + b.y * b.y;
        sr += (a.x * b.x + a.y * b.y) / den;
        si += (a.y * b.x - a.x * b.y) / den;
    }
    for (int m = 32; m >= 1; m >>= 1) { sr += __shfl_xor(sr, m, 64); si += __shfl_xor(si, m, 64); }
    if ((tid & 63) == 0) { red[2 * (tid >> 6)] = sr; red[2 * (tid >> 6) + 1] = si; }
    __syncthreads();
    double rr = 0, ri = 0;
    for (int w = 0; w < (nthr + 63) / 64; w++) { rr += red[2 * w]; ri += red[2 * w + 1]; }
    rr /= (double)L; ri /= (double)L;
    double m, delta, alpha;
    const double ar = hypot(rr, ri);
    if (ar < 0.5) {
        m = ar; delta = atan2(ri, rr); alpha = m * m / (m * m + 1);
    } else {
        const double d2 = rr * rr + ri * ri; // 1/r
        const double ir = rr / d2, ii = -ri / d2;
        m = hypot(ir, ii); delta = -atan2(ii, ir); alpha = 1 / (m * m + 1);
    }
    double sd, cd;
    sincos(-delta, &sd, &cd);
    const cplx M11 = make_double2(sqrt(alpha) * cd, sqrt(alpha) * sd);
    const cplx M12 = make_double2(-sqrt(1 - alpha) * cd, -sqrt(1 - alpha) * sd);
    const cplx M21 = make_double2(sqrt(1 - alpha), 0), M22 = make_double2(sqrt(alpha), 0);
    if (apply) {
        cplx *y1 = y + f * 2 * (size_t)L, *y2 = y1 + L;
        for (int64_t i = tid; i < L; i += nthr) { // y = x*M
            const cplx a = x1[i], b = x2[i];
            y1[i] = cadd(cmul(a, M11), cmul(b, M21));
            y2[i] = cadd(cmul(a, M12), cmul(b, M22));
        }
    } else if (tid == 0) { // M = rotpolar(1,r).'
        cplx *Mo = Mout + f * 4;
        Mo[0] = M11; Mo[1] = M21; Mo[2] = M12; Mo[3] = M22;
    }
}

// ============================================================ carrier recovery ======
struct CpeArgs {
    const cplx *s;   // [frame][ncol][L] normalised, demultiplexed samples
    cplx *out;       // [frame][ncol][L]
    cplx *wsc;       // global scratch [frame*ncol][2][L] (used when LDS is too small)
    double *wsr;     // global scratch [frame*ncol][L]
    int64_t L;
    int use_lds, modorder, freqavg, phasavg, poworder;
};

__device__ __forceinline__ cplx cpow_int(cplx v, int P)
{
    cplx r = v;
    for (int i = 1; i < P; i++) r = cmul(r, v);
    return r;
}

// circular causal boxcar of 2k+1 taps (vitvit :106-117): dst[n] = mean_{m<N} src[(n-m) mod L]
__device__ __forceinline__ void boxcar(const cplx *src, cplx *dst, int64_t L, int k, int tid, int nthr)
{
    const int64_t N = 2 * (int64_t)k + 1;
    const double invN = 1.0 / (double)N;
    for (int64_t n = tid; n < L; n += nthr) {
        double ar = 0, ai = 0;
        int64_t j = n;
        for (int64_t m = 0; m < N; m++) {
            const cplx v = src[j];
            ar += v.x; ai += v.y;
            j = (j == 0) ? L - 1 : j - 1;
        }
        dst[n] = make_double2(ar * invN, ai * invN);
    }
}

// inclusive scan of w[0..L) in place (cumsum); red: >= nthr+1 doubles of LDS
__device__ __forceinline__ void block_cumsum(double *w, int64_t L, double *red, int tid, int nthr)
{
    const int64_t per = (L + nthr - 1) / nthr, lo = (int64_t)tid * per, hi = (lo + per < L) ? lo + per : L;
    double acc = 0;
    for (int64_t i = lo; i < hi; i++) { acc += w[i]; w[i] = acc; }
    red[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        double run = 0;
        for (int t = 0; t < nthr; t++) { const double v = red[t]; red[t] = run; run += v; }
    }
    __syncthreads();
    const double off = red[tid];
    for (int64_t i = lo; i < hi; i++) w[i] += off;
    __syncthreads();
}

// :44-79 for one column.  One workgroup per (frame, column).
__global__ __launch_bounds__(256) void k_cpe(CpeArgs a)
{
    PLX_DYN_LDS(lds);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int64_t L = a.L;
    const size_t col = blockIdx.x;
    const cplx *s = a.s + col * (size_t)L;
    cplx *out = a.out + col * (size_t)L;
    double *red = (double *)lds; // [nthr+16]
    cplx *bufA, *bufB;
    double *w, *th;
    if (a.use_lds) {
        bufA = (cplx *)(red + nthr + 16);
        bufB = bufA + L;
        w = (double *)(bufB + L);
        th = w + L;
    } else {
        bufA = a.wsc + col * 2 * (size_t)L;
        bufB = bufA + L;
        w = a.wsr + col * 2 * (size_t)L;
        th = w + L;
    }
    const int Mo = 1 << a.modorder;
    const double off = a.modorder > 1 ? 0.78539816339744830962 : 0.0; // +pi/4 :62-66
    if (a.freqavg) {
        // omega = cumsum(vitvit(s.*conj(fastshift(s,1)), M, M, freqavg, false))  :49-50
        for (int64_t n = tid; n < L; n += nthr) {
            const cplx d = cmulc(s[n], s[n == 0 ? L - 1 : n - 1]);
            bufA[n] = cpow_int(d, Mo); // P == M: s.^P
        }
        __syncthreads();
        boxcar(bufA, bufB, L, a.freqavg, tid, nthr);
        __syncthreads();
        for (int64_t n = tid; n < L; n += nthr) w[n] = atan2(bufB[n].y, bufB[n].x) / Mo;
        __syncthreads();
        block_cumsum(w, L, red, tid, nthr);
        // circularity fix :52-55
        const double w1 = w[0], wend = w[L - 1];
        const double closest = w1 + round((wend - w1) / 2 / 3.14159265358979323846) * 2 * 3.14159265358979323846;
        const double ratio = closest / wend;
        __syncthreads();
        for (int64_t n = tid; n < L; n += nthr) w[n] = ((w[n] - w1) * ratio) + w1;
        __syncthreads();
    } else {
        for (int64_t n = tid; n < L; n += nthr) w[n] = 0;
        __syncthreads();
    }
    // theta = vitvit(sigdemod, P, M, phasavg, true)  :57-61
    for (int64_t n = tid; n < L; n += nthr) {
        const cplx sd = a.freqavg ? cmul(s[n], cexpi(-w[n])) : s[n];
        if (a.poworder == Mo) {
            bufA[n] = cpow_int(sd, Mo);
        } else {
            const cplx sm = cpow_int(sd, Mo);
            const double am = pow(hypot(sd.x, sd.y), (double)a.poworder);
            const double ang = atan2(sm.y, sm.x);
            double sp, cp;
            sincos(ang, &sp, &cp);
            bufA[n] = make_double2(am * cp, am * sp);
        }
    }
    __syncthreads();
    const cplx *bs = bufA;
    if (a.phasavg > 0) {
        boxcar(bufA, bufB, L, a.phasavg, tid, nthr);
        __syncthreads();
        bs = bufB;
    }
    for (int64_t n = tid; n < L; n += nthr) th[n] = atan2(bs[n].y, bs[n].x);
    __syncthreads();
    // unwrap (tolerance pi): correction of sample n depends on th[n]-th[n-1] only -> scan
    double *corr = (double *)bufA; // bufA is free again
    const double PI = 3.14159265358979323846;
    for (int64_t n = tid; n < L; n += nthr) {
        double cv = 0;
        if (n > 0) {
            const double dp = th[n] - th[n - 1];
            double dps = fmod(dp + PI, 2 * PI);
            if (dps < 0) dps += 2 * PI;
            dps -= PI;
            if (dps == -PI && dp > 0) dps = PI;
            cv = (fabs(dp) < PI) ? 0.0 : dps - dp;
        }
        corr[n] = cv;
    }
    __syncthreads();
    block_cumsum(corr, L, red, tid, nthr);
    for (int64_t n = tid; n < L; n += nthr) {
        const double theta = (th[n] + corr[n]) / Mo;
        out[n] = cmul(s[n], cexpi(-w[n] - theta + off)); // :67,79
    }
}

// ====================================================== decisions + error count ======
// samp2pat 'coherent' (samp2pat.m:61-66) and err = sum(sum(pat ~= pat_hat)) (ber_estimate.m:119),
// kept per column so that a caller can resolve the pi/2 ambiguity of each polarisation.
// pstride: bytes between the pattern blocks of consecutive frames (0: one pattern shared by all frames)
__global__ __launch_bounds__(256) void k_decide(const cplx *sym, int64_t L, int ncol, const uint8_t *pat, size_t pstride,
                                                uint8_t *pat_hat, unsigned long long *err)
{
    PLX_DYN_LDS(lds);
    unsigned long long *red = (unsigned long long *)lds;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const size_t fc = blockIdx.x, f = fc / ncol;
    const int c = (int)(fc - f * ncol);
    unsigned long long cnt = 0;
    if (pat) pat += f * pstride;
    for (int64_t i = tid; i < L; i += nthr) {
        const cplx v = sym[fc * (size_t)L + i];
        const double ph = atan2(v.y, v.x);
        const uint8_t first = fabs(ph) <= 1.57079632679489661923 ? 1 : 0;
        const uint8_t second = ph > 0 ? 1 : 0;
        if (pat_hat) {
            pat_hat[(f * 2 * ncol + 2 * c) * (size_t)L + i] = first;
            pat_hat[(f * 2 * ncol + 2 * c + 1) * (size_t)L + i] = second;
        }
        if (pat) {
            cnt += (pat[(size_t)(2 * c) * L + i] != first);
            cnt += (pat[(size_t)(2 * c + 1) * L + i] != second);
        }
    }
    for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
    if ((tid & 63) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0 && err) {
        unsigned long long tot = 0;
        for (int w = 0; w < (nthr + 63) / 64; w++) tot += red[w];
        err[fc] = tot;
    }
}

// Per-frame error-vector magnitude of the recovered symbols: mean over the frame's symbols (all columns) of
// |s - s_hat|^2, s_hat the unit-modulus QPSK point of the quadrant samp2pat decides (samp2pat.m:61-66).  A continuous
// per-realisation sample for mc_estimate (mc_estimate.m:133-212) beside the integer error count of ber_estimate.
__global__ __launch_bounds__(256) void k_evm(const cplx *sym, int64_t L, int ncol, double *evm)
{
    PLX_DYN_LDS(lds);
    double *red = (double *)lds;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const size_t f = blockIdx.x;
    const int64_t n = L * ncol;
    double acc = 0;
    for (int64_t i = tid; i < n; i += nthr) {
        const cplx v = sym[f * (size_t)n + i];
        const double hx = v.x >= 0 ? 0.70710678118654752440 : -0.70710678118654752440;
        const double hy = v.y > 0 ? 0.70710678118654752440 : -0.70710678118654752440;
        const double dx = v.x - hx, dy = v.y - hy;
        acc += dx * dx + dy * dy;
    }
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double tot = 0;
        for (int w = 0; w < (nthr + 63) / 64; w++) tot += red[w];
        evm[f] = tot / (double)n;
    }
}

int ilog2i(int64_t v)
{
    int l = 0;
    while (((int64_t)1 << l) < v) l++;
    return l;
}

#ifndef PLX_EMU
template <class K> hipError_t allow_lds(K kern, size_t bytes)
{
    return hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
#else
template <class K> hipError_t allow_lds(K, size_t) { return hipSuccess; }
#endif

} // namespace

// =================================================================== CDE host ===
struct plx_cde {
    int64_t N, L;
    int logN, G, h_in_lds;
    cplx *d_H = nullptr, *d_tw = nullptr;
    size_t lds = 0;
};

static const char *kCdeMsg[] = {"", "Error: x must be one dimensional complex vector", "Error: H must be even length",
                                "Error: L must be > 0", "Error: L must be shorter than filter length",
                                "Error: Signal must be longer or equal filter"};

extern "C" int plx_cde_create(plx_cde **out, int64_t fft_len, int64_t L, const double *H)
{
    if (!out || !H) PLX_FAIL(PLX_ERR_ARG, "plx_cde_create: null argument");
    *out = nullptr;
    if (fft_len % 2 != 0) PLX_FAIL(PLX_ERR_ARG, kCdeMsg[2]);     // CDE_OFDE.m:73-74
    if (L <= 0) PLX_FAIL(PLX_ERR_ARG, kCdeMsg[3]);               // :77-78
    if (L > fft_len) PLX_FAIL(PLX_ERR_ARG, kCdeMsg[4]);          // :79-80
    const int logN = ilog2i(fft_len);
    if (((int64_t)1 << logN) != fft_len || fft_len < 4 || fft_len > 4096)
        PLX_FAIL(PLX_ERR_UNSUPPORTED, "plx_cde_create: FFT length must be a power of two in [4, 4096]");
    if ((fft_len - L) % 2 != 0) PLX_FAIL(PLX_ERR_UNSUPPORTED, "plx_cde_create: overlap N-L must be even");
    plx_cde *P = new plx_cde();
    P->N = fft_len; P->L = L; P->logN = logN;
    int G = 1;
    while (G < 8 && (int64_t)G * 2 * fft_len <= 2048) G *= 2;
    P->G = G;
    P->h_in_lds = fft_len <= 1024 ? 1 : 0;
    // H is given on the fftshift-ordered grid: unshifted bin k takes H[(k+N/2) mod N]
    // (CDE_OFDE.m:108-112); LDS position i holds bin bitrev(i); fold in ifft's 1/N.
    std::vector<cplx> Hp((size_t)fft_len), tw((size_t)(fft_len / 2));
    for (int64_t i = 0; i < fft_len; i++) {
        const int64_t k = (int64_t)plx_bitrev((unsigned)i, logN);
        const int64_t src = (k + fft_len / 2) % fft_len;
        Hp[i] = make_double2(H[2 * src] / (double)fft_len, H[2 * src + 1] / (double)fft_len);
    }
    for (int64_t k = 0; k < fft_len / 2; k++) {
        long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)fft_len;
        tw[k] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    if (hipMalloc((void **)&P->d_H, Hp.size() * sizeof(cplx)) != hipSuccess ||
        hipMalloc((void **)&P->d_tw, tw.size() * sizeof(cplx)) != hipSuccess ||
        hipMemcpy(P->d_H, Hp.data(), Hp.size() * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(P->d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess) {
        plx_cde_destroy(P);
        PLX_FAIL(PLX_ERR_HIP, "plx_cde_create: device allocation/upload failed");
    }
    P->lds = ((size_t)G * fft_len + fft_len / 2 + (P->h_in_lds ? fft_len : 0)) * sizeof(cplx);
    if (allow_lds(k_cde, P->lds) != hipSuccess) { plx_cde_destroy(P); PLX_FAIL(PLX_ERR_HIP, "plx_cde_create: cannot reserve LDS"); }
    *out = P;
    return PLX_OK;
}

extern "C" int plx_cde_destroy(plx_cde *P)
{
    if (P) { hipFree(P->d_H); hipFree(P->d_tw); delete P; }
    return PLX_OK;
}

extern "C" int plx_cde_apply_dev(plx_cde *P, const double *d_x, double *d_y, int64_t nx, int nsig, void *stream)
{
    if (!P || !d_x || !d_y) PLX_FAIL(PLX_ERR_ARG, "plx_cde_apply_dev: null argument");
    if (nx < P->N) PLX_FAIL(PLX_ERR_ARG, kCdeMsg[5]); // CDE_OFDE.m:83-84
    if (nsig < 1) PLX_FAIL(PLX_ERR_ARG, "plx_cde_apply_dev: nsig must be >= 1");
    CdeArgs a;
    a.x = (const cplx *)d_x; a.y = (cplx *)d_y; a.Hp = P->d_H; a.tw = P->d_tw; a.nx = nx;
    a.logN = P->logN; a.L = (int)P->L; a.B2 = (int)((P->N - P->L) / 2); a.G = P->G; a.logG = ilog2i(P->G);
    a.nblocks = (int)((nx + P->L - 1) / P->L); a.h_in_lds = P->h_in_lds;
    const unsigned gx = (unsigned)((a.nblocks + a.G - 1) / a.G);
    PLX_LAUNCH(k_cde, dim3(gx, (unsigned)nsig), dim3(256), P->lds, stream, a);
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_cde_ofde(const double *xr, const double *xi, const double *yr, const double *yi, int64_t nx,
                            double fs, double lambda_ref, double span, double D, double S, int64_t fft_len,
                            int64_t L, double *oxr, double *oxi, double *oyr, double *oyi)
{
    if (!xr || !yr || !oxr || !oxi || !oyr || !oyi) PLX_FAIL(PLX_ERR_ARG, "plx_cde_ofde: null argument");
    if (fft_len > nx) fft_len = nx; // CDE_OFDE.m:24-27
    // transfer function, CDE_OFDE.m:29-38 (host, double)
    const double c = 299792458.0, fc = c / lambda_ref, df = 1.0 / ((double)fft_len / fs);
    std::vector<double> H(2 * (size_t)fft_len);
    for (int64_t i = 0; i < fft_len; i++) {
        const double fg = df * (double)(i - fft_len / 2);
        const double hd = -(D * span * M_PI * c / (fc * fc) * (fg * fg));
        const double hs = S * span * M_PI * (c * c) / 3 / (fc * fc * fc * fc) * (fg * fg * fg);
        H[2 * i] = cos(hd + hs);
        H[2 * i + 1] = sin(hd + hs);
    }
    if (nx < fft_len) PLX_FAIL(PLX_ERR_ARG, kCdeMsg[5]);
    // plan (keyed by fft_len, L and the transfer function), device buffers and pinned staging: the library's (plx_gateway.h)
    std::lock_guard<std::mutex> lk(plxgw::mutex());
    plxgw::count_call();
    int rc = PLX_OK;
    plx_cde *P = plxgw::cde_plan(fft_len, L, H.data(), &rc);
    if (!P) return rc;
    const size_t bytes = 4 * (size_t)nx * sizeof(double);
    double *h = (double *)plxgw::pinned(plxgw::S_IN, bytes);
    double *dx = (double *)plxgw::dev(plxgw::S_IN, bytes), *dy = (double *)plxgw::dev(plxgw::S_OUT, bytes);
    if (!h || !dx || !dy) return PLX_ERR_HIP;
    for (int64_t i = 0; i < nx; i++) {
        h[2 * i] = xr[i]; h[2 * i + 1] = xi ? xi[i] : 0.0;
        h[2 * (nx + i)] = yr[i]; h[2 * (nx + i) + 1] = yi ? yi[i] : 0.0;
    }
    PLX_HIP(hipMemcpyAsync(dx, h, bytes, hipMemcpyHostToDevice, nullptr));
    rc = plx_cde_apply_dev(P, dx, dy, nx, 2, nullptr);
    if (rc) return rc;
    if (hipMemcpyAsync(h, dy, bytes, hipMemcpyDeviceToHost, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)
        PLX_FAIL(PLX_ERR_HIP, "plx_cde_ofde: download failed");
    for (int64_t i = 0; i < nx; i++) {
        oxr[i] = h[2 * i]; oxi[i] = h[2 * i + 1];
        oyr[i] = h[2 * (nx + i)]; oyi[i] = h[2 * (nx + i) + 1];
    }
    return PLX_OK;
}

// ================================================================ pol-demux host ===
// batches of this many frames and more run k_cma16 in four-wave workgroups (fewer CUs carry a CMA wave: see k_cma16)
static const int kCmaPackMin = 64;    // (measured with four Monte-Carlo rounds of 128 in flight: +2...8 %, profiles/r03_mc.txt)
static int launch_demux(int method, DemuxArgs &a, void *stream)
{
    if (method == PLX_DEMUX_CMA && a.taps <= 8 && a.dontskip && a.L >= 16 && a.L < ((int64_t)1 << 30) && (!a.single_pass || a.L - a.taps + 1 >= 1)) {
        // big batches travel four waves to a workgroup (see k_cma16); a few frames keep a CU per wave (sharing one costs the
        // recurrence ~14 %: 16 frames of 2^20 samples, 148 -> 170 ms)
        const int thr = a.nframes >= kCmaPackMin ? CMA16_THREADS : 64, fpw = thr / 16;
        const unsigned gx = (unsigned)((a.nframes + fpw - 1) / fpw);
        PLX_LAUNCH(k_cma16, dim3(gx), dim3(thr), 0, stream, a);
    } else if (method == PLX_DEMUX_CMA) {
        int G = 8;
        while (G < a.taps) G *= 2;
        if (G > 64) PLX_FAIL(PLX_ERR_UNSUPPORTED, "pol-demux: at most 64 taps are supported");
        a.G = G; a.logG = ilog2i(G);
        const int frames_per_block = 256 / G;
        const unsigned gx = (unsigned)((a.nframes + frames_per_block - 1) / frames_per_block);
        PLX_LAUNCH(k_cma, dim3(gx), dim3(256), 0, stream, a);
    } else if (method == PLX_DEMUX_EASI_M) {
        // the driver form (pass loop + convergence test on the four taps) exists for taps == 1 only, as easipolardemux
        // fixes it (DspPdmCohQpsk.m:197); the single-pass gateway form takes any number of taps
        if (!a.single_pass && a.taps != 1) PLX_FAIL(PLX_ERR_ARG, "pol-demux: the EASI driver loop runs with one tap (DspPdmCohQpsk.m:197)");
        const unsigned gx = (unsigned)((a.nframes + 63) / 64);
        PLX_LAUNCH(k_easi_m, dim3(gx), dim3(64), 0, stream, a);
    } else {
        const unsigned gx = (unsigned)((a.nframes + 63) / 64);
        PLX_LAUNCH(k_easi, dim3(gx), dim3(64), 0, stream, a);
    }
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_poldemux_dev(int method, const double *d_x, double *d_y, int64_t L, int nframes, int32_t taps,
                                double mu, const double *R, const double *d_M, double *d_h, int32_t *d_passes,
                                void *stream)
{
    if (!d_x || !d_y || !d_M) PLX_FAIL(PLX_ERR_ARG, "plx_poldemux_dev: null argument");
    if (method != PLX_DEMUX_CMA && method != PLX_DEMUX_EASI && method != PLX_DEMUX_EASI_M)
        PLX_FAIL(PLX_ERR_ARG, "plx_poldemux_dev: unknown method");
    if (L < 1 || nframes < 1 || taps < 1 || !(mu > 0)) PLX_FAIL(PLX_ERR_ARG, "plx_poldemux_dev: bad size");
    DemuxArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = (const cplx *)d_x; a.y = (cplx *)d_y; a.M = (const cplx *)d_M; a.h = (cplx *)d_h; a.passes = d_passes;
    a.L = L; a.nframes = nframes; a.m_stride = 4; a.mu = mu; a.dontskip = 1; // drivers pass sps = 1 (:179,:231)
    double *tmp_h = nullptr;
    if (method == PLX_DEMUX_CMA) {
        if (!R) PLX_FAIL(PLX_ERR_ARG, "plx_poldemux_dev: CMA needs R");
        if (taps % 2 == 0) PLX_FAIL(PLX_ERR_ARG, "Ntaps should be an ODD INTEGER."); // cmaadaptivefilter.c:118-119
        if (taps / 2 >= L) PLX_FAIL(PLX_ERR_ARG, "plx_poldemux_dev: more taps than samples");
        a.taps = taps; a.halftaps = taps / 2; a.R1 = R[0]; a.R2 = R[1];
        a.max_passes = (int)(50.0 * ceil(1.0 / ((double)L * mu))); // :175
    } else {
        a.taps = 1; a.halftaps = 0; // easipolardemux fixes taps = 1 (:197)
        a.max_passes = (int)(20.0 * ceil(1.0 / ((double)L * mu))); // :227
        if (!a.h) { // the EASI kernel keeps its taps in global memory
            PLX_HIP(hipMalloc((void **)&tmp_h, sizeof(cplx) * 4 * (size_t)nframes));
            a.h = (cplx *)tmp_h;
        }
    }
    int rc = launch_demux(method, a, stream);
    if (tmp_h) { hipStreamSynchronize((hipStream_t)stream); hipFree(tmp_h); }
    return rc;
}

// gateway forms: one call of the MEX function on host arrays -----------------------
static int gateway_filter(int method, const double *xr, const double *xi, int32_t Mdim, double *h1r, double *h1i,
                          double *h2r, double *h2i, double Ntap, double mu, const double *R, double sps, double *yr,
                          double *yi, bool twin = false)
{
    if (!xr || !h1r || !h2r || !yr || !yi) PLX_FAIL(PLX_ERR_ARG, "adaptive filter gateway: null argument");
    const int taps = (int)Ntap;
    // the .m twins have neither check (cmaadaptivefilter.m:52-54 sizes everything from h1; sps is unused)
    if (!twin && method == PLX_DEMUX_CMA && taps % 2 == 0) PLX_FAIL(PLX_ERR_ARG, "Ntaps should be an ODD INTEGER.");
    if (!twin && (int)sps != 1 && (int)sps != 2) PLX_FAIL(PLX_ERR_ARG, "Samples x symbol should be either 1 or 2.");
    if (twin && (!h1i || !h2i)) PLX_FAIL(PLX_ERR_ARG, "adaptive filter gateway: the .m twins return complex taps (h1i, h2i required)");
    if (taps < 1 || Mdim < taps) PLX_FAIL(PLX_ERR_ARG, "adaptive filter gateway: input shorter than the filter");
    if (method == PLX_DEMUX_CMA && taps > 64) PLX_FAIL(PLX_ERR_UNSUPPORTED, "pol-demux: at most 64 taps are supported");
    const int dimY = Mdim - taps + 1;
    // One call = ONE pass of the filter; the unchanged drivers make up to 299 of them per frame (DspPdmCohQpsk.m:176-191).
    // Device buffers and the pinned staging area are the library's (plx_gateway.h): a pass allocates nothing, moves
    // x | h up in one copy and h | y down in one copy.
    std::lock_guard<std::mutex> lk(plxgw::mutex());
    plxgw::count_call();
    const size_t nx = 4 * (size_t)Mdim, nh = 8 * (size_t)taps;          // doubles: x [2][Mdim] complex, h [4][taps] complex
    double *hst = (double *)plxgw::pinned(plxgw::S_IN, (2 * nx + nh) * sizeof(double));
    double *dbuf = (double *)plxgw::dev(plxgw::S_IN, (2 * nx + nh) * sizeof(double));
    if (!hst || !dbuf) return PLX_ERR_HIP;
    double *hx = hst, *hh = hst + nx, *hy = hst + nx + nh;                 // staging: x | h | y
    double *dx = dbuf, *dh = dbuf + nx, *dy = dbuf + nx + nh;
    for (int p = 0; p < 2; p++)
        for (int i = 0; i < Mdim; i++) {
            hx[2 * ((size_t)p * Mdim + i)] = xr[(size_t)p * Mdim + i];
            hx[2 * ((size_t)p * Mdim + i) + 1] = xi ? xi[(size_t)p * Mdim + i] : 0.0;
        }
    for (int j = 0; j < 2 * taps; j++) {
        hh[2 * j] = h1r[j]; hh[2 * j + 1] = h1i ? h1i[j] : 0.0;
        hh[2 * (2 * taps + j)] = h2r[j]; hh[2 * (2 * taps + j) + 1] = h2i ? h2i[j] : 0.0;
    }
    PLX_HIP(hipMemcpyAsync(dx, hx, (nx + nh) * sizeof(double), hipMemcpyHostToDevice, nullptr));
    PLX_HIP(hipMemsetAsync(dy, 0, nx * sizeof(double), nullptr));
    DemuxArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = (const cplx *)dx; a.y = (cplx *)dy; a.h = (cplx *)dh; a.L = Mdim; a.nframes = 1; a.taps = taps;
    a.halftaps = 0; a.mu = mu; a.single_pass = 1; a.max_passes = 2;
    a.dontskip = (twin || (int)sps == 1) ? 1 : 0;   // the .m twin updates at every sample (cmaadaptivefilter.m:60-69)
    a.skipk = ((taps - 1) / 2) % 2; // cmaadaptivefilter.c:64
    if (R) { a.R1 = R[0]; a.R2 = R[1]; }
    int rc = launch_demux(method, a, nullptr);
    if (rc) return rc;
    PLX_HIP(hipMemcpyAsync(hh, dh, (nh + nx) * sizeof(double), hipMemcpyDeviceToHost, nullptr));   // h | y are adjacent
    PLX_HIP(hipStreamSynchronize(nullptr));
    // y is [dimY x 2]; the kernel wrote column p at offset p*Mdim
    for (int p = 0; p < 2; p++)
        for (int i = 0; i < dimY; i++) {
            yr[(size_t)p * dimY + i] = hy[2 * ((size_t)p * Mdim + i)];
            yi[(size_t)p * dimY + i] = hy[2 * ((size_t)p * Mdim + i) + 1];
        }
    // in-place tap update, as the reference MEX does through prhs[1..2] (:87-88)
    for (int j = 0; j < 2 * taps; j++) {
        h1r[j] = hh[2 * j]; if (h1i) h1i[j] = hh[2 * j + 1];
        h2r[j] = hh[2 * (2 * taps + j)]; if (h2i) h2i[j] = hh[2 * (2 * taps + j) + 1];
    }
    return PLX_OK;
}

extern "C" int plx_cmaadaptivefilter(const double *xr, const double *xi, int32_t Mdim, double *h1r, double *h1i,
                                     double *h2r, double *h2i, double Ntap, double mu, const double *R, double sps,
                                     double *yr, double *yi)
{
    if (!R) PLX_FAIL(PLX_ERR_ARG, "plx_cmaadaptivefilter: R is required");
    return gateway_filter(PLX_DEMUX_CMA, xr, xi, Mdim, h1r, h1i, h2r, h2i, Ntap, mu, R, sps, yr, yi);
}

extern "C" int plx_easiadaptivefilter(const double *xr, const double *xi, int32_t Mdim, double *h1r, double *h1i,
                                      double *h2r, double *h2i, double Ntap, double mu, double sps, double *yr,
                                      double *yi)
{
    return gateway_filter(PLX_DEMUX_EASI, xr, xi, Mdim, h1r, h1i, h2r, h2i, Ntap, mu, nullptr, sps, yr, yi);
}

extern "C" int plx_cmaadaptivefilter_m(const double *xr, const double *xi, int32_t Mdim, double *h1r, double *h1i,
                                       double *h2r, double *h2i, int32_t ntap, double mu, const double *R, double *yr,
                                       double *yi)
{
    if (!R) PLX_FAIL(PLX_ERR_ARG, "plx_cmaadaptivefilter_m: R is required");
    return gateway_filter(PLX_DEMUX_CMA, xr, xi, Mdim, h1r, h1i, h2r, h2i, (double)ntap, mu, R, 1.0, yr, yi, true);
}

extern "C" int plx_easiadaptivefilter_m(const double *xr, const double *xi, int32_t Mdim, double *h1r, double *h1i,
                                        double *h2r, double *h2i, int32_t ntap, double mu, double *yr, double *yi)
{
    return gateway_filter(PLX_DEMUX_EASI_M, xr, xi, Mdim, h1r, h1i, h2r, h2i, (double)ntap, mu, nullptr, 1.0, yr, yi, true);
}

// The whole DRIVER loop as one gateway call: y = cmapolardemux(x, params) / easipolardemux(x, params)
// (DspPdmCohQpsk.m:142-192, :195-244 == dsp4cohdec.m:374-478).  The unchanged drivers call the filter MEX once per pass, up
// to 299 times per frame, each a PCIe round trip (0.22 ms per pass against 0.08 ms on one host core); here the cyclic
// extension (:161-165), the centre-tap initialisation from M (:160), the pass loop with its budget (:175 / :227) and the
// 5e-5 test (:187) all run on the device (k_cma16 / k_cma / k_easi / k_easi_m) and x crosses the bus once.
static int gateway_poldemux(int method, const double *xr, const double *xi, int64_t L, int32_t taps, double mu, const double *R,
                            const double *M, double *yr, double *yi, double *h1r, double *h1i, double *h2r, double *h2i,
                            int32_t *passes)
{
    if (!xr || !M || !yr || !yi) PLX_FAIL(PLX_ERR_ARG, "pol-demux gateway: null argument");
    if (L < 1 || L > ((int64_t)1 << 28)) PLX_FAIL(PLX_ERR_ARG, "pol-demux gateway: bad length");
    if (method != PLX_DEMUX_CMA) taps = 1;                                  // easipolardemux fixes taps = 1 (:197)
    if (taps < 1 || taps > 64) PLX_FAIL(PLX_ERR_UNSUPPORTED, "pol-demux: at most 64 taps are supported");
    std::lock_guard<std::mutex> lk(plxgw::mutex());
    plxgw::count_call();
    const size_t nx = 4 * (size_t)L, nh = 8 * (size_t)taps;                 // doubles: x / y [2][L] complex, h [4][taps] complex
    // staging: x | M (up, one copy) and y | h | passes (down, one copy)
    double *hst = (double *)plxgw::pinned(plxgw::S_IN, (2 * nx + 8 + nh + 2) * sizeof(double));
    double *dbuf = (double *)plxgw::dev(plxgw::S_IN, (2 * nx + 8 + nh + 2) * sizeof(double));
    if (!hst || !dbuf) return PLX_ERR_HIP;
    double *hx = hst, *hM = hst + nx, *hy = hst + nx + 8, *hh = hy + nx, *hp = hh + nh;
    double *dx = dbuf, *dM = dbuf + nx, *dy = dbuf + nx + 8, *dh = dy + nx, *dp = dh + nh;
    for (int p = 0; p < 2; p++)
        for (int64_t i = 0; i < L; i++) {
            hx[2 * ((size_t)p * L + i)] = xr[(size_t)p * L + i];
            hx[2 * ((size_t)p * L + i) + 1] = xi ? xi[(size_t)p * L + i] : 0.0;
        }
    for (int k = 0; k < 8; k++) hM[k] = M[k];
    PLX_HIP(hipMemcpyAsync(dx, hx, (nx + 8) * sizeof(double), hipMemcpyHostToDevice, nullptr));
    PLX_HIP(hipMemsetAsync(dy, 0, (nx + nh + 2) * sizeof(double), nullptr));
    int rc = plx_poldemux_dev(method, dx, dy, L, 1, taps, mu, R, dM, dh, (int32_t *)dp, nullptr);
    if (rc) return rc;
    PLX_HIP(hipMemcpyAsync(hy, dy, (nx + nh + 2) * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    PLX_HIP(hipStreamSynchronize(nullptr));
    for (size_t i = 0; i < 2 * (size_t)L; i++) { yr[i] = hy[2 * i]; yi[i] = hy[2 * i + 1]; }
    for (int j = 0; j < 2 * taps; j++) {                                    // [h1(:,1) h1(:,2) | h2(:,1) h2(:,2)]
        if (h1r) h1r[j] = hh[2 * j];
        if (h1i) h1i[j] = hh[2 * j + 1];
        if (h2r) h2r[j] = hh[2 * (2 * taps + j)];
        if (h2i) h2i[j] = hh[2 * (2 * taps + j) + 1];
    }
    if (passes) std::memcpy(passes, hp, sizeof(int32_t));
    return PLX_OK;
}

extern "C" int plx_cmapolardemux(const double *xr, const double *xi, int64_t L, int32_t taps, double mu, const double *R,
                                 const double *M, double *yr, double *yi, double *h1r, double *h1i, double *h2r, double *h2i,
                                 int32_t *passes)
{
    if (!R) PLX_FAIL(PLX_ERR_ARG, "plx_cmapolardemux: R is required");
    return gateway_poldemux(PLX_DEMUX_CMA, xr, xi, L, taps, mu, R, M, yr, yi, h1r, h1i, h2r, h2i, passes);
}

extern "C" int plx_easipolardemux(const double *xr, const double *xi, int64_t L, double mu, const double *M, int32_t mfile_twin,
                                  double *yr, double *yi, double *h1r, double *h1i, double *h2r, double *h2i, int32_t *passes)
{
    return gateway_poldemux(mfile_twin ? PLX_DEMUX_EASI_M : PLX_DEMUX_EASI, xr, xi, L, 1, mu, nullptr, M, yr, yi, h1r, h1i, h2r,
                            h2i, passes);
}

// ===================================================================== DSP host ===
struct plx_dsp {
    plx_dsp_params p;
    int64_t Lin, L;
    int ncol, max_frames;
    cplx *d_a = nullptr, *d_b = nullptr, *d_M = nullptr, *d_h = nullptr, *d_wsc = nullptr;
    cplx *d_Mrot = nullptr; // [2][max_frames][4]: constant rotation matrices of the CMA / EASI drivers (:155-156)
    double *d_wsr = nullptr;
    int use_lds = 0;
    size_t lds_cpe = 0;
};

extern "C" int plx_dsp_destroy(plx_dsp *P)
{
    if (P) {
        hipFree(P->d_a); hipFree(P->d_b); hipFree(P->d_M); hipFree(P->d_h); hipFree(P->d_wsc); hipFree(P->d_wsr);
        hipFree(P->d_Mrot);
        delete P;
    }
    return PLX_OK;
}

extern "C" int plx_dsp_create(plx_dsp **out, int64_t Lin, int32_t ncol, int32_t max_frames, const plx_dsp_params *p)
{
    if (!out || !p) PLX_FAIL(PLX_ERR_ARG, "plx_dsp_create: null argument");
    *out = nullptr;
    if (Lin < 2 || (ncol != 1 && ncol != 2) || max_frames < 1) PLX_FAIL(PLX_ERR_ARG, "plx_dsp_create: bad size");
    if (p->applypol && ncol == 2 && (p->polmethod < 0 || p->polmethod > 3))
        PLX_FAIL(PLX_ERR_ARG, "Unknown Polar Rotation method."); // DspPdmCohQpsk.m:40
    if (p->modorder < 1 || p->modorder > 4 || p->poworder < 0 || p->freqavg < 0 || p->phasavg < 0)
        PLX_FAIL(PLX_ERR_ARG, "plx_dsp_create: bad carrier-recovery parameters");
    plx_dsp *P = new plx_dsp();
    P->p = *p; P->Lin = Lin; P->ncol = ncol; P->max_frames = max_frames;
    P->L = p->workatbaudrate ? Lin : (Lin + 1) / 2;
    const size_t n = (size_t)max_frames * ncol * P->L;
    const size_t lds = (256 + 16) * sizeof(double) + (size_t)P->L * (2 * sizeof(cplx) + 2 * sizeof(double));
    P->use_lds = lds <= 64 * 1024 ? 1 : 0;
    P->lds_cpe = P->use_lds ? lds : (256 + 16) * sizeof(double);
    bool ok = hipMalloc((void **)&P->d_a, n * sizeof(cplx)) == hipSuccess &&
              hipMalloc((void **)&P->d_b, n * sizeof(cplx)) == hipSuccess &&
              hipMalloc((void **)&P->d_M, (size_t)max_frames * 4 * sizeof(cplx)) == hipSuccess &&
              hipMalloc((void **)&P->d_h, (size_t)max_frames * 4 * 64 * sizeof(cplx)) == hipSuccess;
    if (ok && !P->use_lds)
        ok = hipMalloc((void **)&P->d_wsc, 2 * n * sizeof(cplx)) == hipSuccess &&
             hipMalloc((void **)&P->d_wsr, 2 * n * sizeof(double)) == hipSuccess;
    if (ok) { // M = [cos sin; -sin cos] of cmapolardemux / easipolardemux with two Tx polarisations
        std::vector<cplx> M((size_t)2 * max_frames * 4);
        for (int m = 0; m < 2; m++) {
            const double phi = m == 0 ? p->cma_phizero : p->easi_phizero;
            const bool has = m == 0 ? p->cma_has_mat != 0 : p->easi_has_mat != 0;
            const double *mat = m == 0 ? p->cma_mat : p->easi_mat;
            for (int f = 0; f < max_frames; f++) {
                cplx *q = &M[((size_t)m * max_frames + f) * 4];
                if (has) { // M = params.mat  (DspPdmCohQpsk.m:148-149, :201-202)
                    for (int k = 0; k < 4; k++) q[k] = make_double2(mat[2 * k], mat[2 * k + 1]);
                } else {
                    q[0] = make_double2(cos(phi), 0); q[1] = make_double2(sin(phi), 0);
                    q[2] = make_double2(-sin(phi), 0); q[3] = make_double2(cos(phi), 0);
                }
            }
        }
        ok = hipMalloc((void **)&P->d_Mrot, M.size() * sizeof(cplx)) == hipSuccess &&
             hipMemcpy(P->d_Mrot, M.data(), M.size() * sizeof(cplx), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { plx_dsp_destroy(P); PLX_FAIL(PLX_ERR_HIP, "plx_dsp_create: device allocation failed"); }
    if (allow_lds(k_cpe, P->lds_cpe) != hipSuccess) { plx_dsp_destroy(P); PLX_FAIL(PLX_ERR_HIP, "plx_dsp_create: cannot reserve LDS"); }
    *out = P;
    return PLX_OK;
}

extern "C" int64_t plx_dsp_out_len(const plx_dsp *P) { return P ? P->L : -1; }

static int demux_stage(plx_dsp *P, int method, cplx *src, cplx *dst, int nframes, void *stream)
{
    const plx_dsp_params &p = P->p;
    const int txpol = method == PLX_DEMUX_CMA ? p.cma_txpolars : p.easi_txpolars;
    const bool has_mat = method == PLX_DEMUX_CMA ? p.cma_has_mat != 0 : p.easi_has_mat != 0;
    const cplx *Min = P->d_M;
    if (txpol == 2 || has_mat) { // M = params.mat, or [cos sin; -sin cos]  :148-156 (constant, uploaded once at plan creation)
        Min = P->d_Mrot + (size_t)(method == PLX_DEMUX_CMA ? 0 : 1) * P->max_frames * 4;
    } else {
        PLX_LAUNCH(k_rotpolar, dim3((unsigned)nframes), dim3(256), 16 * sizeof(double), stream, (const cplx *)src,
                   (cplx *)nullptr, P->d_M, P->L, 0);
    }
    const int kmethod = (method == PLX_DEMUX_EASI && p.mfile_twins) ? PLX_DEMUX_EASI_M : method;   // no MEX compiled: the .m twin
    return plx_poldemux_dev(kmethod, (const double *)src, (double *)dst, P->L, nframes,
                            method == PLX_DEMUX_CMA ? p.cma_taps : 1, method == PLX_DEMUX_CMA ? p.cma_mu : p.easi_mu,
                            p.cma_R, (const double *)Min, (double *)P->d_h, nullptr, stream);
}

extern "C" int plx_dsp_run_dev(plx_dsp *P, const double *d_in, double *d_out, int nframes, void *stream)
{
    if (!P || !d_in || !d_out) PLX_FAIL(PLX_ERR_ARG, "plx_dsp_run_dev: null argument");
    if (nframes < 1 || nframes > P->max_frames) PLX_FAIL(PLX_ERR_ARG, "plx_dsp_run_dev: nframes out of range");
    const plx_dsp_params &p = P->p;
    PreArgs pa;
    pa.in = (const cplx *)d_in; pa.out = P->d_a; pa.Lin = P->Lin; pa.L = P->L; pa.ncol = P->ncol;
    pa.stride = p.workatbaudrate ? 1 : 2; pa.applynlr = p.applynlr; pa.nlralpha = p.nlralpha;
    pa.inv_peak = 1.0 / (4 * sqrt(p.power_mw)); // :22-23 (multiplication by the reciprocal)
    PLX_LAUNCH(k_dsp_pre, dim3((unsigned)nframes), dim3(256), 16 * sizeof(double), stream, pa);
    cplx *cur = P->d_a, *oth = P->d_b;
    if (p.applypol && P->ncol == 2) { // :26-42
        int rc = PLX_OK;
        switch (p.polmethod) {
        case 0:
            PLX_LAUNCH(k_rotpolar, dim3((unsigned)nframes), dim3(256), 16 * sizeof(double), stream, (const cplx *)cur,
                       oth, (cplx *)nullptr, P->L, 1);
            std::swap(cur, oth);
            break;
        case 1:
            rc = demux_stage(P, PLX_DEMUX_CMA, cur, oth, nframes, stream);
            std::swap(cur, oth);
            break;
        case 2:
            rc = demux_stage(P, PLX_DEMUX_EASI, cur, oth, nframes, stream);
            std::swap(cur, oth);
            break;
        case 3:
            rc = demux_stage(P, PLX_DEMUX_EASI, cur, oth, nframes, stream);
            if (!rc) rc = demux_stage(P, PLX_DEMUX_CMA, oth, cur, nframes, stream);
            break;
        default:
            PLX_FAIL(PLX_ERR_ARG, "Unknown Polar Rotation method.");
        }
        if (rc) return rc;
    }
    CpeArgs ca;
    ca.s = cur; ca.out = (cplx *)d_out; ca.wsc = P->d_wsc; ca.wsr = P->d_wsr; ca.L = P->L; ca.use_lds = P->use_lds;
    ca.modorder = p.modorder; ca.freqavg = p.freqavg; ca.phasavg = p.phasavg; ca.poworder = p.poworder;
    PLX_LAUNCH(k_cpe, dim3((unsigned)(nframes * P->ncol)), dim3(256), P->lds_cpe, stream, ca);
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_decide_count_frames_dev(const double *d_sym, int64_t L, int32_t ncol, int nframes, const uint8_t *d_pat,
                                           int64_t pat_frame_stride, uint8_t *d_pat_hat, int64_t *d_err, void *stream)
{
    if (!d_sym || L < 1 || ncol < 1 || nframes < 1 || pat_frame_stride < 0) PLX_FAIL(PLX_ERR_ARG, "plx_decide_count_dev: bad argument");
    PLX_LAUNCH(k_decide, dim3((unsigned)(nframes * ncol)), dim3(256), 16 * sizeof(unsigned long long), stream,
               (const cplx *)d_sym, L, (int)ncol, d_pat, (size_t)pat_frame_stride, d_pat_hat, (unsigned long long *)d_err);
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_evm_dev(const double *d_sym, int64_t L, int32_t ncol, int nframes, double *d_evm, void *stream)
{
    if (!d_sym || !d_evm || L < 1 || ncol < 1 || nframes < 1) PLX_FAIL(PLX_ERR_ARG, "plx_evm_dev: bad argument");
    PLX_LAUNCH(k_evm, dim3((unsigned)nframes), dim3(256), 16 * sizeof(double), stream, (const cplx *)d_sym, L, (int)ncol, d_evm);
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_decide_count_dev(const double *d_sym, int64_t L, int32_t ncol, int nframes, const uint8_t *d_pat,
                                    uint8_t *d_pat_hat, int64_t *d_err, void *stream)
{
    return plx_decide_count_frames_dev(d_sym, L, ncol, nframes, d_pat, 0, d_pat_hat, d_err, stream);
}

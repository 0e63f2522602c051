// plx_fft.h -- in-LDS complex128 FFT building blocks for gfx950.
//
// All transforms of the hot path (fiber.m:904-905,934-935 fft/ifft of the field,
// CDE_OFDE.m:108-112 block transforms, DspPdmCohQpsk.m:109-110 boxcar) are built
// from two workgroup-cooperative routines that work in place on LDS:
//
//   lds_fft_dif : forward DFT, natural order in  -> BIT-REVERSED order out
//   lds_fft_dit : inverse DFT (unscaled), BIT-REVERSED order in -> natural order out
//
// Pairing a DIF forward with a DIT inverse means no bit-reversal pass is ever
// executed: spectral multipliers are simply stored pre-permuted.  Radix-4
// butterflies (one radix-2 stage when log2 M is odd); twiddles are read from an
// LDS-staged half table W_M^k, k < M/2.
//
// Addressing: point i of transform t lives at s[i*IS + t*TS] (cplx units).
// Butterflies are dealt to threads either transform-fastest (column tiles, where
// the T transforms are contiguous in LDS => conflict-free ds_read_b128) or
// point-fastest (row transforms).
#pragma once
#include "plx_common.h"

__device__ __forceinline__ cplx tw3(const cplx *tw, int k3, int halfM)
{ // W^{k3}, k3 < 3M/4, from the half table: W^{k+M/2} = -W^k
    const cplx w = tw[k3 & (halfM - 1)];      // (halfM is a power of two, k3 < 3 halfM / 2: one load, no divergent branch)
    const bool neg = k3 >= halfM;
    return make_double2(neg ? -w.x : w.x, neg ? -w.y : w.y);
}

// T = 1<<logT transforms of length M = 1<<logM.  Caller has synchronised the data
// into LDS; on return all threads have passed a barrier (results visible).
__device__ __forceinline__ void lds_fft_dif(cplx *s, int logM, int IS, int TS, int logT, const cplx *tw,
                                            int tid, int nthr, bool tfast)
{
    const int M = 1 << logM, halfM = M >> 1, T = 1 << logT;
    int lm = logM;
    if (logM & 1) { // radix-2 head stage
        const int total = T * halfM;
        for (int b = tid; b < total; b += nthr) {
            int t, j;
            if (tfast) { t = b & (T - 1); j = b >> logT; } else { j = b & (halfM - 1); t = b >> (logM - 1); }
            cplx *p = s + t * TS;
            cplx a = p[j * IS], c = p[(j + halfM) * IS];
            p[j * IS] = cadd(a, c);
            p[(j + halfM) * IS] = cmul(csub(a, c), tw[j]);
        }
        __syncthreads();
        lm--;
    }
    for (; lm >= 2; lm -= 2) {
        const int lq = lm - 2, q = 1 << lq, sh = logM - lm;
        const int total = T << (logM - 2);
        for (int b = tid; b < total; b += nthr) {
            int t, bi;
            if (tfast) { t = b & (T - 1); bi = b >> logT; } else { bi = b & ((M >> 2) - 1); t = b >> (logM - 2); }
            const int j = bi & (q - 1);
            const int base = ((bi >> lq) << lm) + j;
            cplx *p = s + t * TS + base * IS;
            const int qs = q * IS;
            cplx a0 = p[0], a1 = p[qs], a2 = p[2 * qs], a3 = p[3 * qs];
            cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
            const int k = j << sh;
            cplx y0 = cadd(t0, t2);
            cplx y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
            if (k) { // bit-reversed placement: [y0, y2, y1, y3]
                y1 = cmul(y1, tw[k]);
                y2 = cmul(y2, tw[2 * k]);
                y3 = cmul(y3, tw3(tw, 3 * k, halfM));
            }
            p[0] = y0; p[qs] = y2; p[2 * qs] = y1; p[3 * qs] = y3;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void lds_fft_dit(cplx *s, int logM, int IS, int TS, int logT, const cplx *tw,
                                            int tid, int nthr, bool tfast)
{
    const int M = 1 << logM, halfM = M >> 1, T = 1 << logT;
    const int lmax = (logM & 1) ? logM - 1 : logM;
    for (int lm = 2; lm <= lmax; lm += 2) {
        const int lq = lm - 2, q = 1 << lq, sh = logM - lm;
        const int total = T << (logM - 2);
        for (int b = tid; b < total; b += nthr) {
            int t, bi;
            if (tfast) { t = b & (T - 1); bi = b >> logT; } else { bi = b & ((M >> 2) - 1); t = b >> (logM - 2); }
            const int j = bi & (q - 1);
            const int base = ((bi >> lq) << lm) + j;
            cplx *p = s + t * TS + base * IS;
            const int qs = q * IS;
            cplx c0 = p[0], c2 = p[qs], c1 = p[2 * qs], c3 = p[3 * qs];
            const int k = j << sh;
            if (k) {
                c1 = cmulc(c1, tw[k]);
                c2 = cmulc(c2, tw[2 * k]);
                c3 = cmulc(c3, tw3(tw, 3 * k, halfM));
            }
            cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
            p[0] = cadd(s0, s2); p[qs] = cadd(s1, s3); p[2 * qs] = csub(s0, s2); p[3 * qs] = csub(s1, s3);
        }
        __syncthreads();
    }
    if (logM & 1) { // radix-2 tail stage
        const int total = T * halfM;
        for (int b = tid; b < total; b += nthr) {
            int t, j;
            if (tfast) { t = b & (T - 1); j = b >> logT; } else { j = b & (halfM - 1); t = b >> (logM - 1); }
            cplx *p = s + t * TS;
            cplx a = p[j * IS], c = cmulc(p[(j + halfM) * IS], tw[j]);
            p[j * IS] = cadd(a, c);
            p[(j + halfM) * IS] = csub(a, c);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// Row transforms (point-fastest layout).  A plain in-place radix-4 pass has 64-byte lane
// strides in its last two stages (4-way LDS bank conflicts, measured 51 % of the LDS cycles of
// the SSFM row pass), so rows use a padded layout -- one 16-byte slot after every 16 points,
// physical(i) = i + (i >> 4) -- and finish (DIF) / start (DIT) with a 16-point transform held
// entirely in registers: lane u reads the 16 contiguous points of block u (lane stride 17
// slots => conflict-free), does two radix-4 layers in registers and writes them back.  The
// remaining stages (sub-block length >= 16) keep the radix-4 scheme on 16-aligned runs.
__device__ __forceinline__ int row_phys(int i) { return i + (i >> 4); }
__device__ __forceinline__ int row_pitch(int M) { return M + (M >> 4); }
// Which point a lane takes when consecutive lanes sweep consecutive points of a padded row.  A wave's ds_read_b128 is
// served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- chosen so that 64
// lanes reading 64 CONSECUTIVE 16-byte slots meet no bank conflict.  Behind the pad slots lane l would read slot
// l + (l >> 4): lanes 20-27 land on slots 5..12 (mod 16) and collide with lane 12 -- every stride-1 read of a padded row
// then takes twice its cycles (measured: 32 % of k_row's LDS cycles were conflicts).  So within every aligned block of 16
// points the lanes are ROTATED by the block's pad count: lane l takes point (l & ~15) | ((l - pads) & 15), whose slot is
// congruent to l again (mod 16).  i: the lane's linear point index, M: the (power-of-two) row length.
__device__ __forceinline__ int row_lane_point(int i, int M) { return (i & ~15) | ((i - ((i & (M - 1)) >> 4)) & 15); }

#define PLX_C8 0.92387953251128673848  /* cos(pi/8) */
#define PLX_S8 0.38268343236508978178  /* sin(pi/8) */
#define PLX_R2 0.70710678118654752440  /* sqrt(1/2) */

// in-register 16-point DIF, natural in -> bit-reversed out (same placement rule as lds_fft_dif)
__device__ __forceinline__ void r16_dif(cplx *x)
{
    const cplx w1 = make_double2(PLX_C8, -PLX_S8), w2 = make_double2(PLX_R2, -PLX_R2), w3 = make_double2(PLX_S8, -PLX_C8);
    const cplx w6 = make_double2(-PLX_R2, -PLX_R2), w9 = make_double2(-PLX_C8, PLX_S8);
#pragma unroll
    for (int j = 0; j < 4; j++) { // layer m = 16, q = 4
        const cplx a0 = x[j], a1 = x[j + 4], a2 = x[j + 8], a3 = x[j + 12];
        const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
        cplx y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
        if (j == 1) { y1 = cmul(y1, w1); y2 = cmul(y2, w2); y3 = cmul(y3, w3); }
        if (j == 2) { y1 = cmul(y1, w2); y2 = cmulni(y2); y3 = cmul(y3, w6); }
        if (j == 3) { y1 = cmul(y1, w3); y2 = cmul(y2, w6); y3 = cmul(y3, w9); }
        x[j] = cadd(t0, t2); x[j + 4] = y2; x[j + 8] = y1; x[j + 12] = y3;
    }
#pragma unroll
    for (int b = 0; b < 16; b += 4) { // layer m = 4, q = 1
        const cplx a0 = x[b], a1 = x[b + 1], a2 = x[b + 2], a3 = x[b + 3];
        const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
        x[b] = cadd(t0, t2); x[b + 1] = csub(t0, t2); x[b + 2] = cadd(t1, t3); x[b + 3] = csub(t1, t3);
    }
}

// in-register 16-point unscaled inverse DIT, bit-reversed in -> natural out
__device__ __forceinline__ void r16_dit(cplx *x)
{
    const cplx w1 = make_double2(PLX_C8, -PLX_S8), w2 = make_double2(PLX_R2, -PLX_R2), w3 = make_double2(PLX_S8, -PLX_C8);
    const cplx w6 = make_double2(-PLX_R2, -PLX_R2), w9 = make_double2(-PLX_C8, PLX_S8);
#pragma unroll
    for (int b = 0; b < 16; b += 4) { // layer m = 4
        const cplx c0 = x[b], c2 = x[b + 1], c1 = x[b + 2], c3 = x[b + 3];
        const cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
        x[b] = cadd(s0, s2); x[b + 1] = cadd(s1, s3); x[b + 2] = csub(s0, s2); x[b + 3] = csub(s1, s3);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) { // layer m = 16
        cplx c0 = x[j], c2 = x[j + 4], c1 = x[j + 8], c3 = x[j + 12];
        if (j == 1) { c1 = cmulc(c1, w1); c2 = cmulc(c2, w2); c3 = cmulc(c3, w3); }
        if (j == 2) { c1 = cmulc(c1, w2); c2 = cmuli(c2); c3 = cmulc(c3, w6); }
        if (j == 3) { c1 = cmulc(c1, w3); c2 = cmulc(c2, w6); c3 = cmulc(c3, w9); }
        const cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
        x[j] = cadd(s0, s2); x[j + 4] = cadd(s1, s3); x[j + 8] = csub(s0, s2); x[j + 12] = csub(s1, s3);
    }
}

// Second level of a 256-point transform held in registers: y[k] is point j + 16k of the transform
// (k = r1 + 4 r2).  Stages m = 64 (q = 16) and m = 256 (q = 64) only combine points that share j, so
// with the 16-point blocks done by r16_dit / r16_dif a 256-point column transform needs ONE LDS
// exchange instead of four read+write passes.  tw: half table W_256^k, k < 128.
__device__ __forceinline__ void lvl2_dit256(cplx *y, int j, const cplx *tw)
{
    const int k6 = 4 * j;
    const cplx u1 = tw[k6], u2 = tw[2 * k6], u3 = tw3(tw, 3 * k6, 128);
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++) { // m = 64
        cplx c0 = y[4 * r2], c2 = y[4 * r2 + 1], c1 = y[4 * r2 + 2], c3 = y[4 * r2 + 3];
        c1 = cmulc(c1, u1); c2 = cmulc(c2, u2); c3 = cmulc(c3, u3);   // (W^0 = 1 for j = 0: no divergent guard, one basic block)
        const cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
        y[4 * r2] = cadd(s0, s2); y[4 * r2 + 1] = cadd(s1, s3); y[4 * r2 + 2] = csub(s0, s2); y[4 * r2 + 3] = csub(s1, s3);
    }
#pragma unroll
    for (int r1 = 0; r1 < 4; r1++) { // m = 256
        const int k8 = j + 16 * r1;
        cplx c0 = y[r1], c2 = y[r1 + 4], c1 = y[r1 + 8], c3 = y[r1 + 12];
        c1 = cmulc(c1, tw[k8]); c2 = cmulc(c2, tw[2 * k8]); c3 = cmulc(c3, tw3(tw, 3 * k8, 128));
        const cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
        y[r1] = cadd(s0, s2); y[r1 + 4] = cadd(s1, s3); y[r1 + 8] = csub(s0, s2); y[r1 + 12] = csub(s1, s3);
    }
}
// The same two levels with the ODD twiddles of the m = 256 stage (W^k8, W^3k8) taken from a second table, twa.  With twa = -tw
// the stage's odd outputs change sign, which for the inverse form is the same as delivering the upper and lower halves of
// y[] swapped (y[r1] <-> y[r1+8], y[r1+4] <-> y[r1+12]: s0 +- s2, s1 +- s3 with s2, s3 negated), and for the forward form the
// same as ACCEPTING them swapped -- exactly, bit for bit (negations are exact).  k_colx16 gives the lanes of the second
// polarisation the negated table, so that between its two transforms the register pair (k, k + 8) of a lane pair (t, t ^ 8)
// holds the two polarisations of ONE sample without any per-lane select.
__device__ __forceinline__ void lvl2_dit256s(cplx *y, int j, const cplx *tw, const cplx *twa)
{
    const int k6 = 4 * j;
    const cplx u1 = tw[k6], u2 = tw[2 * k6], u3 = tw3(tw, 3 * k6, 128);
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++) { // m = 64
        cplx c0 = y[4 * r2], c2 = y[4 * r2 + 1], c1 = y[4 * r2 + 2], c3 = y[4 * r2 + 3];
        c1 = cmulc(c1, u1); c2 = cmulc(c2, u2); c3 = cmulc(c3, u3);
        const cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
        y[4 * r2] = cadd(s0, s2); y[4 * r2 + 1] = cadd(s1, s3); y[4 * r2 + 2] = csub(s0, s2); y[4 * r2 + 3] = csub(s1, s3);
    }
#pragma unroll
    for (int r1 = 0; r1 < 4; r1++) { // m = 256
        const int k8 = j + 16 * r1;
        cplx c0 = y[r1], c2 = y[r1 + 4], c1 = y[r1 + 8], c3 = y[r1 + 12];
        c1 = cmulc(c1, twa[k8]); c2 = cmulc(c2, tw[2 * k8]); c3 = cmulc(c3, tw3(twa, 3 * k8, 128));
        const cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
        y[r1] = cadd(s0, s2); y[r1 + 4] = cadd(s1, s3); y[r1 + 8] = csub(s0, s2); y[r1 + 12] = csub(s1, s3);
    }
}
__device__ __forceinline__ void lvl2_dif256s(cplx *y, int j, const cplx *tw, const cplx *twa)
{
#pragma unroll
    for (int r1 = 0; r1 < 4; r1++) { // m = 256
        const int k8 = j + 16 * r1;
        const cplx a0 = y[r1], a1 = y[r1 + 4], a2 = y[r1 + 8], a3 = y[r1 + 12];
        const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
        cplx y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
        y1 = cmul(y1, twa[k8]); y2 = cmul(y2, tw[2 * k8]); y3 = cmul(y3, tw3(twa, 3 * k8, 128));
        y[r1] = cadd(t0, t2); y[r1 + 4] = y2; y[r1 + 8] = y1; y[r1 + 12] = y3;
    }
    const int k6 = 4 * j;
    const cplx u1 = tw[k6], u2 = tw[2 * k6], u3 = tw3(tw, 3 * k6, 128);
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++) { // m = 64
        const cplx a0 = y[4 * r2], a1 = y[4 * r2 + 1], a2 = y[4 * r2 + 2], a3 = y[4 * r2 + 3];
        const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
        cplx y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
        y1 = cmul(y1, u1); y2 = cmul(y2, u2); y3 = cmul(y3, u3);
        y[4 * r2] = cadd(t0, t2); y[4 * r2 + 1] = y2; y[4 * r2 + 2] = y1; y[4 * r2 + 3] = y3;
    }
}
__device__ __forceinline__ void lvl2_dif256(cplx *y, int j, const cplx *tw)
{
#pragma unroll
    for (int r1 = 0; r1 < 4; r1++) { // m = 256
        const int k8 = j + 16 * r1;
        const cplx a0 = y[r1], a1 = y[r1 + 4], a2 = y[r1 + 8], a3 = y[r1 + 12];
        const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
        cplx y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
        y1 = cmul(y1, tw[k8]); y2 = cmul(y2, tw[2 * k8]); y3 = cmul(y3, tw3(tw, 3 * k8, 128));
        y[r1] = cadd(t0, t2); y[r1 + 4] = y2; y[r1 + 8] = y1; y[r1 + 12] = y3;
    }
    const int k6 = 4 * j;
    const cplx u1 = tw[k6], u2 = tw[2 * k6], u3 = tw3(tw, 3 * k6, 128);
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++) { // m = 64
        const cplx a0 = y[4 * r2], a1 = y[4 * r2 + 1], a2 = y[4 * r2 + 2], a3 = y[4 * r2 + 3];
        const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
        cplx y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
        y1 = cmul(y1, u1); y2 = cmul(y2, u2); y3 = cmul(y3, u3);
        y[4 * r2] = cadd(t0, t2); y[4 * r2 + 1] = y2; y[4 * r2 + 2] = y1; y[4 * r2 + 3] = y3;
    }
}

// The register levels lvl2_dif256 / lvl2_dit256 for any length M = 16 S: y[k] is point j + S k (j < S) of a length-M
// transform, W a functor e -> W_M^e (e < 3M/4).  Stages m = M (q = M/4) and m = M/4 (q = S) only combine points that
// share j, so a transform of 16^L points is L such levels with one exchange between consecutive ones (4096-point
// rows: k_row4k).
// (u1, u2, u3 = w(4 j), w(8 j), w(12 j): the second stage's twiddles depend on the lane only; a caller whose lanes would
//  fetch them from the same LDS banks -- sixteen lanes of a transform at strides of 4, 8 and 12 entries -- keeps them in registers)
template <int S, class W> __device__ __forceinline__ void lvl2_dif(cplx *y, int j, W w, cplx u1, cplx u2, cplx u3)
{
#pragma unroll
    for (int r1 = 0; r1 < 4; r1++) { // m = M
        const int e = j + S * r1;
        const cplx a0 = y[r1], a1 = y[r1 + 4], a2 = y[r1 + 8], a3 = y[r1 + 12];
        const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
        cplx y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
        y1 = cmul(y1, w(e)); y2 = cmul(y2, w(2 * e)); y3 = cmul(y3, w(3 * e));
        y[r1] = cadd(t0, t2); y[r1 + 4] = y2; y[r1 + 8] = y1; y[r1 + 12] = y3;
    }
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++) { // m = M/4
        const cplx a0 = y[4 * r2], a1 = y[4 * r2 + 1], a2 = y[4 * r2 + 2], a3 = y[4 * r2 + 3];
        const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
        cplx y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
        y1 = cmul(y1, u1); y2 = cmul(y2, u2); y3 = cmul(y3, u3);
        y[4 * r2] = cadd(t0, t2); y[4 * r2 + 1] = y2; y[4 * r2 + 2] = y1; y[4 * r2 + 3] = y3;
    }
}
template <int S, class W> __device__ __forceinline__ void lvl2_dif(cplx *y, int j, W w)
{
    const int e = 4 * j;
    lvl2_dif<S, W>(y, j, w, w(e), w(2 * e), w(3 * e));
}
template <int S, class W> __device__ __forceinline__ void lvl2_dit(cplx *y, int j, W w, cplx u1, cplx u2, cplx u3)
{
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++) { // m = M/4
        cplx c0 = y[4 * r2], c2 = y[4 * r2 + 1], c1 = y[4 * r2 + 2], c3 = y[4 * r2 + 3];
        c1 = cmulc(c1, u1); c2 = cmulc(c2, u2); c3 = cmulc(c3, u3);
        const cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
        y[4 * r2] = cadd(s0, s2); y[4 * r2 + 1] = cadd(s1, s3); y[4 * r2 + 2] = csub(s0, s2); y[4 * r2 + 3] = csub(s1, s3);
    }
#pragma unroll
    for (int r1 = 0; r1 < 4; r1++) { // m = M
        const int e1 = j + S * r1;
        cplx c0 = y[r1], c2 = y[r1 + 4], c1 = y[r1 + 8], c3 = y[r1 + 12];
        c1 = cmulc(c1, w(e1)); c2 = cmulc(c2, w(2 * e1)); c3 = cmulc(c3, w(3 * e1));
        const cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
        y[r1] = cadd(s0, s2); y[r1 + 4] = cadd(s1, s3); y[r1 + 8] = csub(s0, s2); y[r1 + 12] = csub(s1, s3);
    }
}
template <int S, class W> __device__ __forceinline__ void lvl2_dit(cplx *y, int j, W w)
{
    const int e = 4 * j;
    lvl2_dit<S, W>(y, j, w, w(e), w(2 * e), w(3 * e));
}
// W_256^e (e < 192) from the half table W_256^k, k < 128
struct Tw256half {
    const cplx *t;
    __device__ __forceinline__ cplx operator()(int e) const { return tw3(t, e, 128); }
};
// twiddle functors over the COMPACT table of W_4096 (W^{4k}, k < 512, then W^0..3: 8 KiB instead of the 32 KiB half
// table, so that two row workgroups share a CU): W_4096^e, and W_256^e = W_4096^{16 e}
struct Tw4096 {
    const cplx *t;
    __device__ __forceinline__ cplx operator()(int e) const
    {
        const int i = e & 2047;
        const cplx w = cmul(t[i >> 2], t[512 + (i & 3)]);
        const bool neg = e >= 2048;
        return make_double2(neg ? -w.x : w.x, neg ? -w.y : w.y);
    }
};
struct Tw256of4096 {
    const cplx *t;
    __device__ __forceinline__ cplx operator()(int e) const
    {
        const int e16 = e << 4, i = e16 & 2047;
        const cplx w = t[i >> 2];
        const bool neg = e16 >= 2048;
        return make_double2(neg ? -w.x : w.x, neg ? -w.y : w.y);
    }
};

// W_M^e (e < M) over the compact table of W_M: t[k] = W_M^{4k}, k < M/8, then W_M^0..3 (M/8 + 4 entries)
template <int M> struct TwCompact {
    const cplx *t;
    __device__ __forceinline__ cplx operator()(int e) const
    {
        const int i = e & (M / 2 - 1);
        const cplx w = cmul(t[i >> 2], t[M / 8 + (i & 3)]);
        const bool neg = (e & (M - 1)) >= M / 2;
        return make_double2(neg ? -w.x : w.x, neg ? -w.y : w.y);
    }
};

// W_M^e (e < M) over the half table t[k] = W_M^k, k < M/2
template <int M> struct TwHalf {
    const cplx *t;
    __device__ __forceinline__ cplx operator()(int e) const
    {
        const cplx w = t[e & (M / 2 - 1)];
        const bool neg = (e & (M - 1)) >= M / 2;
        return make_double2(neg ? -w.x : w.x, neg ? -w.y : w.y);
    }
};

// The MIDDLE register level of a row of M = 16 * R * 16 points (R = 2, 4, 8: rows of 512, 1024 and 2048 points, k_rowreg):
// after the first level (lvl2_dif<M/16>) the row is sixteen independent blocks of M/16 = 16 R points; a thread holds, of a
// 256-point chunk of the row, the points j + 16 kk (kk < 16) = point j + 16 k of block h, kk = R h + k: 16 / R radix-R
// butterflies at stride 16, after which blocks of sixteen CONTIGUOUS points remain (r16_dif).  R = 8 is a radix-2 stage
// (m = 128) followed by a radix-4 stage (m = 64), the order of lds_fft_dif's in-place stages: the output order is the
// plain bit reversal whatever the grouping.  w: this lane's twiddles -- R = 2: W_32^j; R = 4: W_64^{j, 2j, 3j};
// R = 8: W_128^{j + 16 k} (k < 4), then W_64^{j, 2j, 3j}.
template <int R> __device__ __forceinline__ void lvlmid_dif(cplx *x, const cplx *w)
{
    if (R == 8) {
#pragma unroll
        for (int h = 0; h < 16; h += 8)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const cplx a = x[h + k], b = x[h + k + 4];
                x[h + k] = cadd(a, b);
                x[h + k + 4] = cmul(csub(a, b), w[k]);
            }
    }
    if (R == 2) {
#pragma unroll
        for (int h = 0; h < 16; h += 2) {
            const cplx a = x[h], b = x[h + 1];
            x[h] = cadd(a, b);
            x[h + 1] = cmul(csub(a, b), w[0]);
        }
    } else {
        const cplx w1 = w[R == 8 ? 4 : 0], w2 = w[R == 8 ? 5 : 1], w3 = w[R == 8 ? 6 : 2];
#pragma unroll
        for (int b = 0; b < 16; b += 4) {
            const cplx a0 = x[b], a1 = x[b + 1], a2 = x[b + 2], a3 = x[b + 3];
            const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
            x[b] = cadd(t0, t2); x[b + 1] = cmul(csub(t0, t2), w2); x[b + 2] = cmul(cadd(t1, t3), w1); x[b + 3] = cmul(csub(t1, t3), w3);
        }
    }
}
template <int R> __device__ __forceinline__ void lvlmid_dit(cplx *x, const cplx *w)
{
    if (R == 2) {
#pragma unroll
        for (int h = 0; h < 16; h += 2) {
            const cplx a = x[h], b = cmulc(x[h + 1], w[0]);
            x[h] = cadd(a, b);
            x[h + 1] = csub(a, b);
        }
    } else {
        const cplx w1 = w[R == 8 ? 4 : 0], w2 = w[R == 8 ? 5 : 1], w3 = w[R == 8 ? 6 : 2];
#pragma unroll
        for (int b = 0; b < 16; b += 4) {
            const cplx c0 = x[b], c2 = cmulc(x[b + 1], w2), c1 = cmulc(x[b + 2], w1), c3 = cmulc(x[b + 3], w3);
            const cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
            x[b] = cadd(s0, s2); x[b + 1] = cadd(s1, s3); x[b + 2] = csub(s0, s2); x[b + 3] = csub(s1, s3);
        }
    }
    if (R == 8) {
#pragma unroll
        for (int h = 0; h < 16; h += 8)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const cplx a = x[h + k], b = cmulc(x[h + k + 4], w[k]);
                x[h + k] = cadd(a, b);
                x[h + k + 4] = csub(a, b);
            }
    }
}

// One R-point set of that level on R consecutive registers (rows of 16 R points, k_rowsm: a thread's sixteen registers are
// 16 / R sets whose lane twiddles differ): x[q] = point i + 16 q, w as in lvlmid_dif for lane index i.
template <int R> __device__ __forceinline__ void radset_dif(cplx *x, const cplx *w)
{
    if (R == 2) {
        const cplx a = x[0], b = x[1];
        x[0] = cadd(a, b);
        x[1] = cmul(csub(a, b), w[0]);
        return;
    }
    if (R == 8) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const cplx a = x[k], b = x[k + 4];
            x[k] = cadd(a, b);
            x[k + 4] = cmul(csub(a, b), w[k]);
        }
    }
    const cplx w1 = w[R == 8 ? 4 : 0], w2 = w[R == 8 ? 5 : 1], w3 = w[R == 8 ? 6 : 2];
#pragma unroll
    for (int b = 0; b < R; b += 4) {
        const cplx a0 = x[b], a1 = x[b + 1], a2 = x[b + 2], a3 = x[b + 3];
        const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
        x[b] = cadd(t0, t2); x[b + 1] = cmul(csub(t0, t2), w2); x[b + 2] = cmul(cadd(t1, t3), w1); x[b + 3] = cmul(csub(t1, t3), w3);
    }
}
template <int R> __device__ __forceinline__ void radset_dit(cplx *x, const cplx *w)
{
    if (R == 2) {
        const cplx a = x[0], b = cmulc(x[1], w[0]);
        x[0] = cadd(a, b);
        x[1] = csub(a, b);
        return;
    }
    const cplx w1 = w[R == 8 ? 4 : 0], w2 = w[R == 8 ? 5 : 1], w3 = w[R == 8 ? 6 : 2];
#pragma unroll
    for (int b = 0; b < R; b += 4) {
        const cplx c0 = x[b], c2 = cmulc(x[b + 1], w2), c1 = cmulc(x[b + 2], w1), c3 = cmulc(x[b + 3], w3);
        const cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
        x[b] = cadd(s0, s2); x[b + 1] = cadd(s1, s3); x[b + 2] = csub(s0, s2); x[b + 3] = csub(s1, s3);
    }
    if (R == 8) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const cplx a = x[k], b = cmulc(x[k + 4], w[k]);
            x[k] = cadd(a, b);
            x[k + 4] = csub(a, b);
        }
    }
}

// T = 1<<logT padded rows of M = 1<<logM (M >= 16) points each, row t at s + t*row_pitch(M).
__device__ __forceinline__ void row_fft_dif(cplx *s, int logM, int logT, const cplx *tw, int tid, int nthr)
{
    const int M = 1 << logM, halfM = M >> 1, T = 1 << logT, TSp = row_pitch(M);
    int lm = logM;
    if (logM & 1) { // radix-2 head (halfM >= 16)
        const int total = T * halfM, hp = row_phys(halfM);
        for (int b = tid; b < total; b += nthr) {
            const int j = row_lane_point(b & (halfM - 1), M), t = b >> (logM - 1);
            cplx *p = s + t * TSp + row_phys(j);
            const cplx a = p[0], c = p[hp];
            p[0] = cadd(a, c);
            p[hp] = cmul(csub(a, c), tw[j]);
        }
        __syncthreads();
        lm--;
    }
    for (; lm >= 6; lm -= 2) { // radix-4 stages with q >= 16
        const int lq = lm - 2, q = 1 << lq, sh = logM - lm, qs = row_phys(q);
        const int total = T << (logM - 2);
        for (int b = tid; b < total; b += nthr) {
            const int bi = b & ((M >> 2) - 1), t = b >> (logM - 2);
            // (q >= 16: the lanes of a block of 16 butterflies are rotated by the block's pad count, see row_lane_point)
            const int blk0 = ((bi >> lq) << lm) + (bi & (q - 1) & ~15);
            const int base = blk0 + ((bi - (blk0 >> 4)) & 15);
            const int j = base & (q - 1);
            cplx *p = s + t * TSp + row_phys(base);
            cplx a0 = p[0], a1 = p[qs], a2 = p[2 * qs], a3 = p[3 * qs];
            cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulni(csub(a1, a3));
            const int k = j << sh;
            cplx y0 = cadd(t0, t2), y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
            y1 = cmul(y1, tw[k]);      // (no guard for k = 0: it diverges inside every wave and splits the basic block)
            y2 = cmul(y2, tw[2 * k]);
            y3 = cmul(y3, tw3(tw, 3 * k, halfM));
            p[0] = y0; p[qs] = y2; p[2 * qs] = y1; p[3 * qs] = y3;
        }
        __syncthreads();
    }
    { // 16-point tail in registers
        const int total = T << (logM - 4);
        for (int u = tid; u < total; u += nthr) {
            const int blk = u & ((M >> 4) - 1), t = u >> (logM - 4);
            cplx *p = s + t * TSp + blk * 17;
            cplx x[16];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = p[k];
            r16_dif(x);
#pragma unroll
            for (int k = 0; k < 16; k++) p[k] = x[k];
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void row_fft_dit(cplx *s, int logM, int logT, const cplx *tw, int tid, int nthr)
{
    const int M = 1 << logM, halfM = M >> 1, T = 1 << logT, TSp = row_pitch(M);
    {
        const int total = T << (logM - 4);
        for (int u = tid; u < total; u += nthr) {
            const int blk = u & ((M >> 4) - 1), t = u >> (logM - 4);
            cplx *p = s + t * TSp + blk * 17;
            cplx x[16];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = p[k];
            r16_dit(x);
#pragma unroll
            for (int k = 0; k < 16; k++) p[k] = x[k];
        }
        __syncthreads();
    }
    const int lmax = (logM & 1) ? logM - 1 : logM;
    for (int lm = 6; lm <= lmax; lm += 2) {
        const int lq = lm - 2, q = 1 << lq, sh = logM - lm, qs = row_phys(q);
        const int total = T << (logM - 2);
        for (int b = tid; b < total; b += nthr) {
            const int bi = b & ((M >> 2) - 1), t = b >> (logM - 2);
            // (q >= 16: the lanes of a block of 16 butterflies are rotated by the block's pad count, see row_lane_point)
            const int blk0 = ((bi >> lq) << lm) + (bi & (q - 1) & ~15);
            const int base = blk0 + ((bi - (blk0 >> 4)) & 15);
            const int j = base & (q - 1);
            cplx *p = s + t * TSp + row_phys(base);
            cplx c0 = p[0], c2 = p[qs], c1 = p[2 * qs], c3 = p[3 * qs];
            const int k = j << sh;
            c1 = cmulc(c1, tw[k]);
            c2 = cmulc(c2, tw[2 * k]);
            c3 = cmulc(c3, tw3(tw, 3 * k, halfM));
            cplx s0 = cadd(c0, c2), s1 = csub(c0, c2), s2 = cadd(c1, c3), s3 = cmuli(csub(c1, c3));
            p[0] = cadd(s0, s2); p[qs] = cadd(s1, s3); p[2 * qs] = csub(s0, s2); p[3 * qs] = csub(s1, s3);
        }
        __syncthreads();
    }
    if (logM & 1) {
        const int total = T * halfM, hp = row_phys(halfM);
        for (int b = tid; b < total; b += nthr) {
            const int j = row_lane_point(b & (halfM - 1), M), t = b >> (logM - 1);
            cplx *p = s + t * TSp + row_phys(j);
            const cplx a = p[0], c = cmulc(p[hp], tw[j]);
            p[0] = cadd(a, c);
            p[hp] = csub(a, c);
        }
        __syncthreads();
    }
}

// stage the half table W_M^k (k < M/2) from global memory into LDS
__device__ __forceinline__ void lds_load_twiddles(cplx *dst, const cplx *__restrict__ src, int halfM, int tid, int nthr)
{
    for (int k = tid; k < halfM; k += nthr) dst[k] = src[k];
}

// ---- index helper (host and device) ----------------------------------------------
__host__ __device__ static inline unsigned plx_bitrev(unsigned i, int bits)
{
    unsigned r = 0;
    for (int b = 0; b < bits; b++) r = (r << 1) | ((i >> b) & 1u);
    return r;
}

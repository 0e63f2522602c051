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
    if (k3 >= halfM) {
        cplx w = tw[k3 - halfM];
        return make_double2(-w.x, -w.y);
    }
    return tw[k3];
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

// stage the half table W_M^k (k < M/2) from global memory into LDS
__device__ __forceinline__ void lds_load_twiddles(cplx *dst, const cplx *__restrict__ src, int halfM, int tid, int nthr)
{
    for (int k = tid; k < halfM; k += nthr) dst[k] = src[k];
}

// ---- host helpers ---------------------------------------------------------------
static inline unsigned plx_bitrev(unsigned i, int bits)
{
    unsigned r = 0;
    for (int b = 0; b < bits; b++) r = (r << 1) | ((i >> b) & 1u);
    return r;
}

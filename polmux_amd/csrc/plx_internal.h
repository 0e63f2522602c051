// Library-internal C++ interfaces shared between translation units (not part of the C ABI, hidden).
#pragma once
#include "plx_common.h"
#include "../../include/polmux_hip.h"

#define PLX_HIDDEN __attribute__((visibility("hidden")))

// The batched four-step FFT engine of an SSFM plan used as a spectral filter:
// u = ifft(fft(u) .* H) column-wise for `nframes` frames of the plan's nfc columns (receiver_cohmix.m:183,
// :232-233, :300-304).  H is uploaded once in the order the row pass visits the spectrum.
PLX_HIDDEN int plx_ssfm_filter_table(plx_ssfm *P, const double *h_re, const double *h_im, cplx **d_out);
// d_umat (optional, dual-polarisation single-field plans): [nframes][N][3] = first row (U11, U12) of a per-frequency
// SU(2) matrix and a scalar factor Hgvd, in the row-pass order; the pass then applies (Hgvd U)^H to [x; y]
// instead of the scalar H (inverse_pmd.m:130-143).
PLX_HIDDEN int plx_ssfm_filter_dev(plx_ssfm *P, cplx *d_ux, cplx *d_uy, const cplx *d_hmul, int nframes, void *stream,
                                   const cplx *d_umat = nullptr);
// natural frequency index k of row-pass position pos (and the plan's N), for kernels that fill such tables
PLX_HIDDEN void plx_ssfm_geometry(const plx_ssfm *P, int *p1, int *p2);

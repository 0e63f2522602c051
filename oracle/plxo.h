/*
 * plxo.h -- CPU ORACLE for the Polmux/Optilux hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory restates, in plain C99 double precision, the algorithm of the
 * reference's hot path (fiber.m SSFM -> CDE_OFDE.m overlap-save -> CMA/EASI 2x2
 * butterfly + DspPdmCohQpsk.m drivers -> ber_estimate.m/mc_estimate.m).  Every
 * function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load it -- as the checker / the timed CPU baseline, never as the product
 * path.  The product (polmux_amd/) never imports anything from here.
 *
 * PARITY PINNING (see DESIGN.md "Oracle"):
 *   - The reference holds no automated tests and no golden vectors for the hot
 *     path.  Its only literal known answers (fastshift.m:8-11, nmod.m:8-9,
 *     pattern.m:48-51) and its stated analytic invariants (fiber.m:172-174
 *     exact SPM, fiber.m:762-773 single-step GVD, unitarity of every sub-step,
 *     CDE_OFDE.m:104-116 identity for H==1, ...) are all checked in
 *     tests/test_oracle_*.py.
 *   - The reference's MATLAB sources cannot run here (no MATLAB/Octave) and its
 *     four C MEX files need MATLAB's <mex.h>, which this image lacks; building
 *     them against a hand-written stand-in header is not a reference build, so
 *     none is made.  For the CMA/EASI recurrences and the .m loops the oracle
 *     is therefore an op-for-op restatement pinned only by those invariants:
 *     "parity unpinned" beyond them.
 *
 * Array conventions follow MATLAB: column-major, fields are [nfft x nfc]
 * matrices of interleaved complex double (C99 double complex).  The CMA/EASI
 * kernels keep the MEX split-plane (re/im) layout of the reference C.
 */
#ifndef PLXO_H
#define PLXO_H

#include <complex.h>
#include <stddef.h>

typedef double complex plxo_c;

/* ---- FFT (MATLAB fft/ifft semantics: forward unscaled, inverse 1/n) ---- */
void plxo_fft(plxo_c *x, long n, int inverse);

/* ---- fastexp.c:37-44 / fastexp.m:28 ---- */
void plxo_fastexp(const double *x, double *yr, double *yi, long m);

/* ---- fiber.m step controller ---- */
double plxo_nextstep(double dzmax, double phimax, const double *gam, int nfc,
                     double alphalin, const plxo_c *ux, const plxo_c *uy,
                     long nfft);
/* returns ntrunk; dzb must hold >= nplates+2 entries */
int plxo_checkstep(double zprop, double dz, double lcorr, double *dz_miss,
                   int nz_old, double *dzb, int *nmem);

/* ---- fiber.m operators ---- */
void plxo_lin_step(const double *betat, double dz, plxo_c *u, long nfft, int nfc);
void plxo_nl_step(double alphalin, const double *gam, double dz, plxo_c *u,
                  long nfft, int nfc, int spm, int xpm);
int plxo_matrix_nl_step(long nfft, int ismanakov, double alphalin,
                        const double *gam, double dz, plxo_c *ux, plxo_c *uy,
                        int nfc, int spm, int xpm);
void plxo_matrix_step(const double *betat, const double *db1, const double *dzb,
                      int ntrunk, plxo_c *ux, plxo_c *uy, long nfft, int nfc,
                      const double *db0, const double *theta,
                      const double *epsilon, double lcorr, int ntot, int nmem);

/* ---- fiber.m propagation loops.  Return 0, or <0 on the reference's error()s */
#define PLXO_ERR_XPM_CNLSE (-1)   /* fiber.m:854 */
int plxo_matrix_ssfm(plxo_c *ux, plxo_c *uy, const double *betat,
                     const double *db1, double dzmaxt, double dphimaxt,
                     const double *gam, double alphalin, int nfc, long nfft,
                     double Lf, int nplates, int manakov, const int *fls,
                     const double *db0, const double *theta,
                     const double *epsilon, double *firstdz, int *ncycle);
/* test diagnostics: record the step lengths nextstep returns inside plxo_matrix_ssfm (fiber.m:512, :534) */
void plxo_set_step_log(double *buf, int cap);
int plxo_step_log_count(void);
/* test diagnostics: step k of the next plxo_matrix_ssfm call takes dz[k] instead of nextstep's result (k < n) */
void plxo_set_step_replay(const double *dz, int n);
int plxo_scalar_ssfm(plxo_c *u, const double *betat, double dzmaxt,
                     double dphimaxt, const double *gam, double alphalin,
                     long nfft, int nfc, double Lf, const int *fls, int tolflag,
                     double trg_err, double trg_safety, double *firstdz,
                     int *ncycle);
int plxo_scalar_a_ssfm(plxo_c *u, const double *betat, double dzmaxt,
                       double dphimaxt, const double *gam, double alphalin,
                       long nfft, int nfc, double Lf, double trg_err,
                       double trg_safety, const int *fls, double *firstdz,
                       int *ncycle, int *nrej);

/* ---- CDE_OFDE.m ---- */
/* H on the fftshift-ordered grid, CDE_OFDE.m:29-38 */
void plxo_cde_transfer(plxo_c *H, long fftlen, double fs, double lambda_ref,
                       double span, double D, double S);
/* CDE_OFDE.m:62-125.  y has nx entries. returns 0, or the 1-based index of the
 * failed argument check (the reference display()s and returns y=[]) */
int plxo_overlap_both_trans(const plxo_c *x, long nx, const plxo_c *H, long N,
                            long L, plxo_c *y);
int plxo_cde_ofde(const plxo_c *inx, const plxo_c *iny, long nx, double fs,
                  double lambda_ref, double span, double D, double S,
                  long fftlen, long L, plxo_c *outx, plxo_c *outy);

/* ---- cmaadaptivefilter.c / easiadaptivefilter.c ---- */
void plxo_cmafilter(const double *xr, const double *xi, int Ndim, double *h1r,
                    double *h1i, double *h2r, double *h2i, int Ntap, double mu,
                    const double *R, double *yr, double *yi, int dontskip);
void plxo_easifilter(const double *xr, const double *xi, int Ndim, double *h1r,
                     double *h1i, double *h2r, double *h2i, int Ntap, double mu,
                     double *yr, double *yi, int dontskip);
/* gateway argument checks (cmaadaptivefilter.c:118-133): 0 ok,
 * 1 "Ntaps should be an ODD INTEGER.", 2 "Samples x symbol should be either 1 or 2." */
int plxo_cma_gateway_check(double Ntap, double sps, int check_odd);
/* the .m twins (cmaadaptivefilter.m:52-72, easiadaptivefilter.m:51-84): xx [2][Ndim], h [2][ntap], y [2][Ndim-ntap+1] */
void plxo_cmafilter_m(const plxo_c *xx, int Ndim, plxo_c *h1, plxo_c *h2, int ntap, double mu,
                      const double *R, plxo_c *y);
void plxo_easifilter_m(const plxo_c *xx, int Ndim, plxo_c *h1, plxo_c *h2, int ntap, double mu, plxo_c *y);

/* ---- DspPdmCohQpsk.m drivers (x: [L x 2] column-major complex) ---- */
/* M: 2x2 complex initial centre-tap matrix, row-major M[0]=M(1,1) M[1]=M(1,2)...
 * y: [L x 2]; h1,h2 out: [taps x 2] column-major; returns passes executed */
int plxo_cmapolardemux(const plxo_c *x, long L, const plxo_c *M, int taps,
                       double mu, const double *R, plxo_c *y, plxo_c *h1,
                       plxo_c *h2);
int plxo_easipolardemux(const plxo_c *x, long L, const plxo_c *M, double mu,
                        plxo_c *y, plxo_c *h1, plxo_c *h2);
int plxo_easipolardemux_m(const plxo_c *x, long L, const plxo_c *M, double mu,
                          plxo_c *y, plxo_c *h1, plxo_c *h2);

typedef struct {
    int workatbaudrate;   /* DspPdmCohQpsk.m:12 */
    int applynlr;         /* :17 */
    double nlralpha;
    double power_mw;      /* GSTATE.POWER(chNum), :22 */
    int applypol;         /* :26 */
    int polmethod;        /* 0 singlepol, 1 cma, 2 easi, 3 combo */
    /* cmaparams / easiparams */
    double cma_R[2];
    double cma_mu;
    int cma_taps;
    int cma_txpolars;
    double cma_phizero;
    double easi_mu;
    int easi_txpolars;
    double easi_phizero;
    int modorder, freqavg, phasavg, poworder;
    /* params.mat (DspPdmCohQpsk.m:148-149, :201-202): row-major 2x2 complex (re,im) */
    int cma_has_mat, easi_has_mat;
    double cma_mat[8], easi_mat[8];
    int mfile_twins;      /* 1: the drivers run the .m twins of the filters (no MEX compiled) */
} plxo_dsp_params;

/* in: [Lin x ncol] (ncol 1 or 2); out: [Lout x ncol], Lout = ceil(Lin/2) unless
 * workatbaudrate.  Returns Lout (<0 on error). */
long plxo_dsp_pdm_coh_qpsk(const plxo_c *in, long Lin, int ncol,
                           const plxo_dsp_params *p, plxo_c *out);
/* vitvit, DspPdmCohQpsk.m:97-123 (s: [L x ncol], theta out real) */
void plxo_vitvit(const plxo_c *s, long L, int ncol, int P, int M, int k,
                 int applyunwrap, double *theta);
/* fastshift.m:45-62 on rows of [L x ncol] */
void plxo_fastshift(const plxo_c *x, long L, int ncol, long n, plxo_c *y);
long plxo_nmod(long A, long N); /* nmod.m:31 */
/* MATLAB unwrap along a column (tolerance pi) */
void plxo_unwrap(double *p, long n);

/* samp2pat.m:61-66, 'coherent': phase [L x ncol] -> pat [L x 2*ncol] (0/1 bytes) */
void plxo_samp2pat_coherent(const double *phase, long L, int ncol,
                            unsigned char *pat);

/* ---- ber_estimate.m:97-143 / mc_estimate.m:133-212 state machines ---- */
#define PLXO_MC_MAXDIM 256
typedef struct {
    int first;                       /* 0 == isempty(first) */
    int dim;
    double n[PLXO_MC_MAXDIM], avg[PLXO_MC_MAXDIM], var[PLXO_MC_MAXDIM];
    double varlim[2][PLXO_MC_MAXDIM];
    int cond[PLXO_MC_MAXDIM];
    double epsilon[2];
} plxo_mc_state;

double plxo_erfcinv(double y);
/* one call of ber_estimate: err = number of differing entries, M = numel(pat).
 * has_stop: x.stop given.  nind is 1-based.  Outputs vectors of length dim. */
void plxo_ber_estimate(plxo_mc_state *st, double err, double M, int dim,
                       int nind, int has_stop, double stop1, double stop2,
                       double nmin, int *cond, double *avgber, double *nruns,
                       double *stdber);
/* one call of mc_estimate for a sample vector s[0..M-1] */
void plxo_mc_estimate(plxo_mc_state *st, const double *s, long M, int dim,
                      int nind, int has_stop, double stop1, double stop2,
                      double nmin, int method_var, int *cond, double *mean,
                      double *var, double *nruns, double *stdmean,
                      double *varlim /* [2 x dim] column-major */);

#endif

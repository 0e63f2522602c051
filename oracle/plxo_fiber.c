/*
 * plxo_fiber.c -- CPU ORACLE (test infrastructure, see plxo.h) for fiber.m's
 * split-step Fourier propagator and fastexp.  Op-for-op restatement of
 * /root/reference/fiber.m:459-1009 and fastexp.c:37-44.
 */
#include "plxo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ FFT --- */
/* Self-contained transform standing in for MATLAB's fft/ifft (FFTW, not part of
 * the reference): iterative radix-2 for powers of two, O(n^2) DFT otherwise.
 * Twiddles come straight from libm cos/sin (one table per size, cached). */
static double *tw_cache[64];

static const double *twiddles(long n, int lg)
{
    if (!tw_cache[lg]) {
        double *t = (double *)malloc(sizeof(double) * n); /* n/2 pairs */
        for (long k = 0; k < n / 2; k++) {
            double a = -2.0 * M_PI * (double)k / (double)n;
            t[2 * k] = cos(a);
            t[2 * k + 1] = sin(a);
        }
        tw_cache[lg] = t;
    }
    return tw_cache[lg];
}

static void dft_naive(plxo_c *x, long n, int inverse)
{
    plxo_c *y = (plxo_c *)malloc(sizeof(plxo_c) * n);
    double sgn = inverse ? 2.0 : -2.0;
    for (long k = 0; k < n; k++) {
        double sr = 0, si = 0;
        for (long j = 0; j < n; j++) {
            long p = (k * j) % n;
            double a = sgn * M_PI * (double)p / (double)n;
            double c = cos(a), s = sin(a);
            sr += creal(x[j]) * c - cimag(x[j]) * s;
            si += creal(x[j]) * s + cimag(x[j]) * c;
        }
        y[k] = sr + I * si;
    }
    memcpy(x, y, sizeof(plxo_c) * n);
    free(y);
}

void plxo_fft(plxo_c *x, long n, int inverse)
{
    if (n <= 1) return;
    int lg = 0;
    while ((1L << lg) < n) lg++;
    if ((1L << lg) != n) {
        dft_naive(x, n, inverse);
    } else {
        const double *tw = twiddles(n, lg);
        /* bit reversal */
        for (long i = 1, j = 0; i < n; i++) {
            long bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) { plxo_c t = x[i]; x[i] = x[j]; x[j] = t; }
        }
        for (long len = 2; len <= n; len <<= 1) {
            long half = len >> 1, stride = n / len;
            for (long i = 0; i < n; i += len) {
                for (long k = 0; k < half; k++) {
                    double wr = tw[2 * k * stride];
                    double wi = inverse ? -tw[2 * k * stride + 1] : tw[2 * k * stride + 1];
                    double br = creal(x[i + k + half]), bi = cimag(x[i + k + half]);
                    double tr = br * wr - bi * wi, ti = br * wi + bi * wr;
                    double ar = creal(x[i + k]), ai = cimag(x[i + k]);
                    x[i + k] = (ar + tr) + I * (ai + ti);
                    x[i + k + half] = (ar - tr) + I * (ai - ti);
                }
            }
        }
    }
    if (inverse) {
        double s = 1.0 / (double)n;
        for (long i = 0; i < n; i++) x[i] = (creal(x[i]) * s) + I * (cimag(x[i]) * s);
    }
}

/* -------------------------------------------------------------- fastexp --- */
/* fastexp.c:37-44: descending index, cos then sin of each element. */
void plxo_fastexp(const double *x, double *yr, double *yi, long m)
{
    while (m > 0) {
        m--;
        yr[m] = cos(x[m]);
        yi[m] = sin(x[m]);
    }
}

static inline plxo_c cexpi(double a) { return cos(a) + I * sin(a); } /* fastexp.m:28 */

/* complex product written out as MATLAB evaluates it: (ac-bd) + i(ad+bc) */
static inline plxo_c cmul(plxo_c a, plxo_c b)
{
    double ar = creal(a), ai = cimag(a), br = creal(b), bi = cimag(b);
    return (ar * br - ai * bi) + I * (ar * bi + ai * br);
}
static inline plxo_c rmul(double r, plxo_c a) { return (r * creal(a)) + I * (r * cimag(a)); }

/* ------------------------------------------------------------- nextstep --- */
/* fiber.m:682-715.  gam has nfc entries (one field: nfc==1).  uy==NULL <=> ~isv. */
double plxo_nextstep(double dzmax, double phimax, const double *gam, int nfc,
                     double alphalin, const plxo_c *ux, const plxo_c *uy, long nfft)
{
    double Pmax = -INFINITY;
    for (int k = 0; k < nfc; k++) {
        double Umax = -INFINITY; /* column-wise max, fiber.m:694-696 */
        for (long n = 0; n < nfft; n++) {
            plxo_c a = ux[k * nfft + n];
            double p = creal(a) * creal(a) + cimag(a) * cimag(a);
            if (uy) {
                plxo_c b = uy[k * nfft + n];
                p = p + creal(b) * creal(b);
                p = p + cimag(b) * cimag(b);
            }
            if (p > Umax) Umax = p;
        }
        double gp = gam[k] * Umax; /* :698 */
        if (gp > Pmax) Pmax = gp;
    }
    double leff = phimax / Pmax; /* :699 */
    double dl = alphalin * leff;  /* :700 */
    double dz_nl;
    if (dl >= 1) {
        dz_nl = dzmax;
    } else {
        double step;
        if (alphalin == 0) step = leff;
        else step = -1 / alphalin * log(1 - dl);
        if (step > dzmax) dz_nl = dzmax;
        else dz_nl = step;
    }
    return dz_nl;
}

/* ------------------------------------------------------------ checkstep --- */
/* fiber.m:718-758 */
int plxo_checkstep(double zprop, double dz, double lcorr, double *dz_miss,
                   int nz_old, double *dzb, int *nmem)
{
    double nz = zprop / lcorr;
    int nzc = (int)ceil(nz);
    int ntrunk;
    if (*dz_miss == 0) {
        *nmem = 0;
        ntrunk = nzc - nz_old;
        double dzlast = dz - lcorr * (ntrunk - 1);
        for (int k = 0; k < ntrunk - 1; k++) dzb[k] = lcorr;
        dzb[ntrunk > 0 ? ntrunk - 1 : 0] = dzlast;
        *dz_miss = lcorr - dzlast;
    } else {
        *nmem = 1;
        ntrunk = nzc - nz_old + 1;
        if (ntrunk == 1) {
            dzb[0] = dz;
            *dz_miss = *dz_miss - dz;
        } else {
            double dzlast = dz - *dz_miss - lcorr * (ntrunk - 2);
            dzb[0] = *dz_miss;
            for (int k = 1; k < ntrunk - 1; k++) dzb[k] = lcorr;
            dzb[ntrunk - 1] = dzlast;
            *dz_miss = lcorr - dzlast;
        }
    }
    return ntrunk;
}

/* ------------------------------------------------------------- lin_step --- */
/* fiber.m:762-773: u = ifft(fft(u).*fastexp(-betat*dz)), column-wise */
void plxo_lin_step(const double *betat, double dz, plxo_c *u, long nfft, int nfc)
{
    for (int k = 0; k < nfc; k++) {
        plxo_c *c = u + (size_t)k * nfft;
        plxo_fft(c, nfft, 0);
        for (long n = 0; n < nfft; n++) {
            double betaxdz = betat[(size_t)k * nfft + n] * dz; /* fiber.m:620 */
            c[n] = cmul(c[n], cexpi(-betaxdz));
        }
        plxo_fft(c, nfft, 1);
    }
}

static double leff_of(double alphalin, double dz)
{ /* fiber.m:787-791, 827-831 */
    if (alphalin == 0) return dz;
    return (1 - exp(-alphalin * dz)) / alphalin;
}

/* -------------------------------------------------------------- nl_step --- */
/* fiber.m:776-804 (gam replicated per column, :584) */
void plxo_nl_step(double alphalin, const double *gam, double dz, plxo_c *u,
                  long nfft, int nfc, int spm, int xpm)
{
    double leff = leff_of(alphalin, dz);
    if (!xpm && !spm) return; /* :800-802 */
    for (long n = 0; n < nfft; n++) {
        double rowsum = 0;
        if (xpm) {
            for (int k = 0; k < nfc; k++) {
                plxo_c a = u[(size_t)k * nfft + n];
                rowsum += creal(a) * creal(a) + cimag(a) * cimag(a);
            }
        }
        for (int k = 0; k < nfc; k++) {
            plxo_c a = u[(size_t)k * nfft + n];
            double pw = creal(a) * creal(a) + cimag(a) * cimag(a); /* :792 */
            if (xpm) {
                if (spm) pw = 2 * rowsum - pw;   /* :795 */
                else pw = 2 * (rowsum - pw);     /* :797 */
            }
            double arg = -gam[k] * pw * leff;    /* :804 */
            u[(size_t)k * nfft + n] = cmul(a, cexpi(arg));
        }
    }
}

/* -------------------------------------------------------- matrix_nl_step --- */
/* fiber.m:807-874 */
int plxo_matrix_nl_step(long nfft, int ismanakov, double alphalin,
                        const double *gam, double dz, plxo_c *ux, plxo_c *uy,
                        int nfc, int spm, int xpm)
{
    double leff = leff_of(alphalin, dz);
    for (int k = 0; k < nfc; k++) {
        plxo_c *cx = ux + (size_t)k * nfft, *cy = uy + (size_t)k * nfft;
        if (spm) {
            double gamleff = gam[k] * leff; /* :836 */
            for (long n = 0; n < nfft; n++) {
                double xr = creal(cx[n]), xi = cimag(cx[n]);
                double yr = creal(cy[n]), yi = cimag(cy[n]);
                double power = xr * xr + xi * xi + yr * yr + yi * yi; /* :834-835 */
                plxo_c nls = cexpi(-gamleff * power);                  /* :837 */
                plxo_c a = cmul(cx[n], nls), b = cmul(cy[n], nls);     /* :838-839 */
                if (!ismanakov) {
                    double s3 = 2 * (creal(a) * cimag(b) - cimag(a) * creal(b)); /* :842 */
                    double ph = gamleff * s3 / 3;                      /* :844 */
                    double cp = cos(ph), sp = sin(ph);
                    plxo_c uxx = rmul(cp, a) + rmul(sp, b);            /* :847 */
                    plxo_c uyy = rmul(-sp, a) + rmul(cp, b);           /* :848 */
                    a = uxx; b = uyy;
                }
                cx[n] = a; cy[n] = b;
            }
        }
        if (xpm) return PLXO_ERR_XPM_CNLSE; /* :854 error(...) */
    }
    return 0;
}

/* ---------------------------------------------------------- matrix_step --- */
/* fiber.m:877-935.  db0/theta/epsilon indexed by 1-based trunk number n. */
void plxo_matrix_step(const double *betat, const double *db1, const double *dzb,
                      int ntrunk, plxo_c *ux, plxo_c *uy, long nfft, int nfc,
                      const double *db0, const double *theta,
                      const double *epsilon, double lcorr, int ntot, int nmem)
{
    for (int c = 0; c < nfc; c++) {
        plxo_fft(ux + (size_t)c * nfft, nfft, 0); /* :904-905 */
        plxo_fft(uy + (size_t)c * nfft, nfft, 0);
    }
    for (int k = 1; k <= ntrunk; k++) {
        int n = ntot + k - nmem; /* :908 */
        double ct = cos(theta[n - 1]), st = sin(theta[n - 1]);
        double ce = cos(epsilon[n - 1]), se = sin(epsilon[n - 1]);
        /* matRth = [ct -st; st ct]; matReps = [ce i*se; i*se ce]; matR = matRth*matReps */
        plxo_c R11 = (ct * ce) + I * (-st * se);
        plxo_c R12 = (-st * ce) + I * (ct * se);
        plxo_c R21 = (st * ce) + I * (ct * se);
        plxo_c R22 = (ct * ce) + I * (st * se);
        double dzk = dzb[k - 1];
        for (int c = 0; c < nfc; c++) {
            plxo_c *cx = ux + (size_t)c * nfft, *cy = uy + (size_t)c * nfft;
            const double *bt = betat + (size_t)c * nfft, *d1 = db1 + (size_t)c * nfft;
            for (long f = 0; f < nfft; f++) {
                plxo_c uux = cmul(conj(R11), cx[f]) + cmul(conj(R21), cy[f]); /* :920 */
                plxo_c uuy = cmul(conj(R12), cx[f]) + cmul(conj(R22), cy[f]); /* :921 */
                double combeta = bt[f] * dzk;                                  /* :924 */
                double deltabeta = 0.5 * (d1[f] + db0[n - 1]) * dzk / lcorr;   /* :925 */
                uux = cmul(cexpi(-(combeta + deltabeta)), uux);                /* :927 */
                uuy = cmul(cexpi(-(combeta - deltabeta)), uuy);                /* :928 */
                cx[f] = cmul(R11, uux) + cmul(R12, uuy);                       /* :931 */
                cy[f] = cmul(R21, uux) + cmul(R22, uuy);                       /* :932 */
            }
        }
    }
    for (int c = 0; c < nfc; c++) {
        plxo_fft(ux + (size_t)c * nfft, nfft, 1); /* :934-935 */
        plxo_fft(uy + (size_t)c * nfft, nfft, 1);
    }
}

static void scale_field(plxo_c *u, size_t n, double s)
{
    for (size_t i = 0; i < n; i++) u[i] = rmul(s, u[i]);
}

/* test diagnostics: the step lengths nextstep returns inside matrix_ssfm (fiber.m:512, :534), in order */
static double *g_dzlog;
static int g_dzcap, g_dzn;
void plxo_set_step_log(double *buf, int cap) { g_dzlog = buf; g_dzcap = buf ? cap : 0; g_dzn = 0; }
int plxo_step_log_count(void) { return g_dzn; }
static void log_dz(double dz) { if (g_dzlog) { if (g_dzn < g_dzcap) g_dzlog[g_dzn] = dz; g_dzn++; } }
/* ... and the reverse: step k (0-based) of the next plxo_matrix_ssfm call takes replay[k] instead of nextstep's result, k < n
 * (parity tests hand the DEVICE's step sequence to the restatement: on noise-loaded fields the step rule itself amplifies
 * rounding differences, so the transforms are compared under one and the same sequence) */
static const double *g_replay;
static int g_replay_n, g_replay_k;
void plxo_set_step_replay(const double *dz, int n) { g_replay = dz; g_replay_n = dz ? n : 0; g_replay_k = 0; }
static double replay_dz(double dz) { const int k = g_replay_k++; return (g_replay && k < g_replay_n) ? g_replay[k] : dz; }

/* ---------------------------------------------------------- matrix_ssfm --- */
/* fiber.m:459-554 */
int plxo_matrix_ssfm(plxo_c *ux, plxo_c *uy, const double *betat,
                     const double *db1, double dzmaxt, double dphimaxt,
                     const double *gam_in, double alphalin, int nfc, long nfft,
                     double Lf, int nplates, int manakov, const int *fls,
                     const double *db0, const double *theta,
                     const double *epsilon, double *firstdz, int *ncycle_out)
{
    double *gam = (double *)malloc(sizeof(double) * nfc);
    for (int k = 0; k < nfc; k++) gam[k] = manakov ? gam_in[k] * 8 / 9 : gam_in[k]; /* :499-501 */
    double *dzb = (double *)malloc(sizeof(double) * (nplates + 4));
    int ncycle = 1;
    double lcorr = Lf / nplates; /* :507 */
    double dz_miss = 0;
    int rc = 0;
    size_t tot = (size_t)nfft * nfc;

    g_replay_k = 0;
    double dz = replay_dz(plxo_nextstep(dzmaxt, dphimaxt, gam, nfc, alphalin, ux, uy, nfft)); /* :512 */
    log_dz(dz);
    double halfalpha = 0.5 * alphalin;
    int ntot = 0;
    *firstdz = dz;
    double zprop = dz;
    while (zprop < Lf) { /* :518 */
        rc = plxo_matrix_nl_step(nfft, manakov, alphalin, gam, dz, ux, uy, nfc, fls[2], fls[3]);
        if (rc) goto out;
        int nmem;
        int ntrunk = plxo_checkstep(zprop, dz, lcorr, &dz_miss, ntot, dzb, &nmem); /* :524 */
        plxo_matrix_step(betat, db1, dzb, ntrunk, ux, uy, nfft, nfc, db0, theta, epsilon,
                         lcorr, ntot, nmem);                                   /* :526 */
        ntot = ntot + ntrunk - nmem;                                           /* :529 */
        double att = exp(-halfalpha * dz);                                     /* :531-532 */
        scale_field(ux, tot, att);
        scale_field(uy, tot, att);
        dz = replay_dz(plxo_nextstep(dzmaxt, dphimaxt, gam, nfc, alphalin, ux, uy, nfft)); /* :534 */
        log_dz(dz);
        zprop = zprop + dz;
        ncycle = ncycle + 1;
    }
    {
        double last_step = Lf - zprop + dz; /* :538 */
        rc = plxo_matrix_nl_step(nfft, manakov, alphalin, gam, last_step, ux, uy, nfc, fls[2], fls[3]);
        if (rc) goto out;
        int nmem;
        int ntrunk = plxo_checkstep(Lf, last_step, lcorr, &dz_miss, ntot, dzb, &nmem); /* :545 */
        plxo_matrix_step(betat, db1, dzb, ntrunk, ux, uy, nfft, nfc, db0, theta, epsilon,
                         lcorr, ntot, nmem);
        double att = exp(-halfalpha * last_step); /* :550-551 */
        scale_field(ux, tot, att);
        scale_field(uy, tot, att);
    }
out:
    *ncycle_out = ncycle;
    free(gam);
    free(dzb);
    return rc;
}

/* ------------------------------------------------------------- adaptssfm --- */
/* fiber.m:938-1009.  One trial step of the adaptive scheme. */
static void adaptssfm(plxo_c *u, double *zdone, double *dz, double alphalin,
                      const double *gam, int nfc, const int *fls,
                      const double *betat, double halfalpha, double trg_err,
                      double trg_safety, int *nrej, int *ncycle, long nfft)
{
    size_t tot = (size_t)nfft * nfc;
    double dz1 = *dz, dz2 = 0.5 * dz1, dz4 = 0.25 * dz1;
    plxo_c *ustack = (plxo_c *)malloc(sizeof(plxo_c) * tot);
    plxo_c *uh = (plxo_c *)malloc(sizeof(plxo_c) * tot);
    memcpy(ustack, u, sizeof(plxo_c) * tot);
    memcpy(uh, u, sizeof(plxo_c) * tot);
    /* one big step (symmetric SSFM), :972-979 */
    plxo_nl_step(alphalin, gam, dz2, u, nfft, nfc, fls[2], fls[3]);
    scale_field(u, tot, exp(-halfalpha * dz2));
    plxo_lin_step(betat, dz1, u, nfft, nfc);
    plxo_nl_step(alphalin, gam, dz2, u, nfft, nfc, fls[2], fls[3]);
    scale_field(u, tot, exp(-halfalpha * dz2));
    /* two small steps, :983-993 */
    plxo_nl_step(alphalin, gam, dz4, uh, nfft, nfc, fls[2], fls[3]);
    scale_field(uh, tot, exp(-halfalpha * dz4));
    plxo_lin_step(betat, dz2, uh, nfft, nfc);
    plxo_nl_step(alphalin, gam, dz2, uh, nfft, nfc, fls[2], fls[3]);
    scale_field(uh, tot, exp(-halfalpha * dz2));
    plxo_lin_step(betat, dz2, uh, nfft, nfc);
    plxo_nl_step(alphalin, gam, dz4, uh, nfft, nfc, fls[2], fls[3]);
    scale_field(uh, tot, exp(-halfalpha * dz4));
    /* :997 */
    double emax = 0;
    for (size_t i = 0; i < tot; i++) {
        double dr = creal(u[i]) - creal(uh[i]), di = cimag(u[i]) - cimag(uh[i]);
        double e = sqrt(dr * dr + di * di);
        if (e > emax) emax = e;
    }
    double est_err = emax / dz1;
    if (est_err > trg_err) { /* reject, :999-1002 */
        *dz = trg_safety * sqrt(trg_err / est_err) * dz1;
        memcpy(u, ustack, sizeof(plxo_c) * tot);
        *nrej = *nrej + 1;
    } else { /* accept, :1003-1008 */
        for (size_t i = 0; i < tot; i++) {
            double c43 = 4.0 / 3, c13 = 1.0 / 3;
            u[i] = (c43 * creal(uh[i]) - c13 * creal(u[i])) + I * (c43 * cimag(uh[i]) - c13 * cimag(u[i]));
        }
        *zdone = *zdone + dz1;
        *dz = trg_safety * sqrt(trg_err / est_err) * dz1;
        *ncycle = *ncycle + 1;
    }
    free(ustack);
    free(uh);
}

/* ----------------------------------------------------------- scalar_ssfm --- */
/* fiber.m:557-636 */
int plxo_scalar_ssfm(plxo_c *u, const double *betat, double dzmaxt,
                     double dphimaxt, const double *gam, double alphalin,
                     long nfft, int nfc, double Lf, const int *fls, int tolflag,
                     double trg_err, double trg_safety, double *firstdz,
                     int *ncycle_out)
{
    size_t tot = (size_t)nfft * nfc;
    double dz = plxo_nextstep(dzmaxt, dphimaxt, gam, nfc, alphalin, u, NULL, nfft); /* :585 */
    double halfalpha = 0.5 * alphalin;
    int ncycle = 1;
    double zprop;
    if (tolflag == 1) { /* :588-611 */
        if (dz >= dzmaxt) {
            double maxpow = -INFINITY;
            for (int k = 0; k < nfc; k++) {
                double Umax = -INFINITY;
                for (long n = 0; n < nfft; n++) {
                    plxo_c a = u[(size_t)k * nfft + n];
                    double p = creal(a) * creal(a) + cimag(a) * cimag(a);
                    if (p > Umax) Umax = p;
                }
                if (gam[k] * Umax > maxpow) maxpow = gam[k] * Umax;
            }
            if (alphalin == 0) dphimaxt = maxpow * dz;
            else dphimaxt = maxpow * (1 - exp(-alphalin * dz)) / alphalin;
        }
        double dzini = dz;
        double zdone = 0;
        int nrej = 0;
        while (zdone == 0) {
            /* called with nrej=0,ncycle=0: the returned ncycle (0 or 1) replaces it, :601-602 */
            int nc = 0;
            nrej = 0;
            adaptssfm(u, &zdone, &dz, alphalin, gam, nfc, fls, betat, halfalpha,
                      trg_err, trg_safety, &nrej, &nc, nfft);
            ncycle = nc;
        }
        if (dz > dzmaxt) dz = dzmaxt;
        dphimaxt = dphimaxt * (1 - exp(-alphalin * zdone)) / (1 - exp(-alphalin * dzini)); /* :607 */
        *firstdz = zdone;
        zprop = zdone + dz;
        ncycle = ncycle + 1;
    } else {
        *firstdz = dz;
        zprop = dz;
    }
    while (zprop < Lf) { /* :616-630 */
        plxo_nl_step(alphalin, gam, dz, u, nfft, nfc, fls[2], fls[3]);
        plxo_lin_step(betat, dz, u, nfft, nfc);
        scale_field(u, tot, exp(-halfalpha * dz));
        dz = plxo_nextstep(dzmaxt, dphimaxt, gam, nfc, alphalin, u, NULL, nfft);
        zprop = zprop + dz;
        ncycle = ncycle + 1;
    }
    double last_step = Lf - zprop + dz; /* :631 */
    plxo_nl_step(alphalin, gam, last_step, u, nfft, nfc, fls[2], fls[3]);
    plxo_lin_step(betat, last_step, u, nfft, nfc);
    scale_field(u, tot, exp(-halfalpha * last_step));
    *ncycle_out = ncycle;
    return 0;
}

/* --------------------------------------------------------- scalar_a_ssfm --- */
/* fiber.m:639-679 */
int plxo_scalar_a_ssfm(plxo_c *u, const double *betat, double dzmaxt,
                       double dphimaxt, const double *gam, double alphalin,
                       long nfft, int nfc, double Lf, double trg_err,
                       double trg_safety, const int *fls, double *firstdz,
                       int *ncycle_out, int *nrej_out)
{
    int ncycle = 1;
    double dz = plxo_nextstep(dzmaxt, dphimaxt, gam, nfc, alphalin, u, NULL, nfft); /* :666 */
    double halfalpha = 0.5 * alphalin;
    *firstdz = dz;
    double zdone = 0;
    int nrej = 0;
    while (zdone < Lf) {
        if (zdone + dz > Lf) dz = Lf - zdone; /* :672-674 */
        adaptssfm(u, &zdone, &dz, alphalin, gam, nfc, fls, betat, halfalpha,
                  trg_err, trg_safety, &nrej, &ncycle, nfft);
        if (dz > dzmaxt) dz = dzmaxt;
    }
    *ncycle_out = ncycle;
    *nrej_out = nrej;
    return 0;
}

/*
 * plxo_rx.c -- CPU ORACLE (test infrastructure, see plxo.h) for the receiver
 * side of the hot path: CDE_OFDE.m, cmaadaptivefilter.c, easiadaptivefilter.c,
 * the DspPdmCohQpsk.m drivers / carrier recovery, samp2pat.m decisions.
 */
#include "plxo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline plxo_c cmul(plxo_c a, plxo_c b)
{
    double ar = creal(a), ai = cimag(a), br = creal(b), bi = cimag(b);
    return (ar * br - ai * bi) + I * (ar * bi + ai * br);
}
static inline plxo_c cexpi(double a) { return cos(a) + I * sin(a); }

/* ===================================================== CDE_OFDE.m:16-47 === */
/* H = exp(H_D + H_S) on fg = df*(-N/2 : N/2-1), CDE_OFDE.m:29-38 */
void plxo_cde_transfer(plxo_c *H, long fftlen, double fs, double lambda_ref,
                       double span, double D, double S)
{
    const double c = 299792458; /* :21 */
    double fc = c / lambda_ref;
    double Tf = (double)fftlen / fs;
    double df = 1 / Tf;
    for (long i = 0; i < fftlen; i++) {
        double fg = df * (double)(i - fftlen / 2);
        /* H_D = -1j*D*span*pi*c / fc^2 * fg.^2 ; H_S = 1j*S*span*pi*c^2 / 3 / fc^4 * fg.^3 */
        double hd = -(D * span * M_PI * c / (fc * fc) * (fg * fg));
        double hs = S * span * M_PI * (c * c) / 3 / (fc * fc * fc * fc) * (fg * fg * fg);
        H[i] = cexpi(hd + hs); /* exp of a purely imaginary argument */
    }
}

/* CDE_OFDE.m:62-125 */
int plxo_overlap_both_trans(const plxo_c *x, long nx, const plxo_c *H, long N,
                            long L, plxo_c *y)
{
    if (N % 2 != 0) return 2;      /* :73-74 H must be even length */
    if (L <= 0) return 3;          /* :77-78 */
    if (L > N) return 4;           /* :79-80 */
    if (nx < N) return 5;          /* :83-84 */
    long m = nx % L;
    long nxL = nx + (m ? L - m : 0); /* :90-94 */
    long B = N - L, B2 = B / 2;      /* :97,101 */
    plxo_c *xe = (plxo_c *)calloc((size_t)(nxL + 2 * B2 + N), sizeof(plxo_c));
    plxo_c *ye = (plxo_c *)calloc((size_t)nxL, sizeof(plxo_c));
    plxo_c *xc = (plxo_c *)malloc(sizeof(plxo_c) * N);
    memcpy(xe + B2, x, sizeof(plxo_c) * nx); /* :102 */
    for (long i = 0; i < nxL; i += L) {       /* :104 */
        memcpy(xc, xe + i, sizeof(plxo_c) * N);
        plxo_fft(xc, N, 0);
        /* Yc = fftshift(fft(xc)).*H ; yc = ifft(ifftshift(Yc)):
         * multiply bin k by H at its fftshift position (k + N/2) mod N */
        for (long k = 0; k < N; k++) xc[k] = cmul(xc[k], H[(k + N / 2) % N]);
        plxo_fft(xc, N, 1);
        for (long j = 0; j < L; j++) ye[i + j] = ye[i + j] + xc[B2 + j]; /* :115 */
    }
    memcpy(y, ye, sizeof(plxo_c) * nx); /* :119 */
    free(xe); free(ye); free(xc);
    return 0;
}

int plxo_cde_ofde(const plxo_c *inx, const plxo_c *iny, long nx, double fs,
                  double lambda_ref, double span, double D, double S,
                  long fftlen, long L, plxo_c *outx, plxo_c *outy)
{
    if (fftlen > nx) fftlen = nx; /* :24-27 */
    plxo_c *H = (plxo_c *)malloc(sizeof(plxo_c) * fftlen);
    plxo_cde_transfer(H, fftlen, fs, lambda_ref, span, D, S);
    int rc = plxo_overlap_both_trans(inx, nx, H, fftlen, L, outx);
    if (!rc) rc = plxo_overlap_both_trans(iny, nx, H, fftlen, L, outy);
    free(H);
    return rc;
}

/* ============================================== cmaadaptivefilter.c:25-91 === */
static double vmac(const double *a, const double *b, int n)
{ /* :25-32 */
    double z = 0;
    for (int i = 0; i < n; i++) z += a[i] * b[i];
    return z;
}

static void updatecoeff(double *hr, double *hi, int Ntap, const double *xr,
                        const double *xi, int Ndim, double outr, double outi,
                        double mu, double R)
{ /* :41-54 */
    double k = mu * (R - outr * outr - outi * outi); /* errorfun :34-38 */
    for (int i = 0; i < Ntap; i++) {
        hr[i] += k * (outr * xr[i] + outi * xi[i]);
        hi[i] += k * (outi * xr[i] - outr * xi[i]);
        hr[i + Ntap] += k * (outr * xr[i + Ndim] + outi * xi[i + Ndim]);
        hi[i + Ntap] += k * (outi * xr[i + Ndim] - outr * xi[i + Ndim]);
    }
}

static void butterfly_out(const double *xr, const double *xi, int Ndim, int i,
                          const double *h1r, const double *h1i,
                          const double *h2r, const double *h2i, int Ntap,
                          double *yr, double *yi, int dimY)
{ /* cmaadaptivefilter.c:71-81 == easiadaptivefilter.c:65-75 */
    yr[i] = vmac(xr + i, h1r, Ntap) - vmac(xi + i, h1i, Ntap)
          + vmac(xr + i + Ndim, h1r + Ntap, Ntap) - vmac(xi + i + Ndim, h1i + Ntap, Ntap);
    yi[i] = vmac(xi + i, h1r, Ntap) + vmac(xr + i, h1i, Ntap)
          + vmac(xi + i + Ndim, h1r + Ntap, Ntap) + vmac(xr + i + Ndim, h1i + Ntap, Ntap);
    yr[i + dimY] = vmac(xr + i, h2r, Ntap) - vmac(xi + i, h2i, Ntap)
                 + vmac(xr + i + Ndim, h2r + Ntap, Ntap) - vmac(xi + i + Ndim, h2i + Ntap, Ntap);
    yi[i + dimY] = vmac(xi + i, h2r, Ntap) + vmac(xr + i, h2i, Ntap)
                 + vmac(xi + i + Ndim, h2r + Ntap, Ntap) + vmac(xr + i + Ndim, h2i + Ntap, Ntap);
}

void plxo_cmafilter(const double *xr, const double *xi, int Ndim, double *h1r,
                    double *h1i, double *h2r, double *h2i, int Ntap, double mu,
                    const double *R, double *yr, double *yi, int dontskip)
{ /* :57-91 */
    int k = ((Ntap - 1) / 2) % 2;
    int dimY = Ndim - Ntap + 1;
    for (int i = 0; i < dimY; i++) {
        butterfly_out(xr, xi, Ndim, i, h1r, h1i, h2r, h2i, Ntap, yr, yi, dimY);
        if (dontskip || (i % 2 == k)) {
            updatecoeff(h1r, h1i, Ntap, xr + i, xi + i, Ndim, yr[i], yi[i], mu, R[0]);
            updatecoeff(h2r, h2i, Ntap, xr + i, xi + i, Ndim, yr[i + dimY], yi[i + dimY], mu, R[1]);
        }
    }
}

/* ============================================= easiadaptivefilter.c:34-93 === */
static double myabs(double n) { return n < 0 ? -n : n; }

void plxo_easifilter(const double *xr, const double *xi, int Ndim, double *h1r,
                     double *h1i, double *h2r, double *h2i, int Ntap, double mu,
                     double *yr, double *yi, int dontskip)
{
    int k = ((Ntap - 1) / 2) % 2;
    int dimY = Ndim - Ntap + 1;
    for (int i = 0; i < dimY; i++) {
        butterfly_out(xr, xi, Ndim, i, h1r, h1i, h2r, h2i, Ntap, yr, yi, dimY);
        if (dontskip || (i % 2 == k)) {
            double a = yr[i], b = yr[i + dimY], M = mu, E[4]; /* :81, errorfun :43-49 */
            E[0] = (a * a - 1) / (1 + M * (a * a + b * b));
            E[1] = (a * b) / (1 + M * (a * a + b * b)) + (a * b * (a * a - b * b)) / (1 + M * (a * myabs(a) + b * myabs(b)));
            E[2] = (a * b) / (1 + M * (a * a + b * b)) + (a * b * (b * b - a * a)) / (1 + M * (a * myabs(a) + b * myabs(b)));
            E[3] = (b * b - 1) / (1 + M * (a * a + b * b));
            double h11 = (1 - mu * E[0]) * h1r[0] + (-mu * E[1]) * h2r[0]; /* :83-90 */
            double h12 = (1 - mu * E[0]) * h1r[1] + (-mu * E[1]) * h2r[1];
            double h21 = (-mu * E[2]) * h1r[0] + (1 - mu * E[3]) * h2r[0];
            double h22 = (-mu * E[2]) * h1r[1] + (1 - mu * E[3]) * h2r[1];
            h1r[0] = h11; h1r[1] = h12; h2r[0] = h21; h2r[1] = h22;
        }
    }
}

int plxo_cma_gateway_check(double Ntap, double sps, int check_odd)
{ /* cmaadaptivefilter.c:118-133; easiadaptivefilter.c has no odd check */
    if (check_odd && ((int)Ntap % 2 == 0)) return 1;
    if ((int)sps != 1 && (int)sps != 2) return 2;
    return 0;
}

/* ============ the .m twins, used when no MEX is compiled ============
 * cmaadaptivefilter.m:52-72: every sample updates (no sps gate), the updated taps are RETURNED.
 * easiadaptivefilter.m:51-84: complex a, b, abs(), ALL taps of the complex h are recombined.
 * Layout: xx [2][Ndim], h1/h2 [2][ntap] (column p of the MATLAB matrix at offset p*rows), y [2][L]. */
static plxo_c colsum2(const plxo_c *xx, int Ndim, int k, const plxo_c *h, int ntap)
{ /* sum(sum(xx(nindex,:).*h)): MATLAB sums each column over the taps first, then the two column sums */
    plxo_c s[2];
    for (int p = 0; p < 2; p++) {
        s[p] = 0;
        for (int t = 0; t < ntap; t++) s[p] = s[p] + cmul(xx[p * Ndim + k + t], h[p * ntap + t]);
    }
    return s[0] + s[1];
}

void plxo_cmafilter_m(const plxo_c *xx, int Ndim, plxo_c *h1, plxo_c *h2, int ntap, double mu,
                      const double *R, plxo_c *y)
{
    int L = Ndim - ntap + 1; /* :54 */
    for (int k = 0; k < L; k++) { /* :60-69 */
        plxo_c Y1 = colsum2(xx, Ndim, k, h1, ntap), Y2 = colsum2(xx, Ndim, k, h2, ntap);
        y[k] = Y1; y[L + k] = Y2;
        double a1 = hypot(creal(Y1), cimag(Y1)), a2 = hypot(creal(Y2), cimag(Y2));
        double m1 = R[0] - a1 * a1, m2 = R[1] - a2 * a2;            /* errorfuncma :74-75: X.*(M - abs(X).^2) */
        plxo_c e1 = (mu * (creal(Y1) * m1)) + I * (mu * (cimag(Y1) * m1));
        plxo_c e2 = (mu * (creal(Y2) * m2)) + I * (mu * (cimag(Y2) * m2));
        for (int p = 0; p < 2; p++)
            for (int t = 0; t < ntap; t++) {
                plxo_c xc = conj(xx[p * Ndim + k + t]);
                h1[p * ntap + t] = h1[p * ntap + t] + cmul(e1, xc);  /* :64-67 */
                h2[p * ntap + t] = h2[p * ntap + t] + cmul(e2, xc);
            }
    }
}

void plxo_easifilter_m(const plxo_c *xx, int Ndim, plxo_c *h1, plxo_c *h2, int ntap, double mu, plxo_c *y)
{
    int L = Ndim - ntap + 1;
    for (int k = 0; k < L; k++) { /* :51-69 */
        plxo_c a = colsum2(xx, Ndim, k, h1, ntap), b = colsum2(xx, Ndim, k, h2, ntap);
        y[k] = a; y[L + k] = b;
        double aa = hypot(creal(a), cimag(a)), ab = hypot(creal(b), cimag(b));
        double den1 = 1 + mu * (aa * aa + ab * ab);                  /* errorfun :78-84, lambda = mu */
        plxo_c den2 = 1 + mu * (a * aa + b * ab);
        plxo_c p = cmul(a, b);
        double E11 = (aa * aa - 1) / den1, E22 = (ab * ab - 1) / den1;
        plxo_c E12 = p / den1 + (p * (aa * aa - ab * ab)) / den2;
        plxo_c E21 = p / den1 + (p * (ab * ab - aa * aa)) / den2;
        for (int t = 0; t < ntap; t++) {                            /* :58-66, all taps, both columns */
            for (int c = 0; c < 2; c++) {
                plxo_c g1 = h1[c * ntap + t], g2 = h2[c * ntap + t];
                h1[c * ntap + t] = (1 - mu * E11) * g1 + cmul(-mu * E12, g2);
                h2[c * ntap + t] = cmul(-mu * E21, g1) + (1 - mu * E22) * g2;
            }
        }
    }
}

/* ================================= DspPdmCohQpsk.m:142-244 pol-demux drivers === */
static int poldemux_driver(const plxo_c *x, long L, const plxo_c *M, int taps,
                           double mu, const double *R, int is_easi, plxo_c *y,
                           plxo_c *h1o, plxo_c *h2o)
{
    int halftaps = taps / 2;           /* :146 */
    int Ndim = (int)L + 2 * halftaps;  /* :161-165 cyclic extension */
    double *xr = (double *)malloc(sizeof(double) * 2 * Ndim), *xi = (double *)malloc(sizeof(double) * 2 * Ndim);
    for (int p = 0; p < 2; p++)
        for (int i = 0; i < Ndim; i++) {
            long src = ((long)i - halftaps) % L;
            if (src < 0) src += L;
            xr[p * Ndim + i] = creal(x[p * L + src]);
            xi[p * Ndim + i] = cimag(x[p * L + src]);
        }
    size_t nh = (size_t)2 * taps;
    double *h1r = (double *)calloc(nh, sizeof(double)), *h1i = (double *)calloc(nh, sizeof(double));
    double *h2r = (double *)calloc(nh, sizeof(double)), *h2i = (double *)calloc(nh, sizeof(double));
    double *o1r = (double *)malloc(sizeof(double) * nh), *o1i = (double *)malloc(sizeof(double) * nh);
    double *o2r = (double *)malloc(sizeof(double) * nh), *o2i = (double *)malloc(sizeof(double) * nh);
    /* hzero(halftaps+1,:,:) = M; h1 = squeeze(hzero(:,1,:)); h2 = squeeze(hzero(:,2,:)) :160-167 */
    for (int j = 0; j < 2; j++) {
        h1r[j * taps + halftaps] = creal(M[0 * 2 + j]); h1i[j * taps + halftaps] = cimag(M[0 * 2 + j]);
        h2r[j * taps + halftaps] = creal(M[1 * 2 + j]); h2i[j * taps + halftaps] = cimag(M[1 * 2 + j]);
    }
    double *yr = (double *)malloc(sizeof(double) * 2 * L), *yi = (double *)malloc(sizeof(double) * 2 * L);
    int convergence = 0, c = 1;
    double rep = (is_easi ? 20.0 : 50.0) * ceil(1.0 / ((double)L * mu)); /* :175 / :227 */
    int passes = 0;
    while (!convergence && (double)c < rep) { /* :176 */
        memcpy(o1r, h1r, sizeof(double) * nh); memcpy(o1i, h1i, sizeof(double) * nh);
        memcpy(o2r, h2r, sizeof(double) * nh); memcpy(o2i, h2i, sizeof(double) * nh);
        /* MEX mutates h1,h2 in place and returns zeros, which the driver keeps (:183-186) */
        if (is_easi == 2) { /* no MEX: the .m twin returns the updated (complex) taps, which the driver takes (:232-235) */
            plxo_c *xx = (plxo_c *)malloc(sizeof(plxo_c) * 2 * Ndim), *yy = (plxo_c *)malloc(sizeof(plxo_c) * 2 * L);
            plxo_c g1[2], g2[2];
            for (int i = 0; i < 2 * Ndim; i++) xx[i] = xr[i] + I * xi[i];
            for (int j = 0; j < 2; j++) { g1[j] = h1r[j] + I * h1i[j]; g2[j] = h2r[j] + I * h2i[j]; }
            plxo_easifilter_m(xx, Ndim, g1, g2, 1, mu, yy);
            for (int j = 0; j < 2; j++) { h1r[j] = creal(g1[j]); h1i[j] = cimag(g1[j]); h2r[j] = creal(g2[j]); h2i[j] = cimag(g2[j]); }
            for (long i = 0; i < 2 * L; i++) { yr[i] = creal(yy[i]); yi[i] = cimag(yy[i]); }
            free(xx); free(yy);
        }
        else if (is_easi) plxo_easifilter(xr, xi, Ndim, h1r, h1i, h2r, h2i, taps, mu, yr, yi, 1);
        else plxo_cmafilter(xr, xi, Ndim, h1r, h1i, h2r, h2i, taps, mu, R, yr, yi, 1);
        double mx = 0; /* :187 max(max(abs([h1_old-h1 h2_old-h2]))) */
        for (size_t t = 0; t < nh; t++) {
            double d1 = hypot(o1r[t] - h1r[t], o1i[t] - h1i[t]);
            double d2 = hypot(o2r[t] - h2r[t], o2i[t] - h2i[t]);
            if (d1 > mx) mx = d1;
            if (d2 > mx) mx = d2;
        }
        if (mx < 5e-5) convergence = 1;
        c = c + 1;
        passes++;
    }
    for (long i = 0; i < 2 * L; i++) y[i] = passes ? yr[i] + I * yi[i] : 0;
    for (size_t t = 0; t < nh; t++) { h1o[t] = h1r[t] + I * h1i[t]; h2o[t] = h2r[t] + I * h2i[t]; }
    free(xr); free(xi); free(h1r); free(h1i); free(h2r); free(h2i);
    free(o1r); free(o1i); free(o2r); free(o2i); free(yr); free(yi);
    return passes;
}

int plxo_cmapolardemux(const plxo_c *x, long L, const plxo_c *M, int taps,
                       double mu, const double *R, plxo_c *y, plxo_c *h1, plxo_c *h2)
{
    return poldemux_driver(x, L, M, taps, mu, R, 0, y, h1, h2);
}

int plxo_easipolardemux(const plxo_c *x, long L, const plxo_c *M, double mu,
                        plxo_c *y, plxo_c *h1, plxo_c *h2)
{
    return poldemux_driver(x, L, M, 1, mu, NULL, 1, y, h1, h2);
}

int plxo_easipolardemux_m(const plxo_c *x, long L, const plxo_c *M, double mu,
                          plxo_c *y, plxo_c *h1, plxo_c *h2)
{ /* easipolardemux around the .m twin of the filter */
    return poldemux_driver(x, L, M, 1, mu, NULL, 2, y, h1, h2);
}

/* ============================================ helpers: fastshift, nmod, unwrap === */
long plxo_nmod(long A, long N)
{ /* nmod.m:31: mod(A-1-N,N)+1 with MATLAB's floored mod */
    long r = (A - 1 - N) % N;
    if (r < 0) r += N;
    return r + 1;
}

void plxo_fastshift(const plxo_c *x, long L, int ncol, long n, plxo_c *y)
{ /* fastshift.m:52-60: y(i) = x(i-n) circularly (n>0 shifts down) */
    for (int c = 0; c < ncol; c++)
        for (long i = 0; i < L; i++) {
            long src = (i - n) % L;
            if (src < 0) src += L;
            y[c * L + i] = x[c * L + src];
        }
}

void plxo_unwrap(double *p, long n)
{ /* MATLAB unwrap (default tolerance pi), column vector */
    double cum = 0;
    double prev = n > 0 ? p[0] : 0;
    for (long i = 1; i < n; i++) {
        double cur = p[i];
        double dp = cur - prev;
        double dps = fmod(dp + M_PI, 2 * M_PI);
        if (dps < 0) dps += 2 * M_PI; /* floored mod */
        dps -= M_PI;
        if (dps == -M_PI && dp > 0) dps = M_PI;
        double corr = dps - dp;
        if (fabs(dp) < M_PI) corr = 0;
        cum += corr;
        prev = cur;
        p[i] = cur + cum;
    }
}

/* ================================================= DspPdmCohQpsk.m:97-123 === */
static plxo_c cpow_int(plxo_c s, int P)
{ /* s.^P for small positive integer P by repeated multiplication */
    plxo_c r = s;
    for (int i = 1; i < P; i++) r = cmul(r, s);
    return r;
}

void plxo_vitvit(const plxo_c *s_in, long L, int ncol, int P, int M, int k,
                 int applyunwrap, double *theta)
{
    plxo_c *s = (plxo_c *)malloc(sizeof(plxo_c) * L * ncol);
    for (long i = 0; i < L * ncol; i++) {
        if (P == M) {
            s[i] = cpow_int(s_in[i], P); /* :101-102 */
        } else {
            plxo_c sm = cpow_int(s_in[i], M); /* :104 abs(s).^P .* fastexp(angle(s.^M)) */
            double a = pow(hypot(creal(s_in[i]), cimag(s_in[i])), (double)P);
            double ang = atan2(cimag(sm), creal(sm));
            s[i] = (a * cos(ang)) + I * (a * sin(ang));
        }
    }
    if (k > 0) { /* :106-118 */
        long N = 2 * (long)k + 1;
        long reps = (N < L) ? 1 : (N + L - 1) / L; /* ceil(N/L) */
        long LL = reps * L;
        plxo_c *filt = (plxo_c *)calloc((size_t)LL, sizeof(plxo_c));
        for (long i = 0; i < N; i++) filt[i] = 1.0 / (double)N;
        plxo_fft(filt, LL, 0);
        plxo_c *buf = (plxo_c *)malloc(sizeof(plxo_c) * LL);
        for (int c = 0; c < ncol; c++) {
            for (long r = 0; r < reps; r++) memcpy(buf + r * L, s + c * L, sizeof(plxo_c) * L);
            plxo_fft(buf, LL, 0);
            for (long i = 0; i < LL; i++) buf[i] = cmul(buf[i], filt[i]);
            plxo_fft(buf, LL, 1);
            memcpy(s + c * L, buf, sizeof(plxo_c) * L);
        }
        free(filt); free(buf);
    }
    for (int c = 0; c < ncol; c++) {
        double *t = theta + c * L;
        for (long i = 0; i < L; i++) t[i] = atan2(cimag(s[c * L + i]), creal(s[c * L + i]));
        if (applyunwrap) plxo_unwrap(t, L); /* :119-123 */
        for (long i = 0; i < L; i++) t[i] = t[i] / M;
    }
    free(s);
}

/* rotpolar, DspPdmCohQpsk.m:126-139: fills 2x2 row-major M */
static void rotpolar_matrix(plxo_c r, plxo_c *Mout)
{
    double m, delta, alpha;
    if (cabs(r) < 0.5) {
        m = cabs(r); delta = carg(r); alpha = m * m / (m * m + 1);
    } else {
        plxo_c ir = 1.0 / r;
        m = cabs(ir); delta = -carg(ir); alpha = 1 / (m * m + 1);
    }
    plxo_c e = cexpi(-delta);
    Mout[0] = sqrt(alpha) * e;  Mout[1] = -sqrt(1 - alpha) * e;
    Mout[2] = sqrt(1 - alpha);  Mout[3] = sqrt(alpha);
}

static plxo_c mean_ratio(const plxo_c *x, long L)
{
    double sr = 0, si = 0;
    for (long i = 0; i < L; i++) { plxo_c q = x[i] / x[L + i]; sr += creal(q); si += cimag(q); }
    return (sr / L) + I * (si / L);
}

/* initial centre-tap matrix of cmapolardemux/easipolardemux, :148-159 / :201-212 */
static void init_matrix(const plxo_c *x, long L, int txpolars, double phizero, plxo_c *M, int has_mat, const double *mat)
{
    if (has_mat) { /* M = params.mat  :148-149 */
        for (int k = 0; k < 4; k++) M[k] = mat[2 * k] + I * mat[2 * k + 1];
    } else if (txpolars != 2) {
        plxo_c Mr[4];
        rotpolar_matrix(mean_ratio(x, L), Mr); /* M = rotpolar(1,r).' */
        M[0] = Mr[0]; M[1] = Mr[2]; M[2] = Mr[1]; M[3] = Mr[3];
    } else {
        M[0] = cos(phizero); M[1] = sin(phizero); M[2] = -sin(phizero); M[3] = cos(phizero);
    }
}

/* ==================================================== DspPdmCohQpsk.m:3-84 === */
long plxo_dsp_pdm_coh_qpsk(const plxo_c *in, long Lin, int ncol,
                           const plxo_dsp_params *p, plxo_c *out)
{
    long L = p->workatbaudrate ? Lin : (Lin + 1) / 2; /* :12-14, 1:2:end */
    plxo_c *s = (plxo_c *)malloc(sizeof(plxo_c) * L * ncol);
    plxo_c *t = (plxo_c *)malloc(sizeof(plxo_c) * L * ncol);
    for (int c = 0; c < ncol; c++)
        for (long i = 0; i < L; i++) s[c * L + i] = in[c * Lin + (p->workatbaudrate ? i : 2 * i)];
    if (p->applynlr) { /* NLRotation :87-94 */
        double mean = 0;
        double *asq = (double *)malloc(sizeof(double) * L);
        for (long i = 0; i < L; i++) {
            double a = 0;
            for (int c = 0; c < ncol; c++) { double m = hypot(creal(s[c * L + i]), cimag(s[c * L + i])); a += m * m; }
            asq[i] = a; mean += a;
        }
        mean /= (double)L;
        for (long i = 0; i < L; i++)
            for (int c = 0; c < ncol; c++) {
                plxo_c v = s[c * L + i];
                double ph = atan2(cimag(v), creal(v)) + p->nlralpha * (asq[i] - mean);
                double am = hypot(creal(v), cimag(v));
                s[c * L + i] = (am * cos(ph)) + I * (am * sin(ph));
            }
        free(asq);
    }
    double peak = 4 * sqrt(p->power_mw); /* :22-23 */
    for (long i = 0; i < L * ncol; i++) s[i] = (creal(s[i]) / peak) + I * (cimag(s[i]) / peak);

    if (p->applypol && ncol == 2) { /* :26-42 */
        plxo_c M[4], h1[64], h2[64];
        if (p->polmethod == 0) { /* singlepol: y = x*M, rotpolar */
            rotpolar_matrix(mean_ratio(s, L), M);
            for (long i = 0; i < L; i++) {
                plxo_c a = s[i], b = s[L + i];
                t[i] = cmul(a, M[0]) + cmul(b, M[2]);
                t[L + i] = cmul(a, M[1]) + cmul(b, M[3]);
            }
            memcpy(s, t, sizeof(plxo_c) * 2 * L);
        }
        if (p->polmethod == 2 || p->polmethod == 3) {
            init_matrix(s, L, p->easi_txpolars, p->easi_phizero, M, p->easi_has_mat, p->easi_mat);
            if (p->mfile_twins) plxo_easipolardemux_m(s, L, M, p->easi_mu, t, h1, h2);
            else plxo_easipolardemux(s, L, M, p->easi_mu, t, h1, h2);
            memcpy(s, t, sizeof(plxo_c) * 2 * L);
        }
        if (p->polmethod == 1 || p->polmethod == 3) {
            if (p->cma_taps > 31) { free(s); free(t); return -1; }
            init_matrix(s, L, p->cma_txpolars, p->cma_phizero, M, p->cma_has_mat, p->cma_mat);
            plxo_cmapolardemux(s, L, M, p->cma_taps, p->cma_mu, p->cma_R, t, h1, h2);
            memcpy(s, t, sizeof(plxo_c) * 2 * L);
        }
    }

    /* carrier phase estimation :44-79 */
    int Mo = 1 << p->modorder;
    double off = p->modorder > 1 ? M_PI / 4 : 0;
    double *omega = (double *)calloc((size_t)L * ncol, sizeof(double));
    double *theta = (double *)malloc(sizeof(double) * L * ncol);
    if (p->freqavg) {
        plxo_fastshift(s, L, ncol, 1, t);
        for (long i = 0; i < L * ncol; i++) t[i] = cmul(s[i], conj(t[i])); /* :49 */
        plxo_vitvit(t, L, ncol, Mo, Mo, p->freqavg, 0, omega);
        for (int c = 0; c < ncol; c++) {
            double *w = omega + c * L;
            for (long i = 1; i < L; i++) w[i] = w[i - 1] + w[i]; /* cumsum */
            double w1 = w[0], wend = w[L - 1];
            double closest = w1 + round((wend - w1) / 2 / M_PI) * 2 * M_PI; /* :52 */
            double ratio = closest / wend;                                    /* :53 */
            for (long i = 0; i < L; i++) w[i] = ((w[i] - w1) * ratio) + w1;   /* :54-55 */
        }
        for (long i = 0; i < L * ncol; i++) t[i] = cmul(s[i], cexpi(-omega[i])); /* :57 */
        plxo_vitvit(t, L, ncol, p->poworder, Mo, p->phasavg, 1, theta);
        for (long i = 0; i < L * ncol; i++)
            out[i] = cmul(s[i], cexpi(-omega[i] - theta[i] + off)); /* :67,79 */
    } else {
        plxo_vitvit(s, L, ncol, p->poworder, Mo, p->phasavg, 1, theta);
        for (long i = 0; i < L * ncol; i++) out[i] = cmul(s[i], cexpi(-theta[i] + off)); /* :77 */
    }
    free(omega); free(theta); free(s); free(t);
    return L;
}

/* ====================================================== samp2pat.m:61-66 === */
void plxo_samp2pat_coherent(const double *phase, long L, int ncol, unsigned char *pat)
{
    for (int c = 0; c < ncol; c++)
        for (long i = 0; i < L; i++) {
            double v = phase[c * L + i];
            pat[(2 * c) * L + i] = fabs(v) <= M_PI / 2;   /* first_bit */
            pat[(2 * c + 1) * L + i] = v > 0;             /* second_bit */
        }
}

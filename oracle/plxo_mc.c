/*
 * plxo_mc.c -- CPU ORACLE (test infrastructure, see plxo.h) for the Monte-Carlo
 * estimators: ber_estimate.m:97-143 (mc_run) and mc_estimate.m:133-212
 * (complete_mc).  The MATLAB `persistent` variables live in plxo_mc_state.
 */
#include "plxo.h"
#include <math.h>
#include <string.h>

/* erfcinv by Newton iteration on libm erfc (MATLAB builtin, not in the reference) */
double plxo_erfcinv(double y)
{
    if (y <= 0) return INFINITY;
    if (y >= 2) return -INFINITY;
    if (y == 1) return 0;
    double yy = y < 1 ? y : 2 - y;
    /* initial guess: asymptotic */
    double t = sqrt(-2 * log(yy / 2));
    double x = -0.70711 * ((2.30753 + t * 0.27061) / (1 + t * (0.99229 + t * 0.04481)) - t);
    for (int it = 0; it < 60; it++) {
        double err = erfc(x) - yy;
        double dx = err / (-2 / sqrt(M_PI) * exp(-x * x));
        /* Halley correction */
        double step = dx / (1 + x * dx);
        x -= step;
        if (fabs(step) <= 1e-17 * fabs(x)) break;
    }
    return y < 1 ? x : -x;
}

static void mc_init(plxo_mc_state *st, int dim, int has_stop, double stop2, int two)
{
    st->dim = dim;
    for (int i = 0; i < dim; i++) {
        st->n[i] = 1; st->avg[i] = 0; st->var[i] = 0;
        st->varlim[0][i] = 0; st->varlim[1][i] = 0; st->cond[i] = 1;
    }
    st->first = 1;
    st->epsilon[0] = st->epsilon[1] = 0;
    if (has_stop) {
        st->epsilon[0] = sqrt(2.0) * plxo_erfcinv(1 - stop2 / 100);          /* ber_estimate.m:113 */
        if (two) st->epsilon[1] = sqrt(2.0) * plxo_erfcinv(1 + stop2 / 100); /* mc_estimate.m:155 */
    }
}

static int any_cond(const plxo_mc_state *st)
{
    for (int i = 0; i < st->dim; i++) if (st->cond[i]) return 1;
    return 0;
}

/* ber_estimate.m:97-143 */
void plxo_ber_estimate(plxo_mc_state *st, double err, double M, int dim,
                       int nind, int has_stop, double stop1, double stop2,
                       double nmin, int *cond, double *avgber, double *nruns,
                       double *stdber)
{
    int k = nind - 1;
    if (!st->first) mc_init(st, dim, has_stop, stop2, 0); /* :106-116 */
    double nnew = st->n[k] * M;                            /* :117 */
    double N = (st->n[k] - 1) * M;                         /* :121 */
    double varerr = (err - err * err / M) / (M - 1);       /* :122 */
    double avgerr = err / M;                               /* :123 */
    st->var[k] = ((N - 1) * st->var[k] + (M - 1) * varerr +
                  N * M / (N + M) * (st->avg[k] - avgerr) * (st->avg[k] - avgerr)) / (N + M - 1); /* :125-126 */
    st->avg[k] = ((st->n[k] - 1) * st->avg[k] + avgerr) / st->n[k];                               /* :127 */
    for (int i = 0; i < dim; i++) {
        double Ni = (st->n[i] - 1) * M;
        stdber[i] = sqrt(st->var[i] / (Ni + M)); /* :128 */
    }
    int clear = 0;
    if (has_stop) { /* :129-135 */
        if ((st->epsilon[0] * stdber[k] < stop1 * st->avg[k]) && (st->avg[k] * nnew >= nmin)) {
            st->cond[k] = 0;
            if (!any_cond(st)) clear = 1;
        }
    } else { /* :136-141 */
        if (st->avg[k] * nnew > nmin) {
            st->cond[k] = 0;
            if (!any_cond(st)) clear = 1;
        }
    }
    st->n[k] = st->n[k] + 1; /* :142 */
    for (int i = 0; i < dim; i++) {
        nruns[i] = (st->n[i] - 1) * M;
        avgber[i] = st->avg[i];
        cond[i] = st->cond[i];
    }
    if (clear) st->first = 0; /* first = [] */
}

/* mc_estimate.m:133-212 (vector s: nind2 = 1) */
void plxo_mc_estimate(plxo_mc_state *st, const double *s, long M, int dim,
                      int nind, int has_stop, double stop1, double stop2,
                      double nmin, int method_var, int *cond, double *mean,
                      double *var, double *nruns, double *stdmean, double *varlim)
{
    int k = nind - 1;
    if (!st->first) mc_init(st, dim, has_stop, stop2, 1); /* :145-159 */
    double runs = st->n[k] * (double)M;                    /* :160 */
    double N = (st->n[k] - 1) * (double)M;                 /* :162 */
    if (M == 1 && st->n[k] == 1) {                         /* :163-165 */
        st->var[k] = 0;
        st->avg[k] = s[0];
    } else {
        double sum = 0;
        for (long i = 0; i < M; i++) sum += s[i];
        double avgblk = sum / (double)M;                   /* :168 */
        double ss = 0;
        for (long i = 0; i < M; i++) ss += (s[i] - avgblk) * (s[i] - avgblk);
        double varblk = M > 1 ? ss / (double)(M - 1) : 0;  /* var(s), :167 */
        st->var[k] = ((N - 1) * st->var[k] + ((double)M - 1) * varblk +
                      N * M / (N + M) * (st->avg[k] - avgblk) * (st->avg[k] - avgblk)) / (N + M - 1); /* :170-171 */
        st->avg[k] = ((st->n[k] - 1) * st->avg[k] + avgblk) / st->n[k];                               /* :172 */
    }
    for (int i = 0; i < dim; i++) {
        double Ni = (st->n[i] - 1) * (double)M;
        stdmean[i] = sqrt(st->var[i] / (Ni + M)); /* :174 */
    }
    double r1 = st->epsilon[0] + sqrt(2 * (N + M) - 3), r2 = st->epsilon[1] + sqrt(2 * (N + M) - 3);
    double x21mdh = 0.5 * r1 * r1; /* :175 */
    double x2dh = 0.5 * r2 * r2;   /* :176 */
    st->varlim[0][k] = (N + M - 1) * st->var[k] / x21mdh; /* :177 */
    st->varlim[1][k] = (N + M - 1) * st->var[k] / x2dh;   /* :178 */
    int clear = 0;
    if (has_stop) {
        if (!method_var) { /* :182-188 */
            double absavg = fabs(st->avg[k]);
            if ((st->epsilon[0] * stdmean[k] < stop1 * absavg) && (runs >= nmin)) {
                st->cond[k] = 0;
                if (!any_cond(st)) clear = 1;
            }
        } else { /* :189-195 */
            if ((st->varlim[1][k] - st->varlim[0][k]) / st->var[k] < stop1 && (runs >= nmin)) {
                st->cond[k] = 0;
                if (!any_cond(st)) clear = 1;
            }
        }
    } else { /* :197-202 */
        if (runs > nmin) {
            st->cond[k] = 0;
            if (!any_cond(st)) clear = 1;
        }
    }
    st->n[k] = st->n[k] + 1; /* :203 */
    for (int i = 0; i < dim; i++) {
        nruns[i] = (st->n[i] - 1) * (double)M;
        mean[i] = st->avg[i];
        var[i] = st->var[i];
        cond[i] = st->cond[i];
        varlim[2 * i] = st->varlim[0][i];
        varlim[2 * i + 1] = st->varlim[1][i];
    }
    if (clear) st->first = 0;
}

/* Runtime of the mex.h test double (see mex.h). */
#include "mex.h"
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static jmp_buf g_jmp;
static char g_err[512];
mxArray *mxCreateDoubleMatrix(size_t m, size_t n, mxComplexity c)
{
    mxArray *a = (mxArray *)calloc(1, sizeof(mxArray));
    a->m = m; a->n = n;
    const size_t cnt = (m * n) > 0 ? m * n : 1;
    a->pr = (double *)calloc(cnt, sizeof(double));
    a->pi = c == mxCOMPLEX ? (double *)calloc(cnt, sizeof(double)) : NULL;
    return a;
}
mxArray *mxCreateString(const char *str)
{
    mxArray *a = mxCreateDoubleMatrix(1, strlen(str), mxREAL);
    a->str = (char *)malloc(strlen(str) + 1);
    strcpy(a->str, str);
    return a;
}
int mxIsChar(const mxArray *a) { return a && a->str != NULL; }
int mxGetString(const mxArray *a, char *buf, size_t buflen)
{
    if (!a || !a->str || buflen == 0) return 1;
    snprintf(buf, buflen, "%s", a->str);
    return strlen(a->str) >= buflen;
}
mxArray *mxCreateDoubleScalar(double v) { mxArray *a = mxCreateDoubleMatrix(1, 1, mxREAL); a->pr[0] = v; return a; }
mxArray *mxDuplicateArray(const mxArray *s)
{
    mxArray *a = mxCreateDoubleMatrix(s->m, s->n, s->pi ? mxCOMPLEX : mxREAL);
    memcpy(a->pr, s->pr, s->m * s->n * sizeof(double));
    if (s->pi) memcpy(a->pi, s->pi, s->m * s->n * sizeof(double));
    return a;
}
void mxDestroyArray(mxArray *a) { if (a) { free(a->pr); free(a->pi); free(a->str); free(a); } }
double *mxGetPr(const mxArray *a) { return a->pr; }
double *mxGetPi(const mxArray *a) { return a->pi; }
void mxSetPi(mxArray *a, double *pi) { a->pi = pi; }
size_t mxGetM(const mxArray *a) { return a->m; }
size_t mxGetN(const mxArray *a) { return a->n; }
size_t mxGetNumberOfElements(const mxArray *a) { return a->m * a->n; }
int mxIsEmpty(const mxArray *a) { return a == NULL || a->m * a->n == 0; }
double mxGetScalar(const mxArray *a) { return a->pr[0]; }
void *mxCalloc(size_t n, size_t size) { return calloc(n ? n : 1, size); }   /* MATLAB frees these at gateway exit; the tests leak them */
void mexErrMsgTxt(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); longjmp(g_jmp, 1); }
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    char b[400]; vsnprintf(b, sizeof(b), fmt, ap); va_end(ap);
    snprintf(g_err, sizeof(g_err), "%s: %s", id ? id : "", b);
    longjmp(g_jmp, 1);
}
void mexPrintf(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); }
int mexstub_call(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    g_err[0] = 0;
    if (setjmp(g_jmp)) return 1;
    mexFunction(nlhs, plhs, nrhs, prhs);
    return 0;
}
const char *mexstub_last_error(void) { return g_err; }
static int g_locks;
static void (*g_atexit)(void);
void mexLock(void) { g_locks++; }
void mexUnlock(void) { if (g_locks > 0) g_locks--; }
int mexIsLocked(void) { return g_locks > 0; }
int mexAtExit(void (*fn)(void)) { g_atexit = fn; return 0; }
int mexstub_lock_count(void) { return g_locks; }
int mexstub_run_atexit(void)
{
    if (!g_atexit) return 0;
    g_atexit();
    return 1;
}

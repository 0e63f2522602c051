/* TEST DOUBLE of MATLAB's <mex.h> (separate-complex API, -R2017b semantics) -- just enough of the interface to
 * compile and exercise THIS repository's MEX shims (integration/mex, the .c files) without MATLAB.  It is never used to build
 * anything under /root/reference.  Arrays are column-major doubles with separate real / imaginary planes. */
#ifndef PLX_MEXSTUB_H
#define PLX_MEXSTUB_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag { size_t m, n; double *pr, *pi; char *str; /* non-NULL: a char row vector */ } mxArray;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
mxArray *mxCreateDoubleMatrix(size_t m, size_t n, mxComplexity c);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateString(const char *s);
int mxIsChar(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, size_t buflen);   /* 0 on success, 1 if truncated / not a char array */
mxArray *mxDuplicateArray(const mxArray *a);
void mxDestroyArray(mxArray *a);
double *mxGetPr(const mxArray *a);
double *mxGetPi(const mxArray *a);
void mxSetPi(mxArray *a, double *pi);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
int mxIsEmpty(const mxArray *a);
double mxGetScalar(const mxArray *a);
void *mxCalloc(size_t n, size_t size);
void mexErrMsgTxt(const char *msg);                       /* does not return (long jump to the caller of the gateway) */
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
void mexPrintf(const char *fmt, ...);
void mexLock(void);                                       /* keep the MEX file (and the library state behind it) in memory */
void mexUnlock(void);
int mexIsLocked(void);
int mexAtExit(void (*fn)(void));                          /* called when the MEX file is cleared / MATLAB exits */
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
/* harness side (tests/test_mex_shims.py): run a gateway, catching mexErrMsgTxt; returns 0 or 1 and the message */
int mexstub_call(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
const char *mexstub_last_error(void);
int mexstub_lock_count(void);                             /* mexLock() calls minus mexUnlock() calls */
int mexstub_run_atexit(void);                             /* what `clear mex` does: run the registered function; 1 if there was one */
#ifdef __cplusplus
}
#endif
#endif

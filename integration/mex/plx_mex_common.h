/* Shared by every MEX shim of this directory: the library behind the gateway calls keeps its plans, device buffers and
 * pinned staging memory between calls (include/polmux_hip.h, tier A), so the MEX file must stay loaded while that state
 * is alive and must release it when MATLAB clears the MEX file or exits ("Ownership"):
 * mexLock() once -- after the first call's arguments have passed their checks --, plx_release_all() from mexAtExit().
 * A locked MEX file cannot be cleared, so every shim also answers  <name>('release') : it frees the library's state and
 * unlocks the file; `clear mex` then unloads it (to rebuild or replace a shim without restarting MATLAB), and the next
 * ordinary call locks it again.  The reference's MEX files are stateless and clearable (cmaadaptivefilter.c:93-174). */
#ifndef PLX_MEX_COMMON_H
#define PLX_MEX_COMMON_H
#include <string.h>
#include "mex.h"
#include "polmux_hip.h"
static int plx_mex_locked;
static void plx_mex_cleanup(void) { plx_release_all(); }
/* keep the file loaded from now on (call when the arguments are known to be good) */
static void plx_mex_once(void)
{
    if (plx_mex_locked) return;
    mexLock();
    mexAtExit(plx_mex_cleanup);
    plx_mex_locked = 1;
}
/* <name>('release'): 1 if this call was the release request (handled), 0 otherwise */
static int plx_mex_release_request(int nrhs, const mxArray *prhs[])
{
    char cmd[16];
    if (nrhs != 1 || !mxIsChar(prhs[0]) || mxGetString(prhs[0], cmd, sizeof(cmd)) || strcmp(cmd, "release")) return 0;
    plx_release_all();
    if (plx_mex_locked) { mexUnlock(); plx_mex_locked = 0; }
    return 1;
}
#endif

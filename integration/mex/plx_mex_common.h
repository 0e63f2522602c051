/* Shared by every MEX shim of this directory: the library behind the gateway calls keeps its plans, device buffers and
 * pinned staging memory between calls (include/polmux_hip.h, tier A), so the MEX file must stay loaded while that state
 * is alive and must release it when MATLAB clears the MEX file or exits ("Ownership"):
 * mexLock() once, plx_release_all() from mexAtExit(). */
#ifndef PLX_MEX_COMMON_H
#define PLX_MEX_COMMON_H
#include "mex.h"
#include "polmux_hip.h"
static void plx_mex_cleanup(void) { plx_release_all(); }
static void plx_mex_once(void)
{
    static int done;
    if (done) return;
    mexLock();
    mexAtExit(plx_mex_cleanup);
    done = 1;
}
#endif

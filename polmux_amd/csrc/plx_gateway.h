// plx_gateway.h -- library-owned persistent state behind the MEX-shaped gateway calls (library-internal).
//
// The unchanged MATLAB drivers call a gateway once per pass -- cmaadaptivefilter up to 299 times per frame
// (DspPdmCohQpsk.m:176-191), fiber's propagator once per span (fiber.m:372-389) -- so a gateway must not build and
// destroy its plan, its device buffers and its staging memory on every call ("Ownership": plans, twiddle
// tables and buffers belong to the library, pinned with mexLock, released from mexAtExit).  This module keeps, per
// process, behind one mutex:
//   * growable device scratch slots and pinned host staging slots (never shrunk, reused by every gateway);
//   * small LRU caches of the plans the gateways need, keyed by a hash of the descriptor's scalars AND the contents of
//     the tables it points to (betat, db1, gam; the CDE transfer function is a function of its scalars);
//   * counters (allocations, plan builds / hits) so that tests can assert "the second call allocates nothing".
// plx_release_all() frees everything; the MEX shims register it with mexAtExit after mexLock.
#pragma once
#include "plx_internal.h"

#include <mutex>

namespace plxgw {

enum Slot { S_IN = 0, S_OUT = 1, S_AUX = 2, S_AUX2 = 3, S_COUNT = 4 };

struct Stats {
    int64_t calls, dev_allocs, host_allocs, plan_builds, plan_hits, dev_bytes, host_bytes, releases, fallbacks;
};

PLX_HIDDEN std::mutex &mutex();
// device / pinned-host scratch of at least `bytes` (nullptr + error message on failure); contents are NOT preserved
// across a growth.  Valid until the next call for the same slot or plx_release_all().
PLX_HIDDEN void *dev(Slot s, size_t bytes);
PLX_HIDDEN void *pinned(Slot s, size_t bytes);
PLX_HIDDEN uint64_t hash_bytes(const void *p, size_t n, uint64_t seed);
PLX_HIDDEN void count_call();
PLX_HIDDEN void count_plan(bool built);
PLX_HIDDEN void count_fallback();

// cached plans (owned by the cache; never destroy them).  nullptr on failure, *rc holds the code.
PLX_HIDDEN plx_ssfm *ssfm_plan(const plx_ssfm_desc &d, int *rc);
PLX_HIDDEN plx_cde *cde_plan(int64_t fft_len, int64_t L, const double *H_interleaved, int *rc);
PLX_HIDDEN plx_front *front_plan(const plx_front_desc &d, int *rc);

} // namespace plxgw

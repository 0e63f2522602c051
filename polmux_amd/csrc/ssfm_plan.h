// ssfm_plan.h -- the propagator's plan object (library-internal).
#pragma once
#include "ssfm_kernels.h"
#include <vector>

using plxs::FrameCtl;
using plxs::SsfmArgs;

struct plx_ssfm {
    plx_ssfm_desc d;
    int p, p1, p2;
    size_t N;
    SsfmArgs a;
    double *d_betat = nullptr, *d_db1 = nullptr, *d_gam = nullptr, *d_brf = nullptr, *d_psum = nullptr;
    cplx *d_tpass = nullptr, *d_tw1 = nullptr, *d_tw2 = nullptr, *d_ctab = nullptr;
    FrameCtl *d_ctl = nullptr;
    unsigned long long *d_umax = nullptr;
    int *d_ndone = nullptr;   // [0] frames done, [1] abort word, [2] frames in the active list, [3] its running sum over the steps
    int *h_ndone = nullptr;   // pinned copy of the four words
    int *d_active = nullptr;  // [max_frames] active list (k_compact)
    hipEvent_t ev = nullptr;  // completion of the last read-back of d_ndone
    std::vector<FrameCtl> h_ctl;
    int brf_sets = 0;
    size_t lds_col = 0, lds_row = 0;
    cplx *d_e1 = nullptr, *d_e2 = nullptr;   // per-frame, per-trunk row / column phasors of PMD plans with a linear db1 (k_pmd_tab)
    unsigned long long *d_slots = nullptr;   // slot barrier of the fused column sweep: [launch parity][frame][tile]
    unsigned long long *d_mbox = nullptr;    // [teams][frames + 4] mailboxes of the fused column sweep's teams, then the two claim counters
    size_t mbox_bytes = 0;
    int fused = 0, fused_grid = 0, tiles_pf = 0;
    uint32_t flags = 0;                      // plx_ssfm_create_ex
    int barrier_timeouts = 0;                // propagate calls of this plan that ended in a frame-barrier time-out (it then takes the three-sweep step
                                             // until plx_ssfm_barrier_timeouts(..., rearm) -- the gateway tier re-arms its cached plans itself)
    int calls_unfused = 0, rearm_after = 16; // gateway tier: three-sweep calls since the last time-out / how many of them before the fused step is tried again
    double *d_dzlist = nullptr, *d_dzlog = nullptr;   // diagnostics: replayed / logged step sequences
    int dzlist_cap = 0;
    int col_threads = 512;         // workgroup size of k_col_fwd / k_col_inv
    int row_threads = ROW_THREADS; // workgroup size of k_row
    int rowr = 0;                  // k_row256r serves the step's row pass
    int row4k_split = 0;           // k_row4k<false, true>: the same for 4096-point rows
    int rowsm = 0;                 // k_rowsm<p2> serves it (rows of 32 / 64 / 128 points; dual polarisation without PMD, scalar)
    int row256_split = 0;          // k_row256r<true, false, true>: the PMD form with tables at three waves per SIMD
    int rowg_pair_split = 0;       // ... and the PMD form with phasor tables as well (k_rowreg<., true, false, true>)
    int rowg_split = 0;            // ... with the exchanges split into real and imaginary halves (three workgroups per CU)
    int rowreg = 0;                // k_rowreg<p2> serves it (dual polarisation, no PMD, rows of 512 / 1024 / 2048 points)
    cplx *d_tw2c = nullptr, *d_twmid = nullptr;
    int row_split = 0, rs_threads = 0; // long rows without PMD: one polarisation per workgroup (scalar row pass twice)
    size_t rs_lds = 0;
    int tw_compact = 0;            // 4096-point rows: compact twiddle table in d_tw2, register-blocked row pass k_row4k
    int row_pair4k = 0;            // ... of a PMD-type plan: both polarisations of a row in one workgroup (k_row4k<true>)
    size_t rs_lds_pair = 0;
    double *h_brf[2] = {nullptr, nullptr}; // pinned staging of the waveplate tables
    hipEvent_t brf_ev[2] = {nullptr, nullptr};
    int brf_slot = 0;
    int64_t row_launches = 0, sample_steps = 0;
    int64_t slots_launched = 0, slots_listed = 0, frame_steps = 0;   // utilisation accounting of the last propagate
    // optional per-kernel timing of the step loop (plx_ssfm_profile): one event between consecutive launches
    int profile = 0;
    // The intervals are read LATER -- while the next call's first launches run, or when the times are asked for: some 160
    // hipEventElapsedTime calls per propagate would otherwise sit between two calls with the GPU idle (~2 ms per 110 ms).
    struct ProfRun { std::vector<hipEvent_t> ev; std::vector<int> cls, step; int maxnc = 0; bool fused = false; };
    std::vector<hipEvent_t> evfree;          // events not in use
    std::vector<ProfRun> prof_pending;       // finished step loops whose intervals have not been read yet
    double k_ms[4] = {0, 0, 0, 0};           // accumulated since the last plx_ssfm_kernel_times
    int64_t k_launches[4] = {0, 0, 0, 0};
};

static const double kInv2Pi = 0.15915494309189533577;

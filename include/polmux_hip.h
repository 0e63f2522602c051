/*
 * polmux_hip.h -- C ABI of libpolmux_hip.so: the MI355X (gfx950) implementation
 * of the Optilux/Polmux hot path.
 *
 * The reference's only native seam is the MATLAB MEX gateway
 *   void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
 * (fastexp.c:46, cmaadaptivefilter.c:93, easiadaptivefilter.c:95).  A MEX shim
 * unpacks mxArrays into plain pointers and sizes and forwards to the functions
 * below; INTEGRATION.md shows those shims and the .m wrappers.  Every entry
 * point cites the reference interface it replaces.
 *
 * Two tiers:
 *   (A) "gateway" calls  plx_<name>(...)      host pointers, MEX-shaped: the
 *       arrays are MATLAB column-major, complex data as SEPARATE re/im planes
 *       (mxGetPr/mxGetPi; an all-real input has im == NULL).  Each call uploads,
 *       runs the kernels, downloads.  Drop-in for one MATLAB call.  The plans
 *       (twiddle / phase tables), device buffers and pinned staging memory these
 *       calls need are OWNED BY THE LIBRARY and kept between calls (the drivers
 *       call a filter gateway up to 299 times per frame, DspPdmCohQpsk.m:176-191,
 *       and the propagator once per span, fiber.m:372-389): plans are found again
 *       by a hash of the descriptor's scalars and table contents, a repeated call
 *       allocates nothing.  plx_release_all() frees that state; a MEX shim calls
 *       mexLock() and registers it with mexAtExit() (integration/mex/plx_mex_common.h).
 *   (B) "resident" calls plx_<name>_dev(...)  device pointers + HIP stream,
 *       interleaved complex128 (re,im), batched over frames.  This is what the
 *       Monte-Carlo runner and bench.py use; nothing crosses PCIe per call.
 *
 * All functions return 0 (PLX_OK) or a negative PLX_ERR_* code;
 * plx_last_error() returns the message (the strings of the reference's
 * error()/mexErrMsgTxt calls where one exists).  Not re-entrant per plan;
 * different plans may be used from different host threads.
 */
#ifndef POLMUX_HIP_H
#define POLMUX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLX_OK 0
#define PLX_ERR_HIP (-1)       /* a HIP runtime call failed                         */
#define PLX_ERR_ARG (-2)       /* bad argument (message mirrors the reference)      */
#define PLX_ERR_UNSUPPORTED (-3)
#define PLX_ERR_REFERENCE (-4) /* the reference itself raises here, e.g. fiber.m:854 */
#define PLX_ERR_TIMEOUT (-5)   /* a frame barrier of the fused sweep timed out: another kernel holds the GPU (see plx_ssfm_propagate_dev) */

const char *plx_last_error(void);
/* ABI version of this header: major*1000 + minor */
int plx_abi_version(void);
/* number of visible HIP devices, and selection of the one this process uses */
int plx_device_count(int *count);
int plx_set_device(int device);

/* Library-owned gateway state (tier A): plx_release_all() destroys the cached plans and frees the scratch and
 * staging buffers of every gateway (safe to call at any time; the next gateway call rebuilds what it needs).
 * plx_gateway_stats(out[8]): calls, device allocations, pinned-host allocations, plans built, plans found in the
 * cache, device bytes held, pinned bytes held, releases -- so that a caller (and the tests) can check that a
 * repeated call allocates nothing.                                                                               */
int plx_release_all(void);
int plx_gateway_stats(int64_t *out);
/* the same with room for later counters: fills min(n, 9) entries; out[8] = gateway propagations that were repeated on the
 * barrier-free three-sweep step after a frame-barrier time-out (another kernel held part of the GPU)                  */
int plx_gateway_stats_ex(int64_t *out, int n);

/* ------------------------------------------------------------------ fastexp --- */
/* fastexp.c:37-47 / fastexp.m:28: y = cos(x) + i*sin(x), x real [m x n].           */
int plx_fastexp(const double *x, double *yr, double *yi, size_t count);
/* resident: y interleaved complex128 */
int plx_fastexp_dev(const double *d_x, double *d_y, size_t count, void *stream);

/* ------------------------------------------------------------ SSFM propagator --- */
/* One plan = one fibre type on a fixed grid: the arguments fiber.m:372-389 hands
 * to matrix_ssfm / scalar_ssfm (fiber.m:459-460, 557-558), minus the field.       */
typedef struct plx_ssfm_desc {
    int64_t nfft;        /* Nfft = NSYMB*NT, power of two, 256 .. 2^20            */
    int32_t nfc;         /* columns of GSTATE.FIELDX (1 = 'unique' field)         */
    int32_t dual_pol;    /* 1: matrix_ssfm (FIELDY present or 'p' flag), 0: scalar */
    int32_t max_frames;  /* batch capacity: independent frames per call           */
    int32_t fls[4];      /* flag -> [gvd pmd spm xpm], fiber.m:157-251            */
    double dzmaxt;       /* fiber.m:160-247                                       */
    double dphimaxt;     /* Inf allowed                                           */
    double alphalin;     /* [1/m] fiber.m:302                                     */
    double length;       /* Lf [m]                                                */
    int32_t nplates;     /* x.nplates (1 when PMD is off, fiber.m:297)            */
    int32_t manakov;     /* strcmp(x.manakov,'yes')                               */
    const double *gam;   /* [nfc] host, [1/mW/m] fiber.m:325,328                  */
    const double *betat; /* [nfft x nfc] host column-major, fiber.m:355-356       */
    const double *db1;   /* [nfft x nfc] host (zeros/NULL without PMD) :358       */
} plx_ssfm_desc;

typedef struct plx_ssfm plx_ssfm;

int plx_ssfm_create(plx_ssfm **plan, const plx_ssfm_desc *desc);
/* flags: PLX_SSFM_SHARE_DEVICE -- the plan takes the barrier-free three-sweep step (no workgroup of it ever waits for
 * another one): for processes that share a GPU, and for plans that propagate beside a long-running kernel of another
 * stream (a one-team plan -- plx_ssfm_info: info[4] == info[3] -- whose receiver runs beside the next batch's fibre). */
#define PLX_SSFM_SHARE_DEVICE 1u
int plx_ssfm_create_ex(plx_ssfm **plan, const plx_ssfm_desc *desc, uint32_t flags);
/* Plan-time tuning: which kernels and which split a plan takes where the library has more than one (the tests compare
 * each kernel with the one it replaced; A/B measurements).  NOT read from the environment: a plan created without a
 * tuning takes the defaults, which consult exactly two deployment variables -- PLX_SSFM_NO_FUSE=1 (every plan of the
 * process on the barrier-free three-sweep step: several processes on one GPU) and PLX_SSFM_BARRIER_TIMEOUT_MS (the frame
 * barrier's patience, default 500).  plx_ssfm_tuning_defaults fills the struct (always call it first: it sets `size`);
 * plx_ssfm_create_tuned(..., tuning) builds one plan with it (NULL: the process-wide override if one is set, else the
 * defaults); plx_ssfm_tuning_override(t) makes t the tuning of every plan created WITHOUT one from now on -- including
 * the plans the gateway tier builds and caches (call plx_release_all() to drop those built earlier) -- NULL ends it.
 * A test hook: deployments never call it (and it is not synchronised: set it while no other thread creates plans).   */
typedef struct plx_ssfm_tuning {
    uint32_t size;           /* sizeof(plx_ssfm_tuning), set by plx_ssfm_tuning_defaults                              */
    int32_t no_fuse;         /* 1: three sweeps per step (k_col_fwd, row pass, k_col_inv) instead of the fused sweep   */
    int32_t short_rows;      /* 1: 2^20-sample frames on the 512 x 2048 split instead of 256 x 4096                    */
    int32_t no_row_split;    /* 1: long rows keep both polarisations in one workgroup of k_row                         */
    int32_t p1, logW;        /* forced log2 of the column length / of the tile width (-1: the plan's choice)           */
    int32_t col_threads;     /* forced workgroup size of k_col_fwd / k_col_inv: 128 ... 1024 (-1: the plan's choice)   */
    int32_t rowr;            /* 0: the LDS-resident k_row also where a register-form row pass applies                  */
    int32_t rowsm;           /* 0: k_rowsm nowhere; 2: wherever it applies (default 1: where it is the faster kernel)  */
    int32_t row256_split, row4k_split, rowg_split;   /* 0: whole-sample LDS exchanges in k_row256r<PMD> / k_row4k / k_rowreg */
    int32_t no_pmd_tab;      /* 1: PMD plans form one exponential per bin and trunk instead of the phasor tables       */
    int32_t store_late;      /* fused sweep: a tile's stores after the next tile's landing (-1: multi-team launches)   */
    int32_t row_rev;         /* 0: the row pass takes the listed frames in ascending order as well                     */
    int32_t safe_landing;    /* 1: the fused sweep also waits for its staged tile with vmcnt(0)                        */
    int32_t reserved_[4];
    double barrier_timeout_ms;
} plx_ssfm_tuning;
int plx_ssfm_tuning_defaults(plx_ssfm_tuning *tuning);
int plx_ssfm_tuning_override(const plx_ssfm_tuning *tuning);
int plx_ssfm_create_tuned(plx_ssfm **plan, const plx_ssfm_desc *desc, uint32_t flags, const plx_ssfm_tuning *tuning);
int plx_ssfm_destroy(plx_ssfm *plan);
/* brf.db0/theta/epsilon (fiber.m:266-276): host arrays [nplates x nsets]; set s is
 * used by frame f = s (nsets == 1: shared by all frames).                          */
int plx_ssfm_set_birefringence(plx_ssfm *plan, const double *db0, const double *theta,
                               const double *epsilon, int nsets);
/* the same, stream-ordered and without waiting: the tables take effect for propagate calls enqueued on
 * `stream` after this call (Monte-Carlo loops draw the next batch's waveplates while the GPU is busy) */
int plx_ssfm_set_birefringence_dev(plx_ssfm *plan, const double *db0, const double *theta,
                                   const double *epsilon, int nsets, void *stream);
/* Propagate nframes frames in place.  d_ux/d_uy: device, interleaved complex128,
 * layout [frame][channel][nfft]; d_uy NULL for a scalar plan.  Asynchronous on
 * `stream` except for the bounded polling of the data-dependent step loop: the call
 * returns when every frame has reached the fibre end.
 * A plan is NOT re-entrant: it owns the per-call step-control state, so ONE propagate
 * (or filter) call may be in flight per plan; different plans may run from different
 * host threads / streams.
 * Dual-polarisation plans with 256-row column tiles use a fused sweep (k_colx16) whose
 * workgroups of one frame meet at a barrier inside the launch (the grid is a whole number
 * of such teams; a team claims frames one at a time, so one that is displaced by another
 * kernel's waves leaves its share to the others).  Its grid is sized at
 * plan creation from hipOccupancyMaxActiveBlocksPerMultiprocessor for that kernel, so
 * the frame's workgroups are co-resident on a GPU the process has to itself; when the
 * frame does not fit the chip that way the plan takes the barrier-free three-sweep
 * step.  If another long-running kernel holds the device (two processes on one GPU, a
 * second plan propagating at the same time), a barrier that cannot complete within
 * 0.5 s raises a sticky abort: nothing is stored or advanced after it, the call returns
 * PLX_ERR_TIMEOUT and the field of THAT call is invalid (it was propagated in place up to
 * the time-out); the plan switches itself to the barrier-free three-sweep step, so the
 * caller restores the field and calls again (the gateway calls plx_matrix_ssfm /
 * plx_scalar_ssfm do exactly that from their staging copy and return success).
 * plx_ssfm_create_ex(..., PLX_SSFM_SHARE_DEVICE) selects that step from the start for
 * such deployments.  Short kernels of another stream
 * of the same process are harmless as long as one frame takes at most half of the grid
 * (plx_ssfm_info: 2 * info[4] <= info[3]); larger frames (2^19, 2^20 samples) should have
 * the device to themselves while they propagate.                                     */
int plx_ssfm_propagate_dev(plx_ssfm *plan, double *d_ux, double *d_uy, int nframes, void *stream);
/* Diagnostics for parity work on ill-conditioned step sequences (noise-loaded WDM fields: the step rule fiber.m:682-715
 * amplifies rounding differences).  set_step_sequence: the step length nextstep would return for step k (0-based) is
 * replaced by dz[k] for k < nsteps, for every frame of later propagate calls (nsteps 0: off); the loop around it
 * (fiber.m:512-551: zprop, the last-step rule) is unchanged.  log_steps(max_steps > 0): later propagate calls record
 * every frame's step lengths; step_sequence copies frame f's first max_steps of them out (ncycle of them are valid). */
int plx_ssfm_set_step_sequence(plx_ssfm *plan, const double *dz, int nsteps);
int plx_ssfm_log_steps(plx_ssfm *plan, int max_steps);
int plx_ssfm_step_sequence(plx_ssfm *plan, int frame, double *dz, int max_steps);
/* per-frame results of the last propagate: firstdz, ncycle (fiber.m:431)           */
int plx_ssfm_results(plx_ssfm *plan, int nframes, double *firstdz, int32_t *ncycle);
/* kernel-time accounting of the last propagate: launches of the dominant
 * (row-pass) kernel and sample-steps processed                                     */
int plx_ssfm_stats(plx_ssfm *plan, int64_t *row_pass_launches, int64_t *sample_steps);
/* Optional per-kernel timing of the step loop, for roofline reports: with profiling enabled every propagate call
 * records a HIP event between consecutive launches on its stream; after the call, ms[k] / launches[k] hold the
 * summed duration and count of the ACTIVE launches (those issued before the slowest frame had finished) of kernel
 * class k: 0 = column sweep that starts a step (fused k_colx16, or k_col_fwd), 1 = the row pass (k_row256r, k_row4k or k_row: plx_ssfm_info), 2 = k_col_inv,
 * 3 = step control / row sums / read-backs.  Both arrays have 4 entries.                                         */
/* Lock-step accounting of the last propagate (frames of a batch need different numbers of steps, fiber.m:518): the
 * frame-steps that had work to do; the frame slots of the device's active list summed over the steps (what the
 * workgroups iterated over: the list is rebuilt before every step, so this equals the first figure); the frame slots the
 * host's launches covered (its view of the list lags by up to two chunks of 8 steps; surplus workgroups exit at once). */
int plx_ssfm_utilisation(plx_ssfm *plan, int64_t *frame_steps, int64_t *slots_listed, int64_t *slots_launched);
/* How the plan runs a step (for reports): info[0] 1 = fused column sweep (two sweeps per step), 0 = three sweeps;
 * [1] log2 N1, [2] log2 N2 of the four-step split; [3] grid of the fused sweep; [4] column tiles per frame;
 * [5] threads per column workgroup, [6] per row workgroup of the step's row pass (64: the register form k_row256r, one wave
 * per 2 rows x 2 polarisations; with [2] = 12: k_row4k, 256 threads on one polarisation of a row or -- PMD plans -- 512 on both);
 * [7] 1 = one polarisation per row workgroup, 2 = the register form for rows of 512 / 1024 / 2048 points (k_rowreg: 256
 * threads on 8 / 4 / 2 row-polarisations), 0 = both polarisations of its rows in one workgroup.  8 entries. */
int plx_ssfm_info(plx_ssfm *plan, int32_t *info);
/* Frame-barrier time-outs of this plan so far (each one made a propagate call return PLX_ERR_TIMEOUT and switched the plan
 * to the barrier-free three-sweep step); rearm != 0 switches a plan that was created with the fused step back to it (the
 * caller knows that the other kernel has left the GPU).  The gateway tier does this for its cached plans by itself: after
 * a fallback the next 16 spans take three sweeps, then the fused step is tried again (32, 64, ... 4096 after further
 * time-outs).  count may be NULL.                                                                                     */
int plx_ssfm_barrier_timeouts(plx_ssfm *plan, int32_t *count, int rearm);
/* Per-kernel timing of the step loop: with profiling enabled an event is recorded between consecutive launches of
 * plx_ssfm_propagate_dev; plx_ssfm_kernel_times returns, per kernel class (0 the column sweep that starts a step, 1 the row
 * pass, 2 k_col_inv, 3 control), the milliseconds and the number of ACTIVE launches accumulated over all propagate calls
 * since its previous call, and resets them (the intervals of a call are read while the next call runs, or here). */
int plx_ssfm_profile(plx_ssfm *plan, int enable);
int plx_ssfm_kernel_times(plx_ssfm *plan, double *ms, int64_t *launches);

/* gateway forms: [firstdz,ncycle,ux,uy,brf]=matrix_ssfm(...) fiber.m:459-460 and
 * [firstdz,ncycle,u]=scalar_ssfm(...) :557-558 on one frame; split planes in/out
 * ([nfft x nfc] each; *_i may be NULL on input only if the out planes are given).   */
int plx_matrix_ssfm(double *uxr, double *uxi, double *uyr, double *uyi, const plx_ssfm_desc *desc,
                    const double *db0, const double *theta, const double *epsilon,
                    double *firstdz, int32_t *ncycle);
int plx_scalar_ssfm(double *ur, double *ui, const plx_ssfm_desc *desc, double *firstdz, int32_t *ncycle);
/* adaptive step size by local error (x.ltol): tolflag 2 = [firstdz,ncycle,u]=scalar_a_ssfm(...) with adaptssfm
 * (fiber.m:639-679, 938-1009); tolflag 1 = x.dphiadapt: adaptive first step, then the constant-phase loop with
 * the corrected dphimax (fiber.m:588-611).  trg.err = ltol, trg.safety = 0.9 in fiber.m:145-146.  Scalar fields
 * only: a dual-polarisation descriptor returns PLX_ERR_REFERENCE with the message of fiber.m:374.               */
int plx_scalar_ssfm_adaptive(double *ur, double *ui, const plx_ssfm_desc *desc, int tolflag, double ltol,
                             double safety, double *firstdz, int32_t *ncycle, int32_t *nrej);

/* ------------------------------------------------------ overlap-save CD equaliser --- */
/* y = OverlapBothTrans(x, H, L)  CDE_OFDE.m:62-125, batched: x [nsig][nx], H [N] on
 * the fftshift-ordered grid (CDE_OFDE.m:30-38), both interleaved complex128.       */
typedef struct plx_cde plx_cde;
int plx_cde_create(plx_cde **plan, int64_t fft_len, int64_t L, const double *H_interleaved);
int plx_cde_destroy(plx_cde *plan);
int plx_cde_apply_dev(plx_cde *plan, const double *d_x, double *d_y, int64_t nx, int nsig, void *stream);
/* gateway: the whole CDE_OFDE(inX, inY, fs, lambdaRef, span, D, S, fftLength, L) call,
 * CDE_OFDE.m:16-47 (H built on the device side of the ABI in double, host code).   */
int plx_cde_ofde(const double *xr, const double *xi, const double *yr, const double *yi, int64_t nx,
                 double fs, double lambda_ref, double span, double D, double S, int64_t fft_len,
                 int64_t L, double *oxr, double *oxi, double *oyr, double *oyi);
/* the reference display()s a message and returns [] on bad arguments (:63-85): the
 * gateway returns PLX_ERR_ARG with that message and leaves the outputs untouched.  */

/* -------------------------------------------------- CMA / EASI 2x2 butterfly --- */
/* [Y,h1,h2] = cmaadaptivefilter(xx,h1,h2,taps,mu,R,sps)  cmaadaptivefilter.c:93-174.
 * xx [Mdim x 2], h1/h2 [taps x 2], y [Mdim-taps+1 x 2]; split planes.  Like the
 * reference, h1/h2 are UPDATED IN PLACE (the MEX shim then returns 0,0 in plhs[1..2],
 * cmaadaptivefilter.c:166-171).  Errors: "Ntaps should be an ODD INTEGER." /
 * "Samples x symbol should be either 1 or 2."                                      */
int plx_cmaadaptivefilter(const double *xr, const double *xi, int32_t Mdim, double *h1r, double *h1i,
                          double *h2r, double *h2i, double Ntap, double mu, const double *R, double sps,
                          double *yr, double *yi);
/* [Y,h1,h2] = easiadaptivefilter(xx,h1,h2,taps,mu,sps)  easiadaptivefilter.c:95-169 */
int plx_easiadaptivefilter(const double *xr, const double *xi, int32_t Mdim, double *h1r, double *h1i,
                           double *h2r, double *h2i, double Ntap, double mu, double sps, double *yr,
                           double *yi);

/* The .m TWINS of the two filters -- what MATLAB runs when the MEX files are not compiled (comp_mex.m not run; the
 * name then resolves to the .m, fastexp.m:6-7) -- with the semantics the twins really have, which differ from the C:
 * cmaadaptivefilter.m:52-72 updates at EVERY sample (no sps gate), has no odd-taps check and RETURNS the updated taps;
 * easiadaptivefilter.m:51-84 forms the error matrix from the complex outputs (abs(), complex denominators) and
 * recombines ALL taps of the complex h1, h2.  ntap = size(h1,1); h1/h2 are updated in place (the shim returns them in
 * plhs[1..2], so the unchanged drivers take their any(any(h1_new)) branch, DspPdmCohQpsk.m:183-186).                */
int plx_cmaadaptivefilter_m(const double *xr, const double *xi, int32_t Mdim, double *h1r, double *h1i,
                            double *h2r, double *h2i, int32_t ntap, double mu, const double *R, double *yr, double *yi);
int plx_easiadaptivefilter_m(const double *xr, const double *xi, int32_t Mdim, double *h1r, double *h1i,
                             double *h2r, double *h2i, int32_t ntap, double mu, double *yr, double *yi);

/* y = cmapolardemux(x, params) / y = easipolardemux(x, params) as ONE gateway call: the driver loops of
 * DspPdmCohQpsk.m:142-192 / :195-244 (== dsp4cohdec.m:374-425 / :427-478) -- cyclic extension, centre taps set from M,
 * up to 50*ceil(1/(L*mu)) - 1 (CMA) or 20*ceil(1/(L*mu)) - 1 (EASI) passes, the 5e-5 convergence test -- run on the
 * device; the unchanged per-pass MEX costs a PCIe round trip per pass (up to 299 per frame).  x [L x 2] split planes
 * (xi may be NULL); M: the initial centre-tap matrix the driver forms at :146-160 from params.mat / params.phizero / the
 * single-polarisation ratio, row-major 2x2 complex as 8 doubles (re,im); R = params.R [2], mu = params.mu, taps =
 * params.taps.  y [L x 2] split planes; optional: final taps h1, h2 [taps x 2] planes, passes made.  mfile_twin (EASI):
 * the loop around the .m twin of the filter (easiadaptivefilter.m:51-84) instead of the C (the CMA twins coincide under
 * the drivers' sps = 1).  Errors as the filter gateways: "Ntaps should be an ODD INTEGER."                              */
int plx_cmapolardemux(const double *xr, const double *xi, int64_t L, int32_t taps, double mu, const double *R,
                      const double *M, double *yr, double *yi, double *h1r, double *h1i, double *h2r, double *h2i,
                      int32_t *passes);
int plx_easipolardemux(const double *xr, const double *xi, int64_t L, double mu, const double *M, int32_t mfile_twin,
                       double *yr, double *yi, double *h1r, double *h1i, double *h2r, double *h2i, int32_t *passes);

/* resident, batched pol-demux driver: cmapolardemux / easipolardemux
 * (DspPdmCohQpsk.m:142-244 == dsp4cohdec.m:374-478): cyclic extension, centre-tap
 * init from M, pass loop with the 5e-5 convergence test, all on the device.
 * d_x, d_y: [frame][2][L] interleaved complex128.  d_M: [frame][4] complex (row-major
 * 2x2) initial centre taps; d_h (optional out): [frame][2][2*taps] final taps;
 * d_passes (optional out): int32 [frame].                                          */
#define PLX_DEMUX_CMA 1
#define PLX_DEMUX_EASI 2
#define PLX_DEMUX_EASI_M 3   /* easipolardemux around the .m twin of the filter (easiadaptivefilter.m:51-84) */
int plx_poldemux_dev(int method, const double *d_x, double *d_y, int64_t L, int nframes, int32_t taps,
                     double mu, const double *R /* host [2], CMA only */, const double *d_M,
                     double *d_h, int32_t *d_passes, void *stream);

/* ------------------------------------------- DspPdmCohQpsk body (resident) --- */
typedef struct plx_dsp_params {
    int32_t workatbaudrate; /* DspPdmCohQpsk.m:12 */
    int32_t applynlr;       /* :17 (NLRotation)   */
    double nlralpha;
    double power_mw;        /* GSTATE.POWER(chNum) :22 */
    int32_t applypol;       /* :26 */
    int32_t polmethod;      /* 0 singlepol 1 cma 2 easi 3 combo */
    double cma_R[2];
    double cma_mu;
    int32_t cma_taps;
    int32_t cma_txpolars;
    double cma_phizero;
    double easi_mu;
    int32_t easi_txpolars;
    double easi_phizero;
    int32_t modorder, freqavg, phasavg, poworder;
    /* params.mat: explicit initial centre-tap matrix (DspPdmCohQpsk.m:148-149, :201-202), row-major 2x2 complex (re,im) */
    int32_t cma_has_mat, easi_has_mat;
    double cma_mat[8], easi_mat[8];
    int32_t mfile_twins;    /* 1: the drivers call the .m twins of the filters (no MEX compiled) */
    int32_t reserved_;
} plx_dsp_params;

typedef struct plx_dsp plx_dsp;
/* Lin: samples per polarisation handed to DspPdmCohQpsk (2 sps unless workatbaudrate) */
int plx_dsp_create(plx_dsp **plan, int64_t Lin, int32_t ncol, int32_t max_frames, const plx_dsp_params *p);
int plx_dsp_destroy(plx_dsp *plan);
/* d_in [frame][ncol][Lin] -> d_out [frame][ncol][Lout], Lout = Lin or ceil(Lin/2)  */
int plx_dsp_run_dev(plx_dsp *plan, const double *d_in, double *d_out, int nframes, void *stream);
int64_t plx_dsp_out_len(const plx_dsp *plan);

/* samp2pat 'coherent' decisions (samp2pat.m:61-66) + error count against the
 * transmitted pattern (ber_estimate.m:119 err = sum(sum(pat ~= pat_hat))):
 * d_sym [frame][ncol][L] complex; d_pat uint8 [ncol*2][L] (shared by frames, may be
 * NULL); d_pat_hat (optional) uint8 [frame][ncol*2][L]; d_err int64 [frame][ncol]
 * (errors of each column's two bit streams, so a caller can resolve the pi/2
 * ambiguity of each polarisation separately; their sum is the reference's err).    */
int plx_decide_count_dev(const double *d_sym, int64_t L, int32_t ncol, int nframes, const uint8_t *d_pat,
                         uint8_t *d_pat_hat, int64_t *d_err, void *stream);
/* the same with a transmitted pattern PER FRAME (frames of a batch that carry different sequences): frame f compares
 * with d_pat + f * pat_frame_stride bytes (0: one shared pattern, as above)                                       */
int plx_decide_count_frames_dev(const double *d_sym, int64_t L, int32_t ncol, int nframes, const uint8_t *d_pat,
                                int64_t pat_frame_stride, uint8_t *d_pat_hat, int64_t *d_err, void *stream);

/* per-frame error-vector magnitude of the recovered symbols, mean |s - s_hat|^2 with s_hat the unit-modulus QPSK point of
 * the decided quadrant: one continuous sample per realisation for mc_estimate (mc_estimate.m:133-212), the way the error
 * count above feeds ber_estimate.  d_sym [frame][ncol][L] complex; d_evm double [frame].                               */
int plx_evm_dev(const double *d_sym, int64_t L, int32_t ncol, int nframes, double *d_evm, void *stream);

/* ----------------------------------------------------------------- ampliflat --- */
/* ampliflat(x,'gain',options), ampliflat.m:60-148 ("next" row): FIELD *= sqrt(gain) and
 * FIELD += sigma(c) * n with n complex Gaussian, E|Re n|^2 = E|Im n|^2 = 1.  sigma: host [nfc] (NULL/0 = no
 * ASE, :91-105).  d_noise: optional injected noise like options.noise [frame][X cols | Y cols][nfft] complex
 * (:123-129, the parity route); otherwise a counter-based Philox-4x32-10 stream keyed by (seed, d_keys[frame] or
 * frame).  asex/asey: ASE on x / y (options.onepol, :107-118).  d_uy may be NULL (single polarisation).        */
int plx_ampliflat_dev(double *d_ux, double *d_uy, int64_t nfft, int32_t nfc, int nframes, double gain_lin,
                      const double *sigma, const double *d_noise, uint64_t seed, const int64_t *d_keys,
                      int32_t asex, int32_t asey, void *stream);

/* ------------------------------------------------------------ coherent front end --- */
/* The step between fiber() and CDE_OFDE(): receiver_cohmix.m:165-307 (optical filter x post-compensation,
 * LO mixing in two 90-degree hybrids, balanced or single photodiodes, electrical low-pass) followed by
 * RxPdmCohQpsk.m:36-72 (ADC quantisation, timing shift, decimation, I/Q recombination), batched over frames.
 * Tables are on the fft-ordered grid GSTATE.FN and are built by the caller (myfilter.m, :143-168, :193-227). */
typedef struct plx_front plx_front;
typedef struct plx_front_desc {
    int64_t nfft;            /* samples per frame                                                        */
    int32_t dual_pol;        /* 1: GSTATE.FIELDY present (isy, :229), 0: X only                          */
    int32_t max_frames;
    int32_t balanced;        /* 1: balanced detection (default), 0: x.pdtype == 'normal' (:265-278)      */
    int32_t adcbits;         /* RxParams.adcbits when RxParams.applyadc, 0: no ADC (RxPdmCohQpsk.m:36-40) */
    int32_t decim;           /* DecimationRate (RxPdmCohQpsk.m:49-53); 1: keep every sample              */
    int32_t ntaps;           /* taps of the decimation FIR (17 for decimate(x,r,16,'fir')), odd          */
    const double *fir;       /* [ntaps] host                                                             */
    const double *hopt_re, *hopt_im;   /* [nfft] Hf of receiver_cohmix.m:165-168 (hopt_im may be NULL)   */
    const double *hel_re, *hel_im;     /* [nfft] Hf of receiver_cohmix.m:296   (hel_im may be NULL)      */
    const double *elo_re, *elo_im;     /* [nfft] Elo of receiver_cohmix.m:227, or NULL: Elo = elo_scalar */
    double elo_scalar;       /* LO_Ecw = 10^(x.lopower/20) (:221-225)                                    */
} plx_front_desc;
int plx_front_create(plx_front **plan, const plx_front_desc *desc);
int plx_front_destroy(plx_front *plan);
int64_t plx_front_out_len(const plx_front *plan);   /* ceil(nfft / decim) */
/* d_ux, d_uy: [nframes][nfft] complex128 fields, OVERWRITTEN: on return they hold the filtered photocurrents
 * of each polarisation as I + jQ (the columns [IricX IricY] of receiver_cohmix.m:300-307) before ADC/shift.
 * shift: host, [1 + dual_pol] circular shifts round(-delay*NT) of fastshift (RxPdmCohQpsk.m:42-44), or NULL.
 * d_out: [nframes][1 + dual_pol][out_len] complex128 RxSamples (RxPdmCohQpsk.m:63-72).                     */
int plx_front_run_dev(plx_front *plan, double *d_ux, double *d_uy, int nframes, const int64_t *shift,
                      double *d_out, void *stream);
/* gateway tier (one frame, host arrays, MATLAB's separate planes; xi/yi may be NULL, yr NULL when !dual_pol):
 * out [out_len x (1 + dual_pol)] column-major = RxSamples; optional cur_r/cur_i [nfft x 2(1 + dual_pol)] = the
 * photocurrent columns [IricX IricY] of receiver_cohmix.m:300-307 (cur_i is filled with zeros).            */
int plx_rx_front(const double *xr, const double *xi, const double *yr, const double *yi, const plx_front_desc *desc,
                 const int64_t *shift, double *outr, double *outi, double *cur_r, double *cur_i);

/* y = ifft(fft(x) .* H), in place, on [nsignals][nfft] complex128 rows sharing one frequency response H (fft
 * order): the dispersion-compensating filter of RxPdmCohQpsk.m:74-84 / dsp4cohdec.m:163-173 (Hfilt built by
 * DispCompFilter on the host) or any other fixed response.  nfft: power of two in [256, 2^20].             */
typedef struct plx_filter plx_filter;
int plx_filter_create(plx_filter **plan, int64_t nfft, int max_signals, const double *h_re, const double *h_im);
int plx_filter_destroy(plx_filter *plan);
int plx_filter_apply_dev(plx_filter *plan, double *d_x, int nsignals, void *stream);

/* ------------------------------------------------------------------ inverse PMD --- */
/* inverse_pmd(brf, options)  inverse_pmd.m:91-145: the per-frequency PMD matrix U of a link of `nfibers` fibres
 * (brf{n} as returned by fiber), Uinv = U^H, and its application to a unique dual-polarisation field.
 * set_link: ntrunk[nfibers]; db0/theta/epsilon concatenated over fibres, [nsets][sum ntrunk] (nsets = 1: one link
 * for every frame; nsets = frames: a waveplate draw per frame); lcorr[nfibers]; betat, db1: [nfibers][nfft]
 * (brf{n}.betat, brf{n}.db1); mat: options.mat as 8 doubles (row-major re,im) or NULL; apply_gvd 0: options.gvd='no'. */
typedef struct plx_pmdinv plx_pmdinv;
int plx_pmdinv_create(plx_pmdinv **plan, int64_t nfft, int max_frames);
int plx_pmdinv_destroy(plx_pmdinv *plan);
int plx_pmdinv_set_link(plx_pmdinv *plan, int nfibers, const int32_t *ntrunk, const double *db0, const double *theta,
                        const double *epsilon, const double *lcorr, const double *betat, const double *db1,
                        const double *mat, int apply_gvd, int nsets);
/* FIELD = ifft(Uinv * fft(FIELD)) in place, d_ux/d_uy [nframes][nfft] complex128 (inverse_pmd.m:139-145) */
int plx_pmdinv_apply_dev(plx_pmdinv *plan, double *d_ux, double *d_uy, int nframes, void *stream);
/* host copies of U and/or Uinv of one frame in MATLAB's [2][2][nfft] column-major layout, interleaved complex */
int plx_pmdinv_matrices(plx_pmdinv *plan, int frame, double *U, double *Uinv);

/* ------------------------------------------------------------ small helpers --- */
/* strided pick + scale used between fibre and CDE when the full front end
 * (receiver_cohmix + decimate) is not in the chain:
 * out[s*out_pitch + i] = scale * in[s*n_in + offset + i*stride], s < nsig
 * (out_pitch 0 = n_out; a larger pitch interleaves the two polarisations of a frame) */
int plx_pick_dev(const double *d_in, double *d_out, int64_t n_in, int64_t n_out, int64_t offset,
                 int64_t stride, double scale, int nsig, int64_t out_pitch, void *stream);

#ifdef __cplusplus
}
#endif
#endif

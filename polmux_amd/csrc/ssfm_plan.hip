// ssfm_plan.hip -- the plan and the step loop of the split-step Fourier propagator of fiber.m on gfx950 (create / destroy,
// waveplate tables, propagate, per-kernel timing, the FFT engine as a spectral filter).  The kernels live in ssfm_small.hip,
// ssfm_col.hip, ssfm_colx.hip, ssfm_row*.hip (ssfm_kernels.h: what this file sees of them), the gateways in ssfm_gateway.hip.
//
// Reference: /root/reference/fiber.m:459-935 (matrix_ssfm, scalar_ssfm, nextstep,
// checkstep, lin_step, nl_step, matrix_nl_step, matrix_step).
//
// MI355X design (not a translation of the MATLAB): the field stays in HBM, a step is two or three
// sweeps over it, organised around a four-step FFT  N = N1 x N2  whose forward half is
// decimation-in-frequency and whose inverse half is decimation-in-time, so the spectrum is only
// ever held in (bit-reversed, transposed) order and no reorder pass exists:
//
//   k_col_fwd  load N1 x 16 column tile -> [Kerr step fused on load] -> N1-point
//              DIF in LDS -> store in place
//   k_row      load rows -> x inter-pass twiddle -> N2-point DIF in LDS ->
//              x exp(-i beta dz) / PMD waveplates (both polarisations of one
//              frequency co-resident) -> N2-point DIT -> x conj twiddle -> store
//   k_row256r  the same for 256-point rows of dual-polarisation plans with every radix level in
//              registers: one wave = 2 rows x 2 polarisations, two LDS exchanges, no workgroup
//              barrier, the multiplier shared between the wave's halves (k_row4k: 4096-point rows)
//   k_col_inv  load column tile -> N1-point DIT -> x exp(-alpha dz/2)/N ->
//              wave-shuffle max of |ux|^2+|uy|^2 -> one atomicMax per workgroup
//   k_ctrl     one lane per frame: nextstep + checkstep + last-step rule; the
//              data-dependent step loop never round-trips to the host
//   k_colx16   dual-polarisation plans with 256-row tiles: k_col_inv of step s, the step
//              controller and k_col_fwd of step s+1 in ONE launch on a register-resident
//              tile (the step is then two sweeps), the tiles of a frame meeting at a barrier
//
// Twiddles of the in-LDS transforms are staged in LDS; frames of a batch carry
// their own step state, so a batch of Monte-Carlo realisations advances in
// lock-step launches while every frame keeps the reference's own step sequence.
// (Variants that were measured and lost -- persistent prefetching sweeps, register-blocked rows,
// an LDS-resident fused sweep -- live in the history and in profiles/r01_notes.md, not here.)
#include "ssfm_plan.h"
#include "ssfm_ctrl.h"
#include "plx_gateway.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
using namespace plxs;

static int ilog2(int64_t v)
{
    int l = 0;
    while (((int64_t)1 << l) < v) l++;
    return l;
}

static void free_plan(plx_ssfm *P)
{
    if (!P) return;
    hipFree(P->d_betat); hipFree(P->d_db1); hipFree(P->d_gam); hipFree(P->d_brf); hipFree(P->d_psum);
    hipFree(P->d_tpass); hipFree(P->d_tw1); hipFree(P->d_tw2); hipFree(P->d_ctab); hipFree(P->d_ctl); hipFree(P->d_umax);
    hipFree(P->d_dzlist); hipFree(P->d_dzlog); hipFree(P->d_tw2c); hipFree(P->d_twmid);
    hipFree(P->d_ndone); hipFree(P->d_slots); hipFree(P->d_mbox); hipFree(P->d_active); hipFree(P->d_e1); hipFree(P->d_e2);
    if (P->h_ndone) hipHostFree(P->h_ndone);
    if (P->ev) hipEventDestroy(P->ev);
    for (hipEvent_t e : P->evfree) hipEventDestroy(e);
    for (auto &r : P->prof_pending) for (hipEvent_t e : r.ev) hipEventDestroy(e);
    for (int k = 0; k < 2; k++) {
        if (P->h_brf[k]) hipHostFree(P->h_brf[k]);
        if (P->brf_ev[k]) hipEventDestroy(P->brf_ev[k]);
    }
    delete P;
}

static void half_table(std::vector<cplx> &t, int M)
{
    t.resize(M / 2 > 0 ? M / 2 : 1);
    for (int k = 0; k < M / 2; k++) {
        long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)M;
        t[k] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
}


// Plan-time tuning (include/polmux_hip.h, plx_ssfm_tuning): the geometry switches the tests use to reach other splits and
// kernels, and the A/B switches of shipped choices.  Nothing here is read from the environment except the two deployment knobs
// plx_ssfm_tuning_defaults documents (PLX_SSFM_NO_FUSE, PLX_SSFM_BARRIER_TIMEOUT_MS).
namespace {
plx_ssfm_tuning g_override;            // plx_ssfm_tuning_override: the process-wide default of plans created without a tuning
bool g_have_override = false;
} // namespace

extern "C" int plx_ssfm_tuning_defaults(plx_ssfm_tuning *t)
{
    if (!t) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_tuning_defaults: null argument");
    std::memset(t, 0, sizeof(*t));
    t->size = (uint32_t)sizeof(*t);
    t->p1 = -1; t->logW = -1; t->col_threads = -1; t->store_late = -1;
    t->row_rev = 1; t->rowr = 1; t->rowsm = 1; t->row256_split = 1; t->row4k_split = 1; t->rowg_split = 1;
    t->barrier_timeout_ms = 500.0;
    // the two deployment knobs: several processes on one GPU, and the frame barrier's patience
    if (const char *e = getenv("PLX_SSFM_NO_FUSE")) t->no_fuse = atoi(e);
    if (const char *e = getenv("PLX_SSFM_BARRIER_TIMEOUT_MS")) t->barrier_timeout_ms = atof(e);
    return PLX_OK;
}

extern "C" int plx_ssfm_tuning_override(const plx_ssfm_tuning *t)
{
    if (t && t->size != sizeof(plx_ssfm_tuning)) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_tuning_override: fill the struct with plx_ssfm_tuning_defaults first");
    g_have_override = t != nullptr;
    if (t) g_override = *t;
    return PLX_OK;
}

extern "C" int plx_ssfm_create(plx_ssfm **out, const plx_ssfm_desc *desc) { return plx_ssfm_create_tuned(out, desc, 0u, nullptr); }

extern "C" int plx_ssfm_create_ex(plx_ssfm **out, const plx_ssfm_desc *desc, uint32_t flags) { return plx_ssfm_create_tuned(out, desc, flags, nullptr); }

extern "C" int plx_ssfm_create_tuned(plx_ssfm **out, const plx_ssfm_desc *desc, uint32_t flags, const plx_ssfm_tuning *tuning)
{
    if (!out || !desc) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create: null argument");
    if (flags & ~(uint32_t)PLX_SSFM_SHARE_DEVICE) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create_ex: unknown flag");
    *out = nullptr;
    const int64_t N = desc->nfft;
    const int p = ilog2(N);
    if (N < 256 || ((int64_t)1 << p) != N || p > 20)
        PLX_FAIL(PLX_ERR_UNSUPPORTED, "plx_ssfm_create: nfft must be a power of two in [256, 2^20]");
    if (desc->nfc < 1 || desc->max_frames < 1) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create: nfc and max_frames must be >= 1");
    if (!desc->gam || !desc->betat) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create: gam and betat are required");
    if (desc->dual_pol && desc->fls[3] && desc->nfc > 1)
        PLX_FAIL(PLX_ERR_REFERENCE, "The CNLSE with separate fields is not yet implemented"); // fiber.m:854
    if (desc->dual_pol && desc->fls[3] && desc->nfc == 1) { /* xpm flag is forced to 0 for one field, :224 */ }
    if (desc->nplates < 1) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create: nplates must be >= 1");
    if (tuning && tuning->size != sizeof(plx_ssfm_tuning)) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create_tuned: fill the struct with plx_ssfm_tuning_defaults first");
    plx_ssfm_tuning tune;
    if (tuning) tune = *tuning;
    else if (g_have_override) tune = g_override;
    else plx_ssfm_tuning_defaults(&tune);
    if (flags & PLX_SSFM_SHARE_DEVICE) tune.no_fuse = 1;      // the barrier-free three-sweep step: no co-residency requirement

    plx_ssfm *P = new plx_ssfm();
    P->d = *desc;
    P->flags = flags;
    P->p = p;
    // Four-step split N = N1 x N2.  The column tile is N1 rows x T complex (T = W columns per
    // polarisation x npol) and is kept at <= 64 KiB so two workgroups share a CU; a wider, shorter
    // tile means longer contiguous row segments in HBM (W*16 B per polarisation).  The row pass
    // holds npol x N2 complex (+ twiddles) in LDS, which bounds N2 at 2048 for dual-pol frames.
    int logW;
    {
        const int npol = desc->dual_pol ? 2 : 1;
        logW = desc->dual_pol ? 3 : 4;                       // 8 (dual) / 16 (scalar) columns per tile (measured best)
        int p1 = 12 - (logW + (npol == 2 ? 1 : 0));          // N1 * T = 4096 complex = 64 KiB
        // 2^20-sample dual-polarisation frames -- one field or several 'sepfields' channels, fused or three-sweep step, with or
        // without PMD -- keep the 256-row column tile and take 4096-point rows instead (k_row4k, compact twiddle table: one
        // polarisation per row workgroup and two workgroups per CU, or with PMD both polarisations in one workgroup of twice
        // the size); everything else stops at 2048-point rows and gets taller column tiles
        // (scalar plans take the same split: k_row4k on the rows of the one field)
        const bool long_rows = !tune.no_row_split && !tune.short_rows;
        const int p2max = long_rows ? 12 : 11;
        if (p - p1 > p2max) p1 = p - p2max;                  // large frames: taller tiles instead
        if (p1 > p - 4) p1 = p - 4;                          // keep N2 >= 16
        if (p1 < 2) p1 = 2;
        while ((((size_t)1 << p1) << (logW + (npol == 2 ? 1 : 0))) * sizeof(cplx) > 128 * 1024 && logW > 3) logW--;
        if (tune.p1 >= 2 && tune.p1 <= p - 4) p1 = tune.p1;
        if (tune.logW >= 2 && tune.logW <= 6) logW = tune.logW;
        while (((int64_t)1 << (p - p1)) < ((int64_t)1 << logW)) logW--;  // tile not wider than a row
        P->p1 = p1;
    }
    P->p2 = p - P->p1;
    P->N = (size_t)N;
    const int N1 = 1 << P->p1, N2 = 1 << P->p2;
    const int nfc = desc->nfc, F = desc->max_frames;
    SsfmArgs &a = P->a;
    std::memset(&a, 0, sizeof(a));
    a.p1 = P->p1; a.p2 = P->p2; a.nfc = nfc; a.dual = desc->dual_pol ? 1 : 0;
    a.logW = logW; a.W = 1 << a.logW;
    a.logT = a.logW + (a.dual ? 1 : 0); a.T = 1 << a.logT;
    // rows per workgroup in the row pass: >= one 16-point register block per thread
    {
        int R = 1, npol = a.dual ? 2 : 1;
        while (R * npol * (N2 / 16) < ROW_THREADS / 2 && R * 2 <= N1) R *= 2;   // measured: 2 rows x 2 pols at N2 = 256
        a.R = R; a.logR = ilog2(R);
        // row pass: ~8 points per thread (128 threads for 2 rows x 2 polarisations x 256 points); long rows leave room
        // for only one or two workgroups per CU, so those get proportionally more waves (up to 1024 threads)
        int rowthr = ROW_THREADS;
        const int64_t pts = (int64_t)npol * R * N2;
        while (rowthr < 1024 && (int64_t)rowthr * 8 < pts) rowthr *= 2;
        P->row_threads = rowthr;
    }
    // Long rows leave room for a single dual-polarisation workgroup per CU.  Without PMD the two polarisations only
    // share the multiplier, so each gets its own workgroup (the scalar form of the row pass, R = 1): half the LDS,
    // 2-3 workgroups per CU.
    P->tw_compact = P->p2 >= 12 ? 1 : 0;
    if (a.dual && (N2 >= 2048 && !tune.no_row_split)) {   // measured: 2^20 frames 74 -> 66 ms; at N2 = 1024 it loses (47 -> 52)
        P->row_split = 1;
        P->rs_threads = N2 / 8 < ROW_THREADS ? ROW_THREADS : (N2 / 8 > 1024 ? 1024 : N2 / 8);
        P->rs_lds = ((size_t)(N2 + N2 / 16) + (P->tw_compact ? N2 / 8 + 4 + 16 + 160 + PLX_CTAB : N2 / 2)) * sizeof(cplx);   // (+16: k_row4k's bk, +160: its padded W_256 table, + the unit-circle table)
    }
    if (P->tw_compact && !a.dual && !tune.no_row_split) P->rs_lds = ((size_t)(N2 + N2 / 16) + N2 / 8 + 4 + 16 + 160 + PLX_CTAB) * sizeof(cplx);
    if (P->tw_compact && !P->row_split && a.dual) {
        free_plan(P);
        PLX_FAIL(PLX_ERR_UNSUPPORTED, "plx_ssfm_create: 4096-point rows need the one-polarisation row pass (PLX_SSFM_NO_ROW_SPLIT is set)");
    }
    if (P->tw_compact) {
        P->row_pair4k = desc->fls[1] ? 1 : 0;
        P->rs_lds_pair = P->rs_lds + (size_t)(N2 + N2 / 16) * sizeof(cplx);
    }
    a.spm = desc->fls[2]; a.xpm = desc->fls[3]; a.manakov = desc->manakov ? 1 : 0; a.pmd = desc->fls[1] ? 1 : 0;
    a.nplates = desc->nplates;
    a.alphalin = desc->alphalin; a.Lf = desc->length; a.dzmax = desc->dzmaxt; a.dphimax = desc->dphimaxt;
    a.lcorr = desc->length / desc->nplates; // fiber.m:507
    a.invN = 1.0 / (double)N;
    a.spin_ticks = (long long)(tune.barrier_timeout_ms * 1e5);
    a.safe_land = tune.safe_landing;
    a.row_rev = tune.row_rev;
    // (the mailbox entries of k_colx16 pack frame + 1 and iteration + 1 into 22-bit fields)
    if ((int64_t)desc->max_frames + 4 >= ((int64_t)1 << 22)) { free_plan(P); PLX_FAIL(PLX_ERR_UNSUPPORTED, "plx_ssfm_create: max_frames must be below 2^22 - 4"); }

    // ---- tables: spectral multipliers in the order the row pass sees them ----
    std::vector<double> bt((size_t)nfc * N), d1;
    std::vector<cplx> tp((size_t)N);
    const bool have_db1 = desc->db1 != nullptr && a.dual;
    if (have_db1) d1.resize((size_t)nfc * N);
    for (int j = 0; j < N1; j++) {
        const unsigned k1 = plx_bitrev((unsigned)j, P->p1);
        for (int i = 0; i < N2; i++) {
            const unsigned k2 = plx_bitrev((unsigned)i, P->p2);
            const size_t k = (size_t)k1 + (size_t)N1 * k2, pos = (size_t)j * N2 + i;
            for (int c = 0; c < nfc; c++) {
                // phases are kept in TURNS (rad / 2 pi) for the exact range reduction of cexp_neg_turns
                bt[(size_t)c * N + pos] = desc->betat[(size_t)c * N + k] * kInv2Pi;
                if (have_db1) d1[(size_t)c * N + pos] = desc->db1[(size_t)c * N + k] * kInv2Pi;
            }
            const uint64_t e = ((uint64_t)i * k1) & (uint64_t)(N - 1); // n2*k1 mod N
            long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)e / (long double)N;
            tp[pos] = make_double2((double)cosl(ang), (double)sinl(ang));
        }
    }
    std::vector<cplx> t1, t2;
    half_table(t1, N1);
    if (P->tw_compact) {   // W_N2^{4k}, k < N2/8, then W_N2^0..3 (plx_fft.h, Tw4096)
        t2.resize(N2 / 8 + 4);
        for (int k = 0; k < N2 / 8 + 4; k++) {
            const int e = k < N2 / 8 ? 4 * k : k - N2 / 8;
            long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)e / (long double)N2;
            t2[k] = make_double2((double)cosl(ang), (double)sinl(ang));
        }
    } else {
        half_table(t2, N2);
    }
    std::vector<double> gam(nfc);
    for (int c = 0; c < nfc; c++) gam[c] = (a.dual && a.manakov) ? desc->gam[c] * 8 / 9 : desc->gam[c]; // :499-501

#define UP(dst, vec, T)                                                                             \
    do {                                                                                            \
        if (hipMalloc((void **)&(dst), (vec).size() * sizeof(T)) != hipSuccess ||                   \
            hipMemcpy((dst), (vec).data(), (vec).size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { \
            free_plan(P);                                                                           \
            PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_create: device allocation/upload failed");              \
        }                                                                                           \
    } while (0)
    UP(P->d_betat, bt, double);
    if (have_db1) UP(P->d_db1, d1, double);
    UP(P->d_tpass, tp, cplx);
    UP(P->d_tw1, t1, cplx);
    UP(P->d_tw2, t2, cplx);
    {
        std::vector<cplx> ctv(PLX_CTAB);
        for (int k = 0; k < PLX_CTAB; k++) {
            const long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)PLX_CTAB;
            ctv[k] = make_double2((double)cosl(ang), (double)-sinl(ang));
        }
        UP(P->d_ctab, ctv, cplx);
    }
    // k_rowsm: register-form row pass for rows of 32 / 64 / 128 points (one wave = 64 / R row-polarisations)
    // (measured, fraction of 8 TB/s: scalar plans 0.55 / 0.67 / 0.68 at 32 / 64 / 128 points against k_row's 0.49 / 0.50 / 0.45;
    //  dual-polarisation plans 0.64 / 0.69 / 0.70 against 0.70 / 0.70 / 0.60 -- k_row's wider workgroups win the short dual rows,
    //  so those take it at 128 points only; PLX_SSFM_ROWSM=2 forces it wherever it applies: tests)
    const int rowsm_min = a.dual ? (tune.rowsm == 2 ? 5 : 7) : 5;
    if (tune.rowr && tune.rowsm && !desc->fls[1] && P->p2 >= rowsm_min && P->p2 <= 7 && (N1 * (a.dual ? 2 : 1)) % (64 / (N2 / 16)) == 0) {
        const long double tau = -2.0L * 3.14159265358979323846264338327950288L;
        std::vector<cplx> tm(7 * 16, make_double2(1.0, 0.0));
        const int R = N2 / 16;
        auto put = [&](int q, int j, int e, int m) { tm[16 * q + j] = make_double2((double)cosl(tau * e / m), (double)sinl(tau * e / m)); };
        for (int j = 0; j < 16; j++) {
            if (R == 2) put(0, j, j, 32);
            const int q0 = R == 8 ? 4 : 0;
            if (R >= 4) for (int q = 0; q < 3; q++) put(q0 + q, j, (q + 1) * j, 64);
            if (R == 8) for (int q = 0; q < 4; q++) put(q, j, j + 16 * q, 128);
        }
        UP(P->d_twmid, tm, cplx);
        if (allow_lds(rowsm_kernel(P->p2, !a.dual), ROWSM_LDS) == hipSuccess) P->rowsm = 1;
    }
    // k_rowreg: register-form row pass for dual-polarisation plans without PMD whose rows have 512, 1024 or 2048 points
    if (tune.rowr && P->p2 >= 9 && P->p2 <= 11 && N1 >= (ROWG_THREADS / (N2 / 16)) / (a.dual ? 2 : 1)) {
        const long double tau = -2.0L * 3.14159265358979323846264338327950288L;
        std::vector<cplx> tc(N2 / 8 + 4), tm(7 * 16, make_double2(1.0, 0.0));
        for (int k = 0; k < N2 / 8 + 4; k++) {
            const int e = k < N2 / 8 ? 4 * k : k - N2 / 8;
            tc[k] = make_double2((double)cosl(tau * e / N2), (double)sinl(tau * e / N2));
        }
        const int R = N2 / 256;
        auto put = [&](int q, int j, int e, int m) { tm[16 * q + j] = make_double2((double)cosl(tau * e / m), (double)sinl(tau * e / m)); };
        for (int j = 0; j < 16; j++) {
            if (R == 2) put(0, j, j, 32);
            const int q0 = R == 8 ? 4 : 0;
            if (R >= 4) for (int q = 0; q < 3; q++) put(q0 + q, j, (q + 1) * j, 64);
            if (R == 8) for (int q = 0; q < 4; q++) put(q, j, j + 16 * q, 128);
        }
        UP(P->d_tw2c, tc, cplx);
        UP(P->d_twmid, tm, cplx);
        const size_t lds_whole = ROWG_LDS((size_t)N2), lds_split = ROWG_LDS_SPLIT((size_t)N2);
        hipError_t e = allow_lds(rowreg_kernel(P->p2, false, false, false), lds_whole);
        if (e == hipSuccess) e = allow_lds(rowreg_kernel(P->p2, true, false, false), lds_whole);
        if (e == hipSuccess) e = allow_lds(rowreg_kernel(P->p2, false, true, false), lds_whole);
        if (e == hipSuccess) P->rowreg = 1;
        // rows of 512 / 1024 points without PMD: the exchanges in real / imaginary halves, three workgroups per CU
        // (PLX_SSFM_ROWG_SPLIT=0: the whole-sample exchange, A/B and tests)
        if (P->rowreg && tune.rowg_split) {
            if (allow_lds(rowreg_kernel(P->p2, false, !a.dual, true), lds_split) == hipSuccess) P->rowg_split = 1;
            if (P->rowg_split && a.dual && allow_lds(rowreg_kernel(P->p2, true, false, true), lds_split) == hipSuccess) P->rowg_pair_split = 1;
        }
    }
    UP(P->d_gam, gam, double);
#undef UP
    bool ok = hipMalloc((void **)&P->d_ctl, sizeof(FrameCtl) * F) == hipSuccess &&
              hipMalloc((void **)&P->d_umax, sizeof(unsigned long long) * F * nfc) == hipSuccess &&
              hipMalloc((void **)&P->d_ndone, 64) == hipSuccess &&
              hipMalloc((void **)&P->d_active, sizeof(int) * (size_t)F) == hipSuccess &&
              hipHostMalloc((void **)&P->h_ndone, 64, hipHostMallocDefault) == hipSuccess &&
              hipEventCreateWithFlags(&P->ev, hipEventDisableTiming) == hipSuccess;
    if (ok && !a.dual && a.xpm) ok = hipMalloc((void **)&P->d_psum, sizeof(double) * (size_t)F * N) == hipSuccess;
    if (!ok) { free_plan(P); PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_create: device allocation failed"); }
    a.betat_p = P->d_betat; a.db1_p = P->d_db1; a.tpass = P->d_tpass; a.tw1 = P->d_tw1; a.tw2 = P->d_tw2; a.ctab = P->d_ctab; a.tw2c = P->d_tw2c; a.twmid = P->d_twmid;
    a.gam = P->d_gam; a.ctl = P->d_ctl; a.umax = P->d_umax; a.ndone = P->d_ndone; a.psum = P->d_psum;
    P->h_ctl.resize(F);

    P->lds_col = (((size_t)N1 << a.logT) + N1 / 2) * sizeof(cplx) + 32 * sizeof(double) + 8 * sizeof(FrameCtl) + 128 + COLX_NFC * sizeof(double) + 128 * sizeof(cplx);   // (128: CtrlK; 128 cplx: k_colx16's negated W_256 table)
    // [stamps:lds]
    P->lds_row = ((size_t)(a.dual ? 2 : 1) * a.R * (N2 + N2 / 16) + N2 / 2) * sizeof(cplx);
    P->col_threads = P->lds_col > 80 * 1024 ? 1024 : 512;   // measured: 512-thread column workgroups (2 per CU, 16 waves) beat
                                                            // 256 by 3-12 %; tall tiles of large frames: one workgroup per CU, 16 waves
    if (tune.col_threads == 128 || tune.col_threads == 256 || tune.col_threads == 512 || tune.col_threads == 1024) P->col_threads = tune.col_threads;
    if (allow_lds(colx16_kernel(true), P->lds_col) != hipSuccess || allow_lds(colx16_kernel(false), P->lds_col) != hipSuccess || allow_lds(col_fwd_kernel(), P->lds_col) != hipSuccess ||
        allow_lds(col_inv_kernel(), P->lds_col) != hipSuccess ||
        (!P->tw_compact && allow_lds(row_kernel(), P->lds_row > P->rs_lds ? P->lds_row : P->rs_lds) != hipSuccess) ||
        (P->tw_compact && (allow_lds(row4k_kernel(false, false), P->rs_lds) != hipSuccess || allow_lds(row4k_kernel(true, false), P->rs_lds_pair) != hipSuccess ||
                           allow_lds(row4k_kernel(false, true), P->rs_lds) != hipSuccess))) {
        free_plan(P);
        PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_create: cannot reserve LDS for the transform kernels");
    }
    P->row4k_split = (P->tw_compact && tune.row4k_split) ? 1 : 0;
    if (tune.rowr && a.dual && a.p1 == 8 && a.p2 == 8 && !P->row_split &&
        allow_lds(row256_kernel(a.pmd != 0, false, false), ROWR_LDS) == hipSuccess) P->rowr = 1;
    if (tune.rowr && !a.dual && a.p1 == 8 && a.p2 == 8 && allow_lds(row256_kernel(false, true, false), ROWR_LDS_SC) == hipSuccess) P->rowr = 1;
    if (P->rowr && a.dual && a.pmd && tune.row256_split && allow_lds(row256_kernel(true, false, true), ROWR_LDS) == hipSuccess) P->row256_split = 1;
    // Fused column sweep (k_colx16): the inverse column pass of step s, the step controller and the forward column
    // pass of step s+1 in ONE launch on a register/LDS-resident tile (2 sweeps over HBM per step instead of 3), for
    // dual-polarisation plans with 256 x (8+8) column tiles.  The tiles of a frame meet at a barrier inside the
    // launch, so all of them must be resident together: the grid is sized from the runtime's own occupancy answer
    // for this kernel (block size and dynamic LDS as launched), a multiple of the tiles per frame; a plan whose
    // frame does not fit the chip that way takes the barrier-free three-sweep step.
    // (scalar plans: the same sweep on sixteen columns of the one field -- not with XPM, whose Kerr step needs the other
    //  channels' powers at the same sample, i.e. other workgroups' tiles)
    if (((a.dual && a.W == 8) || (!a.dual && a.W == 16 && !desc->fls[3])) && !tune.no_fuse && a.p1 == 8 && nfc <= COLX_NFC) {
        int ncu = 256;
        {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
                ncu = v;
        }
        const int tiles_pf = nfc * (N2 / a.W);
        const int per_cu = blocks_per_cu(colx16_kernel(a.dual != 0), 256, P->lds_col);
        const int cap = ncu * per_cu;
        if (tiles_pf <= cap) {
            P->fused = 1;
            P->tiles_pf = tiles_pf;
            P->fused_grid = (cap / tiles_pf) * tiles_pf;
            a.store_late = tune.store_late >= 0 ? (tune.store_late ? 1 : 0) : (P->fused_grid > tiles_pf ? 1 : 0);
            const int mstride = F + 4;       // (iterations of a team in a launch <= frames listed; its first workgroup posts two ahead)
            P->mbox_bytes = sizeof(unsigned long long) * ((size_t)mstride * (P->fused_grid / tiles_pf) + 1);
            if (hipMalloc((void **)&P->d_slots, sizeof(unsigned long long) * 2 * (size_t)F * tiles_pf) != hipSuccess ||
                hipMalloc((void **)&P->d_mbox, P->mbox_bytes) != hipSuccess) {
                free_plan(P);
                PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_create: device allocation failed");
            }
            a.slots = P->d_slots;
            a.mbox = P->d_mbox;
            a.mbox_stride = mstride;
            a.grab = (int *)(P->d_mbox + (size_t)mstride * (P->fused_grid / tiles_pf));
        }
    }
    // PMD plans: is db1 linear in the signed frequency index (fiber.m:358)?  Then the trunk phases factor into row x column
    // phasors (SsfmArgs::e1tab) and the row pass does one complex product per bin and trunk instead of an exponential.
    if (a.pmd && have_db1 && !tune.no_pmd_tab) {
        const double D = desc->db1[1] * kInv2Pi;       // turns per unit of m (k = 1 <-> m = 1)
        bool lin = D != 0.0 && N >= 4;
        for (int c = 0; c < nfc && lin; c++)
            for (int64_t k = 0; k < N && lin; k++) {
                const double m = (double)(k < N / 2 ? k : k - N);
                if (fabs(desc->db1[(size_t)c * N + k] * kInv2Pi - D * m) > 4e-15 * fabs(D) * (double)N) lin = false;
            }
        if (lin) {
            const int tmax = (int)ceil(desc->dzmaxt / a.lcorr) + 2;
            const size_t n1 = (size_t)F * tmax * N1, n2 = (size_t)F * tmax * N2;
            if (tmax <= 64 && (n1 + n2) * sizeof(cplx) <= ((size_t)4 << 30)) {       // (bounded: 27 trunks at 100 plates and dzmax = L / 4)
                if (hipMalloc((void **)&P->d_e1, n1 * sizeof(cplx)) != hipSuccess || hipMalloc((void **)&P->d_e2, n2 * sizeof(cplx)) != hipSuccess) {
                    free_plan(P);
                    PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_create: device allocation failed (trunk phasor tables)");
                }
                a.e1tab = P->d_e1; a.e2tab = P->d_e2; a.d1slope = D; a.tmax = tmax;
            }
        }
    }
    if (a.pmd && !a.dual) { free_plan(P); PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_create: PMD needs a dual-polarisation plan"); }
    if (!a.pmd) { // fiber.m:291-297: birefringence off
        double z = 0;
        int rc = plx_ssfm_set_birefringence(P, &z, &z, &z, 1);
        if (rc) { free_plan(P); return rc; }
    }
    *out = P;
    return PLX_OK;
}

extern "C" int plx_ssfm_destroy(plx_ssfm *P)
{
    free_plan(P);
    return PLX_OK;
}

// Waveplate tables.  Device storage is sized once for max_frames sets; uploads go through two pinned staging
// slots and are stream-ordered (the copy lands after whatever propagate call is still reading the old table on
// that stream and before the next one), so a Monte-Carlo loop can draw fresh birefringence for batch i+1 while
// batch i is still in flight elsewhere on the GPU -- no device-wide synchronisation.
static int set_brf(plx_ssfm *P, const double *db0, const double *theta, const double *epsilon, int nsets, hipStream_t st,
                   bool wait)
{
    if (!P || !db0 || !theta || !epsilon) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_set_birefringence: null argument");
    if (nsets < 1 || (nsets != 1 && nsets > P->d.max_frames)) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_set_birefringence: more sets than frames");
    const int np = P->d.nplates;
    const size_t cap = (size_t)P->d.max_frames * np * BRF_STRIDE, cnt = (size_t)nsets * np * BRF_STRIDE;
    if (!P->d_brf) PLX_HIP(hipMalloc((void **)&P->d_brf, cap * sizeof(double)));
    const int slot = P->brf_slot;
    P->brf_slot ^= 1;
    if (!P->h_brf[slot]) {
        PLX_HIP(hipHostMalloc((void **)&P->h_brf[slot], cap * sizeof(double), hipHostMallocDefault));
        PLX_HIP(hipEventCreateWithFlags(&P->brf_ev[slot], hipEventDisableTiming));
    } else {
        PLX_HIP(hipEventSynchronize(P->brf_ev[slot]));   // the upload that last used this slot has executed
    }
    double *t = P->h_brf[slot];
    for (int sidx = 0; sidx < nsets; sidx++)
        for (int n = 0; n < np; n++) {
            const size_t i = (size_t)sidx * np + n;
            // matR = matRth*matRepsilon, fiber.m:910-912; the kernel needs S = matR * sigma3 * matR' (see pmd_trunks)
            const double ct = cos(theta[i]), sn = sin(theta[i]), ce = cos(epsilon[i]), se = sin(epsilon[i]);
            const double r11x = ct * ce, r11y = -sn * se, r12x = -sn * ce, r12y = ct * se;
            const double r21x = sn * ce, r21y = ct * se, r22x = ct * ce, r22y = sn * se;
            double *m = &t[i * BRF_STRIDE];
            m[0] = (r11x * r11x + r11y * r11y) - (r12x * r12x + r12y * r12y);                       // S11 = |R11|^2 - |R12|^2
            m[1] = (r11x * r21x + r11y * r21y) - (r12x * r22x + r12y * r22y);                       // S12 = R11 conj(R21) - R12 conj(R22)
            m[2] = (r11y * r21x - r11x * r21y) - (r12y * r22x - r12x * r22y);
            m[3] = db0[i] * kInv2Pi; // turns, like betat_p / db1_p
        }
    PLX_HIP(hipMemcpyAsync(P->d_brf, t, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    PLX_HIP(hipEventRecord(P->brf_ev[slot], st));
    if (wait) PLX_HIP(hipStreamSynchronize(st));
    P->a.brf = P->d_brf;
    P->a.brf_per_frame = nsets > 1 ? 1 : 0;
    P->brf_sets = nsets;
    return PLX_OK;
}

extern "C" int plx_ssfm_set_birefringence(plx_ssfm *P, const double *db0, const double *theta,
                                          const double *epsilon, int nsets)
{
    return set_brf(P, db0, theta, epsilon, nsets, nullptr, true);
}

extern "C" int plx_ssfm_set_birefringence_dev(plx_ssfm *P, const double *db0, const double *theta,
                                              const double *epsilon, int nsets, void *stream)
{
    return set_brf(P, db0, theta, epsilon, nsets, (hipStream_t)stream, false);
}

// The row pass of one step / filter pass: one launch over both polarisations, or -- long rows without PMD -- the
// one-polarisation form twice (the polarisations only share the multiplier there).
static void launch_row(plx_ssfm *P, const SsfmArgs &a, unsigned FC, hipStream_t st)
{
    const int N1 = 1 << a.p1, N2 = 1 << a.p2;
    if (P->tw_compact && !a.dual) {              // scalar plan, 4096-point rows: one workgroup per row and frame-channel
        if (P->row4k_split) launch(row4k_kernel(false, true), dim3((unsigned)N1 * FC), dim3(256), P->rs_lds - 4352 * sizeof(double), st, a);
        else launch(row4k_kernel(false, false), dim3((unsigned)N1 * FC), dim3(256), P->rs_lds, st, a);
        return;
    }
    if (P->tw_compact && a.dual && (a.pmd || a.umat)) {      // the multiplier couples the polarisations: both rows in one workgroup
        launch(row4k_kernel(true, false), dim3((unsigned)N1 * FC), dim3(512), P->rs_lds_pair, st, a);
        return;
    }
    if (P->rowsm && !a.pmd && !a.umat) {
        const dim3 g((unsigned)(N1 * (a.dual ? 2 : 1) / (64 / (N2 / 16))), FC), bs(64);
        launch(rowsm_kernel(a.p2, !a.dual), g, bs, ROWSM_LDS, st, a);
        return;
    }
    if (P->rowreg && !a.dual) {                  // scalar plan: every row-polarisation of the workgroup is a row
        const dim3 g((unsigned)(N1 / (ROWG_THREADS / (N2 / 16))), FC), bs(ROWG_THREADS);
        if (P->rowg_split) launch(rowreg_kernel(a.p2, false, true, true), g, bs, ROWG_LDS_SPLIT((size_t)N2), st, a);
        else launch(rowreg_kernel(a.p2, false, true, false), g, bs, ROWG_LDS((size_t)N2), st, a);
        return;
    }
    if (P->rowreg && a.dual) {
        const unsigned gx = (unsigned)(N1 / ((ROWG_THREADS / (N2 / 16)) / 2));
        const dim3 g(gx, FC), bs(ROWG_THREADS);
        if (a.pmd && !a.umat && a.e1tab && P->rowg_split && P->rowg_pair_split)      // ... with phasor tables: the three-waves-per-SIMD form
            launch(rowreg_kernel(a.p2, true, false, true), g, bs, ROWG_LDS_SPLIT((size_t)N2), st, a);
        else if (a.pmd || a.umat)                // the multiplier couples the polarisations: lanes i and i + 32 hold X and Y
            launch(rowreg_kernel(a.p2, true, false, false), g, bs, ROWG_LDS((size_t)N2), st, a);
        else if (P->rowg_split)
            launch(rowreg_kernel(a.p2, false, false, true), g, bs, ROWG_LDS_SPLIT((size_t)N2), st, a);
        else
            launch(rowreg_kernel(a.p2, false, false, false), g, bs, ROWG_LDS((size_t)N2), st, a);
        return;
    }
    if (P->row_split && a.dual && !a.pmd) {
        SsfmArgs b = a;
        b.dual = 0; b.R = 1; b.logR = 0;
        const dim3 gs((unsigned)N1, FC), bs((unsigned)P->rs_threads);
        if (P->tw_compact) {                     // (both polarisations in one launch: one tail instead of two)
            if (P->row4k_split) launch(row4k_kernel(false, true), dim3(gs.x * gs.y * 2u), dim3(256), P->rs_lds - 4352 * sizeof(double), st, b);
            else launch(row4k_kernel(false, false), dim3(gs.x * gs.y * 2u), dim3(256), P->rs_lds, st, b);   // (rows x frame-channels x polarisations: decoded in the kernel)
            return;
        }
        for (int pol = 0; pol < 2; pol++) {
            if (pol) b.ux = a.uy;
            launch(row_kernel(), gs, bs, P->rs_lds, st, b);
        }
        return;
    }
    if (P->rowr && !a.dual && !a.force && !a.hmul) {
        launch(row256_kernel(false, true, false), dim3(64u, FC), dim3(ROWR_THREADS), ROWR_LDS_SC, st, a);
        return;
    }
    if (P->rowr && a.dual && !a.force && !a.hmul && !a.umat) {
        if (a.pmd && P->row256_split && a.e1tab) launch(row256_kernel(true, false, true), dim3(128u, FC), dim3(ROWR_THREADS), ROWR_LDS - 4 * 272 * sizeof(double), st, a);
        else launch(row256_kernel(a.pmd != 0, false, false), dim3(128u, FC), dim3(ROWR_THREADS), ROWR_LDS, st, a);
        return;
    }
    launch(row_kernel(), dim3((unsigned)(N1 / a.R), FC), dim3((unsigned)P->row_threads), P->lds_row, st, a);
}

// Read the event intervals of the step loops that have finished (see plx_ssfm::ProfRun) into k_ms / k_launches.
static int resolve_profiles(plx_ssfm *P)
{
    for (auto &r : P->prof_pending) {
        // ACTIVE launches only: the chunked loop also issues launches after every frame has finished (they return at
        // once).  Step s of the slowest frame is its (s+1)-th; the fused column sweep needs one more round to finish
        // the last step and write the field out.
        for (size_t i = 0; i + 1 < r.ev.size(); i++) {
            const int cls = r.cls[i], step = r.step[i];
            const bool active = (r.fused && cls == 0) ? step <= r.maxnc : step < r.maxnc;
            if (!active) continue;
            float ms = 0;
            PLX_HIP(hipEventElapsedTime(&ms, r.ev[i], r.ev[i + 1]));
            P->k_ms[cls] += ms;
            P->k_launches[cls]++;
        }
        for (hipEvent_t e : r.ev) P->evfree.push_back(e);
    }
    P->prof_pending.clear();
    return PLX_OK;
}

// The frames of one call through the whole step loop (fiber.m:518-552).
static int propagate_frames(plx_ssfm *P, cplx *d_ux, cplx *d_uy, int nframes, hipStream_t st)
{
    SsfmArgs a = P->a;
    const int nfc = a.nfc, N2 = 1 << a.p2;
    a.ux = d_ux;
    a.uy = d_uy;
    a.nframes = nframes;
    a.active = P->d_active;
    a.nactive = P->d_ndone + 2;
    unsigned FC = (unsigned)nframes * nfc;      // frame-channels launched: shrinks with the host's (lagging) view of the active list
    PLX_HIP(hipMemsetAsync(a.ctl, 0, sizeof(FrameCtl) * nframes, st));
    PLX_HIP(hipMemsetAsync(a.umax, 0, sizeof(unsigned long long) * FC, st));
    PLX_HIP(hipMemsetAsync(P->d_ndone, 0, 64, st));
    const bool fused = P->fused != 0;
    if (fused) { // the first fused launch also forms nextstep's initial maximum
        PLX_HIP(hipMemsetAsync(P->d_slots, 0xFF, sizeof(unsigned long long) * 2 * (size_t)nframes * P->tiles_pf, st));   // ~0 = "not arrived"; [parity][frame][tile]
        PLX_HIP(hipMemsetAsync(P->d_mbox, 0, P->mbox_bytes, st));                                                      // no entry posted, nothing claimed
    } else {
        unsigned gx = (unsigned)((P->N + 255) / 256);
        if (gx > 64) gx = 64;
        launch_umax(dim3(gx, FC), st, a);
    }
    const dim3 blk(256);
    const dim3 bcol((unsigned)P->col_threads);
    // Data-dependent trip count (fiber.m:518): steps are enqueued in chunks; the completed-frame counter and the
    // abort word of chunk k are read back while chunk k+1 executes.
    // The sweeps of a step cover the frames of the active list (k_compact, once per step); their grids follow the
    // host's last read-back of its length, an upper bound (frames only ever leave), workgroups beyond the list exit.
    // (the list is rebuilt before every step for batches of 64 frames and more, once per chunk for small ones, whose
    // steps are launch-bound: the sweeps skip a listed frame that has finished meanwhile)
    const bool compact_every_step = nframes >= 64;
    int chunk = 4, steps = 0;
    const int kMaxSteps = 1 << 19;      // (far beyond any physical span; also below the period of the mailbox tags of k_colx16)
    bool pending = false, aborted = false;
    // profiling: an event in front of every launch of the loop (and one after the last); intervals are attributed to
    // the kernel class that follows the event.  Classes: 0 k_colx16 / k_col_fwd, 1 k_row, 2 k_col_inv, 3 control.
    // (the events of a call that ends early -- a HIP failure, a frame-barrier time-out -- go back to the plan's free list)
    struct ProfGuard {
        plx_ssfm *P;
        plx_ssfm::ProfRun run;
        ~ProfGuard() { for (hipEvent_t e : run.ev) P->evfree.push_back(e); }
    } guard{P, {}};
    plx_ssfm::ProfRun &run = guard.run;
    auto mark = [&](int cls, int step) -> int {
        if (!P->profile) return PLX_OK;
        hipEvent_t e;
        if (!P->evfree.empty()) { e = P->evfree.back(); P->evfree.pop_back(); }
        else PLX_HIP(hipEventCreate(&e));
        run.ev.push_back(e); run.cls.push_back(cls); run.step.push_back(step);   // (owned by the guard from here on)
        PLX_HIP(hipEventRecord(e, st));
        return PLX_OK;
    };
#define PLX_MARK(cls, step) do { int rc_ = mark((cls), (step)); if (rc_) return rc_; } while (0)
    for (;;) {
        for (int sidx = 0; sidx < chunk; sidx++) {
            const dim3 gcol((unsigned)(N2 / a.W), FC);
            P->slots_launched += FC / nfc;
            if (fused) {
                const int tcx = (int)gcol.x, tct = (int)(gcol.x * FC);
                const dim3 gx((unsigned)(tct < P->fused_grid ? tct : P->fused_grid));
                if (compact_every_step || sidx == 0) {
                    PLX_MARK(3, steps + sidx);
                    launch_compact(a.ctl, nframes, P->d_active, P->d_ndone + 2, compact_every_step ? 1 : chunk, st);
                }
#ifdef PLX_EMU
                // the emulator must keep one frame's workgroups alive together (PLX_EMU_STARVE: a test starves the barrier)
                emu::g_concurrency = emu::starve_barriers() ? 1 : P->tiles_pf;
#endif
                a.round = steps + sidx;
                PLX_MARK(0, steps + sidx);
                launch(colx16_kernel(a.dual != 0), gx, blk, P->lds_col, st, a, tcx, P->tiles_pf);
#ifdef PLX_EMU
                emu::g_concurrency = 1;
#endif
                if (a.e1tab) {             // PMD: the step's trunk phasors, once per frame (between the controller and the row pass)
                    PLX_MARK(3, steps + sidx);
                    launch_pmd_tab(FC / nfc, st, a);
                }
                PLX_MARK(1, steps + sidx);
                launch_row(P, a, FC, st);
                P->row_launches++;
                continue;
            }
            PLX_MARK(3, steps + sidx);
            launch_ctrl(nframes, st, a);
            if (compact_every_step || sidx == 0)
                launch_compact(a.ctl, nframes, P->d_active, P->d_ndone + 2, compact_every_step ? 1 : chunk, st);
            if (!a.dual && a.xpm) {
                unsigned gx = (unsigned)((P->N + 255) / 256);
                if (gx > 256) gx = 256;
                launch_rowsum(dim3(gx, (unsigned)nframes), st, a);
            }
            if (a.e1tab) launch_pmd_tab(FC / nfc, st, a);
            PLX_MARK(0, steps + sidx);
            launch(col_fwd_kernel(), gcol, bcol, P->lds_col, st, a);
            PLX_MARK(1, steps + sidx);
            launch_row(P, a, FC, st);
            PLX_MARK(2, steps + sidx);
            launch(col_inv_kernel(), gcol, bcol, P->lds_col, st, a);
            P->row_launches++;
        }
        PLX_MARK(3, steps + chunk);     // closes the last interval of the chunk (the read-back below lands in class 3)
        if (steps == 0 && !P->prof_pending.empty()) {      // (the GPU has this call's first chunk to work on meanwhile)
            const int rc_ = resolve_profiles(P);
            if (rc_) return rc_;
        }
        steps += chunk;
        if (pending) {
            PLX_HIP(hipEventSynchronize(P->ev));
            if (P->h_ndone[1]) { aborted = true; break; }
            if (P->h_ndone[0] >= nframes) break;
            const unsigned live = (unsigned)(nframes - P->h_ndone[0]) * nfc;   // as of the previous chunk: an upper bound
            if (live < FC) FC = live;
        }
        PLX_HIP(hipMemcpyAsync(P->h_ndone, P->d_ndone, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
        PLX_HIP(hipEventRecord(P->ev, st));
        pending = true;
        if (chunk > 8) chunk = 4;                                     // (after a predicted first chunk)
        else if (chunk < (compact_every_step ? 8 : 16)) chunk *= 2;   // (small batches are launch-bound: longer chunks keep the queue fed)
        if (steps > kMaxSteps) PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_propagate_dev: step loop did not terminate");
    }
    PLX_HIP(hipMemcpyAsync(P->h_ctl.data(), a.ctl, sizeof(FrameCtl) * nframes, hipMemcpyDeviceToHost, st));
    PLX_HIP(hipMemcpyAsync(P->h_ndone, P->d_ndone, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    PLX_HIP(hipStreamSynchronize(st));
    P->slots_listed += P->h_ndone[3];
    if (aborted || P->h_ndone[1]) {
        // The field of this call is lost (it has been propagated in place up to the time-out).  The plan itself stays usable:
        // from now on it takes the barrier-free three-sweep step, which needs no co-residency.  The gateway tier, whose
        // pristine input is still in its pinned staging buffer, repeats the call that way at once (gateway_ssfm).
        P->barrier_timeouts++;
        P->fused = 0;
        PLX_FAIL(PLX_ERR_TIMEOUT, "plx_ssfm_propagate_dev: frame barrier timed out (the workgroups of a frame were not co-resident: "
                                  "another kernel holds the GPU); nothing was stored after the time-out and the field of this call is "
                                  "INVALID -- the plan now takes the barrier-free three-sweep step: restore the field and call again, "
                                  "or create such plans with plx_ssfm_create_ex(..., PLX_SSFM_SHARE_DEVICE)");
    }
    int maxnc = 0;
    for (int f = 0; f < nframes; f++) {
        P->frame_steps += P->h_ctl[f].ncycle + (fused ? 1 : 0);   // (the fused sweep's last round writes the field out)
        if (!P->h_ctl[f].done) PLX_FAIL(PLX_ERR_HIP, "plx_ssfm_propagate_dev: a frame did not reach the fibre end");
        P->sample_steps += (int64_t)P->h_ctl[f].ncycle * (int64_t)P->N * nfc;
        if (P->h_ctl[f].ncycle > maxnc) maxnc = P->h_ctl[f].ncycle;
    }
    if (!run.ev.empty()) {
        run.maxnc = maxnc; run.fused = fused;
        P->prof_pending.push_back(std::move(run));
        run.ev.clear();                 // (moved out: nothing left for the guard to return)
    }
#undef PLX_MARK
    return PLX_OK;
}

extern "C" int plx_ssfm_propagate_dev(plx_ssfm *P, double *d_ux, double *d_uy, int nframes, void *stream)
{
    if (!P || !d_ux) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_propagate_dev: null argument");
    if (nframes < 1 || nframes > P->d.max_frames) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_propagate_dev: nframes out of range");
    if (P->a.dual && !d_uy) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_propagate_dev: dual-polarisation plan needs d_uy");
    if (P->a.brf_per_frame && P->brf_sets < nframes)
        PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_propagate_dev: fewer birefringence sets than frames");
    P->slots_launched = 0; P->slots_listed = 0; P->row_launches = 0; P->sample_steps = 0; P->frame_steps = 0;
    const int rc = propagate_frames(P, (cplx *)d_ux, (cplx *)d_uy, nframes, (hipStream_t)stream);
    if (rc) return rc;
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_ssfm_set_step_sequence(plx_ssfm *P, const double *dz, int nsteps)
{
    if (!P || nsteps < 0 || (nsteps > 0 && !dz)) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_set_step_sequence: bad argument");
    if (nsteps > P->dzlist_cap) {
        if (P->d_dzlist) { (void)hipFree(P->d_dzlist); P->d_dzlist = nullptr; P->dzlist_cap = 0; }
        PLX_HIP(hipMalloc((void **)&P->d_dzlist, sizeof(double) * (size_t)nsteps));
        P->dzlist_cap = nsteps;
    }
    if (nsteps > 0) PLX_HIP(hipMemcpy(P->d_dzlist, dz, sizeof(double) * (size_t)nsteps, hipMemcpyHostToDevice));
    P->a.dzlist = nsteps > 0 ? P->d_dzlist : nullptr;
    P->a.ndz = nsteps;
    return PLX_OK;
}

extern "C" int plx_ssfm_log_steps(plx_ssfm *P, int max_steps)
{
    if (!P || max_steps < 0) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_log_steps: bad argument");
    if (P->d_dzlog) { (void)hipFree(P->d_dzlog); P->d_dzlog = nullptr; }
    P->a.dzlog = nullptr; P->a.logcap = 0;
    if (max_steps > 0) {
        PLX_HIP(hipMalloc((void **)&P->d_dzlog, sizeof(double) * (size_t)max_steps * P->d.max_frames));
        P->a.dzlog = P->d_dzlog; P->a.logcap = max_steps;
    }
    return PLX_OK;
}

extern "C" int plx_ssfm_step_sequence(plx_ssfm *P, int frame, double *dz, int max_steps)
{
    if (!P || !dz || frame < 0 || frame >= P->d.max_frames || max_steps < 0) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_step_sequence: bad argument");
    if (!P->d_dzlog) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_step_sequence: no log (plx_ssfm_log_steps)");
    const int n = max_steps < P->a.logcap ? max_steps : P->a.logcap;
    PLX_HIP(hipMemcpy(dz, P->d_dzlog + (size_t)frame * P->a.logcap, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return PLX_OK;
}

extern "C" int plx_ssfm_utilisation(plx_ssfm *P, int64_t *frame_steps, int64_t *slots_listed, int64_t *slots_launched)
{
    if (!P) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_utilisation: null plan");
    if (frame_steps) *frame_steps = P->frame_steps;
    if (slots_listed) *slots_listed = P->slots_listed;
    if (slots_launched) *slots_launched = P->slots_launched;
    return PLX_OK;
}

extern "C" int plx_ssfm_info(plx_ssfm *P, int32_t *info)
{
    if (!P || !info) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_info: null argument");
    info[0] = P->fused; info[1] = P->p1; info[2] = P->p2; info[3] = P->fused_grid; info[4] = P->tiles_pf;
    info[5] = P->col_threads; info[6] = (P->rowr || P->rowsm) ? ROWR_THREADS : P->rowreg ? ROWG_THREADS : (P->tw_compact ? (P->row_pair4k ? 512 : 256) : (P->row_split ? P->rs_threads : P->row_threads)); info[7] = (P->rowreg || P->rowsm) ? 2 : (P->row_pair4k ? 0 : ((P->tw_compact && !P->a.dual) ? 1 : P->row_split));
    return PLX_OK;
}

extern "C" int plx_ssfm_barrier_timeouts(plx_ssfm *P, int32_t *count, int rearm)
{
    if (!P) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_barrier_timeouts: null plan");
    if (count) *count = P->barrier_timeouts;
    if (rearm && !P->fused && P->fused_grid > 0 && P->tiles_pf > 0) { P->fused = 1; P->calls_unfused = 0; }
    return PLX_OK;
}

extern "C" int plx_ssfm_profile(plx_ssfm *P, int enable)
{
    if (!P) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_profile: null plan");
    P->profile = enable ? 1 : 0;
    return PLX_OK;
}

extern "C" int plx_ssfm_kernel_times(plx_ssfm *P, double *ms, int64_t *launches)
{
    if (!P || !ms || !launches) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_kernel_times: null argument");
    const int rc = resolve_profiles(P);
    if (rc) return rc;
    for (int k = 0; k < 4; k++) { ms[k] = P->k_ms[k]; launches[k] = P->k_launches[k]; P->k_ms[k] = 0; P->k_launches[k] = 0; }
    return PLX_OK;
}

// ---- the plan's FFT engine as a spectral filter (library-internal, plx_internal.h) ----
int plx_ssfm_filter_table(plx_ssfm *P, const double *h_re, const double *h_im, cplx **d_out)
{
    if (!P || !h_re || !d_out) PLX_FAIL(PLX_ERR_ARG, "filter table: null argument");
    const int N1 = 1 << P->p1, N2 = 1 << P->p2;
    std::vector<cplx> h(P->N);
    for (int j = 0; j < N1; j++) {
        const unsigned k1 = plx_bitrev((unsigned)j, P->p1);
        for (int i = 0; i < N2; i++) {
            const size_t k = (size_t)k1 + (size_t)N1 * plx_bitrev((unsigned)i, P->p2);
            h[(size_t)j * N2 + i] = make_double2(h_re[k], h_im ? h_im[k] : 0.0);
        }
    }
    cplx *d = nullptr;
    if (hipMalloc((void **)&d, P->N * sizeof(cplx)) != hipSuccess ||
        hipMemcpy(d, h.data(), P->N * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess) {
        if (d) (void)hipFree(d);
        PLX_FAIL(PLX_ERR_HIP, "filter table: device allocation/upload failed");
    }
    *d_out = d;
    return PLX_OK;
}

void plx_ssfm_geometry(const plx_ssfm *P, int *p1, int *p2) { *p1 = P->p1; *p2 = P->p2; }

int plx_ssfm_filter_dev(plx_ssfm *P, cplx *d_ux, cplx *d_uy, const cplx *d_hmul, int nframes, void *stream, const cplx *d_umat)
{
    if (!P || !d_ux || (!d_hmul && !d_umat)) PLX_FAIL(PLX_ERR_ARG, "filter: null argument");
    if (d_umat && (!P->a.dual || P->a.nfc != 1)) PLX_FAIL(PLX_ERR_ARG, "filter: matrix tables need a dual-polarisation single-field plan");
    if (nframes < 1 || nframes > P->d.max_frames) PLX_FAIL(PLX_ERR_ARG, "filter: nframes outside [1, max_frames]");
    if (P->a.dual && !d_uy) PLX_FAIL(PLX_ERR_ARG, "filter: dual-polarisation plan needs d_uy");
    hipStream_t st = (hipStream_t)stream;
    SsfmArgs b = P->a;
    b.ux = d_ux; b.uy = d_uy; b.nframes = nframes; b.hmul = d_hmul; b.umat = d_umat;
    b.force = 1; b.spm = 0; b.xpm = 0; b.pmd = 0; b.f_cur = 0; b.f_leff = 0; b.f_sc = b.invN;
    const int N1 = 1 << b.p1, N2 = 1 << b.p2;
    const unsigned FC = (unsigned)nframes * b.nfc;
    PLX_HIP(hipMemsetAsync(P->d_ctl, 0, sizeof(FrameCtl) * nframes, st));   // no frame is "done"
    PLX_HIP(hipMemsetAsync(P->d_ndone, 0, 64, st));
    const dim3 gcol((unsigned)(N2 / b.W), FC), grow((unsigned)(N1 / b.R), FC);
    launch(col_fwd_kernel(), gcol, dim3((unsigned)P->col_threads), P->lds_col, st, b);
    if (d_umat && !P->tw_compact && !P->rowreg) launch(row_kernel(), grow, dim3((unsigned)P->row_threads), P->lds_row, st, b);   // matrix tables couple the polarisations
    else launch_row(P, b, FC, st);
    launch(col_inv_kernel(), gcol, dim3((unsigned)P->col_threads), P->lds_col, st, b);
    PLX_HIP(hipGetLastError());
    return PLX_OK;
}

extern "C" int plx_ssfm_results(plx_ssfm *P, int nframes, double *firstdz, int32_t *ncycle)
{
    if (!P || nframes < 1 || nframes > P->d.max_frames) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_results: bad argument");
    for (int f = 0; f < nframes; f++) {
        if (firstdz) firstdz[f] = P->h_ctl[f].firstdz;
        if (ncycle) ncycle[f] = P->h_ctl[f].ncycle;
    }
    return PLX_OK;
}

extern "C" int plx_ssfm_stats(plx_ssfm *P, int64_t *row_pass_launches, int64_t *sample_steps)
{
    if (!P) PLX_FAIL(PLX_ERR_ARG, "plx_ssfm_stats: null plan");
    if (row_pass_launches) *row_pass_launches = P->row_launches;
    if (sample_steps) *sample_steps = P->sample_steps;
    return PLX_OK;
}

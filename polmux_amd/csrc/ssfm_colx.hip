// ssfm_colx.hip -- the fused column sweep k_colx16: the inverse column pass of step s, the step controller and the forward
// column pass of step s+1 in one persistent launch (fiber.m:512-551, :682-758, :776-874).
#include "ssfm_ctrl.h"
#include "ssfm_kernels.h"
using namespace plxs;

namespace {

__device__ __noinline__ double ctrl_head_call(const PLX_LDS_QUAL CtrlK *k, PLX_LDS_QUAL FrameCtl *rec, int count, double pmax, int f)
{
    if (!ctrl_head<true>(*k, f, *rec, true, pmax, count != 0)) return -1.0;
    return rec->leff;
}
__device__ __noinline__ void ctrl_tail_call(const PLX_LDS_QUAL CtrlK *k, PLX_LDS_QUAL FrameCtl *rec)
{
    ctrl_tail(*k, *rec);
}
// The rare full-range Kerr step of k_colx16 ('--s-' exact single step: |gamma Leff P| not small), on the tile parked in the
// exchange buffer, one lane per polarisation pair.  Out of line for the same reason as ctrl_head_call: the argument
// reduction constants of sincos must not live in the registers of the hot loop.
__device__ __noinline__ void kerr_full_range(int j, int t, double gamleff, int manakov)
{
    PLX_DYN_LDS(lds);
    cplx *s = (cplx *)lds;
    for (int k = 0; k < 16; k++) {
        cplx X = s[((j + 16 * k) << 4) + t], Y = s[((j + 16 * k) << 4) + t + 8];
        const double P = X.x * X.x + X.y * X.y + Y.x * Y.x + Y.y * Y.y;
        const cplx nl = cexpi(-gamleff * P);
        X = cmul(X, nl);
        Y = cmul(Y, nl);
        if (!manakov) {
            const double s3 = 2 * (X.x * Y.y - X.y * Y.x);
            double sp, cp;
            sincos(gamleff * s3 / 3, &sp, &cp);
            const cplx xx = make_double2(cp * X.x + sp * Y.x, cp * X.y + sp * Y.y);
            const cplx yy = make_double2(cp * Y.x - sp * X.x, cp * Y.y - sp * X.y);
            X = xx; Y = yy;
        }
        s[((j + 16 * k) << 4) + t] = X;
        s[((j + 16 * k) << 4) + t + 8] = Y;
    }
}

// ... and of the scalar form of the sweep: nl_step (:792-804, SPM only) on the lane's column of the parked tile.
__device__ __noinline__ void kerr_full_range_scalar(int j, int t, double gam, double leff)
{
    PLX_DYN_LDS(lds);
    cplx *s = (cplx *)lds;
    for (int k = 0; k < 16; k++) {
        const cplx X = s[((j + 16 * k) << 4) + t];
        const double pw = X.x * X.x + X.y * X.y;
        s[((j + 16 * k) << 4) + t] = cmul(X, cexpi(-gam * pw * leff));
    }
}

// ----------------------------------------------------------------------------------------------
// k_colx16: the fused column sweep for the 256 x (8+8) tile with both column transforms held in
// REGISTERS (16 points per thread, r16_* + lvl2_*256): per tile one LDS exchange per transform instead
// of four read+write passes, the Kerr step on registers (the other polarisation of a sample sits in
// lane t^8: one DPP row rotation).  The kernel sits at the VGPR cap (16 FP64 complex points per lane), so
// the NEXT tile is staged by LDS-DMA (global_load_lds_dwordx4: no register destination) straight into the
// exchange buffer as soon as the current tile has left it, and is in flight during the last register
// transform and the stores of the current tile.
// Thread = (j = tid>>4, t = tid&15): t < 8 -> column t of ux, t >= 8 -> column t-8 of uy.
// LDS image of a tile: s[row][16] (256 B per row: 8 columns of ux | 8 of uy); one LDS-DMA instruction of a wave
// fills 4 consecutive rows (64 lanes x 16 B = 1 KiB, lane-linear), wave w owns rows 64w .. 64w+63 -- exactly the
// rows its own threads read first, so the landing needs the wave's own vmcnt wait and no workgroup barrier.
//
// Landing WITHOUT a vmcnt wait.  A gfx9-family wave has one counter for its loads and its stores, and they retire out of
// order with respect to each other: waiting for the staged tile through vmcnt means waiting for the acknowledgement of
// every store of the previous tile as well (~1 us at the top of every tile, the most variable microsecond of the loop, and
// what varies shows up again as waiting at the frame barrier).  So the copies are issued from inline assembly (the
// compiler's wait-count pass does not see them and inserts no wait of its own in front of the LDS reads), the frame record
// goes LAST into a copy whose `done` word holds a sentinel, and the wave spins on that word in LDS: loads return in
// issue order, so the record's arrival implies the tile's.  The workgroup barriers of the tile loop are bare s_barrier +
// lgkmcnt waits for the same reason (__syncthreads carries a release fence = a vmcnt wait while stores are in flight).
#define PLX_REC_SENTINEL 0x7fffffff
#ifdef PLX_EMU
__device__ __forceinline__ void glds16(const cplx *src, cplx *lds_wave_base, int lane) { lds_wave_base[lane] = *src; }
// rows row .. row+63 of a tile: 16 copies of 4 rows each; src: this lane's first element, stride: elements between copies
__device__ __forceinline__ void glds_rows(const cplx *src, size_t stride, cplx *lds_wave_base, int lane)
{
    for (int i = 0; i < 16; i++) glds16(src + i * stride, lds_wave_base + 64 * i, lane);
}
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
__device__ __forceinline__ int lds_peek(const int *p) { return *(const volatile int *)p; }
__device__ __forceinline__ void lds_settle() {}
__device__ __forceinline__ void emu_lockstep() { __syncthreads(); }   // (the emulator's lanes are free-running threads: a wave's lanes meet here)
#else
__device__ __forceinline__ void glds16(const cplx *src, cplx *lds_wave_base, int)
{
    const unsigned lb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)lds_wave_base);   // (wave-uniform by construction)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lb) : "memory", "m0");
}
__device__ __forceinline__ void glds_rows(const cplx *src, size_t stride, cplx *lds_wave_base, int)
{
    unsigned lb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    unsigned long long p = (unsigned long long)src;
    const unsigned long long st = (unsigned long long)stride * sizeof(cplx);
#define PLX_GLDS_STEP "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tv_lshl_add_u64 %0, %0, 0, %2\n\ts_add_u32 %1, %1, 0x400\n\t"
    asm volatile(PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP
                 PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP PLX_GLDS_STEP
                 : "+v"(p), "+s"(lb) : "s"(st) : "memory", "m0", "scc");
#undef PLX_GLDS_STEP
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int lds_peek(const int *p)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) void *)p) : "memory");
    return v;
}
__device__ __forceinline__ void lds_settle() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void emu_lockstep() {}
#endif


// D = false: the same sweep for SCALAR plans (scalar_ssfm, fiber.m:557-636, without XPM -- its row sums across channels would
// need the other channels' tiles): a tile is sixteen columns of the one field, lane t its column t; the frame maximum is
// max |u|^2 (nextstep :694-698 with ~isy), the Kerr step nl_step's u .*= fastexp(-gam |u|^2 leff) (:792-804) on the lane's own
// sixteen points -- no lane pairs, no swapped halves.
template <bool D> __global__ __launch_bounds__(256, 2) void k_colx16(SsfmArgs a, int tiles_x, int tiles_pf)
{
    constexpr int CW = D ? 8 : 16;         // columns of one polarisation in a tile
    PLX_DYN_LDS(lds);
    if (all_done_or_aborted(a)) return;
    const int tid = threadIdx.x, t = tid & 15, j = tid >> 4;
    const int N2 = 1 << a.p2;
    const int LOGN = a.p1 + a.p2;
    cplx *s = (cplx *)lds;                 // [256][16] exchange buffer
    cplx *tw = s + 4096;                   // W_256^k, k < 128
    double *red = (double *)(tw + 128);
    FrameCtl *lctl = (FrameCtl *)(red + 32);
    lds_load_twiddles(tw, a.tw1, 128, tid, 256);
    cplx *const fld = (D && t >= 8) ? a.uy : a.ux;
    const int round = a.round;
    const int colt = D ? t & 7 : t;
    const bool isx = !D || t < 8;
    const int lane = tid & 63, row0 = (tid >> 6) * 64;     // this wave stages rows row0 .. row0+63
    // Teams.  The grid is a whole number of TEAMS of tiles_pf workgroups; a team takes a frame at a time, workgroup ti of the
    // team its tile ti (channel c, column block bx: the same every time round).  A team's first frame is slot `team` of the
    // active list; the further ones are claimed from a counter, one at a time, so a team that runs late -- its workgroups
    // found no room beside another kernel's waves, or its CUs are slow -- leaves its share to the others instead of holding
    // the launch up.  The team's first workgroup claims the frame of iteration k+2 while the team sits at the barrier of
    // iteration k (its second wave does: it has nothing else to do there) and posts it in the team's mailbox; everybody
    // picks the frame of iteration k+1 up with the polls of barrier k, where it has been lying for a whole iteration.
    // The mailbox is a log, one entry per iteration and no reuse within a launch: through finished frames of a stale list the
    // team's workgroups run without meeting, and its first may be any number of iterations ahead of its slowest.
    const int nact = a.nactive[0];
    const int NT = gridDim.x / tiles_pf, team = blockIdx.x / tiles_pf, ti = blockIdx.x - team * tiles_pf;
    const int c = ti / tiles_x, bx = ti - c * tiles_x;
    const unsigned long long rtag = (unsigned long long)(((unsigned)round + 1u) & 0xfffffu) << 22;
    unsigned long long *const mbox = a.mbox + (size_t)a.mbox_stride * team;
    if (blockIdx.x == 0 && tid == 0) a.grab[(round & 1) ^ 1] = 0;        // (the other parity's counter: for the next launch)
    auto post = [&](int k) {               // the team's first workgroup, thread 64: claim the frame of iteration k and post it
        const int sl = NT + atomicAdd(a.grab + (round & 1), 1);
        const int fr = sl < nact ? a.active[sl] : -1;
        if (k < a.mbox_stride) st_agent(mbox + k, ((rtag | (unsigned long long)(k + 1)) << 22) | (unsigned long long)(fr + 1));
    };
    auto posted = [&](int k, unsigned long long v) -> bool { return (v >> 22) == (rtag | (unsigned long long)(k + 1)); };
    // stage(f, par): start the asynchronous copy of this workgroup's tile of frame f into s, and of the frame's step-control
    // record into this wave's own copy (7 lanes x 16 B): everything the next iteration needs arrives without a load between
    // its loop top and its first transform.  The list may be a few steps old (small batches rebuild it once per chunk of
    // steps), so the record's `done` is still checked at the loop top; it cannot change before THIS workgroup has met the
    // frame's barrier.
    // (record copies: [iteration parity][wave] -- the copy of the tile in hand is still needed while the next one lands)
    int it = 0;
    auto stage = [&](int f, int par) {
        const int fc = f * a.nfc + c;
        // (lane & 15 == t: the lane stages a piece of the same column of the same polarisation it later works on)
        int lq = lane >> 4;
        pin(lq);                           // (addresses are formed where they are used: hoisted out of the tile loop they end up in scratch)
        const cplx *src = fld + ((size_t)fc << LOGN) + (size_t)bx * CW + colt + (size_t)(row0 + lq) * N2;
        FrameCtl *const rec = lctl + 4 * par + (tid >> 6);
        if (lane == (int)(offsetof(FrameCtl, done) / 16)) rec->done = PLX_REC_SENTINEL;   // (the lane whose piece of the record holds the word)
        lds_settle();                      // (the sentinel is in place before the copy that replaces it can land)
        glds_rows(src, (size_t)4 * N2, s + (size_t)row0 * 16, lane);
        static_assert(sizeof(FrameCtl) % 16 == 0, "the record travels as 16-byte pieces");
        int ln = lane;
        pin(ln);                           // (the address is formed here: kept across the tile loop it would sit in scratch)
        if (ln < (int)(sizeof(FrameCtl) / 16)) glds16((const cplx *)(a.ctl + f) + ln, (cplx *)rec, ln);
    };
    // [stamps:init]
    double *const gaml = (double *)((char *)(lctl + 8) + 128);    // gam[channel] (at most COLX_NFC channels: checked by the plan)
    for (int k = tid; k < a.nfc; k += 256) gaml[k] = a.gam[k];
    // The lanes of the second polarisation (t >= 8) take the odd twiddles of the two m = 256 stages from a NEGATED copy of the
    // table (lvl2_dit256s / lvl2_dif256s): between the inverse and the forward transform their registers hold the halves of
    // the tile's points swapped, y[k] = point j + 16 (k ^ 8), so that the register pair (k, k + 8) of the lane pair (t, t ^ 8)
    // is the two polarisations of ONE sample -- for k < 8 the X lane's sample k and the Y lane's sample k + 8 -- and the
    // frame maximum and the Kerr step need no per-lane selects (130 v_cndmask per tile before).  Exact: negations only.
    cplx *const twn = (cplx *)(gaml + COLX_NFC);
    if (D && tid < 128) { const cplx w = a.tw1[tid]; twn[tid] = make_double2(-w.x, -w.y); }
    // (the per-lane table pointer and the LDS distance between the halves of a column, 2048 elements for the lanes that swap
    //  them, are re-derived from t where they are used: held across the tile loop they cost the two registers that spill)
#define COLX_TWA(tp) ((D && (tp) >= 8) ? (const cplx *)twn : (const cplx *)tw)
#define COLX_HSW(tp) ((D && (tp) >= 8) ? 2048 : 0)
    int f = team < nact ? a.active[team] : -1;
    if (f < 0) return;                     // (more teams than frames)
    stage(f, 0);
    if (ti == 0 && tid == 64) post(1);     // (the one claim nobody's wait hides: once per launch)
    // The workgroup of a frame's first tile owns the frame's record: it finishes the controller (ctrl_tail) and writes the
    // record back LATER, while it waits at the barrier of its next tile (red[8]: the frame owed, or -1).
    CtrlK *const kk = (CtrlK *)(lctl + 8);
    if (tid == 0) {
        red[8] = -1.0;
        kk->dphimax = a.dphimax; kk->alphalin = a.alphalin; kk->dzmax = a.dzmax; kk->dz0 = a.dz0; kk->zdone0 = a.zdone0; kk->Lf = a.Lf; kk->lcorr = a.lcorr;
        kk->dual = a.dual ? 1 : 0; kk->resume = a.resume ? 1 : 0; kk->ncycle0 = a.ncycle0; kk->nfc = 0; kk->ndone = a.ndone; kk->umax = nullptr; kk->gam = nullptr;
        kk->dzlist = a.dzlist; kk->dzlog = a.dzlog; kk->ndz = a.ndz; kk->logcap = a.logcap;
    }
    auto settle = [&](int par) {           // tid 0 only; par: the parity the owed record was staged in
        const int pf = (int)red[8];
        if (pf < 0) return;
        FrameCtl *pr = lctl + 4 * par;
        if (!pr->done) ctrl_tail_call((const PLX_LDS_QUAL CtrlK *)kk, (PLX_LDS_QUAL FrameCtl *)pr);
        a.ctl[pf] = *pr;
        red[8] = -1.0;
    };
    __syncthreads();                       // twiddles staged
    // (red[10 + (it & 1)]: the team's next frame, by iteration parity -- the waves of a workgroup read it at their own pace at the
    //  end of an iteration, and the next iteration may write its successor with no workgroup barrier in between)
    for (;; it++) {
        FrameCtl *const wrec = lctl + 4 * (it & 1) + (tid >> 6);
        const int fc = f * a.nfc + c;
        // [phase 0] loop top
        if (a.safe_land) drain_vmem();     // (checking mode: the ordinary wait as well -- results must not depend on it)
        while (lds_peek(&wrec->done) == PLX_REC_SENTINEL) nap();   // this wave's rows of the tile and its copy of the record are in LDS
        emu_lockstep();
        // [phase 1] wait for the staged tile
        if (wrec->done) {                  // a listed frame that has finished meanwhile (the same answer in every wave)
            if (tid == 0) {
                settle((it & 1) ^ 1);
                unsigned long long v;
                unsigned spins = 0;
                bool dead = false;
                const long long t0 = plx_clock();
                while (!posted(it + 1, v = ld_agent(mbox + (it + 1)))) {
                    nap();
                    if ((++spins & 255u) == 0 && (ld_agent((const unsigned *)a.ndone + 1) != 0 || plx_clock() - t0 > a.spin_ticks)) {
                        st_agent((unsigned *)a.ndone + 1, 1u);      // (the same sticky abort as the frame barrier's)
                        dead = true;
                        break;
                    }
                }
                red[10 + (it & 1)] = dead ? -2.0 : (double)((int)(v & 0x3fffffull) - 1);
            } else if (ti == 0 && tid == 64) {
                post(it + 2);
            }
            lds_barrier();
            f = (int)red[10 + (it & 1)];
            if (f == -2) return;           // (timed out: uniform over the workgroup)
            if (f < 0) { it++; break; }
            stage(f, (it & 1) ^ 1);        // (s is free here: every path below ends past its last read of s, and so far
            continue;                      //  each wave has only touched its own rows)
        }
        const size_t cbase = ((size_t)fc << LOGN) + (size_t)bx * CW + colt;   // in the caller's arrays
        const bool started = wrec->started != 0;
        {
            cplx x[16];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = s[((16 * j + k) << 4) + t];
            if (started) {                 // rows 16j .. 16j+15 of the (bit-reversed) column spectrum
                r16_dit(x);
#pragma unroll
                for (int k = 0; k < 16; k++) s[((16 * j + k) << 4) + t] = x[k];
            }
        }
        lds_barrier();
        // [phase 2] r16_dit + exchange write + workgroup barrier
        cplx y[16];                        // point j + 16k (lanes t >= 8: j + 16 (k ^ 8) from here to the forward transform)
        {
            // (a frame that has not started yet is read with the halves swapped; a started one gets them swapped by the
            //  negated twiddles of the inverse transform's last stage)
            int tp = t;
            pin(tp);
            const int ysw = started ? 0 : COLX_HSW(tp);
#pragma unroll
            for (int k = 0; k < 8; k++) y[k] = s[((j + 16 * k) << 4) + t + ysw];
#pragma unroll
            for (int k = 8; k < 16; k++) y[k] = s[((j + 16 * k) << 4) + t - ysw];
        }
        double sc = 1.0;
        if (started) {                     // finish step s: ifft (1/N), attenuation (:531-532)
            lvl2_dit256s(y, j, tw, COLX_TWA(t));
            sc = wrec->att * a.invN;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) y[k] = cscale(y[k], sc);
        double m = 0;
        if (!D) {
#pragma unroll
            for (int k = 0; k < 16; k++) m = fmax(fma(y[k].y, y[k].y, y[k].x * y[k].x), m);
        } else
#pragma unroll
        for (int k = 0; k < 8; k++) {
            // |ux|^2 + |uy|^2 of ONE sample: this lane's y[k] and the partner lane's y[k + 8] (lane t ^ 8).  The X lane forms
            // it for the samples k < 8, the Y lane for k + 8: the pair covers the column, each sum once (the same two powers
            // added as before: a + b == b + a, the maximum is the same to the bit)
            const double po = fma(y[k].y, y[k].y, y[k].x * y[k].x);
            const double pq = fma(y[k + 8].y, y[k + 8].y, y[k + 8].x * y[k + 8].x);
            const double p = po + lane_xchg<8>(pq);
            m = fmax(p, m);
        }
        m = wave_max(m);
        if ((tid & 63) == 0) red[tid >> 6] = m;
        lds_barrier();
        // [phase 3] exchange read + lvl2_dit + scale + max
        // Frame barrier (dz of the next step needs the frame-wide maximum, fiber.m:694-698): an all-gather.  Every workgroup
        // stores its tile maximum into its own slot, then its first wave polls the slots of the whole frame and runs the
        // step controller itself on its copy of the record (the same inputs, the same instructions: the same step in every
        // workgroup, to the bit) -- one store-to-load trip across the chip instead of two (members -> leader -> members).
        // One launch = one round, so kernel boundaries order the rounds and the protocol needs no read-modify-write: the slots
        // of the two launch parities alternate, and a workgroup empties its slot of the OTHER parity (last read one launch
        // ago) for the next round.  The workgroup of the frame's first tile writes the record back (k_row reads it) and
        // counts the finished frame.
        if (tid < 64) {
            const unsigned par = (unsigned)round & 1u;
            unsigned long long *slots = a.slots + ((size_t)par * a.nframes + f) * tiles_pf;
            double mm = red[0];
            for (int w = 1; w < 4; w++) mm = red[w] > mm ? red[w] : mm;
            const unsigned long long mine = (unsigned long long)__double_as_longlong(mm);
            if (tid == 0) {
                st_agent(slots + ti, mine);
                st_agent(a.slots + ((size_t)(par ^ 1u) * a.nframes + f) * tiles_pf + ti, ~0ull);
                settle((it & 1) ^ 1);      // (the wait below hides it)
            }
            double pm;
            unsigned long long mv = 0;      // (lane 0: the team's mailbox entry of the next iteration, read with the polls)
            unsigned spins = 0;
            bool dead = false;
            const long long t0 = plx_clock();
            for (;;) {
                bool all = true;
                pm = -INFINITY;
                if (tid == 0) { mv = ld_agent(mbox + (it + 1)); all = posted(it + 1, mv); }
                int i0 = tid;
                pin(i0);
                for (int i = i0; i < tiles_pf; i += 64) {
                    const unsigned long long b = (i == ti) ? mine : ld_agent(slots + i);
                    if (b == ~0ull) all = false;
                    else { const double gp = gaml[i / tiles_x] * __longlong_as_double((long long)b); pm = gp > pm ? gp : pm; }
                }
    // [stamps:poll]
                if (__all(all)) break;
                nap();
                if ((++spins & 255u) == 0) {           // (wave-uniform: every lane evaluates the same test)
                    const int late = ld_agent((const unsigned *)a.ndone + 1) != 0 || plx_clock() - t0 > a.spin_ticks;
                    if (__any(late)) { dead = true; break; }    // the frame's partners are not co-resident: abort, store nothing
                }
            }
            pm = wave_max(pm);
            // [phase 8] (dev) slot store -> every slot of the frame seen
            if (tid == 0) {
                red[10 + (it & 1)] = (double)((int)(mv & 0x3fffffull) - 1);
                if (dead) {
                    st_agent((unsigned *)a.ndone + 1, 1u);
                    red[19] = 1.0;
                } else {
                    const double pv = ctrl_head_call((const PLX_LDS_QUAL CtrlK *)kk, (PLX_LDS_QUAL FrameCtl *)wrec, ti == 0, pm, f);
                    if (ti == 0) red[8] = (double)f;
                    red[16] = pv; red[17] = pv < 0 ? 1.0 : 0.0; red[18] = mm; red[19] = 0.0;
                }
            }
        } else if (ti == 0 && tid == 64) {
            post(it + 2);                  // (this wave only waits for the first one here)
        }
        lds_barrier();
        // [phase 4] frame barrier
        if (red[19] != 0.0) return;        // barrier timed out (uniform over the workgroup): no store, no control update
        const double leff = red[16];
        const bool finished = red[17] != 0.0;
        if (finished) {                    // the frame has reached the fibre end: write the field out
            const int nf = (int)red[10 + (it & 1)];    // (the team's next frame, or -1: none left)
            if (nf >= 0) stage(nf, (it & 1) ^ 1);  // (every thread is past its reads of s: the barrier above)
            {
                int tq = t, jq = j;        // (opaque here: the sixteen row offsets of this once-per-frame store are not loop invariants
                pin(tq);                   //  worth thirty-two registers of the tile loop)
                pin(jq);
                const int rsw = (D && tq >= 8) ? 128 : 0;  // (the second polarisation's lanes hold the halves swapped)
#pragma unroll
                for (int k = 0; k < 8; k++) fld[cbase + (size_t)(jq + 16 * k + rsw) * N2] = y[k];
#pragma unroll
                for (int k = 8; k < 16; k++) fld[cbase + (size_t)(jq + 16 * k - rsw) * N2] = y[k];
            }
        } else {
            if (!D) {
                if (a.spm) {               // nl_step (:792-804, SPM only) on the lane's own sixteen points
                    const double gam = gaml[c];
                    if (fabs(gam * leff) * red[18] < 0.0625) {
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const double pw = y[k].x * y[k].x + y[k].y * y[k].y;
                            double sn, cs;
                            sincos_taylor(-gam * pw * leff, &sn, &cs);
                            y[k] = cmul(y[k], make_double2(cs, sn));
                        }
                    } else {               // ('--s-': the exact single step, radians of phase) through the exchange buffer
#pragma unroll
                        for (int k = 0; k < 16; k++) s[((j + 16 * k) << 4) + t] = y[k];
                        lds_barrier();
                        kerr_full_range_scalar(j, t, gam, leff);
                        lds_barrier();
#pragma unroll
                        for (int k = 0; k < 16; k++) y[k] = s[((j + 16 * k) << 4) + t];
                        lds_barrier();
                    }
                }
            } else if (a.spm) {            // Kerr step of step s+1 (:832-852) on registers
                const double gamleff = gaml[c] * leff;
                // |gamleff*P| <= gamleff * (tile maximum): a few mrad under the step controller, so the
                // Taylor form applies to the whole tile; otherwise ('--s-' exact single step) the rare
                // full-range path goes through LDS, one thread per polarisation pair.
                int tq = t;
                pin(tq);                   // (formed here, per tile: hoisted out of the tile loop the two constants would cost four registers for good)
                const double sgn = tq < 8 ? 1.0 : -1.0, sgn2 = tq < 8 ? 2.0 : -2.0;
                if (fabs(gamleff) * red[18] < 0.0625) {
                    // (one loop per equation: a uniform branch inside the unrolled body would cut it into 16 basic blocks)
                    // A sample's two polarisations sit in the lane pair (t, t^8), and the Kerr rotation is the same arithmetic
                    // on both: the pair shares the work by SAMPLES instead of doing all of it twice -- the X lane takes sample
                    // k (its own ux, the partner's uy), the Y lane sample k+8 (its own uy, the partner's ux); each forms both
                    // outputs of its sample and hands the partner's back.  own / oth = this lane's and the other polarisation.
                    auto kerr16 = [&](auto cn) {
                        constexpr bool CNLSE = decltype(cn)::value;
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            // (y[k] is this lane's polarisation of ITS sample, y[k + 8] the partner's sample: see twn above)
                            const cplx own = y[k], snd = y[k + 8];
                            const cplx oth = make_double2(lane_xchg<8>(snd.x), lane_xchg<8>(snd.y));
                            const double P = fma(own.y, own.y, own.x * own.x) + fma(oth.y, oth.y, oth.x * oth.x);
                            double sn, cs;
                            sincos_taylor(-gamleff * P, &sn, &cs);
                            const cplx nl = make_double2(cs, sn);
                            cplx A = cmul(own, nl), B = cmul(oth, nl);
                            if (CNLSE) {
                                // s3 = 2 (Re ux Im uy - Im ux Re uy) (:841-851): on the Y lane own/oth are swapped, the two
                                // products swap and the difference changes sign exactly
                                const double s3 = sgn2 * __dsub_rn(__dmul_rn(A.x, B.y), __dmul_rn(A.y, B.x));
                                double sp, cp;
                                sincos_taylor(div3(gamleff * s3), &sp, &cp);
                                const double sg = sgn * sp, ng = -sg;   // ux' = cp ux + sp uy,  uy' = cp uy - sp ux
                                const cplx A2 = make_double2(cp * A.x + sg * B.x, cp * A.y + sg * B.y);
                                B = make_double2(cp * B.x + ng * A.x, cp * B.y + ng * A.y);
                                A = A2;
                            }
                            const cplx back = make_double2(lane_xchg<8>(B.x), lane_xchg<8>(B.y));
                            y[k] = A;
                            y[k + 8] = back;
                            sched_fence();         // (one sample pair at a time: a lone wave's FP64 rate does not depend on
                        }                          //  interleaving, and the pairs' operands need not all be selected up front)
                    };
                    if (a.manakov) kerr16(std::false_type{}); else kerr16(std::true_type{});
                } else {
                    int tp = t;            // (the tile goes through the exchange buffer in its natural layout)
                    pin(tp);
                    const int ksw = COLX_HSW(tp);
#pragma unroll
                    for (int k = 0; k < 8; k++) s[((j + 16 * k) << 4) + t + ksw] = y[k];
#pragma unroll
                    for (int k = 8; k < 16; k++) s[((j + 16 * k) << 4) + t - ksw] = y[k];
                    lds_barrier();
                    if (isx) kerr_full_range(j, t, gamleff, a.manakov);
                    lds_barrier();
#pragma unroll
                    for (int k = 0; k < 8; k++) y[k] = s[((j + 16 * k) << 4) + t + ksw];
#pragma unroll
                    for (int k = 8; k < 16; k++) y[k] = s[((j + 16 * k) << 4) + t - ksw];
                    lds_barrier();
                }
            }
            // [phase 5] Kerr step
            lvl2_dif256s(y, j, tw, COLX_TWA(t));
#pragma unroll
            for (int k = 0; k < 16; k++) s[((j + 16 * k) << 4) + t] = y[k];
            lds_barrier();
            cplx x[16];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = s[((16 * j + k) << 4) + t];
            lds_barrier();               // the exchange buffer is free: the next tile may land in it ...
            // [phase 6] lvl2_dif + exchange
            {
                const int nf = (int)red[10 + (it & 1)];
                if (nf >= 0) stage(nf, (it & 1) ^ 1);
            }
            r16_dif(x);                    // ... during the last register transform and the stores of this one
            // store_late (what ships: ON for multi-team launches such as C1, OFF where a frame is the whole grid): the tile's stores
            // are held back until this wave's rows of the NEXT tile are in LDS -- requests first, stores while the next tile
            // computes.  Multi-team: the landing no longer shares the workgroup's memory queue with 64 KiB of stores, k_colx16
            // beside the receiver 1170 -> 1138 us (+3...4 %).  One team (2^20-sample frames): the whole grid would wait for the
            // last landing before ANY store is issued, 359 -> 395 us per 16 frames, so the plan leaves it off there
            // (profiles/r03_store_late_ab.txt).
            if (a.store_late) {
                const int nf = (int)red[10 + (it & 1)];
                if (nf >= 0) {
                    FrameCtl *const nrec = lctl + 4 * ((it & 1) ^ 1) + (tid >> 6);
                    while (lds_peek(&nrec->done) == PLX_REC_SENTINEL) nap();
                }
            }
#pragma unroll
            for (int k = 0; k < 16; k++) fld[cbase + (size_t)(16 * j + k) * N2] = x[k];
            // [phase 7] staging issue + r16_dif + stores issued
        }
        f = (int)red[10 + (it & 1)];
    // [stamps:iter]
        if (f < 0) { it++; break; }
    }
    if (tid == 0) settle((it & 1) ^ 1);
    // [stamps:exit]
}


} // namespace

namespace plxs {
colx_kernel_t colx16_kernel(bool dual) { return dual ? (colx_kernel_t)k_colx16<true> : (colx_kernel_t)k_colx16<false>; }
} // namespace plxs

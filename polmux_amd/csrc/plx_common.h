// plx_common.h -- shared definitions for the gfx950 kernels of libpolmux_hip.
//
// The kernels are written for CDNA4 only (64-lane waves, 160 KiB LDS per CU).
// -DPLX_EMU swaps the HIP runtime header for tests/emu/hip_emu.h, a host-side
// emulator used exclusively by the sanitizer/unit tests; the shipped library is
// always the hipcc build.
#pragma once
#ifdef PLX_EMU
#include "hip_emu.h"
#define PLX_LAUNCH(kern, grid, block, shmem, stream, ...) PLX_EMU_LAUNCH(kern, grid, block, shmem, stream, __VA_ARGS__)
#define PLX_DYN_LDS(name) char *name = emu::dyn_lds()
#define PLX_LDS_QUAL
#else
#include <hip/hip_runtime.h>
#define PLX_LAUNCH(kern, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kern, grid, block, shmem, (hipStream_t)(stream), __VA_ARGS__)
// all LDS lives in the dynamic region, base 16-byte aligned (guide G17)
#define PLX_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define PLX_LDS_QUAL __attribute__((address_space(3)))   // pointer-to-LDS by type (for out-of-line device functions)
#endif

#include <cstddef>
#include <cstdint>
#include <string>

typedef double2 cplx; // interleaved complex128: x = re, y = im

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); } // a*conj(b)
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return make_double2(a.x * s, a.y * s); }
__device__ __forceinline__ cplx cconj(cplx a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ cplx cmuli(cplx a) { return make_double2(-a.y, a.x); }  // a * i
__device__ __forceinline__ cplx cmulni(cplx a) { return make_double2(a.y, -a.x); } // a * (-i)
// component-wise select (a ternary on two struct lvalues becomes a select of POINTERS, which pins
// register arrays into scratch memory)
__device__ __forceinline__ cplx csel(bool c, cplx a, cplx b) { return make_double2(c ? a.x : b.x, c ? a.y : b.y); }
__device__ __forceinline__ cplx cexpi(double a)
{
    double s, c;
    sincos(a, &s, &c);
    return make_double2(c, s);
}

// exp(-i*2*pi*x) for a phase x given in TURNS.  The linear step's phases betat*dz reach 1e4 rad; the
// general sincos spends ~100 instructions on range reduction.  With the phase tables stored in turns the
// reduction is exact (x - rint(x)), a quadrant split brings the angle to |psi| <= pi/4, and odd/even
// Taylor polynomials to x^17 / x^16 (truncation < 1e-16) finish in ~30 FMAs.  The product x = turns*dz
// carries the same 1e-16 relative rounding as the reference's betat*dz, i.e. ~1e-12 rad at 1e4 rad.
__device__ __forceinline__ cplx cexp_neg_turns(double x)
{
    const double r = x - rint(x);            // [-0.5, 0.5] turns, exact
    const double q = rint(4.0 * r);          // quadrant -2..2
    const double psi = (r - 0.25 * q) * 6.28318530717958647692; // [-pi/4, pi/4]
    const double z = psi * psi;
    double ps = fma(z, 1.0 / 355687428096000.0, -1.0 / 1307674368000.0);
    ps = fma(z, ps, 1.0 / 6227020800.0);
    ps = fma(z, ps, -1.0 / 39916800.0);
    ps = fma(z, ps, 1.0 / 362880.0);
    ps = fma(z, ps, -1.0 / 5040.0);
    ps = fma(z, ps, 1.0 / 120.0);
    ps = fma(z, ps, -1.0 / 6.0);
    const double sn = fma(psi * z, ps, psi);
    double pc = fma(z, 1.0 / 20922789888000.0, -1.0 / 87178291200.0);
    pc = fma(z, pc, 1.0 / 479001600.0);
    pc = fma(z, pc, -1.0 / 3628800.0);
    pc = fma(z, pc, 1.0 / 40320.0);
    pc = fma(z, pc, -1.0 / 720.0);
    pc = fma(z, pc, 1.0 / 24.0);
    pc = fma(z, pc, -0.5);
    const double cs = fma(z, pc, 1.0);
    // angle = psi + q*pi/2; result = cos(angle) - i sin(angle):
    //   q mod 4 = 0: (cs, -sn)   1: (-sn, -cs)   2: (-cs, sn)   3: (sn, cs)
    // as a swap for odd q and two sign flips on the high words -- no branch (the four-way selects compiled to exec-masked
    // blocks: ~20 instructions and two branches per call; the values are the same to the bit)
    const int qi = (int)q & 3;               // -2 -> 2, -1 -> 3
    const bool odd = (qi & 1) != 0;
    const double c0 = odd ? sn : cs, s0 = odd ? cs : sn;
    const unsigned long long cflip = (unsigned long long)((qi + 1) & 2) << 62;   // the real part is negated for q = 1, 2
    const unsigned long long sflip = (unsigned long long)((qi & 2) ^ 2) << 62;   // the imaginary part is -s0 for q = 0, 1, +s0 for q = 2, 3
    return make_double2(__longlong_as_double((long long)((unsigned long long)__double_as_longlong(c0) ^ cflip)),
                        __longlong_as_double((long long)((unsigned long long)__double_as_longlong(s0) ^ sflip)));
}

// The same value through a 64-entry table of the unit circle, T[k] = (cos, -sin)(2 pi k / 64) (LDS or L1), and a short
// polynomial on the remainder |d| <= pi/64: 20 FP64 operations and no quadrant logic where cexp_neg_turns needs 26 plus a
// four-way select -- for kernels that are bound by their instruction issue (k_row4k: profiles/r04_notes.md).  The
// remainder r - q/64 is exact (both on r's grid), the truncation terms d^11/11! and d^10/10! are below 1e-20.
#define PLX_CTAB 64
__device__ __forceinline__ cplx cexp_neg_turns_tab(double x, const cplx *T)
{
    const double r = x - rint(x);            // [-0.5, 0.5] turns, exact
    const double q = rint(64.0 * r);         // -32 .. 32
    const double d = fma(q, -0.015625, r) * 6.28318530717958647692;
    const cplx w = T[(int)q & (PLX_CTAB - 1)];
    const double z = d * d;
    double ps = fma(z, 1.0 / 362880.0, -1.0 / 5040.0);
    ps = fma(z, ps, 1.0 / 120.0);
    ps = fma(z, ps, -1.0 / 6.0);
    const double sn = fma(d * z, ps, d);
    double pc = fma(z, 1.0 / 40320.0, -1.0 / 720.0);
    pc = fma(z, pc, 1.0 / 24.0);
    pc = fma(z, pc, -0.5);
    const double cs = fma(z, pc, 1.0);
    // exp(-i (a + d)) with w = (cos a, -sin a): (cos a cos d - sin a sin d, -(sin a cos d + cos a sin d))
    return make_double2(fma(w.y, sn, w.x * cs), fma(-w.x, sn, w.y * cs));
}

// Keep a batch of global loads issued back-to-back: an empty asm that "uses" the loaded value
// stops the compiler from sinking each load next to its (conditional) consumer, where it would
// be followed by s_waitcnt vmcnt(0) and serialise the batch.
#ifdef PLX_EMU
__device__ __forceinline__ void pin(cplx &) {}
__device__ __forceinline__ void pin(double &) {}
__device__ __forceinline__ void pin(int &) {}
__device__ __forceinline__ void pin_uniform(int &) {}
__device__ __forceinline__ void sched_fence() {}
#else
__device__ __forceinline__ void pin(cplx &v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
__device__ __forceinline__ void pin(double &v) { asm volatile("" : "+v"(v)); }
// an index made opaque at this point: loads addressed through it cannot be hoisted above (keeps the live ranges of
// a later phase's operands out of an earlier, register-hungry phase)
__device__ __forceinline__ void pin(int &v) { asm volatile("" : "+v"(v)); }
// the same for a wave-uniform value (stays in a scalar register)
__device__ __forceinline__ void pin_uniform(int &v) { asm volatile("" : "+s"(v)); }
// the instruction scheduler moves nothing across this point (keeps the live ranges of unrolled iterations apart)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif

// ---- DPP lane exchange inside a 16-lane row (no LDS crossbar on the critical path) ----
// xor-1 / xor-2 are quad permutes; 7 and 15 are the row_half_mirror / row_mirror pairings
// (lane i <-> 7-i, i <-> 15-i), valid butterfly partners once the lower levels are reduced;
// 8 is row_ror:8 (lane i <-> i^8: the two halves of a row swap).
#ifdef PLX_EMU
template <int X> __device__ __forceinline__ double lane_xchg(double v) { return __shfl_xor(v, X, 64); }
#else
template <int X> __device__ __forceinline__ double lane_xchg(double v)
{
    constexpr int ctrl = X == 1 ? 0xB1 : X == 2 ? 0x4E : X == 7 ? 0x141 : X == 8 ? 0x128 : 0x140;
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every lane is a valid source under these patterns, so the "old" operand is never used: an untied old value (0,
    // bound_ctrl) lets the DPP move write a fresh register straight from the source instead of copying it first
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
#endif

// The two halves of a wave share a value each of them has computed: a = the copy of lanes 0-31, b = the copy of lanes 32-63,
// both in every lane (lane i and lane i + 32 are partners).  v_permlane32_swap (gfx950) exchanges the upper half of one
// register with the lower half of another: from two copies of m it leaves exactly a and b.
#ifdef PLX_EMU
__device__ __forceinline__ void half_share(double m, double &a, double &b)
{
    const int l = (int)(threadIdx.x & 31u);
    a = __shfl(m, l, 64);
    b = __shfl(m, l + 32, 64);
}
#else
__device__ __forceinline__ void half_share(double m, double &a, double &b)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const unsigned lo = (unsigned)__double2loint(m), hi = (unsigned)__double2hiint(m);
    const u2 r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const u2 r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}
#endif

// (complex forms: on the GPU two scalar swaps; the emulator moves the pair in one rendezvous of its 64 host threads)
#ifdef PLX_EMU
__device__ __forceinline__ void half_share(cplx m, cplx &a, cplx &b)
{
    const int l = (int)(threadIdx.x & 31u);
    const double v[4] = {m.x, m.y, m.x, m.y};
    const int src[4] = {l, l, l + 32, l + 32};
    double o[4];
    emu_shfl4(v, src, o, 4);
    a = make_double2(o[0], o[1]); b = make_double2(o[2], o[3]);
}
#else
__device__ __forceinline__ void half_share(cplx m, cplx &a, cplx &b)
{
    half_share(m.x, a.x, b.x);
    half_share(m.y, a.y, b.y);
}
#endif

// Lanes i and i + 32 trade halves of a register pair: on return v0 holds (own v0 | partner's v0 ... ) as v_permlane32_swap
// defines it -- lanes 0-31 keep v0 and receive the upper half's v0 in v1, lanes 32-63 keep v1 and receive the lower half's v1
// in v0.  Applied twice it is the identity.  (k_row256r with PMD: lane i holds ux, lane i + 32 uy of the same sixteen bins;
// after the trade every lane holds both polarisations of eight of them.)
#ifdef PLX_EMU
__device__ __forceinline__ void half_trade(double &v0, double &v1)
{
    const int l = (int)(threadIdx.x & 31u);
    const bool lower = (threadIdx.x & 32u) == 0;
    const double a = __shfl(v1, l, 64), b = __shfl(v0, l + 32, 64);
    const double n0 = lower ? v0 : a, n1 = lower ? b : v1;
    v0 = n0; v1 = n1;
}
#else
__device__ __forceinline__ void half_trade(double &v0, double &v1)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 rl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v0), (unsigned)__double2loint(v1), false, false);
    const u2 rh = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v0), (unsigned)__double2hiint(v1), false, false);
    v0 = __hiloint2double((int)rh[0], (int)rl[0]);
    v1 = __hiloint2double((int)rh[1], (int)rl[1]);
}
#endif

#ifdef PLX_EMU
__device__ __forceinline__ void half_trade(cplx &c0, cplx &c1)
{
    const int l = (int)(threadIdx.x & 31u);
    const bool lower = (threadIdx.x & 32u) == 0;
    const double v[4] = {c1.x, c1.y, c0.x, c0.y};
    const int src[4] = {l, l, l + 32, l + 32};
    double o[4];
    emu_shfl4(v, src, o, 4);
    const cplx n0 = lower ? c0 : make_double2(o[0], o[1]), n1 = lower ? make_double2(o[2], o[3]) : c1;
    c0 = n0; c1 = n1;
}
#else
__device__ __forceinline__ void half_trade(cplx &c0, cplx &c1)
{
    half_trade(c0.x, c1.x);
    half_trade(c0.y, c1.y);
}
#endif

// ---- agent-scope (whole-GPU) relaxed atomics for words shared between workgroups INSIDE a launch:
// global_load/store ... sc1, served by the memory side, never by a possibly stale per-CU L1 / per-XCD
// L2 line (CDNA guide, Guideline 16: "8-B agent atomics both sides").
#ifdef PLX_EMU
__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return emu_atomic_load_u32(p); }
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) { emu_atomic_store_u32(p, v); }
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) { return emu_atomic_load_u64(p); }
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) { emu_atomic_store_u64(p, v); }
__device__ __forceinline__ void drain_vmem() {}
__device__ __forceinline__ void nap() { std::this_thread::yield(); }
__device__ __forceinline__ long long plx_clock() { return (long long)(emu_now_ms() * 1e5); }   // 10 ns ticks
#else
__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void nap() { __builtin_amdgcn_s_sleep(4); }
// constant-rate wall clock (100 MHz on gfx950: 10 ns ticks), for time-based spin limits
__device__ __forceinline__ long long plx_clock() { return (long long)wall_clock64(); }
#endif
__device__ __forceinline__ double ld_agent_f64(const double *p) { return __longlong_as_double((long long)ld_agent((const unsigned long long *)p)); }
__device__ __forceinline__ void st_agent_f64(double *p, double v) { st_agent((unsigned long long *)p, (unsigned long long)__double_as_longlong(v)); }

// ---- error plumbing (host) ------------------------------------------------------
void plx_set_error(const std::string &msg);
#define PLX_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            plx_set_error(std::string(#call) + ": " + hipGetErrorString(e_));                  \
            return PLX_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)
#define PLX_FAIL(code, msg) do { plx_set_error(msg); return (code); } while (0)

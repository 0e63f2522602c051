// ssfm_kernels.h -- what the plan (ssfm_plan.hip) sees of the kernel translation units: launch geometry constants, one
// selector per kernel family (the kernel templates stay private to their file; a selector returns the instantiation a plan
// asks for, or nullptr), launchers for the small kernels, and the launch / LDS / occupancy helpers on kernel pointers.
#pragma once
#include "ssfm_args.h"
#include <tuple>
#include <utility>

// ---- launch geometry shared by the kernels and the plan ----
#define COMPACT_THREADS 256       // k_compact: one workgroup
#define COL_THREADS_MAX 1024      // k_col_fwd / k_col_inv
#define ROW_THREADS 128           // k_row's smallest workgroup
#define ROWR_THREADS 64           // k_row256r / k_rowsm: one-wave workgroups
#define ROWR_LDS ((4 * 272 + 128 + 48) * sizeof(cplx))        // 20224 B: eight one-wave workgroups per CU
#define ROWR_LDS_SC ((4 * 272 + 128 + 80) * sizeof(cplx))     // scalar plans: four rows' bk entries (seven workgroups per CU)
#define ROWG_THREADS 256          // k_rowreg
#define ROWG_NTW(M) ((M) <= 1024 ? (M) / 2 : (M) / 8 + 4)
#define ROWG_LDS(M) ((size_t)((ROWG_THREADS / ((M) / 16)) * ((M) + (M) / 16) + ROWG_NTW(M) + 7 * 16 + PLX_CTAB + 17 * (ROWG_THREADS / ((M) / 16))) * sizeof(cplx))
#define ROWG_LDS_SPLIT(M) (ROWG_LDS(M) - (size_t)((ROWG_THREADS / ((M) / 16)) * ((M) + (M) / 16)) * sizeof(double))
#define ROWSM_LDS ((size_t)64 * 17 * sizeof(double) + (7 * 16 + PLX_CTAB) * sizeof(cplx))
#define COLX_NFC 64               // channels whose gam the fused sweep keeps in LDS

namespace plxs {

typedef void (*sweep_kernel_t)(SsfmArgs);
typedef void (*colx_kernel_t)(SsfmArgs, int, int);

// ---- selectors (defined next to the kernels) ----
PLX_HIDDEN sweep_kernel_t col_fwd_kernel();                                       // ssfm_col.hip
PLX_HIDDEN sweep_kernel_t col_inv_kernel();
PLX_HIDDEN colx_kernel_t colx16_kernel(bool dual);                                // ssfm_colx.hip
PLX_HIDDEN sweep_kernel_t row_kernel();                                           // ssfm_row.hip
PLX_HIDDEN sweep_kernel_t row256_kernel(bool pmd, bool scalar, bool split);       // ssfm_row256.hip
PLX_HIDDEN sweep_kernel_t row4k_kernel(bool pair, bool split);                    // ssfm_row4k.hip
PLX_HIDDEN sweep_kernel_t rowreg_kernel(int logm, bool pair, bool scalar, bool split);   // ssfm_rowreg.hip (logm 9 ... 11)
PLX_HIDDEN sweep_kernel_t rowsm_kernel(int logm, bool scalar);                    // ssfm_rowsm.hip (logm 5 ... 7)

// ---- the small kernels (ssfm_small.hip) ----
PLX_HIDDEN void launch_umax(dim3 grid, hipStream_t st, const SsfmArgs &a);
PLX_HIDDEN void launch_ctrl(int nframes, hipStream_t st, const SsfmArgs &a);
PLX_HIDDEN void launch_compact(const FrameCtl *ctl, int nframes, int *active, int *nactive, int serves, hipStream_t st);
PLX_HIDDEN void launch_rowsum(dim3 grid, hipStream_t st, const SsfmArgs &a);
PLX_HIDDEN void launch_pmd_tab(unsigned frames, hipStream_t st, const SsfmArgs &a);
PLX_HIDDEN void launch_nl_att(unsigned grid, hipStream_t st, cplx *u, const double *gam, size_t N, int nfc, int spm, int xpm, double leff, double att);
PLX_HIDDEN void launch_maxdiff(unsigned grid, hipStream_t st, const cplx *u, const cplx *uh, size_t n, unsigned long long *out);
PLX_HIDDEN void launch_richardson(unsigned grid, hipStream_t st, cplx *u, const cplx *uh, size_t n);

// ---- helpers on kernel pointers ----
#ifndef PLX_EMU
namespace detail {
template <class... P, size_t... I> inline void launch_ptr(void (*k)(P...), dim3 g, dim3 b, size_t lds, hipStream_t st, std::tuple<P...> &v, std::index_sequence<I...>)
{
    void *ptrs[] = {(void *)&std::get<I>(v)...};
    (void)hipLaunchKernel((const void *)k, g, b, ptrs, lds, st);
}
} // namespace detail
template <class... P, class... A> inline void launch(void (*k)(P...), dim3 g, dim3 b, size_t lds, hipStream_t st, A &&...args)
{
    std::tuple<P...> v{P(std::forward<A>(args))...};
    detail::launch_ptr(k, g, b, lds, st, v, std::index_sequence_for<P...>{});
}
template <class K> inline hipError_t allow_lds(K kern, size_t bytes)
{
    return hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
// workgroups of `kern` the runtime admits per CU at this block size and dynamic LDS (0 on failure)
template <class K> inline int blocks_per_cu(K kern, int threads, size_t lds)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, threads, lds) != hipSuccess) return 0;
    return nb;
}
#else
template <class... P, class... A> inline void launch(void (*k)(P...), dim3 g, dim3 b, size_t lds, hipStream_t, A &&...args)
{
    std::tuple<P...> v{P(std::forward<A>(args))...};
    emu::launch(g, b, lds, [=]() { std::apply(k, v); });
}
template <class K> inline hipError_t allow_lds(K, size_t) { return hipSuccess; }
template <class K> inline int blocks_per_cu(K, int, size_t) { return 2; }
#endif

} // namespace plxs

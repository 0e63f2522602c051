#!/usr/bin/env python3
"""dev tool: make the STAMPS copies of ssfm_colx.hip / ssfm_plan.hip (thread-0 wall-clock stamps of k_colx16's phases).

The product source carries only marker comments (`// [phase N] ...`, `// [stamps:init|poll|iter|exit|lds]`); this script
writes instrumented copies to build_stamps/ (git-ignored) in which the markers are replaced by the
stamping code below.  scripts/experiments/stamps.sh builds libpolmux_hip_stamps.so from that copy.
(A --nowait variant, nobody waiting at the frame barrier, existed in round 2; with the controller run by every workgroup on
its own maximum it no longer terminates -- the step sequence diverges -- and was removed.)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "polmux_amd", "csrc", "ssfm_colx.hip")
DST = os.path.join(ROOT, "build_stamps", "ssfm_colx.hip")
SRC_PLAN = os.path.join(ROOT, "polmux_amd", "csrc", "ssfm_plan.hip")
DST_PLAN = os.path.join(ROOT, "build_stamps", "ssfm_plan.hip")

DEFS = "__device__ long long g_stamps[32];\n__device__ long long g_wgwait[1024];\n__device__ int g_wgxcc[1024];\n__device__ long long g_wgend[1024], g_t0;    // per workgroup: wall clock at its exit of the last launch; earliest start   // per workgroup: time between its slot store and the frame's last arrival\n#define PLX_STAMP(i) do { if (tid == 0) { long long now_ = wall_clock64(); long long *st_ = (long long *)(red + 20); const long long d_ = now_ - st_[0]; st_[1 + (i)] += d_; ((long long *)((char *)(lctl + 8) + 128 + COLX_NFC * sizeof(double)))[i] += d_ * d_; st_[0] = now_; } } while (0)"
BLOCKS = {'init': '    if (tid == 0) { long long *st_ = (long long *)(red + 20); for (int i = 1; i < 12; i++) st_[i] = 0;   /* red[20..31] */ for (int i = 0; i < 16; i++) ((long long *)((char *)(lctl + 8) + 128 + COLX_NFC * sizeof(double)))[i] = 0; st_[0] = wall_clock64(); }', 'poll': '                if (tid == 0) ((long long *)(red + 20))[11] += 1;       // (dev) polls', 'iter': '        if (tid == 0) ((long long *)(red + 20))[10] += 1;', 'exit': '    if (tid == 0 && blockIdx.x < 1024) { g_wgwait[blockIdx.x] += ((const long long *)(red + 20))[1 + 8]; g_wgend[blockIdx.x] = wall_clock64(); g_wgxcc[blockIdx.x] = (int)__builtin_amdgcn_s_getreg(6164) & 15; /* HW_REG_XCC_ID[3:0] */ }\n    if (tid == 0) { const long long *st_ = (const long long *)(red + 20); for (int i = 0; i < 11; i++) atomicAdd((unsigned long long *)&g_stamps[i], (unsigned long long)st_[1 + i]); for (int i = 0; i < 9; i++) atomicAdd((unsigned long long *)&g_stamps[16 + i], (unsigned long long)((const long long *)(lctl + 8))[i]); }', 'lds': '    P->lds_col += 128;'}
HOST = 'extern "C" void plx_ssfm_stamps(long long *out, int reset)\n{\n    hipDeviceSynchronize();\n    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(long long) * 32);\n    hipMemcpyFromSymbol(out + 32, HIP_SYMBOL(g_wgwait), sizeof(long long) * 1024);\n    hipMemcpyFromSymbol(out + 32 + 1024, HIP_SYMBOL(g_wgend), sizeof(long long) * 1024);\n    { static int xc[1024]; hipMemcpyFromSymbol(xc, HIP_SYMBOL(g_wgxcc), sizeof(xc)); for (int i = 0; i < 1024; i++) out[32 + 2048 + i] = xc[i]; }\n    if (reset) { static long long zz[1024]; hipMemcpyToSymbol(HIP_SYMBOL(g_wgwait), zz, sizeof(zz)); }\n    if (reset) { long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }\n}'


def main():
    s = open(SRC).read()
    anchor = "template <bool D> __global__ __launch_bounds__(256, 2) void k_colx16("
    assert anchor in s
    s = s.replace(anchor, DEFS + "\n" + anchor, 1)
    rot = "    const int c = ti / tiles_x, bx = ti - c * tiles_x;"
    assert rot in s
    s = s.replace(rot, "#ifndef PLX_TILE_ROT\n#define PLX_TILE_ROT 0\n#endif\n    const int tiq = (ti + PLX_TILE_ROT) % tiles_pf;\n    const int c = tiq / tiles_x, bx = tiq - c * tiles_x;", 1)
    s, n = re.subn(r"^(\s*)// \[phase (\d+)\]", lambda m: "%sPLX_STAMP(%s); //" % (m.group(1), m.group(2)), s, flags=re.M)
    assert n >= 9, n
    plan = open(SRC_PLAN).read()
    for name, body in BLOCKS.items():
        mark = "    // [stamps:%s]\n" % name
        if name == "lds":
            assert mark in plan, name
            plan = plan.replace(mark, body + "\n")
            continue
        assert mark in s, name
        s = s.replace(mark, body + "\n")
    s += "\n" + HOST + "\n"
    os.makedirs(os.path.dirname(DST), exist_ok=True)
    open(DST, "w").write(s)
    open(DST_PLAN, "w").write(plan)
    print(DST, DST_PLAN)


if __name__ == "__main__":
    main()

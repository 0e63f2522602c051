"""Build/load the emulated (CPU, test-only) copy of the kernels -- see tests/emu/hip_emu.h.
polmux_amd never loads this library; it exists so the kernel sources can be exercised
(and sanitised) without a GPU."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "emu", "_build", "libpolmux_emu.so")
# PLX_EMU_SAN=1 (with LD_PRELOAD=$(gcc -print-file-name=libasan.so)): the ASan + UBSan build, tests/emu/build.sh SAN=1
if os.environ.get("PLX_EMU_SAN") == "1":
    LIB = os.path.join(ROOT, "tests", "emu", "_build", "libpolmux_emu_san.so")


def build():
    import fcntl
    srcs = glob.glob(os.path.join(ROOT, "polmux_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "tests", "emu", "hip_emu.*")) \
        + [os.path.join(ROOT, "include", "polmux_hip.h")]

    def fresh():
        return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs)
    if fresh():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with open(LIB + ".lock", "w") as lk:          # (pytest-xdist workers: one builds, the others wait and find it fresh)
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not fresh():
            env = dict(os.environ, SAN="1") if LIB.endswith("_san.so") else None
            subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build.sh")], stdout=subprocess.DEVNULL, env=env)
    return LIB


def binding():
    from polmux_amd import _abi
    return _abi.Binding(build())

"""dev tool: ns per symbol of the CMA driver kernel (plx_poldemux_dev, 7 taps, mu 1/6000, noisy QPSK) for 1 ... 1024 frames"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polmux_amd import _abi
if len(sys.argv) > 1 and sys.argv[1] != "base":
    _abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libpolmux_hip_%s.so" % sys.argv[1])
lib = _abi.get()
r = np.random.default_rng(1)
for L in (1024, 16384):
    for F in (1, 16, 1024):
        if L > 1024 and F > 16:
            continue
        a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (F, 2, L)))) + 0.3 * (r.standard_normal((F, 2, L)) + 1j * r.standard_normal((F, 2, L)))
        x = torch.from_numpy(a).cuda(); y = torch.empty_like(x)
        M = torch.from_numpy(np.tile(np.array([1, 0, 0, 1], complex), (F, 1))).cuda()
        R = np.array([1.0, 1.0]); passes = torch.zeros(F, dtype=torch.int32, device="cuda")
        ts = []
        for k in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lib.call("plx_poldemux_dev", 1, x.data_ptr(), y.data_ptr(), L, F, 7, 1 / 6000, R.ctypes.data, M.data_ptr(), None, passes.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        pm = int(passes.max())
        print("%-8s L=%5d F=%4d: %7.2f ms, %3d passes, %6.1f ns per symbol-pass  crc %08x" % (sys.argv[1] if len(sys.argv) > 1 else "base", L, F, min(ts) * 1e3, pm, min(ts) * 1e9 / (pm * L), __import__("zlib").crc32(y.cpu().numpy().tobytes())))

"""dev tool: k_cma16 alone (plx_poldemux_dev, 7 taps, mu 1/6000, 1024 symbols, noisy QPSK so that it runs all 299 passes)
for several batch sizes, and once more right behind a heavy kernel (clock state)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polmux_amd import _abi
lib = _abi.get()
L = 1024
r = np.random.default_rng(1)
def run(F, noise):
    a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (F, 2, L)))) + noise * (r.standard_normal((F, 2, L)) + 1j * r.standard_normal((F, 2, L)))
    x = torch.from_numpy(a).cuda(); y = torch.empty_like(x)
    M = torch.from_numpy(np.tile(np.array([1, 0, 0, 1], complex), (F, 1))).cuda()
    R = np.array([1.0, 1.0]); passes = torch.zeros(F, dtype=torch.int32, device="cuda")
    ts = []
    for k in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lib.call("plx_poldemux_dev", 1, x.data_ptr(), y.data_ptr(), L, F, 7, 1 / 6000, R.ctypes.data, M.data_ptr(), None, passes.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), int(passes.max()), int(passes.min())
for noise in (0.05, 0.3):
    for F in (1, 4, 128, 1024, 4096):
        print("noise %.2f F=%d: %.1f ms, passes %d..%d" % ((noise, F) + tuple(run(F, noise)[i] for i in (0, 2, 1))))

# the same call while another stream keeps the GPU busy (memory-bound copies / FP64 GEMMs)
big = torch.empty(1 << 28, dtype=torch.float64, device="cuda"); big2 = torch.empty_like(big)
A = torch.randn(4096, 4096, dtype=torch.float64, device="cuda")
for name, load in (("copy stream", lambda: big2.copy_(big)), ("fp64 gemm", lambda: A @ A)):
    side = torch.cuda.Stream()
    F, noise = 1024, 0.3
    a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (F, 2, L)))) + noise * (r.standard_normal((F, 2, L)) + 1j * r.standard_normal((F, 2, L)))
    x = torch.from_numpy(a).cuda(); y = torch.empty_like(x)
    M = torch.from_numpy(np.tile(np.array([1, 0, 0, 1], complex), (F, 1))).cuda()
    R = np.array([1.0, 1.0]); passes = torch.zeros(F, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for k in range(60): load()
    time.sleep(0.005)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.call("plx_poldemux_dev", 1, x.data_ptr(), y.data_ptr(), L, F, 7, 1 / 6000, R.ctypes.data, M.data_ptr(), None, passes.data_ptr(), torch.cuda.current_stream().cuda_stream)
    e1.record(); e1.synchronize()
    busy = not side.query()
    torch.cuda.synchronize()
    print("F=1024 noise 0.3 beside a %s (still running at the end: %s): %.1f ms, passes %d..%d" % (name, busy, e0.elapsed_time(e1), int(passes.min()), int(passes.max())))

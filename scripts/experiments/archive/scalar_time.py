# dev: time the scalar (single-polarisation) SSFM plan on a batch of frames through the resident tier (plx_ssfm_propagate_dev)
import ctypes as C, os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polmux_amd import _abi, synth
from polmux_amd.fiber import parse_flag, fiber_tables
from polmux_amd.gstate import GSTATE
import polmux_amd as px
lib = _abi.get()
nsymb, nt, F, flag = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "g-s-"
n = nsymb * nt
px.reset_all(nsymb, nt, 1); GSTATE.SYMBOLRATE = 28.0; GSTATE.LAMBDA = np.array([1550.0])
x = dict(length=8e4, alphadB=0.2, aeff=80.0, n2=2.7e-20, disp=17.0, slope=0.0, dphimax=5e-3, dzmax=2e4); x["lambda"] = 1550.0
fls, dph, dzm = parse_flag(flag, 1, x)
t = fiber_tables(x, fls, 1, 0.0)
ux, uy, bits, pw = synth.pdm_qpsk_field(nsymb, nt, 2.0)
d = _abi.SsfmDesc(); d.nfft, d.nfc, d.dual_pol, d.max_frames = n, 1, 0, F
for i in range(4): d.fls[i] = fls[i]
d.dzmaxt, d.dphimaxt, d.alphalin, d.length, d.nplates, d.manakov = dzm, dph, t["alphalin"], 8e4, 1, 0
gam = np.ascontiguousarray(t["gam"]); d.gam, d.betat, d.db1 = gam.ctypes.data, t["betat"].ctypes.data, 0
plan = C.c_void_p(); lib.call("plx_ssfm_create", C.byref(plan), C.byref(d))
lib.call("plx_ssfm_profile", plan, 1)
tx = torch.from_numpy(ux * math.sqrt(2.0)).cuda()
st = torch.cuda.current_stream().cuda_stream
ts = []
for r in range(4):
    u = tx.unsqueeze(0).repeat(F, 1).contiguous()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lib.call("plx_ssfm_propagate_dev", plan, u.data_ptr(), None, F, st)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
rows, steps = C.c_int64(), C.c_int64(); lib.call("plx_ssfm_stats", plan, C.byref(rows), C.byref(steps))
ms = (C.c_double * 4)(); k = (C.c_int64 * 4)(); lib.call("plx_ssfm_kernel_times", plan, ms, k)
info = (C.c_int32 * 8)(); lib.call("plx_ssfm_info", plan, info)
T = min(ts[1:]); ss = steps.value
print("scalar N=2^%d x %d frames '%s': %.2f ms, %.1f steps/frame, info %s" % (int(math.log2(n)), F, flag, T * 1e3, ss / n / F, list(info)))
per = [(ms[i] / max(k[i], 1) * 1e3) for i in range(4)]
act = F * n * 32.0   # bytes per sweep (16 read + 16 written per sample)
fr = lambda us: act / us / 8e6 if us > 0 else 0.0
if info[0]:
    print("  fused: k_colx16<false> %.1f us (%.3f of 8 TB/s)  row %.1f (%.3f)  control %.1f; step group (64 B per sample-step) %.3f" % (
          per[0], fr(per[0]), per[1], fr(per[1]), per[3], 64.0 * ss / T / 8e12))
else:
    print("  kernels us: col_fwd %.1f (%.3f of 8 TB/s)  row %.1f (%.3f)  col_inv %.1f (%.3f)  control %.1f; step group (96 B per sample-step) %.3f" % (
          per[0], fr(per[0]), per[1], fr(per[1]), per[2], fr(per[2]), per[3], 96.0 * ss / T / 8e12))

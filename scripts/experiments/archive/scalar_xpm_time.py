# dev: scalar XPM comb ('g-sx', nfc channels) through the resident tier: fused (default) against PLX_SSFM_NO_FUSE=1
import ctypes as C, os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polmux_amd import _abi, synth
from polmux_amd.fiber import parse_flag, fiber_tables
from polmux_amd.gstate import GSTATE
import polmux_amd as px
lib = _abi.get()
nsymb, nt, nfc, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = nsymb * nt
px.reset_all(nsymb, nt, nfc); GSTATE.SYMBOLRATE = 28.0; GSTATE.NCH = nfc; GSTATE.LAMBDA = 1550.0 + 0.4 * (np.arange(nfc) - (nfc - 1) / 2)
x = dict(length=8e4, alphadB=0.2, aeff=80.0, n2=2.7e-20, disp=17.0, slope=0.0, dphimax=5e-3, dzmax=2e4); x["lambda"] = 1550.0
fls, dph, dzm = parse_flag("g-sx", nfc, x)
t = fiber_tables(x, fls, nfc, 0.0)
cols = np.stack([synth.pdm_qpsk_field(nsymb, nt, 1.0 * (1 + 0.2 * k), 2 + 2 * k, 3 + 2 * k)[0] for k in range(nfc)])
d = _abi.SsfmDesc(); d.nfft, d.nfc, d.dual_pol, d.max_frames = n, nfc, 0, F
for i in range(4): d.fls[i] = fls[i]
d.dzmaxt, d.dphimaxt, d.alphalin, d.length, d.nplates, d.manakov = dzm, dph, t["alphalin"], 8e4, 1, 0
gam = np.ascontiguousarray(t["gam"]); d.gam, d.betat, d.db1 = gam.ctypes.data, t["betat"].ctypes.data, 0
plan = C.c_void_p(); lib.call("plx_ssfm_create", C.byref(plan), C.byref(d))
tx = torch.from_numpy(cols).cuda()
st = torch.cuda.current_stream().cuda_stream
ts = []
for r in range(4):
    u = tx.unsqueeze(0).repeat(F, 1, 1).contiguous()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lib.call("plx_ssfm_propagate_dev", plan, u.data_ptr(), None, F, st)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
rows, steps = C.c_int64(), C.c_int64(); lib.call("plx_ssfm_stats", plan, C.byref(rows), C.byref(steps))
info = (C.c_int32 * 8)(); lib.call("plx_ssfm_info", plan, info)
import zlib
print("scalar XPM comb %d ch x 2^%d x %d frames: %.2f ms, %.1f steps/frame, info %s, crc %08x" % (nfc, int(math.log2(n)), F, min(ts[1:]) * 1e3, steps.value / n / F / nfc, list(info), zlib.crc32(u.cpu().numpy().tobytes())))

# dev tool: batches of frames with very different step counts through plx_ssfm_propagate_dev (fused frame barrier,
# finished frames dropping out, chunked enqueue) vs the oracle on sampled frames.
import ctypes as C, os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import plxo as oracle
from polmux_amd import _abi, synth
from polmux_amd._abi import SsfmDesc
from polmux_amd.fiber import parse_flag, fiber_tables
from polmux_amd.gstate import GSTATE
import polmux_amd as px
lib = _abi.get()
r = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bad = ran = 0
worst = 0.0
for case in range(ncase):
    nsymb, nt = [(1024, 64), (1024, 64), (256, 64), (256, 16), (1024, 16)][int(r.integers(0, 5))]
    n = nsymb * nt
    F = int(r.integers(1, 70))
    flag = str(r.choice(["g-s-", "gps-", "g-s-", "gps-", "--s-", "g---"]))
    nplates = 10 if flag[1] == "p" else 1
    L = float(r.choice([2e4, 8e4]))
    px.reset_all(nsymb, nt, 1); GSTATE.SYMBOLRATE = 28.0; px.lasersource(1.0, 1550.0)
    x = dict(length=L, alphadB=0.2, aeff=80.0, n2=2.7e-20, disp=17.0, slope=0.0, dphimax=5e-3, dzmax=2e4, dgd=0.2, manakov="no"); x["lambda"] = 1550.0
    fls, dph, dzm = parse_flag(flag, 1, x)
    t = fiber_tables(x, fls, 1, math.sqrt(3 * math.pi / 8) * 0.2 / math.sqrt(nplates) if fls[1] else 0.0)
    ux0, uy0, _, _ = synth.pdm_qpsk_field(nsymb, nt, 1.0)
    scale = np.sqrt(10 ** r.uniform(-2.0, 1.0, F))              # 0.01 .. 10 mW: 1 .. ~150 steps
    d = SsfmDesc(); d.nfft, d.nfc, d.dual_pol, d.max_frames = n, 1, 1, F
    for i in range(4): d.fls[i] = fls[i]
    d.dzmaxt, d.dphimaxt, d.alphalin, d.length, d.nplates, d.manakov = dzm, dph, t["alphalin"], L, nplates, 0
    gam = np.ascontiguousarray(t["gam"]); d.gam, d.betat, d.db1 = gam.ctypes.data, t["betat"].ctypes.data, t["db1"].ctypes.data
    plan = C.c_void_p(); lib.call("plx_ssfm_create", C.byref(plan), C.byref(d))
    if fls[1]:
        db0 = r.random((F, nplates)) * 2 * np.pi - np.pi; th = r.random((F, nplates)) * np.pi - np.pi / 2; ep = 0.5 * np.arcsin(r.random((F, nplates)) * 2 - 1)
        lib.call("plx_ssfm_set_birefringence", plan, db0.ctypes.data, th.ctypes.data, ep.ctypes.data, F)
    ux = torch.from_numpy(np.stack([ux0 * s for s in scale])).cuda(); uy = torch.from_numpy(np.stack([uy0 * s for s in scale])).cuda()
    lib.call("plx_ssfm_propagate_dev", plan, ux.data_ptr(), uy.data_ptr(), F, torch.cuda.current_stream().cuda_stream)
    ncyc = np.zeros(F, np.int32); fdz = np.zeros(F)
    lib.call("plx_ssfm_results", plan, F, fdz.ctypes.data, ncyc.ctypes.data)
    lib.call("plx_ssfm_destroy", plan)
    gx, gy = ux.cpu().numpy(), uy.cpu().numpy()
    for f in sorted(set([0, F - 1] + [int(v) for v in r.integers(0, F, 4)])):
        b = (db0[f], th[f], ep[f]) if fls[1] else ([0.0], [0.0], [0.0])
        rc, ofd, onc, ox, oy = oracle.matrix_ssfm(ux0 * scale[f], uy0 * scale[f], t["betat"], t["db1"], dzm, dph, gam, t["alphalin"], L, nplates, False, fls, *b)
        e = max(np.abs(gx[f] - ox[:, 0]).max() / np.abs(ox).max(), np.abs(gy[f] - oy[:, 0]).max() / np.abs(oy).max())
        worst = max(worst, e); ran += 1
        if not (ncyc[f] == onc and e < 1e-9 and abs(fdz[f] - ofd) <= 1e-12 * ofd):
            bad += 1; print("BATCH MISMATCH n=%d F=%d flag=%s frame %d: nc %d vs %d, err %.3g" % (n, F, flag, f, ncyc[f], onc, e))
    print("case n=%d F=%d %s L=%g: ncycle %d..%d" % (n, F, flag, L, ncyc.min(), ncyc.max()))
print("%d frames checked, worst error %.3g, mismatches %d" % (ran, worst, bad))

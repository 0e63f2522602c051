# dev tool: random fibre configurations through the gateways vs the oracle (sizes, flags, channels, waveplates, lengths).
import ctypes as C, os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from oracle import plxo as oracle
from polmux_amd import _abi, synth
from polmux_amd._abi import SsfmDesc
from polmux_amd.fiber import parse_flag, fiber_tables
from polmux_amd.gstate import GSTATE
import polmux_amd as px
lib = _abi.get()
vp = lambda a: C.c_void_p(a.ctypes.data)
r = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
LG0, LG1 = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (8, 13)     # log2 of the frame sizes drawn (large: 16 19)
worst = 0.0
bad = 0
ran = 0
illc = 0
for case in range(ncase):
    lg = int(r.integers(LG0, LG1 + 1))
    ntl = int(r.choice([3, 4, 5] if lg < 16 else [5, 6, 7]))
    nsl = lg - ntl
    if nsl % 2: nsl -= 1; lg = nsl + ntl
    if nsl < 6: continue
    nsymb, nt = 1 << nsl, 1 << ntl
    n = nsymb * nt
    dual = bool(r.integers(0, 2))
    nfc = int(r.choice([1, 1, 2, 3]))
    flag = "".join([r.choice(["g", "-"]), (r.choice(["p", "-"]) if dual else "-"), r.choice(["s", "-"]), (r.choice(["x", "-"]) if (not dual and nfc > 1) else "-")])
    if flag == "----": flag = "g---"
    nplates = int(r.choice([1, 3, 10, 37])) if flag[1] == "p" else 1
    manakov = bool(r.integers(0, 2)) and flag[1] == "p"
    L = float(r.choice([5e3, 2e4, 8e4] if LG1 < 16 else [2e3, 5e3, 1e4]))
    pavg = float(r.choice([0.5, 2.0, 8.0]))
    px.reset_all(nsymb, nt, nfc); GSTATE.SYMBOLRATE = 28.0
    GSTATE.NCH = nfc; GSTATE.LAMBDA = 1550.0 + 0.4 * (np.arange(nfc) - (nfc - 1) / 2) if nfc > 1 else np.array([1550.0])
    x = dict(length=L, alphadB=float(r.choice([0.0, 0.2])), aeff=80.0, n2=2.7e-20, disp=float(r.choice([17.0, 4.0, -2.0])), slope=float(r.choice([0.0, 0.057])),
             dphimax=float(r.choice([5e-3, 2e-2])), dzmax=2e4, dgd=0.3, manakov="yes" if manakov else "no"); x["lambda"] = 1550.0
    try:
        fls, dph, dzm = parse_flag(flag, nfc, x)
    except ValueError:
        continue
    dgdrms = math.sqrt(3 * math.pi / 8) * 0.3 / math.sqrt(nplates) if fls[1] else 0.0
    t = fiber_tables(x, fls, nfc, dgdrms)
    cols = [synth.pdm_qpsk_field(nsymb, nt, pavg * (1 + 0.3 * k), 2 + 2 * k, 3 + 2 * k) for k in range(nfc)]
    sx = np.stack([c[0] for c in cols], 1); sy = np.stack([c[1] for c in cols], 1)
    if fls[1]:
        db0 = r.random(nplates) * 2 * np.pi - np.pi; th = r.random(nplates) * np.pi - np.pi / 2; ep = 0.5 * np.arcsin(r.random(nplates) * 2 - 1)
    else:
        db0 = th = ep = np.zeros(1)
    d = SsfmDesc(); d.nfft, d.nfc, d.dual_pol, d.max_frames = n, nfc, int(dual), 1
    for i in range(4): d.fls[i] = fls[i]
    d.dzmaxt, d.dphimaxt, d.alphalin, d.length, d.nplates, d.manakov = dzm, dph, t["alphalin"], L, nplates, int(manakov)
    gam = np.ascontiguousarray(t["gam"]); d.gam, d.betat, d.db1 = gam.ctypes.data, t["betat"].ctypes.data, t["db1"].ctypes.data
    fd, nc = C.c_double(), C.c_int32()
    tag = "n=2^%d nt=%d dual=%d nfc=%d flag=%s plates=%d manakov=%d L=%g P=%g" % (lg, nt, dual, nfc, flag, nplates, manakov, L, pavg)
    try:
        if dual:
            planes = [np.asfortranarray(v.copy()) for v in (sx.real, sx.imag, sy.real, sy.imag)]
            lib.call("plx_matrix_ssfm", *[vp(p) for p in planes], C.byref(d), vp(db0), vp(th), vp(ep), C.byref(fd), C.byref(nc))
            rc, ofd, onc, ox, oy = oracle.matrix_ssfm(sx, sy, t["betat"], t["db1"], dzm, dph, gam, t["alphalin"], L, nplates, manakov, fls, db0, th, ep)
            if rc: print("oracle refuses:", tag); continue
            g = planes[0] + 1j * planes[1]; err = np.abs(g - ox).max() / np.abs(ox).max()
            err = max(err, np.abs(planes[2] + 1j * planes[3] - oy).max() / max(np.abs(oy).max(), 1e-300))
        else:
            planes = [np.asfortranarray(v.copy()) for v in (sx.real, sx.imag)]
            lib.call("plx_scalar_ssfm", vp(planes[0]), vp(planes[1]), C.byref(d), C.byref(fd), C.byref(nc))
            ofd, onc, ou = oracle.scalar_ssfm(sx, t["betat"], dzm, dph, gam, t["alphalin"], L, fls)
            err = np.abs(planes[0] + 1j * planes[1] - ou).max() / np.abs(ou).max()
    except _abi.PolmuxError as e:
        print("refused (%s): %s" % (e, tag)); continue
    bar, cond = 1e-9, None
    if err >= bar:
        # the step rule can be ill-conditioned ("Conditioning of the step rule": e.g. SPM + channel walk-off without
        # GVD on a coarse grid): measure the ORACLE's own sensitivity to a 1e-15 input perturbation and scale the bar with it
        if dual:
            _, _, _, px2, py2 = oracle.matrix_ssfm(sx * (1 + 1e-15), sy, t["betat"], t["db1"], dzm, dph, gam, t["alphalin"], L, nplates, manakov, fls, db0, th, ep)
            cond = max(np.abs(px2 - ox).max() / np.abs(ox).max(), np.abs(py2 - oy).max() / max(np.abs(oy).max(), 1e-300))
        else:
            _, _, ou2 = oracle.scalar_ssfm(sx * (1 + 1e-15), t["betat"], dzm, dph, gam, t["alphalin"], L, fls)
            cond = np.abs(ou2 - ou).max() / np.abs(ou).max()
        bar = max(bar, 100 * cond)
    ok = nc.value == onc and err < bar and abs(fd.value - ofd) <= 1e-12 * abs(ofd)
    if cond is None:
        worst = max(worst, err)
    else:
        illc += 1
        print("ill-conditioned case (%s): error %.3g against the oracle's own 1e-15 sensitivity %.3g: %s" % (tag, err, cond, "accepted" if ok else "REJECTED"))
    ran += 1
    if not ok:
        bad += 1
        print("MISMATCH", tag, "nc", nc.value, onc, "err %.3g" % err, "fd", fd.value, ofd)
print("%d cases run, worst relative field error %.3g (well-conditioned cases), %d ill-conditioned case(s) judged against the oracle's own sensitivity, mismatches %d" % (ran, worst, illc, bad))

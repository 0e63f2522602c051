# dev tool: CRC32 of the propagated field for a fixed set of fibre configurations through the gateways (no oracle involved):
# a refactoring that is meant to change no behaviour must reproduce every line bit for bit.
#   usage: python scripts/crc_set.py [seed] [small cases] [large cases] > crc.txt ; diff against the file of the previous build
import ctypes as C, os, sys, math, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (first: a development build of the library must find torch's HIP runtime already loaded)
from polmux_amd import _abi, synth
from polmux_amd._abi import SsfmDesc
from polmux_amd.fiber import parse_flag, fiber_tables
from polmux_amd.gstate import GSTATE
import polmux_amd as px
for _t in [t for t in sys.argv[1:] if t.startswith("lib=")]:       # lib=<name>: polmux_amd/lib/libpolmux_hip_<name>.so (a development build)
    _abi.LIB_PATH = os.path.join(os.path.dirname(_abi.LIB_PATH), "libpolmux_hip_%s.so" % _t[4:])
    sys.argv.remove(_t)
lib = _abi.get()
vp = lambda a: C.c_void_p(a.ctypes.data)
# (trailing field=value arguments: a plan-time tuning for every plan of the run, e.g. colx_lite=1)
_tune = {k: int(v) for k, v in (t.split("=") for t in sys.argv[1:] if "=" in t)}
sys.argv = [t for t in sys.argv if "=" not in t]
if _tune:
    lib.call("plx_ssfm_tuning_override", C.byref(lib.tuning(**_tune)))
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
nsmall = int(sys.argv[2]) if len(sys.argv) > 2 else 60
nlarge = int(sys.argv[3]) if len(sys.argv) > 3 else 16
r = np.random.default_rng(seed)


def run(lg, ntl, dual, nfc, flag, nplates, manakov, L, pavg, dphimax=5e-3, disp=17.0, slope=0.0, alphadB=0.2):
    nsl = lg - ntl
    nsymb, nt = 1 << nsl, 1 << ntl
    n = nsymb * nt
    px.reset_all(nsymb, nt, nfc); GSTATE.SYMBOLRATE = 28.0
    GSTATE.NCH = nfc; GSTATE.LAMBDA = 1550.0 + 0.4 * (np.arange(nfc) - (nfc - 1) / 2) if nfc > 1 else np.array([1550.0])
    x = dict(length=L, alphadB=alphadB, aeff=80.0, n2=2.7e-20, disp=disp, slope=slope, dphimax=dphimax, dzmax=2e4, dgd=0.3,
             manakov="yes" if manakov else "no"); x["lambda"] = 1550.0
    try:
        fls, dph, dzm = parse_flag(flag, nfc, x)
    except ValueError:
        return None
    dgdrms = math.sqrt(3 * math.pi / 8) * 0.3 / math.sqrt(nplates) if fls[1] else 0.0
    t = fiber_tables(x, fls, nfc, dgdrms)
    cols = [synth.pdm_qpsk_field(nsymb, nt, pavg * (1 + 0.3 * k), 2 + 2 * k, 3 + 2 * k) for k in range(nfc)]
    sx = np.stack([c[0] for c in cols], 1); sy = np.stack([c[1] for c in cols], 1)
    rr = np.random.default_rng(1000 + lg * 131 + nplates)
    if fls[1]:
        db0 = rr.random(nplates) * 2 * np.pi - np.pi; th = rr.random(nplates) * np.pi - np.pi / 2; ep = 0.5 * np.arcsin(rr.random(nplates) * 2 - 1)
    else:
        db0 = th = ep = np.zeros(1)
    d = SsfmDesc(); d.nfft, d.nfc, d.dual_pol, d.max_frames = n, nfc, int(dual), 1
    for i in range(4): d.fls[i] = fls[i]
    d.dzmaxt, d.dphimaxt, d.alphalin, d.length, d.nplates, d.manakov = dzm, dph, t["alphalin"], L, nplates, int(manakov)
    gam = np.ascontiguousarray(t["gam"]); d.gam, d.betat, d.db1 = gam.ctypes.data, t["betat"].ctypes.data, t["db1"].ctypes.data
    fd, nc = C.c_double(), C.c_int32()
    try:
        if dual:
            planes = [np.asfortranarray(v.copy()) for v in (sx.real, sx.imag, sy.real, sy.imag)]
            lib.call("plx_matrix_ssfm", *[vp(p) for p in planes], C.byref(d), vp(db0), vp(th), vp(ep), C.byref(fd), C.byref(nc))
        else:
            planes = [np.asfortranarray(v.copy()) for v in (sx.real, sx.imag)]
            lib.call("plx_scalar_ssfm", vp(planes[0]), vp(planes[1]), C.byref(d), C.byref(fd), C.byref(nc))
    except _abi.PolmuxError as e:
        return "refused"
    crc = 0
    for p in planes:
        crc = zlib.crc32(p.tobytes(), crc)
    return "%08x nc %d fd %r" % (crc, nc.value, fd.value)


def draw(lo, hi, count):
    done = 0
    while done < count:
        lg = int(r.integers(lo, hi + 1))
        ntl = int(r.choice([3, 4, 5] if lg < 16 else [5, 6, 7]))
        if (lg - ntl) % 2: lg -= 1
        if lg - ntl < 6: continue
        dual = bool(r.integers(0, 2))
        nfc = int(r.choice([1, 1, 2, 3]))
        flag = "".join([r.choice(["g", "-"]), (r.choice(["p", "-"]) if dual else "-"), r.choice(["s", "-"]), (r.choice(["x", "-"]) if (not dual and nfc > 1) else "-")])
        if flag == "----": flag = "g---"
        nplates = int(r.choice([1, 3, 10, 37])) if flag[1] == "p" else 1
        manakov = bool(r.integers(0, 2)) and flag[1] == "p"
        L = float(r.choice([5e3, 2e4, 8e4] if hi < 16 else [2e3, 5e3, 1e4]))
        pavg = float(r.choice([0.5, 2.0, 8.0]))
        res = run(lg, ntl, dual, nfc, flag, nplates, manakov, L, pavg, dphimax=float(r.choice([5e-3, 2e-2])), disp=float(r.choice([17.0, 4.0, -2.0])),
                  slope=float(r.choice([0.0, 0.057])), alphadB=float(r.choice([0.0, 0.2])))
        if res is None: continue
        print("2^%d nt %d dual %d nfc %d %s plates %d manakov %d L %g P %g: %s" % (lg, 1 << ntl, dual, nfc, flag, nplates, manakov, L, pavg, res))
        done += 1


draw(8, 13, nsmall)
draw(16, 19, nlarge)
# the BASELINE shapes and the row-pass families around them
for (lg, ntl, dual, nfc, flag, npl, L, P) in [(16, 6, 1, 1, "g-s-", 1, 8e4, 2.0), (16, 6, 1, 1, "gps-", 100, 8e4, 2.0), (16, 6, 1, 16, "gps-", 20, 2e4, 2.0),
                                              (20, 6, 1, 1, "g-s-", 1, 2e4, 2.0), (20, 6, 1, 1, "gps-", 10, 1e4, 2.0), (20, 6, 0, 1, "g-s-", 1, 1e4, 2.0),
                                              (16, 6, 0, 3, "g-sx", 1, 2e4, 2.0), (18, 6, 1, 1, "gps-", 10, 2e4, 2.0), (18, 6, 1, 1, "g-s-", 1, 2e4, 4.0),
                                              (14, 4, 1, 1, "g-s-", 1, 8e4, 2.0), (13, 5, 0, 1, "g-s-", 1, 8e4, 2.0), (12, 4, 1, 1, "gps-", 5, 8e4, 2.0)]:
    print("2^%d nt %d dual %d nfc %d %s plates %d L %g P %g: %s" % (lg, 1 << ntl, dual, nfc, flag, npl, L, P, run(lg, ntl, bool(dual), nfc, flag, npl, False, L, P)))

# batches through the resident tier (teams, frame barrier, active list, chunked enqueue): frames of very different step counts
import torch
for (nsymb, nt, F, flag, L) in [(1024, 64, 37, "g-s-", 8e4), (1024, 64, 40, "gps-", 2e4), (256, 16, 69, "g-s-", 8e4), (1024, 16, 5, "--s-", 2e4), (16384, 64, 3, "g-s-", 1e4)]:
    n = nsymb * nt
    nplates = 10 if flag[1] == "p" else 1
    px.reset_all(nsymb, nt, 1); GSTATE.SYMBOLRATE = 28.0; px.lasersource(1.0, 1550.0)
    x = dict(length=L, alphadB=0.2, aeff=80.0, n2=2.7e-20, disp=17.0, slope=0.0, dphimax=5e-3, dzmax=2e4, dgd=0.2, manakov="no"); x["lambda"] = 1550.0
    fls, dph, dzm = parse_flag(flag, 1, x)
    t = fiber_tables(x, fls, 1, math.sqrt(3 * math.pi / 8) * 0.2 / math.sqrt(nplates) if fls[1] else 0.0)
    ux0, uy0, _, _ = synth.pdm_qpsk_field(nsymb, nt, 1.0)
    rb = np.random.default_rng(n + F)
    scale = np.sqrt(10 ** rb.uniform(-2.0, 1.0, F))
    d = SsfmDesc(); d.nfft, d.nfc, d.dual_pol, d.max_frames = n, 1, 1, F
    for i in range(4): d.fls[i] = fls[i]
    d.dzmaxt, d.dphimaxt, d.alphalin, d.length, d.nplates, d.manakov = dzm, dph, t["alphalin"], L, nplates, 0
    gam = np.ascontiguousarray(t["gam"]); d.gam, d.betat, d.db1 = gam.ctypes.data, t["betat"].ctypes.data, t["db1"].ctypes.data
    plan = C.c_void_p(); lib.call("plx_ssfm_create", C.byref(plan), C.byref(d))
    if fls[1]:
        db0 = rb.random((F, nplates)) * 2 * np.pi - np.pi; th = rb.random((F, nplates)) * np.pi - np.pi / 2; ep = 0.5 * np.arcsin(rb.random((F, nplates)) * 2 - 1)
        lib.call("plx_ssfm_set_birefringence", plan, db0.ctypes.data, th.ctypes.data, ep.ctypes.data, F)
    ux = torch.from_numpy(np.stack([ux0 * s for s in scale])).cuda(); uy = torch.from_numpy(np.stack([uy0 * s for s in scale])).cuda()
    lib.call("plx_ssfm_propagate_dev", plan, ux.data_ptr(), uy.data_ptr(), F, torch.cuda.current_stream().cuda_stream)
    ncyc = np.zeros(F, np.int32); fdz = np.zeros(F)
    lib.call("plx_ssfm_results", plan, F, fdz.ctypes.data, ncyc.ctypes.data)
    lib.call("plx_ssfm_destroy", plan)
    crc = zlib.crc32(uy.cpu().numpy().tobytes(), zlib.crc32(ux.cpu().numpy().tobytes()))
    print("batch n %d F %d %s L %g: %08x ncycle %d..%d sum %d fd0 %r" % (n, F, flag, L, crc, ncyc.min(), ncyc.max(), int(ncyc.sum()), float(fdz[0])))

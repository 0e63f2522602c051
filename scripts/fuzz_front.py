# dev tool: random coherent front-end configurations (RxPdmCohQpsk on the device) vs oracle/front.py; random inverse_pmd links.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import front, pmdinv
import polmux_amd as px
from polmux_amd import rxfront, synth
from polmux_amd.gstate import GSTATE, to_host_field
r = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = ran = 0
worst = 0.0
ftypes = ["gauss", "butt2", "butt4", "butt6", "bessel5", "rc1", "rc2", "supergauss", "ideal", "movavg"]
for case in range(ncase):
    nsl, ntl = int(r.choice([6, 8, 10])), int(r.choice([3, 4, 5, 6]))
    nsymb, nt = 1 << nsl, 1 << ntl
    if nsymb * nt < 256 or nsymb * nt > (1 << 16): continue
    dual = bool(r.integers(0, 2))
    px.reset_all(nsymb, nt, 1); GSTATE.SYMBOLRATE = float(r.choice([10.0, 28.0])); px.lasersource(2.0, 1550.0)
    sx, sy, _, _ = synth.pdm_qpsk_field(nsymb, nt, 2.0)
    px.create_field("sepfields", sx, sy if dual else None)
    GSTATE.DELAY = r.random((2 if dual else 1, 1)) * 0.4
    of, ef = str(r.choice(ftypes[:8])), str(r.choice(ftypes))
    rp = dict(rec="coherent", ts=0, oftype=of, obw=float(r.choice([1.2, 1.9, 3.0])), oord=int(r.choice([2, 3])), eftype=ef,
              ebw=float(r.choice([0.5, 0.65, 0.9])), eord=4, delay="theory", lopower=float(r.choice([0, 3, -2])), sps=nt,
              workatbaudrate=bool(r.integers(0, 2)), applyadc=bool(r.integers(0, 2)), adcbits=int(r.choice([3, 5, 8])))
    rp["lambda"] = 1550.0
    if r.integers(0, 2): rp["pdtype"] = "normal"
    if r.integers(0, 2): rp["lodetuning"] = float(r.choice([0.0, 1.7, 3.3])) * GSTATE.SYMBOLRATE * 1e9 / nsymb
    if r.integers(0, 2): rp["lophasenoise"] = np.cumsum(r.standard_normal(nsymb * nt)) * 0.01
    if r.integers(0, 3) == 0: rp.update(dpost=float(r.choice([-500.0, 1360.0])), slopez=0.0)
    isy = dual and bool(r.integers(0, 2))
    pat = np.zeros((nsymb, 2 if isy else 1))
    if nt // (1 if rp["workatbaudrate"] else 2) < 1: continue
    fr, shifts, info = rxfront.rx_plan(1, rp, isy, 1)
    ux = GSTATE.FIELDX.clone(); uy = GSTATE.FIELDY.clone() if fr.dual else None
    out = fr.run(ux, uy, shifts); fr.close(); torch.cuda.synchronize()
    hx = to_host_field(GSTATE.FIELDX)[:, 0]; hy = to_host_field(GSTATE.FIELDY)[:, 0] if fr.dual else None
    want = front.receiver_cohmix(hx, hy, info["hopt"], info["elo"], info["hel"], rp.get("pdtype") != "normal")
    cols = [ux[0].real, ux[0].imag] + ([uy[0].real, uy[0].imag] if fr.dual else [])
    got = torch.stack(cols, 1).cpu().numpy()
    e1 = np.abs(got - want).max() / np.abs(want).max()
    bits = rp["adcbits"] if rp["applyadc"] else 0
    rx = front.rx_front(got, fr.dual, bits, shifts, info["decim"], info["fir"])
    o = out[0].cpu().numpy().T
    e2 = np.abs(o - rx).max() / np.abs(rx).max()
    worst = max(worst, e1, e2); ran += 1
    if not (e1 < 1e-10 and e2 < 1e-13):
        bad += 1; print("FRONT MISMATCH", nsymb, nt, dual, isy, {k: v for k, v in rp.items() if k != "lophasenoise"}, "cur %.3g samples %.3g" % (e1, e2))
for case in range(ncase):                                    # ---- inverse_pmd
    nsymb, nt = int(r.choice([64, 256, 1024])), int(r.choice([8, 16, 32]))
    n = nsymb * nt
    if n < 256: continue
    px.reset_all(nsymb, nt, 1); GSTATE.SYMBOLRATE = 10.0; px.lasersource(1.0, 1550.0)
    sx, sy, _, _ = synth.pdm_qpsk_field(nsymb, nt, 1.0)
    px.create_field("sepfields", sx, sy)
    omega = 2 * np.pi * 10.0 * GSTATE.FN
    brf = []
    for k in range(int(r.integers(1, 4))):
        npl = int(r.choice([1, 2, 7, 30]))
        brf.append(dict(db0=r.random(npl) * 2 * np.pi - np.pi, theta=r.random(npl) * np.pi - np.pi / 2, epsilon=0.5 * np.arcsin(r.random(npl) * 2 - 1),
                        lcorr=float(r.choice([1e3, 5e3])), betat=0.5 * omega ** 2 * float(r.choice([-2.17e-8, 5e-9])), db1=float(r.random()) / npl / 10.0 * omega))
    opts = [None, dict(gvd="no"), dict(mat=np.array([[np.cos(0.3), 1j * np.sin(0.3)], [1j * np.sin(0.3), np.cos(0.3)]]))][int(r.integers(0, 3))]
    Uinv, U, wx, wy = pmdinv.inverse_pmd(brf, sx, sy, opts)
    gUinv, gU = px.inverse_pmd(brf, opts, nargout=2)
    gx, gy = to_host_field(GSTATE.FIELDX)[:, 0], to_host_field(GSTATE.FIELDY)[:, 0]
    e = max(np.abs(gx - wx).max(), np.abs(gy - wy).max()) / np.abs(wx).max()
    e = max(e, np.abs(gU - U).max(), np.abs(gUinv - Uinv).max())
    worst = max(worst, e); ran += 1
    if not e < 1e-11:
        bad += 1; print("PMDINV MISMATCH", nsymb, nt, len(brf), opts, "%.3g" % e)
print("%d cases run, worst error %.3g, mismatches %d" % (ran, worst, bad))

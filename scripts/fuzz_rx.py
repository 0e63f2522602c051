# dev tool: random receiver configurations (CDE_OFDE sizes; DspPdmCohQpsk options) through the host mirror vs the oracle.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from oracle import plxo as oracle
import polmux_amd as px
from polmux_amd.gstate import GSTATE
r = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = ran = 0
worst = 0.0
for case in range(ncase):                                   # ---- CDE_OFDE
    N = int(2 ** r.integers(4, 11)); L = int(r.choice([N // 2, N // 4, max(2, N // 2 - 2), N]))
    if L % 2 or L < 2: continue
    nx = int(r.integers(N, 6000 + N))      # (nx < fftLength makes the reference shrink the FFT to nx: powers of two only here)
    x = r.standard_normal(nx) + 1j * r.standard_normal(nx); y = r.standard_normal(nx) + 1j * r.standard_normal(nx)
    D, S, span = float(r.choice([17e-6, -80e-6, 4e-6])), float(r.choice([0.0, 0.057e-6])), float(r.choice([1e3, 8e4, 4e5]))
    ex, ey, rc = oracle.cde_ofde(x, y, 56e9, 1550e-9, span, D, S, N, L)
    gx, gy, _ = px.CDE_OFDE(x, y, 56e9, 1550e-9, span, D, S, N, L)
    if rc:
        ok = np.size(gx) == 0
    else:
        e = max(np.abs(gx - ex).max(), np.abs(gy - ey).max()) / max(np.abs(ex).max(), 1e-300)
        worst = max(worst, e); ok = e < 1e-11
    ran += 1
    if not ok: bad += 1; print("CDE MISMATCH nx=%d N=%d L=%d rc=%d" % (nx, N, L, rc))
GSTATE.POWER = np.array([2.0])
for case in range(ncase):                                   # ---- DspPdmCohQpsk
    nsymb = int(r.choice([64, 256, 1000, 1024])); wab = bool(r.integers(0, 2)); ncol = int(r.choice([1, 2]))
    Lin = nsymb if wab else 2 * nsymb
    sym = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (nsymb, ncol))))
    th = r.random() * 2 - 1
    if ncol == 2:
        M = np.array([[np.cos(th), np.sin(th) * np.exp(0.4j)], [-np.sin(th) * np.exp(-0.4j), np.cos(th)]]); sym = sym @ M
    ph = np.cumsum(r.standard_normal(nsymb) * 0.01)[:, None] + 2 * np.pi * 0.002 * np.arange(nsymb)[:, None]
    sig = 4 * np.sqrt(2.0) * sym * np.exp(1j * ph) + 0.15 * (r.standard_normal((nsymb, ncol)) + 1j * r.standard_normal((nsymb, ncol)))
    rx = np.zeros((Lin, ncol), dtype=complex)
    if wab: rx[:] = sig
    else: rx[0::2] = sig; rx[1::2] = 0.5 * (sig + np.roll(sig, -1, 0))
    method = str(r.choice(["cma", "easi", "combo", "singlepol"])) if ncol == 2 else "cma"
    kw = dict(workatbaudrate=wab, applynlr=bool(r.integers(0, 2)), nlralpha=0.01, applypol=(ncol == 2 and bool(r.integers(0, 4))), polmethod=method,
              cma_mu=float(r.choice([1 / 600, 1 / 2000])), cma_taps=int(r.choice([1, 3, 7, 9])), easi_mu=float(r.choice([1 / 600, 1 / 3000])),
              freqavg=int(r.choice([0, 20, 500])), phasavg=int(r.choice([0, 3, 7])), poworder=int(r.choice([0, 1, 2])), modorder=int(r.choice([1, 2])))
    if kw["freqavg"] * 2 + 1 > nsymb: kw["freqavg"] = 20
    op = oracle.dsp_params(power_mw=2.0, cma_phizero=0.1, easi_phizero=-0.2, **kw)
    ref = oracle.dsp_pdm_coh_qpsk(rx, op)
    dsp = dict(workatbaudrate=wab, applynlr=kw["applynlr"], nlralpha=0.01, applypol=kw["applypol"], polmethod=method,
               cmaparams=dict(R=[1, 1], mu=kw["cma_mu"], taps=kw["cma_taps"], txpolars=2, phizero=0.1),
               easiparams=dict(mu=kw["easi_mu"], txpolars=2, phizero=-0.2), modorder=kw["modorder"], freqavg=kw["freqavg"],
               phasavg=kw["phasavg"], poworder=kw["poworder"])
    got = px.DspPdmCohQpsk(rx, dsp, 1)
    e = np.abs(got - ref).max()
    worst = max(worst, e); ran += 1
    if not (got.shape == ref.shape and e < 1e-7):
        bad += 1; print("DSP MISMATCH", nsymb, ncol, kw, "err %.3g" % e)
print("%d cases run, worst error %.3g, mismatches %d" % (ran, worst, bad))

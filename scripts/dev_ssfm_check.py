"""Developer check (GPU): SSFM kernels vs the CPU oracle + first timings.  Not a test."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import plxo  # noqa: E402  (dev check only)
from polmux_amd._abi import SsfmDesc  # noqa: E402

lib = C.CDLL(os.environ.get("PLX_LIB", os.path.join(ROOT, "polmux_amd", "lib", "libpolmux_hip.so")))
lib.plx_last_error.restype = C.c_char_p
vp = C.c_void_p


def make(n, nfc, fls, nplates, seed, amp):
    from polmux_amd import synth
    nt = 64 if n >= 4096 else 16
    if int(np.log2(n // nt)) % 2:
        nt //= 2
    nsymb = n // nt
    cols_x, cols_y = [], []
    for k in range(nfc):
        x, y, _, _ = synth.pdm_qpsk_field(nsymb, nt, amp * (1 + 0.2 * k), seed_x=seed + 2 + 2 * k, seed_y=seed + 3 + 2 * k)
        cols_x.append(x)
        cols_y.append(y)
    ux, uy = np.asfortranarray(np.stack(cols_x, 1)), np.asfortranarray(np.stack(cols_y, 1))
    omega = 2 * np.pi * 28 * synth.fn_grid(nsymb, nt)
    betat = np.asfortranarray(np.stack([(0.5 * omega ** 2 * -2.17e-8 * fls[0] + 6.8e-9 * k * omega) for k in range(nfc)], axis=1))
    db1 = np.asfortranarray(np.stack([np.sqrt(3 * np.pi / 8) * 0.1 / np.sqrt(nplates) / 28 * omega * fls[1] for k in range(nfc)], axis=1))
    return ux, uy, betat, db1


def run(n, nfc, dual, fls, L, alpha, gam, dzmax, dphimax, nplates=1, manakov=0, seed=0, amp=8.0, check=True):
    ux, uy, betat, db1 = make(n, nfc, fls, nplates, seed, amp)
    gamv = np.ascontiguousarray(np.broadcast_to(np.atleast_1d(gam).astype(float), (nfc,)))
    r = np.random.default_rng(seed + 77)
    if fls[1]:
        db0, th, ep = r.random(nplates) * 2 * np.pi - np.pi, r.random(nplates) * np.pi - 0.5 * np.pi, 0.5 * np.arcsin(r.random(nplates) * 2 - 1)
    else:
        db0, th, ep = np.zeros(nplates), np.zeros(nplates), np.zeros(nplates)
    d = SsfmDesc()
    d.nfft, d.nfc, d.dual_pol, d.max_frames = n, nfc, dual, 1
    for i in range(4):
        d.fls[i] = fls[i]
    d.dzmaxt, d.dphimaxt, d.alphalin, d.length, d.nplates, d.manakov = dzmax, dphimax, alpha, L, nplates, manakov
    d.gam, d.betat, d.db1 = gamv.ctypes.data, betat.ctypes.data, db1.ctypes.data
    xr, xi = np.asfortranarray(ux.real.copy()), np.asfortranarray(ux.imag.copy())
    yr, yi = np.asfortranarray(uy.real.copy()), np.asfortranarray(uy.imag.copy())
    fd, nc = C.c_double(), C.c_int32()
    t = time.time()
    if dual:
        rc = lib.plx_matrix_ssfm(vp(xr.ctypes.data), vp(xi.ctypes.data), vp(yr.ctypes.data), vp(yi.ctypes.data), C.byref(d),
                                 vp(db0.ctypes.data), vp(th.ctypes.data), vp(ep.ctypes.data), C.byref(fd), C.byref(nc))
    else:
        rc = lib.plx_scalar_ssfm(vp(xr.ctypes.data), vp(xi.ctypes.data), C.byref(d), C.byref(fd), C.byref(nc))
    te = time.time() - t
    if rc:
        print("FAILED rc", rc, lib.plx_last_error())
        return False
    tag = f"n=2^{int(np.log2(n))} nfc={nfc} {'dual' if dual else 'scalar'} fls={fls} nplates={nplates} manakov={manakov}"
    if not check:
        print(f"{tag}: ncycle {nc.value} firstdz {fd.value:.6g} gateway {te*1e3:.1f} ms (no oracle check)")
        return True
    t = time.time()
    if dual:
        orc, ofd, onc, ox, oy = plxo.matrix_ssfm(ux, uy, betat, db1, dzmax, dphimax, gamv, alpha, L, nplates, manakov, fls, db0, th, ep)
        err = max(np.abs((xr + 1j * xi) - ox).max() / np.abs(ox).max(), np.abs((yr + 1j * yi) - oy).max() / np.abs(oy).max())
    else:
        ofd, onc, ox = plxo.scalar_ssfm(ux, betat, dzmax, dphimax, gamv, alpha, L, fls)
        err = np.abs((xr + 1j * xi) - ox).max() / np.abs(ox).max()
    to = time.time() - t
    ok = err < 1e-9 and nc.value == onc
    print(f"{'OK ' if ok else 'BAD'} {tag}: ncycle {nc.value}/{onc} firstdz {fd.value:.6g}/{ofd:.6g} relerr {err:.2e} gateway {te*1e3:.1f} ms oracle {to*1e3:.0f} ms")
    return ok


if __name__ == "__main__":
    A, G = 4.6e-5, 1.3e-6
    good = True
    good &= run(256, 1, 1, [1, 0, 0, 0], 8e4, A, G, 8e4, np.inf)
    good &= run(4096, 1, 1, [1, 0, 0, 0], 8e4, A, G, 2e3, np.inf)      # 40 linear steps
    good &= run(4096, 1, 1, [0, 0, 1, 0], 8e4, A, G, 2e3, 5e-3)        # NL only, many steps
    good &= run(4096, 1, 1, [0, 0, 1, 0], 8e4, A, G, 2e3, 5e-3, manakov=1)
    good &= run(4096, 1, 0, [0, 0, 1, 0], 8e4, A, G, 2e3, np.inf)      # scalar NL only fixed steps
    good &= run(1 << 16, 1, 1, [1, 0, 0, 0], 8e4, A, G, 2e3, np.inf)
    good &= run(256, 1, 1, [1, 0, 1, 0], 8e4, A, G, 2e4, 5e-3)
    good &= run(1024, 1, 1, [1, 0, 1, 0], 8e4, A, G, 2e4, 5e-3, manakov=1)
    good &= run(4096, 1, 1, [1, 1, 1, 0], 8e4, A, G, 2e4, 5e-3, nplates=20)
    good &= run(2048, 1, 0, [1, 0, 1, 0], 8e4, A, G, 2e4, 5e-3)
    good &= run(1024, 3, 0, [1, 0, 1, 1], 8e4, A, [1.1e-6, 1.2e-6, 1.3e-6], 2e4, 5e-3)
    good &= run(512, 2, 1, [1, 0, 1, 0], 8e4, A, [1.1e-6, 1.2e-6], 2e4, 5e-3)
    good &= run(1 << 16, 1, 1, [1, 0, 1, 0], 8e4, A, G, 2e4, 5e-3)
    good &= run(1 << 16, 1, 1, [1, 1, 1, 0], 8e4, A, G, 2e4, 5e-3, nplates=100)
    good &= run(1 << 17, 1, 1, [1, 0, 1, 0], 8e4, A, G, 2e4, 5e-3)
    good &= run(1 << 20, 1, 1, [1, 0, 1, 0], 8e4, A, G, 2e4, 5e-3)
    good &= run(1 << 20, 1, 0, [1, 0, 1, 0], 8e4, A, G, 2e4, 5e-3, check=False)
    print("ALL OK" if good else "SOME FAILED")
    sys.exit(0 if good else 1)

"""Run the *kernel sources* (polmux_amd/csrc/*.hip) under the host emulator of
tests/emu and compare with the CPU oracle.  This is a CPU-side sanity net for
indexing and synchronisation (it is also what the ASan/UBSan build runs); the
parity tests proper are the -m gpu tests, which call the hipcc-built library."""
import ctypes as C

import numpy as np
import pytest

from polmux_amd import synth
from polmux_amd._abi import DspParams, SsfmDesc


@pytest.fixture(scope="module")
def emu():
    from tests import _emu
    return _emu.binding()


def _vp(a):
    return C.c_void_p(a.ctypes.data)


def _il(z):
    """complex array -> interleaved float64 copy (C order over the given array's memory order)"""
    z = np.ascontiguousarray(z, dtype=np.complex128)
    return z.view(np.float64).copy()


def _qpsk_field(n, nt, pavg, seeds=(2, 3)):
    ux, uy, bits, pw = synth.pdm_qpsk_field(n // nt, nt, pavg, *seeds)
    return ux, uy, bits, pw


def _desc(n, nfc, dual, fls, L, alpha, gam, dzmax, dphimax, betat, db1, nplates=1, manakov=0, frames=1):
    d = SsfmDesc()
    d.nfft, d.nfc, d.dual_pol, d.max_frames = n, nfc, dual, frames
    for i in range(4):
        d.fls[i] = fls[i]
    d.dzmaxt, d.dphimaxt, d.alphalin, d.length, d.nplates, d.manakov = dzmax, dphimax, alpha, L, nplates, manakov
    d._keep = (np.ascontiguousarray(gam, dtype=float), np.asfortranarray(betat), np.asfortranarray(db1))
    d.gam, d.betat, d.db1 = d._keep[0].ctypes.data, d._keep[1].ctypes.data, d._keep[2].ctypes.data
    return d


def _tables(n, nt, fls, nplates, nfc=1):
    omega = 2 * np.pi * 28 * synth.fn_grid(n // nt, nt)
    betat = np.stack([0.5 * omega ** 2 * -2.17e-8 * fls[0] + 6.8e-9 * k * omega for k in range(nfc)], 1)
    db1 = np.stack([np.sqrt(3 * np.pi / 8) * 0.1 / np.sqrt(nplates) / 28 * omega * fls[1] for _ in range(nfc)], 1)
    return betat, db1


@pytest.mark.parametrize("tables", [True, False])
def test_emu_matrix_ssfm_batch_with_pmd(emu, oracle, tune, tables):
    """two frames with their own PMD realisation and launch power advance in lock-step launches -- with the trunk phasors from
    the row / column tables of k_pmd_tab (db1 linear in the frequency index: the default) and with one exponential per bin and
    trunk (PLX_SSFM_NO_PMD_TAB=1, also what a plan with a non-linear db1 takes)"""
    if not tables:
        tune.setenv("PLX_SSFM_NO_PMD_TAB", "1")
    n, nt, nplates, L = 512, 8, 6, 2e4
    fls = [1, 1, 1, 0]
    betat, db1 = _tables(n, nt, fls, nplates)
    r = np.random.default_rng(3)
    brf = [(r.random(nplates) * 2 * np.pi - np.pi, r.random(nplates) * np.pi - np.pi / 2, 0.5 * np.arcsin(r.random(nplates) * 2 - 1))
           for _ in range(2)]
    fields = [_qpsk_field(n, nt, p)[:2] for p in (6.0, 12.0)]
    d = _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-6], 1e4, 2e-2, betat, db1, nplates=nplates, frames=2)
    plan = C.c_void_p()
    emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
    db0 = np.concatenate([b[0] for b in brf]); th = np.concatenate([b[1] for b in brf]); ep = np.concatenate([b[2] for b in brf])
    emu.call("plx_ssfm_set_birefringence", plan, _vp(db0), _vp(th), _vp(ep), 2)
    ux = _il(np.stack([f[0] for f in fields])); uy = _il(np.stack([f[1] for f in fields]))
    emu.call("plx_ssfm_propagate_dev", plan, _vp(ux), _vp(uy), 2, None)
    first = np.zeros(2); ncyc = np.zeros(2, np.int32)
    emu.call("plx_ssfm_results", plan, 2, _vp(first), _vp(ncyc))
    rows, steps = C.c_int64(), C.c_int64()
    emu.call("plx_ssfm_stats", plan, C.byref(rows), C.byref(steps))
    emu.call("plx_ssfm_destroy", plan)
    gx = ux.view(np.complex128).reshape(2, n); gy = uy.view(np.complex128).reshape(2, n)
    for f in range(2):
        rc, ofd, onc, ox, oy = oracle.matrix_ssfm(fields[f][0], fields[f][1], betat, db1, 1e4, 2e-2, [1.3e-6], 4.6e-5, L,
                                                  nplates, 0, fls, *brf[f])
        assert rc == 0 and ncyc[f] == onc and first[f] == pytest.approx(ofd, rel=1e-13)
        assert np.abs(gx[f] - ox[:, 0]).max() < 1e-11 * np.abs(ox).max()
        assert np.abs(gy[f] - oy[:, 0]).max() < 1e-11 * np.abs(oy).max()
    assert ncyc[0] != ncyc[1]          # the frames really did take different step sequences
    assert steps.value == int(ncyc.sum()) * n and rows.value >= ncyc.max()


def test_emu_scalar_xpm_gateway(emu, oracle):
    n, nt, nfc = 512, 8, 2
    fls = [1, 0, 1, 1]
    betat, db1 = _tables(n, nt, fls, 1, nfc)
    u = np.asfortranarray(np.stack([_qpsk_field(n, nt, 6.0, (2 + k, 5 + k))[0] for k in range(nfc)], 1))
    gam = [1.2e-6, 1.3e-6]
    d = _desc(n, nfc, 0, fls, 3e4, 4.6e-5, gam, 1e4, 2e-2, betat, db1)
    ur, ui = np.asfortranarray(u.real.copy()), np.asfortranarray(u.imag.copy())
    fd, nc = C.c_double(), C.c_int32()
    emu.call("plx_scalar_ssfm", _vp(ur), _vp(ui), C.byref(d), C.byref(fd), C.byref(nc))
    ofd, onc, ou = oracle.scalar_ssfm(u, betat, 1e4, 2e-2, gam, 4.6e-5, 3e4, fls)
    assert nc.value == onc
    assert np.abs((ur + 1j * ui) - ou).max() < 1e-11 * np.abs(ou).max()


def test_emu_reference_error_paths(emu):
    betat = np.zeros((256, 2))
    d = _desc(256, 2, 1, [1, 0, 1, 1], 1e3, 0.0, [1e-6, 1e-6], 1e3, 5e-3, betat, betat)
    plan = C.c_void_p()
    from polmux_amd._abi import PolmuxError
    with pytest.raises(PolmuxError, match="CNLSE with separate fields"):          # fiber.m:854
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
    d = _desc(300, 1, 1, [1, 0, 0, 0], 1e3, 0.0, [1e-6], 1e3, 5e-3, np.zeros((300, 1)), np.zeros((300, 1)))
    with pytest.raises(PolmuxError, match="power of two"):
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))


def test_emu_fastexp(emu):
    x = np.linspace(-50, 50, 1001)
    yr, yi = np.zeros_like(x), np.zeros_like(x)
    emu.call("plx_fastexp", _vp(x), _vp(yr), _vp(yi), x.size)
    np.testing.assert_allclose(yr + 1j * yi, np.exp(1j * x), atol=1e-15)


@pytest.mark.parametrize("nx,N,L", [(2048, 256, 128), (700, 64, 32), (16, 16, 8), (1000, 128, 100)])
def test_emu_cde(emu, oracle, nx, N, L):
    r = np.random.default_rng(nx)
    x = r.standard_normal((2, nx)) + 1j * r.standard_normal((2, nx))
    H = oracle.cde_transfer(N, 56e9, 1.55e-6, 8e4, 17e-6, 0.0)
    plan = C.c_void_p()
    Hi = _il(H)
    emu.call("plx_cde_create", C.byref(plan), N, L, _vp(Hi))
    xi, yo = _il(x), np.zeros(4 * nx)
    emu.call("plx_cde_apply_dev", plan, _vp(xi), _vp(yo), nx, 2, None)
    emu.call("plx_cde_destroy", plan)
    y = yo.view(np.complex128).reshape(2, nx)
    for k in range(2):
        ref, rc = oracle.overlap_both_trans(x[k], H, L)
        assert rc == 0
        np.testing.assert_allclose(y[k], ref, rtol=0, atol=1e-12)


def test_emu_cde_gateway_and_checks(emu, oracle):
    from polmux_amd._abi import PolmuxError
    r = np.random.default_rng(5)
    nx = 600
    x = r.standard_normal(nx) + 1j * r.standard_normal(nx)
    y = r.standard_normal(nx) + 1j * r.standard_normal(nx)
    outs = [np.zeros(nx) for _ in range(4)]
    args = [np.ascontiguousarray(v) for v in (x.real, x.imag, y.real, y.imag)]
    emu.call("plx_cde_ofde", *[_vp(a) for a in args], nx, 56e9, 1.55e-6, 8e4, 17e-6, 0.0, 256, 128, *[_vp(o) for o in outs])
    ox, oy, rc = oracle.cde_ofde(x, y, 56e9, 1.55e-6, 8e4, 17e-6, 0.0, 256, 128)
    np.testing.assert_allclose(outs[0] + 1j * outs[1], ox, atol=1e-12)
    np.testing.assert_allclose(outs[2] + 1j * outs[3], oy, atol=1e-12)
    with pytest.raises(PolmuxError, match="L must be > 0"):                       # CDE_OFDE.m:77-78
        emu.call("plx_cde_ofde", *[_vp(a) for a in args], nx, 56e9, 1.55e-6, 8e4, 17e-6, 0.0, 256, 0, *[_vp(o) for o in outs])
    with pytest.raises(PolmuxError, match="shorter than filter"):                 # :79-80
        emu.call("plx_cde_ofde", *[_vp(a) for a in args], nx, 56e9, 1.55e-6, 8e4, 17e-6, 0.0, 256, 300, *[_vp(o) for o in outs])


def _mixed_qpsk(L, seed, noise=0.05, th=0.4):
    r = np.random.default_rng(seed)
    a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (L, 2))))
    J = np.array([[np.cos(th), np.sin(th) * np.exp(0.3j)], [-np.sin(th) * np.exp(-0.3j), np.cos(th)]])
    return a @ J + noise * (r.standard_normal((L, 2)) + 1j * r.standard_normal((L, 2)))


@pytest.mark.parametrize("taps,sps", [(1, 1), (3, 2), (7, 1), (7, 2), (15, 1)])
def test_emu_cmaadaptivefilter_gateway(emu, oracle, taps, sps):
    x = np.asfortranarray(_mixed_qpsk(60, taps))
    r = np.random.default_rng(taps)
    h1 = np.asfortranarray(0.3 * (r.standard_normal((taps, 2)) + 1j * r.standard_normal((taps, 2))))
    h2 = np.asfortranarray(0.3 * (r.standard_normal((taps, 2)) + 1j * r.standard_normal((taps, 2))))
    xr, xi = np.asfortranarray(x.real.copy()), np.asfortranarray(x.imag.copy())
    hs = [np.asfortranarray(v.copy()) for v in (h1.real, h1.imag, h2.real, h2.imag)]
    dimY = 60 - taps + 1
    yr, yi = np.zeros((dimY, 2), order="F"), np.zeros((dimY, 2), order="F")
    R = np.array([1.0, 1.2])
    emu.call("plx_cmaadaptivefilter", _vp(xr), _vp(xi), 60, *[_vp(h) for h in hs], float(taps), 1e-3, _vp(R), float(sps),
             _vp(yr), _vp(yi))
    y, g1, g2 = oracle.cmaadaptivefilter(x, h1, h2, taps, 1e-3, R, sps)
    np.testing.assert_allclose(yr + 1j * yi, y, atol=1e-13)
    np.testing.assert_allclose(hs[0] + 1j * hs[1], g1, atol=1e-13)       # taps updated IN PLACE (MEX contract)
    np.testing.assert_allclose(hs[2] + 1j * hs[3], g2, atol=1e-13)


def test_emu_filter_gateway_errors(emu):
    from polmux_amd._abi import PolmuxError
    z = np.zeros((16, 2), order="F"); h = np.zeros((4, 2), order="F"); y = np.zeros((16, 2), order="F"); R = np.ones(2)
    with pytest.raises(PolmuxError, match="Ntaps should be an ODD INTEGER."):
        emu.call("plx_cmaadaptivefilter", _vp(z), _vp(z), 16, _vp(h), _vp(h), _vp(h), _vp(h), 4.0, 1e-3, _vp(R), 1.0, _vp(y), _vp(y))
    with pytest.raises(PolmuxError, match="Samples x symbol should be either 1 or 2."):
        emu.call("plx_cmaadaptivefilter", _vp(z), _vp(z), 16, _vp(h), _vp(h), _vp(h), _vp(h), 3.0, 1e-3, _vp(R), 3.0, _vp(y), _vp(y))
    with pytest.raises(PolmuxError, match="Samples x symbol should be either 1 or 2."):
        emu.call("plx_easiadaptivefilter", _vp(z), _vp(z), 16, _vp(h), _vp(h), _vp(h), _vp(h), 1.0, 1e-3, 0.0, _vp(y), _vp(y))


@pytest.mark.parametrize("taps", [1, 3])
def test_emu_easiadaptivefilter_gateway(emu, oracle, taps):
    x = np.asfortranarray(_mixed_qpsk(150, 9))
    r = np.random.default_rng(2)
    h1 = np.asfortranarray(0.5 * (r.standard_normal((taps, 2)) + 1j * r.standard_normal((taps, 2))))
    h2 = np.asfortranarray(0.5 * (r.standard_normal((taps, 2)) + 1j * r.standard_normal((taps, 2))))
    xr, xi = np.asfortranarray(x.real.copy()), np.asfortranarray(x.imag.copy())
    hs = [np.asfortranarray(v.copy()) for v in (h1.real, h1.imag, h2.real, h2.imag)]
    dimY = 150 - taps + 1
    yr, yi = np.zeros((dimY, 2), order="F"), np.zeros((dimY, 2), order="F")
    emu.call("plx_easiadaptivefilter", _vp(xr), _vp(xi), 150, *[_vp(h) for h in hs], float(taps), 1e-2, 1.0, _vp(yr), _vp(yi))
    y, g1, g2 = oracle.easiadaptivefilter(x, h1, h2, taps, 1e-2, 1)
    np.testing.assert_allclose(yr + 1j * yi, y, atol=1e-12)
    np.testing.assert_allclose(hs[0] + 1j * hs[1], g1, atol=1e-12)
    np.testing.assert_allclose(hs[2] + 1j * hs[3], g2, atol=1e-12)


def test_emu_poldemux_batch(emu, oracle):
    """three frames, different convergence behaviour inside one wave"""
    L, taps, mu = 32, 7, 1 / 40        # tiny: every emulated wave shuffle costs two OS-thread barriers
    xs = [_mixed_qpsk(L, 20, noise=0.0, th=0.0), _mixed_qpsk(L, 20, noise=0.002, th=0.0), _mixed_qpsk(L, 20, noise=0.002, th=0.02)]
    xin = _il(np.stack([x.T for x in xs]))                     # [frame][2][L]
    y = np.zeros_like(xin)
    M = _il(np.tile(np.eye(2, dtype=complex).reshape(1, 4), (3, 1)))
    h = np.zeros(3 * 4 * taps * 2); passes = np.zeros(3, np.int32)
    R = np.array([1.0, 1.0])
    emu.call("plx_poldemux_dev", 1, _vp(xin), _vp(y), L, 3, taps, mu, _vp(R), _vp(M), _vp(h), _vp(passes), None)
    yy = y.view(np.complex128).reshape(3, 2, L)
    hh = h.view(np.complex128).reshape(3, 2, 2, taps)
    for f in range(3):
        oy, h1, h2, n = oracle.cmapolardemux(xs[f], np.eye(2), taps, mu, R)
        assert passes[f] == n
        np.testing.assert_allclose(yy[f].T, oy, atol=1e-11)
        np.testing.assert_allclose(hh[f, 0].T, h1, atol=1e-11)
        np.testing.assert_allclose(hh[f, 1].T, h2, atol=1e-11)
    assert len(set(passes.tolist())) > 1
    # EASI driver
    y2 = np.zeros_like(xin); p2 = np.zeros(3, np.int32)
    emu.call("plx_poldemux_dev", 2, _vp(xin), _vp(y2), L, 3, 1, mu, None, _vp(M), None, _vp(p2), None)
    for f in range(3):
        oy, h1, h2, n = oracle.easipolardemux(xs[f], np.eye(2), mu)
        assert p2[f] == n
        np.testing.assert_allclose(y2.view(np.complex128).reshape(3, 2, L)[f].T, oy, atol=1e-11)


def _dsp_params(**kw):
    p = DspParams()
    d = dict(workatbaudrate=0, applynlr=0, nlralpha=0.0, power_mw=2.0, applypol=0, polmethod=1, cma_mu=1 / 40,
             cma_taps=7, cma_txpolars=2, cma_phizero=0.0, easi_mu=1 / 40, easi_txpolars=2, easi_phizero=0.0,
             modorder=2, freqavg=5, phasavg=3, poworder=2)
    d.update(kw)
    for k, v in d.items():
        setattr(p, k, v)
    p.cma_R[0], p.cma_R[1] = 1.0, 1.0
    return p


@pytest.mark.parametrize("kw", [dict(), dict(applypol=1, polmethod=1), dict(applypol=1, polmethod=3, freqavg=0),
                                dict(applypol=1, polmethod=0, applynlr=1, nlralpha=0.05),
                                dict(applypol=1, polmethod=2, easi_txpolars=1, workatbaudrate=1, poworder=4, freqavg=40)])
def test_emu_dsp_chain(emu, oracle, kw):
    L = 32
    p = _dsp_params(**kw)
    Lin = L if p.workatbaudrate else 2 * L
    frames = 2
    r = np.random.default_rng(7)
    ins = []
    for f in range(frames):
        s = _mixed_qpsk(L, 30 + f, noise=0.002, th=0.02) * np.exp(1j * (2 * np.pi * 1 / L * np.arange(L) + 0.3))[:, None]
        x = np.zeros((Lin, 2), complex)
        x[:: (1 if p.workatbaudrate else 2)] = s * 4 * np.sqrt(2.0)
        if not p.workatbaudrate:
            x[1::2] = r.standard_normal((L, 2))
        ins.append(x)
    din = _il(np.stack([x.T for x in ins]))
    plan = C.c_void_p()
    emu.call("plx_dsp_create", C.byref(plan), Lin, 2, frames, C.byref(p))
    Lout = emu.lib.plx_dsp_out_len(plan)
    assert Lout == L
    dout = np.zeros(frames * 2 * L * 2)
    emu.call("plx_dsp_run_dev", plan, _vp(din), _vp(dout), frames, None)
    emu.call("plx_dsp_destroy", plan)
    out = dout.view(np.complex128).reshape(frames, 2, L)
    names = {0: "singlepol", 1: "cma", 2: "easi", 3: "combo"}
    op = oracle.dsp_params(power_mw=2.0, workatbaudrate=bool(p.workatbaudrate), applynlr=bool(p.applynlr), nlralpha=p.nlralpha,
                           applypol=bool(p.applypol), polmethod=names[p.polmethod], cma_mu=p.cma_mu, cma_taps=p.cma_taps,
                           cma_txpolars=p.cma_txpolars, easi_mu=p.easi_mu, easi_txpolars=p.easi_txpolars,
                           modorder=p.modorder, freqavg=p.freqavg, phasavg=p.phasavg, poworder=p.poworder)
    for f in range(frames):
        ref = oracle.dsp_pdm_coh_qpsk(ins[f], op)
        np.testing.assert_allclose(out[f].T, ref, atol=2e-10)
    # decisions + error count
    pat = oracle.samp2pat_coherent(np.angle(out[0].T))
    dp = np.ascontiguousarray(pat.T.copy())
    dp[0, :5] ^= 1
    err = np.zeros((frames, 2), np.int64); hat = np.zeros((frames, 4, L), np.uint8)
    emu.call("plx_decide_count_dev", _vp(dout), L, 2, frames, _vp(dp), _vp(hat), _vp(err), None)
    np.testing.assert_array_equal(hat[0], pat.T)
    assert err[0].tolist() == [5, 0]
    assert err[1].sum() == int(np.sum(oracle.samp2pat_coherent(np.angle(out[1].T)).T != dp))


def test_emu_evm(emu):
    r = np.random.default_rng(4)
    F, L = 3, 130
    sym = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (F, 2, L)))) + 0.1 * (r.standard_normal((F, 2, L)) + 1j * r.standard_normal((F, 2, L)))
    d = _il(sym.reshape(F, -1))
    out = np.zeros(F)
    emu.call("plx_evm_dev", _vp(d), L, 2, F, _vp(out), None)
    hat = (np.where(sym.real >= 0, 1, -1) + 1j * np.where(sym.imag > 0, 1, -1)) / np.sqrt(2)
    np.testing.assert_allclose(out, (np.abs(sym - hat) ** 2).mean(axis=(1, 2)), rtol=1e-13)


def test_emu_sweep_variants(emu, oracle, tune):
    """The fused column sweep (default at this geometry: k_colx16 with its per-frame barrier), the barrier-free
    three-sweep step (PLX_SSFM_NO_FUSE) and a 16 x 256 split (PLX_SSFM_P1=4: 256-point rows through the general
    k_row) give the same fields and step counts."""
    n, nt, L = 4096, 64, 1.5e3
    fls = [1, 0, 1, 0]
    betat, db1 = _tables(n, nt, fls, 1)
    fields = [_qpsk_field(n, nt, p)[:2] for p in (6.0, 9.0, 12.0)]
    ref = [oracle.matrix_ssfm(f[0], f[1], betat, db1, 4e2, 5e-3, [1.3e-6], 4.6e-5, L, 1, 0, fls, [0.0], [0.0], [0.0]) for f in fields]
    # (PLX_EMU_CUS=1: a one-CU device -- the fused grid is two workgroups, so each walks the tiles of all three frames, with the
    # emulator's two-entry window of the frame list moving on under it)
    # (the three-sweep step alone is test_emu_register_form_row_pass's and test_emu_fused_sweep_scalar_plan's second leg)
    for env in ({}, {"PLX_EMU_CUS": "1"}, {"PLX_SSFM_P1": "4"}):
        nf = 1 if ("PLX_SSFM_P1" in env or "PLX_SSFM_NO_FUSE" in env) else 3
        for k, v in env.items():
            tune.setenv(k, v)
        d = _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-6], 4e2, 5e-3, betat, db1, frames=nf)
        plan = C.c_void_p()
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
        for k in env:
            tune.delenv(k)
        ux = _il(np.stack([f[0] for f in fields[:nf]])); uy = _il(np.stack([f[1] for f in fields[:nf]]))
        emu.call("plx_ssfm_propagate_dev", plan, _vp(ux), _vp(uy), nf, None)
        ncyc = np.zeros(nf, np.int32)
        emu.call("plx_ssfm_results", plan, nf, None, _vp(ncyc))
        emu.call("plx_ssfm_destroy", plan)
        gx = ux.view(np.complex128).reshape(nf, n)
        gy = uy.view(np.complex128).reshape(nf, n)
        for f in range(nf):
            rc, ofd, onc, ox, oy = ref[f]
            assert ncyc[f] == onc
            assert np.abs(gx[f] - ox[:, 0]).max() < 1e-11 * np.abs(ox).max()
            assert np.abs(gy[f] - oy[:, 0]).max() < 1e-11 * np.abs(oy).max()


def test_emu_register_form_row_pass(emu, oracle, tune):
    """256 x 256 frames without PMD take the register form of the row pass (k_row256r: one wave per 2 rows x 2 polarisations,
    the multiplier shared between the wave's halves); PLX_SSFM_ROWR=0 keeps the LDS-resident k_row.  Both against the oracle,
    and against each other (they differ by the rounding of the inter-pass twiddles only)."""
    n, nt, L = 65536, 16, 7e2
    fls = [1, 0, 1, 0]
    betat, db1 = _tables(n, nt, fls, 1)
    tune.setenv("PLX_SSFM_NO_FUSE", "1")    # (three-sweep step: the emulated frame barrier of the fused sweep is slow and not under test here)
    fields = [_qpsk_field(n, nt, p)[:2] for p in (9.0,)]
    ref = [oracle.matrix_ssfm(f[0], f[1], betat, db1, 4e2, 5e-3, [1.3e-6], 4.6e-5, L, 1, 0, fls, [0.0], [0.0], [0.0]) for f in fields]
    got = {}
    for mode in ("1",):      # ("0", the same frame through k_row, is compared on the GPU: test_sentinel_landing_... 'ldsrow', test_register_form_row_pass_with_pmd_vs_oracle_three_ways)
        tune.setenv("PLX_SSFM_ROWR", mode)
        d = _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-6], 4e2, 5e-3, betat, db1, frames=1)
        plan = C.c_void_p()
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
        tune.delenv("PLX_SSFM_ROWR")
        info = (C.c_int32 * 8)()
        emu.call("plx_ssfm_info", plan, info)
        assert info[6] == 64                                     # one wave per 2 rows x 2 polarisations: k_row256r
        ux = _il(np.stack([f[0] for f in fields])); uy = _il(np.stack([f[1] for f in fields]))
        emu.call("plx_ssfm_propagate_dev", plan, _vp(ux), _vp(uy), 1, None)
        ncyc = np.zeros(1, np.int32)
        emu.call("plx_ssfm_results", plan, 1, None, _vp(ncyc))
        emu.call("plx_ssfm_destroy", plan)
        gx = ux.view(np.complex128).reshape(1, n); gy = uy.view(np.complex128).reshape(1, n)
        for f in range(1):
            rc, ofd, onc, ox, oy = ref[f]
            assert ncyc[f] == onc and onc >= 2
            assert np.abs(gx[f] - ox[:, 0]).max() < 1e-11 * np.abs(ox).max()
            assert np.abs(gy[f] - oy[:, 0]).max() < 1e-11 * np.abs(oy).max()
        got[mode] = (gx.copy(), gy.copy())


@pytest.mark.parametrize("tables", [True])      # (False = one exponential per bin and trunk: on the GPU, test_register_form_row_pass_with_pmd_vs_oracle_three_ways)
def test_emu_register_form_row_pass_with_pmd(emu, oracle, tune, tables):
    """k_row256r<PMD>: the wave's halves trade so that a lane holds both polarisations of eight bins, then the waveplate trunks
    of matrix_step (fiber.m:907-933) with the phasor tables of k_pmd_tab or one exponential per bin and trunk
    (PLX_SSFM_NO_PMD_TAB=1).  256 x 256 frame with its own waveplates, against the oracle and against k_row's PMD branch."""
    if not tables:
        tune.setenv("PLX_SSFM_NO_PMD_TAB", "1")
    tune.setenv("PLX_SSFM_NO_FUSE", "1")    # (three-sweep step, as in test_emu_register_form_row_pass)
    n, nt, nplates, L = 65536, 16, 5, 8e2
    fls = [1, 1, 1, 0]
    betat, db1 = _tables(n, nt, fls, nplates)
    r = np.random.default_rng(11)
    brf = (r.random(nplates) * 2 * np.pi - np.pi, r.random(nplates) * np.pi - np.pi / 2, 0.5 * np.arcsin(r.random(nplates) * 2 - 1))
    fx, fy = _qpsk_field(n, nt, 7.0)[:2]
    rc, ofd, onc, ox, oy = oracle.matrix_ssfm(fx, fy, betat, db1, 5e2, 5e-3, [1.3e-6], 4.6e-5, L, nplates, 0, fls, *brf)
    assert rc == 0 and onc >= 2
    got = {}
    for mode in ("1",):      # ("0", k_row's PMD branch on the same frame: on the GPU, test_register_form_row_pass_with_pmd_vs_oracle_three_ways)
        tune.setenv("PLX_SSFM_ROWR", mode)
        d = _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-6], 5e2, 5e-3, betat, db1, nplates=nplates, frames=1)
        plan = C.c_void_p()
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
        tune.delenv("PLX_SSFM_ROWR")
        emu.call("plx_ssfm_set_birefringence", plan, _vp(brf[0]), _vp(brf[1]), _vp(brf[2]), 1)
        ux = _il(fx[None]); uy = _il(fy[None])
        emu.call("plx_ssfm_propagate_dev", plan, _vp(ux), _vp(uy), 1, None)
        ncyc = np.zeros(1, np.int32)
        emu.call("plx_ssfm_results", plan, 1, None, _vp(ncyc))
        emu.call("plx_ssfm_destroy", plan)
        gx = ux.view(np.complex128).reshape(n); gy = uy.view(np.complex128).reshape(n)
        assert ncyc[0] == onc
        assert np.abs(gx - ox[:, 0]).max() < 1e-11 * np.abs(ox).max()
        assert np.abs(gy - oy[:, 0]).max() < 1e-11 * np.abs(oy).max()
        got[mode] = (gx.copy(), gy.copy())


def test_emu_fused_sweep_exact_nonlinear_step(emu, oracle, tune):
    """The fused sweep's rare path on a one-CU device (one team walks both frames): the '--s-' exact single step
    (fiber.m:172-174: nonlinear phases of radians, so the Kerr step takes the full-range sincos through the exchange buffer)
    and a stale frame list (a batch under 64 frames rebuilds it once per chunk: the second launch meets finished frames) --
    against the oracle, and against the three-sweep step."""
    n, nt, L = 4096, 64, 2e4
    fls = [0, 0, 1, 0]
    betat, db1 = _tables(n, nt, fls, 1)
    fields = [_qpsk_field(n, nt, p)[:2] for p in (30.0, 60.0)]
    ref = [oracle.matrix_ssfm(f[0], f[1], betat, db1, L, np.inf, [1.3e-3], 4.6e-5, L, 1, 0, fls, [0.0], [0.0], [0.0]) for f in fields]
    got = []
    for env in ({"PLX_EMU_CUS": "1"}, {"PLX_SSFM_NO_FUSE": "1"}):
        for k, v in env.items():
            tune.setenv(k, v)
        d = _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-3], L, np.inf, betat, db1, frames=2)
        plan = C.c_void_p()
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
        for k in env:
            tune.delenv(k)
        ux = _il(np.stack([f[0] for f in fields])); uy = _il(np.stack([f[1] for f in fields]))
        emu.call("plx_ssfm_propagate_dev", plan, _vp(ux), _vp(uy), 2, None)
        ncyc = np.zeros(2, np.int32)
        emu.call("plx_ssfm_results", plan, 2, None, _vp(ncyc))
        emu.call("plx_ssfm_destroy", plan)
        got.append((ux.copy(), uy.copy(), ncyc.copy()))
    assert np.abs(got[0][0] - got[1][0]).max() < 1e-11 * np.abs(got[1][0]).max()
    gx, gy = got[0][0].view(np.complex128).reshape(2, n), got[0][1].view(np.complex128).reshape(2, n)
    for f in range(2):
        rc, ofd, onc, ox, oy = ref[f]
        assert got[0][2][f] == onc == 1
        assert np.abs(ox).max() > 0 and np.abs(np.angle(gx[f] * np.conj(fields[f][0]))).max() > 0.5      # radians of nonlinear phase
        assert np.abs(gx[f] - ox[:, 0]).max() < 1e-11 * np.abs(ox).max()
        assert np.abs(gy[f] - oy[:, 0]).max() < 1e-11 * np.abs(oy).max()


@pytest.mark.parametrize("flag", ["g-s-", "--s-"])
def test_emu_fused_sweep_scalar_plan(emu, oracle, tune, flag):
    """k_colx16<false>: scalar_ssfm (fiber.m:557-636) without XPM on the fused column sweep -- sixteen columns of the one field
    to a tile, max |u|^2 at the frame barrier, nl_step on the lane's own points; two 'sepfields' channels, three frames at
    different powers claimed by the teams of a one-CU device.  'g-s-': against oracle.scalar_ssfm and the three-sweep step;
    '--s-': the exact single step (radians of phase: the full-range path through the exchange buffer)."""
    n, nt, nfc = 4096, 64, 2
    fls = [1, 0, 1, 0] if flag == "g-s-" else [0, 0, 1, 0]
    L = 1.2e3 if flag == "g-s-" else 2e4
    gam = [1.3e-6, 1.2e-6] if flag == "g-s-" else [1.3e-3, 1.1e-3]
    dzmax, dph = (4e2, 5e-3) if flag == "g-s-" else (L, np.inf)
    betat, db1 = _tables(n, nt, fls, 1, nfc)
    frames = [np.asfortranarray(np.stack([_qpsk_field(n, nt, p * (1 + 0.3 * k), (2 + k, 5 + k))[0] for k in range(nfc)], 1)) for p in (6.0, 9.0, 14.0)]
    ref = [oracle.scalar_ssfm(u, betat, dzmax, dph, gam, 4.6e-5, L, fls) for u in frames]
    got = []
    for env in ({"PLX_EMU_CUS": "1"}, {"PLX_SSFM_NO_FUSE": "1"}):
        for k, v in env.items():
            tune.setenv(k, v)
        d = _desc(n, nfc, 0, fls, L, 4.6e-5, gam, dzmax, dph, betat, db1, frames=3)
        plan = C.c_void_p()
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
        for k in env:
            tune.delenv(k)
        info = (C.c_int32 * 8)()
        emu.call("plx_ssfm_info", plan, info)
        assert info[0] == (0 if "PLX_SSFM_NO_FUSE" in env else 1) and info[1] == 8 and info[4] == (0 if "PLX_SSFM_NO_FUSE" in env else nfc)
        g = _il(np.stack([np.ascontiguousarray(u.T) for u in frames]))          # [frame][channel][nfft]
        emu.call("plx_ssfm_propagate_dev", plan, _vp(g), None, 3, None)
        nc = np.zeros(3, np.int32)
        emu.call("plx_ssfm_results", plan, 3, None, _vp(nc))
        emu.call("plx_ssfm_destroy", plan)
        got.append((g.view(np.complex128).reshape(3, nfc, n).copy(), nc.copy()))
    for f in range(3):
        ofd, onc, ou = ref[f]
        for gg, nc in got:
            assert nc[f] == onc and (onc >= 3 if flag == "g-s-" else onc == 1)
            assert np.abs(gg[f].T - ou).max() < 1e-11 * np.abs(ou).max()
    if flag == "--s-":
        assert np.abs(np.angle(got[0][0][2][0] * np.conj(frames[2][:, 0]))).max() > 0.5      # radians of nonlinear phase


def test_emu_long_rows_compact_twiddles(emu, oracle, tune):
    """4096-point rows (the row pass of 2^20-sample frames: one polarisation per workgroup, compact twiddle table W^{4k} +
    four fine factors, plx_fft.h row_tw) on a 4 x 4096 split of a 2^14 frame: field and step count against the oracle."""
    n, nt, L = 16384, 64, 6e2                       # 4 x 4096 split, two steps
    fls = [1, 0, 1, 0]
    betat, db1 = _tables(n, nt, fls, 1)
    f = _qpsk_field(n, nt, 6.0)
    tune.setenv("PLX_SSFM_P1", "2")
    tune.setenv("PLX_SSFM_COL_THREADS", "128")   # (fewer, wider column workgroups: fewer emulated threads)
    tune.setenv("PLX_SSFM_LOGW", "6")
    d = _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-6], 4e2, 5e-3, betat, db1, frames=1)
    plan = C.c_void_p()
    emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
    tune.delenv("PLX_SSFM_P1")
    tune.delenv("PLX_SSFM_COL_THREADS")
    tune.delenv("PLX_SSFM_LOGW")
    info = (C.c_int32 * 8)()
    emu.call("plx_ssfm_info", plan, info)
    assert list(info)[:3] == [0, 2, 12] and info[7] == 1          # three sweeps, 4 x 4096, one polarisation per row workgroup
    ux = _il(f[0][None]); uy = _il(f[1][None])
    emu.call("plx_ssfm_propagate_dev", plan, _vp(ux), _vp(uy), 1, None)
    nc = np.zeros(1, np.int32)
    emu.call("plx_ssfm_results", plan, 1, None, _vp(nc))
    emu.call("plx_ssfm_destroy", plan)
    rc, fd, onc, ox, oy = oracle.matrix_ssfm(f[0], f[1], betat, db1, 4e2, 5e-3, [1.3e-6], 4.6e-5, L, 1, 0, fls, [0.0], [0.0], [0.0])
    assert nc[0] == onc
    assert np.abs(ux.view(np.complex128).reshape(n) - ox[:, 0]).max() < 1e-11 * np.abs(ox).max()
    assert np.abs(uy.view(np.complex128).reshape(n) - oy[:, 0]).max() < 1e-11 * np.abs(oy).max()


@pytest.mark.parametrize("tables", [True, False])
def test_emu_long_rows_both_polarisations_pmd(emu, oracle, tune, tables):
    """k_row4k<true>: 4096-point rows of a PMD plan -- both polarisations of a row in one 512-thread workgroup, lanes i and
    i + 32 of a wave trade halves around the waveplate trunks of matrix_step (fiber.m:907-933), phasor tables of k_pmd_tab or
    one exponential per bin and trunk (PLX_SSFM_NO_PMD_TAB=1).  4 x 4096 split of a 2^14 frame with its own waveplates against
    the oracle, and against the 2048-point rows of k_row's PMD branch (PLX_SSFM_SHORT_ROWS=1)."""
    if not tables:
        tune.setenv("PLX_SSFM_NO_PMD_TAB", "1")
    n, nt, nplates, L = 16384, 64, 4, 9e2
    fls = [1, 1, 1, 0]
    betat, db1 = _tables(n, nt, fls, nplates)
    r = np.random.default_rng(23)
    brf = (r.random(nplates) * 2 * np.pi - np.pi, r.random(nplates) * np.pi - np.pi / 2, 0.5 * np.arcsin(r.random(nplates) * 2 - 1))
    fx, fy = _qpsk_field(n, nt, 6.0)[:2]
    rc, ofd, onc, ox, oy = oracle.matrix_ssfm(fx, fy, betat, db1, 4e2, 5e-3, [1.3e-6], 4.6e-5, L, nplates, 0, fls, *brf)
    assert rc == 0 and onc >= 3
    got = {}
    for short in ((False, True) if tables else (False,)):      # (the 8 x 2048 comparison once: it doubles the emulated work)
        tune.setenv("PLX_SSFM_P1", "3" if short else "2")
        tune.setenv("PLX_SSFM_COL_THREADS", "128")
        tune.setenv("PLX_SSFM_LOGW", "6")
        d = _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-6], 4e2, 5e-3, betat, db1, nplates=nplates, frames=1)
        plan = C.c_void_p()
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
        for k in ("PLX_SSFM_P1", "PLX_SSFM_COL_THREADS", "PLX_SSFM_LOGW"):
            tune.delenv(k)
        info = (C.c_int32 * 8)()
        emu.call("plx_ssfm_info", plan, info)
        assert list(info)[:3] == ([0, 3, 11] if short else [0, 2, 12])
        assert short or (info[6] == 512 and info[7] == 0)      # both rows in one workgroup of k_row4k
        emu.call("plx_ssfm_set_birefringence", plan, _vp(brf[0]), _vp(brf[1]), _vp(brf[2]), 1)
        ux = _il(fx[None]); uy = _il(fy[None])
        emu.call("plx_ssfm_propagate_dev", plan, _vp(ux), _vp(uy), 1, None)
        ncyc = np.zeros(1, np.int32)
        emu.call("plx_ssfm_results", plan, 1, None, _vp(ncyc))
        emu.call("plx_ssfm_destroy", plan)
        gx = ux.view(np.complex128).reshape(n); gy = uy.view(np.complex128).reshape(n)
        assert ncyc[0] == onc
        assert np.abs(gx - ox[:, 0]).max() < 1e-11 * np.abs(ox).max()
        assert np.abs(gy - oy[:, 0]).max() < 1e-11 * np.abs(oy).max()
        got[short] = (gx.copy(), gy.copy())
    if tables:
        assert np.abs(got[True][0] - got[False][0]).max() < 1e-12 * np.abs(got[False][0]).max()


@pytest.mark.parametrize("nsymb,nt", [(256, 64), (64, 64), (256, 8)])
def test_emu_inverse_pmd_long_rows(emu, oracle, tune, nsymb, nt):
    """inverse_pmd's matrix tables (inverse_pmd.m:130-141) through the register-form row passes: the plan of plx_pmdinv is a
    PMD-type plan, so k_row4k<true> (4 x 4096 split) / k_rowreg<., true> (4 x 1024, 4 x 512) apply (Hgvd U)^H bin by bin after
    the halves' trade.  Against oracle/pmdinv.py."""
    from oracle import pmdinv
    n = nsymb * nt
    fn = synth.fn_grid(nsymb, nt)
    omega = 2 * np.pi * 10.0 * fn
    betat = 0.5 * omega ** 2 * -2.17e-8 + omega ** 3 * 1.3e-10 / 6
    nplates, L, dgd = 5, 4e4, 0.4
    r = np.random.default_rng(5)
    sets = (r.random(nplates) * 2 * np.pi - np.pi, r.random(nplates) * np.pi - np.pi / 2, 0.5 * np.arcsin(r.random(nplates) * 2 - 1))
    db1 = dgd / nplates / 10.0 * omega
    ux, uy, _, _ = synth.pdm_qpsk_field(nsymb, nt, 2.0)
    fx = np.stack([ux]); fy = np.stack([1j * np.roll(uy, 7)])
    tune.setenv("PLX_SSFM_P1", "2")
    tune.setenv("PLX_SSFM_COL_THREADS", "128")
    tune.setenv("PLX_SSFM_LOGW", "6")
    plan = C.c_void_p()
    emu.call("plx_pmdinv_create", C.byref(plan), n, 1)
    for k in ("PLX_SSFM_P1", "PLX_SSFM_COL_THREADS", "PLX_SSFM_LOGW"):
        tune.delenv(k)
    ntr = np.array([nplates], dtype=np.int32)
    db0, th, ep = (np.ascontiguousarray(v[None]) for v in sets)
    lc = np.array([L / nplates])
    emu.call("plx_pmdinv_set_link", plan, 1, _vp(ntr), _vp(db0), _vp(th), _vp(ep), _vp(lc), _vp(np.ascontiguousarray(betat[None])),
             _vp(np.ascontiguousarray(db1[None])), None, 1, 1)
    gx, gy = _il(fx), _il(fy)
    emu.call("plx_pmdinv_apply_dev", plan, _vp(gx), _vp(gy), 1, None)
    emu.call("plx_pmdinv_destroy", plan)
    gx = gx.view(np.complex128).reshape(n); gy = gy.view(np.complex128).reshape(n)
    brf = [dict(db0=sets[0], theta=sets[1], epsilon=sets[2], lcorr=L / nplates, betat=betat, db1=db1)]
    Uinv, U, wx, wy = pmdinv.inverse_pmd(brf, fx[0], fy[0], None)
    assert np.abs(gx - wx).max() < 1e-12 * np.abs(wx).max()
    assert np.abs(gy - wy).max() < 1e-12 * np.abs(wy).max()


@pytest.mark.parametrize("logm", [9, 10, 11])
def test_emu_register_form_rows_512_to_2048(emu, oracle, tune, logm):
    """k_rowreg<9 / 10 / 11>: rows of 512, 1024 and 2048 points (frames of 2^17 ... 2^19 samples on the 256-row split; 2^18 is the
    size Run_my_PDM_QPSK.m:21-24 ships with) as three register levels -- radix 16, radix 2 / 4 / 8 at stride 16, radix 16 -- on a
    4 x M split, against the oracle and against the LDS-resident k_row (PLX_SSFM_ROWR=0)."""
    M = 1 << logm
    n, nt, L = 4 * M, (16 if logm == 10 else 32), 9e2
    fls = [1, 0, 1, 0]
    betat, db1 = _tables(n, nt, fls, 1)
    f = _qpsk_field(n, nt, 6.0)
    rc, fd, onc, ox, oy = oracle.matrix_ssfm(f[0], f[1], betat, db1, 4e2, 5e-3, [1.3e-6], 4.6e-5, L, 1, 0, fls, [0.0], [0.0], [0.0])
    assert rc == 0 and onc >= 3
    got = {}
    for mode in ("1", "0"):
        tune.setenv("PLX_SSFM_ROWR", mode)
        tune.setenv("PLX_SSFM_P1", "2")
        tune.setenv("PLX_SSFM_COL_THREADS", "128")
        tune.setenv("PLX_SSFM_LOGW", "6")
        nf = 2 if logm == 9 else 1                             # (a two-frame batch where the emulated work is small)
        d = _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-6], 4e2, 5e-3, betat, db1, frames=nf)
        plan = C.c_void_p()
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
        for k in ("PLX_SSFM_ROWR", "PLX_SSFM_P1", "PLX_SSFM_COL_THREADS", "PLX_SSFM_LOGW"):
            tune.delenv(k)
        info = (C.c_int32 * 8)()
        emu.call("plx_ssfm_info", plan, info)
        assert list(info)[:3] == [0, 2, logm] and (info[7] == 2) == (mode == "1")
        ux = _il(np.stack([f[0], 0.5 * f[1]])[:nf]); uy = _il(np.stack([f[1], f[0]])[:nf])
        emu.call("plx_ssfm_propagate_dev", plan, _vp(ux), _vp(uy), nf, None)
        nc = np.zeros(nf, np.int32)
        emu.call("plx_ssfm_results", plan, nf, None, _vp(nc))
        gx = ux.view(np.complex128).reshape(nf, n); gy = uy.view(np.complex128).reshape(nf, n)
        assert nc[0] == onc
        assert np.abs(gx[0] - ox[:, 0]).max() < 1e-11 * np.abs(ox).max()
        assert np.abs(gy[0] - oy[:, 0]).max() < 1e-11 * np.abs(oy).max()
        emu.call("plx_ssfm_destroy", plan)
        got[mode] = (gx.copy(), gy.copy())
    assert np.abs(got["1"][0] - got["0"][0]).max() < 1e-12 * np.abs(got["0"][0]).max()
    assert not np.array_equal(got["1"][0], got["0"][0])      # (the switch really selects another kernel)


@pytest.mark.parametrize("logm,scalar", [(5, False), (6, False), (7, False), (5, True), (7, True)])
def test_emu_register_form_small_rows(emu, oracle, tune, logm, scalar):
    """k_rowsm<5 / 6 / 7>: rows of 32, 64 and 128 points (frames of 2^13 ... 2^15 samples on the 256-row split: the sizes of the
    reference's own examples) -- R = 2 / 4 / 8 threads per row and polarisation, sixteen points each, one radix-R set per
    i = j + R par, one exchange in real / imaginary halves, r16 -- on a 32 x M split: a dual-polarisation batch of two frames
    against oracle.matrix_ssfm, a two-channel scalar XPM comb against oracle.scalar_ssfm, and both against k_row (PLX_SSFM_ROWR=0)."""
    M = 1 << logm
    n, nt, L = 32 * M, (16 if logm != 6 else 32), 9e2
    got = {}
    if scalar:
        fls = [1, 0, 1, 1]
        betat, db1 = _tables(n, nt, fls, 1, nfc=2)
        u = np.asfortranarray(np.stack([_qpsk_field(n, nt, p)[0] for p in (6.0, 8.0)], 1))
        gam = [1.3e-6, 1.25e-6]
        ofd, onc, ou = oracle.scalar_ssfm(u, betat, 4e2, 5e-3, gam, 4.6e-5, L, fls)
    else:
        fls = [1, 0, 1, 0]
        betat, db1 = _tables(n, nt, fls, 1)
        f = _qpsk_field(n, nt, 6.0)
        rc, fd, onc, ox, oy = oracle.matrix_ssfm(f[0], f[1], betat, db1, 4e2, 5e-3, [1.3e-6], 4.6e-5, L, 1, 0, fls, [0.0], [0.0], [0.0])
    assert onc >= 3
    tune.setenv("PLX_SSFM_ROWSM", "2")              # (also for the short dual rows, which take k_row by default)
    for mode in ("1", "0"):
        tune.setenv("PLX_SSFM_ROWR", mode)
        tune.setenv("PLX_SSFM_P1", "5")
        tune.setenv("PLX_SSFM_COL_THREADS", "128")
        d = _desc(n, 2, 0, fls, L, 4.6e-5, gam, 4e2, 5e-3, betat, db1, frames=1) if scalar else \
            _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-6], 4e2, 5e-3, betat, db1, frames=2)
        plan = C.c_void_p()
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
        for k in ("PLX_SSFM_ROWR", "PLX_SSFM_P1", "PLX_SSFM_COL_THREADS"):
            tune.delenv(k)
        info = (C.c_int32 * 8)()
        emu.call("plx_ssfm_info", plan, info)
        assert list(info)[:3] == [0, 5, logm] and (info[7] == 2 and info[6] == 64) == (mode == "1")
        if scalar:
            g = _il(np.ascontiguousarray(u.T)[None])
            emu.call("plx_ssfm_propagate_dev", plan, _vp(g), None, 1, None)
            nc = np.zeros(1, np.int32)
            emu.call("plx_ssfm_results", plan, 1, None, _vp(nc))
            res = g.view(np.complex128).reshape(2, n).T
            assert nc[0] == onc and np.abs(res - ou).max() < 1e-11 * np.abs(ou).max()
        else:
            ux = _il(np.stack([f[0], 0.5 * f[1]])); uy = _il(np.stack([f[1], f[0]]))
            emu.call("plx_ssfm_propagate_dev", plan, _vp(ux), _vp(uy), 2, None)
            nc = np.zeros(2, np.int32)
            emu.call("plx_ssfm_results", plan, 2, None, _vp(nc))
            res = ux.view(np.complex128).reshape(2, n)
            gy = uy.view(np.complex128).reshape(2, n)
            assert nc[0] == onc
            assert np.abs(res[0] - ox[:, 0]).max() < 1e-11 * np.abs(ox).max()
            assert np.abs(gy[0] - oy[:, 0]).max() < 1e-11 * np.abs(oy).max()
        emu.call("plx_ssfm_destroy", plan)
        got[mode] = np.array(res).copy()
    assert np.abs(got["1"] - got["0"]).max() < 1e-12 * np.abs(got["0"]).max()
    assert not np.array_equal(got["1"], got["0"])      # (the switch really selects another kernel)


@pytest.mark.parametrize("logm", [9, 11, 12])
def test_emu_register_form_rows_scalar_plan(emu, oracle, tune, logm):
    """k_rowreg<., false, true>: the scalar plan's rows of 512 / 2048 points (every row-polarisation of the workgroup is a row of
    the one field), two 'sepfields' channels with XPM, 4 x M split (8 x M for 512 points: eight rows to a workgroup) against
    oracle.scalar_ssfm.  4096 points: k_row4k<false> with one workgroup per row and frame-channel."""
    M = 1 << logm
    p1 = 3 if logm == 9 else 2
    n, nt, L = (1 << p1) * M, (32 if logm == 11 else 64), 9e2
    nfc = 1 if logm == 12 else 2                         # (the 4096-point rows on one field: half the emulated work)
    fls = [1, 0, 1, 1 if nfc > 1 else 0]
    betat, db1 = _tables(n, nt, fls, 1, nfc=nfc)
    cols = [_qpsk_field(n, nt, p)[0] for p in (6.0, 8.0)[:nfc]]
    u = np.asfortranarray(np.stack(cols, 1))
    gam = [1.3e-6, 1.25e-6][:nfc]
    ofd, onc, ou = oracle.scalar_ssfm(u, betat, 4e2, 5e-3, gam, 4.6e-5, L, fls)
    assert onc >= 3
    tune.setenv("PLX_SSFM_P1", str(p1))
    tune.setenv("PLX_SSFM_COL_THREADS", "128")
    tune.setenv("PLX_SSFM_LOGW", "6")
    d = _desc(n, nfc, 0, fls, L, 4.6e-5, gam, 4e2, 5e-3, betat, db1, frames=1)
    plan = C.c_void_p()
    emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
    for k in ("PLX_SSFM_P1", "PLX_SSFM_COL_THREADS", "PLX_SSFM_LOGW"):
        tune.delenv(k)
    info = (C.c_int32 * 8)()
    emu.call("plx_ssfm_info", plan, info)
    assert list(info)[:3] == [0, p1, logm] and info[7] == (2 if logm < 12 else 1)
    g = _il(np.ascontiguousarray(u.T)[None])                     # [frame][channel][nfft]
    emu.call("plx_ssfm_propagate_dev", plan, _vp(g), None, 1, None)
    nc = np.zeros(1, np.int32)
    emu.call("plx_ssfm_results", plan, 1, None, _vp(nc))
    emu.call("plx_ssfm_destroy", plan)
    assert nc[0] == onc
    got = g.view(np.complex128).reshape(nfc, n).T
    assert np.abs(got - ou).max() < 1e-11 * np.abs(ou).max()


@pytest.mark.parametrize("logm,tables", [(9, True), (10, True), (10, False), (11, True)])
def test_emu_register_form_rows_both_polarisations_pmd(emu, oracle, tune, logm, tables):
    """k_rowreg<., true>: rows of 512 / 1024 / 2048 points of a PMD plan -- lanes i and i + 32 of every wave hold the same thread
    of the X and the Y row and trade halves around the waveplate trunks (pair_multiplier, shared with k_row4k<true>).  4 x M
    split with its own waveplates against the oracle and against k_row's PMD branch (PLX_SSFM_ROWR=0)."""
    if not tables:
        tune.setenv("PLX_SSFM_NO_PMD_TAB", "1")
    M = 1 << logm
    n, nt, nplates, L = 4 * M, (16 if logm == 10 else 32), 4, 9e2
    fls = [1, 1, 1, 0]
    betat, db1 = _tables(n, nt, fls, nplates)
    r = np.random.default_rng(29)
    brf = (r.random(nplates) * 2 * np.pi - np.pi, r.random(nplates) * np.pi - np.pi / 2, 0.5 * np.arcsin(r.random(nplates) * 2 - 1))
    fx, fy = _qpsk_field(n, nt, 6.0)[:2]
    rc, ofd, onc, ox, oy = oracle.matrix_ssfm(fx, fy, betat, db1, 4e2, 5e-3, [1.3e-6], 4.6e-5, L, nplates, 0, fls, *brf)
    assert rc == 0 and onc >= 3
    got = {}
    for mode in ("1", "0"):
        tune.setenv("PLX_SSFM_ROWR", mode)
        tune.setenv("PLX_SSFM_P1", "2")
        tune.setenv("PLX_SSFM_COL_THREADS", "128")
        tune.setenv("PLX_SSFM_LOGW", "6")
        d = _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-6], 4e2, 5e-3, betat, db1, nplates=nplates, frames=1)
        plan = C.c_void_p()
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
        for k in ("PLX_SSFM_ROWR", "PLX_SSFM_P1", "PLX_SSFM_COL_THREADS", "PLX_SSFM_LOGW"):
            tune.delenv(k)
        info = (C.c_int32 * 8)()
        emu.call("plx_ssfm_info", plan, info)
        assert list(info)[:3] == [0, 2, logm] and (info[7] == 2) == (mode == "1")
        emu.call("plx_ssfm_set_birefringence", plan, _vp(brf[0]), _vp(brf[1]), _vp(brf[2]), 1)
        ux = _il(fx[None]); uy = _il(fy[None])
        emu.call("plx_ssfm_propagate_dev", plan, _vp(ux), _vp(uy), 1, None)
        ncyc = np.zeros(1, np.int32)
        emu.call("plx_ssfm_results", plan, 1, None, _vp(ncyc))
        emu.call("plx_ssfm_destroy", plan)
        gx = ux.view(np.complex128).reshape(n); gy = uy.view(np.complex128).reshape(n)
        assert ncyc[0] == onc
        assert np.abs(gx - ox[:, 0]).max() < 1e-11 * np.abs(ox).max()
        assert np.abs(gy - oy[:, 0]).max() < 1e-11 * np.abs(oy).max()
        got[mode] = (gx.copy(), gy.copy())
    assert np.abs(got["1"][1] - got["0"][1]).max() < 1e-12 * np.abs(got["0"][1]).max()
    assert not np.array_equal(got["1"][0], got["0"][0])


def test_emu_frame_barrier_timeout_aborts_cleanly(emu, tune):
    """A fused-sweep frame whose workgroups are not co-resident (here: the emulator runs ONE workgroup at a time) must
    end in a clean error, never a hang: the barrier times out (wall clock), raises the sticky abort word, nothing is
    stored or advanced after it, and propagate reports PLX_ERR_TIMEOUT.  The timeout hits in the first launch, before any
    store: the caller's field is bit-for-bit untouched."""
    n, nt, L = 4096, 64, 1.5e3
    fls = [1, 0, 1, 0]
    betat, db1 = _tables(n, nt, fls, 1)
    f = _qpsk_field(n, nt, 6.0)
    tune.setenv("PLX_SSFM_BARRIER_TIMEOUT_MS", "30")
    d = _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-6], 4e2, 5e-3, betat, db1, frames=2)
    plan = C.c_void_p()
    emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
    tune.delenv("PLX_SSFM_BARRIER_TIMEOUT_MS")
    ux = _il(np.stack([f[0], f[0]])); uy = _il(np.stack([f[1], f[1]]))
    ux0, uy0 = ux.copy(), uy.copy()
    tune.setenv("PLX_EMU_STARVE", "1")
    rc = emu.lib.plx_ssfm_propagate_dev(plan, _vp(ux), _vp(uy), 2, None)
    tune.delenv("PLX_EMU_STARVE")
    assert rc == -5                                     # PLX_ERR_TIMEOUT
    assert b"frame barrier timed out" in emu.lib.plx_last_error()
    np.testing.assert_array_equal(ux, ux0)
    np.testing.assert_array_equal(uy, uy0)
    # the plan has switched itself to the barrier-free three-sweep step: the same call again (still one workgroup at a
    # time) now succeeds -- what the gateway tier does at once from its staging copy (fiber.m:372-389 always returns a field)
    info = (C.c_int32 * 8)()
    emu.call("plx_ssfm_info", plan, info)
    assert info[0] == 0
    tune.setenv("PLX_EMU_STARVE", "1")
    rc = emu.lib.plx_ssfm_propagate_dev(plan, _vp(ux), _vp(uy), 2, None)
    tune.delenv("PLX_EMU_STARVE")
    assert rc == 0 and np.abs(ux - ux0).max() > 0
    np.testing.assert_array_equal(ux[0], ux[1])
    three = ux.copy()
    # the time-out is on record, and the caller can switch the plan back once the GPU is its own again: the fused step then
    # reproduces the three-sweep result (frames co-resident this time)
    cnt = C.c_int32(-1)
    emu.call("plx_ssfm_barrier_timeouts", plan, C.byref(cnt), 0)
    emu.call("plx_ssfm_info", plan, info)
    assert cnt.value == 1 and info[0] == 0
    emu.call("plx_ssfm_barrier_timeouts", plan, None, 1)
    emu.call("plx_ssfm_info", plan, info)
    assert info[0] == 1
    ux2, uy2 = ux0.copy(), uy0.copy()
    emu.call("plx_ssfm_propagate_dev", plan, _vp(ux2), _vp(uy2), 2, None)
    assert np.abs(ux2 - three).max() < 1e-12 * np.abs(three).max()
    emu.call("plx_ssfm_destroy", plan)


@pytest.mark.parametrize("tolflag", [2, 1])
def test_emu_adaptive_ssfm(emu, oracle, tolflag):
    """scalar_a_ssfm / adaptssfm (fiber.m:639-679, 938-1009) and the dphiadapt first step (:588-611)."""
    n, nt, nfc, L = 512, 8, 2, 0.5e4
    fls = [1, 0, 1, 1]
    betat, db1 = _tables(n, nt, fls, 1, nfc)
    u = np.asfortranarray(np.stack([_qpsk_field(n, nt, 6.0, (2 + k, 5 + k))[0] for k in range(nfc)], 1))
    gam = [1.2e-6, 1.3e-6]
    dph = np.inf if tolflag == 2 else 2e-2
    d = _desc(n, nfc, 0, fls, L, 4.6e-5, gam, L, dph, betat, db1)
    ur, ui = np.asfortranarray(u.real.copy()), np.asfortranarray(u.imag.copy())
    fd, nc, nr = C.c_double(), C.c_int32(), C.c_int32()
    emu.call("plx_scalar_ssfm_adaptive", _vp(ur), _vp(ui), C.byref(d), tolflag, 1e-6, 0.9, C.byref(fd), C.byref(nc), C.byref(nr))
    if tolflag == 2:
        ofd, onc, onrej, ou = oracle.scalar_a_ssfm(u, betat, L, dph, gam, 4.6e-5, L, 1e-6, 0.9, fls)
        assert nr.value == onrej and onc > 3
    else:
        ofd, onc, ou = oracle.scalar_ssfm(u, betat, L, dph, gam, 4.6e-5, L, fls, tolflag=1, trg_err=1e-6, trg_safety=0.9)
        assert onc > 3
    assert nc.value == onc and fd.value == pytest.approx(ofd, rel=1e-9)
    assert np.abs((ur + 1j * ui) - ou).max() < 1e-9 * np.abs(ou).max()


def test_emu_ampliflat_injected_and_philox(emu):
    """ampliflat.m:78-148: gain, injected options.noise, one-polarisation ASE, and the statistics of the
    counter-based device noise (keyed per frame)."""
    n, nfc, F = 2048, 2, 2
    r = np.random.default_rng(4)
    ux = r.standard_normal((F, nfc, n)) + 1j * r.standard_normal((F, nfc, n))
    uy = r.standard_normal((F, nfc, n)) + 1j * r.standard_normal((F, nfc, n))
    noise = r.standard_normal((F, 2 * nfc, n)) + 1j * r.standard_normal((F, 2 * nfc, n))
    sigma = np.array([0.3, 0.5])
    gx, gy, nz = _il(ux), _il(uy), _il(noise)
    emu.call("plx_ampliflat_dev", _vp(gx), _vp(gy), n, nfc, F, 4.0, _vp(sigma), _vp(nz), 0, None, 1, 1, None)
    ox = gx.view(np.complex128).reshape(F, nfc, n); oy = gy.view(np.complex128).reshape(F, nfc, n)
    np.testing.assert_allclose(ox, 2.0 * ux + sigma[None, :, None] * noise[:, :nfc], rtol=1e-15, atol=1e-15)
    np.testing.assert_allclose(oy, 2.0 * uy + sigma[None, :, None] * noise[:, nfc:], rtol=1e-15, atol=1e-15)
    # 'asex': noise on x only; no sigma: pure gain
    gx, gy = _il(ux), _il(uy)
    emu.call("plx_ampliflat_dev", _vp(gx), _vp(gy), n, nfc, F, 4.0, _vp(sigma), _vp(nz), 0, None, 1, 0, None)
    np.testing.assert_allclose(gy.view(np.complex128).reshape(F, nfc, n), 2.0 * uy, rtol=1e-15)
    gx, gy = _il(ux), _il(uy)
    emu.call("plx_ampliflat_dev", _vp(gx), None, n, nfc, F, 0.25, None, None, 0, None, 1, 1, None)
    np.testing.assert_allclose(gx.view(np.complex128).reshape(F, nfc, n), 0.5 * ux, rtol=1e-15)
    # device Philox noise: zero field in, unit-variance quadratures out, frames keyed independently
    z = np.zeros(F * nfc * n * 2)
    zy = np.zeros_like(z)
    keys = np.array([7, 7], dtype=np.int64)
    one = np.ones(nfc)
    emu.call("plx_ampliflat_dev", _vp(z), _vp(zy), n, nfc, F, 1.0, _vp(one), None, 1234, _vp(keys), 1, 1, None)
    zz = z.view(np.complex128).reshape(F, nfc, n)
    np.testing.assert_array_equal(zz[0], zz[1])                          # same key -> same realisation
    assert abs(zz.real.std() - 1) < 0.03 and abs(zz.imag.std() - 1) < 0.03 and abs(zz.mean()) < 0.05
    assert abs(np.corrcoef(zz[0, 0].real, zz[0, 1].real)[0, 1]) < 0.08    # channels / polarisations independent
    assert abs(np.corrcoef(zz[0, 0].real, zy.view(np.complex128).reshape(F, nfc, n)[0, 0].real)[0, 1]) < 0.08
    z2 = np.zeros_like(z)
    keys2 = np.array([7, 8], dtype=np.int64)
    emu.call("plx_ampliflat_dev", _vp(z2), None, n, nfc, F, 1.0, _vp(one), None, 1234, _vp(keys2), 1, 1, None)
    w = z2.view(np.complex128).reshape(F, nfc, n)
    np.testing.assert_array_equal(w[0], zz[0])
    assert not np.array_equal(w[1], zz[1])


def _front_case(nsymb, nt, seed=5):
    from polmux_amd import rxfront
    n = nsymb * nt
    ux, uy, bits, pw = synth.pdm_qpsk_field(nsymb, nt, 2.0, 4, 5)
    fn = synth.fn_grid(nsymb, nt)
    omega = 2 * np.pi * 28 * fn
    hopt = np.exp(-1j * (0.5 * omega ** 2 * 3.1e-3)) * rxfront.myfilter("gauss", fn, 0.95)
    hel = rxfront.myfilter("bessel5", fn, 0.65)
    r = np.random.default_rng(seed)
    pn = np.cumsum(r.standard_normal(n)) * 0.01
    elo = 10 ** (1.5 / 20) * np.exp(1j * (2 * np.pi * 3 / n * np.arange(1, n + 1) + pn))
    return n, ux, uy, hopt, hel, elo


def _front_desc(n, dual, frames, hopt, hel, elo, balanced, bits, r, fir):
    from polmux_amd._abi import FrontDesc
    d = FrontDesc()
    d.nfft, d.dual_pol, d.max_frames, d.balanced, d.adcbits, d.decim = n, dual, frames, balanced, bits, r
    keep = [np.ascontiguousarray(v, dtype=np.float64) for v in (hopt.real, hopt.imag, hel.real, hel.imag)]
    d.hopt_re, d.hopt_im, d.hel_re, d.hel_im = (k.ctypes.data for k in keep)
    if np.ndim(elo):
        keep += [np.ascontiguousarray(elo.real), np.ascontiguousarray(elo.imag)]
        d.elo_re, d.elo_im = keep[-2].ctypes.data, keep[-1].ctypes.data
    else:
        d.elo_scalar = float(elo)
    if r > 1:
        keep.append(np.ascontiguousarray(fir, dtype=np.float64))
        d.ntaps, d.fir = len(fir), keep[-1].ctypes.data
    d._keep = keep
    return d


@pytest.mark.parametrize("dual,balanced,bits,r,lo_table", [(1, 1, 5, 16, True), (0, 0, 0, 1, False), (1, 1, 0, 8, False)])
def test_emu_front_end(emu, dual, balanced, bits, r, lo_table):
    """plx_front_*: optical filter + hybrids + photodiodes + low-pass (receiver_cohmix.m:183-307), ADC + fastshift +
    decimation (RxPdmCohQpsk.m:36-72) against oracle/front.py; two frames with different content."""
    from oracle import front
    from polmux_amd import rxfront
    n, ux, uy, hopt, hel, elo = _front_case(64, 16)
    if not lo_table:
        elo = 1.25
    fir = rxfront.fir1_lowpass(16, 1.0 / r) if r > 1 else None
    d = _front_desc(n, dual, 2, hopt, hel, elo, balanced, bits, r, fir)
    plan = C.c_void_p()
    emu.call("plx_front_create", C.byref(plan), C.byref(d))
    nout = emu.lib.plx_front_out_len(plan)
    assert nout == -(-n // r)
    fx = np.stack([ux, 0.7j * np.roll(ux, 37)]); fy = np.stack([uy, 1.3 * np.roll(uy, -11)])
    gx, gy = _il(fx), _il(fy)
    out = np.zeros(2 * (dual + 1) * nout * 2)
    sh = (C.c_int64 * 2)(-13, 21)
    emu.call("plx_front_run_dev", plan, _vp(gx), _vp(gy) if dual else None, 2, sh, _vp(out), None)
    emu.call("plx_front_destroy", plan)
    out = out.view(np.complex128).reshape(2, dual + 1, nout)
    cur_x = gx.view(np.complex128).reshape(2, n); cur_y = gy.view(np.complex128).reshape(2, n)
    for f in range(2):
        want = front.receiver_cohmix(fx[f], fy[f] if dual else None, hopt, elo, hel, bool(balanced))
        got = np.stack([cur_x[f].real, cur_x[f].imag] + ([cur_y[f].real, cur_y[f].imag] if dual else []), 1)
        assert np.abs(got - want).max() < 1e-12 * np.abs(want).max()
        # downstream of the (discontinuous) ADC: the oracle continues from the device's own currents
        rx = front.rx_front(got, bool(dual), bits, [-13, 21 if dual else -13], r, fir)
        assert np.abs(out[f].T - rx).max() <= 1e-14 * np.abs(rx).max()
        full = front.rx_front(want, bool(dual), bits, [-13, 21 if dual else -13], r, fir)
        assert np.mean(np.abs(out[f].T - full) > 1e-9 * np.abs(full).max()) < 0.01   # at most isolated LSB flips


def test_emu_inverse_pmd(emu, oracle):
    """plx_pmdinv_*: U / Uinv per frequency (inverse_pmd.m:103-136) and the application to two frames with their own
    waveplate draws (:139-145) against oracle/pmdinv.py; options.mat and options.gvd = 'no'."""
    from oracle import pmdinv
    nsymb, nt, n = 64, 16, 1024
    fn = synth.fn_grid(nsymb, nt)
    omega = 2 * np.pi * 10.0 * fn
    betat = 0.5 * omega ** 2 * -2.17e-8 + omega ** 3 * 1.3e-10 / 6
    links = []
    for nplates, seed, L, dgd in ((6, 1, 5e4, 0.3), (3, 2, 2e4, 0.6)):
        r = np.random.default_rng(seed)
        sets = [(r.random(nplates) * 2 * np.pi - np.pi, r.random(nplates) * np.pi - np.pi / 2, 0.5 * np.arcsin(r.random(nplates) * 2 - 1))
                for _ in range(2)]
        links.append(dict(nplates=nplates, lcorr=L / nplates, betat=betat * (1 + 0.1 * seed), db1=dgd / nplates / 10.0 * omega, sets=sets))
    ux, uy, _, _ = synth.pdm_qpsk_field(nsymb, nt, 2.0)
    fx = np.stack([ux, 1j * np.roll(uy, 5)]); fy = np.stack([uy, 0.5 * np.roll(ux, -9)])
    c, s = np.cos(0.3), np.sin(0.3)
    M = np.array([[c, s * 1j], [s * 1j, c]])
    for opts in (None, dict(mat=M), dict(gvd="no")):
        plan = C.c_void_p()
        emu.call("plx_pmdinv_create", C.byref(plan), n, 2)
        ntr = np.array([l["nplates"] for l in links], dtype=np.int32)
        cat = lambda i: np.ascontiguousarray(np.stack([np.concatenate([l["sets"][f][i] for l in links]) for f in range(2)]))
        db0, th, ep = cat(0), cat(1), cat(2)
        lc = np.array([l["lcorr"] for l in links])
        bt = np.ascontiguousarray(np.stack([l["betat"] for l in links])); d1 = np.ascontiguousarray(np.stack([l["db1"] for l in links]))
        mat = None
        if opts and "mat" in opts:
            mat = np.ascontiguousarray(np.stack([M.real, M.imag], -1).reshape(-1))
        emu.call("plx_pmdinv_set_link", plan, 2, _vp(ntr), _vp(db0), _vp(th), _vp(ep), _vp(lc), _vp(bt), _vp(d1),
                 _vp(mat) if mat is not None else None, 0 if (opts and opts.get("gvd") == "no") else 1, 2)
        gx, gy = _il(fx), _il(fy)
        emu.call("plx_pmdinv_apply_dev", plan, _vp(gx), _vp(gy), 2, None)
        gx = gx.view(np.complex128).reshape(2, n); gy = gy.view(np.complex128).reshape(2, n)
        for f in range(2):
            brf = [dict(db0=l["sets"][f][0], theta=l["sets"][f][1], epsilon=l["sets"][f][2], lcorr=l["lcorr"], betat=l["betat"],
                        db1=l["db1"]) for l in links]
            Uinv, U, wx, wy = pmdinv.inverse_pmd(brf, fx[f], fy[f], opts)
            assert np.abs(gx[f] - wx).max() < 1e-12 * np.abs(wx).max()
            assert np.abs(gy[f] - wy).max() < 1e-12 * np.abs(wy).max()
            gU = np.zeros((n, 2, 2), dtype=complex); gUi = np.zeros_like(gU)
            emu.call("plx_pmdinv_matrices", plan, f, _vp(gU), _vp(gUi))
            np.testing.assert_allclose(np.transpose(gU, (2, 1, 0)), U, atol=1e-13)
            np.testing.assert_allclose(np.transpose(gUi, (2, 1, 0)), Uinv, atol=1e-13)
        emu.call("plx_pmdinv_destroy", plan)
    plan = C.c_void_p()
    emu.call("plx_pmdinv_create", C.byref(plan), n, 1)
    with pytest.raises(Exception, match="set_link has not been called"):
        emu.call("plx_pmdinv_apply_dev", plan, _vp(_il(fx)), _vp(_il(fy)), 1, None)
    emu.call("plx_pmdinv_destroy", plan)


def test_emu_new_plan_argument_errors(emu):
    """plx_front_*, plx_pmdinv_*, plx_filter_*: bad arguments are refused with PLX_ERR_ARG / PLX_ERR_UNSUPPORTED and a message."""
    from polmux_amd._abi import PLX_ERR_ARG, PLX_ERR_UNSUPPORTED, PolmuxError
    n = 1024
    one = np.ones(n)
    plan = C.c_void_p()

    def code(fn, *args):
        with pytest.raises(PolmuxError) as ei:
            emu.call(fn, *args)
        assert emu.last_error()
        return ei.value.code

    d = _front_desc(n, 1, 1, one + 0j, one + 0j, 1.0, 1, 5, 16, np.ones(16) / 16)       # even tap count
    assert code("plx_front_create", C.byref(plan), C.byref(d)) == PLX_ERR_ARG
    d = _front_desc(n, 1, 1, one + 0j, one + 0j, 1.0, 1, 40, 1, None)                   # adcbits out of range
    assert code("plx_front_create", C.byref(plan), C.byref(d)) == PLX_ERR_ARG
    d = _front_desc(1000, 1, 1, np.ones(1000) + 0j, np.ones(1000) + 0j, 1.0, 1, 0, 1, None)   # not a power of two
    assert code("plx_front_create", C.byref(plan), C.byref(d)) == PLX_ERR_UNSUPPORTED
    d = _front_desc(n, 1, 2, one + 0j, one + 0j, 1.0, 1, 0, 1, None)
    emu.call("plx_front_create", C.byref(plan), C.byref(d))
    x = np.zeros(2 * 2 * n)
    out = np.zeros(2 * 2 * 2 * n)
    assert code("plx_front_run_dev", plan, _vp(x), None, 1, None, _vp(out), None) == PLX_ERR_ARG      # dual plan needs uy
    assert code("plx_front_run_dev", plan, _vp(x), _vp(x), 3, None, _vp(out), None) == PLX_ERR_ARG    # more than max_frames
    emu.call("plx_front_destroy", plan)
    assert code("plx_filter_create", C.byref(plan), n, 0, _vp(one), None) == PLX_ERR_ARG
    assert code("plx_filter_create", C.byref(plan), 100, 1, _vp(one), None) == PLX_ERR_UNSUPPORTED
    emu.call("plx_filter_create", C.byref(plan), n, 2, _vp(one), None)
    assert code("plx_filter_apply_dev", plan, _vp(x), 3, None) == PLX_ERR_ARG
    y = _il(np.arange(2 * n) + 1j)
    emu.call("plx_filter_apply_dev", plan, _vp(y), 2, None)                                # H = 1: identity
    np.testing.assert_allclose(y.view(np.complex128), np.arange(2 * n) + 1j, atol=1e-9)
    emu.call("plx_filter_destroy", plan)
    assert code("plx_pmdinv_create", C.byref(plan), n, 0) == PLX_ERR_ARG
    emu.call("plx_pmdinv_create", C.byref(plan), n, 1)
    nt0 = np.array([0], dtype=np.int32)
    z = np.zeros(4)
    assert code("plx_pmdinv_set_link", plan, 1, _vp(nt0), _vp(z), _vp(z), _vp(z), _vp(z), _vp(one), _vp(one), None, 1, 1) == PLX_ERR_ARG
    nt1 = np.array([2], dtype=np.int32)
    assert code("plx_pmdinv_set_link", plan, 1, _vp(nt1), _vp(z), _vp(z), _vp(z), _vp(z), _vp(one), _vp(one), None, 1, 2) == PLX_ERR_ARG
    emu.call("plx_pmdinv_destroy", plan)


def test_emu_poldemux_driver_gateways(emu, oracle):
    """plx_cmapolardemux / plx_easipolardemux: the whole driver loop (DspPdmCohQpsk.m:142-244) as one gateway call on host arrays
    with MATLAB's separate planes -- staging layout x | M up, y | h | passes down --, against the oracle's driver loop.  L = 44
    leaves a tail behind the last whole chunk of eight symbols; three taps make a one-sample cyclic extension."""
    for L, taps, mu, phi in ((44, 7, 1 / 40, 0.0), (40, 3, 1 / 60, 0.25), (24, 1, 1 / 30, 0.0)):
        x = _mixed_qpsk(L, 31 + taps, noise=0.003, th=0.03)
        M = np.array([[np.cos(phi), np.sin(phi)], [-np.sin(phi), np.cos(phi)]], dtype=complex)       # :157-158
        xr, xi = np.asfortranarray(x.real), np.asfortranarray(x.imag)
        yr, yi = np.zeros((L, 2), order="F"), np.zeros((L, 2), order="F")
        h = [np.zeros((taps, 2), order="F") for _ in range(4)]
        n = C.c_int32()
        R = np.array([1.0, 1.0])
        Mi = _il(M.reshape(-1))
        emu.call("plx_cmapolardemux", _vp(xr), _vp(xi), L, taps, mu, _vp(R), _vp(Mi), _vp(yr), _vp(yi), _vp(h[0]), _vp(h[1]), _vp(h[2]),
                 _vp(h[3]), C.byref(n))
        oy, h1, h2, on = oracle.cmapolardemux(x, M, taps, mu, R)
        assert n.value == on and 1 <= on < 50 * int(np.ceil(1 / (L * mu)))
        np.testing.assert_allclose(yr + 1j * yi, oy, atol=1e-11)
        np.testing.assert_allclose(h[0] + 1j * h[1], h1.reshape(taps, 2), atol=1e-11)
        np.testing.assert_allclose(h[2] + 1j * h[3], h2.reshape(taps, 2), atol=1e-11)
    # a real-valued input without an imaginary plane, outputs without the optional taps / pass count
    xr = np.asfortranarray(np.sign(np.random.default_rng(2).standard_normal((32, 2))))
    yr, yi = np.zeros((32, 2), order="F"), np.zeros((32, 2), order="F")
    emu.call("plx_cmapolardemux", _vp(xr), None, 32, 3, 1 / 50, _vp(R), _vp(_il(np.eye(2, dtype=complex).reshape(-1))), _vp(yr), _vp(yi),
             None, None, None, None, None)
    oy = oracle.cmapolardemux(xr.astype(complex), np.eye(2), 3, 1 / 50, R)[0]
    np.testing.assert_allclose(yr + 1j * yi, oy, atol=1e-11)
    with pytest.raises(Exception, match="Ntaps should be an ODD INTEGER"):
        emu.call("plx_cmapolardemux", _vp(xr), None, 32, 4, 1 / 50, _vp(R), _vp(Mi), _vp(yr), _vp(yi), None, None, None, None, None)
    # EASI: the C filter and its .m twin
    L, mu = 40, 1 / 40
    x = _mixed_qpsk(L, 77, noise=0.003, th=0.02)
    xr, xi = np.asfortranarray(x.real), np.asfortranarray(x.imag)
    for twin, ref in ((0, oracle.easipolardemux), (1, oracle.easipolardemux_m)):
        yr, yi = np.zeros((L, 2), order="F"), np.zeros((L, 2), order="F")
        n = C.c_int32()
        emu.call("plx_easipolardemux", _vp(xr), _vp(xi), L, mu, _vp(_il(np.eye(2, dtype=complex).reshape(-1))), twin, _vp(yr), _vp(yi),
                 None, None, None, None, C.byref(n))
        oy, _, _, on = ref(x, np.eye(2), mu)
        assert n.value == on
        np.testing.assert_allclose(yr + 1j * yi, oy, atol=1e-11)


def test_emu_step_sequence_replay_and_log(emu, oracle, tune):
    """plx_ssfm_set_step_sequence / plx_ssfm_log_steps (the diagnostics behind the config[2] parity test), fused sweep and
    three-sweep step: a plan that replays the oracle's list of step lengths makes the oracle's steps (same ncycle, first step
    bit-equal, field to 1e-12); left to itself it logs a sequence that agrees with the oracle's to 1e-12 on this
    band-limited frame; a replayed list that differs from the rule's changes the result (the list really is used)."""
    n, nt, L = 4096, 64, 4e3
    fls = [1, 0, 1, 0]
    betat, db1 = _tables(n, nt, fls, 1)
    f = _qpsk_field(n, nt, 40.0)
    rc, ofd, onc, ox, oy, odz = oracle.matrix_ssfm(f[0], f[1], betat, db1, 2e3, 0.08, [1.3e-6], 4.6e-5, L, 1, 0, fls, [0.0], [0.0], [0.0],
                                                   return_dz=True)
    assert rc == 0 and onc == len(odz) and 4 <= onc <= 12
    for env in ({"PLX_EMU_CUS": "1"}, {"PLX_SSFM_NO_FUSE": "1"}):
        for k, v in env.items():
            tune.setenv(k, v)
        d = _desc(n, 1, 1, fls, L, 4.6e-5, [1.3e-6], 2e3, 0.08, betat, db1)
        plan = C.c_void_p()
        emu.call("plx_ssfm_create", C.byref(plan), C.byref(d))
        for k in env:
            tune.delenv(k)
        nc, fd = np.zeros(1, np.int32), np.zeros(1)

        def run():
            ux, uy = _il(f[0]), _il(f[1])
            emu.call("plx_ssfm_propagate_dev", plan, _vp(ux), _vp(uy), 1, None)
            emu.call("plx_ssfm_results", plan, 1, _vp(fd), _vp(nc))
            return ux.view(np.complex128), uy.view(np.complex128)
        emu.call("plx_ssfm_log_steps", plan, 64)
        gx, gy = run()                                              # free-running, logged
        log = np.zeros(64)
        emu.call("plx_ssfm_step_sequence", plan, 0, _vp(log), 64)
        assert nc[0] == onc
        np.testing.assert_allclose(log[:onc], odz, rtol=1e-12)
        emu.call("plx_ssfm_log_steps", plan, 0)
        emu.call("plx_ssfm_set_step_sequence", plan, _vp(np.ascontiguousarray(odz)), len(odz))
        rx, ry = run()                                              # the oracle's sequence replayed
        assert nc[0] == onc and fd[0] == ofd
        assert np.abs(rx - ox[:, 0]).max() < 1e-12 * np.abs(ox).max() and np.abs(ry - oy[:, 0]).max() < 1e-12 * np.abs(oy).max()
        other = np.full(3, L / 3 + 1.0)                             # three equal steps instead of the rule's
        emu.call("plx_ssfm_set_step_sequence", plan, _vp(other), 3)
        sx, _ = run()
        assert nc[0] == 3 and np.abs(sx - rx).max() > 1e-6 * np.abs(rx).max()
        emu.call("plx_ssfm_set_step_sequence", plan, None, 0)
        tx, _ = run()                                               # switched off again: the free-running result
        np.testing.assert_array_equal(tx, gx)
        emu.call("plx_ssfm_destroy", plan)

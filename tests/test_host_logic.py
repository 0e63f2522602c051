"""CPU tests (-m "not gpu") of the host side: fiber.m flag/physics mirror, the MC estimators against the
oracle, the synthetic Tx against the reference's literal, the C ABI surface (header vs library), and the
no-fallback rule."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --------------------------------------------------------------------- C ABI ---
def _header_symbols():
    src = open(os.path.join(ROOT, "include", "polmux_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(plx_[a-z0-9_]+)\s*\(", src)))


def test_abi_library_exports_every_declared_symbol():
    """The hipcc-built library loads (no GPU needed) and exports every symbol include/polmux_hip.h declares;
    the ctypes table binds exactly that set."""
    from polmux_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    syms = _header_symbols()
    assert len(syms) >= 25
    lib = C.CDLL(_abi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "library does not export %s" % s
    assert sorted(_abi.SIGNATURES) + ["plx_last_error"] == sorted(syms + []) or \
        sorted(list(_abi.SIGNATURES) + ["plx_last_error"]) == syms
    b = _abi.Binding()
    assert b.lib.plx_abi_version() == 1002
    # no compute without a GPU, but argument validation works everywhere
    with pytest.raises(_abi.PolmuxError, match="null argument"):
        b.call("plx_ssfm_create", None, None)


def test_plan_tuning_is_a_struct_not_the_environment(monkeypatch):
    """include/polmux_hip.h, plx_ssfm_tuning (ABI 1002): the defaults consult exactly two environment variables -- the deployment
    knobs PLX_SSFM_NO_FUSE and PLX_SSFM_BARRIER_TIMEOUT_MS -- and none of the seventeen A/B switches of ABI 1001; a struct that was
    not filled by plx_ssfm_tuning_defaults (size 0) is refused; the library's sources read nothing else from the environment."""
    from polmux_amd import _abi
    b = _abi.Binding()
    for k in ("PLX_SSFM_ROWR", "PLX_SSFM_P1", "PLX_SSFM_NO_PMD_TAB", "PLX_SSFM_ROW_REV", "PLX_SSFM_STORE_LATE", "PLX_SSFM_SHORT_ROWS"):
        monkeypatch.setenv(k, "0" if k != "PLX_SSFM_P1" else "4")
    t = b.tuning()
    assert t.size == C.sizeof(_abi.SsfmTuning) and (t.rowr, t.p1, t.no_pmd_tab, t.row_rev, t.store_late, t.short_rows) == (1, -1, 0, 1, -1, 0)
    assert t.no_fuse == 0 and t.barrier_timeout_ms == 500.0
    monkeypatch.setenv("PLX_SSFM_NO_FUSE", "1")
    monkeypatch.setenv("PLX_SSFM_BARRIER_TIMEOUT_MS", "125")
    t = b.tuning()
    assert t.no_fuse == 1 and t.barrier_timeout_ms == 125.0
    with pytest.raises(KeyError):
        b.tuning(no_such_switch=1)
    bad = _abi.SsfmTuning()                                    # size 0: never went through plx_ssfm_tuning_defaults
    with pytest.raises(_abi.PolmuxError, match="plx_ssfm_tuning_defaults first"):
        b.call("plx_ssfm_tuning_override", C.byref(bad))
    b.call("plx_ssfm_tuning_override", C.byref(b.tuning(rowr=0)))
    b.call("plx_ssfm_tuning_override", None)
    reads = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "polmux_amd", "csrc"))):
        src = open(os.path.join(ROOT, "polmux_amd", "csrc", fn)).read()
        reads += [(fn, ln.strip()) for ln in src.splitlines() if "getenv(" in ln]
    assert len(reads) == 2 and all(fn == "ssfm_plan.hip" for fn, _ in reads), reads
    assert "PLX_SSFM_NO_FUSE" in reads[0][1] and "PLX_SSFM_BARRIER_TIMEOUT_MS" in reads[1][1]


def test_no_cpu_fallback_and_oracle_is_never_imported():
    """The product fails loudly when the HIP library is missing, and no module of polmux_amd touches oracle/."""
    from polmux_amd import _abi
    with pytest.raises(_abi.PolmuxError, match="no CPU fallback"):
        _abi.Binding(os.path.join(ROOT, "polmux_amd", "lib", "missing", "libpolmux_hip.so"))
    for fn in os.listdir(os.path.join(ROOT, "polmux_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "polmux_amd", fn)).read()
            assert "oracle" not in src.replace("# noqa", ""), fn
    for fn in os.listdir(os.path.join(ROOT, "polmux_amd", "csrc")):
        src = open(os.path.join(ROOT, "polmux_amd", "csrc", fn)).read()
        assert "plxo" not in src and "oracle/" not in src, fn


def test_fiber_needs_gpu_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import polmux_amd as px
    px.reset_all(16, 16, 1)
    px.GSTATE.SYMBOLRATE = 10.0
    px.lasersource(1.0, 1550.0)
    with pytest.raises(px.PolmuxError, match="no CPU path"):
        px.create_field("sepfields", np.ones((256, 1), complex), np.ones((256, 1), complex))


# ------------------------------------------------------------------- fiber.m ---
def test_parse_flag_table():
    """fiber.m:157-251."""
    from polmux_amd.fiber import parse_flag
    x = dict(length=8e4, dzmax=2e4, dphimax=5e-3)
    inf = math.inf
    assert parse_flag("g---", 1, x) == ([1, 0, 0, 0], inf, 8e4)
    assert parse_flag("----", 1, x) == ([0, 0, 0, 0], inf, 8e4)
    assert parse_flag("gp--", 1, x) == ([1, 1, 0, 0], inf, 8e4)
    assert parse_flag("--s-", 1, x) == ([0, 0, 1, 0], inf, 8e4)            # exact solution, one field (:172-174)
    assert parse_flag("--s-", 3, x) == ([0, 0, 1, 0], 5e-3, 2e4)
    assert parse_flag("--sx", 1, x) == ([0, 0, 1, 0], inf, 8e4)
    assert parse_flag("--sx", 2, x) == ([0, 0, 1, 1], 5e-3, 2e4)
    assert parse_flag("g-sx", 1, x) == ([1, 0, 1, 0], 5e-3, 2e4)            # xpm only with separate fields (:224)
    assert parse_flag("g-sx", 5, x) == ([1, 0, 1, 1], 5e-3, 2e4)
    assert parse_flag("GPS-", 1, x) == ([1, 1, 1, 0], 5e-3, 2e4)            # lower(flag)
    assert parse_flag("gpsx", 2, x) == ([1, 1, 1, 1], 5e-3, 2e4)
    assert parse_flag("gp-x", 2, x) == ([1, 1, 0, 1], 5e-3, 2e4)
    for f in ("---x", "g--x", "-p-x", "gp-x"):
        with pytest.raises(ValueError, match="available only for channels separated"):
            parse_flag(f, 1, x)
    with pytest.raises(ValueError, match="wrong flag"):
        parse_flag("-s--", 1, x)


def test_fiber_tables_physics():
    """fiber.m:302-362 for SSMF at 1550 nm."""
    from polmux_amd import synth
    from polmux_amd.fiber import fiber_tables
    from polmux_amd.gstate import GSTATE
    GSTATE.NSYMB, GSTATE.NT, GSTATE.NCH = 64, 16, 1
    GSTATE.SYMBOLRATE = 28.0
    GSTATE.FN = synth.fn_grid(64, 16)
    GSTATE.LAMBDA = np.array([1550.0])
    x = {"alphadB": 0.2, "aeff": 80.0, "n2": 2.7e-20, "lambda": 1550.0, "disp": 17.0, "slope": 0.0}
    t = fiber_tables(x, [1, 1, 1, 0], 1, 0.05)
    assert t["alphalin"] == pytest.approx(0.2 * math.log(10) * 1e-4)
    assert t["gam"][0] == pytest.approx(1.368e-6, rel=1e-3)                    # ~1.37 /W/km
    omega = 2 * math.pi * 28.0 * GSTATE.FN
    b2 = -1550.0 ** 2 / (2 * math.pi * 299792458.0) * 17.0 * 1e-6            # ns^2/m: -21.7 ps^2/km
    assert b2 == pytest.approx(-2.168e-8, rel=1e-3)
    b3 = (1550.0 / (2 * math.pi * 299792458.0)) ** 2 * (2 * 1550.0 * 17.0) * 1e-6   # fiber.m:309 with slope = 0
    np.testing.assert_allclose(t["betat"][:, 0], 0.5 * omega ** 2 * b2 + omega ** 3 * b3 / 6, rtol=1e-12, atol=1e-20)
    np.testing.assert_allclose(t["db1"][:, 0], 0.05 / 28.0 * omega, rtol=1e-14)
    assert GSTATE.FN[0] == 0 and GSTATE.FN[1] == 1 / 64 and GSTATE.FN[512] == -8.0   # reset_all.m:152-153
    # GVD flag off: no beta2/beta3, walk-off only (fiber.m:332-334)
    t0 = fiber_tables(x, [0, 0, 1, 0], 1, 0.0)
    assert not t0["betat"].any() and not t0["db1"].any()
    # separate channels: per-channel beta1 = b1 and gamma (fiber.m:326-328)
    GSTATE.NCH, GSTATE.LAMBDA = 3, np.array([1549.6, 1550.0, 1550.4])
    t3 = fiber_tables(x, [1, 0, 1, 1], 3, 0.0)
    assert t3["betat"].shape == (1024, 3) and t3["gam"].shape == (3,)
    assert t3["gam"][0] > t3["gam"][2]
    assert t3["b1"][0] * t3["b1"][2] < 0 and abs(t3["b1"][1]) < 1e-3 * abs(t3["b1"][0])


def test_reset_all_and_lasersource():
    import polmux_amd as px
    g = px.reset_all(32, 8, 3)
    assert (g.NSYMB, g.NT, g.NCH) == (32, 8, 3) and g.FN.size == 256
    px.lasersource(2.0, 1550.0, 0.4)
    np.testing.assert_allclose(px.GSTATE.LAMBDA, [1549.6, 1550.0, 1550.4])        # lasersource.m:160-163
    np.testing.assert_array_equal(px.GSTATE.POWER, [2.0, 2.0, 2.0])
    with pytest.raises(ValueError, match="missing the channel-spacing"):
        px.lasersource(2.0, 1550.0)
    # the reference's own documented example, lasersource.m:18: Nch=4, lam=1550, spac=0.8 -> lambda=[1548.8 1549.6 1550.4 1551.2]
    px.reset_all(32, 8, 4)
    px.lasersource(1.0, 1550.0, 0.8)
    np.testing.assert_allclose(px.GSTATE.LAMBDA, [1548.8, 1549.6, 1550.4, 1551.2], rtol=0, atol=1e-9)


def test_corrdelay_reference_signal_of_the_documented_example():
    """corrdelay.m:12-15: 'if PAT='1101' and Nt=4 the reference transmitted signal turns out to be ... x=1111111100001111': a
    current equal to that signal delayed by k samples is found k samples late, reported as (k + Nt/2) / Nt symbols (:113: the
    delay is counted from the symbol's centre), with rho = 2 max(c) / Nfft (:114) = 2 x 12 ones / 16."""
    import polmux_amd as px
    pat, nt = np.array([1, 1, 0, 1]), 4
    x = np.array([int(c) for c in "1111111100001111"], dtype=float)
    np.testing.assert_array_equal(np.repeat(pat, nt), x)                     # the signal the function builds, :91
    for k in (0, 3, 6):
        delay, wrn, rho, _ = px.corrdelay(np.roll(x, k), pat, nt, 4)
        assert delay == pytest.approx((k + nt / 2) / nt) and rho == pytest.approx(1.5) and not wrn


# --------------------------------------------------------------------- synth ---
def test_debruijn_against_reference_literal():
    """pattern.m:48-51 documents pattern('debruijn',0,alphabet 4) @ Nsymb=16 as
    [1 1 2 3 0 3 1 3 3 2 2 1 0 2 0 0].  The shipped code weights the two bit planes [2 1]
    (pattern.m:305: 2.^(q-1:-1:0)); the doc string's sequence is the same pair of planes weighted [1 2].
    Both are checked: identical binary De Bruijn planes, and the code's weighting."""
    from polmux_amd import synth
    pat, bmat = synth.pattern_debruijn(16, 0, 4)
    doc = np.array([1, 1, 2, 3, 0, 3, 1, 3, 3, 2, 2, 1, 0, 2, 0, 0])
    np.testing.assert_array_equal(bmat[:, 0] + 2 * bmat[:, 1], doc)                # the literal, bit planes swapped
    np.testing.assert_array_equal(2 * bmat[:, 0] + bmat[:, 1], pat)                # the code's weighting
    np.testing.assert_array_equal(bmat[:, 1], np.roll(bmat[:, 0], 2))             # plane 2 = plane 1 shifted by ns/q
    for nsymb, q in ((64, 4), (256, 4), (1024, 4), (32, 2)):
        p, _ = synth.pattern_debruijn(nsymb, 3, q)
        k = int(round(math.log(nsymb, q)))
        words = {tuple(p[(i + np.arange(k)) % nsymb]) for i in range(nsymb)}
        assert len(words) == nsymb                                                 # every k-tuple exactly once
    with pytest.raises(ValueError, match="does not exist"):
        synth.pattern_debruijn(32, 0, 4)


def test_tx_waveform_power_and_symbols():
    from polmux_amd import synth
    ux, uy, bits, p = synth.pdm_qpsk_field(64, 16, 2.0)
    assert np.mean(np.abs(ux) ** 2 + np.abs(uy) ** 2) == pytest.approx(2.0, rel=1e-12)   # create_field.m:113-124
    centre = ux[::16]
    np.testing.assert_array_equal(centre.real > 0, bits[:, 0] == 1)               # I = 2*bit-1 at the symbol centre
    np.testing.assert_array_equal(centre.imag > 0, bits[:, 1] == 1)
    assert np.allclose(np.abs(centre), np.abs(centre[0]))


# ------------------------------------------------------------ MC estimators ---
def test_ber_estimate_matches_oracle_bitwise(oracle):
    from polmux_amd import mc
    r = np.random.default_rng(1)
    for stop, nmin in ((None, 30), ((0.1, 95), 1), ((0.05, 68), 5)):
        mc.reset_persistent()
        st = oracle.McState()
        x = dict(nmin=nmin)
        if stop:
            x["stop"] = stop
        for it in range(4000):
            pat = r.integers(0, 2, (64, 4))
            hat = pat ^ (r.random((64, 4)) < 0.02)
            a = mc.ber_estimate(hat, pat, x)
            b = oracle.ber_estimate(st, hat, pat, stop=stop, nmin=nmin)
            for u, v in zip(a, b):
                np.testing.assert_array_equal(np.asarray(u, dtype=float), np.asarray(v, dtype=float))
            if not a[0][0]:
                break
        assert not a[0][0] and it > 2
        assert mc._ber_state.first is False                                       # persistent state cleared (:132,137)


def test_ber_estimate_vector_mode_and_errors(oracle):
    from polmux_amd import mc
    mc.reset_persistent()
    st = oracle.McState()
    r = np.random.default_rng(2)
    x = dict(stop=(0.2, 90), nmin=1, dim=3)
    conds = np.ones(3, bool)
    for it in range(600):
        for nind in (1, 2, 3):
            if not conds[nind - 1]:
                continue                                  # the scripts stop calling an index once its cond is false
            pat = r.integers(0, 2, (32, 2))
            hat = pat ^ (r.random((32, 2)) < 0.03 * nind)
            a = mc.ber_estimate(hat, pat, x, nind)
            b = oracle.ber_estimate(st, hat, pat, stop=(0.2, 90), nmin=1, dim=3, nind=nind)
            for u, v in zip(a, b):
                np.testing.assert_array_equal(np.asarray(u, dtype=float), np.asarray(v, dtype=float))
            conds = a[0]
        if not conds.any():
            break
    assert not conds.any() and mc._ber_state.first is False
    with pytest.raises(ValueError, match="missing variable nind"):
        mc.ber_estimate(hat, pat, dict(nmin=1), 2)
    with pytest.raises(ValueError, match="Gaussian confidence"):
        mc.ber_estimate(hat, pat, dict(stop=(0.1, 120)))


def test_mc_estimate_matches_oracle(oracle):
    from polmux_amd import mc
    r = np.random.default_rng(3)
    for method in ("mean", "var"):
        mc.reset_persistent()
        st = oracle.McState()
        x = dict(stop=(0.05, 95), nmin=50, method=method)
        for it in range(3000):
            s = 3.0 + r.standard_normal(25)
            c1, o1 = mc.mc_estimate(s, x)
            c2, o2 = oracle.mc_estimate(st, s, stop=(0.05, 95), nmin=50, method=method)
            assert c1[0] == c2[0]
            for k in ("mean", "var", "nruns", "stdmean", "varlim"):
                np.testing.assert_allclose(o1[k], o2[k], rtol=1e-13, atol=0)
            if not c1[0]:
                break
        assert not c1[0]
    assert mc.erfcinv(0.05) == pytest.approx(oracle.erfcinv(0.05), rel=1e-15)


def test_ex23_style_coverage():
    """(x) ex23_test_mc_estimate.m:36-49: the confidence interval contains the true mean about conf % of the time."""
    from polmux_amd import mc
    r = np.random.default_rng(5)
    hit = 0
    for rep in range(100):
        mc.reset_persistent()
        cond = [True]
        while cond[0]:
            cond, out = mc.mc_estimate(2.0 + r.standard_normal(20), dict(stop=(0.05, 95), nmin=50))
        eps = math.sqrt(2) * mc.erfcinv(1 - 0.95)
        hit += abs(out["mean"][0] - 2.0) < eps * out["stdmean"][0]
    assert 85 <= hit <= 100


def test_samp2pat_host():
    import polmux_amd as px
    ph = np.array([[math.pi / 4, -math.pi / 4], [3 * math.pi / 4, -3 * math.pi / 4]])
    np.testing.assert_array_equal(px.samp2pat(dict(rec="coherent"), None, ph), [[1, 1, 1, 0], [0, 1, 0, 0]])
    with pytest.raises(ValueError, match="Wrong modulation format"):
        px.samp2pat(dict(rec="ook"), None, ph)


def test_pat_decoder_host_logic_matches_reference_text():
    """pat_decoder.m:66-79 / pat2stars.m / stars2pat.m: differential QPSK decoding, quaternary and binary routes."""
    from polmux_amd import patterns, synth
    pat, _ = synth.pattern_debruijn(64, 1, 4)
    p, pm = patterns.pat_decoder(pat, "dqpsk")
    st = patterns.pat2stars(pat, "dqpsk")
    np.testing.assert_array_equal(st, np.array([1, 1j, -1j, -1])[pat])
    d = np.conj(st) * np.roll(st, 1)                                  # conj(stars_t).*fastshift(stars_t,1)
    q, qm = patterns.stars2pat(d, "dqpsk")
    np.testing.assert_array_equal(p, 3 - q)
    np.testing.assert_array_equal(pm, 1 - qm)
    _, bm = patterns.stars2pat(st, "dqpsk")
    p2, pm2 = patterns.pat_decoder(bm, "dqpsk", dict(binary=True))
    np.testing.assert_array_equal(p2, p)
    np.testing.assert_array_equal(pm2, pm)
    # dpsk: stars +-1, conj(s).*fastshift(s,1) = [1,-1,1,-1] -> stars2pat [0,1,0,1] -> inverted
    np.testing.assert_array_equal(patterns.pat_decoder(np.array([0, 1, 1, 0]), "dpsk"), [1, 0, 1, 0])
    with pytest.raises(ValueError, match="wrong modulation format"):
        patterns.pat_decoder(pat, "nope")


def test_corrdelay_recovers_delay_and_phase_ambiguity():
    """corrdelay.m:62-116: binary route on a delayed NRZ current; 'phase' route on a delayed, rotated QPSK phasor."""
    import polmux_amd as px
    rng = np.random.default_rng(11)
    nt, nsymb = 16, 256
    bits = rng.integers(0, 2, nsymb).astype(float)
    nrz = np.repeat(bits, nt)
    for d in (0, 5, 37, nt * nsymb - 3):
        cur = np.roll(nrz, d) + 0.05 * rng.standard_normal(nrz.size)
        delay, wrn, rho, out = px.corrdelay(cur, bits, nt, nsymb)
        assert delay == (d + nt / 2) / nt and not wrn and np.array_equal(out, cur)       # +Nt/2: the first bit is centred on sample 1
        assert abs(rho - 2 * np.mean(bits)) < 0.05                        # maxc/Nfft*2 of a 0/1 signal
    sym = rng.integers(0, 4, nsymb)
    phases = np.array([-0.75, 0.75, -0.25, 0.25])[sym] * math.pi          # RxPdmCohQpsk.m:118-121
    ref = np.repeat(np.exp(1j * phases), nt)
    for d, rot in ((0, 0.0), (21, 0.9), (100, -2.4), (7, math.pi)):
        cur = np.roll(ref, d) * np.exp(1j * rot) + 0.05 * (rng.standard_normal(ref.size) + 1j * rng.standard_normal(ref.size))
        delay, wrn, rho, ang = px.corrdelay(cur, phases, nt, nsymb, "phase")
        assert delay == (d + nt / 2) / nt
        # 4th output: the angle next to the reference symbol, the rotation removed up to half a trial step (2*pi/35/2)
        # (the representative is chosen next to the UNDELAYED reference, corrdelay.m:86-88: equal mod 2*pi otherwise)
        err = np.roll(ang, -d) - np.angle(ref)
        assert np.max(np.abs(np.angle(np.exp(1j * err)))) < math.pi / 35 + 0.25
        assert np.max(np.abs(ang - np.angle(ref))) < math.pi + math.pi / 35 + 0.25
        if d == 0:
            assert np.max(np.abs(err)) < math.pi / 35 + 0.25
        assert abs(rho - 2.0) < 0.05
    with pytest.raises(ValueError, match="wrong flag"):
        px.corrdelay(nrz, bits, nt, nsymb, "angle")
    with pytest.raises(ValueError, match="Nsymb"):
        px.corrdelay(nrz[:-1], bits, nt, nsymb)


def test_mygeteyeinfo_best_instant_and_measured_delay():
    """RxPdmCohQpsk.m:100-215: smooth QPSK currents delayed by a known amount: corrdelay finds the delay, the widest
    eye sits at the symbol centre (best_ts ~ 0, parabolic refinement), a fixed x.ts reads the same table."""
    import polmux_amd as px
    from polmux_amd import synth
    nt, nsymb = 16, 256
    px.reset_all(nsymb, nt, 1)
    pat, _ = synth.pattern_debruijn(nsymb, 1, 4)
    pat2, _ = synth.pattern_debruijn(nsymb, 2, 4)
    cols = []
    d = nt // 2          # (within a symbol: corrdelay's 4th output is unwrapped next to the UNDELAYED reference, :86-88)
    for p_ in (pat, pat2):
        ph = np.array([-0.75, 0.75, -0.25, 0.25])[p_] * math.pi
        e = np.repeat(np.exp(1j * ph), nt)
        e = np.roll(e, -nt // 2)                                         # first symbol centred on sample 1
        h = np.hanning(nt + 1); h /= h.sum()
        e = np.fft.ifft(np.fft.fft(e) * np.fft.fft(np.roll(np.pad(h, (0, e.size - h.size)), -(nt // 2))))
        e = np.roll(e, d)
        cols += [e.real, e.imag]
    irx = np.stack(cols, 1)
    eyeb, best_ts, delay, xopt = px.mygeteyeinfo(irx, np.stack([pat, pat2], 1), None, None)
    np.testing.assert_allclose(delay, d / nt, atol=1.0 / nt)
    assert eyeb.shape == (8,) and abs(best_ts) < 1.5 / nt and 0.9 < np.min(eyeb) <= math.pi / 2 + 2e-3
    eyef, ts, delay_f, xf = px.mygeteyeinfo(irx, np.stack([pat, pat2], 1), delay, 0.0)
    assert ts == 0.0 and xf == nt / 2 and np.all(delay_f == delay)
    assert np.nanmin(eyef) <= np.min(eyeb) + 1e-9                        # the refined optimum is at least as open
    w = px.eye_opening(irx, np.stack([pat, pat2], 1), delay, 0.0)
    assert w == np.nanmin(eyef[eyef < math.pi / 2]) and 0.9 < w < math.pi / 2
    # a delay of several symbols is still measured (modulo the frame)
    big = np.roll(irx, 5 * nt + 3, axis=0)
    np.testing.assert_allclose(px.mygeteyeinfo(big, np.stack([pat, pat2], 1), None, None)[2], (d + 5 * nt + 3) / nt, atol=1e-12)
    # X only
    e1 = px.mygeteyeinfo(irx[:, :2], pat, None, None)
    np.testing.assert_allclose(e1[0], eyeb[:4], atol=1e-12)


def test_myseq_doc_examples():
    """pattern.m:241-242: the two literal examples of the periodic-repetition helper."""
    from polmux_amd import synth
    assert synth.myseq([0, 1, 0], 8).tolist() == [0, 1, 0, 0, 1, 0, 0, 1]
    assert synth.myseq([0, 1, 0, 0, 1, 0], 4).tolist() == [0, 1, 0, 0]
    assert synth.myseq([1, 0], 6).tolist() == [1, 0, 1, 0, 1, 0]

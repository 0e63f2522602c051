"""Pin the CPU oracle's receiver-side restatement (CDE_OFDE, CMA/EASI, carrier
recovery, decisions, MC estimators) with the reference's literal known answers
and stated invariants (SURVEY 8c vii-x)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rng(seed):
    return np.random.default_rng(seed)


# ------------------------------------------------ literal doc-string answers ---
def test_fastshift_literal(oracle):
    """fastshift.m:8-11 and :17-23."""
    x = np.arange(11, dtype=complex)
    np.testing.assert_array_equal(oracle.fastshift(x, 2).real, [9, 10, 0, 1, 2, 3, 4, 5, 6, 7, 8])
    X = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], dtype=complex)
    np.testing.assert_array_equal(oracle.fastshift(X, 2).real, [[4, 5, 6], [7, 8, 9], [1, 2, 3]])
    np.testing.assert_array_equal(oracle.fastshift(X, -1).real, np.roll(X.real, -1, axis=0))


def test_nmod_literal(oracle):
    """nmod.m:8-9, N=8."""
    A = list(range(-2, 11))
    assert [oracle.nmod(a, 8) for a in A] == [6, 7, 8, 1, 2, 3, 4, 5, 6, 7, 8, 1, 2]


# ----------------------------------------------------------------- CDE_OFDE ---
def test_overlap_both_identity_and_checks(oracle):
    """(vii) H == 1 is the identity (CDE_OFDE.m:104-116); argument checks :63-85."""
    x = _rng(1).standard_normal(700) + 1j * _rng(2).standard_normal(700)
    y, rc = oracle.overlap_both_trans(x, np.ones(256), 128)
    assert rc == 0
    np.testing.assert_allclose(y, x, rtol=0, atol=1e-13)
    assert oracle.overlap_both_trans(x, np.ones(255), 128)[1] == 2
    assert oracle.overlap_both_trans(x, np.ones(256), 0)[1] == 3
    assert oracle.overlap_both_trans(x, np.ones(256), 300)[1] == 4
    assert oracle.overlap_both_trans(x[:100], np.ones(256), 128)[1] == 5


def test_overlap_both_is_truncated_linear_convolution(oracle):
    """Away from the ends, overlap-both == zero-padded linear filtering with h = ifft(ifftshift(H))."""
    r = _rng(3)
    N, L = 64, 32
    x = r.standard_normal(512) + 1j * r.standard_normal(512)
    # a short impulse response (|delay| < B/2) so that aliasing inside a block is nil
    h = np.zeros(N, complex)
    h[[0, 1, 2, N - 1, N - 2]] = r.standard_normal(5) + 1j * r.standard_normal(5)
    H = np.fft.fftshift(np.fft.fft(h))
    y, rc = oracle.overlap_both_trans(x, H, L)
    xp = np.concatenate([np.zeros(2, complex), x, np.zeros(2, complex)])
    ref = np.array([sum(h[d % N] * xp[2 + n - d] for d in (-2, -1, 0, 1, 2)) for n in range(512)])
    np.testing.assert_allclose(y, ref, rtol=0, atol=1e-12)


def test_cde_inverts_gvd_fibre(oracle):
    """(vii) CDE_OFDE with D*span equal to the link's dispersion undoes 'g---' propagation
    away from the block edges (CDE_OFDE.m:30-38 sign convention vs fiber.m:308,355)."""
    c = 299792458.0
    lam_nm, D, Lf, R = 1550.0, 17.0, 2e4, 10.0          # ps/nm/km, m, Gbaud
    nsymb, nt = 512, 2
    n = nsymb * nt
    r = _rng(4)
    sym = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, nsymb)))
    u = np.repeat(sym, nt).astype(complex)
    u = np.fft.ifft(np.fft.fft(u) * np.exp(-(np.fft.fftfreq(n, 1 / nt) / 0.8) ** 4))   # band-limit
    fn = np.fft.fftfreq(n, 1.0 / nt)
    omega = 2 * np.pi * R * fn                                                        # rad/ns
    b20 = -lam_nm ** 2 / 2 / np.pi / c * D * 1e-6                                     # fiber.m:308
    betat = (0.5 * omega ** 2 * b20).reshape(n, 1)
    _, _, v = oracle.scalar_ssfm(u, betat, Lf, np.inf, 0.0, 0.0, Lf, [1, 0, 0, 0])
    ox, oy, rc = oracle.cde_ofde(v[:, 0], v[:, 0].conj(), R * nt * 1e9, lam_nm * 1e-9, Lf, D * 1e-6, 0.0, 256, 128)
    assert rc == 0
    np.testing.assert_allclose(ox[200:-200], u[200:-200], rtol=0, atol=2e-3)
    # wrong sign does NOT invert
    bx, _, _ = oracle.cde_ofde(v[:, 0], v[:, 0], R * nt * 1e9, lam_nm * 1e-9, Lf, -D * 1e-6, 0.0, 256, 128)
    assert np.abs(bx[200:-200] - u[200:-200]).max() > 0.1


def test_cde_on_reference_fixture(oracle):
    """LtdeTest.m:1-38 input fixture: fftLength 300 is clipped to the 16 samples (CDE_OFDE.m:24-27);
    result equals the closed form ifft(fft(zero-extended block).*ifftshift(H))."""
    g = json.load(open(os.path.join(GOLD, "ltde_test_input.json")))
    sig = np.array(g["sig"])
    x, y = sig[:, 0] + 1j * sig[:, 1], sig[:, 2] + 1j * sig[:, 3]
    fs = 2 * g["bitrate"] / g["bits_per_symbol"]
    lam = g["c"] / g["fref"]
    ox, oy, rc = oracle.cde_ofde(x, y, fs, lam, g["span"], g["D"], g["S"], g["ntaps"], 8)
    assert rc == 0
    H = oracle.cde_transfer(16, fs, lam, g["span"], g["D"], g["S"])
    fg = fs / 16 * np.arange(-8, 8)
    fc = g["c"] / lam
    np.testing.assert_allclose(H, np.exp(-1j * g["D"] * g["span"] * np.pi * g["c"] / fc ** 2 * fg ** 2), rtol=1e-12)
    xe = np.concatenate([np.zeros(4), x, np.zeros(4)])
    ref = np.concatenate([np.fft.ifft(np.fft.fft(xe[i:i + 16]) * np.fft.ifftshift(H))[4:12] for i in (0, 8)])
    np.testing.assert_allclose(ox, ref, rtol=0, atol=1e-13)


# ----------------------------------------------------------------- CMA / EASI ---
def _rotated_qpsk(L, phi, seed, noise=0.0):
    r = _rng(seed)
    a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (L, 2))))
    M = np.array([[np.cos(phi), np.sin(phi)], [-np.sin(phi), np.cos(phi)]])
    x = a @ M
    if noise:
        x = x + noise * (r.standard_normal((L, 2)) + 1j * r.standard_normal((L, 2)))
    return a, x


def _numpy_cma(xx, h1, h2, taps, mu, R, sps):
    """straight numpy transcription of the recurrence, SURVEY A.5"""
    h1, h2 = h1.copy(), h2.copy()
    L = xx.shape[0] - taps + 1
    y = np.zeros((L, 2), complex)
    k = ((taps - 1) // 2) % 2
    for i in range(L):
        w = xx[i:i + taps]
        y[i, 0] = np.sum(w * h1)
        y[i, 1] = np.sum(w * h2)
        if sps == 1 or i % 2 == k:
            h1 = h1 + mu * (R[0] - abs(y[i, 0]) ** 2) * y[i, 0] * w.conj()
            h2 = h2 + mu * (R[1] - abs(y[i, 1]) ** 2) * y[i, 1] * w.conj()
    return y, h1, h2


@pytest.mark.parametrize("taps,sps", [(1, 1), (3, 1), (7, 1), (7, 2), (15, 2)])
def test_cmafilter_matches_recurrence(oracle, taps, sps):
    _, x = _rotated_qpsk(300, 0.4, taps, noise=0.05)
    r = _rng(taps + 100)
    h1 = 0.3 * (r.standard_normal((taps, 2)) + 1j * r.standard_normal((taps, 2)))
    h2 = 0.3 * (r.standard_normal((taps, 2)) + 1j * r.standard_normal((taps, 2)))
    y, g1, g2 = oracle.cmaadaptivefilter(x, h1, h2, taps, 1e-3, [1.0, 1.2], sps)
    yr, r1, r2 = _numpy_cma(x, h1, h2, taps, 1e-3, [1.0, 1.2], sps)
    np.testing.assert_allclose(y, yr, rtol=0, atol=1e-12)
    np.testing.assert_allclose(g1, r1, rtol=0, atol=1e-12)
    np.testing.assert_allclose(g2, r2, rtol=0, atol=1e-12)


def test_cma_gateway_errors(oracle):
    """cmaadaptivefilter.c:118-119,132."""
    _, x = _rotated_qpsk(16, 0.1, 1)
    with pytest.raises(ValueError, match="ODD INTEGER"):
        oracle.cmaadaptivefilter(x, np.zeros((4, 2)), np.zeros((4, 2)), 4, 1e-3, [1, 1], 1)
    with pytest.raises(ValueError, match="either 1 or 2"):
        oracle.cmaadaptivefilter(x, np.zeros((3, 2)), np.zeros((3, 2)), 3, 1e-3, [1, 1], 3)


def test_cma_fixed_point(oracle):
    """(viii) noise-free rotated QPSK: |y| -> R and h -> inverse rotation; driver stops on the 5e-5 test."""
    phi = 0.3
    a, x = _rotated_qpsk(1024, phi, 5)
    M0 = np.eye(2)
    y, h1, h2, passes = oracle.cmapolardemux(x, M0, 7, 1 / 600, [1.0, 1.0])
    assert 1 < passes < 50 * int(np.ceil(600 / 1024)) * 1 + 50
    np.testing.assert_allclose(np.abs(y), 1.0, atol=2e-3)
    np.testing.assert_allclose(h1[3], [np.cos(phi), np.sin(phi)], atol=2e-3)    # a = x @ M.T
    np.testing.assert_allclose(h2[3], [-np.sin(phi), np.cos(phi)], atol=2e-3)
    np.testing.assert_allclose(y, a, atol=5e-3)


def test_cma_driver_pass_budget(oracle):
    """repetitions = 50*ceil(1/(L*mu)); loop runs while c < repetitions => at most repetitions-1 passes
    (DspPdmCohQpsk.m:175-191)."""
    _, x = _rotated_qpsk(64, 0.5, 6, noise=0.2)
    _, _, _, passes = oracle.cmapolardemux(x, np.eye(2), 3, 1 / 100, [1.0, 1.0])
    assert passes == 50 * int(np.ceil(100 / 64)) - 1
    _, _, _, passes = oracle.easipolardemux(x, np.eye(2), 1 / 100)
    assert passes <= 20 * int(np.ceil(100 / 64)) - 1


def test_easi_updates_real_parts_of_tap0_only(oracle):
    """easiadaptivefilter.c:81-90: only Re(h1[0..1]), Re(h2[0..1]) move; imaginary parts untouched."""
    _, x = _rotated_qpsk(200, 0.2, 7, noise=0.05)
    h1 = np.array([[0.9 + 0.1j, 0.2 - 0.3j]])
    h2 = np.array([[-0.2 + 0.05j, 1.1 + 0.2j]])
    y, g1, g2 = oracle.easiadaptivefilter(x, h1, h2, 1, 1e-2, 1)
    np.testing.assert_array_equal(g1.imag, h1.imag)
    np.testing.assert_array_equal(g2.imag, h2.imag)
    assert np.abs(g1.real - h1.real).max() > 1e-4
    # first output uses the initial taps
    np.testing.assert_allclose(y[0], [x[0] @ h1[0], x[0] @ h2[0]], atol=1e-14)
    # one-step check of the E matrix (errorfun :43-49)
    a, b, mu = y[0, 0].real, y[0, 1].real, 1e-2
    d1, d2 = 1 + mu * (a * a + b * b), 1 + mu * (a * abs(a) + b * abs(b))
    E = np.array([[(a * a - 1) / d1, a * b / d1 + a * b * (a * a - b * b) / d2],
                  [a * b / d1 + a * b * (b * b - a * a) / d2, (b * b - 1) / d1]])
    _, s1, s2 = oracle.easiadaptivefilter(x[:1], h1, h2, 1, mu, 1)
    W = (np.eye(2) - mu * E) @ np.array([h1[0].real, h2[0].real])
    np.testing.assert_allclose(np.array([s1[0].real, s2[0].real]), W, atol=1e-15)


# ------------------------------------------------------------ carrier recovery ---
def test_unwrap_matches_numpy(oracle):
    p = np.cumsum(_rng(8).standard_normal(500) * 1.5)
    w = np.angle(np.exp(1j * p))
    np.testing.assert_allclose(oracle.unwrap(w), np.unwrap(w), atol=1e-12)


def test_vitvit_boxcar_is_causal_circular(oracle):
    """boxcar taps at 0..N-1 => k-sample lag, circular (DspPdmCohQpsk.m:106-110)."""
    L, k = 64, 3
    s = np.exp(1j * 0.1 * np.arange(L)).reshape(L, 1)
    th = oracle.vitvit(s, 4, 4, k, False)
    s4 = s[:, 0] ** 4
    box = np.array([np.mean(s4[(n - np.arange(2 * k + 1)) % L]) for n in range(L)])
    np.testing.assert_allclose(th[:, 0], np.angle(box) / 4, atol=1e-12)
    # N >= L branch (:111-116): tiled signal
    th2 = oracle.vitvit(s, 4, 4, 40, False)
    N = 81
    tiled = np.tile(s4, 2)
    box2 = np.array([np.mean(tiled[(n - np.arange(N)) % (2 * L)]) for n in range(L)])
    np.testing.assert_allclose(th2[:, 0], np.angle(box2) / 4, atol=1e-12)


def test_dsp_recovers_qpsk_with_frequency_offset(oracle):
    """DspPdmCohQpsk body: decimate 1:2:end, /peak, Bell-Labs frequency estimate + V&V phase, +pi/4."""
    L = 1024
    r = _rng(9)
    q = r.integers(0, 4, (L, 2))
    sym = np.exp(1j * (np.pi / 2 * q))            # after +pi/4 the decisions sit mid-quadrant... see below
    dw = 2 * np.pi * 3 / L                        # integer number of cycles: circular
    rot = np.exp(1j * (dw * np.arange(L) + 0.2))[:, None]
    rx = np.zeros((2 * L, 2), complex)
    rx[0::2] = 4 * np.sqrt(2.0) * np.exp(1j * np.pi / 4) * sym * rot
    rx[1::2] = 99.0                                # odd samples are dropped (:12-14)
    p = oracle.dsp_params(power_mw=2.0, applypol=False, freqavg=20, phasavg=3)
    out = oracle.dsp_pdm_coh_qpsk(rx, p)
    assert out.shape == (L, 2)
    np.testing.assert_allclose(np.abs(out), 1.0, atol=1e-9)
    # 4-fold phase ambiguity aside, symbols come back on the (2k+1)pi/4 grid.  The endpoint
    # "circularity" rescaling (:52-55) leaves a slow residual ramp of at most ~dw, tracked by V&V.
    ph = np.angle(out)
    np.testing.assert_allclose(np.abs(np.abs(ph) % (np.pi / 2) - np.pi / 4), 0, atol=0.03)
    d = np.exp(1j * np.angle(out * np.conj(np.exp(1j * np.pi / 4) * sym)))
    assert np.abs(d - d[0, 0]).max() < 0.04


def test_samp2pat_decisions(oracle):
    """samp2pat.m:61-66."""
    ph = np.array([[np.pi / 4, -np.pi / 4], [3 * np.pi / 4, -3 * np.pi / 4], [np.pi / 2, 0.0]])
    pat = oracle.samp2pat_coherent(ph)
    np.testing.assert_array_equal(pat, [[1, 1, 1, 0], [0, 1, 0, 0], [1, 1, 1, 0]])


# ------------------------------------------------------------------ MC estimators ---
def test_erfcinv(oracle):
    from scipy.special import erfcinv
    for y in (1e-12, 1e-3, 0.05, 0.32, 0.9, 1.0, 1.5, 1.95):
        assert oracle.erfcinv(y) == pytest.approx(float(erfcinv(y)), rel=1e-13, abs=1e-15)


def test_ber_estimate_pooled_statistics(oracle):
    """(ix) pooled mean/variance equal the batch values of the concatenated blocks
    (ber_estimate.m:121-127)."""
    r = _rng(11)
    st = oracle.McState()
    blocks = []
    for i in range(7):
        pat = r.integers(0, 2, (64, 4))
        hat = pat ^ (r.random((64, 4)) < 0.1)
        blocks.append((pat != hat).astype(float).ravel())
        cond, avg, nruns, std = oracle.ber_estimate(st, hat, pat, stop=None, nmin=1e9)
        allb = np.concatenate(blocks)
        assert cond[0]
        assert nruns[0] == allb.size
        assert avg[0] == pytest.approx(allb.mean(), rel=1e-13)
        assert std[0] == pytest.approx(np.sqrt(allb.var(ddof=1) / allb.size), rel=1e-12)


def test_ber_estimate_stop_rules_and_state_reset(oracle):
    """ber_estimate.m:128-141: nmin-only rule and the Gaussian-confidence rule; state clears when all cond false."""
    st = oracle.McState()
    pat = np.zeros((100, 2), int)
    hat = pat.copy(); hat[:3, 0] = 1                # 3 errors / 200 bits per block
    n = 0
    cond = [True]
    while cond[0]:
        cond, avg, nruns, std = oracle.ber_estimate(st, hat, pat, stop=None, nmin=10)
        n += 1
    assert n == 4 and avg[0] == pytest.approx(0.015)     # avgber*n*M = 12 > 10 at n = 4
    assert st.first == 0                                 # persistent state cleared
    cond, avg, nruns, std = oracle.ber_estimate(st, hat, pat, stop=None, nmin=10)
    assert nruns[0] == 200 and cond[0]
    st = oracle.McState()
    r = _rng(12)
    k = 0
    cond = [True]
    while cond[0] and k < 10000:
        hat = (r.random((100, 2)) < 0.05).astype(int)
        cond, avg, nruns, std = oracle.ber_estimate(st, hat, pat, stop=(0.1, 95), nmin=1)
        k += 1
    eps = np.sqrt(2) * oracle.erfcinv(1 - 0.95)
    assert not cond[0] and eps * std[0] < 0.1 * avg[0]
    assert abs(avg[0] - 0.05) < 3 * 0.1 * 0.05


def test_mc_estimate_vector_mode_and_limits(oracle):
    """mc_estimate.m:160-203 with nind; (x) variance limits bracket the estimate."""
    r = _rng(13)
    st = oracle.McState()
    data = {1: [], 2: []}
    for it in range(5):
        for nind in (1, 2):
            s = r.standard_normal(40) * nind + nind
            data[nind].append(s)
            cond, out = oracle.mc_estimate(st, s, stop=(1e-9, 68), nmin=50, dim=2, nind=nind)
    for nind in (1, 2):
        allb = np.concatenate(data[nind])
        assert out["mean"][nind - 1] == pytest.approx(allb.mean(), rel=1e-12)
        assert out["var"][nind - 1] == pytest.approx(allb.var(ddof=1), rel=1e-12)
        assert out["nruns"][nind - 1] == allb.size
        lo, hi = out["varlim"][:, nind - 1]
        assert lo < out["var"][nind - 1] < hi
    assert cond.all()


# ============================================================ the .m twins ===
def test_cma_mfile_twin_is_the_c_filter_at_one_sample_per_symbol_and_differs_at_two(oracle):
    """cmaadaptivefilter.m:52-72 updates at EVERY sample: with sps = 1 it is the C filter (same recurrence, MATLAB's
    column-then-row summation order: 1e-13), with sps = 2 the C filter skips every other update (cmaadaptivefilter.c:64,85)
    and the twins part ways.  The twin returns the updated taps and leaves its inputs alone."""
    r = np.random.default_rng(11)
    L, taps = 300, 5
    a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (L + taps - 1, 2))))
    xx = a @ np.array([[np.cos(0.4), np.sin(0.4)], [-np.sin(0.4), np.cos(0.4)]])
    h1 = np.zeros((taps, 2), complex); h1[2, 0] = 1
    h2 = np.zeros((taps, 2), complex); h2[2, 1] = 1
    k1, k2 = h1.copy(), h2.copy()
    y, g1, g2 = oracle.cmaadaptivefilter_m(xx, h1, h2, taps, 2e-3, [1.0, 1.0])
    np.testing.assert_array_equal(h1, k1); np.testing.assert_array_equal(h2, k2)
    yc, c1, c2 = oracle.cmaadaptivefilter(xx, h1.copy(order="F"), h2.copy(order="F"), taps, 2e-3, [1.0, 1.0], 1)
    np.testing.assert_allclose(y, yc, atol=1e-13); np.testing.assert_allclose(g1, c1, atol=1e-13); np.testing.assert_allclose(g2, c2, atol=1e-13)
    y2, d1, d2 = oracle.cmaadaptivefilter(xx, h1.copy(order="F"), h2.copy(order="F"), taps, 2e-3, [1.0, 1.0], 2)
    assert np.abs(d1 - g1).max() > 1e-4                      # half the updates are missing in the C filter at sps = 2
    # an even number of taps is fine for the twin (the odd check is in the C gateway only, cmaadaptivefilter.c:118-119)
    y4, _, _ = oracle.cmaadaptivefilter_m(xx, h1[:4], h2[:4], 4, 2e-3, [1.0, 1.0])
    assert y4.shape == (L + taps - 1 - 4 + 1, 2)


def test_easi_mfile_twin_reduces_to_the_c_filter_on_real_data_and_is_complex_otherwise(oracle):
    """easiadaptivefilter.m:51-84 uses complex a, b and recombines all taps; easiadaptivefilter.c:81-90 uses Re(y) and
    touches the real parts of tap 0.  On REAL inputs with one REAL tap the two coincide exactly (every imaginary part is
    zero) -- that pins the twin's restatement to the C one; on complex inputs they differ ("the twins are
    not equivalent").  First update checked against the formula written out."""
    r = np.random.default_rng(12)
    L = 200
    xr = r.standard_normal((L, 2))
    h1 = np.array([[0.9, 0.1]], complex); h2 = np.array([[-0.1, 0.9]], complex)
    y, g1, g2 = oracle.easiadaptivefilter_m(xr, h1, h2, 1, 1e-3)
    yc, c1, c2 = oracle.easiadaptivefilter(xr.astype(complex), h1.copy(order="F"), h2.copy(order="F"), 1, 1e-3, 1)
    np.testing.assert_array_equal(y, yc); np.testing.assert_array_equal(g1, c1); np.testing.assert_array_equal(g2, c2)
    xc = xr + 1j * r.standard_normal((L, 2))
    y, g1, g2 = oracle.easiadaptivefilter_m(xc, h1, h2, 1, 1e-3)
    yc, c1, c2 = oracle.easiadaptivefilter(xc, h1.copy(order="F"), h2.copy(order="F"), 1, 1e-3, 1)
    assert np.abs(g1 - c1).max() > 1e-4 and np.abs(g1.imag).max() > 1e-4
    # one sample by hand (errorfun :78-84, update :58-66)
    mu = 1e-3
    a = xc[0] @ h1[0]; b = xc[0] @ h2[0]
    d1 = 1 + mu * (abs(a) ** 2 + abs(b) ** 2); d2 = 1 + mu * (a * abs(a) + b * abs(b))
    E = np.array([[(abs(a) ** 2 - 1) / d1, a * b / d1 + a * b * (abs(a) ** 2 - abs(b) ** 2) / d2],
                  [a * b / d1 + a * b * (abs(b) ** 2 - abs(a) ** 2) / d2, (abs(b) ** 2 - 1) / d1]])
    n1 = (1 - mu * E[0, 0]) * h1 + (-mu * E[0, 1]) * h2
    n2 = (-mu * E[1, 0]) * h1 + (1 - mu * E[1, 1]) * h2
    y1, f1, f2 = oracle.easiadaptivefilter_m(xc[:1], h1, h2, 1, mu)
    np.testing.assert_allclose(y1[0], [a, b], atol=1e-15)
    np.testing.assert_allclose(f1, n1, atol=1e-15); np.testing.assert_allclose(f2, n2, atol=1e-15)


def test_params_mat_and_mfile_twin_driver(oracle):
    """cmapolardemux / easipolardemux start from params.mat when it is given (DspPdmCohQpsk.m:148-149, :201-202), whatever
    txpolars says; with the .m twin of the EASI filter the driver takes the returned taps (:232-235)."""
    r = np.random.default_rng(13)
    L = 256
    a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (L, 2))))
    th = 0.5
    Rm = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
    x = (a @ Rm) * 4.0                                           # DspPdmCohQpsk divides by 4 sqrt(P), P = 1
    x2 = np.repeat(x, 2, axis=0)                                 # 2 samples per symbol
    base = dict(power_mw=1.0, applypol=True, polmethod="cma", cma_mu=1 / 500, cma_taps=3, freqavg=0, phasavg=0)
    s0 = oracle.dsp_pdm_coh_qpsk(x2, oracle.dsp_params(**base))
    # y_r = sum_p x_p h_r(p) with h_r = M(r,:) (:160-167): M = Rm undoes x = a*Rm exactly (Rm*Rm.' = I)
    s1 = oracle.dsp_pdm_coh_qpsk(x2, oracle.dsp_params(cma_mat=Rm, **base))
    assert np.abs(np.abs(s1) - 1).max() < 1e-12                  # already demultiplexed: the CMA has nothing to do
    assert 1e-6 < np.abs(np.abs(s0) - 1).max() < 0.2             # from phizero = 0 it has to converge first
    e = dict(base, polmethod="easi", easi_mu=1 / 500)
    t0 = oracle.dsp_pdm_coh_qpsk(x2, oracle.dsp_params(**e))
    t1 = oracle.dsp_pdm_coh_qpsk(x2, oracle.dsp_params(mfile_twins=True, **e))
    assert np.abs(t0 - t1).max() > 1e-3                          # the C filter and its .m twin separate differently
    y, h1, h2, n = oracle.easipolardemux_m(x / 4.0, np.eye(2), 1 / 500)
    assert n >= 1 and np.abs(h1.imag).max() + np.abs(h2.imag).max() > 0

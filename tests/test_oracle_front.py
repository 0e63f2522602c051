"""CPU tests: the front-end oracle (oracle/front.py) against the analytic identities the reference states or
implies, and the host tables of polmux_amd/rxfront.py (myfilter.m, evaldelay.m, receiver_cohmix.m:97-227)."""
import math

import numpy as np
import pytest

from oracle import front
from polmux_amd import rxfront, synth


def _field(nsymb=64, nt=16):
    ux, uy, _, _ = synth.pdm_qpsk_field(nsymb, nt, 2.0, 2, 3)
    return ux, uy, synth.fn_grid(nsymb, nt)


def test_balanced_hybrid_is_four_times_s_conj_lo():
    """receiver_cohmix.m:254-279: I1-I2 = 4 Re(s Elo*), I3-I4 = 4 Im(s Elo*); unit filters leave the field alone."""
    ux, uy, fn = _field()
    n = ux.size
    one = np.ones(n)
    for elo in (1.0, 10 ** (3 / 20) * np.exp(1j * np.linspace(0, 2, n))):
        i = front.receiver_cohmix(ux, uy, one, elo, one, True)
        assert i.shape == (n, 4)
        zx, zy = i[:, 0] + 1j * i[:, 1], i[:, 2] + 1j * i[:, 3]
        np.testing.assert_allclose(zx, 4 * ux * np.conj(elo), atol=1e-12 * np.abs(ux).max() * 4 * np.abs(elo).max())
        np.testing.assert_allclose(zy, 4 * uy * np.conj(elo), atol=1e-12 * np.abs(uy).max() * 4 * np.abs(elo).max())
    i = front.receiver_cohmix(ux, None, one, 1.0, one, True)
    assert i.shape == (n, 2)


def test_single_photodiodes():
    """x.pdtype == 'normal' keeps fields 1 and 3: |s + Elo|^2 and |j s - Elo|^2 (:262-265, :279)."""
    ux, uy, fn = _field()
    one = np.ones(ux.size)
    i = front.receiver_cohmix(ux, None, one, 0.5, one, False)
    np.testing.assert_allclose(i[:, 0], np.abs(ux + 0.5) ** 2, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(i[:, 1], np.abs(1j * ux - 0.5) ** 2, rtol=1e-11, atol=1e-13)


def test_post_compensation_undoes_linear_fibre():
    """x.dpost: Hf = fastexp(-betat) (:149-165) with dpost = -D L inverts a 'g---' span (fiber.m:762-773)."""
    ux, uy, fn = _field()
    omega = 2 * np.pi * 28 * fn
    beta = 0.5 * omega ** 2 * (-21.67e-6) * 8e1
    prop = np.fft.ifft(np.fft.fft(ux) * np.exp(-1j * beta))
    i = front.receiver_cohmix(prop, None, np.exp(+1j * beta), 1.0, np.ones(ux.size), True)
    np.testing.assert_allclose(i[:, 0] + 1j * i[:, 1], 4 * ux, atol=1e-11 * np.abs(ux).max())


def test_lowpass_acts_on_each_current_as_a_real_filter():
    """:300 real(ifft(fft(I).*Hf)): for a Hermitian-symmetric Hf (gauss) the currents stay exactly real-filtered,
    and a complex (bessel5) Hf is applied through its real-impulse-response part."""
    ux, uy, fn = _field()
    n = ux.size
    one = np.ones(n)
    raw = front.receiver_cohmix(ux, None, one, 1.0, one, True)
    for ft in ("gauss", "bessel5"):
        h = rxfront.myfilter(ft, fn, 0.65)
        got = front.receiver_cohmix(ux, None, one, 1.0, h, True)
        he = 0.5 * (h + np.conj(h[(-np.arange(n)) % n]))
        z = np.fft.ifft(np.fft.fft(raw[:, 0] + 1j * raw[:, 1]) * he)
        np.testing.assert_allclose(got[:, 0] + 1j * got[:, 1], z, atol=1e-12 * np.abs(z).max())


def test_adc_levels_and_error_bound():
    """RxPdmCohQpsk.m:36-40: 2^bits+1 levels spanning [-M, M], error <= half a step, extremes preserved."""
    r = np.random.default_rng(1)
    x = r.standard_normal((500, 4))
    M = np.abs(x).max()
    for bits in (1, 3, 5, 8):
        q = front.adc(x, bits)
        step = 2 * M / 2 ** bits
        k = (q + M) / step
        np.testing.assert_allclose(k, np.round(k), atol=1e-9)
        assert np.abs(q - x).max() <= step / 2 * (1 + 1e-12)
        assert np.abs(q).max() == pytest.approx(M, rel=1e-15)
    # round() is half away from zero: a value exactly between two levels goes up
    y = np.array([[-1.0, 1.0, 0.25, -0.25]])
    np.testing.assert_allclose(front.adc(y, 2), [[-1.0, 1.0, 0.5, 0.0]])


def test_decimate_definition_properties():
    """The project's decimate: unit DC gain, ramps preserved through both edges (odd reflection), ceil(n/r)
    outputs centred on samples 0, r, 2r, ..., a tone in the pass-band scaled by |B(f)|."""
    for n, r in ((1024, 16), (1000, 7), (2048, 32), (100, 3)):
        b = rxfront.fir1_lowpass(16, 1.0 / r)
        assert b.size == 17 and b.sum() == pytest.approx(1.0, abs=1e-15)
        np.testing.assert_allclose(b, b[::-1], atol=1e-17)
        nout = -(-n // r)
        np.testing.assert_allclose(front.decimate_fir(np.full(n, 2.5), r, b), np.full(nout, 2.5), rtol=1e-14)
        ramp = 0.1 * np.arange(n) - 3
        np.testing.assert_allclose(front.decimate_fir(ramp, r, b), ramp[::r], rtol=1e-12, atol=1e-12)
    n, r = 4096, 16
    b = rxfront.fir1_lowpass(16, 1.0 / r)
    f0 = 5 / n
    tone = np.cos(2 * np.pi * f0 * np.arange(n))
    gain = np.abs(np.sum(b * np.exp(-2j * np.pi * f0 * np.arange(17))))
    y = front.decimate_fir(tone, r, b)
    np.testing.assert_allclose(y[2:-2], gain * tone[::r][2:-2], atol=1e-12)


def test_rx_front_recombination_and_shift():
    """RxPdmCohQpsk.m:42-44, :63-72: pairs (1,2) and (3,4) shifted by their own delay, complex(I,Q) per polarisation."""
    r = np.random.default_rng(2)
    cur = r.standard_normal((64, 4))
    out = front.rx_front(cur, True, 0, [3, -5], 1, None)
    np.testing.assert_array_equal(out[:, 0], np.roll(cur[:, 0] + 1j * cur[:, 1], 3))
    np.testing.assert_array_equal(out[:, 1], np.roll(cur[:, 2] + 1j * cur[:, 3], -5))
    assert front.rx_front(cur, False, 0, [0], 1, None).shape == (64, 1)


# ------------------------------------------------------------------ host tables ---
def test_myfilter_three_db_points_and_symmetries():
    f = np.array([-1.0, 0.0, 1.0]) * 0.7
    for ft in ("gauss", "butt2", "butt4", "butt6", "rc1", "rc2"):
        h = rxfront.myfilter(ft, f, 0.7)
        np.testing.assert_allclose(np.abs(h) ** 2, [0.5, 1.0, 0.5], rtol=2e-15 if ft == "gauss" else 1e-12)
        assert np.conj(h[0]) == pytest.approx(h[2])                 # real impulse response
    np.testing.assert_allclose(np.abs(rxfront.myfilter("bessel5", f, 0.7)) ** 2, [0.5, 1.0, 0.5], rtol=2e-4)  # Bb = 0.3863
    np.testing.assert_allclose(rxfront.myfilter("supergauss", f, 0.7, 3), [2 ** -0.5, 1, 2 ** -0.5], rtol=1e-15)
    np.testing.assert_array_equal(rxfront.myfilter("ideal", np.array([0.69, 0.71]), 0.7), [1.0, 0.0])
    assert rxfront.myfilter("movavg", np.array([0.7]), 0.7)[0] == pytest.approx(0.0, abs=1e-16)
    np.testing.assert_allclose(rxfront.myfilter("gauss_off", np.array([0.2]), 0.7, 0.2), [1.0])
    with pytest.raises(ValueError, match="does not exist"):
        rxfront.myfilter("nope", f, 1.0)
    with pytest.raises(ValueError, match="missing superGauss order"):
        rxfront.myfilter("supergauss", f, 1.0)


def test_evaldelay_matches_filter_group_delay():
    """evaldelay.m gives the low-frequency group delay (in symbols) of the same responses."""
    df = 1e-4
    # ('rc2' is left out: evaldelay.m states (sqrt(2)-1)/(pi bw), not the DC group delay of myfilter's rc2; mirrored as is)
    for ft in ("bessel5", "butt2", "butt4", "butt6", "rc1", "gauss"):
        h = rxfront.myfilter(ft, np.array([-df, df]), 0.65)
        gd = -(np.angle(h[1]) - np.angle(h[0])) / (2 * np.pi * 2 * df)
        want = rxfront.evaldelay(ft, 0.65)
        assert gd == pytest.approx(want, rel=0.12, abs=1e-9)        # the 1.1x factors of evaldelay.m are empirical
    assert rxfront.evaldelay("bessel5", 0.65) == 0.3863 / 0.65
    assert rxfront.evaldelay("rc2", 0.65) == (math.sqrt(2) - 1) / (math.pi * 0.65)


def test_front_tables_follow_receiver_cohmix():
    """_front_tables: post-compensation phase (:149-165), LO detuning/phase-noise/power (:193-227), filters (:169, :296)."""
    import polmux_amd as px
    from polmux_amd.gstate import GSTATE
    nsymb, nt = 64, 16
    px.reset_all(nsymb, nt, 1)
    GSTATE.SYMBOLRATE = 28.0
    px.lasersource(1.0, 1550.0)
    GSTATE.FIELDX = np.zeros((1, nsymb * nt), dtype=complex)      # only .shape is read
    x = dict(oftype="gauss", obw=1.9, oord=3, eftype="bessel5", ebw=0.65, eord=4, lopower=3.0)
    hopt, elo, hel, pd, b2b = rxfront._front_tables(1, x)
    assert elo == pytest.approx(10 ** 0.15) and pd == 0.0 and not b2b
    np.testing.assert_allclose(hopt, rxfront.myfilter("gauss", GSTATE.FN, 0.95))
    np.testing.assert_allclose(hel, rxfront.myfilter("bessel5", GSTATE.FN, 0.65))
    x2 = dict(x, dpost=-1360.0, slopez=0.0, lodetuning=3.1 * 28e9 / nsymb, lophasenoise=np.linspace(0, 1, nsymb * nt))
    x2["lambda"] = 1550.0
    hopt, elo, hel, pd, b2b = rxfront._front_tables(1, x2)
    omega = 2 * np.pi * 28 * GSTATE.FN
    b20z = -1550.0 ** 2 / 2 / np.pi / 299792458.0 * -1360.0 * 1e-3
    b30z = (1550.0 / 2 / np.pi / 299792458.0) ** 2 * (2 * 1550.0 * -1360.0) * 1e-3     # :150-151 with slopez = 0
    np.testing.assert_allclose(hopt, np.exp(-1j * (0.5 * omega ** 2 * b20z + omega ** 3 * b30z / 6)) * rxfront.myfilter("gauss", GSTATE.FN, 0.95), atol=1e-12)
    n = nsymb * nt
    np.testing.assert_allclose(elo, 10 ** 0.15 * np.exp(1j * (2 * np.pi * 3 / n * np.arange(1, n + 1) + np.linspace(0, 1, n))), atol=1e-12)
    with pytest.raises(ValueError, match="Incompatible vector"):
        rxfront._front_tables(1, dict(x, lophasenoise=np.zeros(5)))
    with pytest.raises(ValueError, match="b2b"):
        rxfront._front_tables(1, dict(x, b2b="yes"))
    assert rxfront._front_tables(1, dict(x2, b2b="b2b"))[3] == 0.0      # b2b removes dpost (:135)
    GSTATE.DELAY = np.array([[0.25], [0.75]])
    assert rxfront.theory_delay(1, x, True, 0.1) == pytest.approx(0.5 + 0.0 + 0.3863 / 0.65 + 0.1)
    assert rxfront.theory_delay(1, x, False, 0.0) == pytest.approx(0.25 + 0.3863 / 0.65)
    assert rxfront.theory_delay(1, dict(x, b2b="b2b"), True, 0.0) == pytest.approx(0.3863 / 0.65)
    assert rxfront._mround(2.5) == 3 and rxfront._mround(-2.5) == -3 and rxfront._mround(-0.4) == 0


def test_unique_field_channel_selection_folds_into_the_tables():
    """receiver_cohmix.m:104-107,184: sigx = sigx(nind) on a 'unique' WDM field.  The device path leaves the field alone
    and moves the optical filter table and the LO instead (rxfront._front_tables): same currents as the literal form."""
    import polmux_amd as px
    from polmux_amd.gstate import GSTATE, unique_field_shifts
    nsymb, nt, nch = 64, 32, 3
    px.reset_all(nsymb, nt, nch)
    GSTATE.SYMBOLRATE = 10.0
    px.lasersource(1.0, 1550.0, 0.4)
    nd = unique_field_shifts()
    assert nd[1] == 0 and nd[0] == -nd[2] and nd[0] < 0              # 0.4 nm = ~50 GHz = 5 symbol rates = ~320 bins; ch 1 = shortest lambda
    assert abs(-nd[0] - round(299792458.0 * (1 / 1549.6 - 1 / 1550.0) / 10.0 * nsymb)) <= 1
    cols = [synth.pdm_qpsk_field(nsymb, nt, 1.0, 2 + k, 5 + k)[0] for k in range(nch)]
    z = sum(np.roll(np.fft.fft(c), -int(nd[k])) for k, c in enumerate(cols))           # create_field.m:186-189
    field = np.fft.ifft(z)
    GSTATE.FIELDX = np.zeros((1, nsymb * nt), dtype=complex)                          # unique: one row, NCH = 3
    x = dict(oftype="gauss", obw=1.9, eftype="bessel5", ebw=0.65, lopower=0.0)
    for ich in (1, 2, 3):
        hopt, elo, hel, pd, _ = rxfront._front_tables(ich, x)
        got = front.receiver_cohmix(field, None, hopt, elo, hel, True)                 # what the device computes
        fn = GSTATE.FN
        lit = front.receiver_cohmix(field, None, rxfront.myfilter("gauss", fn, 0.95), 1.0, hel, True, ndfn=int(nd[ich - 1]))
        np.testing.assert_allclose(got, lit, atol=1e-11 * np.abs(lit).max())
        # and the selected channel is the transmitted one, up to the in-band sidelobes of its neighbours 5 symbol rates
        # away (almost square pulses: about -20 dB)
        clean = front.receiver_cohmix(cols[ich - 1], None, rxfront.myfilter("gauss", fn, 0.95), 1.0, hel, True)
        assert np.abs(got - clean).max() < 0.2 * np.abs(clean).max()
    # a single channel in the unique field comes back exactly
    GSTATE.FIELDX = np.zeros((1, nsymb * nt), dtype=complex)
    alone = np.fft.ifft(np.roll(np.fft.fft(cols[0]), -int(nd[0])))
    hopt, elo, hel, pd, _ = rxfront._front_tables(1, x)
    got = front.receiver_cohmix(alone, None, hopt, elo, hel, True)
    clean = front.receiver_cohmix(cols[0], None, rxfront.myfilter("gauss", GSTATE.FN, 0.95), 1.0, hel, True)
    np.testing.assert_allclose(got, clean, atol=1e-11 * np.abs(clean).max())

"""BASELINE.json's configurations AS STATED (SURVEY 8d "Concrete configs"), on the MI355X through the C ABI, against
the CPU oracle on the same inputs.  test_gpu_parity.py holds the component tests; these are the whole configurations:

  C0  32 Gbaud single-polarisation QPSK, 1 x 80 km linear span 'g---', post-compensation x.dpost in the receiver
  C1  Run_my_PDM_QPSK frame (2^16 dual-pol, 80 km 'g-s-', CDE 256/128, CMA 7 taps at mu = 1/6000, CPE)
  C3  Monte-Carlo realisations of the C1 frame: 'gps-' with 100 random waveplates per realisation + amplifier ASE
  C4  2^20-sample dual-pol frame, spans with in-line amplifiers, two launch powers of the ladder
(C2, the 16-channel WDM chain, is test_wdm_16ch_multispan_chain_vs_oracle_c2 in test_gpu_parity.py.)

Bars: optical field <= 1e-9 relative (stated bar 1e-6), ncycle identical, decisions bit-exact.
"""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELD_RTOL = 1e-9


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from polmux_amd import _abi
    b = _abi.get()
    assert b.path.endswith("polmux_amd/lib/libpolmux_hip.so")
    return b


def _sync():
    import torch
    torch.cuda.synchronize()


def _rx_oracle(oracle, cfg, hp, ox, oy):
    """front end (2-sps pick, or receiver_cohmix + timing + decimate) -> CDE_OFDE -> DspPdmCohQpsk on the oracle;
    returns (symbols [nsymb x 2], decided bits)"""
    if hp.front is not None:
        from oracle import front
        t = hp.front_tables
        cur = front.receiver_cohmix(ox, oy, t["hopt"], t["elo"], t["hel"], True)
        rx = front.rx_front(cur, True, cfg.adcbits, hp.front_shifts, t["decim"], t["fir"])
    else:
        half = cfg.nt // 2
        rx = np.stack([ox[::half], oy[::half]], 1) * hp.rx_scale
    ex, ey, _ = oracle.cde_ofde(rx[:, 0], rx[:, 1], 2 * cfg.symbolrate * 1e9, cfg.lam * 1e-9, cfg.length * cfg.nspans,
                                cfg.disp * 1e-6, 0.0, cfg.fft_length, cfg.cde_L)
    op = oracle.dsp_params(power_mw=hp.power_mw, applypol=True, polmethod="cma", cma_mu=cfg.cma_mu, cma_taps=cfg.cma_taps,
                           freqavg=cfg.freqavg, phasavg=cfg.phasavg, poworder=cfg.poworder)
    ref = oracle.dsp_pdm_coh_qpsk(np.stack([ex, ey], 1), op)
    return ref, oracle.samp2pat_coherent(np.angle(ref))


def _screened_equal(got_bits, ref_sym, want_bits, tol=1e-10):
    """decisions bit-exact, except symbols whose phase sits within tol rad of a decision boundary (there the two sides'
    1e-12 symbol difference may legitimately flip the bit: none observed)"""
    ph = np.angle(ref_sym)
    near = (np.abs(np.abs(ph) - np.pi / 2) < tol) | (np.abs(ph) < tol) | (np.abs(np.abs(ph) - np.pi) < tol)    # [nsymb x 2]
    mask = np.repeat(~near, 2, axis=1)
    np.testing.assert_array_equal(got_bits[mask], want_bits[mask])
    assert near.mean() < 1e-3


# ================================================================================ C1 ===
def test_c1_end_to_end_at_the_stated_cma_step(lib, oracle):
    """BASELINE config[1] with EVERY stated parameter (CMA taps 7, mu = 1/6000, R = [1 1], L = 1024 symbols,
    CDE 256/128, freqavg 500, phasavg 3, poworder 2): the CMA driver runs its full pass budget here (50*ceil(1/(L*mu)) - 1
    = 299 passes unless the 5e-5 test stops it, DspPdmCohQpsk.m:175-191).  Symbols 1e-7, decisions identical."""
    from polmux_amd import pipeline
    cfg = pipeline.HotPathConfig()                       # the defaults ARE config[1]
    assert cfg.cma_mu == 1 / 6000 and cfg.cma_taps == 7 and cfg.nfft == 65536 and cfg.flag == "g-s-"
    hp = pipeline.HotPath(cfg, max_frames=2)
    ux, uy = hp.make_batch(2)
    err = hp.run(ux, uy)
    _sync()
    gam, betat, db1 = hp._keep
    rc, fd, nc, ox, oy = oracle.matrix_ssfm(hp.tx_host[0], hp.tx_host[1], betat, db1, cfg.dzmax, cfg.dphimax, gam, hp.alphalin,
                                            cfg.length, 1, 0, hp.fls, [0.0], [0.0], [0.0])
    assert hp.last_ncycle(2).tolist() == [nc, nc]
    assert np.abs(ux[1].cpu().numpy() - ox[:, 0]).max() <= FIELD_RTOL * np.abs(ox).max()
    assert np.abs(uy[0].cpu().numpy() - oy[:, 0]).max() <= FIELD_RTOL * np.abs(oy).max()
    ref, want = _rx_oracle(oracle, cfg, hp, ox[:, 0], oy[:, 0])
    sym = hp.sym[0].cpu().numpy().T
    np.testing.assert_allclose(sym, ref, atol=1e-11)
    e = [int((want[:, :2] != hp.bits[:, :2]).sum()), int((want[:, 2:] != hp.bits[:, 2:]).sum())]
    assert err.cpu().numpy()[0].tolist() == e
    assert int(hp.errors_min_over_rotations(2).sum()) == 0      # noise-free span: error-free once the pi/2 ambiguity is resolved
    hp.close()


# ================================================================================ C3 ===
@pytest.mark.parametrize("mu,sampled", [(1 / 600, (0, 5, 11, 15)), (1 / 6000, (0, 2, 4, 7, 9, 11, 13, 15))])
def test_c3_monte_carlo_realisations_pmd_and_ase_vs_oracle(lib, oracle, mu, sampled):
    """BASELINE config[3] (ex24_pmd.m-style PMD + ex20_coherent_polmux.m:132-175's noisy amplifier): McCampaign at the
    C1 frame (1024 x 64), fiber('gps-') with 100 waveplates drawn per realisation (fiber.m:260-276, set_random_pmd),
    then ampliflat(Gerbio,'gain',{f, noise}) with the ASE INJECTED (options.noise, ampliflat.m:123-129) so that the
    oracle sees the same noise, then the reference's own front end (receiver_cohmix gauss 1.9 / bessel5 0.65 + timing +
    decimate: the white ASE of the 1.8 THz simulation band must be filtered; no ADC, as in ex20) -> CDE_OFDE -> CMA + CPE.
    Sampled realisations of a 16-realisation batch -- four at mu = 1/600, eight at the mu = 1/6000 that config[1] states (the
    CMA's full 299-pass budget on every noise-loaded realisation) -- are compared with oracle.matrix_ssfm + amplifier + receiver
    chain: field 1e-9, ncycle, symbols, error counts."""
    from polmux_amd import pipeline
    from polmux_amd.ampliflat import ase_sigma
    cfg = pipeline.HotPathConfig(flag="gps-", nplates=100, dgd=0.1, rx_amp=True, span_nf_db=31.0, cma_mu=mu,
                                 frontend="cohmix", adcbits=0)
    n = cfg.nfft

    def noise_of(r):
        g = np.random.default_rng(777000 + int(r))
        return g.standard_normal((2, n)) + 1j * g.standard_normal((2, n))

    camp = pipeline.McCampaign(cfg, frames_per_call=16, noise_provider=lambda idx: np.stack([noise_of(r) for r in idx]))
    hp = camp.hp
    idx = list(range(40, 56))
    kept = {}

    def keep(i0, ids, ux, uy):
        _sync()
        kept["x"], kept["y"] = ux.cpu().numpy(), uy.cpu().numpy()
        kept["nc"] = hp.last_ncycle(len(ids))
    errs = camp.simulate(idx, keep=keep)
    _sync()
    sym_dev = hp.sym[:16].cpu().numpy()
    gam, betat, db1 = hp._keep
    gain = math.exp(hp.alphalin * cfg.length)
    sigma = float(ase_sigma(cfg.span_nf_db, gain, 1)[0])
    ncs = set()
    for pos in sampled:
        r = idx[pos]
        db0, th, ep = (a[0] for a in hp.set_random_pmd([r]))           # the draw of realisation r (keyed by r alone)
        rc, fd, nc, ox, oy = oracle.matrix_ssfm(hp.tx_host[0], hp.tx_host[1], betat, db1, cfg.dzmax, cfg.dphimax, gam, hp.alphalin,
                                                cfg.length, cfg.nplates, 0, hp.fls, db0, th, ep)
        assert rc == 0 and kept["nc"][pos] == nc
        ncs.add(nc)
        nz = noise_of(r)
        ox = math.sqrt(gain) * ox[:, 0] + sigma * nz[0]                # ampliflat.m:78-82, 123-136
        oy = math.sqrt(gain) * oy[:, 0] + sigma * nz[1]
        sc = max(np.abs(ox).max(), np.abs(oy).max())
        assert np.abs(kept["x"][pos] - ox).max() <= FIELD_RTOL * sc
        assert np.abs(kept["y"][pos] - oy).max() <= FIELD_RTOL * sc
        ref, want = _rx_oracle(oracle, cfg, hp, ox, oy)
        np.testing.assert_allclose(sym_dev[pos].T, ref, atol=1e-10)
        # the campaign's count = errors after pol-swap / pi/2 resolution (ex20:160-173): same resolution on the oracle's symbols
        best = None
        for swap in (False, True):
            tx = hp.bits if not swap else np.concatenate([hp.bits[:, 2:], hp.bits[:, :2]], 1)
            tot = 0
            for pol in (0, 1):
                tot += min(int((oracle.samp2pat_coherent(np.angle(ref[:, pol:pol + 1] * 1j ** k)) != tx[:, 2 * pol:2 * pol + 2]).sum())
                           for k in range(4))
            best = tot if best is None else min(best, tot)
        assert int(errs[pos]) == best
    assert 0 < errs.sum() < 16 * camp.bits_per_realisation // 8       # noise-loaded but locked
    # the device generator (Philox keyed by the realisation index) drives the same chain when nothing is injected
    camp.noise_provider = None
    e2 = camp.simulate(idx[:4])
    assert 0 <= e2.min() and e2.max() < camp.bits_per_realisation // 4
    camp.close()


# ================================================================================ C4 ===
def test_c4_2pow20_frame_two_spans_two_powers_vs_oracle(lib, oracle):
    """BASELINE config[4] at FULL frame size: 2^20-sample dual-pol frames (Nsymb 16384 x Nt 64), two 80 km 'g-s-' spans
    with the in-line amplifier between them, at two points of the launch-power ladder (-4 dBm and +5 dBm: different step
    counts in one batch), against the oracle loop (about a minute of CPU)."""
    from polmux_amd import pipeline
    cfg = pipeline.HotPathConfig(nsymb=16384, nt=64, nspans=2)
    assert cfg.nfft == 1 << 20
    hp = pipeline.HotPath(cfg, max_frames=2)
    p_mw = np.array([10 ** (-4 / 10), 10 ** (5 / 10)])
    scale = p_mw / cfg.pavg_mw
    ux, uy = hp.make_batch(2, scale)
    hp.fibre(ux, uy)
    _sync()
    gx, gy = ux.cpu().numpy(), uy.cpu().numpy()
    steps_dev = hp.ssfm_stats()[1] // cfg.nfft
    gam, betat, db1 = hp._keep
    g = math.exp(hp.alphalin * cfg.length)
    tot = 0
    for f in (0, 1):
        hx, hy = hp.tx_host[0] * math.sqrt(scale[f]), hp.tx_host[1] * math.sqrt(scale[f])
        ncs = []
        for s in range(2):
            rc, fd, nc, hx, hy = oracle.matrix_ssfm(hx, hy, betat, db1, cfg.dzmax, cfg.dphimax, gam, hp.alphalin, cfg.length, 1, 0,
                                                    hp.fls, [0.0], [0.0], [0.0])
            assert rc == 0
            hx, hy = hx[:, 0], hy[:, 0]
            ncs.append(nc)
            if s == 0:
                hx, hy = math.sqrt(g) * hx, math.sqrt(g) * hy
        tot += sum(ncs)
        assert hp.last_ncycle(2)[f] == ncs[1]                          # the last span's fingerprint of this frame
        assert np.abs(gx[f] - hx).max() <= FIELD_RTOL * np.abs(hx).max()
        assert np.abs(gy[f] - hy).max() <= FIELD_RTOL * np.abs(hy).max()
    assert steps_dev == tot                                            # every step of both spans, both frames
    hp.close()


# ================================================================================ C0 ===
def test_c0_32gbaud_single_pol_linear_span_with_post_compensation(lib, oracle):
    """BASELINE config[0] AT ITS STATED PARAMETERS: 32 Gbaud single-polarisation QPSK, Nsymb 256 x
    Nt 64, one 80 km span fiber(x,'g---') (D = 17, alpha = 0.2: a single exact step, fiber.m:162-165), amplifier, and
    the receiver's own post-compensating fibre x.dpost = -D*L (receiver_cohmix.m:149-168) in front of dsp4cohdec --
    against scalar_ssfm + front.py + dsp_pdm_coh_qpsk of the oracle.  Linear and fully compensated: no bit errors."""
    import polmux_amd as px
    from oracle import front
    from polmux_amd import rxfront, synth
    from polmux_amd.fiber import fiber_tables, parse_flag
    from polmux_amd.gstate import GSTATE, to_host_field
    nsymb, nt = 256, 64
    px.reset_all(nsymb, nt, 1)
    GSTATE.SYMBOLRATE = 32.0
    E = px.lasersource(1.0, 1550.0, 0.4)
    pat, patmat = synth.pattern_debruijn(nsymb, 1, 4)
    eopt = synth.qi_modulator(E[:, 0], synth.electricsource_qpsk(patmat[:, 0], nt, 1.0, 0.2),
                              synth.electricsource_qpsk(patmat[:, 1], nt, 1.0, 0.2))
    px.create_field("unique", eopt.reshape(-1, 1), None, dict(power="average"))
    tx = to_host_field(GSTATE.FIELDX)[:, 0].copy()
    fib = dict(length=8e4, alphadB=0.2, aeff=80.0, n2=2.7e-20, disp=17.0, slope=0.0, dphimax=5e-3, dzmax=2e4)
    fib["lambda"] = 1550.0
    px.fiber(fib, "g---")
    px.ampliflat(0.2 * 80.0, "gain")                                   # Gerbio = alphadB * L
    field = to_host_field(GSTATE.FIELDX)[:, 0]
    fls, dph, dzm = parse_flag("g---", 1, fib)
    t = fiber_tables(fib, fls, 1, 0.0)
    fd, nc, ou = oracle.scalar_ssfm(tx.reshape(-1, 1), t["betat"], dzm, dph, t["gam"], t["alphalin"], fib["length"], fls)
    assert nc == 1
    ou = ou[:, 0] * math.sqrt(10 ** (0.2 * 80.0 / 10))
    assert np.abs(field - ou).max() <= FIELD_RTOL * np.abs(ou).max()
    assert np.abs(field - tx).max() > 0.1 * np.abs(tx).max()           # 1360 ps/nm at 32 Gbaud: the eye is gone
    x = dict(rec="coherent", ts=0, oftype="gauss", obw=1.9, eftype="bessel5", ebw=0.65, delay="theory", lopower=0,
             dpost=-17.0 * 80.0, slopez=0.0)
    x["lambda"] = 1550.0
    p = dict(sps=nt, workatbaudrate=False, applyadc=False, adcbits=5, samplingrate=64.0, applydcf=False, applynlr=False,
             applypol=False, modorder=2, freqavg=500, phasavg=3, poworder=2)
    phase, amp, eye = px.dsp4cohdec(1, pat, x, p)
    hopt, elo, hel, post_delay, _ = rxfront._front_tables(1, x)
    cur = front.receiver_cohmix(ou, None, hopt, elo, hel, True)
    shift = rxfront._mround(-rxfront.theory_delay(1, x, False, post_delay) * nt)
    rx = front.rx_front(cur, False, 0, [shift], nt // 2, rxfront.fir1_lowpass(16, 2.0 / nt))
    op = oracle.dsp_params(power_mw=float(GSTATE.POWER[0]), applypol=False, freqavg=500, phasavg=3, poworder=2)
    ref = oracle.dsp_pdm_coh_qpsk(rx, op)
    got = amp.cpu().numpy() * np.exp(1j * phase.cpu().numpy())
    np.testing.assert_allclose(got, ref, atol=1e-11)
    pat_hat = px.samp2pat(x, None, phase.cpu().numpy())
    want = oracle.samp2pat_coherent(np.angle(ref))
    np.testing.assert_array_equal(pat_hat, want)
    # no noise, dispersion undone: the transmitted bits come back up to the blind phase estimate's pi/2 ambiguity
    errs = min(int((oracle.samp2pat_coherent(np.angle(ref * 1j ** k)) != patmat).sum()) for k in range(4))
    assert errs == 0


# ===================================================== data-dependent trip counts in one batch ===
def test_power_ladder_batch_every_frame_keeps_the_reference_step_count(lib, oracle):
    """fiber.m:518,534: the number of steps depends on the frame's own peak power.  A batch whose 64 frames sit on BASELINE
    config[4]'s launch-power ladder (-4...+8 dBm, different de Bruijn sequences as well) advances in lock-step launches;
    EVERY frame must leave with the oracle's ncycle and firstdz, sampled frames with the oracle's field -- whatever the
    other frames of the batch are doing (finished frames drop out of the launches, fuzz_batch.py's check as a test)."""
    from polmux_amd import pipeline
    cfg = pipeline.HotPathConfig(nsymb=256, nt=64, variants=4, length=8e4)
    F = 64
    hp = pipeline.HotPath(cfg, max_frames=F)
    assert hp.fused()
    dbm = -4.0 + 12.0 * np.arange(F) / 63.0
    scale = 10 ** (dbm / 10) / cfg.pavg_mw
    ux, uy = hp.make_batch(F, scale)
    hp.fibre(ux, uy)
    _sync()
    nc = hp.last_ncycle(F)
    fd = np.zeros(F)
    hp.lib.call("plx_ssfm_results", hp.ssfm, F, fd.ctypes.data, None)
    gam, betat, db1 = hp._keep
    gx, gy = ux.cpu().numpy(), uy.cpu().numpy()
    want = []
    for f in range(F):
        vx, vy, _ = hp.var_host[f % hp.nvar]
        rc, ofd, onc, ox, oy = oracle.matrix_ssfm(vx * math.sqrt(scale[f]), vy * math.sqrt(scale[f]), betat, db1, cfg.dzmax, cfg.dphimax,
                                                  gam, hp.alphalin, cfg.length, 1, 0, hp.fls, [0.0], [0.0], [0.0])
        want.append(onc)
        assert abs(fd[f] - ofd) <= 1e-12 * ofd
        if f in (0, 17, 40, 63):
            assert np.abs(gx[f] - ox[:, 0]).max() <= FIELD_RTOL * np.abs(ox).max()
            assert np.abs(gy[f] - oy[:, 0]).max() <= FIELD_RTOL * np.abs(oy).max()
    assert nc.tolist() == want
    assert max(want) >= 4 * min(want)                                # the ladder really spreads the trip counts
    assert hp.ssfm_stats()[1] == sum(want) * cfg.nfft
    hp.close()


@pytest.mark.parametrize("nspans", [3, 40])
def test_small_ladder_batch_of_2pow20_frames_fused_equals_three_sweeps(lib, tune, nspans):
    """Config[4]'s shape: 8 frames of 2^20 samples on a steep launch-power ladder, three spans -- and the FORTY spans
    config[4] states (the shape whose stale-list walk once stalled the fused sweep).  A 2^20 frame is ONE
    team of the fused column sweep (512 tiles = the whole grid), and batches under 64 frames rebuild the active list only once
    per chunk of steps -- so most of the time the team walks a STALE list, running through finished frames without meeting at
    a barrier while its first workgroup posts the next frames ahead.  (A ring of four mailbox entries used to be overwritten
    there before the slowest workgroup had read it: the launch never ended.  The mailbox is a log now.)  Property instead of
    the oracle at this size: the fused sweep and the barrier-free three-sweep step give the same step counts and fields."""
    import torch
    from polmux_amd import pipeline
    F = 8
    dbm = -4.0 + 12.0 * np.arange(F) / (F - 1)
    out = []
    # (the second plan also keeps k_row4k's whole-sample exchanges: the default splits them into real / imaginary halves)
    for env in ({}, {"PLX_SSFM_NO_FUSE": "1", "PLX_SSFM_ROW4K_SPLIT": "0"}):
        for k, v in env.items():
            tune.setenv(k, v)
        cfg = pipeline.HotPathConfig(nsymb=16384, nt=64, nspans=nspans)
        hp = pipeline.HotPath(cfg, max_frames=F)
        for k in env:
            tune.delenv(k)
        assert hp.fused() == ("PLX_SSFM_NO_FUSE" not in env)
        scale = 10 ** (dbm / 10) / cfg.pavg_mw
        ux, uy = hp.make_batch(F, scale)
        hp.fibre(ux, uy)
        _sync()
        out.append((hp.last_ncycle(F).copy(), hp.ssfm_stats()[1], ux.cpu().numpy(), uy.cpu().numpy()))
        hp.close()
        del ux, uy
        torch.cuda.empty_cache()
    (nc0, st0, x0, y0), (nc1, st1, x1, y1) = out
    assert nc0.tolist() == nc1.tolist() and st0 == st1
    assert max(nc0) >= 4 * min(nc0)
    # (forty strongly nonlinear spans amplify the two transforms' different rounding: the bar grows with the span count)
    bar = FIELD_RTOL * (1 if nspans <= 3 else 1e3)
    for f in range(F):
        assert np.abs(x0[f] - x1[f]).max() <= bar * np.abs(x1[f]).max()
        assert np.abs(y0[f] - y1[f]).max() <= bar * np.abs(y1[f]).max()


def test_fibre_beside_a_busy_stream_is_bit_identical_to_the_fibre_alone(lib):
    """The fused column sweep's teams claim their frames as they go: when another stream's kernels hold part of the chip, some
    teams start late (or never get a frame) and the others take their share.  Which team propagates a frame must not show in
    the result: a batch beside a stream of large FP64 matrix products leaves with the bits and step counts of the same batch on
    a quiet GPU."""
    import torch
    from polmux_amd import pipeline
    cfg = pipeline.HotPathConfig(nsymb=1024, nt=64, variants=4)
    F = 256
    hp = pipeline.HotPath(cfg, max_frames=F)
    assert hp.fused()
    dbm = -2.0 + 6.0 * np.arange(F) / (F - 1)                  # different trip counts: frames leave the list on the way
    scale = 10 ** (dbm / 10) / cfg.pavg_mw
    ux0, uy0 = hp.make_batch(F, scale)
    ux1, uy1 = ux0.clone(), uy0.clone()
    hp.fibre(ux0, uy0)
    _sync()
    nc0 = hp.last_ncycle(F).copy()
    side = torch.cuda.Stream()
    a = torch.randn(6144, 6144, dtype=torch.float64, device="cuda")
    b = torch.randn(6144, 6144, dtype=torch.float64, device="cuda")
    _sync()
    with torch.cuda.stream(side):
        for _ in range(12):
            c = a @ b                                           # ~0.5 TFLOP each: the chip is busy for the whole pass
    hp.fibre(ux1, uy1)
    _sync()
    assert float(c.abs().max()) > 0
    assert hp.last_ncycle(F).tolist() == nc0.tolist() and max(nc0) > min(nc0)
    assert torch.equal(ux0, ux1) and torch.equal(uy0, uy1)
    hp.close()


@pytest.mark.parametrize("nsymb", [16384, 4096])
def test_front_end_at_2pow20_uses_the_long_row_filter_pass(lib, oracle, nsymb):
    """The coherent front end of a 2^20-sample frame: its two spectral filters run on the plan's FFT engine, i.e. through
    the 256 x 4096 split and the 4096-point row pass k_row4k with a general multiplier table -- photocurrents against the
    numpy restatement of receiver_cohmix.m:165-307 (oracle/front.py).  And of a 2^18-sample frame (the size Run_my_PDM_QPSK.m
    ships with): the optical filter through k_rowreg's 1024-point rows."""
    from oracle import front
    from polmux_amd import pipeline
    cfg = pipeline.HotPathConfig(nsymb=nsymb, nt=64, frontend="cohmix", adcbits=0)
    hp = pipeline.HotPath(cfg, max_frames=1)
    ux, uy = hp.make_batch(1)
    tx, ty = ux[0].cpu().numpy(), uy[0].cpu().numpy()
    hp.front.run(ux, uy, hp.front_shifts, out=hp.rx[:1])
    _sync()
    t = hp.front_tables
    cur = front.receiver_cohmix(tx, ty, t["hopt"], t["elo"], t["hel"], True)
    got = np.stack([ux[0].real.cpu().numpy(), ux[0].imag.cpu().numpy(), uy[0].real.cpu().numpy(), uy[0].imag.cpu().numpy()], 1)
    assert np.abs(got - cur).max() < 1e-10 * np.abs(cur).max()
    rx = front.rx_front(cur, True, 0, hp.front_shifts, t["decim"], t["fir"])
    assert np.abs(hp.rx[0].cpu().numpy().T - rx).max() < 1e-10 * np.abs(rx).max()
    hp.close()


def test_sentinel_landing_of_the_staged_tile_equals_the_ordinary_wait(lib, tune):
    """k_colx16 lands its staged tile WITHOUT a vmcnt wait: the copies are issued from inline assembly, the frame record goes
    last and the wave spins on the record's last word in LDS (in-order return of a wave's loads).  PLX_SSFM_SAFE_LANDING=1
    adds the ordinary s_waitcnt vmcnt(0) in front of that spin: fields, step counts and first steps must agree to the bit, on
    a batch whose frames leave the loop at different steps, three calls in a row."""
    import torch
    from polmux_amd import pipeline
    F = 5
    dbm = np.array([-3.0, 0.0, 2.0, 4.0, 6.0])
    res = {}
    # ("ldsrow": the LDS-resident k_row, which rounds the inter-pass twiddles differently from the register form k_row256r)
    # ("early": the fused sweep's stores issued before the next tile has landed, the order one-team launches keep; "fwdrows":
    #  the row pass over the listed frames in ascending order -- both only move work in time)
    for name, env in (("eager", {}), ("safe", {"PLX_SSFM_SAFE_LANDING": "1"}), ("early", {"PLX_SSFM_STORE_LATE": "0"}),
                      ("fwdrows", {"PLX_SSFM_ROW_REV": "0"}), ("ldsrow", {"PLX_SSFM_ROWR": "0"})):
        for k, v in env.items():
            tune.setenv(k, v)
        cfg = pipeline.HotPathConfig(nsymb=1024, nt=64, variants=2)
        hp = pipeline.HotPath(cfg, max_frames=F)
        for k in env:
            tune.delenv(k)
        assert hp.fused()
        scale = 10 ** (dbm / 10) / cfg.pavg_mw
        outs = []
        ux, uy = hp.make_batch(F, scale)
        keep = (ux.clone(), uy.clone())
        for rep in range(3):
            if rep == 1:                       # the same buffers again
                ux.copy_(keep[0]); uy.copy_(keep[1])
            if rep == 2:                       # other buffers
                ux, uy = keep[0].clone(), keep[1].clone()
            hp.fibre(ux, uy)
            _sync()
            fd = np.zeros(F)
            hp.lib.call("plx_ssfm_results", hp.ssfm, F, fd.ctypes.data, None)
            outs.append((ux.clone(), uy.clone(), hp.last_ncycle(F).copy(), fd))
        res[name] = outs
        hp.close()
    ref = res["eager"][0]
    assert max(ref[2]) > min(ref[2])           # frames leave the loop at different steps
    for name, base in (("eager", "eager"), ("safe", "eager"), ("early", "eager"), ("fwdrows", "eager"), ("ldsrow", "ldsrow")):
        ref = res[base][0]
        for o in res[name]:
            assert torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]), name
            assert o[2].tolist() == ref[2].tolist() and np.array_equal(o[3], ref[3])
    a, b = res["eager"][0], res["ldsrow"][0]       # the two row passes agree to rounding
    assert a[2].tolist() == b[2].tolist() and float((a[0] - b[0]).abs().max()) < 1e-12 * float(b[0].abs().max())


def test_campaign_on_a_plan_that_cannot_share_the_gpu_keeps_one_stream(lib):
    """A 2^20-sample frame takes the whole fused grid, so HotPath.overlap_ok() is False and receive() runs on the fibre's
    stream.  McCampaign.launch() must then put the EVM, the error resolution and its completion event on that SAME stream (it
    used to leave them on the receiver stream, which never waited for the fibre's: counts and EVM samples read symbols that
    were still being written).  The pipelined campaign equals the same realisations computed stage by stage with a device
    synchronisation after every stage, to the bit."""
    import torch
    from polmux_amd import pipeline
    cfg = pipeline.HotPathConfig(nsymb=16384, nt=64, pavg_mw=1.0, cma_mu=1 / 600)
    camp = pipeline.McCampaign(cfg, frames_per_call=2, noise_sigma=0.45)
    hp = camp.hp
    assert not hp.overlap_ok()
    idx = [0, 1, 2, 3]
    h1 = camp.launch(idx[:2])
    h2 = camp.launch(idx[2:])                      # the second batch is enqueued before the first one is read
    got, evm = camp.collect(h1 + h2, with_samples=True)
    want, wevm = [], []
    for i0 in (0, 2):                              # stage by stage, nothing in flight across a stage
        ux, uy = hp.make_batch(2)
        hp.fibre(ux, uy, span_keys=idx[i0:i0 + 2])
        _sync()
        hp.receive(ux, uy, camp.sigma, 20260101, None, idx[i0:i0 + 2])
        _sync()
        v = hp.evm(2)
        _sync()
        e = hp.errors_resolved(2)
        _sync()
        want.append(e.cpu().numpy()); wevm.append(v.cpu().numpy())
    assert got.tolist() == np.concatenate(want).tolist()
    assert np.array_equal(evm, np.concatenate(wevm)) and evm.min() > 0      # (the EVM is a continuous sample: equality to the bit)
    camp.close()


def _spin_helper():
    """Source of a helper PROCESS that holds the LDS of half the CUs for 3 s (tests/gpuhelpers/spin.hip): 128 workgroups x 120 KiB,
    so that no column workgroup (70 KiB) fits beside one and at most 2 x 128 of a one-team frame's 512 are resident.  It prints
    'spinning' once its kernel is launched and exits when the kernel has finished."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tests", "gpuhelpers", "_build", "libplxspin.so")
    if not os.path.exists(so):                          # (built by __graft_entry__.build(); a tree that skipped it builds it here)
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(root, "tests", "gpuhelpers", "spin.hip"), "-o", so])
    return ("import ctypes as C, sys\n"
            "s = C.CDLL(%r)\n"
            "s.plx_test_spin.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]\n"
            "rc = s.plx_test_spin(128, 256, 120 * 1024, 3.0, None)\n"
            "print('spinning' if rc == 0 else 'failed %%d' %% rc, flush=True)\n"
            "sys.exit(s.plx_test_spin_wait() if rc == 0 else 1)\n") % so


def test_fiber_wrapper_repeats_the_span_when_another_kernel_holds_the_gpu(lib, tune):
    """The Python fiber(x, flag) (fiber.m:372-389: a span ALWAYS returns a field) on a 16-channel 'sepfields' frame -- one team of
    the fused sweep -- while another process holds half the CUs: the frame barrier times out, fiber() restores GSTATE's field from
    its copy and repeats the span on the three-sweep step.  Same field (1e-12) and step count as the span on a quiet GPU with a
    fresh, fused plan; the replay list and the step log are disarmed whatever happened; the time-out is on record."""
    import subprocess
    import sys
    import torch
    import polmux_amd as px
    from polmux_amd import synth
    fibermod = sys.modules["polmux_amd.fiber"]          # (the package re-exports the FUNCTION under the module's name)
    from polmux_amd.gstate import GSTATE, to_host_field
    nsymb, nt, nch = 1024, 64, 16
    x = dict(length=1.5e4, alphadB=0.2, aeff=80.0, n2=2.7e-20, disp=17.0, slope=0.0, dphimax=5e-3, dzmax=2e4)
    x["lambda"] = 1550.0

    def stage():
        px.reset_all(nsymb, nt, nch)
        GSTATE.SYMBOLRATE = 28.0
        px.lasersource(np.full(nch, 1.0), 1550.0, 0.4)
        cols = [synth.pdm_qpsk_field(nsymb, nt, 1.0, 2 + 2 * k, 3 + 2 * k) for k in range(nch)]
        px.create_field("sepfields", np.stack([c[0] for c in cols], 1), np.stack([c[1] for c in cols], 1), dict(power="average"))
    fibermod.release_plans()
    tune.setenv("PLX_SSFM_BARRIER_TIMEOUT_MS", "150")
    stage()
    torch.cuda.synchronize()
    proc = subprocess.Popen([sys.executable, "-c", _spin_helper()], stdout=subprocess.PIPE, text=True)
    try:
        assert proc.stdout.readline().strip() == "spinning"
        px.fiber(dict(x, _log_dz=True), "g-s-")
        busy = (to_host_field(GSTATE.FIELDX), to_host_field(GSTATE.FIELDY), dict(px.fiber.last))
    finally:
        assert proc.wait(timeout=60) == 0
    (plan, _d), = fibermod._plans.values()
    cnt, info = C.c_int32(), (C.c_int32 * 8)()
    lib.call("plx_ssfm_barrier_timeouts", plan, C.byref(cnt), 0)
    lib.call("plx_ssfm_info", plan, info)
    assert cnt.value == 1 and info[0] == 0, "the span did not time out (was the frame co-resident after all?)"
    assert len(busy[2]["dz"]) == busy[2]["ncycle"] > 3
    fibermod.release_plans()
    stage()
    px.fiber(x, "g-s-")                                # a fresh plan on a quiet GPU: the fused step
    (plan, _d), = fibermod._plans.values()
    lib.call("plx_ssfm_info", plan, info)
    assert info[0] == 1 and px.fiber.last["ncycle"] == busy[2]["ncycle"] and px.fiber.last["firstdz"] == busy[2]["firstdz"]
    qx, qy = to_host_field(GSTATE.FIELDX), to_host_field(GSTATE.FIELDY)
    assert np.abs(busy[0] - qx).max() <= 1e-12 * np.abs(qx).max() and np.abs(busy[1] - qy).max() <= 1e-12 * np.abs(qy).max()
    fibermod.release_plans()


def test_gateway_falls_back_to_three_sweeps_when_another_kernel_holds_the_gpu(lib, oracle, tune):
    """fiber.m:372-389 always returns a field.  A 16-channel frame is ONE team of the fused sweep (512 tiles = the whole grid):
    its workgroups must all be resident to meet at the frame barrier.  With a bounded spinning kernel of ANOTHER PROCESS holding
    the LDS of half the CUs (tests/gpuhelpers/spin.hip, launched by a helper process: streams of one process may share a
    hardware queue, in which case the two kernels would simply run one after the other) they cannot; the barrier times out,
    nothing is stored after the time-out, and plx_matrix_ssfm repeats the span from its pinned staging copy on the barrier-free
    three-sweep step: the call SUCCEEDS, the field is the oracle's, plx_gateway_stats_ex counts the fallback, and the cached
    plan stays on the three-sweep step afterwards."""
    import os
    import subprocess
    import sys
    import torch
    from tests.test_gpu_parity import _desc, _fibre_case, _vp
    helper = _spin_helper()
    lib.call("plx_release_all")
    tune.setenv("PLX_SSFM_BARRIER_TIMEOUT_MS", "150")
    c = _fibre_case(1024, 64, "g-s-", 1.0, nfc=16, length=1.5e4)
    d = _desc(c)
    planes = [np.asfortranarray(v.copy()) for v in (c["ux"].real, c["ux"].imag, c["uy"].real, c["uy"].imag)]
    z = np.zeros(1)
    fd, nc = C.c_double(), C.c_int32()

    def stats():
        v = np.zeros(9, np.int64)
        lib.call("plx_gateway_stats_ex", v.ctypes.data, 9)
        return v
    s0 = stats()
    torch.cuda.synchronize()
    proc = subprocess.Popen([sys.executable, "-c", helper], stdout=subprocess.PIPE, text=True)
    try:
        assert proc.stdout.readline().strip() == "spinning"
        lib.call("plx_matrix_ssfm", *[_vp(p) for p in planes], C.byref(d), _vp(z), _vp(z), _vp(z), C.byref(fd), C.byref(nc))
        s1 = stats()
    finally:
        assert proc.wait(timeout=60) == 0
    torch.cuda.synchronize()
    assert s1[8] == s0[8] + 1, "the span did not take the fallback (was the frame co-resident after all?)"
    rc, ofd, onc, ox, oy = oracle.matrix_ssfm(c["ux"], c["uy"], c["t"]["betat"], c["t"]["db1"], c["dzm"], c["dph"], c["t"]["gam"],
                                              c["t"]["alphalin"], c["length"], 1, False, c["fls"], z, z, z)
    assert rc == 0 and nc.value == onc and fd.value == pytest.approx(ofd, rel=1e-12)
    gx, gy = planes[0] + 1j * planes[1], planes[2] + 1j * planes[3]
    assert np.abs(gx - ox).max() <= FIELD_RTOL * np.abs(ox).max()
    assert np.abs(gy - oy).max() <= FIELD_RTOL * np.abs(oy).max()
    # the same span again on a quiet GPU: the cached plan is found and no longer waits for anybody (no second fallback)
    planes2 = [np.asfortranarray(v.copy()) for v in (c["ux"].real, c["ux"].imag, c["uy"].real, c["uy"].imag)]
    lib.call("plx_matrix_ssfm", *[_vp(p) for p in planes2], C.byref(d), _vp(z), _vp(z), _vp(z), C.byref(fd), C.byref(nc))
    s2 = stats()
    assert s2[8] == s1[8] and s2[4] == s1[4] + 1
    for a, b in zip(planes, planes2):
        np.testing.assert_array_equal(a, b)
    lib.call("plx_release_all")


def test_share_device_plan_runs_the_receiver_beside_the_next_fibre_bit_equal(lib):
    """plx_ssfm_create_ex(..., PLX_SSFM_SHARE_DEVICE): a 2^20-sample frame is the whole grid of the fused sweep, so its receiver
    used to wait on the fibre's stream.  On the barrier-free three-sweep step the receiver of batch i runs on its own stream
    beside the fibre of batch i + 1: symbols, error counts and step counts equal the same batches run one stage after the other."""
    import torch
    from polmux_amd import pipeline
    cfg = pipeline.HotPathConfig(nsymb=16384, nt=64, cma_mu=1 / 600, share_device=True)
    F = 2
    hp = pipeline.HotPath(cfg, max_frames=F)
    assert not hp.fused() and hp.overlap_ok() and hp.info()[2] == 12        # three sweeps on the 256 x 4096 split
    side = torch.cuda.Stream()
    scales = [np.array([1.0, 1.6]), np.array([0.7, 2.2])]
    serial = []
    for sc in scales:
        ux, uy = hp.make_batch(F, sc)
        hp.fibre(ux, uy)
        _sync()
        e = hp.receive(ux, uy, noise_sigma=0.05, noise_seed=5)
        _sync()
        serial.append((hp.sym[:F].clone(), e.clone(), hp.last_ncycle(F).copy()))
    got = []
    ux0, uy0 = hp.make_batch(F, scales[0])
    hp.fibre(ux0, uy0)
    nc0 = hp.last_ncycle(F).copy()
    e0 = hp.receive(ux0, uy0, noise_sigma=0.05, noise_seed=5, side_stream=side)     # enqueued, not waited for ...
    with torch.cuda.stream(side):
        keep = (hp.sym[:F].clone(), e0.clone())
    ux1, uy1 = hp.make_batch(F, scales[1])
    hp.fibre(ux1, uy1)                                                              # ... while the next batch propagates
    nc1 = hp.last_ncycle(F).copy()
    _sync()
    got.append((keep[0], keep[1], nc0))
    e1 = hp.receive(ux1, uy1, noise_sigma=0.05, noise_seed=5)
    _sync()
    got.append((hp.sym[:F].clone(), e1.clone(), nc1))
    for (s_, e_, n_), (gs, ge, gn) in zip(serial, got):
        assert torch.equal(s_, gs) and torch.equal(e_, ge) and n_.tolist() == gn.tolist()
    hp.close()


def test_two_channel_2pow20_frames_long_rows_equal_the_short_row_split(lib, tune):
    """A 'sepfields' field of two channels of 2^20 samples each (1024 column tiles per frame: more than the fused grid, so the
    plan takes the three-sweep step) on the 256 x 4096 split -- k_col_fwd / k_col_inv on 256-row tiles, k_row4k with the channel
    index in its workgroup map -- against the same frames on the 512 x 2048 split (PLX_SSFM_SHORT_ROWS=1: taller column tiles,
    the LDS-resident k_row per polarisation): same step counts, fields to 1e-9; the channels walk off and carry different powers."""
    import torch
    from polmux_amd import pipeline
    out = []
    for env in ({}, {"PLX_SSFM_SHORT_ROWS": "1"}):
        for k, v in env.items():
            tune.setenv(k, v)
        cfg = pipeline.HotPathConfig(nsymb=16384, nt=64, nch=2, flag="g-s-", length=3e4, pavg_mw=4.0, variants=3)
        hp = pipeline.HotPath(cfg, max_frames=2)
        for k in env:
            tune.delenv(k)
        info = hp.info()
        assert info[0] == 0 and info[2] == (11 if env else 12)
        ux, uy = hp.make_batch(2, np.array([1.0, 2.5]))
        hp.fibre(ux, uy)
        _sync()
        out.append((hp.last_ncycle(2).copy(), ux.cpu().numpy(), uy.cpu().numpy()))
        hp.close()
        del ux, uy
        torch.cuda.empty_cache()
    (nc0, x0, y0), (nc1, x1, y1) = out
    assert nc0.tolist() == nc1.tolist() and nc0[1] > nc0[0] > 3
    for a, b in ((x0, x1), (y0, y1)):
        assert np.abs(a - b).max() <= FIELD_RTOL * np.abs(b).max()
    assert np.abs(x0[0, 0] - x0[0, 1]).max() > 0.1 * np.abs(x0).max()          # the two channels really differ


def test_pmd_2pow20_frames_fused_sweep_and_both_rows_in_one_workgroup_vs_oracle(lib, oracle, tune):
    """fiber('gps-') has no size restriction (fiber.m:877-935).  2^20-sample frames with waveplates keep the 256 x 4096 split:
    the fused column sweep (one team = the whole grid) and k_row4k<true> -- both polarisations of a 4096-point row in one
    workgroup, the halves of every wave traded around the trunk loop.  Two frames with their own waveplate draws and launch
    powers: the stronger one against oracle.matrix_ssfm (field 1e-9, ncycle), both against the three-sweep step on the same
    split (PLX_SSFM_NO_FUSE=1) and against the 512 x 2048 split with the LDS-resident k_row (PLX_SSFM_SHORT_ROWS=1)."""
    import torch
    from polmux_amd import pipeline
    out = []
    scale = np.array([1.0, 2.0])
    for env in ({}, {"PLX_SSFM_NO_FUSE": "1"}, {"PLX_SSFM_SHORT_ROWS": "1"}):
        for k, v in env.items():
            tune.setenv(k, v)
        cfg = pipeline.HotPathConfig(nsymb=16384, nt=64, flag="gps-", nplates=50, dgd=0.1, length=4e4, dphimax=2e-2)
        hp = pipeline.HotPath(cfg, max_frames=2)
        for k in env:
            tune.delenv(k)
        info = hp.info()
        if not env:
            assert list(info[:3]) == [1, 8, 12] and info[4] == info[3] == 512 and info[6] == 512 and info[7] == 0
        elif "PLX_SSFM_NO_FUSE" in env:
            assert list(info[:3]) == [0, 8, 12] and info[6] == 512
        else:
            assert info[0] == 0 and info[2] == 11
        brf = hp.set_random_pmd([31, 32])
        ux, uy = hp.make_batch(2, scale)
        hp.fibre(ux, uy)
        _sync()
        out.append((hp.last_ncycle(2).copy(), ux.cpu().numpy(), uy.cpu().numpy()))
        if not env:
            gam, betat, db1 = hp._keep
            hx, hy = hp.tx_host[0] * math.sqrt(scale[1]), hp.tx_host[1] * math.sqrt(scale[1])
            rc, fd, nc, ox, oy = oracle.matrix_ssfm(hx, hy, betat, db1, cfg.dzmax, cfg.dphimax, gam, hp.alphalin, cfg.length, cfg.nplates,
                                                    0, hp.fls, brf[0][1], brf[1][1], brf[2][1])
            assert rc == 0 and nc == out[0][0][1] and nc > 4
            sc = max(np.abs(ox).max(), np.abs(oy).max())
            assert np.abs(out[0][1][1] - ox[:, 0]).max() <= FIELD_RTOL * sc
            assert np.abs(out[0][2][1] - oy[:, 0]).max() <= FIELD_RTOL * sc
        hp.close()
        del ux, uy
        torch.cuda.empty_cache()
    nc0, x0, y0 = out[0]
    assert nc0[1] > nc0[0]
    for nc1, x1, y1 in out[1:]:
        assert nc1.tolist() == nc0.tolist()
        for a, b in ((x0, x1), (y0, y1)):
            assert np.abs(a - b).max() <= FIELD_RTOL * np.abs(b).max()
    # the waveplates really couple the polarisations: with equal launch fields per polarisation pattern the outputs differ per draw
    assert np.abs(x0[0] * math.sqrt(2.0) - x0[1]).max() > 0.05 * np.abs(x0[1]).max()


@pytest.mark.parametrize("nsymb,nt", [(1024, 128), (4096, 64), (4096, 128), (256, 128)])
def test_frames_of_2pow17_to_2pow19_register_form_rows_vs_oracle(lib, oracle, tune, nsymb, nt):
    """Frames between the BASELINE shapes -- 2^18 = 4096 symbols x 64 samples is what Run_my_PDM_QPSK.m:21-24 ships with -- on
    the 256-row split: fused column sweep + k_rowreg (rows of 512 / 1024 / 2048 points in registers).  Three frames at different
    launch powers (different step counts in one batch): the strongest against oracle.matrix_ssfm (field 1e-9, ncycle), all
    three against the LDS-resident k_row (PLX_SSFM_ROWR=0) to 1e-9 with equal step counts."""
    import torch
    from polmux_amd import pipeline
    scale = np.array([0.5, 1.0, 2.0])
    out = []
    for env in ({}, {"PLX_SSFM_ROWR": "0"}, {"PLX_SSFM_ROWG_SPLIT": "0"}):
        for k, v in env.items():
            tune.setenv(k, v)
        cfg = pipeline.HotPathConfig(nsymb=nsymb, nt=nt, length=4e4)
        hp = pipeline.HotPath(cfg, max_frames=3)
        for k in env:
            tune.delenv(k)
        info = hp.info()
        p2 = (nsymb * nt).bit_length() - 1 - 8
        assert list(info[:3]) == [1, 8, p2] and (info[7] == 2) == ("PLX_SSFM_ROWR" not in env)
        assert hp.row_kernel() == (("k_rowreg" if p2 >= 9 else "k_rowsm") if "PLX_SSFM_ROWR" not in env else "k_row")   # (2^15: 128-point rows)
        ux, uy = hp.make_batch(3, scale)
        hp.fibre(ux, uy)
        _sync()
        out.append((hp.last_ncycle(3).copy(), ux.cpu().numpy(), uy.cpu().numpy()))
        if not env:
            gam, betat, db1 = hp._keep
            hx, hy = hp.tx_host[0] * math.sqrt(scale[2]), hp.tx_host[1] * math.sqrt(scale[2])
            rc, fd, nc, ox, oy = oracle.matrix_ssfm(hx, hy, betat, db1, cfg.dzmax, cfg.dphimax, gam, hp.alphalin, cfg.length, 1, 0,
                                                    hp.fls, [0.0], [0.0], [0.0])
            assert rc == 0 and nc == out[0][0][2]
            assert np.abs(out[0][1][2] - ox[:, 0]).max() <= FIELD_RTOL * np.abs(ox).max()
            assert np.abs(out[0][2][2] - oy[:, 0]).max() <= FIELD_RTOL * np.abs(oy).max()
        hp.close()
        del ux, uy
        torch.cuda.empty_cache()
    nc0, x0, y0 = out[0]
    assert nc0[2] > nc0[1] > nc0[0]
    for nc1, x1, y1 in out[1:]:          # the LDS-resident k_row; k_rowreg with whole-sample exchanges (rows of 512 / 1024 points split them by default)
        assert nc0.tolist() == nc1.tolist()
        for a, b in ((x0, x1), (y0, y1)):
            assert np.abs(a - b).max() <= FIELD_RTOL * np.abs(b).max()


@pytest.mark.parametrize("nsymb,nt", [(4096, 64), (4096, 128)])
def test_pmd_frames_of_2pow18_and_2pow19_register_form_rows_vs_oracle(lib, oracle, tune, nsymb, nt):
    """fiber('gps-') on frames of 2^18 and 2^19 samples: fused column sweep + k_rowreg<., true> (lanes i and i + 32 of a wave
    hold X and Y of the same bins and trade halves around the trunk loop).  Two frames with their own waveplate draws and
    launch powers: the stronger against oracle.matrix_ssfm (1e-9, ncycle), both against k_row's PMD branch (PLX_SSFM_ROWR=0)."""
    import torch
    from polmux_amd import pipeline
    out = []
    scale = np.array([1.0, 2.0])
    for env in ({}, {"PLX_SSFM_ROWR": "0"}):
        for k, v in env.items():
            tune.setenv(k, v)
        cfg = pipeline.HotPathConfig(nsymb=nsymb, nt=nt, flag="gps-", nplates=50, dgd=0.1, length=4e4, dphimax=2e-2)
        hp = pipeline.HotPath(cfg, max_frames=2)
        for k in env:
            tune.delenv(k)
        info = hp.info()
        assert info[0] == 1 and info[1] == 8 and (info[7] == 2) == (not env)
        brf = hp.set_random_pmd([41, 42])
        ux, uy = hp.make_batch(2, scale)
        hp.fibre(ux, uy)
        _sync()
        out.append((hp.last_ncycle(2).copy(), ux.cpu().numpy(), uy.cpu().numpy()))
        if not env:
            gam, betat, db1 = hp._keep
            hx, hy = hp.tx_host[0] * math.sqrt(scale[1]), hp.tx_host[1] * math.sqrt(scale[1])
            rc, fd, nc, ox, oy = oracle.matrix_ssfm(hx, hy, betat, db1, cfg.dzmax, cfg.dphimax, gam, hp.alphalin, cfg.length, cfg.nplates,
                                                    0, hp.fls, brf[0][1], brf[1][1], brf[2][1])
            assert rc == 0 and nc == out[0][0][1] and nc > 4
            sc = max(np.abs(ox).max(), np.abs(oy).max())
            assert np.abs(out[0][1][1] - ox[:, 0]).max() <= FIELD_RTOL * sc
            assert np.abs(out[0][2][1] - oy[:, 0]).max() <= FIELD_RTOL * sc
        hp.close()
        del ux, uy
        torch.cuda.empty_cache()
    (nc0, x0, y0), (nc1, x1, y1) = out
    assert nc0.tolist() == nc1.tolist() and nc0[1] > nc0[0]
    for a, b in ((x0, x1), (y0, y1)):
        assert np.abs(a - b).max() <= FIELD_RTOL * np.abs(b).max()

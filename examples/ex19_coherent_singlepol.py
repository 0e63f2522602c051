#!/usr/bin/env python
"""ex19_coherent_singlepol.m (BASELINE config[0]) on the device path: single-polarisation QPSK, noiseless + noisy flat
amplifier setting the OSNR, coherent receiver + DSP (dsp4cohdec), differential decoding, Monte-Carlo BER with
ber_estimate until its stop criterion -- the reference's loop, one realisation per iteration (ex19:112-154).

(The batched, sharded form of the same experiment is polmux_amd.pipeline.McCampaign / bench.py --mc.)
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(osnr=(3.0, 7.5), stop=(0.1, 68), max_runs=2000, seed=2, quiet=False):
    import polmux_amd as px
    from polmux_amd import mc as pmc
    from polmux_amd import patterns, synth
    samp = dict(logic=[0, 1], thr=0)
    mcx = dict(stop=stop)
    Nsymb, Nt, Nch = 2 ** 8, 64, 1
    Pavg, lam, spac, symbrate, duty, roll = 1.0, 1550.0, 0.4, 10.0, 1.0, 0.2
    x = dict(rec="coherent", ts=0, oftype="gauss", obw=1.9, eftype="bessel5", ebw=0.65, delay="theory", lopower=0)
    dsp = dict(sps=Nt, workatbaudrate=False, applyadc=False, adcbits=5, samplingrate=2 * symbrate, applydcf=False,
               dispersion=4000, ndispsym=16, applynlr=False, applypol=False, modorder=2, freqavg=500, phasavg=3, poworder=2)
    dsp["lambda"] = 1550
    Gerbio, nampli, osnrbw = 1.0, 1, 0.1
    C = px.CONSTANTS
    hvdl = -30 - 10 * math.log10(C.HPLANCK * C.CLIGHT / lam * C.CLIGHT * osnrbw / lam ** 2 * 1e18)
    osnr = np.asarray(osnr, dtype=float)
    nsp = 10 * math.log10(Pavg) + hvdl - 10 * math.log10(10 ** (Gerbio / 10) - 1) - 3 - 10 * math.log10(nampli) - osnr
    F = nsp + 3
    px.reset_all(Nsymb, Nt, Nch)
    pat, patmat = synth.pattern_debruijn(Nsymb, 1, 4)                          # ex19:108
    pat_rx, patmat_rx = patterns.pat_decoder(pat, "dqpsk")                     # ex19:109
    rng = np.random.default_rng(seed)
    out = []
    for knf in range(F.size):
        if not quiet:
            print("OSNR = %g" % osnr[knf])
        pmc.reset_persistent()
        cond, nruns, avgber, stdber = True, 0, float("nan"), float("nan")
        while cond and nruns < max_runs:
            px.reset_all(Nsymb, Nt, Nch)
            px.GSTATE.SYMBOLRATE = symbrate
            E = px.lasersource(Pavg, lam, spac)
            el_i = synth.electricsource_qpsk(patmat[:, 0], Nt, duty, roll)
            el_q = synth.electricsource_qpsk(patmat[:, 1], Nt, duty, roll)
            Eopt = synth.qi_modulator(E[:, 0], el_i, el_q)
            px.create_field("unique", Eopt.reshape(-1, 1), None, dict(power="average"))
            px.ampliflat(-Gerbio, "gain")                                      # noiseless amplifier
            px.ampliflat(Gerbio, "gain", dict(f=float(F[knf])), seed=int(rng.integers(1 << 62)))   # noisy amplifier
            phase, amplitude, _ = px.dsp4cohdec(1, pat, x, dsp)
            pat_hat = px.samp2pat(x, samp, phase.cpu().numpy())
            _, patmat_hat = patterns.pat_decoder(pat_hat, "dqpsk", dict(binary=True))
            cond, avgber, nruns, stdber = px.ber_estimate(patmat_hat, patmat_rx, mcx)
        if not quiet:
            a_, s_ = float(np.ravel(avgber)[0]), float(np.ravel(stdber)[0])
            print("P{error} = %5.2e  std/P{error} = %.3f  # runs=%11d\n" % (a_, s_ / a_ if a_ else float("nan"), int(np.ravel(nruns)[0])))
        out.append((float(osnr[knf]), float(np.ravel(avgber)[0]), float(np.ravel(stdber)[0]), int(np.ravel(nruns)[0])))
    return out


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""The reference's entry script, Run_my_PDM_QPSK.m, on the MI355X path: same parameters, same call
sequence, the reference's function names (polmux_amd mirrors them), every per-sample operation on the GPU.

  reset_all -> lasersource -> pattern/electricsource/qi_modulator (host Tx) -> create_field
  -> fiber(fib,'g---') -> RxPdmCohQpsk -> DspPdmCohQpsk (uncompensated) -> CDE_OFDE -> DspPdmCohQpsk
  -> samp2pat -> error count                                    (Run_my_PDM_QPSK.m:100-199)

The figures of the script are not drawn.  Returns/prints the per-polarisation match lines of :190-199
(which, like the script, compare columns 1:2 for both lines).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(Nsymb=4096, Nt=64, quiet=False):
    import polmux_amd as px
    from polmux_amd import synth
    # ---- 'Signal parameters' :12-27
    sigParams = dict(logic=[0, 1], thr=0)
    Pavg, chSpac, symbolRate, duty, roll = 2.0, 0.4, 10.0, 1.0, 0.2
    Nch, lam = 1, 1310.0
    # ---- 'Fiber parameters' :29-48
    fib = dict(length=1e3, alphadB=0.2, aeff=80.0, n2=2.7e-20, disp=60.0, slope=0.0, dphimax=5e-3, dzmax=2e4,
               pmd=0.04, nplates=10, manakov="no")
    fib["lambda"] = lam
    fib["dgd"] = fib["pmd"] * np.sqrt(fib["length"] / 1e3) * symbolRate * 10 - 3      # :46 (as written)
    # ---- 'Receiver parameters' :50-74
    RxParams = dict(rec="coherent", ts=0, oftype="gauss", obw=1.9, oord=3, eftype="bessel5", ebw=0.65, eord=4,
                    delay="theory", lopower=0, sps=Nt, workatbaudrate=False, applyadc=True, adcbits=5,
                    baudrate=symbolRate, samplingrate=2 * symbolRate, applydcf=False, dispersion=17e5, ndispsym=16)
    RxParams["lambda"] = lam
    # ---- 'DSP parameters' :76-97
    dspParams = dict(workatbaudrate=False, applynlr=False, applypol=False, polmethod="cma",
                     cmaparams=dict(R=[1, 1], mu=1 / 6000, taps=7, txpolars=2, phizero=0),
                     easiparams=dict(mu=1 / 6000, txpolars=2, phizero=0), modorder=2, freqavg=500, phasavg=3, poworder=2)
    # ---- Sending data :100-117
    px.reset_all(Nsymb, Nt, Nch)
    px.GSTATE.SYMBOLRATE = symbolRate                                                   # set by electricsource.m
    carrier = px.lasersource(Pavg, lam, chSpac)
    TxBits4D, TxSymbolsXY, sigx, sigy = [], [], [], []
    for chNum in range(1, Nch + 1):
        sx_, bx = synth.pattern_debruijn(Nsymb, chNum + 1, 4)
        sy_, by = synth.pattern_debruijn(Nsymb, chNum + 2, 4)
        bits = np.concatenate([bx, by], 1)
        TxBits4D.append(bits)
        TxSymbolsXY.append(np.stack([sx_, sy_], 1))
        el = [synth.electricsource_qpsk(bits[:, k], Nt, duty, roll) for k in range(4)]
        sigx.append(synth.qi_modulator(carrier[:, chNum - 1], el[0], el[1]))
        sigy.append(synth.qi_modulator(carrier[:, chNum - 1], el[2], el[3]))
    px.create_field("sepfields", np.stack(sigx, 1), np.stack(sigy, 1), dict(power="average"))
    # ---- Transmission :119-123
    px.fiber(fib, "g---")
    out = dict(TxBits4D=TxBits4D)
    lines = []
    for chNum in range(1, Nch + 1):
        # ---- Receiving data :126-186
        RxSamplCompXY, worsteyeop = px.RxPdmCohQpsk(chNum, TxSymbolsXY[chNum - 1], RxParams)
        OutSampCompNonCDXY = px.DspPdmCohQpsk(RxSamplCompXY.transpose(0, 1), dspParams, chNum)
        fs = RxParams["samplingrate"] * 1e9
        cdx, cdy, _ = px.CDE_OFDE(RxSamplCompXY[:, 0], RxSamplCompXY[:, 1], fs, lam * 1e-9, fib["length"],
                                  fib["disp"] * 1e-6, fib["slope"] * 1e-6, 256, 128)
        import torch
        OutSampCompXY = px.DspPdmCohQpsk(torch.stack([cdx, cdy]), dspParams, chNum)     # [2, Nsymb]
        RxSignalPhase = torch.angle(OutSampCompXY).cpu().numpy().T
        RxBits4D = px.samp2pat(RxParams, sigParams, RxSignalPhase)
        # ---- Calc Errors :189-199 (both lines use columns 1:2, as in the script)
        tx = TxBits4D[chNum - 1]
        r = 2 * tx.shape[0]
        m = int((tx[:, :2] == RxBits4D[:, :2]).sum())
        for pol in "XY":
            lines.append("Ch %d Pol %s Match: %d / %d | Errors: %d" % (chNum, pol, m, r, r - m))
        out.update(RxSamplCompXY=RxSamplCompXY, OutSampCompNonCDXY=OutSampCompNonCDXY, RxSamplCompCD=(cdx, cdy),
                   OutSampCompXY=OutSampCompXY, RxBits4D=RxBits4D, fib=fib, RxParams=RxParams, dspParams=dspParams)
    if not quiet:
        print("\n".join(lines))
    out["lines"] = lines
    return out


if __name__ == "__main__":
    main()

"""Batched, device-resident hot path: fibre -> (pick to 2 sps) -> CDE_OFDE ->
DspPdmCohQpsk (CMA + carrier recovery) -> decisions/error count.

This is what bench.py times and what the Monte-Carlo runner shards.  It is the
chain of Run_my_PDM_QPSK.m:122-193 minus the analogue front end
(receiver_cohmix + decimate, SURVEY 8f-1, "next"): the harness takes the
symbol-centre and mid-symbol samples of the propagated field directly.  Every
stage is a call into libpolmux_hip through the resident tier of the C ABI;
torch only owns the HBM buffers and the stream.
"""
import ctypes as C
import math

import numpy as np

from . import _abi, synth
from .fiber import fiber_tables, parse_flag
from .gstate import GSTATE
from .rx import cde_transfer, dsp_params_struct


def _u01(keys):
    """splitmix64 finaliser of uint64 keys -> doubles in [0, 1) (53 bits)."""
    with np.errstate(over="ignore"):
        z = keys + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


class HotPathConfig:
    """Run_my_PDM_QPSK-style parameters (BASELINE config C1 by default)."""

    def __init__(self, nsymb=1024, nt=64, symbolrate=28.0, pavg_mw=2.0, lam=1550.0, flag="g-s-",
                 length=8e4, alphadB=0.2, aeff=80.0, n2=2.7e-20, disp=17.0, slope=0.0, dphimax=5e-3, dzmax=2e4,
                 dgd=0.1, nplates=100, manakov="no", nspans=1, fft_length=256, cde_L=128, applypol=True,
                 polmethod="cma", cma_taps=7, cma_mu=1 / 6000, freqavg=500, phasavg=3, poworder=2,
                 frontend="pick", oftype="gauss", obw=1.9, oord=3, eftype="bessel5", ebw=0.65, eord=4, lopower=0.0,
                 adcbits=5, span_nf_db=None, rx_amp=False, variants=1, nch=1, chspacing=0.4, share_device=False):
        """frontend: 'pick' = 2-sps sampling supplied by the harness; 'cohmix' = the reference's own
        receiver_cohmix + ADC + decimate chain (RxPdmCohQpsk.m, Run_my_PDM_QPSK.m:52-73 defaults) on the device.
        nspans > 1: every span but the last is followed by an in-line flat amplifier restoring its loss
        (ampliflat(G,'gain'), noiseless or with noise figure span_nf_db and ASE keyed per frame); the last span's loss is
        undone in the receiver scale, as for one span -- unless rx_amp: then the last span is followed by an amplifier as
        well (gain = span loss, ASE from span_nf_db), the `fiber(...); ampliflat(Gerbio,'gain',ampli)` of
        ex20_coherent_polmux.m:147-148 / ex24_pmd.m:86-87, and the receiver sees the amplified, noise-loaded field.
        variants: number of distinct Tx waveforms (de Bruijn seed pairs s+1, s+2 as Run_my_PDM_QPSK.m:104-105 does per
        channel); frame f carries variant f % variants, with its own transmitted bits for the error count.
        nch > 1: a frame is a 'sepfields' WDM field of nch columns (create_field.m:17-18, ex10_wdm.m:9-11; BASELINE
        config[2]) chspacing nm apart around lam: the channels share the step length (fiber.m:694-698) and each keeps its
        own walk-off and gamma (fiber.m:326-328); channel c of frame f carries variant (f nch + c) % variants, and every
        channel has its own receiver (receiver_cohmix.m:104-125 picks the column).
        share_device: the fibre plan takes the barrier-free three-sweep step (plx_ssfm_create_ex, PLX_SSFM_SHARE_DEVICE): a
        frame that is the whole grid of the fused sweep (2^20 samples, 16 channels) can then propagate beside the receiver of
        the previous batch on another stream (the fused sweep would wait for its frame's workgroups to be co-resident)."""
        self.__dict__.update(locals())
        del self.__dict__["self"]

    @property
    def nfft(self):
        return self.nsymb * self.nt


class HotPath:
    def __init__(self, cfg, max_frames):
        import torch
        self.torch = torch
        self.cfg = cfg
        self.F = int(max_frames)
        self.lib = _abi.get()
        self.dev = torch.device("cuda", torch.cuda.current_device())
        n = cfg.nfft
        nch = self.nch = int(cfg.nch)
        self.CF = self.F * nch                           # channel-frames: what the receiver's plans count
        # --- host side of fiber(): flag, conversions, tables (fiber.m:157-362) ---
        GSTATE.NSYMB, GSTATE.NT, GSTATE.NCH = cfg.nsymb, cfg.nt, nch
        GSTATE.SYMBOLRATE = cfg.symbolrate
        GSTATE.FN = synth.fn_grid(cfg.nsymb, cfg.nt)
        GSTATE.LAMBDA = cfg.lam + cfg.chspacing * (np.arange(nch) - (nch - 1) / 2)      # lasersource.m: equally spaced comb
        x = {"length": cfg.length, "alphadB": cfg.alphadB, "aeff": cfg.aeff, "n2": cfg.n2, "lambda": cfg.lam,
             "disp": cfg.disp, "slope": cfg.slope, "dphimax": cfg.dphimax, "dzmax": min(cfg.dzmax, cfg.length)}
        self.fls, dphimaxt, dzmaxt = parse_flag(cfg.flag, nch, x)
        self.pmd = self.fls[1] == 1
        nplates = cfg.nplates if self.pmd else 1
        dgdrms = math.sqrt(3 * math.pi / 8) * cfg.dgd / math.sqrt(nplates) if self.pmd else 0.0   # fiber.m:277
        t = fiber_tables(x, self.fls, nch, dgdrms)
        self.alphalin = t["alphalin"]
        d = _abi.SsfmDesc()
        d.nfft, d.nfc, d.dual_pol, d.max_frames = n, nch, 1, self.F
        for i in range(4):
            d.fls[i] = self.fls[i]
        d.dzmaxt, d.dphimaxt, d.alphalin, d.length = dzmaxt, dphimaxt, t["alphalin"], cfg.length
        d.nplates, d.manakov = nplates, int(str(cfg.manakov).lower() == "yes")
        self._keep = (np.ascontiguousarray(t["gam"]), t["betat"], t["db1"])
        d.gam, d.betat, d.db1 = (a.ctypes.data for a in self._keep)
        self.ssfm = C.c_void_p()
        self.lib.call("plx_ssfm_create_ex", C.byref(self.ssfm), C.byref(d), _abi.PLX_SSFM_SHARE_DEVICE if cfg.share_device else 0)
        self.nplates = nplates
        self._profiling = False
        if self.pmd:   # Monte-Carlo style: an independent random birefringence draw per frame (fiber.m:274-276)
            self.set_random_pmd(range(self.F))
        # --- Tx (host, once): Run_my_PDM_QPSK.m:101-117 ---
        ux, uy, bits, power = synth.pdm_qpsk_field(cfg.nsymb, cfg.nt, cfg.pavg_mw)
        self.tx_host = (ux, uy)
        self.bits = bits
        self.power_mw = power
        GSTATE.POWER = np.full(nch, power)
        self.tx = torch.from_numpy(np.stack([ux, uy])).to(self.dev)          # [2, n]
        self.pat = torch.from_numpy(np.ascontiguousarray(bits.T.astype(np.uint8))).to(self.dev)   # [4, nsymb]
        # further Tx waveforms (other de Bruijn seeds): heterogeneous batches whose frames differ in data
        self.var_host = [(ux, uy, bits)]
        maxseed = cfg.nsymb * (cfg.nsymb - 1) // 4
        for v in range(1, max(1, int(cfg.variants))):
            vx, vy, vb, vp = synth.pdm_qpsk_field(cfg.nsymb, cfg.nt, cfg.pavg_mw, (2 + 2 * v) % maxseed, (3 + 2 * v) % maxseed)
            assert abs(vp - power) <= 1e-9 * power      # de Bruijn sequences share their symbol statistics
            self.var_host.append((vx, vy, vb))
        self.nvar = len(self.var_host)
        if self.nvar > 1:
            self.tx_var = torch.from_numpy(np.stack([np.stack([v[0], v[1]]) for v in self.var_host])).to(self.dev)   # [V, 2, n]
            pv = np.stack([np.ascontiguousarray(v[2].T.astype(np.uint8)) for v in self.var_host])               # [V, 4, nsymb]
            self.pat_frames = torch.from_numpy(pv[np.arange(self.CF) % self.nvar].copy()).to(self.dev)         # [F nch, 4, nsymb]
        self.rx_gain = None                              # per-frame receiver scale of a launch-power ladder (make_batch)
        # --- Rx plans ---
        self.Lrx = 2 * cfg.nsymb
        fs = 2 * cfg.symbolrate * 1e9                                         # Run_my_PDM_QPSK.m:66,149
        N = min(cfg.fft_length, self.Lrx)
        H = cde_transfer(N, fs, cfg.lam * 1e-9, cfg.length * cfg.nspans, cfg.disp * 1e-6, cfg.slope * 1e-6)
        Hi = np.ascontiguousarray(H).view(np.float64)
        self.cde = C.c_void_p()
        self.lib.call("plx_cde_create", C.byref(self.cde), N, cfg.cde_L, Hi.ctypes.data)
        dsp = dict(workatbaudrate=False, applynlr=False, applypol=cfg.applypol, polmethod=cfg.polmethod,
                   cmaparams=dict(R=[1, 1], mu=cfg.cma_mu, taps=cfg.cma_taps, txpolars=2, phizero=0),
                   easiparams=dict(mu=cfg.cma_mu, txpolars=2, phizero=0), modorder=2, freqavg=cfg.freqavg,
                   phasavg=cfg.phasavg, poworder=cfg.poworder)
        self.dsp_p = dsp_params_struct(dsp, power)
        self.dsp = C.c_void_p()
        self.lib.call("plx_dsp_create", C.byref(self.dsp), self.Lrx, 2, self.CF, C.byref(self.dsp_p))
        # receive scale: undo the span loss and bring symbols to the 4*sqrt(P) full scale that
        # DspPdmCohQpsk divides by (DspPdmCohQpsk.m:22-23, "2* -> see receiver_cohmix")
        self.rx_scale = 4.0 * math.sqrt(power) / math.sqrt(power / 2.0)
        if not cfg.rx_amp:
            self.rx_scale *= math.exp(0.5 * self.alphalin * cfg.length)
        self.front = None
        if cfg.frontend == "cohmix":
            from . import rxfront
            rp = dict(oftype=cfg.oftype, obw=cfg.obw, oord=cfg.oord, eftype=cfg.eftype, ebw=cfg.ebw, eord=cfg.eord,
                      lopower=cfg.lopower)
            hopt, elo, hel, post_delay, _ = rxfront._front_tables(1, rp, nfc=nch)
            # in-line amplifier restoring the span loss, folded into the optical filter table (no extra sweep)
            if not cfg.rx_amp:
                hopt = hopt * math.exp(0.5 * self.alphalin * cfg.length)
            r = cfg.nt // 2                                                    # RxPdmCohQpsk.m:49-53, 2 samples/symbol
            delay = rxfront.evaldelay(cfg.oftype, cfg.obw * 0.5) + rxfront.evaldelay(cfg.eftype, cfg.ebw) + post_delay
            self.front_shifts = [rxfront._mround(-delay * cfg.nt)] * 2         # 'theory' delay, RxPdmCohQpsk.m:124-137
            self.front_tables = dict(hopt=hopt, elo=elo, hel=hel, fir=rxfront.fir1_lowpass(16, 1.0 / r), decim=r)
            self.front = rxfront._Front(n, True, self.CF, hopt, elo, hel, True, cfg.adcbits, r, self.front_tables["fir"])
        elif cfg.frontend != "pick":
            raise ValueError("frontend must be 'pick' or 'cohmix'")
        c128 = torch.complex128
        self.rx = torch.empty((self.CF, 2, self.Lrx), dtype=c128, device=self.dev)
        self.eq = torch.empty_like(self.rx)
        self.sym = torch.empty((self.CF, 2, cfg.nsymb), dtype=c128, device=self.dev)
        self.err = torch.zeros((self.CF, 2), dtype=torch.int64, device=self.dev)

    def close(self):
        for name, h in (("plx_ssfm_destroy", self.ssfm), ("plx_cde_destroy", self.cde), ("plx_dsp_destroy", self.dsp)):
            if h:
                self.lib.call(name, h)
        self.ssfm = self.cde = self.dsp = None
        if self.front is not None:
            self.front.close()
            self.front = None

    # ------------------------------------------------------------------ inputs ---
    def make_batch(self, nframes, launch_scale=None):
        """Synthetic inputs -> (ux, uy), each [F, n] complex128 ([F, nch, n] for 'sepfields' WDM frames): channel c of
        frame f carries Tx waveform (f nch + c) % variants, with an optional per-frame launch-power scaling (power sweep;
        the receiver then normalises each frame by its own launch power, as a per-run GSTATE.POWER does in
        DspPdmCohQpsk.m:22-23)."""
        torch = self.torch
        nch, n = self.nch, self.cfg.nfft
        ncf = nframes * nch
        if self.nvar > 1:
            idx = torch.arange(ncf, device=self.dev) % self.nvar
            ux = self.tx_var[idx, 0].contiguous()
            uy = self.tx_var[idx, 1].contiguous()
        else:
            ux = self.tx[0].unsqueeze(0).repeat(ncf, 1).contiguous()
            uy = self.tx[1].unsqueeze(0).repeat(ncf, 1).contiguous()
        self.rx_gain = None
        if launch_scale is not None:
            ls = np.repeat(np.asarray(launch_scale, dtype=float).reshape(-1), nch)
            k = torch.as_tensor(np.sqrt(ls), device=self.dev).reshape(-1, 1)
            ux, uy = ux * k, uy * k
            self.rx_gain = torch.as_tensor(1.0 / np.sqrt(ls), device=self.dev).reshape(-1, 1, 1)   # per channel-frame
        if nch > 1:
            ux, uy = ux.view(nframes, nch, n), uy.view(nframes, nch, n)
        return ux, uy

    def set_random_pmd(self, seeds):
        """brf draws of fiber.m:274-276, one independent set per frame, keyed by the realisation index alone (a
        counter-based generator: splitmix64 of (master seed, realisation, plate, stream) -> U[0,1)), so any batching
        or sharding of the indices gives the same waveplates.  Uploaded stream-ordered, without a device sync."""
        np_ = self.nplates
        r = np.asarray(list(seeds), dtype=np.uint64).reshape(-1, 1, 1)
        plate = np.arange(np_, dtype=np.uint64).reshape(1, -1, 1)
        stream = np.arange(3, dtype=np.uint64).reshape(1, 1, 3)
        with np.errstate(over="ignore"):       # uint64 arithmetic wraps by design
            keys = (np.uint64(20260101) * np.uint64(0x9E3779B97F4A7C15) + r) * np.uint64(0xD1342543DE82EF95) \
                + plate * np.uint64(3) + stream
        u = _u01(keys)
        db0 = np.ascontiguousarray(u[:, :, 0] * 2 * math.pi - math.pi)
        th = np.ascontiguousarray(u[:, :, 1] * math.pi - 0.5 * math.pi)
        ep = np.ascontiguousarray(0.5 * np.arcsin(u[:, :, 2] * 2 - 1))
        self.lib.call("plx_ssfm_set_birefringence_dev", self.ssfm, db0.ctypes.data, th.ctypes.data, ep.ctypes.data, db0.shape[0],
                      self.stream())
        return db0, th, ep

    # ------------------------------------------------------------------- stages ---
    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def fibre(self, ux, uy, span_keys=None, inject_noise=None):
        """ux, uy: [F, n] (or [F, nch, n]) complex128 device tensors ([frame][channel][nfft]), propagated in place.
        span_keys: per-frame keys of the amplifiers' ASE streams (realisation indices).  inject_noise: optional list,
        one entry per amplifier, of [F, 2, n] complex128 device tensors used INSTEAD of the device generator
        (ampliflat's options.noise, ampliflat.m:123-129: the parity route)."""
        F = ux.shape[0]
        self._rows = self._steps = 0
        cfg = self.cfg
        namp = 0
        for span in range(cfg.nspans):
            self.lib.call("plx_ssfm_propagate_dev", self.ssfm, ux.data_ptr(), uy.data_ptr(), F, self.stream())
            rows, steps = C.c_int64(), C.c_int64()
            self.lib.call("plx_ssfm_stats", self.ssfm, C.byref(rows), C.byref(steps))
            self._rows += rows.value
            self._steps += steps.value
            if span + 1 < cfg.nspans or cfg.rx_amp:   # in-line amplifier (ampliflat.m), stays on the device
                gain = math.exp(self.alphalin * cfg.length)
                sig = None
                if cfg.span_nf_db is not None:
                    from .ampliflat import ase_sigma
                    sig = np.ascontiguousarray(ase_sigma(cfg.span_nf_db, gain, self.nch), dtype=float)
                kt = None
                if span_keys is not None:
                    kt = self.torch.as_tensor(np.asarray(list(span_keys), dtype=np.int64), device=self.dev)
                inj = inject_noise[namp] if inject_noise is not None else None
                self.lib.call("plx_ampliflat_dev", ux.data_ptr(), uy.data_ptr(), cfg.nfft, self.nch, F, gain,
                              sig.ctypes.data if sig is not None else None, inj.data_ptr() if inj is not None else None,
                              (20260101 + 7919 * span) & (2 ** 64 - 1),
                              kt.data_ptr() if kt is not None else None, 1, 1, self.stream())
                namp += 1

    def receive(self, ux, uy, noise_sigma=0.0, noise_seed=None, side_stream=None, noise_keys=None):
        """Front end (2-sps pick, or receiver_cohmix + ADC + decimate), CDE, DSP, decisions.  Returns err [F,2] (device).
        Receiver noise (sigma per quadrature on the 2-sps samples, an ASE stand-in) comes from the device Philox
        generator of plx_ampliflat_dev keyed by (noise_seed, noise_keys[frame] or frame): with noise_keys = the
        realisation indices the noise of a realisation does not depend on batching or sharding.
        With side_stream the whole receiver is enqueued on that stream behind the fibre of this batch, so
        the latency-bound CMA recurrence overlaps the HBM-bound fibre sweeps of the NEXT batch."""
        if side_stream is not None and not self.overlap_ok():
            side_stream = None
        if side_stream is not None:
            torch = self.torch
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            side_stream.wait_event(ready)
            with torch.cuda.stream(side_stream):
                return self.receive(ux, uy, noise_sigma, noise_seed, None, noise_keys)
        F = ux.shape[0] * self.nch             # channel-frames: every channel of a 'sepfields' frame has its own receiver
        cfg = self.cfg
        half = cfg.nt // 2
        st = self.stream()
        rx = self.rx[:F]
        if self.nch > 1:
            ux, uy = ux.view(F, cfg.nfft), uy.view(F, cfg.nfft)
        if self.front is not None:             # receiver_cohmix + ADC + decimate; ux, uy are consumed
            if self.rx_gain is not None:       # launch-power ladder: each frame normalised by its own power
                ux.mul_(self.rx_gain[:F, :, 0])
                uy.mul_(self.rx_gain[:F, :, 0])
            self.front.run(ux, uy, self.front_shifts, out=rx)
        else:
            for pol, src in enumerate((ux, uy)):   # rx[f][pol][i] = scale * u_pol[f][i*half]
                self.lib.call("plx_pick_dev", src.data_ptr(), rx.data_ptr() + pol * self.Lrx * 16, cfg.nfft, self.Lrx, 0,
                              half, self.rx_scale, F, 2 * self.Lrx, st)
        if self.rx_gain is not None and self.front is None:   # launch-power ladder: each frame normalised by its own power
            rx.mul_(self.rx_gain[:F])
        if noise_sigma:
            sig = np.array([float(noise_sigma)])
            kt = None
            if noise_keys is not None:
                kt = self.torch.as_tensor(np.asarray(list(noise_keys), dtype=np.int64), device=self.dev)
            # one frame of rx = [X | Y] contiguous: a single-field, single-"polarisation" ampliflat with unit gain
            self.lib.call("plx_ampliflat_dev", rx.data_ptr(), None, 2 * self.Lrx, 1, F, 1.0, sig.ctypes.data, None,
                          int(noise_seed or 0) & (2 ** 64 - 1), kt.data_ptr() if kt is not None else None, 1, 0, st)
        self.lib.call("plx_cde_apply_dev", self.cde, rx.data_ptr(), self.eq.data_ptr(), self.Lrx, 2 * F, st)
        self.lib.call("plx_dsp_run_dev", self.dsp, self.eq.data_ptr(), self.sym.data_ptr(), F, st)
        if self.nvar > 1:     # frames carry different sequences: each compares with its own transmitted bits
            self.lib.call("plx_decide_count_frames_dev", self.sym.data_ptr(), cfg.nsymb, 2, F, self.pat_frames.data_ptr(),
                          4 * cfg.nsymb, None, self.err.data_ptr(), st)
        else:
            self.lib.call("plx_decide_count_dev", self.sym.data_ptr(), cfg.nsymb, 2, F, self.pat.data_ptr(), None,
                          self.err.data_ptr(), st)
        return self.err[:F]

    def _count_errors(self, F, swap):
        """err [F, 2] of the symbols now in self.sym against the transmitted bits (tributaries exchanged if swap)."""
        torch = self.torch
        if self.nvar > 1:
            pat = self.pat_frames[:F]
            if swap:
                pat = torch.cat([pat[:, 2:], pat[:, :2]], 1).contiguous()
            self._pat_keep = pat          # alive until the kernel has run
            self.lib.call("plx_decide_count_frames_dev", self.sym.data_ptr(), self.cfg.nsymb, 2, F, pat.data_ptr(),
                          4 * self.cfg.nsymb, None, self.err.data_ptr(), self.stream())
        else:
            pat = torch.cat([self.pat[2:], self.pat[:2]]).contiguous() if swap else self.pat
            self._pat_keep = pat
            self.lib.call("plx_decide_count_dev", self.sym.data_ptr(), self.cfg.nsymb, 2, F, pat.data_ptr(), None,
                          self.err.data_ptr(), self.stream())
        return self.err[:F].clone()

    def errors_resolved(self, F):
        """Per-frame bit errors after resolving what a blind receiver cannot know: the pi/2 phase
        ambiguity of the Viterbi&Viterbi estimate (per polarisation) and which CMA output carries which
        transmitted tributary (the pol-swap check of ex20_coherent_polmux.m:160-173).  Eight calls of
        the device decision/count kernel; returns an int64 tensor [F]."""
        torch = self.torch
        base = self.sym[:F].clone()
        best = []
        for swap in (False, True):
            b = None
            for k in range(4):
                self.sym[:F] = base * (1j ** k)
                e = self._count_errors(F, swap)
                b = e if b is None else torch.minimum(b, e)
            best.append(b.sum(1))
        self.sym[:F] = base
        return torch.minimum(best[0], best[1])

    def evm(self, F):
        """per-frame error-vector magnitude (mean |s - s_hat|^2) of the symbols now in self.sym: float64 tensor [F]"""
        out = self.torch.empty(F, dtype=self.torch.float64, device=self.dev)
        self.lib.call("plx_evm_dev", self.sym.data_ptr(), self.cfg.nsymb, 2, F, out.data_ptr(), self.stream())
        return out

    def errors_min_over_rotations(self, F):
        """Resolve the pi/2 ambiguity of the blind phase estimate per polarisation (host-side
        convenience for BER sanity; the reference's scripts use differential decoding instead)."""
        torch = self.torch
        best = None
        base = self.sym[:F].clone()
        for k in range(4):
            self.sym[:F] = base * (1j ** k)
            e = self._count_errors(F, False)
            best = e if best is None else torch.minimum(best, e)
        self.sym[:F] = base
        return best

    def run(self, ux, uy, noise_sigma=0.0, noise_seed=None):
        self.fibre(ux, uy)
        return self.receive(ux, uy, noise_sigma, noise_seed)

    def profile(self, on):
        """per-kernel HIP-event timing of the step loop (plx_ssfm_profile); read with kernel_times() after fibre()"""
        self._profiling = bool(on)
        self.lib.call("plx_ssfm_profile", self.ssfm, int(bool(on)))

    def kernel_times(self):
        """(ms[4], active launches[4]) accumulated over the fibre() calls since the previous call of this method: column sweep
        that starts a step, k_row, k_col_inv, control (plx_ssfm_kernel_times: the event intervals are read lazily)"""
        ms, nl = np.zeros(4), np.zeros(4, np.int64)
        if self._profiling:
            self.lib.call("plx_ssfm_kernel_times", self.ssfm, ms.ctypes.data, nl.ctypes.data)
        return ms, nl

    def last_ncycle(self, F):
        """ncycle (fiber.m:431) of each frame of the last propagate call"""
        nc = np.zeros(F, np.int32)
        self.lib.call("plx_ssfm_results", self.ssfm, F, None, nc.ctypes.data)
        return nc

    def utilisation(self):
        """(frame-steps with work, frame slots of the device's active list, frame slots the launches covered) of the last
        propagate call"""
        v = [C.c_int64(), C.c_int64(), C.c_int64()]
        self.lib.call("plx_ssfm_utilisation", self.ssfm, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def info(self):
        """plx_ssfm_info of the fibre plan: [fused, log2 N1, log2 N2, fused grid, column tiles per frame, ...]"""
        info = (C.c_int32 * 8)()
        self.lib.call("plx_ssfm_info", self.ssfm, info)
        return list(info)

    def fused(self):
        return bool(self.info()[0])

    def bind_gstate(self):
        """GSTATE as this plan's grid and comb need it (another HotPath built meanwhile has set its own)."""
        cfg = self.cfg
        GSTATE.NSYMB, GSTATE.NT, GSTATE.NCH = cfg.nsymb, cfg.nt, self.nch
        GSTATE.SYMBOLRATE = cfg.symbolrate
        GSTATE.FN = synth.fn_grid(cfg.nsymb, cfg.nt)
        GSTATE.LAMBDA = cfg.lam + cfg.chspacing * (np.arange(self.nch) - (self.nch - 1) / 2)
        GSTATE.POWER = np.full(self.nch, self.power_mw)

    def tx_columns(self):
        """Tx field of ONE frame as MATLAB holds it: (ux, uy), each [nfft x nch] (column c = variant c % variants)"""
        cols = [self.var_host[c % self.nvar] for c in range(self.nch)]
        return (np.asfortranarray(np.stack([c[0] for c in cols], 1)), np.asfortranarray(np.stack([c[1] for c in cols], 1)))

    def row_kernel(self):
        """name of the kernel that serves the step's row pass (for reports)"""
        info = (C.c_int32 * 8)()
        self.lib.call("plx_ssfm_info", self.ssfm, info)
        if info[6] == 64:
            return "k_row256r" if info[2] == 8 else "k_rowsm"    # (rows of 32 / 64 / 128 points)
        if info[7] == 2:
            return "k_rowreg"                                     # rows of 512 / 1024 / 2048 points, register form
        if info[2] == 12:
            return "k_row4k" if info[7] else "k_row4k<pair>"     # (PMD: both polarisations of a row in one workgroup)
        return "k_row"

    def overlap_ok(self):
        """May a second stream (the receiver of the previous batch) share the GPU with fibre()?  The fused column sweep needs
        the tiles of a frame co-resident; when ONE frame takes more than half of the grid (2^19- and 2^20-sample frames)
        a long-running receiver kernel that holds registers on every CU keeps the frame's second half from being placed
        until it ends: the two serialise (or the barrier times out).  Such plans run the receiver on the fibre's stream."""
        info = (C.c_int32 * 8)()
        self.lib.call("plx_ssfm_info", self.ssfm, info)
        return (not info[0]) or 2 * info[4] <= info[3]

    def ssfm_stats(self):
        """(row-pass launches, sample-steps) of the last fibre() call, summed over its frame groups"""
        return self._rows, self._steps


class McCampaign:
    """Monte-Carlo BER over random PMD + ASE realisations (the ex20-style loop around ber_estimate,
    with ex24's random-birefringence fibre): realisation r gets its own birefringence draw and its own
    noise, both keyed by r alone, so any sharding of the indices over GPUs gives the same counts.

    ASE: with cfg.rx_amp the span is followed by ampliflat(Gerbio,'gain',{f: cfg.span_nf_db}) as in
    ex20_coherent_polmux.m:147-148 (device Philox stream keyed by r, or `noise_provider(indices)` -> host array
    [n, 2, nfft] complex, the options.noise injection of ampliflat.m:123-129, for parity tests); `noise_sigma`
    additionally (or instead) loads the 2-sps receiver samples, the cheap stand-in used by small tests."""

    def __init__(self, cfg, frames_per_call, noise_sigma=0.0, noise_provider=None):
        self.hp = HotPath(cfg, frames_per_call)
        self.F = frames_per_call
        self.sigma = noise_sigma
        self.noise_provider = noise_provider
        self._rx_stream = None

    @property
    def bits_per_realisation(self):
        return 4 * self.hp.cfg.nsymb

    def launch(self, indices, keep=None):
        """Enqueue the realisations `indices` and return a handle WITHOUT waiting for the receiver: the fibre runs on the
        current stream (plx_ssfm_propagate_dev returns when its data-dependent step loop has ended), the receiver and the
        error counts go to a second stream, so the receiver of this batch overlaps the fibre of the next one (the CMA is
        latency-bound: ~25 ms whatever the batch size).  collect(handle) -> int64 error counts."""
        import torch
        hp = self.hp
        if self._rx_stream is None:
            self._rx_stream = torch.cuda.Stream()
        out = []
        for i0 in range(0, len(indices), self.F):
            idx = list(indices[i0:i0 + self.F])
            n = len(idx)
            if hp.pmd:
                hp.set_random_pmd(idx)
            ux, uy = hp.make_batch(n)
            inj = None
            if self.noise_provider is not None:
                inj = [torch.from_numpy(np.ascontiguousarray(self.noise_provider(idx))).to(hp.dev)]
            hp.fibre(ux, uy, span_keys=idx, inject_noise=inj)
            if keep is not None:
                keep(i0, idx, ux, uy)
            # ONE stream for the receiver and everything that reads its outputs (hp.sym, hp.err): the side stream when the
            # plan may share the GPU with the next fibre, the fibre's own stream otherwise (receive() would fall back to
            # it by itself, and the EVM / error kernels must follow the DSP in stream order)
            rxs = self._rx_stream if hp.overlap_ok() else torch.cuda.current_stream()
            side = rxs if rxs is self._rx_stream else None
            hp.receive(ux, uy, self.sigma, 20260101, side, idx)   # receiver noise keyed by realisation index
            with torch.cuda.stream(rxs):
                v = hp.evm(n)              # a continuous per-realisation sample (mc_estimate) beside the error count
                e = hp.errors_resolved(n)
                if side is not None:
                    ux.record_stream(rxs); uy.record_stream(rxs)
                done = torch.cuda.Event()
                done.record(rxs)
            out.append((e, v, done))
        return out

    def collect(self, handle, with_samples=False):
        """int64 error counts of a launch() handle (and, with_samples, the float64 EVM samples beside them)"""
        res, smp = [], []
        for e, v, done in handle or []:
            done.synchronize()             # the counts were formed on the receiver's stream
            res.append(e.cpu().numpy())
            smp.append(v.cpu().numpy())
        counts = np.concatenate(res) if res else np.zeros(0, np.int64)
        if with_samples:
            return counts, (np.concatenate(smp) if smp else np.zeros(0))
        return counts

    def simulate(self, indices, keep=None):
        """Error counts (pol swap and pi/2 ambiguities resolved, ex20_coherent_polmux.m:160-173) of the realisations
        `indices`.  keep(i0, idx, ux, uy): optional callback after the fibre + amplifier of each batch (tests read the
        field back there)."""
        return self.collect(self.launch(indices, keep))

    def close(self):
        self.hp.close()


class McRankShare:
    """The share ONE rank of `world` has in a campaign, run on its own: local index i stands for realisation rank + world * i
    (realisation r on GPU r mod world).  launch / collect / simulate of the wrapped campaign or pool."""

    def __init__(self, camp, rank, world):
        self.camp, self.rank, self.world = camp, int(rank), int(world)

    def _map(self, indices):
        return [self.rank + self.world * int(i) for i in indices]

    def launch(self, indices, keep=None):
        return self.camp.launch(self._map(indices), keep)

    def collect(self, handle, with_samples=False):
        return self.camp.collect(handle, with_samples)

    def simulate(self, indices, keep=None):
        return self.collect(self.launch(indices, keep))


class McCampaignPool:
    """`n` McCampaign instances taking the rounds of a campaign in turn, each with its own plans, receiver buffers and
    receiver stream: the receivers of up to n rounds are in flight at once (ShardedBer.run(depth=n - 1)).  A noise-loaded
    realisation's CMA runs all of its 299 passes -- ~58 ms of a serial recurrence whatever the batch size -- while its
    fibre takes a few ms: with one receiver in flight a round of 128 realisations is latency-bound on that."""

    def __init__(self, cfg, frames_per_call, n=2, noise_sigma=0.0, noise_provider=None, split=False):
        """split: ONE round is dealt over all n instances (contiguous parts, launched back to back, collected in order) instead
        of the rounds taking turns: the receivers of a round's parts run beside each other and the round still ends in ONE
        exchange -- what a rank of a strong-scaling run does with its fixed share (bench.py, mc.strong_scaling_rank_share)."""
        self.camps = [McCampaign(cfg, frames_per_call, noise_sigma, noise_provider) for _ in range(max(1, int(n)))]
        self._turn = 0
        self.split = bool(split)

    @property
    def bits_per_realisation(self):
        return self.camps[0].bits_per_realisation

    @property
    def hp(self):
        return self.camps[0].hp

    def launch(self, indices, keep=None):
        if self.split:
            idx = list(indices)
            per = -(-len(idx) // len(self.camps))
            return [(i, c.launch(idx[i * per:(i + 1) * per], keep)) for i, c in enumerate(self.camps) if idx[i * per:(i + 1) * per]]
        i = self._turn % len(self.camps)
        self._turn += 1
        return i, self.camps[i].launch(indices, keep)

    def collect(self, handle, with_samples=False):
        if self.split:
            parts = [self.camps[i].collect(h, with_samples) for i, h in handle]
            if with_samples:
                return (np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, np.int64),
                        np.concatenate([p[1] for p in parts]) if parts else np.zeros(0))
            return np.concatenate(parts) if parts else np.zeros(0, np.int64)
        i, h = handle
        return self.camps[i].collect(h, with_samples)

    def simulate(self, indices, keep=None):
        return self.collect(self.launch(indices, keep))

    def close(self):
        for c in self.camps:
            c.close()

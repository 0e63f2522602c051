"""Coherent front end: the step between fiber() and CDE_OFDE(), mirroring the reference signatures

  Hf = myfilter(ftype, f, bw, ord)                         myfilter.m:40-152
  y = evaldelay(ftype, bw)                                 evaldelay.m
  [Iric, x] = receiver_cohmix(ich, x)                      receiver_cohmix.m:95-307
  [RxSamples, worsteyeop] = RxPdmCohQpsk(chNum, symbolPattern, RxParams)   RxPdmCohQpsk.m:3-87

Host code builds the O(Nfft) tables (filters, local oscillator, decimation FIR) once; the per-sample work
(two spectral filters, hybrids + photodiodes, ADC, timing shift, decimation) runs on the GPU behind
plx_front_* (include/polmux_hip.h).  No CPU implementation exists here.

decimate(x,r,16,'fir') is MathWorks Signal Processing Toolbox code, not part of the reference: the FIR is
the published fir1 design (Hamming-windowed ideal low-pass, unit DC gain) and the edge/phase convention is
the one stated in DESIGN.md ("front end"); parity is unpinned at that boundary.
"""
import ctypes as C
import math

import numpy as np

from . import _abi
from .gstate import GSTATE, CONSTANTS

# myfilter.m:59-71
_R4P2R2 = 2.61312592975275
_B1, _B2, _B3 = 3.86370330515627315, 7.4641016151377546, 9.1416201726856413
_B4, _B5 = _B2, _B1
_BB = 0.3863
_D0, _D1, _D2, _D3, _D4 = 945, 945, 420, 105, 15


def myfilter(ftype, f, bw, ord=None):
    """Hf = myfilter(ftype, f, bw, ord): frequency response on the grid f, 3-dB bandwidth bw (myfilter.m:73-152)."""
    x = np.asarray(f, dtype=float).reshape(-1) / bw
    ftype = ftype.lower()
    if ftype == "movavg":
        return np.sinc(x)
    if ftype == "gauss":
        return np.exp(-0.5 * math.log(2) * x * x)
    if ftype == "gauss_off":
        return np.exp(-0.5 * math.log(2) * (x - ord / bw) * (x - ord / bw))
    if ftype == "butt2":
        return 1.0 / (1 - x * x + 1j * math.sqrt(2) * x)
    if ftype == "butt4":
        x2 = x * x
        umx2 = 1 - x2
        return 1.0 / (umx2 * umx2 - math.sqrt(2) * x2 + 1j * _R4P2R2 * x * umx2)
    if ftype == "butt6":
        x2 = x * x; x3 = x2 * x; x4 = x3 * x; x5 = x4 * x; x6 = x5 * x
        return 1.0 / (1. - _B2 * x2 + _B4 * x4 - x6 + 1j * (_B1 * x - _B3 * x3 + _B5 * x5))
    if ftype == "ideal":
        return (np.abs(x) <= 1).astype(float)
    if ftype == "bessel5":
        om = 2 * math.pi * x * _BB
        om2 = om * om; om3 = om2 * om; om4 = om3 * om; om5 = om4 * om
        pre = _D0 - _D2 * om2 + _D4 * om4
        pim = _D1 * om - _D3 * om3 + om5
        return _D0 / (pre + 1j * pim)
    if ftype == "rc1":
        return 1.0 / (1 + 1j * x)
    if ftype == "rc2":
        return 1.0 / (1 + 1j * math.sqrt(math.sqrt(2) - 1) * x) ** 2
    if ftype == "supergauss":
        if ord is None:
            raise ValueError("missing superGauss order")
        return np.exp(-0.5 * math.log(2) * x ** (2 * ord))
    raise ValueError("the filter ftype does not exist.")


def evaldelay(ftype, bw):
    """Group delay of myfilter's responses in symbols (evaldelay.m)."""
    ftype = ftype.lower()
    if ftype in ("movavg", "gauss", "gauss_off", "ideal", "supergauss"):
        return 0.0
    if ftype == "butt2":
        return 1.11 * math.sqrt(2) / (2 * math.pi * bw)
    if ftype == "butt4":
        return 1.1 * _R4P2R2 / (2 * math.pi * bw)
    if ftype == "butt6":
        return 1.1 * _B1 / (2 * math.pi * bw)
    if ftype == "bessel5":
        return _BB / bw
    if ftype == "rc1":
        return 1 / (2 * math.pi * bw)
    if ftype == "rc2":
        return (math.sqrt(2) - 1) / (math.pi * bw)
    raise ValueError("the filter ftype does not exist.")


def fir1_lowpass(order, wn):
    """fir1(order, wn): Hamming-windowed ideal low-pass, cut-off wn (1 = Nyquist), scaled to unit gain at DC
    (the documented window-method design; order+1 taps)."""
    k = np.arange(order + 1)
    w = 0.54 - 0.46 * np.cos(2 * math.pi * k / order)
    b = wn * np.sinc(wn * (k - order / 2.0)) * w
    return b / b.sum()


def _nmod(a, n):
    return (np.asarray(a) - 1) % n + 1


def _front_tables(ich, x, rng=None, nfc=None):
    """Everything receiver_cohmix.m:97-227,296 derives from x and GSTATE: (hopt, elo or scalar, hel, post_delay, b2b)."""
    CL = CONSTANTS.CLIGHT
    fn = np.asarray(GSTATE.FN, dtype=float)
    nfft = fn.size
    if nfc is None:
        nfc = GSTATE.FIELDX.shape[0]
    ndfn = 0
    if nfc != GSTATE.NCH:                                             # a 'unique' field: select channel ich, :104-125
        from .gstate import unique_field_shifts
        ndfn = int(unique_field_shifts()[ich - 1])
    b2b = False
    if "b2b" in x:                                                    # :132-141
        if x["b2b"] != "b2b":
            raise ValueError("the b2b field must be 'b2b'")
        b2b = True
    post_delay = 0.0
    if "dpost" in x and not b2b:                                      # :149-168
        lamv = np.atleast_1d(np.asarray(GSTATE.LAMBDA, dtype=float))
        maxl, minl = lamv.max(), lamv.min()
        lamc = 2 * maxl * minl / (maxl + minl)
        lam = x["lambda"]
        b20z = -lam ** 2 / 2 / math.pi / CL * x["dpost"] * 1e-3
        b30z = (lam / 2 / math.pi / CL) ** 2 * (2 * lam * x["dpost"] + lam ** 2 * x["slopez"]) * 1e-3
        d_i0 = 2 * math.pi * CL * (1.0 / lamv[ich - 1] - 1 / lam)
        d_ic = 2 * math.pi * CL * (1.0 / lamv[ich - 1] - 1 / lamc)
        d_c0 = 2 * math.pi * CL * (1.0 / lamc - 1 / lam)
        beta1z = b20z * d_ic + 0.5 * b30z * (d_i0 ** 2 - d_c0 ** 2)
        beta2z = b20z + b30z * d_i0
        omega = 2 * math.pi * GSTATE.SYMBOLRATE * fn
        betat = omega * beta1z + 0.5 * omega ** 2 * beta2z + omega ** 3 * b30z / 6
        post_delay = GSTATE.SYMBOLRATE * beta1z
        hf = np.cos(-betat) + 1j * np.sin(-betat)                     # fastexp(-betat)
    else:
        hf = np.ones(nfft, dtype=complex)
    hopt = hf * myfilter(x["oftype"], fn, 0.5 * x["obw"], x.get("oord"))   # :169
    if x.get("lodetuning"):                                           # :193-203
        minfreq = GSTATE.SYMBOLRATE * 1e9 / GSTATE.NSYMB
        kdet = math.floor(x["lodetuning"] / minfreq)
        det = 2 * math.pi * kdet / nfft * np.arange(1, nfft + 1)
    else:
        det = None
    if "lophasenoise" in x:                                           # :204-208
        pn = np.asarray(x["lophasenoise"], dtype=float).reshape(-1)
        if pn.size != nfft:
            raise ValueError("Incompatible vector.")
    elif "lolinewidth" in x:                                          # :209-216 (randn -> numpy Generator)
        rng = rng or np.random.default_rng()
        fnz = math.sqrt(2 * math.pi * x["lolinewidth"] / GSTATE.NT) * rng.standard_normal(nfft)
        fnz[0] = 0
        pn = np.cumsum(fnz)
        pn = pn - np.arange(nfft) / (nfft - 1) * pn[-1]               # Brownian bridge
    else:
        pn = None
    ecw = 10 ** (x["lopower"] / 20) if "lopower" in x else 1.0        # :220-224
    if det is None and pn is None:
        elo = ecw                                                     # fastexp(0) = 1
    else:
        ph = (0 if det is None else det) + (0 if pn is None else pn)
        elo = ecw * (np.cos(ph) + 1j * np.sin(ph))                    # :227
    hel = myfilter(x["eftype"], fn, x["ebw"], x.get("eord"))          # :296
    if ndfn:
        # sigx = fft(sigx); sigx = sigx(nind) moves the spectrum by ndfn bins before the optical filter (:104-107, :184).
        # The device path gets the same currents without touching the field: filter with the table moved back by
        # ndfn bins, and give the (unit-modulus) time-domain phasor of the shift to the local oscillator instead --
        # |j s p + j Elo|^2 = |j s + j Elo conj(p)|^2 for each of the four mixer outputs (:254-291).
        hopt = np.roll(hopt, -ndfn)
        p = np.exp(2j * math.pi * ndfn * np.arange(nfft) / nfft)
        elo = (elo * np.ones(nfft)) * np.conj(p)
    return hopt, elo, hel, post_delay, b2b


class _Front:
    """A plx_front plan plus the host tables that built it."""

    def __init__(self, nfft, dual, max_frames, hopt, elo, hel, balanced=True, adcbits=0, decim=1, fir=None):
        self.lib = _abi.get()
        d = _abi.FrontDesc()
        d.nfft, d.dual_pol, d.max_frames, d.balanced, d.adcbits, d.decim = nfft, int(dual), max_frames, int(balanced), adcbits, decim
        keep = []

        def vec(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data

        hopt = np.asarray(hopt, dtype=complex); hel = np.asarray(hel, dtype=complex)
        d.hopt_re, d.hopt_im = vec(hopt.real), vec(hopt.imag)
        d.hel_re, d.hel_im = vec(hel.real), vec(hel.imag)
        if np.ndim(elo) == 0:
            if np.imag(elo) != 0:
                elo = np.full(nfft, elo, dtype=complex)
            else:
                d.elo_scalar = float(np.real(elo))
        if np.ndim(elo) != 0:
            elo = np.asarray(elo, dtype=complex)
            d.elo_re, d.elo_im = vec(elo.real), vec(elo.imag)
        if decim > 1:
            d.ntaps, d.fir = len(fir), vec(fir)
        self.plan = C.c_void_p()
        self.lib.call("plx_front_create", C.byref(self.plan), C.byref(d))
        self.nout = int(self.lib.lib.plx_front_out_len(self.plan))
        self.dual, self.nfft = bool(dual), nfft

    def run(self, ux, uy, shifts=None, out=None):
        """ux, uy: torch complex128 [F, nfft] (overwritten with the photocurrents I + jQ); returns [F, npol, nout]."""
        import torch
        F = ux.shape[0]
        npol = 2 if self.dual else 1
        if out is None:
            out = torch.empty((F, npol, self.nout), dtype=torch.complex128, device=ux.device)
        sh = (C.c_int64 * 2)(*(list(shifts) + [0, 0])[:2]) if shifts is not None else None
        self.lib.call("plx_front_run_dev", self.plan, ux.data_ptr(), uy.data_ptr() if self.dual else None, F, sh,
                      out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out

    def close(self):
        if self.plan:
            self.lib.call("plx_front_destroy", self.plan)
            self.plan = None


def _channel_fields(ich, b2b):
    fx = GSTATE.FIELDX_TX if b2b else GSTATE.FIELDX
    fy = GSTATE.FIELDY_TX if b2b else GSTATE.FIELDY
    if fx is None:
        raise ValueError("GSTATE.FIELDX is empty")
    row = ich - 1 if fx.shape[0] == GSTATE.NCH else 0                   # nch = ich / nch = 1 for a unique field, :108,:123
    ux = fx[row:row + 1].clone()
    if GSTATE.FIELDY is None:
        return ux, None
    uy = fy[row:row + 1].clone() if fy is not None else ux * 0          # :230-236
    return ux, uy


def receiver_cohmix(ich, x, rng=None):
    """[Iric, x] = receiver_cohmix(ich, x): photocurrents [Nfft x 2] (X: I, Q) or [Nfft x 4] ([X Y]) as a torch
    float64 CUDA tensor, and x with post_delay added (receiver_cohmix.m:95-307).  x.avgebx/avgeby (:185-187,
    diagnostics) are not produced."""
    import torch
    hopt, elo, hel, post_delay, b2b = _front_tables(ich, x, rng)
    ux, uy = _channel_fields(ich, b2b)
    balanced = not (x.get("pdtype") == "normal")                      # :265-269
    fr = _Front(ux.shape[1], uy is not None, 1, hopt, elo, hel, balanced)
    try:
        fr.run(ux, uy)
    finally:
        fr.close()
    cols = [ux[0].real, ux[0].imag] + ([uy[0].real, uy[0].imag] if uy is not None else [])
    x = dict(x)
    x["post_delay"] = post_delay
    return torch.stack(cols, 1), x


def theory_delay(ich, x, isy, post_delay):
    """delay of mygeteyeinfo with x.delay == 'theory' (RxPdmCohQpsk.m:124-137)."""
    if x.get("b2b") == "b2b":
        avg = 0.0
    elif isy:
        avg = 0.5 * (GSTATE.DELAY[0, ich - 1] + GSTATE.DELAY[1, ich - 1])
    else:
        avg = GSTATE.DELAY[0, ich - 1]
    return avg + evaldelay(x["oftype"], x["obw"] * 0.5) + evaldelay(x["eftype"], x["ebw"]) + post_delay


def rx_plan(chNum, RxParams, isy, max_frames=1, rng=None):
    """The plx_front plan + timing shifts of one RxPdmCohQpsk configuration, for batched use (pipeline.HotPath)."""
    if RxParams.get("rec") != "coherent":
        raise ValueError("Flag X.rec must be 'coherent'")
    hopt, elo, hel, post_delay, b2b = _front_tables(chNum, RxParams, rng)
    if RxParams.get("delay") != "theory" and "delay_symbols" not in RxParams:
        raise ValueError("a batched front-end plan needs its timing up front: RxParams.delay = 'theory' or "
                         "RxParams.delay_symbols (RxPdmCohQpsk measures it with corrdelay for a single frame)")
    dual = GSTATE.FIELDY is not None
    npol = 2 if (dual and isy) else 1
    if "delay_symbols" in RxParams:
        delay = np.ones(npol) * np.asarray(RxParams["delay_symbols"], dtype=float)
    else:
        delay = np.ones(npol) * theory_delay(chNum, RxParams, isy, post_delay)
    shifts = [int(_mround(-d * GSTATE.NT)) for d in delay]            # RxPdmCohQpsk.m:43
    r = RxParams["sps"] if RxParams.get("workatbaudrate") else RxParams["sps"] / 2   # :49-53
    if r != int(r) or r < 1:
        raise ValueError("the decimation rate must be a positive integer")
    r = int(r)
    fir = fir1_lowpass(16, 1.0 / r) if r > 1 else None
    nfft = GSTATE.FIELDX.shape[1]
    balanced = not (RxParams.get("pdtype") == "normal")
    adc = int(RxParams["adcbits"]) if RxParams.get("applyadc") else 0
    # a Y field created along propagation carries no pattern: only X is used then (RxPdmCohQpsk.m:20-33)
    fr = _Front(nfft, npol == 2, max_frames, hopt, elo, hel, balanced, adc, r, fir)
    return fr, shifts, dict(post_delay=post_delay, delay=delay, decim=r, fir=fir, hopt=hopt, elo=elo, hel=hel, b2b=b2b)


def DispCompFilter(Beta2L, B, N, FilterLength):
    """Hfilt = DispCompFilter(Beta2L, B, N, FilterLength)  RxPdmCohQpsk.m:90-98: frequency response of a
    FilterLength+1-tap FIR approximating the inverse of the accumulated dispersion (host table)."""
    k = np.arange(N)
    freq = -B / 2 + B / N * k                                        # (-B/2 : B/N : B/2*(N-2)/N)
    freq = np.fft.ifftshift(freq)
    delay = 2 * math.pi * freq / B * (FilterLength / 2)
    argum = (2 * math.pi * freq) ** 2 * Beta2L / 2 - delay
    H = np.cos(argum) + 1j * np.sin(argum)                           # fastexp
    b = np.fft.ifft(H)[: int(FilterLength) + 1]
    return np.fft.fft(b, N) * (np.cos(delay) + 1j * np.sin(delay))


def _apply_dcf(samples, p):
    """RxSamples = ifft(fft(RxSamples).*Hfilt) column-wise (RxPdmCohQpsk.m:74-84): samples torch [L, ncol] -> same."""
    import torch
    L, ncol = samples.shape
    Beta2L = -p["dispersion"] * p["lambda"] ** 2 / 2 / math.pi / CONSTANTS.CLIGHT * 1e-21
    sps = 1 + (0 if p.get("workatbaudrate") else 1)
    H = DispCompFilter(Beta2L, sps * p["baudrate"], L, p["ndispsym"] * sps)
    lib = _abi.get()
    hr, hi = np.ascontiguousarray(H.real), np.ascontiguousarray(H.imag)
    plan = C.c_void_p()
    lib.call("plx_filter_create", C.byref(plan), L, ncol, hr.ctypes.data, hi.ctypes.data)
    try:
        rows = samples.transpose(0, 1).contiguous()                  # [ncol, L]
        lib.call("plx_filter_apply_dev", plan, rows.data_ptr(), ncol, torch.cuda.current_stream().cuda_stream)
        torch.cuda.current_stream().synchronize()
    finally:
        lib.call("plx_filter_destroy", plan)
    return rows.transpose(0, 1).contiguous()


_QPSK_PHASE = np.array([-0.75 * math.pi, 0.75 * math.pi, -0.25 * math.pi, 0.25 * math.pi])   # RxPdmCohQpsk.m:118-121


def corrdelay(iric, pat, nt, nsymb, opt=None):
    """[delay, wrn, rho, Iric] = corrdelay(Iric, pat, Nt, Nsymb, opt)  corrdelay.m:62-116 -- system delay in symbols from
    the peak of the circular cross-correlation of the received current with the pattern held for Nt samples per
    symbol.  opt='phase': Iric complex, pat the symbol phases; 36 trial rotations remove the phase ambiguity (:75-90)
    and the 4th output is the angle of Iric moved next to the reference symbol.  Host diagnostic (numpy FFTs)."""
    if opt not in (None, "phase"):
        raise ValueError("wrong flag for opt.")
    MINERR, MAXERR, NPHI = 1e-4, 0.5, 36
    nfft = nsymb * nt
    iric = np.asarray(iric).reshape(-1)
    pat = np.asarray(pat, dtype=float).reshape(-1)
    if iric.size != nfft or pat.size != nsymb:
        raise ValueError("corrdelay: Iric must hold Nsymb*Nt samples and pat Nsymb symbols")
    if opt == "phase":
        ref = np.repeat(np.cos(pat) + 1j * np.sin(pat), nt)                        # every column of refsig, :76
        phi = np.linspace(0, 2 * math.pi, NPHI)
        rep = iric[:, None] * (np.cos(phi) + 1j * np.sin(phi))[None, :]            # :78
        cc = np.real(np.fft.ifft(np.fft.fft(rep, axis=0) * np.conj(np.fft.fft(ref))[:, None], axis=0))
        posc1 = np.argmax(cc, axis=0)                                              # :82-85 (first maximum, as MATLAB)
        col = int(np.argmax(cc[posc1, np.arange(NPHI)]))
        posc = int(posc1[col])
        maxc = cc[posc, col]
        out = np.angle(rep[:, col] * np.conj(ref)) + np.angle(ref)                 # :86-88
        c = cc[:, col]
    else:
        ref = np.repeat(pat, nt)                                                   # :91
        c = np.real(np.fft.ifft(np.fft.fft(iric) * np.conj(np.fft.fft(ref))))
        posc = int(np.argmax(c))
        maxc = c[posc]
        out = iric
    delay = float(posc)                                                            # nmod(posc,Nfft)-1, posc 1-based there
    ii = np.nonzero(np.diff(np.sign(np.diff(c))) == -2)[0] + 1                     # interior maxima, :101
    allmax = np.sort(c[ii])[::-1]
    wrn = False
    if allmax.size:
        first = ii[np.argmax(c[ii])]
        inderr = 1 if first == posc else 0                                         # :103-107
        if inderr < allmax.size and allmax[inderr] > 0 and maxc > 0:
            relerr = 10 * math.log10(maxc / allmax[inderr])
            wrn = MINERR < relerr < MAXERR                                         # :108-111
    return (delay + nt / 2) / nt, wrn, maxc / nfft * 2, out                        # :113-114


def mygeteyeinfo(irx, pat, delay=None, ts=None):
    """[eyeb, best_ts, delay, xopt] = mygeteyeinfo(ich, Iric, x, pat)  RxPdmCohQpsk.m:100-215 on the photocurrents irx
    [Nfft x 2 or 4] (after the ADC).  delay: the 'theory' delay in symbols (scalar or per polarisation, :124-133), or None
    -> measured by corrdelay(...,'phase') per polarisation (:134-137).  ts: fixed sampling time x.ts (:171-175), or None
    -> the instant of the widest worst-case eye with the three-point parabolic refinement (:176-201)."""
    irx = np.asarray(irx, dtype=float)
    pat = np.asarray(pat)
    if pat.ndim == 1:
        pat = pat.reshape(-1, 1)
    nt, nsymb = GSTATE.NT, GSTATE.NSYMB
    npol = irx.shape[1] // 2
    minv = np.full((nt, 4 * npol), np.nan)
    maxv = np.full((nt, 4 * npol), np.nan)
    dl = np.zeros(npol)
    if delay is not None:
        dl[:] = np.asarray(delay, dtype=float)
    for p in range(npol):
        ipat = pat[:, p].astype(int)
        cur = irx[:, 2 * p] + 1j * irx[:, 2 * p + 1]
        if delay is None:
            dl[p], _, _, ph = corrdelay(cur, _QPSK_PHASE[ipat], nt, nsymb, "phase")
        else:
            ph = np.angle(cur)                                                     # Iric_t, :131-132
        nshift = _mround(nt / 2 - dl[p] * nt)                                      # :139-141
        mat = np.roll(ph, nshift).reshape(nsymb, nt)                               # reshape(...,NT,NSYMB)' : rows = symbols
        for v in range(int(ipat.max()) + 1):                                       # :143-146
            sel = mat[ipat == v]
            if sel.size:
                minv[:, v + 4 * p] = sel.min(0)
                maxv[:, v + 4 * p] = sel.max(0)
    eyeop = np.full((nt, 4 * npol), np.nan)
    for p in range(npol):                                                          # :159-168
        o = 4 * p
        eyeop[:, o + 0] = minv[:, o + 2] - maxv[:, o + 0]
        eyeop[:, o + 1] = minv[:, o + 1] - maxv[:, o + 3]
        eyeop[:, o + 2] = minv[:, o + 3] - maxv[:, o + 2]
        eyeop[:, o + 3] = minv[:, o + 0] - (maxv[:, o + 1] - 2 * math.pi)
    if ts is not None:
        xopt = int(_mround((ts + 0.5) * nt))                                       # :172 (1-based row)
        eyeb = eyeop[xopt - 1].copy()
        eyeb[eyeb < 0] = np.nan                                                    # :174
        return eyeb, float(ts), dl, float(xopt)
    with np.errstate(all="ignore"):
        worst = np.nanmin(eyeop, axis=1)                                           # :169
    b = int(np.nanargmax(worst)) + 1                                               # best_tsn, 1-based (:177)
    eyeb = eyeop[b - 1].copy()
    if nt == 2:
        return eyeb, b / nt - 0.5, dl, float(b)
    to = (1, 2, 3) if b == 1 else ((b - 2, b - 1, b) if b == nt else (b - 1, b, b + 1))   # :183-189
    w = [worst[k - 1] for k in to]
    den = to[2] * (w[0] - w[1]) + to[0] * (w[1] - w[2]) + to[1] * (w[2] - w[0])
    num = to[2] ** 2 * (w[0] - w[1]) + to[0] ** 2 * (w[1] - w[2]) + to[1] ** 2 * (w[2] - w[0])
    with np.errstate(all="ignore"):
        xopt = 0.5 * np.float64(num) / np.float64(den)                             # :191-193 (0/0 -> NaN as there)
        l0 = (xopt - to[1]) * (xopt - to[2]) / ((to[0] - to[1]) * (to[0] - to[2]))
        l1 = (xopt - to[0]) * (xopt - to[2]) / ((to[1] - to[0]) * (to[1] - to[2]))
        l2 = (xopt - to[0]) * (xopt - to[1]) / ((to[2] - to[0]) * (to[2] - to[1]))
        eyeb = l0 * eyeop[to[0] - 1] + l1 * eyeop[to[1] - 1] + l2 * eyeop[to[2] - 1]   # :195-199
    return eyeb, float(xopt / nt - 0.5), dl, float(xopt)


def _worst_eye(eyeb):
    with np.errstate(invalid="ignore"):
        ok = np.abs(eyeb - np.mod(eyeb, math.pi / 2)) < 1e-10                      # RxPdmCohQpsk.m:87
    return float(np.min(eyeb[ok])) if ok.any() else float("nan")


def eye_opening(irx, pat, delay, ts=0.0):
    """worsteyeop of RxPdmCohQpsk.m:87 / dsp4cohdec.m:283: the smallest of mygeteyeinfo's per-level phase-eye openings
    that lies in [0, pi/2).  delay None -> corrdelay, ts None -> best sampling instant (see mygeteyeinfo)."""
    return _worst_eye(mygeteyeinfo(irx, pat, delay, ts)[0])


def _adc_numpy(irx, bits):
    M = np.max(np.abs(irx))
    q = (irx + M) / 2 / M * 2 ** bits
    fl = np.floor(q)
    return (fl + ((q - fl) >= 0.5)) * 2 * M / 2 ** bits - M                        # RxPdmCohQpsk.m:36-40


def _mround(v):
    """MATLAB round (half away from zero)."""
    return math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1)


def RxPdmCohQpsk(chNum, symbolPattern, RxParams, rng=None):
    """[RxSamples, worsteyeop] = RxPdmCohQpsk(chNum, symbolPattern, RxParams)  RxPdmCohQpsk.m:3-87.
    RxSamples: torch complex128 [nout, 1 or 2] on the GPU.  The front end (hybrid, photodiodes, filters, ADC, timing,
    decimation) runs on the device; mygeteyeinfo (:100-215) is a host diagnostic on the downloaded photocurrents.  With
    RxParams.delay == 'theory' the timing is known up front and the eye is evaluated only on request (RxParams.ts, or
    RxParams.evaleye for the best-instant search); otherwise the delay is MEASURED as the reference does (corrdelay on
    the currents, :134-137): one device pass for the currents, the host correlation, a second pass with the timing.
    RxParams.applydcf runs the DispCompFilter response (:74-98) through the device FFT engine (sample counts that are
    powers of two >= 256)."""
    import torch
    sp = np.asarray(symbolPattern)
    isy = GSTATE.FIELDY is not None and sp.ndim == 2 and sp.shape[1] != 1      # :27-33
    measured = RxParams.get("delay") != "theory" and "delay_symbols" not in RxParams
    ts = float(RxParams["ts"]) if "ts" in RxParams else None

    def currents(fr, ux, uy):
        cols = [ux[0].real, ux[0].imag] + ([uy[0].real, uy[0].imag] if fr.dual else [])   # ux/uy now hold the currents
        cur = torch.stack(cols, 1).cpu().numpy()
        if RxParams.get("applyadc"):
            cur = _adc_numpy(cur, int(RxParams["adcbits"]))
        return cur

    eye = float("nan")
    if measured:
        if GSTATE.NSYMB * GSTATE.NT != GSTATE.FIELDX.shape[1]:
            raise ValueError("corrdelay needs the whole NSYMB*NT frame")
        rng = rng or np.random.default_rng()
        state = rng.bit_generator.state                                # both passes see the same LO phase noise
        fr, _, info = rx_plan(chNum, dict(RxParams, delay_symbols=0.0), isy, 1, rng)
        rng.bit_generator.state = state
        try:
            ux, uy = _channel_fields(chNum, info["b2b"])
            fr.run(ux, uy if fr.dual else None, [0] * (2 if fr.dual else 1))
            eyeb, _, delay, _ = mygeteyeinfo(currents(fr, ux, uy), sp, None, ts)      # :41
        finally:
            fr.close()
        eye = _worst_eye(eyeb)
        RxParams = dict(RxParams, delay_symbols=delay)
    fr, shifts, info = rx_plan(chNum, RxParams, isy, 1, rng)
    try:
        ux, uy = _channel_fields(chNum, info["b2b"])
        out = fr.run(ux, uy if fr.dual else None, shifts)
        if not measured and (ts is not None or RxParams.get("evaleye")) and GSTATE.NSYMB * GSTATE.NT == ux.shape[1]:
            eye = eye_opening(currents(fr, ux, uy), sp, info["delay"], ts)
    finally:
        fr.close()
    samples = out[0].transpose(0, 1).contiguous()
    if RxParams.get("applydcf"):
        samples = _apply_dcf(samples, RxParams)                       # :74-84
    return samples, eye


def dsp4cohdec(ich, pat, x, p, rng=None):
    """[Phases, Amplitudes, worsteyeop] = dsp4cohdec(ich, pat, x, p)  dsp4cohdec.m:100-282 -- Optilux's own coherent
    receiver + DSP (ex19/ex20): receiver_cohmix with the receiver struct x, ADC / timing / decimation with the DSP
    struct p (:121-160), then the DSP body that DspPdmCohQpsk shares with it (:212-282).  Phases/Amplitudes are torch
    float64 [Nsymb, 1 or 2] on the GPU; worsteyeop as RxPdmCohQpsk returns it."""
    import torch
    from .rx import DspPdmCohQpsk
    rxp = dict(x)
    for k in ("sps", "workatbaudrate", "applyadc", "adcbits", "applydcf", "dispersion", "ndispsym", "baudrate", "lambda"):
        if k in p:
            rxp[k] = p[k]
    samples, eye = RxPdmCohQpsk(ich, pat, rxp, rng)                    # dsp4cohdec.m:108-160 = RxPdmCohQpsk.m:18-72
    sig = DspPdmCohQpsk(samples.transpose(0, 1), p, ich)               # [ncol, Nsymb]
    return torch.angle(sig).transpose(0, 1).contiguous(), torch.abs(sig).transpose(0, 1).contiguous(), eye

"""Global simulation state: the GSTATE / CONSTANTS structs of reset_all.m.

Mirror of /root/reference/reset_all.m:104-174 for the fields the hot path reads
or writes.  The optical field lives in HBM: FIELDX / FIELDY are torch complex128
CUDA tensors of shape [nfc, nfft] (one contiguous row per MATLAB column), so that
``fiber`` works in place on the device exactly as the reference works in place on
the global.  The print/file side of reset_all (:176-227) is out of scope.
"""
import numpy as np


class _Struct:
    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join("%s=%r" % kv for kv in sorted(vars(self).items())
                                                          if not kv[0].startswith("FIELD")))


class _GState(_Struct):
    def __init__(self):
        self.clear()

    def clear(self):
        self.NSYMB = self.NT = self.NCH = 0
        self.FN = None
        self.SYMBOLRATE = None
        self.FIELDX = self.FIELDY = None
        self.FIELDX_TX = self.FIELDY_TX = None
        self.DELAY = None
        self.DISP = None
        self.LAMBDA = None
        self.POWER = None
        self.PRINT = False
        self.DIR = None


class _Constants(_Struct):
    CLIGHT = 299792458.0          # reset_all.m:105
    HPLANCK = 6.62606896e-34      # :108
    ECHARGE = 1.602176487e-19
    KBOLTZMANN = 1.3806504e-23


GSTATE = _GState()
CONSTANTS = _Constants()


def reset_all(Nsymb, Nt, Nch, *_ignored):
    """reset_all(Nsymb,Nt,Nch[,dir][,'noprint']) -- reset_all.m:104-174 (state only)."""
    GSTATE.clear()
    GSTATE.NSYMB, GSTATE.NT, GSTATE.NCH = int(Nsymb), int(Nt), int(Nch)
    stepf = 1.0 / Nsymb
    fn = np.arange(-Nt / 2, Nt / 2, stepf)[: int(Nsymb) * int(Nt)]
    GSTATE.FN = np.fft.fftshift(fn)                       # :152-153
    GSTATE.DELAY = np.zeros((1, Nch))
    GSTATE.DISP = np.zeros((1, Nch))
    GSTATE.PRINT = False
    return GSTATE


def _torch():
    import torch
    return torch


def device():
    torch = _torch()
    if not torch.cuda.is_available():
        from ._abi import PolmuxError, PLX_ERR_HIP
        raise PolmuxError(PLX_ERR_HIP, "polmux_amd needs an MI355X (torch.cuda is not available); there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def to_device_field(a):
    """numpy [nfft, nfc] (MATLAB column-major semantics) or torch tensor -> torch complex128 [nfc, nfft] on the GPU."""
    torch = _torch()
    if isinstance(a, torch.Tensor):
        t = a
        if t.dtype != torch.complex128:
            t = t.to(torch.complex128)
        return t.to(device()).contiguous()
    a = np.asarray(a, dtype=np.complex128)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    return torch.from_numpy(np.ascontiguousarray(a.T)).to(device())


def to_host_field(t):
    """torch [nfc, nfft] -> numpy [nfft, nfc]"""
    return t.detach().cpu().numpy().T.copy()


def create_field(ftype, sigx, sigy=None, options=None, rng=None):
    """create_field('sepfields'|'unique', sigx, sigy, options) -- create_field.m:100-199: average-power
    normalisation (:113-124), options.delay (:127-146), 'sepfields' (:156-163) and 'unique' (one field with the
    channels at their carrier offsets, :165-199; Tx-side host work).  The fields end up in HBM."""
    sigx = np.asarray(sigx, dtype=np.complex128)
    if sigx.ndim == 1:
        sigx = sigx.reshape(-1, 1)
    GSTATE.FIELDY = GSTATE.FIELDY_TX = None
    isy = sigy is not None and np.size(sigy) > 0
    if isy:
        sigy = np.asarray(sigy, dtype=np.complex128).reshape(sigx.shape)
    if sigx.shape[1] != GSTATE.NCH:
        raise ValueError("the number of columns of sigx,sigy must be equal to the number of channels")
    if options and str(options.get("power", "")).lower() == "average":
        avge = np.mean(np.abs(sigx) ** 2 + (np.abs(sigy) ** 2 if isy else 0), axis=0)
        k = np.sqrt(np.asarray(GSTATE.POWER, dtype=float) / avge)
        sigx = sigx * k
        if isy:
            sigy = sigy * k
        GSTATE.POWER = np.asarray(GSTATE.POWER, dtype=float) ** 2 / avge
    npol = 2 if isy else 1
    if options and options.get("delay") is not None:                                  # create_field.m:127-146
        if isinstance(options["delay"], str):
            if options["delay"] != "rand":
                raise ValueError("options.delay must be 'rand' or numeric")
            tau = np.round((rng or np.random.default_rng()).random((npol, GSTATE.NCH)) * GSTATE.NT)
        else:
            d = np.atleast_2d(np.asarray(options["delay"], dtype=float))
            if d.shape != (npol, GSTATE.NCH):
                raise ValueError("the delay must be of size [number of polarizations,number of channels]")
            tau = np.sign(d) * np.floor(np.abs(d * GSTATE.NT) + 0.5)                   # MATLAB round
        sigx = sigx.copy()
        sigy = sigy.copy() if isy else sigy
        for k in range(GSTATE.NCH):
            sigx[:, k] = np.roll(sigx[:, k], int(tau[0, k]))                          # fastshift(x, n): y(i) = x(i-n)
            if isy:
                sigy[:, k] = np.roll(sigy[:, k], int(tau[1, k]))
        GSTATE.DELAY = tau
    else:
        GSTATE.DELAY = np.zeros((npol, GSTATE.NCH))
    GSTATE.DISP = np.zeros((npol, GSTATE.NCH))
    if ftype.lower() == "sepfields":
        GSTATE.FIELDX = to_device_field(sigx)
        GSTATE.FIELDX_TX = GSTATE.FIELDX.clone()
        if isy:
            GSTATE.FIELDY = to_device_field(sigy)
            GSTATE.FIELDY_TX = GSTATE.FIELDY.clone()
    elif ftype.lower() == "unique":
        # one field carrying every channel at its own carrier: spectra shifted by ndfn bins and summed
        # (create_field.m:165-199; Tx-side, O(Nch N log N) once, on the host)
        ndfn = unique_field_shifts()
        zx = np.zeros(sigx.shape[0], dtype=np.complex128)
        zy = np.zeros_like(zx)
        for k in range(GSTATE.NCH):
            zx = zx + np.roll(np.fft.fft(sigx[:, k]), -int(ndfn[k]))                   # fastshift(z, -ndfn(kch))
            if isy:
                zy = zy + np.roll(np.fft.fft(sigy[:, k]), -int(ndfn[k]))
        GSTATE.FIELDX = to_device_field(np.fft.ifft(zx))
        GSTATE.FIELDX_TX = GSTATE.FIELDX.clone()
        if isy:
            GSTATE.FIELDY = to_device_field(np.fft.ifft(zy))
            GSTATE.FIELDY_TX = GSTATE.FIELDY.clone()
    else:
        raise ValueError("the field type must be 'unique' or 'sepfields'")


def unique_field_shifts():
    """ndfn of create_field.m:181-184 / receiver_cohmix.m:104-107: carrier offsets of the channels from the central
    wavelength, in frequency bins of GSTATE.FN."""
    lamt = np.atleast_1d(np.asarray(GSTATE.LAMBDA, dtype=float))
    maxl, minl = lamt.max(), lamt.min()
    fnyqmin = (CONSTANTS.CLIGHT / minl - CONSTANTS.CLIGHT / maxl) / GSTATE.SYMBOLRATE
    if GSTATE.NT < fnyqmin and fnyqmin != 0:
        raise ValueError("number of samples per symbol is too small")               # :170-179 (the prompt is not asked)
    lamc = 2 * maxl * minl / (maxl + minl)
    deltafn = CONSTANTS.CLIGHT * (1 / lamc - 1.0 / lamt)
    minfreq = GSTATE.FN[1] - GSTATE.FN[0]
    v = deltafn / GSTATE.SYMBOLRATE / minfreq
    return (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(np.int64)                # MATLAB round


def lasersource(Ptx, lam, spac=None):
    """GSTATE.LAMBDA / GSTATE.POWER bookkeeping of lasersource.m:150-178; returns the CW carriers."""
    nch = GSTATE.NCH
    Pin = np.full(nch, float(Ptx)) if np.size(Ptx) == 1 else np.asarray(Ptx, dtype=float)
    if np.size(lam) == 1 and nch > 1:
        if spac is None:
            raise ValueError("missing the channel-spacing SPAC")
        lamt = np.array([lam + spac * (ch - (nch + 1) / 2) for ch in range(1, nch + 1)], dtype=float)
    else:
        lamt = np.atleast_1d(np.asarray(lam, dtype=float))
        if lamt.size != nch:
            raise ValueError("wrong length for LAM (must be 1 or # channels)")
    GSTATE.LAMBDA = lamt
    GSTATE.POWER = Pin
    return np.ones((GSTATE.NSYMB * GSTATE.NT, 1)) * np.sqrt(Pin)

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


def create_field(ftype, sigx, sigy=None, options=None):
    """create_field('sepfields'|'unique', sigx, sigy, struct('power','average')) -- the part of
    create_field.m:100-199 the hot path needs: average-power normalisation (:113-124) and the
    'sepfields' assignment (:156-163).  'unique' (frequency-shifted sum, :165-199) is Tx-side and
    stays on the host."""
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
    GSTATE.DELAY = np.zeros((npol, GSTATE.NCH))
    GSTATE.DISP = np.zeros((npol, GSTATE.NCH))
    if ftype.lower() == "sepfields" or GSTATE.NCH == 1:
        GSTATE.FIELDX = to_device_field(sigx)
        GSTATE.FIELDX_TX = GSTATE.FIELDX.clone()
        if isy:
            GSTATE.FIELDY = to_device_field(sigy)
            GSTATE.FIELDY_TX = GSTATE.FIELDY.clone()
    else:
        raise NotImplementedError("create_field('unique') with several channels is Tx-side host code outside the "
                                  "accelerated path (create_field.m:165-199)")


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

"""ampliflat(x, atype, options): ideal optical amplifier with ASE noise -- ampliflat.m:1-148.

"Next" row of the scope table: the step between fibre spans.  Gain and noise are
applied in place to GSTATE.FIELDX/FIELDY in HBM by k_ampliflat; options.noise (the reference's own
injection hook, ampliflat.m:123-129) is honoured, otherwise the ASE comes from a counter-based
Philox stream on the device (MATLAB's legacy randn('state') stream cannot be reproduced).
"""
import math

import numpy as np

from . import _abi
from .gstate import CONSTANTS, GSTATE


def ase_sigma(f_db, gain, nfc):
    """sigma of ampliflat.m:91-102 [sqrt(mW)] per column."""
    Flin = 10 ** (f_db * 0.1)
    lam = np.atleast_1d(np.asarray(GSTATE.LAMBDA, dtype=float))
    if nfc == 1:
        maxl, minl = lam.max(), lam.min()
        lam = np.array([2 * maxl * minl / (maxl + minl)])
    return np.sqrt(Flin / 4 * CONSTANTS.HPLANCK * CONSTANTS.CLIGHT / lam * (gain - 1) * GSTATE.NT * GSTATE.SYMBOLRATE * 1e21)


def ampliflat(x, atype, options=None, seed=0, keys=None):
    import torch
    fx = GSTATE.FIELDX
    if fx is None:
        raise ValueError("create_field must be called before ampliflat")
    nfc, nfr = fx.shape[-2], fx.shape[-1]
    atype = str(atype).lower()
    if atype == "gain":
        gain = 10 ** (x * 0.1)                                               # :62-63
    elif atype == "fixpower":
        if nfc != GSTATE.NCH:
            raise ValueError("'fixpower' works only for channels separated")   # :70-72
        mid = int(math.ceil(nfc / 2)) - 1
        avge = float((fx[..., mid, :].abs() ** 2).mean())
        if GSTATE.FIELDY is not None:
            avge += float((GSTATE.FIELDY[..., mid, :].abs() ** 2).mean())
        gain = x / avge                                                       # avg_power(midch,'abs') :66-68
    else:
        raise ValueError("wrong string atype")                                # :75
    sigma = None
    noise = None
    asex = asey = 1
    if options is not None:                                                   # :87-148
        f = options.get("f")
        if f is not None and not math.isinf(f):
            sigma = np.ascontiguousarray(ase_sigma(f, gain, nfc), dtype=float)
            if not sigma.any():
                sigma = None
        if sigma is not None:
            one = options.get("onepol")
            if one is not None:
                if str(one).lower() == "asex":
                    asey = 0
                elif str(one).lower() == "asey":
                    asex = 0
                else:
                    raise ValueError("ONEPOL, if exists, must be 'asex' or 'asey'")
            if options.get("noise") is not None:
                n = np.asarray(options["noise"], dtype=np.complex128)         # [nfr x 2*nfc]: [X cols | Y cols]
                if n.shape != (nfr, 2 * nfc):
                    raise ValueError("options.noise must have the size of [GSTATE.FIELDX, GSTATE.FIELDY]")
                noise = torch.from_numpy(np.ascontiguousarray(n.T)).to(fx.device)
    fy = GSTATE.FIELDY
    if sigma is not None and asey and fy is None:                            # :139-142: FIELDY = noiseY
        fy = GSTATE.FIELDY = torch.zeros_like(fx)
        GSTATE.DELAY = np.vstack([np.atleast_2d(GSTATE.DELAY)[:1], np.zeros((1, GSTATE.NCH))])
    lib = _abi.get()
    frames = int(np.prod(fx.shape[:-2])) if fx.dim() > 2 else 1
    kt = None
    if keys is not None:
        kt = torch.as_tensor(np.asarray(keys, dtype=np.int64), device=fx.device)
    lib.call("plx_ampliflat_dev", fx.data_ptr(), fy.data_ptr() if fy is not None else None, nfr, nfc, frames, float(gain),
             sigma.ctypes.data if sigma is not None else None, noise.data_ptr() if noise is not None else None,
             int(seed) & (2 ** 64 - 1), kt.data_ptr() if kt is not None else None, asex, asey,
             torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().synchronize()
    return gain

"""Receiver-side functions of the hot path, mirroring the reference signatures:

  fastexp(x)                                              fastexp.m:28 / fastexp.c
  CDE_OFDE(inX, inY, fs, lambdaRef, span, D, S, fftLength, L)      CDE_OFDE.m:16-47
  cmaadaptivefilter(xx, h1, h2, taps, mu, R, sps)         cmaadaptivefilter.c:93-174
  easiadaptivefilter(xx, h1, h2, taps, mu, sps)           easiadaptivefilter.c:95-169
  DspPdmCohQpsk(RxSamples4D, dspParams, chNum)            DspPdmCohQpsk.m:3-84
  samp2pat(x, s, outvalue)  ('coherent' only)             samp2pat.m:61-66

Arrays follow MATLAB shapes ([samples x columns]); numpy in -> numpy out through
the gateway tier of the C ABI, torch CUDA tensors ([columns, samples] rows) in ->
torch out through the resident tier.  No CPU implementation exists here.
"""
import ctypes as C
import math

import numpy as np

from . import _abi
from .gstate import GSTATE


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _split(z):
    z = np.asarray(z)
    return _f(z.real.copy()), _f(z.imag.copy() if np.iscomplexobj(z) else np.zeros(z.shape))


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------ fastexp ---
def fastexp(x):
    """y = exp(i*x), x real."""
    lib = _abi.get()
    if _is_torch(x):
        import torch
        xx = x.to(torch.float64).contiguous()
        y = torch.empty(xx.shape, dtype=torch.complex128, device=xx.device)
        lib.call("plx_fastexp_dev", xx.data_ptr(), y.data_ptr(), xx.numel(), _stream())
        return y
    x = _f(np.real(x))
    yr, yi = np.empty_like(x), np.empty_like(x)
    lib.call("plx_fastexp", x.ctypes.data, yr.ctypes.data, yi.ctypes.data, x.size)
    return yr + 1j * yi


# ----------------------------------------------------------------- CDE_OFDE ---
def cde_transfer(fftLength, samplingRateIn, lambdaRef, span, D, S):
    """H on the fftshift-ordered grid, CDE_OFDE.m:21-38."""
    c = 299792458.0
    fc = c / lambdaRef
    df = 1.0 / (fftLength / samplingRateIn)
    fg = df * np.arange(-fftLength // 2, fftLength // 2)
    H_D = -1j * D * span * math.pi * c / fc ** 2 * fg ** 2
    H_S = 1j * S * span * math.pi * c ** 2 / 3 / fc ** 4 * fg ** 3
    return np.exp(H_D + H_S)


_CDE_CHECKS = ("H must be even length", "L must be > 0", "L must be shorter than filter length",
               "Signal must be longer or equal filter")


def CDE_OFDE(inCompIxQx, inCompIyQy, samplingRateIn, lambdaRef, span, D, S, fftLength, L):
    """[outX, outY, samplingRateOut] = CDE_OFDE(...).  Like OverlapBothTrans (CDE_OFDE.m:63-85) a bad
    argument prints 'Error: ...' and yields empty outputs instead of raising."""
    lib = _abi.get()
    try:
        if _is_torch(inCompIxQx):
            import torch
            x = inCompIxQx.reshape(-1).to(torch.complex128).contiguous()
            y = inCompIyQy.reshape(-1).to(torch.complex128).contiguous()
            nx = x.numel()
            N = min(int(fftLength), nx)                                      # :24-27
            H = np.ascontiguousarray(cde_transfer(N, samplingRateIn, lambdaRef, span, D, S)).view(np.float64)
            plan = C.c_void_p()
            lib.call("plx_cde_create", C.byref(plan), N, int(L), H.ctypes.data)
            try:
                xy = torch.stack([x, y])
                out = torch.empty_like(xy)
                lib.call("plx_cde_apply_dev", plan, xy.data_ptr(), out.data_ptr(), nx, 2, _stream())
                torch.cuda.current_stream().synchronize()
            finally:
                lib.call("plx_cde_destroy", plan)
            return out[0].reshape(inCompIxQx.shape), out[1].reshape(inCompIyQy.shape), samplingRateIn
        x = np.asarray(inCompIxQx).reshape(-1)
        y = np.asarray(inCompIyQy).reshape(-1)
        xr, xi = _split(x)
        yr, yi = _split(y)
        outs = [np.zeros(x.size) for _ in range(4)]
        lib.call("plx_cde_ofde", xr.ctypes.data, xi.ctypes.data, yr.ctypes.data, yi.ctypes.data, x.size,
                 float(samplingRateIn), float(lambdaRef), float(span), float(D), float(S), int(fftLength), int(L),
                 *[o.ctypes.data for o in outs])
        shp = np.asarray(inCompIxQx).shape
        return (outs[0] + 1j * outs[1]).reshape(shp), (outs[2] + 1j * outs[3]).reshape(shp), samplingRateIn
    except _abi.PolmuxError as e:
        if e.code == _abi.PLX_ERR_ARG and any(m in str(e) for m in _CDE_CHECKS):
            print(str(e))                                                     # display('Error: ...'); return
            return np.zeros(0, complex), np.zeros(0, complex), samplingRateIn
        raise


# ------------------------------------------------------- CMA / EASI gateways ---
def _filter_gateway(name, xx, h1, h2, taps, mu, R, sps):
    """The MEX contract: h1 and h2 are updated IN PLACE when they are complex numpy arrays of the
    right shape (the reference writes through prhs[1..2]) and the 2nd/3rd outputs are 0."""
    lib = _abi.get()
    xx = np.asarray(xx)
    Mdim = xx.shape[0]
    xr, xi = _split(xx)
    h1r, h1i = _split(h1)
    h2r, h2i = _split(h2)
    ntap = int(taps)
    dimY = max(Mdim - ntap + 1, 0)
    yr, yi = np.zeros((dimY, 2), order="F"), np.zeros((dimY, 2), order="F")
    if name == "cma":
        Rv = np.ascontiguousarray(np.atleast_1d(R), dtype=float)
        lib.call("plx_cmaadaptivefilter", xr.ctypes.data, xi.ctypes.data, Mdim, h1r.ctypes.data, h1i.ctypes.data,
                 h2r.ctypes.data, h2i.ctypes.data, float(taps), float(mu), Rv.ctypes.data, float(sps),
                 yr.ctypes.data, yi.ctypes.data)
    else:
        lib.call("plx_easiadaptivefilter", xr.ctypes.data, xi.ctypes.data, Mdim, h1r.ctypes.data, h1i.ctypes.data,
                 h2r.ctypes.data, h2i.ctypes.data, float(taps), float(mu), float(sps), yr.ctypes.data, yi.ctypes.data)
    for h, hr, hi in ((h1, h1r, h1i), (h2, h2r, h2i)):
        if isinstance(h, np.ndarray) and np.iscomplexobj(h) and h.shape == hr.shape:
            h[...] = hr + 1j * hi
    return yr + 1j * yi, 0.0, 0.0


def cmaadaptivefilter(xx, h1, h2, taps, mu, R, sps):
    """[Y,h1,h2] = cmaadaptivefilter(xx,h1,h2,taps,mu,R,sps): returns (Y, 0, 0); h1,h2 mutated."""
    return _filter_gateway("cma", xx, h1, h2, taps, mu, R, sps)


def easiadaptivefilter(xx, h1, h2, taps, mu, sps):
    """[Y,h1,h2] = easiadaptivefilter(xx,h1,h2,taps,mu,sps): returns (Y, 0, 0); h1,h2 mutated."""
    return _filter_gateway("easi", xx, h1, h2, taps, mu, None, sps)


def _twin_gateway(name, xx, h1, h2, mu, R):
    """The .m twins (no MEX compiled): [Y h1 h2] with the UPDATED taps returned (cmaadaptivefilter.m:1,
    easiadaptivefilter.m:1); taps/sps arguments are unused there."""
    lib = _abi.get()
    xx = np.asarray(xx)
    Mdim = xx.shape[0]
    xr, xi = _split(xx)
    g1 = np.asarray(h1, dtype=np.complex128).reshape(-1, 2)
    g2 = np.asarray(h2, dtype=np.complex128).reshape(-1, 2)
    ntap = g1.shape[0]
    h1r, h1i = (np.asfortranarray(g1.real.copy()), np.asfortranarray(g1.imag.copy()))
    h2r, h2i = (np.asfortranarray(g2.real.copy()), np.asfortranarray(g2.imag.copy()))
    dimY = max(Mdim - ntap + 1, 0)
    yr, yi = np.zeros((dimY, 2), order="F"), np.zeros((dimY, 2), order="F")
    if name == "cma":
        Rv = np.ascontiguousarray(np.atleast_1d(R), dtype=float)
        lib.call("plx_cmaadaptivefilter_m", xr.ctypes.data, xi.ctypes.data, Mdim, h1r.ctypes.data, h1i.ctypes.data,
                 h2r.ctypes.data, h2i.ctypes.data, ntap, float(mu), Rv.ctypes.data, yr.ctypes.data, yi.ctypes.data)
    else:
        lib.call("plx_easiadaptivefilter_m", xr.ctypes.data, xi.ctypes.data, Mdim, h1r.ctypes.data, h1i.ctypes.data,
                 h2r.ctypes.data, h2i.ctypes.data, ntap, float(mu), yr.ctypes.data, yi.ctypes.data)
    return yr + 1j * yi, h1r + 1j * h1i, h2r + 1j * h2i


def cmaadaptivefilter_m(xx, h1, h2, taps, mu, R, sps=1):
    """[Y h1 h2] = cmaadaptivefilter(...) as the .m twin computes it (cmaadaptivefilter.m:52-72): every sample updates,
    the updated taps are returned."""
    return _twin_gateway("cma", xx, h1, h2, mu, R)


def easiadaptivefilter_m(xx, h1, h2, taps, mu, sps=1):
    """[Y h1 h2] = easiadaptivefilter(...) as the .m twin computes it (easiadaptivefilter.m:51-84)."""
    return _twin_gateway("easi", xx, h1, h2, mu, None)


# ------------------------------------------------------------- DspPdmCohQpsk ---
_POLMETHOD = {"singlepol": 0, "cma": 1, "easi": 2, "combo": 3}


def _g(d, name, default=None):
    if isinstance(d, dict):
        return d.get(name, default)
    return getattr(d, name, default)


def dsp_params_struct(dspParams, power_mw):
    p = _abi.DspParams()
    p.workatbaudrate = int(bool(_g(dspParams, "workatbaudrate", False)))
    p.applynlr = int(bool(_g(dspParams, "applynlr", False)))
    p.nlralpha = float(_g(dspParams, "nlralpha", 0.0))
    p.power_mw = float(power_mw)
    p.applypol = int(bool(_g(dspParams, "applypol", False)))
    method = str(_g(dspParams, "polmethod", "cma")).lower()
    if p.applypol and method not in _POLMETHOD:
        raise ValueError("Unknown Polar Rotation method.")                   # DspPdmCohQpsk.m:40
    p.polmethod = _POLMETHOD.get(method, 1)
    cma = _g(dspParams, "cmaparams", {}) or {}
    easi = _g(dspParams, "easiparams", {}) or {}
    R = np.atleast_1d(_g(cma, "R", [1.0, 1.0])).astype(float)
    p.cma_R[0], p.cma_R[1] = R[0], R[-1]
    p.cma_mu = float(_g(cma, "mu", 1 / 6000))
    p.cma_taps = int(_g(cma, "taps", 7))
    p.cma_txpolars = int(_g(cma, "txpolars", 2))
    p.cma_phizero = float(_g(cma, "phizero", 0.0))
    p.easi_mu = float(_g(easi, "mu", 1 / 6000))
    p.easi_txpolars = int(_g(easi, "txpolars", 2))
    p.easi_phizero = float(_g(easi, "phizero", 0.0))
    for prm, has, dst in ((cma, "cma_has_mat", p.cma_mat), (easi, "easi_has_mat", p.easi_mat)):
        m = _g(prm, "mat")                                                   # explicit initial matrix, DspPdmCohQpsk.m:148-149
        if m is not None:
            m = np.asarray(m, dtype=np.complex128)
            if m.shape != (2, 2):
                raise ValueError("params.mat must be a 2x2 matrix")
            setattr(p, has, 1)
            for k, v in enumerate(m.reshape(-1)):
                dst[2 * k], dst[2 * k + 1] = v.real, v.imag
    # no MEX compiled (comp_mex.m not run): MATLAB resolves the filter names to the .m twins, whose EASI differs
    p.mfile_twins = int(bool(_g(dspParams, "mfiletwins", False)))
    p.modorder = int(_g(dspParams, "modorder", 2))
    p.freqavg = int(_g(dspParams, "freqavg", 0))
    p.phasavg = int(_g(dspParams, "phasavg", 0))
    p.poworder = int(_g(dspParams, "poworder", 2))
    return p


def DspPdmCohQpsk(RxSamples4D, dspParams, chNum=1):
    """DSP for a PolMUX coherent QPSK signal (DspPdmCohQpsk.m:3).  chNum is 1-based."""
    import torch
    lib = _abi.get()
    power = float(np.atleast_1d(GSTATE.POWER)[chNum - 1])
    p = dsp_params_struct(dspParams, power)
    if _is_torch(RxSamples4D):
        x = RxSamples4D.to(torch.complex128).contiguous()                    # [ncol, Lin] or [frames, ncol, Lin]
        batched = x.dim() == 3
    else:
        a = np.asarray(RxSamples4D, dtype=np.complex128)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        from .gstate import device
        x = torch.from_numpy(np.ascontiguousarray(a.T)).to(device())
        batched = False
    xb = x if batched else x.unsqueeze(0)
    frames, ncol, Lin = xb.shape
    plan = C.c_void_p()
    lib.call("plx_dsp_create", C.byref(plan), Lin, ncol, frames, C.byref(p))
    try:
        Lout = lib.lib.plx_dsp_out_len(plan)
        out = torch.empty((frames, ncol, Lout), dtype=torch.complex128, device=xb.device)
        lib.call("plx_dsp_run_dev", plan, xb.data_ptr(), out.data_ptr(), frames, _stream())
        torch.cuda.current_stream().synchronize()
    finally:
        lib.call("plx_dsp_destroy", plan)
    if _is_torch(RxSamples4D):
        return out if batched else out[0]
    return out[0].cpu().numpy().T.copy()


def samp2pat(x, s, outvalue):
    """pat_rx = samp2pat(x, s, outvalue) for x.rec == 'coherent' (samp2pat.m:61-66).  outvalue are
    phases [L x ncol]; pure host decision logic on already-downloaded phases (the device path is
    plx_decide_count_dev, which also counts errors)."""
    if _g(x, "rec") != "coherent":
        raise ValueError("Wrong modulation format\n")
    v = np.asarray(outvalue, dtype=float)
    if v.ndim == 1:
        v = v.reshape(-1, 1)
    second = v > 0
    first = np.abs(v) <= math.pi / 2
    if v.shape[1] == 1:
        return np.concatenate([first, second], axis=1).astype(np.uint8)
    return np.stack([first[:, 0], second[:, 0], first[:, 1], second[:, 1]], axis=1).astype(np.uint8)

"""ctypes binding of the CPU ORACLE (oracle/libplxo) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/plxo.h).  The product package polmux_amd never does.

Arrays follow MATLAB conventions: fields are complex128 [nfft, nfc] matrices in
column-major (Fortran) order.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libplxo.so")


def build(force=False):
    import fcntl
    srcs = [os.path.join(_HERE, f) for f in ("plxo_fiber.c", "plxo_rx.c", "plxo_mc.c", "plxo.h")]

    def fresh():
        return os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)
    if not force and fresh():
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    with open(_LIB + ".lock", "w") as lk:         # (parallel test workers: one builds, the others wait and find it fresh)
        fcntl.flock(lk, fcntl.LOCK_EX)
        if force or not fresh():
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.plxo_nextstep.restype = C.c_double
        _lib.plxo_erfcinv.restype = C.c_double
        _lib.plxo_erfcinv.argtypes = [C.c_double]
        _lib.plxo_dsp_pdm_coh_qpsk.restype = C.c_long
        _lib.plxo_nmod.restype = C.c_long
    return _lib


def _c(a):
    """complex128 Fortran-contiguous copy"""
    return np.array(a, dtype=np.complex128, order="F", copy=True)


def _d(a):
    return np.array(a, dtype=np.float64, order="F", copy=True)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _fls(fls):
    return (C.c_int * 4)(*[int(v) for v in fls])


# ------------------------------------------------------------------ basics ---
def fft(x, inverse=False):
    y = _c(x).reshape(-1)
    lib().plxo_fft(_p(y), C.c_long(y.size), C.c_int(int(inverse)))
    return y


def fastexp(x):
    x = _d(x)
    yr = np.empty_like(x)
    yi = np.empty_like(x)
    lib().plxo_fastexp(_p(x), _p(yr), _p(yi), C.c_long(x.size))
    return yr + 1j * yi


def nextstep(dzmax, phimax, gam, alphalin, ux, uy=None):
    ux = _c(ux)
    if ux.ndim == 1:
        ux = ux.reshape(-1, 1, order="F")
    nfft, nfc = ux.shape
    gam = _d(np.broadcast_to(np.atleast_1d(gam), (nfc,)))
    uyp = None
    if uy is not None:
        uy = _c(uy).reshape(nfft, nfc, order="F")
        uyp = _p(uy)
    return lib().plxo_nextstep(C.c_double(dzmax), C.c_double(phimax), _p(gam), C.c_int(nfc),
                               C.c_double(alphalin), _p(ux), uyp, C.c_long(nfft))


def checkstep(zprop, dz, lcorr, dz_miss, nz_old, cap=4096):
    dzb = np.zeros(cap)
    miss = C.c_double(dz_miss)
    nmem = C.c_int(0)
    nt = lib().plxo_checkstep(C.c_double(zprop), C.c_double(dz), C.c_double(lcorr), C.byref(miss),
                              C.c_int(nz_old), _p(dzb), C.byref(nmem))
    return dzb[:nt].copy(), miss.value, nmem.value, nt


# ------------------------------------------------------------------- fiber ---
def _shape2(u):
    u = _c(u)
    if u.ndim == 1:
        u = u.reshape(-1, 1, order="F")
    return u


def lin_step(betat, dz, u):
    u = _shape2(u)
    nfft, nfc = u.shape
    b = _d(betat).reshape(nfft, nfc, order="F")
    lib().plxo_lin_step(_p(b), C.c_double(dz), _p(u), C.c_long(nfft), C.c_int(nfc))
    return u


def nl_step(alphalin, gam, dz, u, spm, xpm):
    u = _shape2(u)
    nfft, nfc = u.shape
    g = _d(np.broadcast_to(np.atleast_1d(gam), (nfc,)))
    lib().plxo_nl_step(C.c_double(alphalin), _p(g), C.c_double(dz), _p(u), C.c_long(nfft),
                       C.c_int(nfc), C.c_int(int(spm)), C.c_int(int(xpm)))
    return u


def matrix_nl_step(ismanakov, alphalin, gam, dz, ux, uy, spm, xpm):
    ux, uy = _shape2(ux), _shape2(uy)
    nfft, nfc = ux.shape
    g = _d(np.broadcast_to(np.atleast_1d(gam), (nfc,)))
    rc = lib().plxo_matrix_nl_step(C.c_long(nfft), C.c_int(int(ismanakov)), C.c_double(alphalin), _p(g),
                                   C.c_double(dz), _p(ux), _p(uy), C.c_int(nfc), C.c_int(int(spm)),
                                   C.c_int(int(xpm)))
    return rc, ux, uy


def matrix_step(betat, db1, dzb, ux, uy, db0, theta, epsilon, lcorr, ntot, nmem):
    ux, uy = _shape2(ux), _shape2(uy)
    nfft, nfc = ux.shape
    b = _d(betat).reshape(nfft, nfc, order="F")
    d = _d(db1).reshape(nfft, nfc, order="F")
    dzb = _d(dzb)
    db0, theta, epsilon = _d(np.atleast_1d(db0)), _d(np.atleast_1d(theta)), _d(np.atleast_1d(epsilon))
    lib().plxo_matrix_step(_p(b), _p(d), _p(dzb), C.c_int(dzb.size), _p(ux), _p(uy), C.c_long(nfft),
                           C.c_int(nfc), _p(db0), _p(theta), _p(epsilon), C.c_double(lcorr),
                           C.c_int(ntot), C.c_int(nmem))
    return ux, uy


def matrix_ssfm(ux, uy, betat, db1, dzmaxt, dphimaxt, gam, alphalin, Lf, nplates, manakov, fls,
                db0, theta, epsilon, return_dz=False, replay_dz=None):
    """fiber.m:459-554.  Returns (rc, firstdz, ncycle, ux, uy) and, with return_dz, the list of the step lengths nextstep
    returned (fiber.m:512, :534: ncycle of them).  replay_dz: step k takes replay_dz[k] instead of nextstep's result."""
    ux, uy = _shape2(ux), _shape2(uy)
    nfft, nfc = ux.shape
    b = _d(betat).reshape(nfft, nfc, order="F")
    d = _d(db1).reshape(nfft, nfc, order="F")
    g = _d(np.broadcast_to(np.atleast_1d(gam), (nfc,)))
    db0, theta, epsilon = _d(np.atleast_1d(db0)), _d(np.atleast_1d(theta)), _d(np.atleast_1d(epsilon))
    first = C.c_double(0)
    ncyc = C.c_int(0)
    log = np.zeros(1 << 16) if return_dz else None
    rp = None
    if replay_dz is not None:
        rp = _d(np.atleast_1d(replay_dz))
        lib().plxo_set_step_replay(_p(rp), C.c_int(rp.size))
    if return_dz:
        lib().plxo_set_step_log(_p(log), C.c_int(log.size))
    rc = lib().plxo_matrix_ssfm(_p(ux), _p(uy), _p(b), _p(d), C.c_double(dzmaxt), C.c_double(dphimaxt),
                                _p(g), C.c_double(alphalin), C.c_int(nfc), C.c_long(nfft), C.c_double(Lf),
                                C.c_int(nplates), C.c_int(int(manakov)), _fls(fls), _p(db0), _p(theta),
                                _p(epsilon), C.byref(first), C.byref(ncyc))
    if rp is not None:
        lib().plxo_set_step_replay(None, C.c_int(0))
    if return_dz:
        n = lib().plxo_step_log_count()
        lib().plxo_set_step_log(None, C.c_int(0))
        return rc, first.value, ncyc.value, ux, uy, log[:min(n, log.size)].copy()
    return rc, first.value, ncyc.value, ux, uy


def scalar_ssfm(u, betat, dzmaxt, dphimaxt, gam, alphalin, Lf, fls, tolflag=0, trg_err=0.0,
                trg_safety=0.9):
    """fiber.m:557-636.  Returns (firstdz, ncycle, u)."""
    u = _shape2(u)
    nfft, nfc = u.shape
    b = _d(betat).reshape(nfft, nfc, order="F")
    g = _d(np.broadcast_to(np.atleast_1d(gam), (nfc,)))
    first = C.c_double(0)
    ncyc = C.c_int(0)
    lib().plxo_scalar_ssfm(_p(u), _p(b), C.c_double(dzmaxt), C.c_double(dphimaxt), _p(g),
                           C.c_double(alphalin), C.c_long(nfft), C.c_int(nfc), C.c_double(Lf), _fls(fls),
                           C.c_int(tolflag), C.c_double(trg_err), C.c_double(trg_safety), C.byref(first),
                           C.byref(ncyc))
    return first.value, ncyc.value, u


def scalar_a_ssfm(u, betat, dzmaxt, dphimaxt, gam, alphalin, Lf, trg_err, trg_safety, fls):
    """fiber.m:639-679.  Returns (firstdz, ncycle, nrej, u)."""
    u = _shape2(u)
    nfft, nfc = u.shape
    b = _d(betat).reshape(nfft, nfc, order="F")
    g = _d(np.broadcast_to(np.atleast_1d(gam), (nfc,)))
    first = C.c_double(0)
    ncyc = C.c_int(0)
    nrej = C.c_int(0)
    lib().plxo_scalar_a_ssfm(_p(u), _p(b), C.c_double(dzmaxt), C.c_double(dphimaxt), _p(g),
                             C.c_double(alphalin), C.c_long(nfft), C.c_int(nfc), C.c_double(Lf),
                             C.c_double(trg_err), C.c_double(trg_safety), _fls(fls), C.byref(first),
                             C.byref(ncyc), C.byref(nrej))
    return first.value, ncyc.value, nrej.value, u


# --------------------------------------------------------------------- CDE ---
def cde_transfer(fftlen, fs, lambda_ref, span, D, S):
    H = np.empty(fftlen, dtype=np.complex128)
    lib().plxo_cde_transfer(_p(H), C.c_long(fftlen), C.c_double(fs), C.c_double(lambda_ref),
                            C.c_double(span), C.c_double(D), C.c_double(S))
    return H


def overlap_both_trans(x, H, L):
    x = _c(x).reshape(-1)
    H = _c(H).reshape(-1)
    y = np.zeros_like(x)
    rc = lib().plxo_overlap_both_trans(_p(x), C.c_long(x.size), _p(H), C.c_long(H.size), C.c_long(L), _p(y))
    return (None if rc else y), rc


def cde_ofde(inx, iny, fs, lambda_ref, span, D, S, fftlen, L):
    inx, iny = _c(inx).reshape(-1), _c(iny).reshape(-1)
    ox, oy = np.zeros_like(inx), np.zeros_like(iny)
    rc = lib().plxo_cde_ofde(_p(inx), _p(iny), C.c_long(inx.size), C.c_double(fs), C.c_double(lambda_ref),
                             C.c_double(span), C.c_double(D), C.c_double(S), C.c_long(fftlen), C.c_long(L),
                             _p(ox), _p(oy))
    return ox, oy, rc


# ---------------------------------------------------------------- CMA/EASI ---
def _split(a):
    a = np.asarray(a)
    return _d(a.real), _d(a.imag)


def cmaadaptivefilter(xx, h1, h2, taps, mu, R, sps):
    """MEX semantics (cmaadaptivefilter.c:93-174): returns (y, h1_updated, h2_updated);
    the reference writes the taps back into its inputs and returns 0,0."""
    rc = lib().plxo_cma_gateway_check(C.c_double(taps), C.c_double(sps), C.c_int(1))
    if rc == 1:
        raise ValueError("Ntaps should be an ODD INTEGER.")
    if rc == 2:
        raise ValueError("Samples x symbol should be either 1 or 2.")
    xx = np.asarray(xx)
    Ndim = xx.shape[0]
    taps = int(taps)
    xr, xi = _split(xx)
    h1r, h1i = _split(h1)
    h2r, h2i = _split(h2)
    dimY = Ndim - taps + 1
    yr = np.zeros((dimY, 2), order="F")
    yi = np.zeros((dimY, 2), order="F")
    R = _d(np.atleast_1d(R))
    lib().plxo_cmafilter(_p(xr), _p(xi), C.c_int(Ndim), _p(h1r), _p(h1i), _p(h2r), _p(h2i), C.c_int(taps),
                         C.c_double(mu), _p(R), _p(yr), _p(yi), C.c_int(int(int(sps) == 1)))
    return yr + 1j * yi, h1r + 1j * h1i, h2r + 1j * h2i


def easiadaptivefilter(xx, h1, h2, taps, mu, sps):
    rc = lib().plxo_cma_gateway_check(C.c_double(taps), C.c_double(sps), C.c_int(0))
    if rc == 2:
        raise ValueError("Samples x symbol should be either 1 or 2.")
    xx = np.asarray(xx)
    Ndim = xx.shape[0]
    taps = int(taps)
    xr, xi = _split(xx)
    h1r, h1i = _split(h1)
    h2r, h2i = _split(h2)
    dimY = Ndim - taps + 1
    yr = np.zeros((dimY, 2), order="F")
    yi = np.zeros((dimY, 2), order="F")
    lib().plxo_easifilter(_p(xr), _p(xi), C.c_int(Ndim), _p(h1r), _p(h1i), _p(h2r), _p(h2i), C.c_int(taps),
                          C.c_double(mu), _p(yr), _p(yi), C.c_int(int(int(sps) == 1)))
    return yr + 1j * yi, h1r + 1j * h1i, h2r + 1j * h2i


def _twin_args(xx, h1, h2):
    xx = np.asarray(xx, dtype=np.complex128)
    x = np.ascontiguousarray(xx.T)                      # [2][Ndim]
    g1 = np.array(np.asarray(h1, dtype=np.complex128).reshape(-1, 2).T, order="C", copy=True)   # [2][ntap], never aliasing the caller's
    g2 = np.array(np.asarray(h2, dtype=np.complex128).reshape(-1, 2).T, order="C", copy=True)
    ntap = g1.shape[1]
    y = np.zeros((2, x.shape[1] - ntap + 1), dtype=np.complex128)
    return x, g1, g2, ntap, y


def cmaadaptivefilter_m(xx, h1, h2, taps, mu, R, sps=1):
    """The .m twin (cmaadaptivefilter.m:1,52-72): [Y h1 h2]; `taps` and `sps` are ignored there (size(h1,1) rules,
    every sample updates).  Returns (Y [L x 2], h1 [ntap x 2], h2 [ntap x 2])."""
    x, g1, g2, ntap, y = _twin_args(xx, h1, h2)
    R = _d(np.atleast_1d(R))
    lib().plxo_cmafilter_m(_p(x), C.c_int(x.shape[1]), _p(g1), _p(g2), C.c_int(ntap), C.c_double(mu), _p(R), _p(y))
    return y.T.copy(), g1.T.copy(), g2.T.copy()


def easiadaptivefilter_m(xx, h1, h2, taps, mu, sps=1):
    """The .m twin (easiadaptivefilter.m:1,51-84): complex error matrix, all taps recombined."""
    x, g1, g2, ntap, y = _twin_args(xx, h1, h2)
    lib().plxo_easifilter_m(_p(x), C.c_int(x.shape[1]), _p(g1), _p(g2), C.c_int(ntap), C.c_double(mu), _p(y))
    return y.T.copy(), g1.T.copy(), g2.T.copy()


def cmapolardemux(x, M, taps, mu, R):
    x = _c(x)
    L = x.shape[0]
    M = np.ascontiguousarray(np.asarray(M, dtype=np.complex128))
    y = np.zeros((L, 2), dtype=np.complex128, order="F")
    h1 = np.zeros((taps, 2), dtype=np.complex128, order="F")
    h2 = np.zeros((taps, 2), dtype=np.complex128, order="F")
    R = _d(np.atleast_1d(R))
    n = lib().plxo_cmapolardemux(_p(x), C.c_long(L), _p(M), C.c_int(taps), C.c_double(mu), _p(R), _p(y),
                                 _p(h1), _p(h2))
    return y, h1, h2, n


def easipolardemux(x, M, mu):
    x = _c(x)
    L = x.shape[0]
    M = np.ascontiguousarray(np.asarray(M, dtype=np.complex128))
    y = np.zeros((L, 2), dtype=np.complex128, order="F")
    h1 = np.zeros((1, 2), dtype=np.complex128, order="F")
    h2 = np.zeros((1, 2), dtype=np.complex128, order="F")
    n = lib().plxo_easipolardemux(_p(x), C.c_long(L), _p(M), C.c_double(mu), _p(y), _p(h1), _p(h2))
    return y, h1, h2, n


def easipolardemux_m(x, M, mu):
    """easipolardemux (DspPdmCohQpsk.m:195-244) around the .m twin of the filter (no MEX compiled)."""
    x = _c(x)
    L = x.shape[0]
    M = np.ascontiguousarray(np.asarray(M, dtype=np.complex128))
    y = np.zeros((L, 2), dtype=np.complex128, order="F")
    h1 = np.zeros((1, 2), dtype=np.complex128, order="F")
    h2 = np.zeros((1, 2), dtype=np.complex128, order="F")
    n = lib().plxo_easipolardemux_m(_p(x), C.c_long(L), _p(M), C.c_double(mu), _p(y), _p(h1), _p(h2))
    return y, h1, h2, n


# --------------------------------------------------------------------- DSP ---
class DspParams(C.Structure):
    _fields_ = [("workatbaudrate", C.c_int), ("applynlr", C.c_int), ("nlralpha", C.c_double),
                ("power_mw", C.c_double), ("applypol", C.c_int), ("polmethod", C.c_int),
                ("cma_R", C.c_double * 2), ("cma_mu", C.c_double), ("cma_taps", C.c_int),
                ("cma_txpolars", C.c_int), ("cma_phizero", C.c_double), ("easi_mu", C.c_double),
                ("easi_txpolars", C.c_int), ("easi_phizero", C.c_double), ("modorder", C.c_int),
                ("freqavg", C.c_int), ("phasavg", C.c_int), ("poworder", C.c_int),
                ("cma_has_mat", C.c_int), ("easi_has_mat", C.c_int), ("cma_mat", C.c_double * 8),
                ("easi_mat", C.c_double * 8), ("mfile_twins", C.c_int)]


_POLMETHOD = {"singlepol": 0, "cma": 1, "easi": 2, "combo": 3}


def dsp_params(power_mw, workatbaudrate=False, applynlr=False, nlralpha=0.0, applypol=False,
               polmethod="cma", cma_R=(1.0, 1.0), cma_mu=1 / 6000, cma_taps=7, cma_txpolars=2,
               cma_phizero=0.0, easi_mu=1 / 6000, easi_txpolars=2, easi_phizero=0.0, modorder=2,
               freqavg=500, phasavg=3, poworder=2, cma_mat=None, easi_mat=None, mfile_twins=False):
    p = DspParams()
    for m, has, dst in ((cma_mat, "cma_has_mat", p.cma_mat), (easi_mat, "easi_has_mat", p.easi_mat)):
        if m is not None:
            setattr(p, has, 1)
            for k, v in enumerate(np.asarray(m, dtype=np.complex128).reshape(-1)):
                dst[2 * k], dst[2 * k + 1] = v.real, v.imag
    p.mfile_twins = int(bool(mfile_twins))
    p.workatbaudrate, p.applynlr, p.nlralpha, p.power_mw = int(workatbaudrate), int(applynlr), nlralpha, power_mw
    p.applypol, p.polmethod = int(applypol), _POLMETHOD[polmethod.lower()]
    p.cma_R[0], p.cma_R[1] = cma_R
    p.cma_mu, p.cma_taps, p.cma_txpolars, p.cma_phizero = cma_mu, cma_taps, cma_txpolars, cma_phizero
    p.easi_mu, p.easi_txpolars, p.easi_phizero = easi_mu, easi_txpolars, easi_phizero
    p.modorder, p.freqavg, p.phasavg, p.poworder = modorder, freqavg, phasavg, poworder
    return p


def dsp_pdm_coh_qpsk(samples, params):
    s = _shape2(samples)
    Lin, ncol = s.shape
    out = np.zeros((Lin, ncol), dtype=np.complex128, order="F")
    L = lib().plxo_dsp_pdm_coh_qpsk(_p(s), C.c_long(Lin), C.c_int(ncol), C.byref(params), _p(out))
    if L < 0:
        raise ValueError("dsp_pdm_coh_qpsk failed")
    return np.asfortranarray(out.reshape(-1, order="F")[: L * ncol].reshape(L, ncol, order="F"))


def vitvit(s, P, M, k, applyunwrap):
    s = _shape2(s)
    L, ncol = s.shape
    th = np.zeros((L, ncol), order="F")
    lib().plxo_vitvit(_p(s), C.c_long(L), C.c_int(ncol), C.c_int(P), C.c_int(M), C.c_int(k),
                      C.c_int(int(applyunwrap)), _p(th))
    return th


def fastshift(x, n):
    x2 = _shape2(x)
    L, ncol = x2.shape
    y = np.zeros_like(x2)
    lib().plxo_fastshift(_p(x2), C.c_long(L), C.c_int(ncol), C.c_long(n), _p(y))
    return y.reshape(np.asarray(x).shape, order="F") if np.asarray(x).ndim == 1 else y


def nmod(A, N):
    return lib().plxo_nmod(C.c_long(A), C.c_long(N))


def unwrap(p):
    p = _d(p).reshape(-1)
    lib().plxo_unwrap(_p(p), C.c_long(p.size))
    return p


def samp2pat_coherent(phase):
    ph = _d(phase)
    if ph.ndim == 1:
        ph = ph.reshape(-1, 1, order="F")
    L, ncol = ph.shape
    pat = np.zeros((L, 2 * ncol), dtype=np.uint8, order="F")
    lib().plxo_samp2pat_coherent(_p(ph), C.c_long(L), C.c_int(ncol), _p(pat))
    return pat


# ---------------------------------------------------------------------- MC ---
class McState(C.Structure):
    _fields_ = [("first", C.c_int), ("dim", C.c_int), ("n", C.c_double * 256), ("avg", C.c_double * 256),
                ("var", C.c_double * 256), ("varlim", (C.c_double * 256) * 2), ("cond", C.c_int * 256),
                ("epsilon", C.c_double * 2)]


def erfcinv(y):
    return lib().plxo_erfcinv(C.c_double(y))


def ber_estimate(state, pat_hat, pat, stop=None, nmin=1, dim=1, nind=1):
    pat_hat, pat = np.asarray(pat_hat), np.asarray(pat)
    err = float(np.sum(pat != pat_hat))
    M = float(pat.size)
    cond = (C.c_int * dim)()
    avg = (C.c_double * dim)()
    nr = (C.c_double * dim)()
    sd = (C.c_double * dim)()
    lib().plxo_ber_estimate(C.byref(state), C.c_double(err), C.c_double(M), C.c_int(dim), C.c_int(nind),
                            C.c_int(int(stop is not None)), C.c_double(stop[0] if stop else 0),
                            C.c_double(stop[1] if stop else 0), C.c_double(nmin), cond, avg, nr, sd)
    return (np.array(cond[:], dtype=bool), np.array(avg[:]), np.array(nr[:]), np.array(sd[:]))


def mc_estimate(state, s, stop=None, nmin=50, method="mean", dim=1, nind=1):
    s = _d(s).reshape(-1)
    cond = (C.c_int * dim)()
    mean = (C.c_double * dim)()
    var = (C.c_double * dim)()
    nr = (C.c_double * dim)()
    sd = (C.c_double * dim)()
    vl = (C.c_double * (2 * dim))()
    lib().plxo_mc_estimate(C.byref(state), _p(s), C.c_long(s.size), C.c_int(dim), C.c_int(nind),
                           C.c_int(int(stop is not None)), C.c_double(stop[0] if stop else 0),
                           C.c_double(stop[1] if stop else 0), C.c_double(nmin),
                           C.c_int(int(method == "var")), cond, mean, var, nr, sd, vl)
    return (np.array(cond[:], dtype=bool),
            dict(mean=np.array(mean[:]), var=np.array(var[:]), nruns=np.array(nr[:]),
                 stdmean=np.array(sd[:]), varlim=np.array(vl[:]).reshape(2, dim, order="F")))

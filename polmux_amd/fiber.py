"""fiber(x, flag): optical fibre in the nonlinear regime -- host side of fiber.m.

Mirror of /root/reference/fiber.m:126-389: flag parsing, PMD draws, physical
conversions, the betat/db1 tables and the DELAY/DISP bookkeeping stay on the
host (O(Nfft) once per call); lines 372-389 -- the dispatch to matrix_ssfm /
scalar_ssfm -- go to the MI355X through plx_ssfm_* (include/polmux_hip.h).
The field is updated in place in GSTATE.FIELDX / GSTATE.FIELDY (device memory).
"""
import ctypes as C
import math

import numpy as np

from . import _abi
from .gstate import CONSTANTS, GSTATE

SAFETYFCT = 0.9      # fiber.m:130
DEF_PLATES = 100     # fiber.m:131

_FLAGS = {           # fiber.m:158-251: flag -> (fls, uses x.dphimax/x.dzmax?)
    "----": ([0, 0, 0, 0], False), "g---": ([1, 0, 0, 0], False), "-p--": ([0, 1, 0, 0], False),
    "--s-": ([0, 0, 1, 0], None), "---x": ([0, 0, 0, 1], True), "gp--": ([1, 1, 0, 0], False),
    "g-s-": ([1, 0, 1, 0], True), "g--x": ([1, 0, 0, 1], True), "-ps-": ([0, 1, 1, 0], True),
    "-p-x": ([0, 1, 0, 1], True), "--sx": ([0, 0, 1, 0], None), "g-sx": ([1, 0, 1, 0], True),
    "-psx": ([0, 1, 1, 0], True), "gps-": ([1, 1, 1, 0], True), "gp-x": ([1, 1, 0, 1], True),
    "gpsx": ([1, 1, 1, 0], True),
}
_NEED_SEP = {"---x", "g--x", "-p-x", "gp-x"}
_XPM_IF_SEP = {"--sx", "g-sx", "-psx", "gpsx"}


def _get(x, name, default=None):
    if isinstance(x, dict):
        return x.get(name, default)
    return getattr(x, name, default)


def _has(x, name):
    return (name in x) if isinstance(x, dict) else hasattr(x, name)


def parse_flag(flag, nfc, x):
    """fiber.m:157-251 -> (fls, dphimaxt, dzmaxt)."""
    key = str(flag).lower()
    if key not in _FLAGS:
        raise ValueError("wrong flag. E.g. flag can be 'g---','gp--','-s--', etc")
    if key in _NEED_SEP and nfc == 1:
        raise ValueError("flag '%s' available only for channels separated" % key)
    fls, uses = _FLAGS[key]
    fls = list(fls)
    length, dzmax, dphimax = _get(x, "length"), _get(x, "dzmax"), _get(x, "dphimax", math.inf)
    if key in _XPM_IF_SEP and nfc != 1:
        fls[3] = 1
    if uses is None:            # '--s-' and '--sx': exact solution with one field (:172-178, :214-221)
        uses = nfc != 1
    if uses:
        return fls, dphimax, dzmax
    return fls, math.inf, length


def fiber_tables(x, fls, nfc, dgdrms_symbols):
    """Physical conversions, fiber.m:302-362.  Returns dict(alphalin, gam, betat, db1, b1, Dch)."""
    CL = CONSTANTS.CLIGHT
    lam, disp, slope = _get(x, "lambda"), _get(x, "disp"), _get(x, "slope")
    alphalin = (math.log(10) * 1e-4) * _get(x, "alphadB")                    # :302
    b20 = -lam ** 2 / 2 / math.pi / CL * disp * 1e-6                          # :308
    b30 = (lam / 2 / math.pi / CL) ** 2 * (2 * lam * disp + lam ** 2 * slope) * 1e-6   # :309
    b30 = b30 * fls[0]                                                        # :311
    lamv = np.atleast_1d(np.asarray(GSTATE.LAMBDA, dtype=float))
    maxl, minl = lamv.max(), lamv.min()
    lamc = 2 * maxl * minl / (maxl + minl)                                    # :315
    Domega_i0 = 2 * math.pi * CL * (1.0 / lamv - 1 / lam)                     # :318
    Domega_ic = 2 * math.pi * CL * (1.0 / lamv - 1 / lamc)
    Domega_c0 = 2 * math.pi * CL * (1.0 / lamc - 1 / lam)
    b1 = b20 * Domega_ic + 0.5 * b30 * (Domega_i0 ** 2 - Domega_c0 ** 2)      # :321
    n2, aeff = _get(x, "n2"), _get(x, "aeff")
    if nfc == 1:
        beta1 = np.zeros(1)                                                   # :323
        Domega_i0 = np.array([2 * math.pi * CL * (1.0 / lamc - 1 / lam)])
        gam = np.array([2 * math.pi * n2 / (lamc * aeff) * 1e18])             # :325
    else:
        beta1 = b1                                                            # :327
        gam = 2 * math.pi * n2 / (lamv * aeff) * 1e18                         # :328
    beta2 = (b20 + b30 * Domega_i0) * fls[0]                                  # :330-332
    Dch = disp + slope * (lamv - lam)                                         # :336
    omega = 2 * math.pi * GSTATE.SYMBOLRATE * np.asarray(GSTATE.FN, dtype=float)   # :352
    nfft = omega.size
    betat = np.zeros((nfft, nfc), order="F")
    db1 = np.zeros((nfft, nfc), order="F")
    for k in range(nfc):
        betat[:, k] = omega * beta1[k] + 0.5 * omega ** 2 * beta2[k] + omega ** 3 * b30 / 6   # :355-356
        if fls[1] == 1:
            db1[:, k] = dgdrms_symbols / GSTATE.SYMBOLRATE * omega            # :284,358
    return dict(alphalin=alphalin, gam=gam, betat=betat, db1=db1, b1=b1, Dch=Dch)


_plans = {}
_rearm = {}       # plan key -> [time-outs seen, unfused spans since, patience] (see fiber())


def _plan_for(desc_key, build):
    plan = _plans.get(desc_key)
    if plan is None:
        if len(_plans) > 8:
            release_plans()
        plan = build()
        _plans[desc_key] = plan
    return plan


def release_plans():
    """Destroy the propagator plans fiber() keeps between calls (one per fibre type and grid)."""
    for k in list(_plans):
        _abi.get().call("plx_ssfm_destroy", _plans.pop(k)[0])
    _rearm.clear()


def fiber(x, flag=None, rng=None):
    """FIBER optical fibre in the nonlinear regime (fiber.m:1).  Works in place on
    GSTATE.FIELDX/FIELDY; returns the birefringence struct ``brf`` (dict) when PMD is on."""
    import torch
    if flag is None:
        raise ValueError("Missing propagation type")                         # fiber.m:137
    x = dict(x) if isinstance(x, dict) else {k: getattr(x, k) for k in dir(x) if not k.startswith("_")}
    if GSTATE.FIELDX is None:
        raise ValueError("create_field must be called before fiber")
    fx = GSTATE.FIELDX
    nfc, nfft = fx.shape
    if not _has(x, "dzmax") or x["dzmax"] > x["length"]:
        x["dzmax"] = x["length"]                                              # :139-141
    tolflag = 0
    if _has(x, "ltol"):                                                       # adaptive step-size, :143-151
        if not _has(x, "dphimax"):
            x["dphimax"] = math.inf
        tolflag = 1 if x.get("dphiadapt") else 2
    fls, dphimaxt, dzmaxt = parse_flag(flag, nfc, x)
    isy = GSTATE.FIELDY is not None and GSTATE.FIELDY.numel() > 0
    isv = fls[1] == 1 or isy
    brf = None
    if fls[1] == 1:                                                           # :255-289
        manakov = str(x.get("manakov", "no")).lower() == "yes"
        if not _has(x, "dgd"):
            raise ValueError("Missing DGD in fiber")
        ispmf = sum(_has(x, k) for k in ("db0", "theta", "epsilon"))
        if ispmf == 3:
            theta = np.atleast_1d(np.asarray(x["theta"], dtype=float))
            nplates = theta.size
            db0 = np.atleast_1d(np.asarray(x["db0"], dtype=float))
            eps = np.atleast_1d(np.asarray(x["epsilon"], dtype=float))
            dgdrms = x["dgd"] / nplates                                       # :269
        elif ispmf == 0:
            nplates = int(x.get("nplates", DEF_PLATES))
            rng = rng or np.random.default_rng()
            db0 = rng.random(nplates) * 2 * math.pi - math.pi                 # :274
            theta = rng.random(nplates) * math.pi - 0.5 * math.pi             # :275
            eps = 0.5 * np.arcsin(rng.random(nplates) * 2 - 1)                # :276
            dgdrms = math.sqrt(3 * math.pi / 8) * x["dgd"] / math.sqrt(nplates)   # :277
        else:
            raise ValueError("Missing one of db0, theta or epsilon in fiber")
        brf = dict(db0=db0, theta=theta, epsilon=eps, dgd=x["dgd"])
        if not isy:
            GSTATE.FIELDY = torch.zeros_like(fx)                              # :285-289
            isy = True
    else:
        dgdrms, manakov, nplates = 0.0, False, 1                             # :291-297
        db0 = theta = eps = np.zeros(1)
    t = fiber_tables(x, fls, nfc, dgdrms)
    # DELAY / DISP bookkeeping, :367-369
    loc_delay = x["length"] * GSTATE.SYMBOLRATE * t["b1"]
    rows = 2 if isy else 1
    GSTATE.DELAY = np.asarray(GSTATE.DELAY if GSTATE.DELAY is not None else 0.0) + np.ones((rows, 1)) * loc_delay
    GSTATE.DISP = np.asarray(GSTATE.DISP if GSTATE.DISP is not None else 0.0) + \
        np.ones((rows, 1)) * fls[0] * t["Dch"] * x["length"] * 1e-3

    lib = _abi.get()
    gam = np.ascontiguousarray(t["gam"], dtype=float)
    if tolflag == 2 and isv:
        raise ValueError("adaptive step available in absence of polarization effects")   # :374
    if tolflag and not isv:
        # scalar_a_ssfm / dphiadapt (:376-377, :386-387): host-driven scheme behind one gateway call
        from .gstate import to_device_field, to_host_field
        d = _abi.SsfmDesc()
        d.nfft, d.nfc, d.dual_pol, d.max_frames = nfft, nfc, 0, 1
        for i in range(4):
            d.fls[i] = fls[i]
        d.dzmaxt, d.dphimaxt, d.alphalin, d.length, d.nplates, d.manakov = dzmaxt, dphimaxt, t["alphalin"], x["length"], 1, 0
        d.gam, d.betat, d.db1 = gam.ctypes.data, t["betat"].ctypes.data, t["db1"].ctypes.data
        u = to_host_field(fx)
        ur, ui = np.asfortranarray(u.real.copy()), np.asfortranarray(u.imag.copy())
        first, ncyc, nrej = C.c_double(), C.c_int32(), C.c_int32()
        lib.call("plx_scalar_ssfm_adaptive", ur.ctypes.data, ui.ctypes.data, C.byref(d), tolflag, float(x["ltol"]),
                 SAFETYFCT, C.byref(first), C.byref(ncyc), C.byref(nrej))
        GSTATE.FIELDX = to_device_field(ur + 1j * ui)
        fiber.last = dict(firstdz=first.value, ncycle=ncyc.value, nrej=nrej.value)
        return None
    key = (nfft, nfc, int(isv), tuple(fls), dzmaxt, dphimaxt, t["alphalin"], x["length"], nplates, manakov,
           gam.tobytes(), t["betat"].tobytes(), t["db1"].tobytes())

    def build():
        d = _abi.SsfmDesc()
        d.nfft, d.nfc, d.dual_pol, d.max_frames = nfft, nfc, int(isv), 1
        for i in range(4):
            d.fls[i] = fls[i]
        d.dzmaxt, d.dphimaxt, d.alphalin, d.length = dzmaxt, dphimaxt, t["alphalin"], x["length"]
        d.nplates, d.manakov = nplates, int(manakov)
        d.gam, d.betat, d.db1 = gam.ctypes.data, t["betat"].ctypes.data, t["db1"].ctypes.data
        p = C.c_void_p()
        lib.call("plx_ssfm_create", C.byref(p), C.byref(d))
        return (p, d)

    plan = _plan_for(hash(key), build)[0]
    # diagnostics for parity work (include/polmux_hip.h, plx_ssfm_set_step_sequence): x["_replay_dz"] = the step lengths to use
    # instead of nextstep's (fiber.m:682-715), x["_log_dz"] = True returns the device's own sequence in fiber.last["dz"]
    replay = x.get("_replay_dz")
    stream = torch.cuda.current_stream().cuda_stream
    if not fx.is_contiguous():
        fx = GSTATE.FIELDX = fx.contiguous()
    fy = GSTATE.FIELDY if isv else None
    if fy is not None and not fy.is_contiguous():
        fy = GSTATE.FIELDY = fy.contiguous()
    # fiber.m:372-389 always returns a field.  The fused sweep's frame barrier times out when another kernel holds part of
    # the GPU (PLX_ERR_TIMEOUT: the field of that call is invalid and the plan has switched itself to the three-sweep step),
    # so a plan on the fused step keeps a copy of the input to repeat the span from, as the gateway tier does from its
    # staging buffer.  The replay list and the step log are armed for THIS call only, whatever happens in it.
    pinfo = (C.c_int32 * 8)()
    lib.call("plx_ssfm_info", plan, pinfo)
    if pinfo[0] == 0:
        # a cached plan that fell back after a time-out tries the fused step again after 16 (32, 64, ... 4096) spans, as the
        # gateway tier does (include/polmux_hip.h, plx_ssfm_barrier_timeouts): the stall that caused it is usually gone
        nto = C.c_int32()
        lib.call("plx_ssfm_barrier_timeouts", plan, C.byref(nto), 0)
        if nto.value > 0:
            st = _rearm.setdefault(hash(key), [0, 0, 16])          # [time-outs seen, unfused spans since, patience]
            if nto.value > st[0]:
                st[:] = [nto.value, 0, min(4096, st[2] * 2 if st[0] else 16)]
            st[1] += 1
            if st[1] >= st[2]:
                lib.call("plx_ssfm_barrier_timeouts", plan, None, 1)
                lib.call("plx_ssfm_info", plan, pinfo)
                st[1] = 0
    keep = (fx.clone(), fy.clone() if fy is not None else None) if pinfo[0] == 1 else None
    info = None
    try:
        if replay is not None:
            rz = np.ascontiguousarray(replay, dtype=float)
            lib.call("plx_ssfm_set_step_sequence", plan, rz.ctypes.data, rz.size)
        if x.get("_log_dz"):
            lib.call("plx_ssfm_log_steps", plan, 1 << 14)
        if fls[1] == 1:
            a, b, c_ = (np.ascontiguousarray(v, dtype=float) for v in (db0, theta, eps))
            lib.call("plx_ssfm_set_birefringence", plan, a.ctypes.data, b.ctypes.data, c_.ctypes.data, 1)
        try:
            lib.call("plx_ssfm_propagate_dev", plan, fx.data_ptr(), fy.data_ptr() if fy is not None else None, 1, stream)
        except _abi.PolmuxError as exc:
            if exc.code != _abi.PLX_ERR_TIMEOUT or keep is None:
                raise
            fx.copy_(keep[0])
            if fy is not None:
                fy.copy_(keep[1])
            lib.call("plx_ssfm_propagate_dev", plan, fx.data_ptr(), fy.data_ptr() if fy is not None else None, 1, stream)
        first, ncyc = C.c_double(), C.c_int32()
        lib.call("plx_ssfm_results", plan, 1, C.byref(first), C.byref(ncyc))
        info = dict(firstdz=first.value, ncycle=ncyc.value)
        if x.get("_log_dz"):
            dzs = np.zeros(min(ncyc.value, 1 << 14))
            lib.call("plx_ssfm_step_sequence", plan, 0, dzs.ctypes.data, dzs.size)
            info["dz"] = dzs
    finally:
        if replay is not None:
            lib.call("plx_ssfm_set_step_sequence", plan, None, 0)
        if x.get("_log_dz"):
            lib.call("plx_ssfm_log_steps", plan, 0)
    fiber.last = info                                                         # fiber.m:431 prints these
    if brf is not None:
        brf.update(lcorr=x["length"] / nplates, betat=t["betat"], db1=t["db1"], **info)
        return brf
    return None


fiber.last = None

"""The MEX shims of INTEGRATION.md (integration/mex/*.c) compiled and driven without MATLAB through a test double of
mex.h (tests/mexstub): the calling convention of the reference's gateways -- separate real/imaginary planes, missing
imaginary planes allocated on the inputs, h1/h2 updated in the caller's arrays with 0, 0 returned, mexErrMsgTxt on bad
arguments (cmaadaptivefilter.c:93-174, easiadaptivefilter.c:95-169, fastexp.c:46-67)."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = sorted(glob.glob(os.path.join(ROOT, "integration", "mex", "*.c")))
STUB = os.path.join(ROOT, "tests", "mexstub")
BUILD = os.path.join(STUB, "_build")


class MxArray(C.Structure):
    _fields_ = [("m", C.c_size_t), ("n", C.c_size_t), ("pr", C.POINTER(C.c_double)), ("pi", C.POINTER(C.c_double)),
                ("str", C.c_char_p)]


def test_every_shim_compiles_against_the_abi_header():
    assert len(SHIMS) >= 5
    for f in SHIMS:
        subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + STUB,
                               "-I" + os.path.join(ROOT, "include"), f])


def _load(shim):
    """shim + stub runtime -> one shared object linked against libpolmux_hip.so"""
    import torch  # noqa: F401  (the HIP runtime of torch must be loaded first, see polmux_amd/_abi.py)
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "lib%s.so" % shim)
    libdir = os.path.join(ROOT, "polmux_amd", "lib")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-shared", "-fPIC", "-I" + STUB, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "integration", "mex", shim + ".c"), os.path.join(STUB, "mexstub.c"),
                           "-L" + libdir, "-lpolmux_hip", "-Wl,-rpath," + libdir, "-o", so])
    from polmux_amd import _abi
    _abi.get()
    lib = C.CDLL(so)
    lib.mxCreateDoubleMatrix.restype = C.POINTER(MxArray)
    lib.mxCreateDoubleMatrix.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
    lib.mxCreateString.restype = C.POINTER(MxArray)
    lib.mxCreateString.argtypes = [C.c_char_p]
    lib.mexstub_call.argtypes = [C.c_int, C.POINTER(C.POINTER(MxArray)), C.c_int, C.POINTER(C.POINTER(MxArray))]
    lib.mexstub_last_error.restype = C.c_char_p
    lib.mexstub_lock_count.restype = C.c_int
    lib.mexstub_run_atexit.restype = C.c_int
    return lib


def _mx(lib, a, force_complex=False):
    """numpy [m x n] -> mxArray with separate planes; a real array gets NO imaginary plane (like MATLAB)"""
    a = np.atleast_2d(np.asarray(a))
    cplx = np.iscomplexobj(a) or force_complex
    a = a.astype(np.complex128 if cplx else np.float64)
    p = lib.mxCreateDoubleMatrix(a.shape[0], a.shape[1], 1 if cplx else 0)
    flat = np.asfortranarray(a)
    re = np.ascontiguousarray(flat.real.T, dtype=np.float64)          # column-major order = C order of the transpose
    C.memmove(p.contents.pr, re.ctypes.data, a.size * 8)
    if cplx:
        im = np.ascontiguousarray(flat.imag.T, dtype=np.float64)
        C.memmove(p.contents.pi, im.ctypes.data, a.size * 8)
    return p


def _np(p):
    m, n = p.contents.m, p.contents.n
    re = np.ctypeslib.as_array(p.contents.pr, shape=(n, m)).T.copy() if m * n else np.zeros((m, n))
    if p.contents.pi:
        return re + 1j * np.ctypeslib.as_array(p.contents.pi, shape=(n, m)).T
    return re


def _call(lib, nlhs, *args):
    prhs = (C.POINTER(MxArray) * len(args))(*args)
    plhs = (C.POINTER(MxArray) * max(nlhs, 1))()
    rc = lib.mexstub_call(nlhs, plhs, len(args), prhs)
    return rc, [plhs[i] for i in range(nlhs)], lib.mexstub_last_error().decode()


@pytest.mark.gpu
def test_fastexp_shim():
    lib = _load("plx_fastexp_mex")
    x = np.linspace(-40, 40, 35).reshape(7, 5)
    rc, out, err = _call(lib, 1, _mx(lib, x))
    assert rc == 0, err
    np.testing.assert_allclose(_np(out[0]), np.cos(x) + 1j * np.sin(x), atol=1e-15)


@pytest.mark.gpu
def test_cma_shim_updates_taps_in_place_and_returns_zeros():
    from oracle import plxo as oracle
    lib = _load("plx_cmaadaptivefilter_mex")
    r = np.random.default_rng(5)
    taps, L = 7, 200
    a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (L + taps - 1, 2))))
    c, s = np.cos(0.3), np.sin(0.3)
    xx = a @ np.array([[c, s], [-s, c]])
    h1 = np.zeros((taps, 2)); h1[3, 0] = 1.0                     # REAL arrays: no imaginary plane until the gateway adds one
    h2 = np.zeros((taps, 2)); h2[3, 1] = 1.0
    mh1, mh2 = _mx(lib, h1), _mx(lib, h2)
    assert not mh1.contents.pi
    rc, out, err = _call(lib, 3, _mx(lib, xx), mh1, mh2, _mx(lib, [[taps]]), _mx(lib, [[1e-2]]), _mx(lib, [[1.0, 1.0]]), _mx(lib, [[1]]))
    assert rc == 0, err
    y, oh1, oh2 = oracle.cmaadaptivefilter(xx, h1.astype(complex), h2.astype(complex), taps, 1e-2, [1.0, 1.0], 1)
    np.testing.assert_allclose(_np(out[0]), y, atol=1e-10)
    assert bool(mh1.contents.pi)                                  # imaginary plane allocated ON THE INPUT (:141-155)
    np.testing.assert_allclose(_np(mh1), oh1, atol=1e-10)         # taps updated in the caller's arrays
    np.testing.assert_allclose(_np(mh2), oh2, atol=1e-10)
    assert _np(out[1]).tolist() == [[0.0]] and _np(out[2]).tolist() == [[0.0]]   # returns 0, 0 (:166-171)
    rc, _, err = _call(lib, 3, _mx(lib, xx), _mx(lib, h1), _mx(lib, h2), _mx(lib, [[4]]), _mx(lib, [[1e-2]]), _mx(lib, [[1.0, 1.0]]),
                       _mx(lib, [[1]]))
    assert rc == 1 and err == "Ntaps should be an ODD INTEGER."   # mexErrMsgTxt, :118-119
    rc, _, err = _call(lib, 3, _mx(lib, xx), _mx(lib, h1), _mx(lib, h2), _mx(lib, [[taps]]), _mx(lib, [[1e-2]]), _mx(lib, [[1.0, 1.0]]),
                       _mx(lib, [[3]]))
    assert rc == 1 and err == "Samples x symbol should be either 1 or 2."          # :132


@pytest.mark.gpu
def test_matrix_ssfm_and_cde_shims():
    from oracle import plxo as oracle
    from polmux_amd import synth
    lib = _load("plx_matrix_ssfm_mex")
    nsymb, nt = 64, 16
    ux, uy, _, _ = synth.pdm_qpsk_field(nsymb, nt, 6.0)
    omega = 2 * np.pi * 28 * synth.fn_grid(nsymb, nt)
    betat = (0.5 * omega ** 2 * -2.17e-8).reshape(-1, 1)
    db1 = np.zeros_like(betat)
    args = [_mx(lib, ux.reshape(-1, 1)), _mx(lib, uy.reshape(-1, 1)), _mx(lib, betat), _mx(lib, db1), _mx(lib, [[2e4]]), _mx(lib, [[5e-3]]),
            _mx(lib, [[1.3e-6]]), _mx(lib, [[4.6e-5]]), _mx(lib, [[1]]), _mx(lib, [[2e4]]), _mx(lib, [[1]]), _mx(lib, [[0]]),
            _mx(lib, [[1, 0, 1, 0]]), _mx(lib, [[0.0]]), _mx(lib, [[0.0]]), _mx(lib, [[0.0]])]
    rc, out, err = _call(lib, 4, *args)
    assert rc == 0, err
    orc, ofd, onc, ox, oy = oracle.matrix_ssfm(ux, uy, betat, db1, 2e4, 5e-3, [1.3e-6], 4.6e-5, 2e4, 1, False, [1, 0, 1, 0], [0.0], [0.0], [0.0])
    assert _np(out[1])[0, 0] == onc and _np(out[0])[0, 0] == pytest.approx(ofd, rel=1e-12)
    assert np.abs(_np(out[2]) - ox).max() <= 1e-9 * np.abs(ox).max()
    assert np.abs(_np(out[3]) - oy).max() <= 1e-9 * np.abs(oy).max()
    lib2 = _load("plx_cde_ofde_mex")
    r = np.random.default_rng(2)
    x = r.standard_normal(700) + 1j * r.standard_normal(700)
    y = r.standard_normal(700) + 1j * r.standard_normal(700)
    sc = [_mx(lib2, [[v]]) for v in (56e9, 1550e-9, 8e4, 17e-6, 0.0, 256, 128)]
    rc, out, err = _call(lib2, 2, _mx(lib2, x.reshape(-1, 1)), _mx(lib2, y.reshape(-1, 1)), *sc)
    assert rc == 0, err
    ex, ey, _ = oracle.cde_ofde(x, y, 56e9, 1550e-9, 8e4, 17e-6, 0.0, 256, 128)
    np.testing.assert_allclose(_np(out[0])[:, 0], ex, atol=1e-11)
    np.testing.assert_allclose(_np(out[1])[:, 0], ey, atol=1e-11)
    sc[6] = _mx(lib2, [[300]])                                    # L > N: OverlapBothTrans display()s and returns [] (CDE_OFDE.m:63-85)
    rc, out, err = _call(lib2, 2, _mx(lib2, x.reshape(-1, 1)), _mx(lib2, y.reshape(-1, 1)), *sc)
    assert rc == 0 and out[0].contents.m == 0


@pytest.mark.gpu
def test_rx_front_shim_matches_the_resident_tier_and_oracle():
    """plx_rx_front_mex: the front end called the MATLAB way (host arrays, tables from the .m side) equals oracle/front.py."""
    from oracle import front
    from polmux_amd import rxfront, synth
    lib = _load("plx_rx_front_mex")
    nsymb, nt = 256, 16
    n = nsymb * nt
    ux, uy, _, _ = synth.pdm_qpsk_field(nsymb, nt, 2.0)
    fn = synth.fn_grid(nsymb, nt)
    hopt = rxfront.myfilter("gauss", fn, 0.95)                     # real table: no imaginary plane
    hel = rxfront.myfilter("bessel5", fn, 0.65)
    b = rxfront.fir1_lowpass(16, 2.0 / nt)
    rc, out, err = _call(lib, 1, _mx(lib, ux.reshape(-1, 1)), _mx(lib, uy.reshape(-1, 1)), _mx(lib, hopt.reshape(-1, 1)),
                         _mx(lib, hel.reshape(-1, 1)), _mx(lib, [[1.0]]), _mx(lib, [[1]]), _mx(lib, [[5]]), _mx(lib, [[nt // 2]]),
                         _mx(lib, b.reshape(1, -1)), _mx(lib, [[-9, -9]]))
    assert rc == 0, err
    got = _np(out[0])
    assert got.shape == (2 * nsymb, 2)
    cur = front.receiver_cohmix(ux, uy, hopt, 1.0, hel, True)
    want = front.rx_front(cur, True, 5, [-9, -9], nt // 2, b)
    assert np.mean(np.abs(got - want) > 1e-9 * np.abs(want).max()) < 2e-3
    rc, _, err = _call(lib, 1, _mx(lib, ux.reshape(-1, 1)), _mx(lib, uy.reshape(-1, 1)), _mx(lib, hopt.reshape(-1, 1)),
                       _mx(lib, hel.reshape(-1, 1)), _mx(lib, [[1.0]]), _mx(lib, [[1]]), _mx(lib, [[5]]), _mx(lib, [[nt // 2]]),
                       _mx(lib, np.ones((1, 16)) / 16), _mx(lib, [[0, 0]]))
    assert rc == 1 and "odd number of taps" in err


@pytest.mark.gpu
def test_scalar_ssfm_shim_with_xpm():
    from oracle import plxo as oracle
    from polmux_amd import synth
    lib = _load("plx_scalar_ssfm_mex")
    nsymb, nt, nfc = 64, 16, 3
    n = nsymb * nt
    u = np.stack([synth.pdm_qpsk_field(nsymb, nt, 4.0 + k, 2 + k, 7 + k)[0] for k in range(nfc)], 1)
    omega = 2 * np.pi * 28 * synth.fn_grid(nsymb, nt)
    betat = np.stack([0.5 * omega ** 2 * -2.17e-8 + 6.8e-9 * k * omega for k in range(nfc)], 1)
    gam = np.array([[1.2e-6, 1.3e-6, 1.25e-6]])
    rc, out, err = _call(lib, 3, _mx(lib, u), _mx(lib, betat), _mx(lib, [[1e4]]), _mx(lib, [[5e-3]]), _mx(lib, gam), _mx(lib, [[4.6e-5]]),
                         _mx(lib, [[nfc]]), _mx(lib, [[1e4]]), _mx(lib, [[1, 0, 1, 1]]))
    assert rc == 0, err
    ofd, onc, ou = oracle.scalar_ssfm(u, betat, 1e4, 5e-3, gam[0], 4.6e-5, 1e4, [1, 0, 1, 1])
    assert _np(out[1])[0, 0] == onc and _np(out[0])[0, 0] == pytest.approx(ofd, rel=1e-12)
    assert np.abs(_np(out[2]) - ou).max() <= 1e-9 * np.abs(ou).max()


@pytest.mark.gpu
def test_easi_shim_six_inputs_in_place_real_parts():
    """easiadaptivefilter.c:95-169 calling convention: six inputs (sps is prhs[5]), in-place taps, 0, 0 returned; only the
    real parts of tap 0 move (:81-90)."""
    from oracle import plxo as oracle
    lib = _load("plx_easiadaptivefilter_mex")
    r = np.random.default_rng(9)
    L = 150
    a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (L, 2))))
    xx = a @ np.array([[np.cos(0.2), np.sin(0.2)], [-np.sin(0.2), np.cos(0.2)]]) + 0.05 * (r.standard_normal((L, 2)) + 1j * r.standard_normal((L, 2)))
    h1 = np.array([[0.9 + 0.1j, 0.2 - 0.3j]]); h2 = np.array([[-0.2 + 0.05j, 1.1 + 0.2j]])
    mh1, mh2 = _mx(lib, h1), _mx(lib, h2)
    rc, out, err = _call(lib, 3, _mx(lib, xx), mh1, mh2, _mx(lib, [[1]]), _mx(lib, [[1e-2]]), _mx(lib, [[1]]))
    assert rc == 0, err
    y, g1, g2 = oracle.easiadaptivefilter(xx, h1, h2, 1, 1e-2, 1)
    np.testing.assert_allclose(_np(out[0]), y, atol=1e-11)
    np.testing.assert_allclose(_np(mh1), g1, atol=1e-11)
    np.testing.assert_allclose(_np(mh2), g2, atol=1e-11)
    np.testing.assert_array_equal(_np(mh1).imag, h1.imag)
    assert _np(out[1]).tolist() == [[0.0]] and _np(out[2]).tolist() == [[0.0]]
    rc, _, err = _call(lib, 3, _mx(lib, xx), mh1, mh2, _mx(lib, [[1]]), _mx(lib, [[1e-2]]))
    assert rc == 1 and err == "Six inputs required."


@pytest.mark.gpu
def test_mfile_twin_shims_return_the_updated_taps():
    """The shims with the semantics of the .m twins (cmaadaptivefilter.m:52-72, easiadaptivefilter.m:51-84): the taps come
    back in plhs[1..2] (non-zero, so the unchanged drivers take them, DspPdmCohQpsk.m:183-186), the inputs are untouched,
    sps = 2 still updates every sample, an even number of taps is accepted."""
    from oracle import plxo as oracle
    r = np.random.default_rng(6)
    taps, L = 4, 120
    a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (L + taps - 1, 2))))
    xx = a @ np.array([[np.cos(0.3), np.sin(0.3)], [-np.sin(0.3), np.cos(0.3)]])
    h1 = np.zeros((taps, 2), complex); h1[1, 0] = 1
    h2 = np.zeros((taps, 2), complex); h2[1, 1] = 1
    lib = _load("plx_cmaadaptivefilter_m_mex")
    mh1, mh2 = _mx(lib, h1), _mx(lib, h2)
    rc, out, err = _call(lib, 3, _mx(lib, xx), mh1, mh2, _mx(lib, [[taps]]), _mx(lib, [[5e-3]]), _mx(lib, [[1.0, 1.0]]), _mx(lib, [[2]]))
    assert rc == 0, err
    y, g1, g2 = oracle.cmaadaptivefilter_m(xx, h1, h2, taps, 5e-3, [1.0, 1.0])
    np.testing.assert_allclose(_np(out[0]), y, atol=1e-11)
    np.testing.assert_allclose(_np(out[1]), g1, atol=1e-11)
    np.testing.assert_allclose(_np(out[2]), g2, atol=1e-11)
    np.testing.assert_array_equal(_np(mh1), h1)                   # inputs untouched
    lib = _load("plx_easiadaptivefilter_m_mex")
    k1 = np.array([[0.9 + 0.1j, 0.2 - 0.3j]]); k2 = np.array([[-0.2 + 0.05j, 1.1 + 0.2j]])
    x1 = xx[:L]
    rc, out, err = _call(lib, 3, _mx(lib, x1), _mx(lib, k1), _mx(lib, k2), _mx(lib, [[1]]), _mx(lib, [[1e-2]]), _mx(lib, [[1]]))
    assert rc == 0, err
    y, g1, g2 = oracle.easiadaptivefilter_m(x1, k1, k2, 1, 1e-2)
    np.testing.assert_allclose(_np(out[0]), y, atol=1e-11)
    np.testing.assert_allclose(_np(out[1]), g1, atol=1e-11)
    np.testing.assert_allclose(_np(out[2]), g2, atol=1e-11)
    assert np.abs(_np(out[1]).imag - k1.imag).max() > 1e-5


@pytest.mark.gpu
def test_poldemux_driver_shims_make_the_whole_pass_loop_in_one_call():
    """plx_cmapolardemux_mex / plx_easipolardemux_mex: y = cmapolardemux(x, params) / easipolardemux(x, params) as ONE MEX
    call each (DspPdmCohQpsk.m:142-244): output of the last pass, final taps and the number of passes equal the oracle's
    driver loop around the per-pass filter (C1's L = 1024, 7 taps, mu = 1/6000 among the cases: 47 passes on this input)."""
    from oracle import plxo as oracle
    lib = _load("plx_cmapolardemux_mex")
    r = np.random.default_rng(3)
    for L, taps, mu, noise, phi in ((1024, 7, 1 / 6000, 0.05, 0.0), (400, 5, 1 / 500, 0.0, 0.3), (256, 1, 1 / 300, 0.02, 0.0)):
        a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (L, 2))))
        x = a @ np.array([[np.cos(0.35), np.sin(0.35)], [-np.sin(0.35), np.cos(0.35)]]) + noise * (r.standard_normal((L, 2)) + 1j * r.standard_normal((L, 2)))
        M = np.array([[np.cos(phi), np.sin(phi)], [-np.sin(phi), np.cos(phi)]])        # :157-158
        rc, out, err = _call(lib, 4, _mx(lib, x), _mx(lib, [[1.0, 1.0]]), _mx(lib, [[mu]]), _mx(lib, [[taps]]), _mx(lib, M))
        assert rc == 0, err
        oy, h1, h2, n = oracle.cmapolardemux(x, M, taps, mu, [1.0, 1.0])
        assert int(_np(out[3])[0, 0]) == n
        assert 1 < n < 50 * int(np.ceil(1 / (L * mu)))                                 # converged inside the budget of :175-176
        np.testing.assert_allclose(_np(out[0]), oy, atol=1e-9)
        np.testing.assert_allclose(_np(out[1]), h1.reshape(taps, 2), atol=1e-9)
        np.testing.assert_allclose(_np(out[2]), h2.reshape(taps, 2), atol=1e-9)
    rc, _, err = _call(lib, 1, _mx(lib, x), _mx(lib, [[1.0, 1.0]]), _mx(lib, [[1e-3]]), _mx(lib, [[4]]), _mx(lib, M))
    assert rc == 1 and err == "Ntaps should be an ODD INTEGER."
    rc, _, err = _call(lib, 1, _mx(lib, x), _mx(lib, [[1.0, 1.0]]), _mx(lib, [[1e-3]]), _mx(lib, [[3]]))
    assert rc == 1 and err == "Five inputs required."
    assert lib.mexstub_lock_count() == 1
    lib2 = _load("plx_easipolardemux_mex")
    L, mu = 512, 1 / 400
    a = np.exp(1j * (np.pi / 4 + np.pi / 2 * r.integers(0, 4, (L, 2))))
    x = a @ np.array([[np.cos(0.2), np.sin(0.2)], [-np.sin(0.2), np.cos(0.2)]]) + 0.03 * (r.standard_normal((L, 2)) + 1j * r.standard_normal((L, 2)))
    M = np.eye(2) + 0j
    rc, out, err = _call(lib2, 4, _mx(lib2, x), _mx(lib2, [[mu]]), _mx(lib2, M))
    assert rc == 0, err
    oy, h1, h2, n = oracle.easipolardemux(x, M, mu)
    assert int(_np(out[3])[0, 0]) == n
    np.testing.assert_allclose(_np(out[0]), oy, atol=1e-9)
    np.testing.assert_allclose(_np(out[1]).ravel(), np.ravel(h1), atol=1e-9)
    rc, out, err = _call(lib2, 4, _mx(lib2, x), _mx(lib2, [[mu]]), _mx(lib2, M), _mx(lib2, [[1]]))     # around the .m twin of the filter
    assert rc == 0, err
    oy, h1, h2, n = oracle.easipolardemux_m(x, M, mu)
    assert int(_np(out[3])[0, 0]) == n
    np.testing.assert_allclose(_np(out[0]), oy, atol=1e-9)


@pytest.mark.gpu
def test_scalar_a_ssfm_shim_and_release_request():
    """plx_scalar_a_ssfm_mex: [firstdz,ncycle,u,nrej] = scalar_a_ssfm(...) (fiber.m:639-679, 938-1009) against the oracle;
    then <shim>('release'): the library's state is freed and the MEX file unlocked (clearable), and the next call works."""
    from oracle import plxo as oracle
    from polmux_amd import _abi, synth
    lib = _load("plx_scalar_a_ssfm_mex")
    nsymb, nt = 64, 16
    u = synth.pdm_qpsk_field(nsymb, nt, 12.0)[0].reshape(-1, 1)
    omega = 2 * np.pi * 28 * synth.fn_grid(nsymb, nt)
    betat = (0.5 * omega ** 2 * -2.17e-8).reshape(-1, 1)
    gam = [1.3e-6]

    def call():
        return _call(lib, 4, _mx(lib, u), _mx(lib, betat), _mx(lib, [[4e4]]), _mx(lib, [[np.inf]]), _mx(lib, [gam]), _mx(lib, [[4.6e-5]]),
                     _mx(lib, [[1]]), _mx(lib, [[4e4]]), _mx(lib, [[1, 0, 1, 0]]), _mx(lib, [[2]]), _mx(lib, [[1e-6]]), _mx(lib, [[0.9]]))
    rc, out, err = call()
    assert rc == 0, err
    ofd, onc, onrej, ou = oracle.scalar_a_ssfm(u, betat, 4e4, np.inf, gam, 4.6e-5, 4e4, 1e-6, 0.9, [1, 0, 1, 0])
    assert int(_np(out[1])[0, 0]) == onc and onc > 3 and int(_np(out[3])[0, 0]) == onrej
    assert _np(out[0])[0, 0] == pytest.approx(ofd, rel=1e-9)
    assert np.abs(_np(out[2]) - ou).max() <= 1e-9 * np.abs(ou).max()
    first = _np(out[2])
    assert lib.mexstub_lock_count() == 1 and _gw_stats()["dev_bytes"] > 0
    rc, _, err = _call(lib, 0, lib.mxCreateString(b"release"))
    assert rc == 0, err
    assert lib.mexstub_lock_count() == 0 and _gw_stats()["dev_bytes"] == 0      # clearable, nothing held
    rc, _, err = _call(lib, 0, lib.mxCreateString(b"relax"))
    assert rc == 1 and err == "Twelve inputs required."                          # any other string is just a bad call
    rc, out, err = call()
    assert rc == 0, err
    np.testing.assert_array_equal(_np(out[2]), first)
    assert lib.mexstub_lock_count() == 1                                          # locked again by the first ordinary call
    _abi.get().call("plx_release_all")


def _gw_stats():
    from polmux_amd import _abi
    out = np.zeros(8, np.int64)
    _abi.get().call("plx_gateway_stats", out.ctypes.data)
    return dict(zip(("calls", "dev_allocs", "host_allocs", "plan_builds", "plan_hits", "dev_bytes", "host_bytes", "releases"), out.tolist()))


@pytest.mark.gpu
def test_gateway_state_is_kept_between_calls_and_released_at_exit():
    """SURVEY 8(b) "Ownership": the unchanged drivers call the filter MEX once per pass (DspPdmCohQpsk.m:176-191) and the
    propagator once per span (fiber.m:372-389).  The SECOND call of a gateway allocates nothing and builds no plan (the
    library keeps plans, device buffers and pinned staging); the shim holds the MEX file with mexLock and releases the
    state from mexAtExit; after a release the next call works again and gives the same result."""
    from polmux_amd import _abi, synth
    lib = _load("plx_cmaadaptivefilter_mex")
    _abi.get().call("plx_release_all")
    r = np.random.default_rng(11)
    xx = (r.standard_normal((1024, 2)) + 1j * r.standard_normal((1024, 2))) / np.sqrt(2)
    taps = 7
    h1 = np.zeros((taps, 2)); h1[taps // 2, 0] = 1.0
    h2 = np.zeros((taps, 2)); h2[taps // 2, 1] = 1.0

    def one_pass():
        mh1, mh2 = _mx(lib, h1), _mx(lib, h2)
        rc, out, err = _call(lib, 3, _mx(lib, xx), mh1, mh2, _mx(lib, [[taps]]), _mx(lib, [[1e-3]]), _mx(lib, [[1.0, 1.0]]), _mx(lib, [[1]]))
        assert rc == 0, err
        return _np(out[0]), _np(mh1), _np(mh2)
    first = one_pass()
    s1 = _gw_stats()
    assert s1["dev_allocs"] >= 1 and s1["host_allocs"] >= 1 and s1["dev_bytes"] > 0
    for _ in range(5):                                    # the driver's pass loop
        again = one_pass()
    s2 = _gw_stats()
    assert s2["calls"] == s1["calls"] + 5
    assert (s2["dev_allocs"], s2["host_allocs"], s2["plan_builds"]) == (s1["dev_allocs"], s1["host_allocs"], s1["plan_builds"])
    for a, b in zip(first, again):
        np.testing.assert_array_equal(a, b)
    assert lib.mexstub_lock_count() == 1                  # mexLock once, not once per call
    # the propagator: one plan per fibre type and grid, found again by content
    lib2 = _load("plx_matrix_ssfm_mex")
    nsymb, nt = 64, 16
    ux, uy, _, _ = synth.pdm_qpsk_field(nsymb, nt, 6.0)
    omega = 2 * np.pi * 28 * synth.fn_grid(nsymb, nt)
    betat = (0.5 * omega ** 2 * -2.17e-8).reshape(-1, 1)

    def span(bt):
        args = [_mx(lib2, ux.reshape(-1, 1)), _mx(lib2, uy.reshape(-1, 1)), _mx(lib2, bt), _mx(lib2, np.zeros_like(bt)), _mx(lib2, [[2e4]]),
                _mx(lib2, [[5e-3]]), _mx(lib2, [[1.3e-6]]), _mx(lib2, [[4.6e-5]]), _mx(lib2, [[1]]), _mx(lib2, [[2e4]]), _mx(lib2, [[1]]),
                _mx(lib2, [[0]]), _mx(lib2, [[1, 0, 1, 0]]), _mx(lib2, [[0.0]]), _mx(lib2, [[0.0]]), _mx(lib2, [[0.0]])]
        rc, out, err = _call(lib2, 4, *args)
        assert rc == 0, err
        return _np(out[2]), _np(out[3]), _np(out[1])[0, 0]
    a = span(betat)
    s3 = _gw_stats()
    assert s3["plan_builds"] == s2["plan_builds"] + 1
    b = span(betat.copy())                                # the same fibre in ANOTHER host array: found by content
    s4 = _gw_stats()
    assert s4["plan_builds"] == s3["plan_builds"] and s4["plan_hits"] == s3["plan_hits"] + 1
    assert (s4["dev_allocs"], s4["host_allocs"]) == (s3["dev_allocs"], s3["host_allocs"])
    np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1]); assert a[2] == b[2]
    c = span(betat * 1.01)                                # another fibre: its own plan
    assert _gw_stats()["plan_builds"] == s4["plan_builds"] + 1 and np.abs(c[0] - a[0]).max() > 0
    # `clear mex`: the registered exit function releases everything; the library rebuilds on demand
    assert lib.mexstub_run_atexit() == 1
    s5 = _gw_stats()
    assert s5["releases"] == s4["releases"] + 1 and s5["dev_bytes"] == 0 and s5["host_bytes"] == 0
    for x_, y_ in zip(first, one_pass()):
        np.testing.assert_array_equal(x_, y_)
    d = span(betat)
    np.testing.assert_array_equal(a[0], d[0])
    _abi.get().call("plx_release_all")

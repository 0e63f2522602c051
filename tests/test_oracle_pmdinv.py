"""CPU tests: oracle/pmdinv.py (inverse_pmd.m) against the identities the reference states, using the oracle's
own fibre (fiber.m matrix_ssfm) as the forward link."""
import numpy as np
import pytest

from oracle import plxo as oracle
from oracle import pmdinv
from polmux_amd import synth


def _brf(nplates, seed):
    r = np.random.default_rng(seed)
    return (r.random(nplates) * 2 * np.pi - np.pi, r.random(nplates) * np.pi - np.pi / 2,
            0.5 * np.arcsin(r.random(nplates) * 2 - 1))


def _link(nsymb, nt, nplates, seed, length=5e4, dgd=0.3):
    fn = synth.fn_grid(nsymb, nt)
    omega = 2 * np.pi * 10.0 * fn
    betat = (0.5 * omega ** 2 * -2.17e-8 + omega ** 3 * 1.3e-10 / 6).reshape(-1, 1)
    db1 = (dgd / nplates / 10.0 * omega).reshape(-1, 1)
    db0, th, ep = _brf(nplates, seed)
    return dict(db0=db0, theta=th, epsilon=ep, lcorr=length / nplates, betat=betat, db1=db1, length=length, nplates=nplates)


def _fibre(b, ux, uy, alphalin=4.6e-5):
    rc, fd, nc, ox, oy = oracle.matrix_ssfm(ux, uy, b["betat"], b["db1"], b["length"], 5e-3, [0.0], alphalin, b["length"], b["nplates"],
                                            False, [1, 1, 0, 0], b["db0"], b["theta"], b["epsilon"])
    assert rc == 0 and nc == 1
    return ox[:, 0], oy[:, 0]


def test_u_is_special_unitary_and_uinv_its_inverse():
    b = _link(64, 16, 12, 3)
    Uinv, U, _, _ = pmdinv.inverse_pmd([b], None, None, dict(apply="no"))
    assert U.shape == (2, 2, 1024)
    for k in (0, 1, 511, 1023):
        np.testing.assert_allclose(Uinv[:, :, k] @ U[:, :, k], np.eye(2), atol=1e-13)
        assert np.linalg.det(U[:, :, k] / np.exp(1j * np.angle(np.linalg.det(U[:, :, k])) / 2)) == pytest.approx(1.0, abs=1e-12)
    Uinv2, U2, _, _ = pmdinv.inverse_pmd([b], None, None, dict(apply="no", gvd="no"))
    np.testing.assert_array_equal(U2[1, 0], -np.conj(U2[0, 1]))                # update_U :155-156 (without the GVD factor)
    np.testing.assert_array_equal(U2[1, 1], np.conj(U2[0, 0]))
    hg = np.exp(-1j * b["betat"][:, 0] * b["length"])
    np.testing.assert_allclose(U, hg * U2, atol=1e-12)                         # :130-134


def test_inverse_pmd_restores_the_field_after_linear_pmd_fibre():
    """fiber(.,'gp--') then inverse_pmd(brf): the input field up to the attenuation (SURVEY 8c iv)."""
    ux, uy, _, _ = synth.pdm_qpsk_field(64, 16, 2.0)
    b = _link(64, 16, 20, 5)
    ox, oy = _fibre(b, ux, uy)
    _, _, rx, ry = pmdinv.inverse_pmd([b], ox, oy)
    att = np.exp(-0.5 * 4.6e-5 * b["length"])
    np.testing.assert_allclose(rx, ux * att, atol=1e-11)
    np.testing.assert_allclose(ry, uy * att, atol=1e-11)
    # two fibres in cascade, inverted by one call with brf = {brf1, brf2}
    b2 = _link(64, 16, 7, 6, length=3e4, dgd=0.5)
    px_, py_ = _fibre(b2, ox, oy)
    _, _, rx, ry = pmdinv.inverse_pmd([b, b2], px_, py_)
    att2 = att * np.exp(-0.5 * 4.6e-5 * b2["length"])
    np.testing.assert_allclose(rx, ux * att2, atol=1e-11)
    np.testing.assert_allclose(ry, uy * att2, atol=1e-11)
    # options.gvd = 'no': PMD undone, the scalar dispersion of the link remains
    _, _, rx, ry = pmdinv.inverse_pmd([b], ox, oy, dict(gvd="no"))
    want = np.fft.ifft(np.fft.fft(ux) * np.exp(-1j * b["betat"][:, 0] * b["length"])) * att
    np.testing.assert_allclose(rx, want, atol=1e-11)


def test_options_apply_and_mat_follow_the_reference_text():
    ux, uy, _, _ = synth.pdm_qpsk_field(64, 16, 2.0)
    b = _link(64, 16, 5, 8)
    assert pmdinv.inverse_pmd([b], ux, uy, dict(apply="no"))[2] is None       # :138: 'no' does not apply ...
    assert pmdinv.inverse_pmd([b], ux, uy, dict(apply="n"))[2] is not None    # ... 'n' does (as written)
    assert pmdinv.inverse_pmd([b], ux, uy, dict(gvd="no"))[2] is not None
    c, s = np.cos(0.4), np.sin(0.4)
    M = np.array([[c, s], [-s, c]], dtype=complex)
    Uinv, U, _, _ = pmdinv.inverse_pmd([b], None, None, dict(apply="no", mat=M))
    _, U0, _, _ = pmdinv.inverse_pmd([b], None, None, dict(apply="no"))
    for k in (0, 77, 1023):
        np.testing.assert_allclose(U[:, :, k], U0[:, :, k] @ M, atol=1e-13)   # reference system rotated first, :105-107

"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of /root/reference/inverse_pmd.m:91-168.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned by the reference's stated identities: Uinv = U' is unitary per frequency (inverse_pmd.m:39-40,
:135-136) and inverse_pmd(brf) after fiber(.,'gp--') restores the input field up to the attenuation
(inverse_pmd.m:1-6 of the help; SURVEY 8c iv, v) -- tests/test_oracle_pmdinv.py.  No golden vectors exist.
"""
import numpy as np

_SIG0 = np.eye(2)
_SIG2 = np.array([[0.0, 1.0], [1.0, 0.0]])
_SIG3I = np.array([[0.0, 1.0], [-1.0, 0.0]])


def getmatR(theta, epsilon):
    """:164-168"""
    rth = np.cos(theta) * _SIG0 - np.sin(theta) * _SIG3I
    reps = np.cos(epsilon) * _SIG0 + 1j * np.sin(epsilon) * _SIG2
    return rth @ reps


def update_U(l1, l2, matR, Uold):
    """:143-161 (U is [2,2,Nfft]; rows 2 are forced to the SU(2) form, l2 is not used)"""
    t11, t12 = l1 * matR[0, 0], l1 * matR[0, 1]
    U = np.empty_like(Uold)
    U[0, 0] = t11 * Uold[0, 0] + t12 * Uold[1, 0]
    U[0, 1] = t11 * Uold[0, 1] + t12 * Uold[1, 1]
    U[1, 0] = -np.conj(U[0, 1])
    U[1, 1] = np.conj(U[0, 0])
    return U


def inverse_pmd(brf, fieldx, fieldy, options=None):
    """brf: list of dicts (db0, theta, epsilon, lcorr, betat [Nfft], db1 [Nfft]).  Returns (Uinv, U, outx, outy);
    outx/outy are None when the reference would not apply the matrix (:138)."""
    nfft = np.asarray(brf[0]["betat"]).reshape(-1).size
    isopt = options is not None
    isnotgvd = isopt and options.get("gvd") == "no"
    U = np.zeros((2, 2, nfft), dtype=complex)
    U[0, 0] = 1
    U[1, 1] = 1
    allgvd = np.zeros(nfft)
    one = np.ones(nfft)
    if isopt and "mat" in options:
        U = update_U(one, one, np.asarray(options["mat"], dtype=complex), U)          # :105-107
    for b in brf:                                                                      # :110-133
        theta, eps, db0 = (np.atleast_1d(np.asarray(b[k], dtype=float)) for k in ("theta", "epsilon", "db0"))
        db1 = np.asarray(b["db1"], dtype=float).reshape(-1)
        ntrunk = theta.size
        matR = getmatR(theta[0], eps[0])
        deltabeta = 0.5 * (db1 + db0[0])
        l1 = np.cos(-deltabeta) + 1j * np.sin(-deltabeta)                              # fastexp(-deltabeta)
        U = update_U(l1, 1.0 / l1, matR.conj().T, U)
        for k in range(1, ntrunk):
            matR = getmatR(theta[k], eps[k]).conj().T @ getmatR(theta[k - 1], eps[k - 1])
            deltabeta = 0.5 * (db1 + db0[k])
            l1 = np.cos(-deltabeta) + 1j * np.sin(-deltabeta)
            U = update_U(l1, 1.0 / l1, matR, U)
        U = update_U(one, one, getmatR(theta[-1], eps[-1]), U)
        allgvd = allgvd + np.asarray(b["betat"], dtype=float).reshape(-1) * b["lcorr"] * ntrunk
    hgvd = np.cos(-allgvd) + 1j * np.sin(-allgvd)
    if not isnotgvd:
        U = hgvd * U
    Uinv = np.conj(np.transpose(U, (1, 0, 2)))                                         # :135-136
    outx = outy = None
    if (not isopt) or ("apply" not in options) or options["apply"] == "n":             # :138 (as written)
        uux, uuy = np.fft.fft(fieldx), np.fft.fft(fieldy)
        outx = np.fft.ifft(Uinv[0, 0] * uux + Uinv[0, 1] * uuy)
        outy = np.fft.ifft(Uinv[1, 0] * uux + Uinv[1, 1] * uuy)
    return Uinv, U, outx, outy

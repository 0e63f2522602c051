"""Synthetic PDM-QPSK transmitter for tests, bench.py and the Monte-Carlo runner.

Host-side numpy, O(Nfft) once per campaign -- not part of the accelerated path
(SURVEY 8f-3).  It follows the reference's Tx chain closely enough that the
waveforms are the ones the hot path sees in Run_my_PDM_QPSK.m:100-117:

  pattern('debruijn',seed,struct('alphabet',4))   pattern.m:104-133, 281-353
  electricsource(bits,'qpsk',R,'cosroll',duty,roll) electricsource.m:133-147, 187-269
  qi_modulator(E,I,Q) with default options          qi_modulator.m:48-101, mz_modulator.m:60-75
  create_field('sepfields',X,Y,struct('power','average'))  create_field.m:113-124
"""
import numpy as np


# ------------------------------------------------------------------ pattern.m ---
def _xor_iscan(x):
    return (np.cumsum(x) % 2) == 1                      # pattern.m:335-341


def _next_debruijn(w, i, k):
    C = _xor_iscan(w)                                   # pattern.m:323-331 (1-based slices)
    Cbar = ~C
    return np.concatenate([C[: i - k], Cbar[i + k - 1:], Cbar[: i - 1 + k], C[i - k:]])


def _generate_debruijn(n, x):
    if n == 2:                                          # pattern.m:310-321
        return np.array([1, 1, 0, 0]) == 1
    if n == 3:
        return (np.array([1, 0, 1, 1, 1, 0, 0, 0]) if x[0] == 0 else np.array([1, 1, 1, 0, 1, 0, 0, 0])) == 1
    return _next_debruijn(_generate_debruijn(n - 1, x), 2 ** (n - 2) + (-1) ** int(x[n - 4]), int(x[n - 3]))


def _fastshift_row(y, n):
    return np.roll(y, n, axis=-1)                       # fastshift.m: circshift semantics


def pattern_debruijn(nsymb, seed, alphabet=2):
    """[pat,bmat] = pattern('debruijn',seed,struct('alphabet',alphabet)); bmat is [nsymb x q]."""
    q = int(round(np.log2(alphabet)))
    lg2 = int(round(np.log2(nsymb)))
    if 2 ** lg2 != nsymb or lg2 % q:
        raise ValueError("The De Bruijn sequence does not exist! log2(number of symbols)=%d is not a "
                         "multiple of log2(alphabet length)=%d." % (lg2, q))
    maxseed = nsymb * (nsymb - 1) / 4
    if seed > maxseed:
        raise ValueError("nseed must be <= %d" % maxseed)
    ns = lg2                                            # q*n, pattern.m:298
    N = 2 ** (ns - 2)
    nseed = seed % N
    x = np.array([int(b) for b in np.binary_repr(nseed, ns - 2)]) if ns > 2 else np.zeros(0, int)
    y = _generate_debruijn(ns, x).astype(float)
    tmat = None
    if q > 1:
        mvm = np.floor(ns / q * np.arange(q)).astype(int)
        tmat = np.stack([_fastshift_row(y, int(m)) for m in mvm])
        y = (2.0 ** np.arange(q - 1, -1, -1)) @ tmat
    if seed > N:
        nseed2 = int(np.ceil((seed - N + 1) / N))
        nshift = (97 * nseed2) % (2 ** ns - 1) + 1
        y = _fastshift_row(y, -nshift)
        if tmat is not None:
            tmat = _fastshift_row(tmat, -nshift)
    pat = y.astype(int)
    bmat = tmat.T.astype(int) if tmat is not None else pat.reshape(-1, 1)
    return pat, bmat


def myseq(locpat, nsymb):
    """pattern.m:236-258: periodic repetition of locpat up to nsymb symbols, truncated if necessary (user-supplied
    patterns, pattern(array, ...))."""
    locpat = np.asarray(locpat).reshape(-1)
    lp = locpat.size
    if lp > nsymb:
        return locpat[:nsymb].copy()
    return np.concatenate([np.tile(locpat, nsymb // lp), locpat[: nsymb % lp]])


# ------------------------------------------------------------- electricsource.m ---
def _pulse_cosroll(roll, duty, nt):
    el = np.zeros(2 * nt)                               # electricsource.m:246-262
    nl = int(round(0.5 * (1 - roll) * duty * nt))
    nr = int(duty * nt - nl - 1)
    el[nt: nt + nl] = 1
    hperiod = duty * nt - 2 * nl
    if hperiod != 0:
        ncos = np.arange(nl, nr + 1)
        el[ncos + nt] = 0.5 * (1 + np.cos(np.pi / hperiod * (ncos - nl + 0.5)))
    el[:nt] = el[nt:][::-1]
    return el


def electricsource_qpsk(bits, nt, duty=1.0, roll=0.2):
    """electricsource(bits,'qpsk',R,'cosroll',duty,roll): +-1 NRZ with raised-cosine edges."""
    pat = 2.0 * np.asarray(bits, dtype=float) - 1.0     # :143
    nsymb = pat.size
    nfft = nsymb * nt
    el = _pulse_cosroll(roll, duty, nt)
    elec = np.zeros(nfft)
    elec[nfft - nt:] = pat[0] * el[:nt]                 # :219-222 first pulse wraps cyclically
    elec[:nt] = pat[0] * el[nt:]
    for k in range(1, nsymb):                           # :223-227
        elec[(k - 1) * nt:(k + 1) * nt] += pat[k] * el
    return elec


def qi_modulator(E, I, Q):
    """qi_modulator with default options: (E/sqrt2)*(sin(pi/2*I) + i*sin(pi/2*Q))."""
    return E / np.sqrt(2.0) * (np.sin(0.5 * np.pi * I) + 1j * np.sin(0.5 * np.pi * Q))


def pdm_qpsk_field(nsymb, nt, pavg_mw, seed_x=2, seed_y=3):
    """One channel of Run_my_PDM_QPSK.m:101-117.  Returns (ux, uy, bits[nsymb x 4], power_mw):
    power_mw is GSTATE.POWER after create_field's 'average' normalisation."""
    _, bx = pattern_debruijn(nsymb, seed_x, 4)
    _, by = pattern_debruijn(nsymb, seed_y, 4)
    bits = np.concatenate([bx, by], axis=1)
    carrier = np.sqrt(pavg_mw)                          # lasersource.m:178
    sx = qi_modulator(carrier, electricsource_qpsk(bits[:, 0], nt), electricsource_qpsk(bits[:, 1], nt))
    sy = qi_modulator(carrier, electricsource_qpsk(bits[:, 2], nt), electricsource_qpsk(bits[:, 3], nt))
    avge = np.mean(np.abs(sx) ** 2 + np.abs(sy) ** 2)   # create_field.m:113-124
    k = np.sqrt(pavg_mw / avge)
    return sx * k, sy * k, bits, pavg_mw * pavg_mw / avge


# --------------------------------------------------------------- fiber.m tables ---
CLIGHT = 299792458.0                                    # reset_all.m:105


def fn_grid(nsymb, nt):
    """GSTATE.FN = fftshift(-Nt/2:1/Nsymb:Nt/2-1/Nsymb), reset_all.m:152-153."""
    stepf = 1.0 / nsymb
    return np.fft.fftshift(np.arange(-nt / 2, nt / 2, stepf)[: nsymb * nt])

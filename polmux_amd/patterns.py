"""Symbol-pattern bookkeeping around the hot path (host integer logic, O(Nsymb)):

  stars = pat2stars(pat, format, options)          pat2stars.m
  pat[, patmat] = stars2pat(stars, format)         stars2pat.m
  pat[, patmat] = pat_decoder(pat, modformat, options)   pat_decoder.m:34-85

used by the Monte-Carlo scripts (ex19/ex20: differential decoding of the transmitted and the decided pattern
before ber_estimate).  Constellation points are exact (+-1, +-i), so the equality tests of the reference hold.
"""
import numpy as np


def pat2stars(pat, fmt, options=None):
    binary = bool(options.get("binary", False)) if options else False
    pat = np.asarray(pat)
    if fmt in ("ook", "psbt"):
        if pat.max() > 1:
            raise ValueError("pattern of %s must be binary" % fmt)
        return np.zeros(pat.shape)
    if fmt in ("bpsk", "dpsk", "nf-dpsk"):
        if pat.max() > 1:
            raise ValueError("pattern of %s must be binary" % fmt)
        return np.where(pat == 0, 1.0, -1.0)
    if fmt in ("dqpsk", "nf-dqpsk", "qpsk"):
        if not binary:
            if pat.max() > 3:
                raise ValueError("pattern of %s must be quaternary" % fmt)
            lut = np.array([1, 1j, -1j, -1], dtype=complex)          # 0 -> 1, 1 -> i, 2 -> -i, 3 -> -1
            return lut[pat.astype(int)]
        if pat.ndim != 2 or pat.shape[1] != 2 or pat.max() > 1:
            raise ValueError("pattern must be a binary matrix with size [Nsymb,2]")
        lut = {(0, 0): 1, (0, 1): 1j, (1, 1): -1, (1, 0): -1j}
        return np.array([lut[(int(a), int(b))] for a, b in pat], dtype=complex)
    raise ValueError("unknown modulation format")


def stars2pat(stars, fmt):
    stars = np.asarray(stars)
    if fmt in ("ook", "psbt"):
        return np.zeros(stars.shape)
    if fmt in ("bpsk", "dpsk", "nf-dpsk"):
        return np.where(stars == -1, 1, 0)
    if fmt in ("dqpsk", "nf-dqpsk", "qpsk"):
        pat = np.zeros(stars.shape, dtype=int)
        pat[stars == 1j] = 1
        pat[stars == -1] = 3
        pat[stars == -1j] = 2
        patmat = np.zeros((stars.size, 2), dtype=int)
        patmat[stars.reshape(-1) == 1j] = (0, 1)
        patmat[stars.reshape(-1) == -1] = (1, 1)
        patmat[stars.reshape(-1) == -1j] = (1, 0)
        return pat, patmat
    raise ValueError("unknown modulation format")


def _fastshift(x, n):
    return np.roll(x, int(n), axis=0)


def pat_decoder(pat, modformat, options=None):
    """Differentially decoded pattern (pat_decoder.m:41-79); 'dqpsk' returns (pat, patmat)."""
    binary = bool(options.get("binary", False)) if options else False
    if modformat == "ook":
        return np.asarray(pat)
    if modformat == "dpsk":
        st = pat2stars(pat, "dpsk")
        return 1 - stars2pat(np.conj(st) * _fastshift(st, 1), "dpsk")
    if modformat in ("nf-dpsk", "psbt"):
        st = pat2stars(pat, "nf-dpsk")
        return _fastshift(1 - stars2pat(np.conj(st) * _fastshift(st, 1), "nf-dpsk"), -1)
    if modformat in ("dqpsk", "nf-dqpsk"):
        st = pat2stars(pat, "dqpsk", dict(binary=True)) if binary else pat2stars(pat, "dqpsk")
        p, pm = stars2pat(np.conj(st) * _fastshift(st, 1), "dqpsk")
        pm = 1 - pm                                                  # both patterns are inverted, :71-72
        p = 3 - p
        if modformat == "nf-dqpsk":
            p, pm = _fastshift(p, -1), _fastshift(pm, -1)
        return p, pm
    raise ValueError("wrong modulation format in pat_decoder")

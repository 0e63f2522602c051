"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the coherent front end.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(polmux_amd/) never does.

Follows, statement by statement, the data path of
  /root/reference/receiver_cohmix.m:183-307   (optical filter, hybrids, photodiodes, low-pass)
  /root/reference/RxPdmCohQpsk.m:36-72        (ADC, timing shift, decimation, I/Q recombination)
with the filter tables (Hf, Elo) supplied by the caller.  FFTs are numpy's (pocketfft); MATLAB's are FFTW:
both are plain DFTs, pinned to each other only to rounding.

PARITY UNPINNED for `decimate`: decimate(x,r,16,'fir') is MathWorks Signal Processing Toolbox code that is
not in /root/reference and has no version pin; `decimate_fir` below is this project's stated
definition (DESIGN.md "front end"), which the HIP kernel must match bit-for-bit up to FMA contraction.
receiver_cohmix itself is pinned by the reference's analytic identities only (balanced detection gives
4*Re, 4*Im of s*conj(Elo); unit filters make it the identity), tests/test_oracle_front.py.
"""
import numpy as np


def receiver_cohmix(sigx, sigy, hf_opt, elo, hf_el, balanced=True, ndfn=0):
    """Iric [n x 2] or [n x 4] = receiver_cohmix data path.  sigy None -> X only.  elo: scalar or [n].
    ndfn: channel offset in bins when the field is 'unique' (nind = nmod(npoints-ndfn, Nfft), :104-107)."""
    sigx = np.asarray(sigx, dtype=np.complex128)
    n = sigx.shape[0]
    elo = np.asarray(elo, dtype=np.complex128) * np.ones(n)
    nind = (np.arange(n) - ndfn) % n                                # 0-based nmod(npoints-ndfn, Nfft)
    sx = np.fft.fft(sigx)[nind] * hf_opt                            # :183-189
    isy = sigy is not None
    if isy:
        sy = np.fft.fft(np.asarray(sigy, dtype=np.complex128))[nind] * hf_opt   # :238-242
        sy = np.fft.ifft(sy)                                        # :246
    sx = np.fft.ifft(sx)                                            # :247 / :252

    def currents(s):
        emix = np.stack([s * 1j + elo * 1j, s - elo, s * 1j - elo, -s + elo * 1j], 1)   # :262-265, :283-286
        iric = np.real(emix * np.conj(emix))                        # :274, :287
        if balanced:
            return np.stack([iric[:, 0] - iric[:, 1], iric[:, 2] - iric[:, 3]], 1)     # :277, :289
        return np.stack([iric[:, 0], iric[:, 2]], 1)                # :279, :291

    hf2 = np.asarray(hf_el, dtype=np.complex128)[:, None] * np.ones((1, 2))           # :296
    out = np.real(np.fft.ifft(np.fft.fft(currents(sx), axis=0) * hf2, axis=0))        # :300
    if isy:
        iy = np.real(np.fft.ifft(np.fft.fft(currents(sy), axis=0) * hf2, axis=0))     # :304
        out = np.concatenate([out, iy], 1)                          # :305
    return out


def adc(irx, bits):
    """RxPdmCohQpsk.m:36-40 (MATLAB round = half away from zero; the argument is >= 0 here)."""
    M = np.max(np.abs(irx))
    q = (irx + M) / 2 / M * 2 ** bits
    fl = np.floor(q)
    rq = fl + ((q - fl) >= 0.5)                                      # round(), exact for q >= 0
    return rq * 2 * M / 2 ** bits - M


def fastshift(x, n):
    """fastshift.m:52-60: y(i) = x(i-n) circularly, column-wise."""
    return np.roll(x, int(n), axis=0)


def decimate_fir(x, r, b):
    """This project's definition of decimate(x,r,numel(b)-1,'fir') for one real column (see module header):
    odd reflection about both end points, causal FIR (oldest sample accumulated first), output taken at
    0-based positions gd + m*r, gd = (numel(b)-1)/2, m = 0..ceil(n/r)-1."""
    x = np.asarray(x, dtype=np.float64)
    n, nt = x.size, len(b)
    gd = (nt - 1) // 2
    nout = -(-n // r)
    out = np.zeros(nout)
    for m in range(nout):
        p = gd + m * r
        acc = 0.0
        for k in range(nt - 1, -1, -1):
            i = p - k
            if i < 0:
                v = 2 * x[0] - x[min(-i, n - 1)]
            elif i >= n:
                v = 2 * x[n - 1] - x[max(2 * (n - 1) - i, 0)]
            else:
                v = x[i]
            acc = acc + b[k] * v
        out[m] = acc
    return out


def rx_front(irxt, isy, adcbits, shifts, r, b):
    """RxPdmCohQpsk.m:27-72 from the photocurrents to RxSamples [nout x (1 or 2)]."""
    irx = irxt if isy else irxt[:, :2]                               # :27-33
    if adcbits:
        irx = adc(irx, adcbits)                                      # :36-40
    irx = irx.copy()
    for npol in range(irx.shape[1] // 2):                            # :42-44
        irx[:, 2 * npol:2 * npol + 2] = fastshift(irx[:, 2 * npol:2 * npol + 2], shifts[npol])
    if r > 1:
        dec = np.stack([decimate_fir(irx[:, c], r, b) for c in range(irx.shape[1])], 1)   # :54-62
    else:
        dec = irx
    cols = [dec[:, 0] + 1j * dec[:, 1]]                              # :66 / :68 (COS_POL1=1, SIN_POL1=2)
    if dec.shape[1] == 4:
        cols.append(dec[:, 2] + 1j * dec[:, 3])                      # :69 (COS_POL2=3, SIN_POL2=4)
    return np.stack(cols, 1)


def disp_comp_filter(beta2l, B, N, filter_length):
    """RxPdmCohQpsk.m:90-98 DispCompFilter, statement by statement."""
    freq = -B / 2 + B / N * np.arange(N)
    freq = np.fft.ifftshift(freq)
    delay = 2 * np.pi * freq / B * (filter_length / 2)
    argum = (2 * np.pi * freq) ** 2 * beta2l / 2 - delay
    H = np.cos(argum) + 1j * np.sin(argum)
    b = np.fft.ifft(H)
    b = b[: int(filter_length) + 1]
    return np.fft.fft(b, N) * (np.cos(delay) + 1j * np.sin(delay))


def apply_dcf(rx, dispersion, lam, baudrate, ndispsym, workatbaudrate, clight=299792458.0):
    """RxPdmCohQpsk.m:74-84"""
    beta2l = -dispersion * lam ** 2 / 2 / np.pi / clight * 1e-21
    sps = 1 + (0 if workatbaudrate else 1)
    H = disp_comp_filter(beta2l, sps * baudrate, rx.shape[0], ndispsym * sps)
    return np.fft.ifft(np.fft.fft(rx, axis=0) * H[:, None], axis=0)

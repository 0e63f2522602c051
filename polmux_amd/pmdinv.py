"""inverse_pmd(brf, options) -- inverse PMD matrix of a link (inverse_pmd.m:91-145), applied on the GPU.

brf: the struct fiber() returns, or a list of them (one per fibre of the link, in propagation order, each
with db0, theta, epsilon, lcorr, betat, db1).  GSTATE.FIELDX/FIELDY are updated in place in HBM; with two
requested outputs the [2,2,Nfft] matrices Uinv and U are downloaded (numpy).  No CPU implementation exists here.
"""
import ctypes as C

import numpy as np

from . import _abi
from .gstate import GSTATE


class PmdInverse:
    """A plx_pmdinv plan: batched use (one waveplate draw per frame) for Monte-Carlo PMD studies."""

    def __init__(self, nfft, max_frames=1):
        self.lib = _abi.get()
        self.plan = C.c_void_p()
        self.lib.call("plx_pmdinv_create", C.byref(self.plan), int(nfft), int(max_frames))
        self.nfft = int(nfft)

    def set_link(self, brf, options=None, nsets=1):
        """brf: list of fibre structs; for nsets > 1 each db0/theta/epsilon is [nsets, ntrunk]."""
        nt = np.array([np.asarray(b["theta"]).reshape(nsets, -1).shape[1] for b in brf], dtype=np.int32)

        def cat(name):
            return np.ascontiguousarray(np.concatenate([np.asarray(b[name], dtype=float).reshape(nsets, -1) for b in brf], 1))

        db0, th, ep = cat("db0"), cat("theta"), cat("epsilon")
        lcorr = np.array([float(b["lcorr"]) for b in brf])
        betat = np.ascontiguousarray(np.stack([np.asarray(b["betat"], dtype=float).reshape(-1) for b in brf]))
        db1 = np.ascontiguousarray(np.stack([np.asarray(b["db1"], dtype=float).reshape(-1) for b in brf]))
        if betat.shape[1] != self.nfft:
            raise ValueError("brf.betat must have one column of length(GSTATE.FN) entries")
        mat = None
        if options and "mat" in options:
            m = np.asarray(options["mat"], dtype=complex).reshape(2, 2)
            mat = np.ascontiguousarray(np.stack([m.real, m.imag], -1).reshape(-1))      # row-major (re, im)
        gvd = not (options and options.get("gvd") == "no")
        self.lib.call("plx_pmdinv_set_link", self.plan, len(brf), nt.ctypes.data, db0.ctypes.data, th.ctypes.data,
                      ep.ctypes.data, lcorr.ctypes.data, betat.ctypes.data, db1.ctypes.data,
                      mat.ctypes.data if mat is not None else None, int(gvd), int(nsets))

    def apply(self, ux, uy):
        import torch
        self.lib.call("plx_pmdinv_apply_dev", self.plan, ux.data_ptr(), uy.data_ptr(), ux.shape[0],
                      torch.cuda.current_stream().cuda_stream)

    def matrices(self, frame=0):
        U = np.zeros((self.nfft, 2, 2), dtype=complex)       # memory order of MATLAB's [2][2][Nfft] column-major
        Ui = np.zeros_like(U)
        self.lib.call("plx_pmdinv_matrices", self.plan, frame, U.ctypes.data, Ui.ctypes.data)
        # memory [k][c][r] -> index [r, c, k]
        return np.transpose(Ui, (2, 1, 0)).copy(), np.transpose(U, (2, 1, 0)).copy()

    def close(self):
        if self.plan:
            self.lib.call("plx_pmdinv_destroy", self.plan)
            self.plan = None


def inverse_pmd(brf, options=None, nargout=0):
    """inverse_pmd(brf[, options]); nargout 1 -> Uinv, 2 -> (Uinv, U), like the reference's varargout."""
    if GSTATE.FIELDX is None or GSTATE.FIELDX.shape[0] > 1:
        raise ValueError("inverse_pmd can be used only with a unique field.")          # :88
    if isinstance(brf, dict):
        brf = [brf]
    nfft = GSTATE.FIELDX.shape[1]
    p = PmdInverse(nfft, 1)
    try:
        p.set_link(brf, options)
        if (options is None) or ("apply" not in options) or options["apply"] == "n":   # :138 (as written)
            if GSTATE.FIELDY is None:
                raise ValueError("inverse_pmd needs GSTATE.FIELDY")
            p.apply(GSTATE.FIELDX, GSTATE.FIELDY)
            GSTATE.DISP = np.zeros((2, GSTATE.NCH))                                     # :145
        if nargout >= 1:
            Uinv, U = p.matrices(0)
            return Uinv if nargout == 1 else (Uinv, U)
    finally:
        p.close()
    return None

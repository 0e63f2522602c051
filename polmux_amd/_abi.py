"""ctypes binding of the C ABI declared in include/polmux_hip.h.

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc,
--offload-arch=gfx950) as ``polmux_amd/lib/libpolmux_hip.so``.  There is no
fallback of any kind: if the library is missing or a call fails, a
``PolmuxError`` is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpolmux_hip.so")

PLX_OK = 0
PLX_ERR_HIP = -1
PLX_ERR_ARG = -2
PLX_ERR_UNSUPPORTED = -3
PLX_ERR_REFERENCE = -4
PLX_ERR_TIMEOUT = -5
PLX_SSFM_SHARE_DEVICE = 1


class PolmuxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class SsfmDesc(C.Structure):
    _fields_ = [("nfft", C.c_int64), ("nfc", C.c_int32), ("dual_pol", C.c_int32), ("max_frames", C.c_int32),
                ("fls", C.c_int32 * 4), ("dzmaxt", C.c_double), ("dphimaxt", C.c_double),
                ("alphalin", C.c_double), ("length", C.c_double), ("nplates", C.c_int32),
                ("manakov", C.c_int32), ("gam", C.c_void_p), ("betat", C.c_void_p), ("db1", C.c_void_p)]


class SsfmTuning(C.Structure):
    """plx_ssfm_tuning (include/polmux_hip.h): fill with plx_ssfm_tuning_defaults first (Binding.tuning does)."""
    _fields_ = [("size", C.c_uint32), ("no_fuse", C.c_int32), ("short_rows", C.c_int32), ("no_row_split", C.c_int32),
                ("p1", C.c_int32), ("logW", C.c_int32), ("col_threads", C.c_int32), ("rowr", C.c_int32), ("rowsm", C.c_int32),
                ("row256_split", C.c_int32), ("row4k_split", C.c_int32), ("rowg_split", C.c_int32), ("no_pmd_tab", C.c_int32),
                ("store_late", C.c_int32), ("row_rev", C.c_int32), ("safe_landing", C.c_int32), ("reserved_", C.c_int32 * 4),
                ("barrier_timeout_ms", C.c_double)]


class DspParams(C.Structure):
    _fields_ = [("workatbaudrate", C.c_int32), ("applynlr", C.c_int32), ("nlralpha", C.c_double),
                ("power_mw", C.c_double), ("applypol", C.c_int32), ("polmethod", C.c_int32),
                ("cma_R", C.c_double * 2), ("cma_mu", C.c_double), ("cma_taps", C.c_int32),
                ("cma_txpolars", C.c_int32), ("cma_phizero", C.c_double), ("easi_mu", C.c_double),
                ("easi_txpolars", C.c_int32), ("easi_phizero", C.c_double), ("modorder", C.c_int32),
                ("freqavg", C.c_int32), ("phasavg", C.c_int32), ("poworder", C.c_int32),
                ("cma_has_mat", C.c_int32), ("easi_has_mat", C.c_int32), ("cma_mat", C.c_double * 8),
                ("easi_mat", C.c_double * 8), ("mfile_twins", C.c_int32), ("reserved_", C.c_int32)]


class FrontDesc(C.Structure):
    _fields_ = [("nfft", C.c_int64), ("dual_pol", C.c_int32), ("max_frames", C.c_int32), ("balanced", C.c_int32),
                ("adcbits", C.c_int32), ("decim", C.c_int32), ("ntaps", C.c_int32), ("fir", C.c_void_p),
                ("hopt_re", C.c_void_p), ("hopt_im", C.c_void_p), ("hel_re", C.c_void_p), ("hel_im", C.c_void_p),
                ("elo_re", C.c_void_p), ("elo_im", C.c_void_p), ("elo_scalar", C.c_double)]


_vp, _i32, _i64, _dbl, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_size_t

# name -> argtypes; every symbol include/polmux_hip.h declares (tests check the list
# against the header and against the built library)
SIGNATURES = {
    "plx_abi_version": [],
    "plx_device_count": [C.POINTER(C.c_int)],
    "plx_set_device": [C.c_int],
    "plx_release_all": [],
    "plx_gateway_stats": [_vp],
    "plx_gateway_stats_ex": [_vp, C.c_int],
    "plx_fastexp": [_vp, _vp, _vp, _sz],
    "plx_fastexp_dev": [_vp, _vp, _sz, _vp],
    "plx_ssfm_create": [C.POINTER(_vp), C.POINTER(SsfmDesc)],
    "plx_ssfm_create_ex": [C.POINTER(_vp), C.POINTER(SsfmDesc), C.c_uint32],
    "plx_ssfm_tuning_defaults": [C.POINTER(SsfmTuning)],
    "plx_ssfm_tuning_override": [C.POINTER(SsfmTuning)],
    "plx_ssfm_create_tuned": [C.POINTER(_vp), C.POINTER(SsfmDesc), C.c_uint32, C.POINTER(SsfmTuning)],
    "plx_ssfm_destroy": [_vp],
    "plx_ssfm_set_step_sequence": [_vp, _vp, C.c_int],
    "plx_ssfm_log_steps": [_vp, C.c_int],
    "plx_ssfm_step_sequence": [_vp, C.c_int, _vp, C.c_int],
    "plx_ssfm_set_birefringence": [_vp, _vp, _vp, _vp, C.c_int],
    "plx_ssfm_set_birefringence_dev": [_vp, _vp, _vp, _vp, C.c_int, _vp],
    "plx_ssfm_propagate_dev": [_vp, _vp, _vp, C.c_int, _vp],
    "plx_ssfm_results": [_vp, C.c_int, _vp, _vp],
    "plx_ssfm_stats": [_vp, C.POINTER(_i64), C.POINTER(_i64)],
    "plx_ssfm_info": [_vp, _vp],
    "plx_ssfm_utilisation": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "plx_ssfm_barrier_timeouts": [_vp, C.POINTER(_i32), C.c_int],
    "plx_ssfm_profile": [_vp, C.c_int],
    "plx_ssfm_kernel_times": [_vp, _vp, _vp],
    "plx_matrix_ssfm": [_vp, _vp, _vp, _vp, C.POINTER(SsfmDesc), _vp, _vp, _vp, C.POINTER(_dbl),
                        C.POINTER(_i32)],
    "plx_scalar_ssfm": [_vp, _vp, C.POINTER(SsfmDesc), C.POINTER(_dbl), C.POINTER(_i32)],
    "plx_scalar_ssfm_adaptive": [_vp, _vp, C.POINTER(SsfmDesc), C.c_int, _dbl, _dbl, C.POINTER(_dbl), C.POINTER(_i32),
                                 C.POINTER(_i32)],
    "plx_cde_create": [C.POINTER(_vp), _i64, _i64, _vp],
    "plx_cde_destroy": [_vp],
    "plx_cde_apply_dev": [_vp, _vp, _vp, _i64, C.c_int, _vp],
    "plx_cde_ofde": [_vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _dbl, _dbl, _dbl, _i64, _i64, _vp, _vp, _vp, _vp],
    "plx_cmaadaptivefilter": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _dbl, _dbl, _vp, _dbl, _vp, _vp],
    "plx_easiadaptivefilter": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _vp, _vp],
    "plx_cmaadaptivefilter_m": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _dbl, _vp, _vp, _vp],
    "plx_easiadaptivefilter_m": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _dbl, _vp, _vp],
    "plx_cmapolardemux": [_vp, _vp, _i64, _i32, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "plx_easipolardemux": [_vp, _vp, _i64, _dbl, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "plx_poldemux_dev": [C.c_int, _vp, _vp, _i64, C.c_int, _i32, _dbl, _vp, _vp, _vp, _vp, _vp],
    "plx_dsp_create": [C.POINTER(_vp), _i64, _i32, _i32, C.POINTER(DspParams)],
    "plx_dsp_destroy": [_vp],
    "plx_dsp_run_dev": [_vp, _vp, _vp, C.c_int, _vp],
    "plx_dsp_out_len": [_vp],
    "plx_decide_count_dev": [_vp, _i64, _i32, C.c_int, _vp, _vp, _vp, _vp],
    "plx_decide_count_frames_dev": [_vp, _i64, _i32, C.c_int, _vp, _i64, _vp, _vp, _vp],
    "plx_evm_dev": [_vp, _i64, _i32, C.c_int, _vp, _vp],
    "plx_ampliflat_dev": [_vp, _vp, _i64, _i32, C.c_int, _dbl, _vp, _vp, C.c_uint64, _vp, _i32, _i32, _vp],
    "plx_front_create": [C.POINTER(_vp), C.POINTER(FrontDesc)],
    "plx_front_destroy": [_vp],
    "plx_front_out_len": [_vp],
    "plx_front_run_dev": [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp],
    "plx_rx_front": [_vp, _vp, _vp, _vp, C.POINTER(FrontDesc), _vp, _vp, _vp, _vp, _vp],
    "plx_filter_create": [C.POINTER(_vp), _i64, C.c_int, _vp, _vp],
    "plx_filter_destroy": [_vp],
    "plx_filter_apply_dev": [_vp, _vp, C.c_int, _vp],
    "plx_pmdinv_create": [C.POINTER(_vp), _i64, C.c_int],
    "plx_pmdinv_destroy": [_vp],
    "plx_pmdinv_set_link": [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int],
    "plx_pmdinv_apply_dev": [_vp, _vp, _vp, C.c_int, _vp],
    "plx_pmdinv_matrices": [_vp, C.c_int, _vp, _vp],
    "plx_pick_dev": [_vp, _vp, _i64, _i64, _i64, _i64, _dbl, C.c_int, _i64, _vp],
}
_RESTYPES = {"plx_dsp_out_len": _i64, "plx_front_out_len": _i64}


class Binding:
    """One loaded copy of the library with checked calls."""

    def __init__(self, path=None):
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise PolmuxError(PLX_ERR_HIP,
                              "libpolmux_hip.so not found at %s: build it with "
                              "`python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        self.path = path
        if os.path.basename(path) == "libpolmux_hip.so":
            # torch owns the HBM buffers and the streams handed to the library: load its HIP
            # runtime FIRST so both sides share one runtime (torch bundles its own libamdhip64;
            # loading ours first would bind torch to a second, different copy).
            import torch  # noqa: F401
        self.lib = C.CDLL(path)
        _BINDINGS.append(self)
        self.lib.plx_last_error.restype = C.c_char_p
        for name, args in SIGNATURES.items():
            fn = getattr(self.lib, name)          # AttributeError if a symbol is missing
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)

    def tuning(self, **fields):
        """A plx_ssfm_tuning holding the defaults with `fields` changed (for plx_ssfm_create_tuned / plx_ssfm_tuning_override)."""
        t = SsfmTuning()
        self.call("plx_ssfm_tuning_defaults", C.byref(t))
        for k, v in fields.items():
            if k not in dict(SsfmTuning._fields_) or k in ("size", "reserved_"):
                raise KeyError("plx_ssfm_tuning has no field %r" % k)
            setattr(t, k, v)
        return t

    def last_error(self):
        return self.lib.plx_last_error().decode("utf-8", "replace")

    def call(self, name, *args):
        rc = getattr(self.lib, name)(*args)
        if rc != PLX_OK:
            raise PolmuxError(rc, self.last_error() or ("%s failed with code %d" % (name, rc)))
        return rc


_default = None
_BINDINGS = []


def bindings():
    """Every loaded copy of the library in this process (the tests' tuning override addresses them all)."""
    return list(_BINDINGS)


def get():
    """The process-wide binding of the real (hipcc-built) library."""
    global _default
    if _default is None:
        _default = Binding()
    return _default

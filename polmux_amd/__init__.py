"""polmux_amd -- MI355X-native hot path of the Optilux/Polmux optical-link simulator.

Host-side mirror of the reference's function surface for the accelerated path
(fiber -> CDE_OFDE -> DspPdmCohQpsk/cma/easi -> ber_estimate); the compute runs in
hand-written gfx950 kernels behind the C ABI of include/polmux_hip.h
(polmux_amd/lib/libpolmux_hip.so).  There is no CPU fallback.
"""
from ._abi import PolmuxError  # noqa: F401
from .gstate import CONSTANTS, GSTATE, create_field, lasersource, reset_all  # noqa: F401
from .fiber import fiber  # noqa: F401
from .ampliflat import ampliflat  # noqa: F401
from .rx import (CDE_OFDE, DspPdmCohQpsk, cmaadaptivefilter, easiadaptivefilter, fastexp, samp2pat)  # noqa: F401
from .rxfront import (RxPdmCohQpsk, corrdelay, dsp4cohdec, evaldelay, eye_opening, myfilter,  # noqa: F401
                      mygeteyeinfo, receiver_cohmix)
from .pmdinv import inverse_pmd  # noqa: F401
from .mc import ber_estimate, mc_estimate  # noqa: F401

__all__ = ["PolmuxError", "GSTATE", "CONSTANTS", "reset_all", "create_field", "lasersource", "fiber", "ampliflat", "CDE_OFDE", "receiver_cohmix", "RxPdmCohQpsk", "dsp4cohdec", "myfilter", "evaldelay",
           "DspPdmCohQpsk", "cmaadaptivefilter", "easiadaptivefilter", "fastexp", "samp2pat", "ber_estimate",
           "mc_estimate", "inverse_pmd"]

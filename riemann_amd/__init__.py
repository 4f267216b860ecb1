"""riemann_amd -- many-chain Metropolis-Hastings on AMD MI355X behind the
plugin API of rscalzo/riemann (``Sampler`` / ``Proposal`` / ``Model``).

Top-level names as /root/reference/riemann/__init__.py:10-13.  The host side is
pure Python; the hot path runs in hand-written HIP kernels (gfx950) reached
through a ctypes C ABI (include/riemann_hip.h).  No CPU fallback exists.
"""
from .sampling_errors import ParameterError, RiemannBaseError
from .models.model import Model
from .proposals.proposal import Proposal
from .samplers.sampler import Sampler

__all__ = ["Model", "Sampler", "Proposal", "ParameterError", "RiemannBaseError"]

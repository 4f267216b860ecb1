from .sampler import Sampler
from .ptsampler import PTSampler, TemperedModel

__all__ = ["Sampler", "PTSampler", "TemperedModel"]

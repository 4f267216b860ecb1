from .model import Model, DeviceModel
from .gaussian import MultiGaussianDist
from .logistic import LogisticRegression
from .changepoint import ChangepointParams, ChangepointRegression1D
from .benchmarks import MixtureDist

__all__ = ["Model", "DeviceModel", "MultiGaussianDist", "LogisticRegression",
           "ChangepointParams", "ChangepointRegression1D", "MixtureDist"]

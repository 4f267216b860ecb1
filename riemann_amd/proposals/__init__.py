from .proposal import Proposal, DeviceProposal
from .adaptive import AdaptScaleProposal
from .randomwalk import MetropolisRandomWalk, AdaptScaleRandomWalk, pCN
from .hamiltonian import VanillaHMC, MALA, AdaptScaleHMC
from .changepoint import ChangepointRegression1DProp

__all__ = ["Proposal", "DeviceProposal", "MetropolisRandomWalk", "VanillaHMC", "MALA",
           "AdaptScaleProposal", "AdaptScaleRandomWalk", "AdaptScaleHMC", "pCN",
           "ChangepointRegression1DProp"]

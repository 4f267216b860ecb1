from .proposal import Proposal, DeviceProposal
from .adaptive import AdaptScaleProposal
from .randomwalk import MetropolisRandomWalk, AdaptScaleRandomWalk, pCN
from .hamiltonian import VanillaHMC, MALA, AdaptScaleHMC

__all__ = ["Proposal", "DeviceProposal", "MetropolisRandomWalk", "VanillaHMC", "MALA",
           "AdaptScaleProposal", "AdaptScaleRandomWalk", "AdaptScaleHMC", "pCN"]

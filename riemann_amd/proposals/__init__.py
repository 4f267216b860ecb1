from .proposal import Proposal, DeviceProposal
from .adaptive import AdaptScaleProposal, AdaptCovProposal
from .randomwalk import (MetropolisRandomWalk, AdaptScaleRandomWalk, AdaptCovRandomWalk,
                         AdaptScaleCovRandomWalk, AdaptiveMetropolisRandomWalk, HaarioRandomWalk, pCN, AdaptScalepCN)
from .hamiltonian import VanillaHMC, MALA, AdaptScaleHMC, AdaptCovHMC, AdaptScaleCovHMC
from .changepoint import ChangepointRegression1DProp

__all__ = ["Proposal", "DeviceProposal", "MetropolisRandomWalk", "VanillaHMC", "MALA",
           "AdaptScaleProposal", "AdaptScaleRandomWalk", "AdaptScaleHMC", "pCN",
           "ChangepointRegression1DProp", "AdaptCovProposal", "AdaptCovRandomWalk", "AdaptScaleCovRandomWalk",
           "AdaptiveMetropolisRandomWalk", "HaarioRandomWalk", "AdaptCovHMC", "AdaptScaleCovHMC", "AdaptScalepCN"]

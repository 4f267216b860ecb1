from .proposal import Proposal, DeviceProposal
from .randomwalk import MetropolisRandomWalk
from .hamiltonian import VanillaHMC, MALA

__all__ = ["Proposal", "DeviceProposal", "MetropolisRandomWalk", "VanillaHMC", "MALA"]

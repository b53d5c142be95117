"""Abstract policy-network API kept from the reference (networks/abstracts.py:6-24): a strategy only
ever talks to a network through these four methods plus __call__(obs[1,S]) -> action."""
from abc import abstractmethod

from torch import nn


class BaseNetwork(nn.Module):
    """Subclass contract: `zero_init`, `reset`, `get_param_list`, `apply_param`."""

    def __init__(self):
        super().__init__()

    @abstractmethod
    def zero_init(self):
        """set every parameter to 0 (ESLoop does this once before the first population)"""

    @abstractmethod
    def reset(self):
        """clear recurrent state at the start of an episode"""

    @abstractmethod
    def get_param_list(self):
        """list of numpy arrays, one per parameter tensor, in parameters() order"""

    @abstractmethod
    def apply_param(self, param_lst: list):
        """load a list shaped like get_param_list()"""

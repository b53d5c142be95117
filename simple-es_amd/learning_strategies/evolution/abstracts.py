"""Abstract bases kept from the reference (learning_strategies/evolution/abstracts.py:5-34) so user
code that subclasses them keeps importing."""
from abc import ABCMeta, abstractmethod


class BaseESLoop(metaclass=ABCMeta):
    @abstractmethod
    def __init__(self):
        ...

    @abstractmethod
    def run(self):
        ...


class BaseOffspringStrategy(metaclass=ABCMeta):
    @abstractmethod
    def __init__(self):
        ...

    @abstractmethod
    def _gen_offsprings(self):
        ...

    @abstractmethod
    def get_elite_model(self):
        ...

    @abstractmethod
    def init_offspring(self):
        ...

    @abstractmethod
    def evaluate(self):
        ...

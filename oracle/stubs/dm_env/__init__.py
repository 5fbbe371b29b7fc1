"""Minimal stand-in for dm_env (absent from the image): only the surface the
reference's step loop touches.  Test infrastructure -- used only by
oracle/gen_golden.py in the development container, never shipped to the GPU box
as part of the product path."""
import abc
import enum
from typing import Any, NamedTuple

from . import specs  # noqa: F401


class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2

    def first(self):
        return self is StepType.FIRST

    def mid(self):
        return self is StepType.MID

    def last(self):
        return self is StepType.LAST


class TimeStep(NamedTuple):
    step_type: Any
    reward: Any
    discount: Any
    observation: Any

    def first(self):
        return self.step_type == StepType.FIRST

    def mid(self):
        return self.step_type == StepType.MID

    def last(self):
        return self.step_type == StepType.LAST


def restart(observation):
    return TimeStep(StepType.FIRST, None, None, observation)


def transition(reward, observation, discount=1.0):
    return TimeStep(StepType.MID, reward, discount, observation)


def termination(reward, observation):
    return TimeStep(StepType.LAST, reward, 0.0, observation)


def truncation(reward, observation, discount=1.0):
    return TimeStep(StepType.LAST, reward, discount, observation)


class Environment(abc.ABC):
    @abc.abstractmethod
    def reset(self):
        ...

    @abc.abstractmethod
    def step(self, action):
        ...

    def reward_spec(self):
        return specs.Array(shape=(), dtype=float, name="reward")

    def discount_spec(self):
        return specs.BoundedArray(shape=(), dtype=float, minimum=0.0, maximum=1.0, name="discount")

    def close(self):
        pass

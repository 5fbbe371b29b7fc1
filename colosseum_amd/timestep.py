"""dm_env's TimeStep vocabulary (the reference's `step`/`reset` return type, colosseum/mdp/base.py:1268-1317).
The real `dm_env` is used when importable, otherwise field-compatible local definitions."""
try:  # pragma: no cover - depends on the environment
    from dm_env import StepType, TimeStep, restart, termination, transition  # noqa: F401
    from dm_env.specs import Array, BoundedArray, DiscreteArray  # noqa: F401
except Exception:  # dm_env is not part of this image
    import enum
    from typing import Any, NamedTuple

    import numpy as np

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

    class Array:
        def __init__(self, shape, dtype, name=None):
            self.shape, self.dtype, self.name = tuple(shape), np.dtype(dtype), name

        def generate_value(self):
            return np.zeros(self.shape, self.dtype)

    class BoundedArray(Array):
        def __init__(self, shape, dtype, minimum, maximum, name=None):
            super().__init__(shape, dtype, name)
            self.minimum, self.maximum = np.asarray(minimum), np.asarray(maximum)

    class DiscreteArray(BoundedArray):
        def __init__(self, num_values, dtype=np.int32, name=None):
            super().__init__((), dtype, 0, num_values - 1, name)
            self.num_values = num_values

import numpy as np


class Array:
    def __init__(self, shape, dtype, name=None):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.name = name

    def generate_value(self):
        return np.zeros(self.shape, self.dtype)

    def validate(self, value):
        return np.asarray(value)


class BoundedArray(Array):
    def __init__(self, shape, dtype, minimum, maximum, name=None):
        super().__init__(shape, dtype, name)
        self.minimum = np.asarray(minimum)
        self.maximum = np.asarray(maximum)


class DiscreteArray(BoundedArray):
    def __init__(self, num_values, dtype=np.int32, name=None):
        super().__init__((), dtype, 0, num_values - 1, name)
        self.num_values = num_values

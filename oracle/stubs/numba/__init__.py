"""Identity stand-in for numba (absent): njit returns the function unchanged,
so the reference's jitted loops run as plain numpy/Python."""
from . import core, typed, types  # noqa: F401

bool_ = bool


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(f):
        return f

    return deco


jit = njit

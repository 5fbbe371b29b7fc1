"""No-op stand-in for gin-config (absent)."""
import contextlib


def configurable(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(f):
        return f

    return deco


def constants_from_enum(cls=None, **kwargs):
    if cls is None:
        return lambda c: c
    return cls


@contextlib.contextmanager
def config_scope(*a, **k):
    yield


def parse_config_files_and_bindings(*a, **k):
    pass


def parse_config_file(*a, **k):
    pass


def parse_config(*a, **k):
    pass


def clear_config(*a, **k):
    pass


def bind_parameter(*a, **k):
    pass

from . import tune, util  # noqa: F401

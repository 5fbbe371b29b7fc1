from . import multiprocessing  # noqa: F401

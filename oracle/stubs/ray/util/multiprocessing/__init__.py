from multiprocessing import Pool  # noqa: F401

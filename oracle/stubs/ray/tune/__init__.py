"""No-op stand-in for ray.tune (absent): only the names the reference's agent classes mention in annotations and
hyper-parameter search-space helpers."""


class _S:
    def __init__(self, *a, **k):
        pass


def uniform(*a, **k):
    return _S()


choice = randint = loguniform = quniform = uniform


class _Sample:
    Domain = _S
    Float = _S
    Integer = _S
    Categorical = _S


class _Search:
    sample = _Sample


search = _Search
sample = _Sample

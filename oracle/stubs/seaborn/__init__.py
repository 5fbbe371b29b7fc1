"""No-op stand-in for seaborn (absent); plotting is out of scope."""


def set_theme(*a, **k):
    pass


def __getattr__(name):
    def _noop(*a, **k):
        return None

    return _noop

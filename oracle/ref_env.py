"""Bootstrap for running the reference (read-only at /root/reference) in the
development container: puts the stub modules and the reference on sys.path and
installs the two py3.10 / numpy-2 aliases the reference's imports need
(colosseum/mdp/utils/mdp_creation.py:8, colosseum/utils/miscellanea.py:34).

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py to emit the fixtures in
tests/golden/.  Nothing under colosseum_amd/ imports this, and it is never run
on the GPU box (/root/reference does not exist there)."""
import collections
import collections.abc
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("COLOSSEUM_REFERENCE", "/root/reference")


def install():
    if not os.path.isdir(os.path.join(REFERENCE, "colosseum")):
        raise RuntimeError(f"reference tree not found at {REFERENCE}")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    for p in (os.path.join(HERE, "stubs"), REFERENCE):
        if p not in sys.path:
            sys.path.insert(0, p)
    if not hasattr(collections, "Container"):
        collections.Container = collections.abc.Container
    import numpy as np

    try:
        import numpy.core._exceptions  # noqa: F401
    except Exception:
        exc = types.ModuleType("numpy.core._exceptions")
        exc._ArrayMemoryError = MemoryError
        sys.modules["numpy.core._exceptions"] = exc
        try:
            import numpy.core as npcore

            npcore._exceptions = exc
        except Exception:
            pass
    # bypass colosseum.agent.agents.* package __init__ (imports TensorFlow / sonnet / bsuite)
    return np

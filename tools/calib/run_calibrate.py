import ctypes as C, sys
sys.path.insert(0, '.')
from colosseum_amd import _lib as L
lib = L.load()
for what in (0, 1):
    for n in (20000, 200000):
        out = C.c_double()
        L.check(lib.cmdp_calibrate(what, n, C.byref(out)))
        print("calib", what, n, out.value, flush=True)

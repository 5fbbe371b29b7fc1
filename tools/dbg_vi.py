"""Times the Jacobi sweep kernels on config C3 (4096 x FrozenLake 20x20, gamma .99, eps 1e-6) and on policy evaluation,
checking that every kernel returns the same bits.  python tools/dbg_vi.py [kernel ids ...]   (default 5 7)"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from colosseum_amd import _lib as L  # noqa: E402
from colosseum_amd.batched import BatchedMDP  # noqa: E402
from colosseum_amd.mdp.fast_batch import frozenlake_dp_tables  # noqa: E402

kernels = [int(a) for a in sys.argv[1:]] or [5, 7]
lib = L.load()
out = {}
tables = {size: frozenlake_dp_tables(np.arange(n), 20, 16) for size, n in ((20, 4096), (21, 16384))}  # before HIP starts (fork)
for size, fl in tables.items():
    dp = BatchedMDP(tables=fl, with_env=False)
    ref = None
    for k in kernels:
        dp.set_option(L.OPT_DP_KERNEL, k)
        ms = []
        for rep in range(4):
            Q, V, sw = dp.value_iteration(0.99, 1e-6)
            kms = C.c_double()
            L.check(lib.cmdp_stat(dp.handle, L.STAT_DP_KERNEL_MS, C.byref(kms)))
            ms.append(kms.value)
        if ref is None:
            ref = (Q.copy(), V.copy(), sw.copy())
        same = bool((Q == ref[0]).all() and (V == ref[1]).all() and (sw == ref[2]).all())
        out["size%d_k%d" % (size, k)] = dict(kernel_ms=min(ms), sweeps=int(sw.sum()), sweeps_per_s=float(sw.sum()) / (min(ms) * 1e-3), bit_equal=same,
                                             max_S=int(np.max(np.diff(fl["state_off"]))), n=len(sw), sw_min=int(sw.min()), sw_mean=float(sw.mean()), sw_max=int(sw.max()),
                                             sw_p90=float(np.percentile(sw, 90)))
        print(size, k, out["size%d_k%d" % (size, k)], flush=True)
    dp.close()
json.dump(out, sys.stdout, indent=1)

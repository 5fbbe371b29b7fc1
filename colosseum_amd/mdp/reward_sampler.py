"""Reference-exact sampling of stochastic (Beta) rewards on the host.

`BaseMDP.sample_reward` (colosseum/mdp/base.py:1187-1207) fills, per visited (s, a, s') triple and in first-visit
order, a cache of 5000 samples drawn from the MDP's single numpy stream (`scipy.stats.beta(a, b).rvs(5000,
random_state=self._rng)` == `RandomState.beta(a, b, 5000)`, the legacy gamma-ratio / Johnk sampler built on libm
`log`/`exp`/`pow`/`sqrt`), pops the samples FIFO, refills on exhaustion, and rescales `r * (max - min) - min`.
The draw order depends on the trajectory, so the procedure is sequential by construction; it runs on the host (the
very numpy generator the reference uses, positioned by the builder) next to the device-side transition sampling.
Deterministic distributions draw nothing (colosseum/utils/miscellanea.py:259-270)."""
from typing import Dict, List, Tuple

import numpy as np

from .builder import REWARD_BETA, TabularModel

CACHE = 5000


class CompatRewardSampler:
    def __init__(self, model: TabularModel):
        self._m = model
        self._rng: np.random.RandomState = model.extra["rng"]
        self._cache: Dict[Tuple[int, int, int], List[float]] = {}
        self._pos: Dict[Tuple[int, int, int], int] = {}
        self._lo, self._hi = model.rewards_range

    def _dist(self, s: int, a: int, s_next: int):
        m = self._m
        r = s * m.n_actions + a
        lo, hi = int(m.sp_ptr[r]), int(m.sp_ptr[r + 1])
        for e in range(lo, hi):
            if m.sp_next[e] == s_next:
                return int(m.sp_rkind[e]), float(m.sp_rp0[e]), float(m.sp_rp1[e])
        raise KeyError(f"({s}, {a}) has no successor {s_next}")

    def sample(self, s: int, a: int, s_next: int) -> float:
        key = (s, a, s_next)
        buf = self._cache.get(key)
        pos = self._pos.get(key, 0)
        if buf is None or pos == len(buf):
            kind, p0, p1 = self._dist(s, a, s_next)
            if kind == REWARD_BETA:
                buf = self._rng.beta(p0, p1, CACHE)
            else:
                buf = np.full(CACHE, p0)
            self._cache[key] = buf
            pos = 0
        self._pos[key] = pos + 1
        r = float(buf[pos])
        return r * (self._hi - self._lo) - self._lo

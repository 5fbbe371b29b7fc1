"""Emission maps and observation noise (reference colosseum/emission_maps/*.py,
colosseum/noises/*.py).  An emission map is a table of float32 feature vectors, one per state (and per in-episode
time for episodic MDPs: `EmissionMap.all_observations`, emission_maps/base.py:56-83); an observation is a row of that
table plus, optionally, additive noise.

Built: `Tabular` (the state index itself, no table), `StateInfo` (emission_maps/state_info.py:20-30: the node's fields,
preceded by the in-episode time when episodic), `OneHotEncoding` (emission_maps/one_hot_encoding.py:19-27) and the
`GaussianUncorrelated` noise (noises/gaussian_uncorrelated.py).  Not built: `TensorEncoding` / `ImageEncoding` (they need
every family's grid drawing) and `StateLinear*` (their features come from the unseeded global `np.random`, so there is
nothing to be equal to).

Two ways to get observations, like rewards: `GpuMDP(..., emission_map="StateInfo", noise="GaussianUncorrelated")`
reproduces the reference's observation stream exactly (the noise is numpy's own `RandomState(seed).normal`, cached
5000 samples at a time, noises/base.py:51-56, continued on the host); `BatchedMDP.set_observation_table` +
`BatchedMDP.observe` gather the rows of all instances on the device and add Philox normal noise there (kernel
`k_emit`, throughput mode)."""
from typing import Optional, Tuple

import numpy as np

NOISE_CACHE = 5000  # config.get_size_cache_noise()


def observation_table(model, name: str) -> Optional[np.ndarray]:
    """float32 [S, F] (continuous) or [H, S, F] (episodic) -- `EmissionMap.all_observations`; None for Tabular."""
    S, H = model.n_states, model.H
    if name in (None, "Tabular"):
        return None
    if name == "OneHotEncoding":
        base = np.eye(S, dtype=np.float32)
        return np.broadcast_to(base, (H, S, S)).copy() if H else base
    if name == "StateInfo":
        nodes = np.asarray(model.nodes, np.float32).reshape(S, -1)
        if not H:
            return nodes
        out = np.empty((H, S, 1 + nodes.shape[1]), np.float32)
        out[:, :, 0] = np.arange(H, dtype=np.float32)[:, None]
        out[:, :, 1:] = nodes[None]
        return out
    raise NotImplementedError(f"emission map {name!r} is not built (Tabular, StateInfo, OneHotEncoding are)")


class CompatNoise:
    """`GaussianUncorrelated.__next__` draw for draw: RandomState(seed).normal(0, scale, (5000, *shape)) as float32,
    refilled when the cache is empty."""

    def __init__(self, seed: int, shape: Tuple[int, ...], scale: float = 0.1):
        self._rng = np.random.RandomState(seed)
        self.shape, self.scale = tuple(shape), scale
        self._cache = []

    def __next__(self) -> np.ndarray:
        if not self._cache:
            self._cache = list(self._rng.normal(loc=0, scale=self.scale, size=(NOISE_CACHE, *self.shape)).astype(np.float32))
        return self._cache.pop(0)

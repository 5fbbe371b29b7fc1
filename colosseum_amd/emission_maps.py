"""Emission maps and observation noise (reference colosseum/emission_maps/*.py,
colosseum/noises/*.py).  An emission map is a table of float32 observations, one per state (and per in-episode time for
episodic MDPs: `EmissionMap.all_observations`, emission_maps/base.py:56-83); an observation is an entry of that table
plus, optionally, additive noise.

All eight maps of the reference are built:
  Tabular            the state index itself, no table
  StateInfo          emission_maps/state_info.py:20-30: the node's fields, preceded by the in-episode time when episodic
  OneHotEncoding     emission_maps/one_hot_encoding.py:19-27
  TensorEncoding     emission_maps/tensor_encoding.py:34-52: one-hot planes of the family's ASCII drawing (+ a time plane)
  ImageEncoding      emission_maps/image_encoding.py:34-49: the drawing as symbol indices (+ a time row)
  StateLinearOptimal / StateLinearRandom   emission_maps/base.py:144-228: features in which V* / V of the uniform policy is
                     linear, drawn from the GLOBAL numpy stream exactly as `_sample_linear_value_features` does (seed
                     `np.random` to reproduce a reference run)
and the four noise classes (`GaussianUncorrelated`, `GaussianCorrelated`, `StudentTUncorrelated`, `StudentTCorrelated`)
with the reference's sampling calls, quirks included (noises/*.py; see the classes below).

The family drawings (`_get_grid_representation` of every family) are restated in `grid_of`.  Quirk kept: the two
MiniGrid families draw the agent of `self.cur_node`, not of the node asked for (minigrid_empty/base.py:252-264,
minigrid_rooms/base.py:323-331), so every entry of their Tensor/Image tables shows the state the MDP was in when the
table was first needed (the first `reset()`); `observation_table(..., cur_state=...)` takes that state.

Two ways to get observations, like rewards: `GpuMDP(..., emission_map="StateInfo", noise="GaussianUncorrelated")`
reproduces the reference's observation stream exactly (noise caches of 5000 samples from `RandomState(seed)`,
noises/base.py:51-56, continued on the host); `BatchedMDP.set_observation_table` + `BatchedMDP.observe` gather the rows
of all instances on the device and add Philox noise there (kernel `k_emit`, throughput mode: Gaussian / Student-t,
uncorrelated or through a Cholesky factor)."""
from typing import Optional, Tuple

import numpy as np

NOISE_CACHE = 5000  # config.get_size_cache_noise()
MIN_LINEAR_FEATURE_DIM = 10  # config.get_min_linear_feature_dim()

SYMBOLS = {  # `get_unique_symbols()` of every family, in the reference's order (the order IS the encoding)
    "DeepSea": ["A", " "],
    "FrozenLake": ["A", "F", "H", "G"],
    "MiniGridEmpty": [" ", ">", "<", "v", "^", "G"],
    "MiniGridRooms": [" ", ">", "<", "v", "^", "G", "W"],
    "RiverSwim": [" ", "A", "S", "G"],
    "SimpleGrid": [" ", "A", "+", "-"],
    "Taxi": [" ", "A", "X", "D", "P"],
}
_ARROWS = {0: "^", 1: ">", 2: "v", 3: "<"}  # MiniGrid*Direction UP, RIGHT, DOWN, LEFT


def grid_of(model, state: int, cur_state: Optional[int] = None) -> np.ndarray:
    """`_get_grid_representation(node)` of the model's family as a 2-D array of one-character strings."""
    fam = model.extra["family"]
    name = fam.name
    node = tuple(int(v) for v in model.nodes[state])
    if name == "DeepSea":  # deep_sea/base.py:309-313
        grid = np.full((fam.size, fam.size), " ", dtype="<U1")
        grid[node[1], node[0]] = "A"
        return grid[::-1, :]
    if name == "FrozenLake":  # frozen_lake/base.py:201-205
        grid = np.array(fam.lake, dtype="<U1").copy()
        grid[0, 0] = "F"
        grid[node[0], node[1]] = "A"
        return grid.T[::-1, :]
    if name in ("MiniGridEmpty", "MiniGridRooms"):
        if name == "MiniGridEmpty":  # minigrid_empty/base.py:252-264
            grid = np.full((fam.size, fam.size), " ", dtype="<U1")
        else:  # minigrid_rooms/base.py:291-331
            per_row = int(np.sqrt(fam.n_rooms))
            doors = [int(fam.room_size // 2) + i * (fam.room_size + 1) + 1 for i in range(per_row)]
            n = per_row * fam.room_size + per_row - 1
            grid = np.full((n, n), " ", dtype="<U1")
            for x in range(1, n + 1):
                for y in range(1, n + 1):
                    if x != n and x % (fam.room_size + 1) == 0 and y not in doors:
                        grid[y - 1, x - 1] = "W"
                    elif y != n and y % (fam.room_size + 1) == 0 and x not in doors:
                        grid[y - 1, x - 1] = "W"
        gx, gy = fam.goal_position
        grid[gy, gx] = "G"
        if cur_state is None:
            raise AttributeError("'NoneType' object has no attribute 'Dir'")  # the reference draws self.cur_node: None before reset()
        cx, cy, cd = (int(v) for v in model.nodes[cur_state])
        grid[cy, cx] = _ARROWS[cd]
        return grid[::-1, :]
    if name == "RiverSwim":  # river_swim/base.py:296-302
        grid = np.full((1, fam.size), " ", dtype="<U1")
        grid[0, 0] = "S"
        grid[0, -1] = "G"
        grid[0, node[0]] = "A"
        return grid
    if name == "SimpleGrid":  # simple_grid/base.py:267-294
        grid = np.full((fam.size, fam.size), " ", dtype="<U1")
        corners = {0: "---+", 1: "+++-", 2: "-+++", 3: "-++-"}[fam.reward_type]  # AND, NAND, OR, XOR
        grid[0, 0], grid[0, -1], grid[-1, 0], grid[-1, -1] = corners
        grid[node[1], node[0]] = "A"
        return grid[::-1, :]
    if name == "Taxi":  # taxi/base.py:355-365
        grid = np.full((fam.size, fam.size), "X", dtype="<U1")
        for cx, cy in fam.admissible:
            grid[cx, cy] = " "
        x, y, xp, yp, xd, yd = node
        grid[xd, yd] = "D"
        if xp != -1:
            grid[xp, yp] = "P"
        grid[x, y] = "A"
        return grid[::-1, :]
    raise NotImplementedError(f"no drawing for family {name!r}")


def _episodic_grid(grid: np.ndarray, h_now: int) -> np.ndarray:
    """What `EpisodicMDP.get_grid_representation` (base_finite.py:390-405) leaves after the encodings drop its two title
    rows (`grid[2:]`): the drawing, widened with 'X' columns while it is narrower than 2 + len(str(self.h)) -- the
    MDP's CURRENT in-episode time, not the one asked for."""
    while grid.shape[1] < 2 + len(str(h_now)):
        adder = np.full((grid.shape[1], 1), "X", dtype="<U1")
        grid = np.hstack((grid, adder))
    return grid


def sample_linear_value_features(v: np.ndarray, d: int, H: Optional[int] = None) -> np.ndarray:
    """`_sample_linear_value_features` (emission_maps/base.py:213-228), drawing from the global numpy stream."""
    psi = np.random.randn(v.size, d)
    psi[:, 0] = 1
    psi[:, 1] = v
    P = psi @ np.linalg.inv(psi.T @ psi) @ psi.T
    W = np.random.randn(v.size, d)
    W[:, 0] = 1
    W_p = P @ W
    features = W_p / np.linalg.norm(W_p, axis=0, keepdims=True)
    if H is not None:
        features = features.reshape(H + 1, -1, d)
    return features


def observation_table(model, name: str, cur_state: Optional[int] = None, h_now: int = 0, values=None,
                      d: Optional[int] = None) -> Optional[np.ndarray]:
    """float32 [S, *shape] (continuous) or [H, S, *shape] (episodic) -- `EmissionMap.all_observations`; None for Tabular.
    `cur_state` / `h_now`: the MDP's current state and in-episode time when the table is first needed (the MiniGrid
    drawings and the episodic padding rule read them, see the module docstring).  `values`: for the StateLinear maps, the
    value function the features are linear in (`optimal_value_functions[1]` / `random_value_functions[1]`), `d` their
    dimension (default max(10, int(0.1 * S)), emission_maps/base.py:168-172)."""
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
    if name in ("StateLinearOptimal", "StateLinearRandom"):
        assert values is not None, "the StateLinear maps need the value function they are linear in"
        dim = max(MIN_LINEAR_FEATURE_DIM, int(S * 0.1)) if d is None else d
        feats = sample_linear_value_features(np.asarray(values).ravel(), dim, H if H else None).astype(np.float32)
        return feats[:H] if H else feats  # node_to_observation indexes features[in_episode_time, state], time < H
    if name in ("TensorEncoding", "ImageEncoding"):
        symbols = SYMBOLS[model.extra["family"].name]
        index = {c: i for i, c in enumerate(symbols)}
        per_state = []
        for s in range(S):
            grid = grid_of(model, s, cur_state)
            if H:
                grid = _episodic_grid(grid, h_now)
            codes = np.vectorize(lambda c: index[c])(grid)  # a KeyError here is the reference's own (e.g. 'X' padding)
            per_state.append(codes)
        codes = np.stack(per_state)  # [S, rows, cols]
        if name == "ImageEncoding":
            img = codes.astype(np.float32)
            if not H:
                return img
            out = np.empty((H, S, img.shape[1] + 1, img.shape[2]), np.float32)
            out[:, :, 1:, :] = img[None]
            out[:, :, 0, :] = np.arange(H, dtype=np.float32)[:, None, None]
            return out
        planes = np.zeros((*codes.shape, len(symbols)), np.float32)
        np.put_along_axis(planes, codes[..., None], 1.0, axis=-1)
        if not H:
            return planes
        out = np.empty((H, S, *codes.shape[1:], len(symbols) + 1), np.float32)
        out[..., :-1] = planes[None]
        out[..., -1] = np.arange(H, dtype=np.float32)[:, None, None, None]
        return out
    raise NotImplementedError(f"emission map {name!r} is not one of the reference's")


# ---- noise -----------------------------------------------------------------------------------------------------------
class CompatNoise:
    """`Noise.__next__` (noises/base.py:51-56) for the four noise classes, draw for draw: a cache of samples from
    `RandomState(seed)`, converted to float32, popped from the front, refilled when empty.

    kind "GaussianUncorrelated"  rng.normal(0, scale, (5000, *shape))                          (gaussian_uncorrelated.py:13-14)
    kind "StudentTUncorrelated"  rng.standard_t(df, *shape) -- the requested count is IGNORED, the cache holds ONE
                                 array of the observation's shape and `list()` of it yields its leading-axis slices:
                                 scalars for vector observations (student_t_uncorrelated.py:13-14)
    kind "GaussianCorrelated"    W = wishart(scale=[scale] * prod(shape)).rvs(1, rng) once, then
                                 multivariate_normal(cov=W).rvs(5000, rng)                        (gaussian_correlated.py:14-18)
    kind "StudentTCorrelated"    the same W, multivariate_t(shape=W).rvs(5000, rng) (df = 1)      (student_t_correlated.py:14-18)"""

    KINDS = ("GaussianUncorrelated", "StudentTUncorrelated", "GaussianCorrelated", "StudentTCorrelated")

    def __init__(self, seed: int, shape: Tuple[int, ...], kind: str = "GaussianUncorrelated", scale: float = 0.1,
                 df: float = 3):
        assert kind in self.KINDS, kind
        self._rng = np.random.RandomState(seed)
        self.shape, self.kind, self.scale, self.df = tuple(shape), kind, scale, df
        self._cache = []
        self._rv = None

    def _sample(self, n: int) -> np.ndarray:
        if self.kind == "GaussianUncorrelated":
            return self._rng.normal(loc=0, scale=self.scale, size=(n, *self.shape))
        if self.kind == "StudentTUncorrelated":
            return self._rng.standard_t(self.df, *self.shape)
        from scipy.stats import multivariate_normal, multivariate_t, wishart

        if self._rv is None:
            W = wishart(scale=[self.scale] * int(np.prod(self.shape))).rvs(1, self._rng)
            self._rv = multivariate_normal(cov=W) if self.kind == "GaussianCorrelated" else multivariate_t(shape=W)
        return self._rv.rvs(n, self._rng).reshape(n, *self.shape)

    def __next__(self) -> np.ndarray:
        if not self._cache:
            self._cache = list(self._sample(NOISE_CACHE).astype(np.float32))
        return self._cache.pop(0)

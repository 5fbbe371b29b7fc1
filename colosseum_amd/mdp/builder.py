"""Host model builder: constructor arguments -> the tables the HIP kernels consume.

This is the build side of the hot path.  It restates, iteratively and
over integer tuples, the reference's recursive graph instantiation so that the *state indexing*, the
per-state *action permutation*, the per-(s,a) successor lists with their probabilities and the
per-(s,a) sampler seeds come out identical for the same constructor arguments:

  colosseum/mdp/base.py:408-409        the two per-MDP generators (numpy legacy MT19937, CPython MT19937)
  colosseum/mdp/base.py:463-503        instantiate_MDP: start sampler, DFS, discarded rand(S, A), index maps
  colosseum/mdp/base.py:505-539        action mapping drawn on first touch; sampler seed per (node, action)
  colosseum/mdp/utils/mdp_creation.py:212-231,234-243,276-310   DFS order, edge insertion, p_lazy/p_rand mixture
  colosseum/mdp/utils/mdp_creation.py:41-95                   dense T (float32 `+=` of duplicates), R = sum p*mean

numpy's legacy `RandomState` and CPython's `random.Random` are used directly: they are the very
generators the reference draws from, and both are frozen, documented algorithms.
"""
import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .families import Family, Node, StartSpec, dist_mean

REWARD_DETERMINISTIC = 0
REWARD_BETA = 1


@dataclass
class TabularModel:
    """One MDP instance as flat tables.  Row r = s * n_actions + a, with `a` the action the agent passes
    to `step` (i.e. after the reference's per-state action permutation)."""

    n_states: int
    n_actions: int
    H: int  # 0 = continuous (infinite horizon)
    nodes: np.ndarray  # int32 [S, k] node tuples in state-index order
    # sampler tables (creation order, duplicates kept -- the order `random.choices` bisects over)
    sp_ptr: np.ndarray  # int64 [S*A + 1]
    sp_next: np.ndarray  # int32 [nnz]
    sp_prob: np.ndarray  # float64 [nnz]
    sp_cum: np.ndarray  # float64 [nnz]  itertools.accumulate(probs) per row
    sp_seed: np.ndarray  # int32 [S*A]   seed handed to NextStateSampler (drawn for every row)
    sp_rkind: np.ndarray  # uint8 [nnz]   REWARD_DETERMINISTIC | REWARD_BETA
    sp_rp0: np.ndarray  # float64 [nnz] loc | beta a
    sp_rp1: np.ndarray  # float64 [nnz] 0   | beta b
    sp_rmean: np.ndarray  # float64 [nnz]
    start_states: np.ndarray  # int32 [n_start]
    start_probs: np.ndarray  # float64 [n_start]
    start_seed: int  # -1 when the start sampler is deterministic / unseeded
    rewards_range: Tuple[float, float] = (0.0, 1.0)
    extra: Dict = field(default_factory=dict)
    _dense: Optional[Tuple[np.ndarray, np.ndarray]] = None
    _csr: Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]] = None
    _R: Optional[np.ndarray] = None

    @property
    def is_episodic(self) -> bool:
        return self.H > 0

    @property
    def deterministic_rewards(self) -> bool:
        return bool((self.sp_rkind == REWARD_DETERMINISTIC).all())

    # -- DP views ------------------------------------------------------------------------------------
    def csr(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(ptr int32 [S*A+1], col int32 [nnz_u], val float32 [nnz_u]): unique successors per row in
        ascending column order, duplicates accumulated in float32 in creation order -- the non-zeros of
        the reference's dense `T` (mdp_creation.py:66-80) in `sparse.COO(T)` coordinate order."""
        if self._csr is None:
            SA = self.n_states * self.n_actions
            ptr = np.zeros(SA + 1, np.int32)
            cols: List[np.ndarray] = []
            vals: List[np.ndarray] = []
            for r in range(SA):
                lo, hi = int(self.sp_ptr[r]), int(self.sp_ptr[r + 1])
                acc: Dict[int, np.float32] = {}
                for j, p in zip(self.sp_next[lo:hi].tolist(), self.sp_prob[lo:hi].tolist()):
                    # numpy-2 (NEP 50) semantics of `T[i, a, j] += prob`: float32 + float32(prob)
                    acc[j] = np.float32(acc.get(j, np.float32(0.0)) + np.float32(p))
                ks = sorted(acc)
                cols.append(np.array(ks, np.int32))
                vals.append(np.array([acc[k] for k in ks], np.float32))
                ptr[r + 1] = ptr[r] + len(ks)
            self._csr = (ptr, np.concatenate(cols), np.concatenate(vals))
        return self._csr

    def reward_matrix(self) -> np.ndarray:
        """R[s, a] = float32(sum_k p_k * mean_k), float64 accumulation in creation order (mdp_creation.py:73-81).
        Computed once (the array is shared between calls and read-only); rows are padded to the longest one and added up
        entry by entry -- `acc + 0.0 * 0.0 == acc` exactly, so the padding leaves the left-to-right sums untouched."""
        if self._R is None:
            SA = self.n_states * self.n_actions
            ptr = np.asarray(self.sp_ptr, np.int64)
            n = np.diff(ptr)
            rows = np.repeat(np.arange(SA), n)
            pos = np.arange(len(self.sp_prob)) - np.repeat(ptr[:-1], n)
            K = int(n.max()) if SA else 0
            prod = np.zeros((SA, K))
            prod[rows, pos] = np.asarray(self.sp_prob, np.float64) * np.asarray(self.sp_rmean, np.float64)
            acc = np.zeros(SA)
            for k in range(K):
                acc = acc + prod[:, k]
            R = acc.astype(np.float32).reshape(self.n_states, self.n_actions)
            R.setflags(write=False)
            self._R = R
        return self._R

    def dense(self) -> Tuple[np.ndarray, np.ndarray]:
        """Dense (T float32 [S,A,S], R float32 [S,A]) as `BaseMDP.transition_matrix_and_rewards`."""
        if self._dense is None:
            ptr, col, val = self.csr()
            S, A = self.n_states, self.n_actions
            T = np.zeros((S * A, S), np.float32)
            rows = np.repeat(np.arange(S * A), np.diff(ptr))
            T[rows, col] = val
            self._dense = (T.reshape(S, A, S), self.reward_matrix())
        return self._dense


class _Row:
    __slots__ = ("next_nodes", "probs", "seed")

    def __init__(self, next_nodes, probs, seed):
        self.next_nodes, self.probs, self.seed = next_nodes, probs, seed


def build_model(
    family: Family,
    seed: int,
    episodic: bool,
    H: Optional[int] = None,
    randomize_actions: bool = True,
    p_lazy: Optional[float] = None,
    p_rand: Optional[float] = None,
    rewards_range: Tuple[float, float] = (0.0, 1.0),
) -> TabularModel:
    p_rand = p_rand if p_rand is None or p_rand > 0.0 else None  # base.py:389-390
    p_lazy = p_lazy if p_lazy is None or p_lazy > 0.0 else None
    if p_lazy is not None:
        assert 0 < p_lazy < 0.9999
    if p_rand is not None:
        assert 0 < p_rand < 0.9999
    family.check(p_lazy, p_rand)
    rewards_range = tuple(rewards_range) if rewards_range[0] < rewards_range[1] else tuple(rewards_range[::-1])

    rng = np.random.RandomState(seed)  # mdp._rng
    fast_rng = random.Random(seed)  # mdp._fast_rng
    A = family.n_actions

    start: StartSpec = family.start(rng, fast_rng)
    start_seed = fast_rng.randint(0, 10_000) if start.wants_seed else -1

    order: Dict[Node, int] = {}  # G.nodes insertion order
    has_succ = set()  # nodes with at least one outgoing edge
    adjacency: Dict[Node, Dict[Node, None]] = {}  # G.successors(node) order = order of first add_edge(node, .)
    action_map: Dict[Node, List[int]] = {}
    rdist_cache: Dict[Tuple[Node, int, Node], Tuple] = {}
    rows: Dict[Node, Dict[int, _Row]] = {}
    state = {"all_det": True}

    def get_action_mapping(node):
        m = action_map.get(node)
        if m is None:
            m = rng.rand(A).argsort().tolist() if randomize_actions else list(range(A))
            action_map[node] = m
        return m

    def get_reward_distribution(node, action, next_node):
        key = (node, action, next_node)
        d = rdist_cache.get(key)
        if d is None:
            d = family.reward_dist(node, get_action_mapping(node)[action], next_node)
            rdist_cache[key] = d
        return d

    def compute_transition(next_nodes, probs, node, action, next_node, p):
        next_nodes.append(next_node)
        probs.append(p)
        if state["all_det"] and get_reward_distribution(node, action, next_node)[0] != "deterministic":
            state["all_det"] = False
        if node not in order:
            order[node] = len(order)
        if next_node not in order:
            order[next_node] = len(order)
        has_succ.add(node)
        adjacency.setdefault(node, {}).setdefault(next_node, None)

    def individual_transition(node, action) -> _Row:
        next_nodes, probs = [], []
        p1_lazy = 1.0 if p_lazy is None else (1 - p_lazy)
        for nn, p in family.next_nodes(node, action):
            p = p1_lazy * p
            p = p if p_rand is None else ((1 - p_rand) * p + p * p_rand / A)
            compute_transition(next_nodes, probs, node, action, nn, p)
        if p_lazy is not None:
            compute_transition(next_nodes, probs, node, action, node, p_lazy)
        if p_rand is not None:
            for a in range(A):
                if a == action:
                    continue
                for nn, p in family.next_nodes(node, a):
                    p = p1_lazy * p_rand * p / A
                    compute_transition(next_nodes, probs, node, action, nn, p)
        assert np.isclose(sum(probs), 1.0)
        return _Row(next_nodes, probs, fast_rng.randint(0, 10_000))

    # iterative form of the recursive `instantiate_transitions` (frames: [node, action, row, child position])
    for sn in start.nodes:
        stack = [[sn, -1, None, 0, None]]
        while stack:
            fr = stack[-1]
            node = fr[0]
            if fr[1] == -1:
                if node in has_succ:  # G.has_node(node) and it already has successors
                    stack.pop()
                    continue
                fr[4] = {}
                fr[1] = 0
                fr[2] = None
            if fr[2] is None:
                if fr[1] == A:
                    assert all(a in fr[4] for a in range(A))
                    rows[node] = fr[4]
                    stack.pop()
                    continue
                fr[2] = individual_transition(node, fr[1])
                fr[3] = 0
            row = fr[2]
            if fr[3] < len(row.next_nodes):
                child = row.next_nodes[fr[3]]
                fr[3] += 1
                if child not in has_succ:
                    stack.append([child, -1, None, 0, None])
                continue
            fr[4][get_action_mapping(node)[fr[1]]] = row
            fr[1] += 1
            fr[2] = None

    S = len(order)
    rng.rand(S, A)  # drawn and discarded by the reference (base.py:487)
    nodes = list(order)
    index = order

    sp_ptr = np.zeros(S * A + 1, np.int64)
    sp_next, sp_prob, sp_cum, sp_kind, sp_p0, sp_p1, sp_mean = [], [], [], [], [], [], []
    sp_seed = np.zeros(S * A, np.int32)
    for i, node in enumerate(nodes):
        for a in range(A):
            row = rows[node][a]
            c = 0.0
            first = True
            for nn, p in zip(row.next_nodes, row.probs):
                sp_next.append(index[nn])
                sp_prob.append(p)
                c = p if first else c + p  # itertools.accumulate
                first = False
                sp_cum.append(c)
                d = get_reward_distribution(node, a, nn)
                if d[0] == "deterministic":
                    sp_kind.append(REWARD_DETERMINISTIC)
                    sp_p0.append(d[1])
                    sp_p1.append(0.0)
                else:
                    sp_kind.append(REWARD_BETA)
                    sp_p0.append(d[1])
                    sp_p1.append(d[2])
                sp_mean.append(dist_mean(d))
            sp_ptr[i * A + a + 1] = len(sp_next)
            sp_seed[i * A + a] = row.seed

    n_start = len(start.nodes)
    start_probs = [1.0] if start.probs is None or n_start == 1 else list(start.probs)
    model = TabularModel(
        n_states=S,
        n_actions=A,
        H=0,
        nodes=np.array(nodes, np.int32),
        sp_ptr=sp_ptr,
        sp_next=np.array(sp_next, np.int32),
        sp_prob=np.array(sp_prob, np.float64),
        sp_cum=np.array(sp_cum, np.float64),
        sp_seed=sp_seed,
        sp_rkind=np.array(sp_kind, np.uint8),
        sp_rp0=np.array(sp_p0, np.float64),
        sp_rp1=np.array(sp_p1, np.float64),
        sp_rmean=np.array(sp_mean, np.float64),
        start_states=np.array([index[n] for n in start.nodes], np.int32),
        start_probs=np.array(start_probs, np.float64),
        start_seed=int(start_seed) if n_start > 1 else -1,
        rewards_range=rewards_range,
    )
    model.extra["node_index"] = index
    model.extra["action_map"] = action_map
    # the MDP's numpy stream, positioned exactly where the reference's `mdp._rng` stands after construction: the
    # reward caches (mdp/base.py:1196-1203) and `random_step` (:1336,1353) continue from here
    model.extra["rng"] = rng
    # the same position as an immutable snapshot: batched handles with reference-exact reward caches
    # (CMDP_FLAG_REWARD_CACHE) continue a COPY of the stream, whatever a host sampler did to the object above meanwhile
    model.extra["rng_state"] = rng.get_state()
    model.extra["successors"] = [[index[x] for x in adjacency[n]] for n in nodes]  # networkx adjacency order
    if episodic:
        model.H = _time_horizon(model, family, H)
    return model


def _time_horizon(model: TabularModel, family: Family, H: Optional[int]) -> int:
    """EpisodicMDP._set_time_horizon (base_finite.py:103-122): 1 + the largest BFS distance from any
    *possible* starting node, raised to the requested H."""
    S, A = model.n_states, model.n_actions
    succ = [set() for _ in range(S)]
    for r in range(S * A):
        succ[r // A].update(model.sp_next[model.sp_ptr[r]: model.sp_ptr[r + 1]].tolist())
    index = model.extra["node_index"]
    worst = 0
    for sn in family.possible_starting_nodes():
        if sn not in index:
            raise KeyError(f"starting node {sn} is not part of the graph")  # networkx raises NodeNotFound
        dist = {index[sn]: 0}
        frontier = [index[sn]]
        while frontier:
            nxt = []
            for u in frontier:
                for v in succ[u]:
                    if v not in dist:
                        dist[v] = dist[u] + 1
                        nxt.append(v)
            frontier = nxt
        worst = max(worst, max(dist.values()))
    minimal = worst + 1
    return minimal if H is None else max(minimal, H)

"""Markov chain of a policy: transition probabilities, stationary distribution, average reward
(reference colosseum/mdp/utils/markov_chain.py:12-136).  The recurrent-class bookkeeping is host side (scipy's
strongly connected components instead of networkx's `attracting_components`: the same sets); the stationary
distribution of every recurrent class is computed on the GPU by the GTH kernel (`cmdp_gth`).

GTH for every size IS the reference's behaviour: above 500 x 500 entries `_get_stationary_distribution`
(markov_chain.py:206-233) first calls ARPACK's shift-invert solver (`_eigen_method`, :188-203), but when that succeeds the
function does not return -- it falls through to `sd = _gth_solve_numba(tps)` (:230), which overwrites the result; when it
fails, the fallback inside the branch is GTH as well.  Golden G9 holds a 576-state chain (S*S > 500*500) produced by the
reference: its distribution equals the GTH result to 1e-12, not the float32 eigenvector ARPACK returns (1e-8 off)."""
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import breadth_first_order, connected_components

from . import _lib as L


def get_average_rewards(R: np.ndarray, policy: np.ndarray) -> np.ndarray:
    return np.einsum("sa,sa->s", R, policy)


def get_transition_probabilities(T: np.ndarray, policy: np.ndarray) -> np.ndarray:
    return np.minimum(1.0, np.einsum("saj,sa->sj", T, policy))


def gth_batch(mats: Sequence[np.ndarray]) -> List[np.ndarray]:
    """Stationary distributions of single-recurrent-class chains on the device (float64 GTH).  The matrices are widened to
    float64 straight into the one buffer the call takes (no per-matrix copies: a batch of baselines is ~1 GB of them)."""
    lib = L.load()
    if not mats:
        return []
    dims = np.array([m.shape[0] for m in mats], np.int32)
    moff = np.concatenate([[0], np.cumsum(dims.astype(np.int64) ** 2)])
    flat = np.empty(int(moff[-1]), np.float64)
    for m, o, n in zip(mats, moff, dims):
        flat[o: o + int(n) * int(n)].reshape(n, n)[...] = m
    out = np.zeros(int(dims.sum()), np.float64)
    L.check(lib.cmdp_gth(len(mats), L.ptr(dims), L.ptr(flat), L.ptr(out)))
    off = np.concatenate([[0], np.cumsum(dims)])
    return [out[off[i]: off[i + 1]] for i in range(len(mats))]


def _dfs_preorder(indptr: np.ndarray, indices: np.ndarray, n: int) -> np.ndarray:
    """Preorder numbers of the depth-first search networkx's `strongly_connected_components` performs: sources in node
    order, neighbours in adjacency (= column) order."""
    pre = np.full(n, -1, np.int64)
    nxt = indptr[:-1].copy()
    cnt = 0
    for src in range(n):
        if pre[src] >= 0:
            continue
        pre[src] = cnt
        cnt += 1
        stack = [src]
        while stack:
            v = stack[-1]
            k, end = nxt[v], indptr[v + 1]
            while k < end and pre[indices[k]] >= 0:
                k += 1
            if k < end:
                w = indices[k]
                nxt[v] = k + 1
                pre[w] = cnt
                cnt += 1
                stack.append(w)
            else:
                nxt[v] = k
                stack.pop()
    return pre


def recurrent_classes(tps: np.ndarray) -> List[np.ndarray]:
    """Closed communicating classes (networkx's `attracting_components` of the chain's digraph), each in ascending
    state order, listed in the reference's order (markov_chain.py:95): networkx emits strongly connected components as
    its depth-first search completes them, and a component without outgoing edges is completed before the search leaves
    it -- so the attracting components come in the order in which that search first touches them."""
    n = tps.shape[0]  # dense array or scipy sparse matrix
    g = csr_matrix(tps > 0)
    ncomp, label = connected_components(g, directed=True, connection="strong")
    src, dst = g.nonzero()
    leaks = np.zeros(ncomp, bool)
    leaks[label[src[label[src] != label[dst]]]] = True
    classes = [np.flatnonzero(label == c) for c in range(ncomp) if not leaks[c]]
    if len(classes) > 1:
        g.sort_indices()
        pre = _dfs_preorder(g.indptr, g.indices, n)
        classes.sort(key=lambda cls: pre[cls].min())
    return classes


def stationary_sparse(P) -> np.ndarray:
    """Stationary distribution of a single-class chain given as a scipy sparse matrix, for chains too large for a dense
    float64 copy (config C5: 50 272 states; BUILD-DEFINED helper of `hardness.mixing_time`, no reference counterpart --
    the reference's own routines take dense arrays): the linear system pi (P - I) = 0 with the last equation replaced
    by sum(pi) = 1, solved by sparse LU (SuperLU through scipy), float64."""
    from scipy.sparse import identity, lil_matrix
    from scipy.sparse.linalg import spsolve

    n = P.shape[0]
    A = (csr_matrix(P, dtype=np.float64).T - identity(n, dtype=np.float64, format="csr")).tolil()
    A[n - 1, :] = 1.0
    b = np.zeros(n)
    b[n - 1] = 1.0
    pi = spsolve(A.tocsc(), b)
    pi = np.maximum(pi, 0.0)
    return pi / pi.sum()


def _class_distribution(tps: np.ndarray, cls: np.ndarray) -> np.ndarray:
    if len(cls) == 1:
        return np.ones(1)
    return gth_batch([tps[np.ix_(cls, cls)]])[0]


def get_stationary_distribution(tps: np.ndarray,
                                starting_states_and_probs: Optional[Iterable[Tuple[int, float]]]) -> np.ndarray:
    """markov_chain.py:64-136."""
    n = len(tps)
    classes = recurrent_classes(tps)
    if len(classes) == 1 and len(classes[0]) < n:
        sd = np.zeros(n, np.float32)
        sd[classes[0]] = _class_distribution(tps, classes[0])
        return sd
    if len(classes) > 1:
        sd = np.zeros(n)
        g = csr_matrix(tps > 0)
        for ss, p in starting_states_and_probs:
            reach = set(breadth_first_order(g, ss, directed=True, return_predecessors=False).tolist())
            for cls in classes:  # the first recurrent class the starting state is connected to
                if int(cls[0]) in reach:
                    sd[cls] += p * _class_distribution(tps, cls)
                    break
        return sd
    return _class_distribution(tps, np.arange(n))


def _stationary_plan(tps: np.ndarray, starting_states_and_probs):
    """Which recurrent classes enter the stationary distribution and with which weight: (result dtype, [(cls, weight)])."""
    n = len(tps)
    classes = recurrent_classes(tps)
    if len(classes) == 1 and len(classes[0]) < n:
        return np.float32, [(classes[0], None)]
    if len(classes) > 1:
        g = csr_matrix(tps > 0)
        plan = []
        for ss, p in starting_states_and_probs:
            reach = set(breadth_first_order(g, ss, directed=True, return_predecessors=False).tolist())
            for cls in classes:
                if int(cls[0]) in reach:
                    plan.append((cls, p))
                    break
        return np.float64, plan
    return None, [(np.arange(n), None)]


def get_average_reward_batch(problems, builtin_sum: bool = False) -> List[float]:
    """`get_average_reward` for many (T, R, policy, starting_states_and_probs) at once: the recurrent-class bookkeeping
    per problem on the host, ALL GTH eliminations in one device call.  `builtin_sum`: the final dot product as Python's
    `sum(sd * ars)` (index order) instead of numpy's pairwise `.sum()` -- how `BaseMDP.optimal_average_reward` /
    `worst_average_reward` / `random_average_reward` add it up (colosseum/mdp/base.py:895-941), which is what MDPLoop's
    normalisers read (agent_mdp_interaction.py:362-386); the two differ by an ulp on some chains."""
    prepared, mats = [], []
    for T, R, policy, starts in problems:
        assert np.isclose(policy.sum(-1), 1).all(), "the policy specification is incorrect."
        ars = get_average_rewards(R, policy)
        tps = get_transition_probabilities(T, policy)
        dtype, plan = _stationary_plan(tps, starts)
        slots = []
        for cls, w in plan:
            if len(cls) == 1:
                slots.append((cls, w, None))
            else:
                slots.append((cls, w, len(mats)))
                mats.append(tps if len(cls) == len(tps) else tps[np.ix_(cls, cls)])   # the whole chain: no gather
        prepared.append((ars, len(tps), dtype, slots))
    sols = gth_batch(mats)
    out = []
    for ars, n, dtype, slots in prepared:
        if dtype is None:
            cls, _, k = slots[0]
            sd = np.ones(1) if k is None else sols[k]
        else:
            sd = np.zeros(n, dtype)
            for cls, w, k in slots:
                x = np.ones(1) if k is None else sols[k]
                if w is None:
                    sd[cls] = x
                else:
                    sd[cls] += w * x
        out.append(sum(ars * sd) if builtin_sum else (ars * sd).sum())
    return out


def get_average_reward(T: np.ndarray, R: np.ndarray, policy: np.ndarray, next_states_and_probs) -> float:
    """markov_chain.py:12-31."""
    assert np.isclose(policy.sum(-1), 1).all(), "the policy specification is incorrect."
    average_rewards = get_average_rewards(R, policy)
    tps = get_transition_probabilities(T, policy)
    sd = get_stationary_distribution(tps, next_states_and_probs)
    return (average_rewards * sd).sum()


class AverageRewardCache:
    """`get_average_reward(T, R, policy, [(current state, 1.0)])` for the B instances of a batch at every logging step
    of the continuous loop, memoised per instance on the policy: late in a run the greedy policy of a Q-learning agent
    changes rarely, and a chain with one recurrent class gives the same value from every state.

    Per new policy: the chain (rows of one-hot policy states are plain gathers T[s, a(s), :] -- exactly what the einsum
    of the reference produces for them), its recurrent classes, ONE batched GTH call for all missing classes of all
    instances, and the average reward of every class.  Chains with several recurrent classes additionally memoise, per
    current state, the first class (in class-list order) that state reaches."""

    def __init__(self, TR: Sequence[Tuple[np.ndarray, np.ndarray]], max_entries: int = 4096):
        self.TR = TR
        self.max_entries = max_entries
        self.memo: List[dict] = [dict() for _ in TR]
        self.hits = self.misses = 0

    @staticmethod
    def _chain(T, R, policy):
        onehot = (policy == 1).any(-1)
        if onehot.all():
            a = policy.argmax(-1)
            idx = np.arange(len(policy))
            return R[idx, a] * np.float32(1.0), np.minimum(1.0, T[idx, a])
        return get_average_rewards(R, policy), get_transition_probabilities(T, policy)

    def __call__(self, need: np.ndarray, policies: Sequence[np.ndarray], current: Sequence[int]) -> list:
        todo, mats = [], []
        for b in np.flatnonzero(need):
            key = policies[b].tobytes()
            e = self.memo[b].get(key)
            if e is not None:
                self.hits += 1
                continue
            self.misses += 1
            T, R = self.TR[b]
            policy = policies[b]
            assert np.isclose(policy.sum(-1), 1).all(), "the policy specification is incorrect."
            ars, tps = self._chain(T, R, policy)
            n = len(tps)
            classes = recurrent_classes(tps)
            if len(classes) == 1:
                mode = "sub" if len(classes[0]) < n else "all"
            else:
                mode = "multi"
            slots = []
            for cls in classes:
                if len(cls) == 1:
                    slots.append(None)
                else:
                    slots.append(len(mats))
                    mats.append(tps[np.ix_(cls, cls)])
            if len(self.memo[b]) >= self.max_entries:
                self.memo[b].clear()
            e = dict(mode=mode, classes=classes, graph=csr_matrix(tps > 0) if mode == "multi" else None, first={})
            self.memo[b][key] = e
            todo.append((e, ars, n, slots))
        if todo:
            sols = gth_batch(mats)
            for e, ars, n, slots in todo:
                vals = []
                for cls, k in zip(e["classes"], slots):
                    x = np.ones(1) if k is None else sols[k]
                    if e["mode"] == "all":
                        sd = x
                    else:
                        sd = np.zeros(n, np.float32 if e["mode"] == "sub" else np.float64)
                        if e["mode"] == "sub":
                            sd[cls] = x
                        else:
                            sd[cls] += 1.0 * x
                    vals.append((ars * sd).sum())
                e["values"] = vals
        out = []
        for b in np.flatnonzero(need):
            e = self.memo[b][policies[b].tobytes()]
            if e["mode"] != "multi":
                out.append(e["values"][0])
                continue
            cur = int(current[b])
            k = e["first"].get(cur)
            if k is None:
                reach = set(breadth_first_order(e["graph"], cur, directed=True, return_predecessors=False).tolist())
                k = next((i for i, cls in enumerate(e["classes"]) if int(cls[0]) in reach), -1)
                e["first"][cur] = k
            # no class reached cannot happen in a finite chain; the reference would return an all-zero distribution
            out.append(e["values"][k] if k >= 0 else np.float64(0.0))
        return out

"""Hardness measures of batches of MDPs on the GPU (reference colosseum/hardness/measures/*, called through
`BaseMDP.diameter` / `.value_norm`, colosseum/mdp/base.py:996-1016,1042-1081).

`diameter(models)` and `value_norm(models)` take TabularModels (colosseum_amd.mdp.make_model) and return one float
per model.  Episodic MDPs use the time-augmented diameter and the value norm of their continuous form, as the
reference does."""
from typing import List, Sequence

import numpy as np

from .. import _lib as L
from ..batched import BatchedMDP
from ..dp_handle import DPBatch
from ..mdp.builder import TabularModel
from ..mdp.episodic import continuous_form


def _vi_rule(S, A, nnz):
    size = float(S) * A * S
    return L.SCHEME_JACOBI if (size > 300 * 3 * 300 and nnz / size < 0.2) else L.SCHEME_GAUSS_SEIDEL


def diameter(models: Sequence[TabularModel], epsilon: float = 1e-3, variant: str = "per_target") -> np.ndarray:
    """`get_diameter` per model.  Models are grouped by (episodic?, H, A, scheme) into batches.

    variant "per_target" (default): every target solved to the stopping rule, the maximum taken -- what the reference
    computes on a machine with >= 3 cores (diameter.py:35-36,108-124) and, below 1000 states, on any machine.
    variant "reference_single_core": the reference's dispatch with one core (its default, config.py:19): continuous MDPs
    above 1000 states take `_get_sparse_diameter` (float64, order-dependent early exit; `BatchedMDP.diameter_sparse_f64`),
    everything else as above."""
    assert variant in ("per_target", "reference_single_core")
    out = np.zeros(len(models), np.float64)
    if variant == "reference_single_core":
        big = [i for i, m in enumerate(models) if not m.is_episodic and m.n_states > 1000]
        small = [i for i in range(len(models)) if i not in set(big)]
        if small:
            out[small] = diameter([models[i] for i in small], epsilon)
        by_A = {}
        for i in big:
            by_A.setdefault(models[i].n_actions, []).append(i)
        for A, idx in by_A.items():
            dp = BatchedMDP([models[i] for i in idx], with_env=False)
            out[idx] = dp.diameter_sparse_f64(epsilon)[0]
            dp.close()
        return out
    groups = {}
    for i, m in enumerate(models):
        key = (m.H, m.n_actions, 0 if m.is_episodic else _vi_rule(m.n_states, m.n_actions, len(m.csr()[1])))
        groups.setdefault(key, []).append(i)
    for (H, A, scheme), idx in groups.items():
        dp = BatchedMDP([models[i] for i in idx], with_env=False)
        d = dp.diameter_episodic(epsilon)[0] if H > 0 else dp.diameter(epsilon, scheme)[0]
        out[idx] = d
        dp.close()
    return out


def sum_reciprocals_suboptimality_gaps(models: Sequence[TabularModel], regularization: float = 0.1) -> np.ndarray:
    """`get_sum_reciprocals_suboptimality_gaps` (colosseum/hardness/measures/sum_reciprocals_suboptimality_gaps.py:6-28)
    on the optimal value functions (`BaseMDP.sum_reciprocals_suboptimality_gaps`, mdp/base.py:1018-1040): discounted VI
    (gamma 0.99, eps 1e-3) for continuous MDPs, backward induction restricted to the reachable (h, s) pairs for
    episodic ones.  The DP runs on the GPU; the final reduction is the reference's numpy expression."""
    from ..mdp.episodic import episodic_graph_nodes

    out = np.zeros(len(models), np.float64)
    groups = {}
    for i, m in enumerate(models):
        key = (m.H, m.n_actions, 0 if m.is_episodic else _vi_rule(m.n_states, m.n_actions, len(m.csr()[1])))
        groups.setdefault(key, []).append(i)
    for (H, A, scheme), idx in groups.items():
        ms = [models[i] for i in idx]
        dp = BatchedMDP(ms, with_env=False)
        if H > 0:
            Q, V = dp.episodic_value_iteration()
            for j, m in enumerate(ms):
                S = m.n_states
                q = dp.split_rows(Q, H + 1)[j].reshape(H + 1, S, A)
                v = dp.split_states(V, H + 1)[j].reshape(H + 1, S)
                gaps = v[..., None] - q
                reach = episodic_graph_nodes(m)[0]
                gaps = np.vstack([gaps[h, s] for h, s in reach])
                out[idx[j]] = (1 / (gaps + regularization)).sum()
        else:
            Q, V, _ = dp.value_iteration(0.99, 1e-3, scheme)
            for j, m in enumerate(ms):
                q = dp.split_rows(Q)[j].reshape(m.n_states, A)
                v = dp.split_states(V)[j]
                out[idx[j]] = (1 / (v[..., None] - q + regularization)).sum()
        dp.close()
    return out


def value_norm(models: Sequence[TabularModel]) -> np.ndarray:
    """`BaseMDP.discounted_value_norm`: 0 for fully deterministic MDPs, else
    `calculate_norm_discounted(T, V*)` with V* = discounted_value_iteration(T, R) (gamma 0.99, eps 1e-3, the
    reference's scheme rule) -- on (T_cf, R_cf) for episodic MDPs."""
    out = np.zeros(len(models), np.float64)
    problems: List = []
    owner: List[int] = []
    for i, m in enumerate(models):
        if (np.diff(m.sp_ptr) == 1).all() and m.deterministic_rewards:
            continue  # mdp/base.py:1070-1074
        if m.is_episodic:
            N, A, csr, R = continuous_form(m)
        else:
            N, A, csr, R = m.n_states, m.n_actions, m.csr(), m.reward_matrix()
        problems.append((N, A, csr, R))
        owner.append(i)
    groups = {}
    for j, (N, A, csr, R) in enumerate(problems):
        groups.setdefault((A, _vi_rule(N, A, len(csr[1]))), []).append(j)
    for (A, scheme), js in groups.items():
        with DPBatch([problems[j] for j in js]) as dp:
            Q, V, _ = dp.value_iteration(0.99, 1e-3, scheme, 1_000_000)
            vn = dp.value_norm(V)
        for j, v in zip(js, vn):
            out[owner[j]] = v
    return out


def mixing_time(models: Sequence[TabularModel], policies=None, threshold: float = 0.25, max_steps: int = 1_000_000):
    """BUILD-DEFINED hardness measure (the reference has no mixing time, parity unpinned): for the
    chain of `policies[i]` ([S, A] action probabilities; default the uniform policy) on `models[i]`, the smallest t with
    max_s TV(P^t(s, .), pi) <= threshold, where pi is the stationary distribution of the chain's single recurrent class
    (GTH kernel).  Returns (t_mix, tv): t_mix = -1 when the chain has several recurrent classes or does not get below
    the threshold within max_steps (periodic chains)."""
    from scipy.sparse import csr_matrix

    from ..markov_chain import gth_batch, recurrent_classes, stationary_sparse

    n = len(models)
    t_out = np.full(n, -1, np.int64)
    tv_out = np.full(n, np.nan)
    groups = {}
    for i, m in enumerate(models):
        groups.setdefault((m.n_actions, m.n_states > 1024), []).append(i)  # large chains one at a time (S x S matrices in HBM)
    for (A, large), idx in groups.items():
        stats, keep, pols = [], [], []
        for i in idx:
            m = models[i]
            S = m.n_states
            pol = np.full((S, A), 1.0 / A, np.float32) if policies is None else np.asarray(policies[i], np.float32)
            # the chain as a sparse matrix: P[s, j] = sum_a pi[s, a] T[s, a, j] in float64 (never a dense S x A x S array)
            ptr, col, val = m.csr()
            rows = np.repeat(np.arange(S * A) // A, np.diff(ptr))
            w = np.repeat(pol.astype(np.float64).ravel(), np.diff(ptr))
            P = csr_matrix((w * val.astype(np.float64), (rows, col)), shape=(S, S))
            P.sum_duplicates()
            P.eliminate_zeros()
            from scipy.sparse import diags

            P = csr_matrix(diags(1.0 / np.asarray(P.sum(axis=1)).ravel()) @ P)  # rows normalised, as the library defines the chain
            classes = recurrent_classes(P)
            if len(classes) != 1:
                continue
            cls = classes[0]
            sd = np.zeros(S)
            if len(cls) == 1:
                sd[cls] = 1.0
            elif len(cls) > 2048:
                sd[cls] = stationary_sparse(P[cls][:, cls])
            else:
                sd[cls] = gth_batch([P[cls][:, cls].toarray()])[0]
            stats.append(sd)
            keep.append(i)
            pols.append(pol)
        if not keep:
            continue
        for part in ([[k] for k in range(len(keep))] if large else [list(range(len(keep)))]):
            dp = BatchedMDP([models[keep[k]] for k in part], with_env=False)
            t, tv = dp.mixing_time([stats[k] for k in part], None if policies is None else [pols[k] for k in part],
                                   threshold, max_steps)
            dp.close()
            t_out[[keep[k] for k in part]] = t
            tv_out[[keep[k] for k in part]] = tv
    return t_out, tv_out

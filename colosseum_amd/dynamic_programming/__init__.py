"""Drop-in for `colosseum.dynamic_programming` (reference colosseum/dynamic_programming/*.py): the same
free functions on dense float32 `T[S,A,S]`, `R[S,A]`, executed by the HIP sweep kernels of libcmdp.so.

Signatures, defaults, return conventions (`None` when `max_abs_value` is exceeded, the
`DynamicProgrammingMaxIterationExceeded` exception after 10**6 sweeps) follow the reference."""
import numpy as np

from .. import _lib as L
from .._lib import DynamicProgrammingMaxIterationExceeded  # noqa: F401
from ..dp_handle import DPBatch, csr_from_dense

DP_MAX_ITERATION = int(1e6)
ARGMAX_SEED = 42

# The reference's functions take dense arrays and own no state; an `MDPLoop` calls them with the SAME (T, R) at every
# logging step.  The device handle of a (T, R) pair (CSR extraction, allocation, upload) is therefore kept and found
# again by content: a few of them, least recently used first out.
_HANDLES = {}
_MAX_HANDLES = 4


def _handle(T, R):
    import hashlib

    T = np.ascontiguousarray(T, np.float32)
    R = np.ascontiguousarray(R, np.float32)
    key = (T.shape, hashlib.blake2b(T.data, digest_size=16).digest(), hashlib.blake2b(R.data, digest_size=16).digest())
    dp = _HANDLES.pop(key, None)
    if dp is None:
        S, A, _ = T.shape
        dp = DPBatch([(S, A, csr_from_dense(T), R)])
        dp.nnz = int(dp._keep["csr_ptr"][-1])
        while len(_HANDLES) >= _MAX_HANDLES:
            _HANDLES.pop(next(iter(_HANDLES))).close()
    _HANDLES[key] = dp  # most recently used last
    return dp


def release_handles():
    """Frees the cached device handles (they are also freed at interpreter exit)."""
    while _HANDLES:
        _HANDLES.popitem()[1].close()


def _vi_rule(T_size, nnz, sparse_n_states_threshold, sparse_nnz_per_threshold):
    """reference infinite_horizon.py:28-36."""
    if T_size > sparse_n_states_threshold and nnz / T_size < sparse_nnz_per_threshold:
        return L.SCHEME_JACOBI
    return L.SCHEME_GAUSS_SEIDEL


def discounted_value_iteration(T, R, gamma=0.99, epsilon=1e-3, max_abs_value=None,
                               sparse_n_states_threshold=300 * 3 * 300, sparse_nnz_per_threshold=0.2):
    """reference infinite_horizon.py:14-44.  Returns (Q [S,A], V [S]) float32, or None."""
    S, A, _ = T.shape
    dp = _handle(T, R)
    scheme = _vi_rule(T.size, dp.nnz, sparse_n_states_threshold, sparse_nnz_per_threshold)
    try:
        Q, V, _ = dp.value_iteration(gamma, epsilon, scheme, DP_MAX_ITERATION, max_abs_value)
    except L.CmdpError as e:
        if e.code == L.ERR_MAX_VALUE:
            return None
        raise
    return Q.reshape(S, A), V


def discounted_policy_evaluation(T, R, pi, gamma=0.99, epsilon=1e-7, sparse_n_states_threshold=200,
                                 sparse_nnz_per_threshold=0.2):
    """reference infinite_horizon.py:47-64."""
    S, A, _ = T.shape
    dp = _handle(T, R)
    scheme = (L.SCHEME_JACOBI if (S > sparse_n_states_threshold and dp.nnz / T.size < sparse_nnz_per_threshold)
              else L.SCHEME_GAUSS_SEIDEL)
    Q, V, _ = dp.policy_evaluation(np.asarray(pi, np.float32).ravel(), gamma, epsilon, scheme, DP_MAX_ITERATION)
    return Q.reshape(S, A), V


def episodic_value_iteration(H, T, R, max_value=None):
    """reference finite_horizon.py:11-26.  Returns (Q [H+1,S,A], V [H+1,S]) or None when a value exceeds
    `max_value`."""
    S, A, _ = T.shape
    Q, V = _handle(T, R).episodic_value_iteration(int(H))
    Q, V = Q.reshape(H + 1, S, A), V.reshape(H + 1, S)
    if max_value is not None and (V > max_value).any():
        return None
    return Q, V


def episodic_policy_evaluation(H, T, R, policy):
    """reference finite_horizon.py:29-42; policy [H,S,A]."""
    S, A, _ = T.shape
    Q, V = _handle(T, R).episodic_policy_evaluation(np.asarray(policy, np.float32).ravel(), int(H))
    return Q.reshape(H + 1, S, A), V.reshape(H + 1, S)


def discounted_policy_iteration(T, R, gamma=0.99, epsilon=1e-7):
    """reference infinite_horizon.py:208-219 (the initial Q is drawn from the unseeded global numpy RNG
    there too)."""
    S, A, _ = T.shape
    Q = np.random.rand(S, A)
    pi = argmax_2d(Q)
    for _ in range(DP_MAX_ITERATION):
        old_pi = pi.copy()
        Q, V = discounted_policy_evaluation(T, R, pi, gamma, epsilon)
        pi = argmax_2d(Q)
        if (pi != old_pi).sum() == 0:
            return Q, V, pi
    raise DynamicProgrammingMaxIterationExceeded()


# ---- argmax with uniform random tie-break (reference dynamic_programming/utils.py:12-100) ----------------
# The reference re-seeds *numba's* generator with 42 on every call; without numba the tie winner cannot be
# reproduced (SURVEY 8c caveat 1).  Tie-free rows are exact; ties are broken by RandomState(42).choice,
# which is what the reference computes when its njit decorators are the identity.
def _pick(rs, row):
    return rs.choice(np.where(row == row.max())[0])


def _pick_rows(M):
    """`_pick` for every row of M [n, A], in row order, from ONE RandomState(ARGMAX_SEED): rows with a single maximum are
    taken in one vectorised pass -- `RandomState.choice` on one candidate draws nothing, so the stream is consumed by the
    tie rows only, in their order, exactly as the row-by-row loop does."""
    M = np.asarray(M)
    n = len(M)
    if n == 0:
        return np.zeros(0, np.int64)
    is_max = M == M.max(axis=1, keepdims=True)
    picks = is_max.argmax(axis=1)
    ties = np.flatnonzero(is_max.sum(axis=1) != 1)   # also rows of NaNs (no element equals the maximum): left to `_pick`
    if len(ties):
        rs = np.random.RandomState(ARGMAX_SEED)
        for s in ties.tolist():
            picks[s] = _pick(rs, M[s])
    return picks


def argmax_2d(A):
    X = np.zeros_like(A, np.float32)
    X[np.arange(len(A)), _pick_rows(A)] = 1
    return X


def argmax_3d(A):
    H, S = A.shape[:2]
    X = np.zeros(A.shape, np.float32)
    X.reshape(H * S, -1)[np.arange(H * S), _pick_rows(A.reshape(H * S, -1))] = 1.0
    return X


def get_deterministic_policy_from_q_values(Q):
    return _pick_rows(Q).astype(np.int32)


def get_deterministic_policy_from_q_values_finite_horizon(Q):
    H, S = Q.shape[:2]
    return _pick_rows(Q.reshape(H * S, -1)).astype(np.int32).reshape(H, S)


def get_policy_from_q_values(Q, stochastic_form=False):
    if Q.ndim == 3:
        return argmax_3d(Q) if stochastic_form else get_deterministic_policy_from_q_values_finite_horizon(Q)
    return argmax_2d(Q) if stochastic_form else get_deterministic_policy_from_q_values(Q)

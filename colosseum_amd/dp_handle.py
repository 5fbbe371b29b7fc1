"""DP-only handles: a batch of (S, A, CSR, R) problems on the device, without the sampler half."""
import ctypes as C

import numpy as np

from . import _lib as L


def csr_from_dense(T):
    """Non-zeros of a dense [S,A,S] array in `sparse.COO(T)` coordinate order -> (ptr, col, val)."""
    S, A, S2 = T.shape
    T2 = np.ascontiguousarray(T, np.float32).reshape(S * A, S2)
    rows, cols = np.nonzero(T2)
    ptr = np.zeros(S * A + 1, np.int64)
    np.add.at(ptr, rows + 1, 1)
    return np.cumsum(ptr).astype(np.int64), cols.astype(np.int32), T2[rows, cols].astype(np.float32)


class DPBatch:
    """problems: sequence of (S, A, (ptr, col, val), R[S,A])."""

    def __init__(self, problems):
        lib = L.load()
        problems = list(problems)
        A = problems[0][1]
        assert all(p[1] == A for p in problems)
        self.B, self.A = len(problems), A
        S = np.array([p[0] for p in problems], np.int64)
        self.state_off = np.concatenate([[0], np.cumsum(S)]).astype(np.int64)
        self.row_off = self.state_off * A
        nz = np.array([len(p[2][1]) for p in problems], np.int64)
        nz_off = np.concatenate([[0], np.cumsum(nz)])
        keep = dict(
            state_off=self.state_off,
            csr_ptr=np.concatenate([np.asarray(p[2][0][:-1], np.int64) + nz_off[i] for i, p in enumerate(problems)]
                                   + [nz_off[-1:]]).astype(np.int64),
            csr_col=np.concatenate([p[2][1] for p in problems]).astype(np.int32),
            csr_val=np.concatenate([p[2][2] for p in problems]).astype(np.float32),
            R=np.concatenate([np.asarray(p[3], np.float32).ravel() for p in problems]).astype(np.float32),
        )
        d = L.CmdpDesc()
        d.n_instances, d.n_actions, d.horizon, d.rng_mode, d.layout = self.B, A, 0, L.RNG_PHILOX, L.LAYOUT_CSR
        d.reward_min, d.reward_max = 0.0, 1.0
        for k, v in keep.items():
            keep[k] = np.ascontiguousarray(v)
            setattr(d, k, L.ptr(keep[k]))
        self._keep = keep
        self._h = C.c_void_p()
        self._lib = lib
        L.check(lib.cmdp_create(C.byref(self._h), C.byref(d)))

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.cmdp_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _out(self, lead=1):
        return (np.zeros(lead * int(self.row_off[-1]), np.float32), np.zeros(lead * int(self.state_off[-1]), np.float32))

    def value_iteration(self, gamma, epsilon, scheme, max_sweeps, max_abs_value=None, R=None):
        Q, V = self._out()
        sw = np.zeros(self.B, np.int64)
        Rov = L.carr(R, np.float32)
        L.check(self._lib.cmdp_vi_discounted(self._h, gamma, epsilon, scheme, max_sweeps,
                                             0.0 if max_abs_value is None else float(max_abs_value), L.ptr(Rov),
                                             L.ptr(Q), L.ptr(V), L.ptr(sw)))
        return Q, V, sw

    def policy_evaluation(self, pi, gamma, epsilon, scheme, max_sweeps, R=None):
        Q, V = self._out()
        sw = np.zeros(self.B, np.int64)
        p = L.carr(pi, np.float32)
        Rov = L.carr(R, np.float32)
        L.check(self._lib.cmdp_pe_discounted(self._h, L.ptr(p), gamma, epsilon, scheme, max_sweeps, L.ptr(Rov),
                                             L.ptr(Q), L.ptr(V), L.ptr(sw)))
        return Q, V, sw

    def episodic_value_iteration(self, H, R=None):
        Q, V = self._out(H + 1)
        Rov = L.carr(R, np.float32)
        L.check(self._lib.cmdp_vi_episodic(self._h, H, L.ptr(Rov), L.ptr(Q), L.ptr(V)))
        return Q, V

    def episodic_policy_evaluation(self, pi, H, R=None):
        Q, V = self._out(H + 1)
        p = L.carr(pi, np.float32)
        Rov = L.carr(R, np.float32)
        L.check(self._lib.cmdp_pe_episodic(self._h, H, L.ptr(p), L.ptr(Rov), L.ptr(Q), L.ptr(V)))
        return Q, V

    def diameter(self, epsilon=1e-3, scheme=L.SCHEME_AUTO, max_sweeps=1_000_000):
        per = np.zeros(int(self.state_off[-1]), np.float32)
        diam = np.zeros(self.B, np.float32)
        L.check(self._lib.cmdp_diameter(self._h, epsilon, scheme, max_sweeps, L.ptr(per), L.ptr(diam)))
        return diam, per

    def value_norm(self, V):
        v = L.carr(V, np.float32)
        out = np.zeros(self.B, np.float32)
        L.check(self._lib.cmdp_value_norm(self._h, L.ptr(v), L.ptr(out)))
        return out

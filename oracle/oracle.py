"""ctypes front-end of the CPU oracle (oracle/cmdp_oracle.c).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by anything
under colosseum_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C")


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libcmdp_oracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, "cmdp_oracle.c")):
            build()
        L = C.CDLL(path)
        L.oracle_env_create.restype = C.c_void_p
        L.oracle_env_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_int32, C.c_uint64]
        L.oracle_env_destroy.argtypes = [C.c_void_p]
        L.oracle_env_set_reward_dists.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_env_set_beta_gammas.argtypes = [C.c_void_p, C.c_int]
        L.oracle_env_set_dense.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_env_reset.restype = C.c_int32
        L.oracle_env_reset.argtypes = [C.c_void_p]
        L.oracle_env_step.restype = C.c_int
        L.oracle_env_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_env_rollout.restype = C.c_int
        L.oracle_env_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.oracle_env_visits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_env_reset_visits.argtypes = [C.c_void_p]
        L.oracle_env_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_python_random.argtypes = [C.c_uint32, C.c_int, _f64p]
        L.oracle_philox.argtypes = [C.c_uint32] * 6 + [C.c_void_p]
        L.oracle_vi_scheme_rule.argtypes = [C.c_int, C.c_int, C.c_int64]
        L.oracle_pe_scheme_rule.argtypes = [C.c_int, C.c_int, C.c_int64]
        L.oracle_vi_discounted.restype = C.c_int64
        L.oracle_vi_discounted.argtypes = [C.c_int, C.c_int, _i64p, _i32p, _f32p, _f32p, C.c_float, C.c_double,
                                           C.c_int, C.c_int64, C.c_double, _f32p, _f32p]
        L.oracle_pe_discounted.restype = C.c_int64
        L.oracle_pe_discounted.argtypes = [C.c_int, C.c_int, _i64p, _i32p, _f32p, _f32p, _f32p, C.c_float,
                                           C.c_double, C.c_int, C.c_int64, _f32p, _f32p]
        L.oracle_episodic.argtypes = [C.c_int, C.c_int, C.c_int, _i64p, _i32p, _f32p, _f32p, C.c_void_p, _f32p, _f32p]
        L.oracle_diameter_continuous.argtypes = [C.c_int, C.c_int, _i64p, _i32p, _f32p, C.c_double, C.c_int,
                                                 C.c_int64, C.c_void_p, C.c_void_p]
        L.oracle_value_norm.restype = C.c_float
        L.oracle_value_norm.argtypes = [C.c_int, C.c_int, _i64p, _i32p, _f32p, _f32p]
        L.oracle_diameter_episodic.restype = C.c_int
        L.oracle_diameter_episodic.argtypes = [C.c_int, C.c_int, C.c_int, _i64p, _i32p, _f32p, C.c_int, _i32p, _f32p,
                                               C.c_double, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.oracle_gth.restype = C.c_int
        L.oracle_gth.argtypes = [C.c_int, _f64p, _f64p]
        L.oracle_batch_rollout.restype = C.c_int
        L.oracle_batch_rollout.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double] + \
            [C.c_void_p] * 11 + [C.c_int64] + [C.c_void_p] * 4
        L.oracle_batch_vi.restype = C.c_int64
        L.oracle_batch_vi.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + \
            [C.c_float, C.c_double, C.c_int, C.c_int64] + [C.c_void_p] * 3
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleEnv:
    """One instance of the interaction loop on the CPU oracle.  `model` is a colosseum_amd TabularModel
    (only its plain arrays are read)."""

    def __init__(self, model, rng_mode=0, philox_key=0, dense=False, sample_beta=False, beta_gammas=False):
        L = lib()
        self._keep = dict(
            sp_ptr=np.ascontiguousarray(model.sp_ptr, np.int64),
            sp_next=np.ascontiguousarray(model.sp_next, np.int32),
            sp_cum=np.ascontiguousarray(model.sp_cum, np.float64),
            sp_reward=np.ascontiguousarray(np.where(model.sp_rkind == 0, model.sp_rp0, model.sp_rmean), np.float64),
            sp_seed=np.ascontiguousarray(model.sp_seed, np.int32),
            start_state=np.ascontiguousarray(model.start_states, np.int32),
            start_cum=np.ascontiguousarray(np.cumsum(model.start_probs), np.float64),
        )
        # itertools.accumulate == sequential float64 adds == np.cumsum on float64
        if not model.deterministic_rewards and not sample_beta:
            raise NotImplementedError("Beta rewards: pass sample_beta=True (Philox recipe); the reference-exact "
                                      "stream is colosseum_amd.mdp.reward_sampler")
        k = self._keep
        self.S, self.A, self.H = model.n_states, model.n_actions, model.H
        self._e = L.oracle_env_create(self.S, self.A, self.H, rng_mode, model.rewards_range[0],
                                      model.rewards_range[1], _ptr(k["sp_ptr"]), _ptr(k["sp_next"]),
                                      _ptr(k["sp_cum"]), _ptr(k["sp_reward"]), _ptr(k["sp_seed"]),
                                      len(k["start_state"]), _ptr(k["start_state"]), _ptr(k["start_cum"]),
                                      int(model.start_seed), int(philox_key))
        if sample_beta:
            assert rng_mode == 1
            k.update(r_kind=np.ascontiguousarray(model.sp_rkind, np.uint8), r_p0=np.ascontiguousarray(model.sp_rp0, np.float64),
                     r_p1=np.ascontiguousarray(model.sp_rp1, np.float64))
            L.oracle_env_set_reward_dists(self._e, _ptr(k["r_kind"]), _ptr(k["r_p0"]), _ptr(k["r_p1"]))
            L.oracle_env_set_beta_gammas(self._e, int(bool(beta_gammas)))
        if dense:
            assert rng_mode == 1
            ptr, col, val = _csr64(model.csr())
            k.update(d_ptr=ptr, d_col=col, d_val=val)
            L.oracle_env_set_dense(self._e, _ptr(ptr), _ptr(col), _ptr(val))

    def __del__(self):
        if getattr(self, "_e", None):
            lib().oracle_env_destroy(self._e)
            self._e = None

    def reset(self):
        return int(lib().oracle_env_reset(self._e))

    def step(self, action):
        obs, rew, act = C.c_int32(), C.c_double(), C.c_int32()
        ty = lib().oracle_env_step(self._e, int(action), C.byref(obs), C.byref(rew), C.byref(act))
        if ty < 0:
            raise AssertionError("reset necessary")
        return ty, obs.value, rew.value, act.value

    def rollout(self, n_steps, actions=None, trace=True):
        L = lib()
        n_steps = int(n_steps)
        acts = None if actions is None else np.ascontiguousarray(actions, np.int8)
        tr_obs = np.zeros(n_steps, np.int32) if trace else None
        tr_rew = np.zeros(n_steps, np.float64) if trace else None
        tr_type = np.zeros(n_steps, np.uint8) if trace else None
        last, rsum = C.c_int32(), C.c_double()
        rc = L.oracle_env_rollout(self._e, _ptr(acts), 1, n_steps, C.byref(last), C.byref(rsum), _ptr(tr_obs),
                                  _ptr(tr_rew), _ptr(tr_type), 1)
        if rc != 0:
            raise AssertionError(f"oracle rollout failed ({rc})")
        return dict(last_obs=last.value, reward_sum=rsum.value, obs=tr_obs, rew=tr_rew, stype=tr_type)

    def visits(self):
        vs = np.zeros(self.S, np.int64)
        vsa = np.zeros(self.S * self.A, np.int64)
        lib().oracle_env_visits(self._e, _ptr(vs), _ptr(vsa))
        return vs, vsa.reshape(self.S, self.A)

    def reset_visits(self):
        lib().oracle_env_reset_visits(self._e)

    def state(self):
        cur, h, nr = C.c_int32(), C.c_int32(), C.c_uint8()
        lib().oracle_env_state(self._e, C.byref(cur), C.byref(h), C.byref(nr))
        return cur.value, h.value, bool(nr.value)


def _csr64(model_or_csr):
    ptr, col, val = model_or_csr
    return (np.ascontiguousarray(ptr, np.int64), np.ascontiguousarray(col, np.int32),
            np.ascontiguousarray(val, np.float32))


def vi_discounted(S, A, csr, R, gamma=0.99, eps=1e-3, scheme=0, max_sweeps=1_000_000, max_abs=0.0):
    L = lib()
    ptr, col, val = _csr64(csr)
    if scheme == 0:
        scheme = L.oracle_vi_scheme_rule(S, A, int(ptr[-1]))
    Q = np.zeros(S * A, np.float32)
    V = np.zeros(S, np.float32)
    it = L.oracle_vi_discounted(S, A, ptr, col, val, np.ascontiguousarray(R, np.float32).ravel(), gamma, eps, scheme,
                                max_sweeps, max_abs, Q, V)
    return Q.reshape(S, A), V, int(it), scheme


def pe_discounted(S, A, csr, R, pi, gamma=0.99, eps=1e-7, scheme=0, max_sweeps=1_000_000):
    L = lib()
    ptr, col, val = _csr64(csr)
    if scheme == 0:
        scheme = L.oracle_pe_scheme_rule(S, A, int(ptr[-1]))
    Q = np.zeros(S * A, np.float32)
    V = np.zeros(S, np.float32)
    it = L.oracle_pe_discounted(S, A, ptr, col, val, np.ascontiguousarray(R, np.float32).ravel(),
                                np.ascontiguousarray(pi, np.float32).ravel(), gamma, eps, scheme, max_sweeps, Q, V)
    return Q.reshape(S, A), V, int(it), scheme


def episodic(S, A, H, csr, R, pi=None):
    L = lib()
    ptr, col, val = _csr64(csr)
    Q = np.zeros((H + 1) * S * A, np.float32)
    V = np.zeros((H + 1) * S, np.float32)
    p = None if pi is None else np.ascontiguousarray(pi, np.float32)
    L.oracle_episodic(S, A, H, ptr, col, val, np.ascontiguousarray(R, np.float32).ravel(), _ptr(p), Q, V)
    return Q.reshape(H + 1, S, A), V.reshape(H + 1, S)


def diameter_continuous(S, A, csr, eps=1e-3, scheme=0, max_sweeps=1_000_000):
    L = lib()
    ptr, col, val = _csr64(csr)
    per = np.zeros(S, np.float32)
    d = C.c_float()
    rc = L.oracle_diameter_continuous(S, A, ptr, col, val, eps, scheme, max_sweeps, _ptr(per), C.byref(d))
    if rc != 0:
        raise RuntimeError(f"oracle diameter failed ({rc})")
    return float(d.value), per


def diameter_episodic(model, eps=1e-3, use_running_max=True, max_sweeps=1_000_000):
    """Episodic diameter of a TabularModel: (diameter, per-target values), single-thread reference order."""
    L = lib()
    ptr, col, val = _csr64(model.csr())
    S, A, H = model.n_states, model.n_actions, model.H
    per = np.zeros(S, np.float32)
    d = C.c_double()
    st = np.ascontiguousarray(model.start_states, np.int32)
    sp = np.ascontiguousarray(model.start_probs, np.float32)  # `T_epi[H - 1, :, :, sn] = p` stores float32
    rc = L.oracle_diameter_episodic(S, A, H, ptr, col, val, len(st), st, sp, eps, int(use_running_max), max_sweeps,
                                    _ptr(per), C.byref(d))
    if rc != 0:
        raise RuntimeError(f"oracle episodic diameter failed ({rc})")
    return float(d.value), per


def value_norm(S, A, csr, V):
    ptr, col, val = _csr64(csr)
    return float(lib().oracle_value_norm(S, A, ptr, col, val, np.ascontiguousarray(V, np.float32)))


def python_random(seed, n):
    out = np.zeros(n, np.float64)
    lib().oracle_python_random(int(seed), int(n), out)
    return out


def philox(c, k):
    out = np.zeros(4, np.uint32)
    lib().oracle_philox(*[int(x) for x in c], *[int(x) for x in k], _ptr(out))
    return out


def batch_rollout(tables, b0, b1, n_steps, rng_mode=1, philox_keys=None, want_visits=False):
    """Instances [b0, b1) of concatenated tables (colosseum_amd.batched.tables_from_models layout): reset, then
    n_steps transitions under the Philox random policy.  Returns last_obs, reward_sum[, visits_s, visits_sa]."""
    L = lib()
    t = tables
    A, H = int(t["A"]), int(t["H"])
    f = {k: np.ascontiguousarray(t[k], dt) for k, dt in (
        ("state_off", np.int64), ("sp_ptr", np.int64), ("sp_next", np.int32), ("sp_cum", np.float64),
        ("sp_reward", np.float64), ("sp_seed", np.int32), ("start_off", np.int64), ("start_state", np.int32),
        ("start_cum", np.float64), ("start_seed", np.int32))}
    keys = np.ascontiguousarray(philox_keys if philox_keys is not None else np.arange(int(t["B"])), np.uint64)
    n = b1 - b0
    last = np.zeros(n, np.int32)
    rsum = np.zeros(n, np.float64)
    ns = int(f["state_off"][b1] - f["state_off"][b0])
    vs = np.zeros(ns, np.int64) if want_visits else None
    vsa = np.zeros(ns * A, np.int64) if want_visits else None
    rc = L.oracle_batch_rollout(b0, b1, A, H, rng_mode, float(t["rewards_range"][0]), float(t["rewards_range"][1]),
                                _ptr(f["state_off"]), _ptr(f["sp_ptr"]), _ptr(f["sp_next"]), _ptr(f["sp_cum"]),
                                _ptr(f["sp_reward"]), _ptr(f["sp_seed"]), _ptr(f["start_off"]), _ptr(f["start_state"]),
                                _ptr(f["start_cum"]), _ptr(f["start_seed"]), _ptr(keys), int(n_steps), _ptr(last),
                                _ptr(rsum), _ptr(vs), _ptr(vsa))
    if rc != 0:
        raise RuntimeError(f"oracle batch rollout failed ({rc})")
    return (last, rsum, vs, vsa) if want_visits else (last, rsum)


def batch_vi(tables, b0, b1, gamma=0.99, eps=1e-6, scheme=0, max_sweeps=1_000_000):
    L = lib()
    t = tables
    A = int(t["A"])
    so = np.ascontiguousarray(t["state_off"], np.int64)
    ptr = np.ascontiguousarray(t["csr_ptr"], np.int64)
    col = np.ascontiguousarray(t["csr_col"], np.int32)
    val = np.ascontiguousarray(t["csr_val"], np.float32)
    R = np.ascontiguousarray(t["R"], np.float32)
    Q = np.zeros(int(so[-1]) * A, np.float32)
    V = np.zeros(int(so[-1]), np.float32)
    sw = np.zeros(b1 - b0, np.int64)
    tot = L.oracle_batch_vi(b0, b1, A, _ptr(so), _ptr(ptr), _ptr(col), _ptr(val), _ptr(R), gamma, eps, scheme,
                            max_sweeps, _ptr(Q), _ptr(V), _ptr(sw))
    if tot < 0:
        raise RuntimeError(f"oracle batch VI failed ({tot})")
    return Q, V, sw


def gth(tps):
    """Stationary distribution of a single-recurrent-class chain (float64 GTH, the reference's numba routine)."""
    a = np.array(tps, np.float64, order="C", copy=True)
    n = a.shape[0]
    x = np.zeros(n, np.float64)
    lib().oracle_gth(n, a, x)
    return x


def sparse_diameter_f64(S, A, csr, eps=1e-3, return_log=False):
    """Restatement of the reference's `_get_sparse_diameter` (colosseum/hardness/measures/diameter.py:382-420), the path a
    single-core reference takes for continuous MDPs above 1000 states: targets in index order, float64 expected hitting
    times ET over the states other than the target, Jacobi sweeps
        ET'[s] = min_a ( T[s,a,target] + sum_{j != target, ascending} float64(T[s,a,j]) * (1 + ET[j]) ),
    stop at diff < eps OR (diff < 0.05 and max(ET') - 1 < running maximum) -- order dependent --, running maximum updated
    with max(ET').  Returns (diameter, running maximum after every target)."""
    ptr, col, val = csr
    ptr = np.asarray(ptr, np.int64)
    col = np.asarray(col, np.int64)
    val64 = np.asarray(val, np.float32).astype(np.float64)
    rows = np.repeat(np.arange(S * A), np.diff(ptr))
    diameter = -np.inf
    running = []
    for i in range(S):
        hit = col == i
        Te = np.zeros(S * A)
        Te[rows[hit]] = val64[hit]          # at most one entry per row
        keep = ~hit
        r_k, c_k, v_k = rows[keep], col[keep], val64[keep]
        starts = np.searchsorted(r_k, np.arange(S * A))
        others = np.arange(S) != i
        ET = np.zeros(S)                     # entry i is never read (its column is excluded)
        for _ in range(1_000_000):
            old = ET
            prod = v_k * (1.0 + old[c_k])
            acc = np.zeros(S * A)
            # ascending-j sequential float64 accumulation per row
            pos = np.arange(len(r_k)) - starts[r_k]
            for k in range(int(pos.max()) + 1 if len(pos) else 0):
                sel = pos == k
                acc[r_k[sel]] = acc[r_k[sel]] + prod[sel]
            Q = (Te + acc).reshape(S, A)
            ET = Q.min(1)
            diff = np.abs(old[others] - ET[others]).max()
            mx = ET[others].max()
            if diff < eps or (diff < 0.05 and mx - 1 < diameter):
                break
        diameter = max(diameter, mx)
        running.append(float(diameter))
    return float(diameter), running

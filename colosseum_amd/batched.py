"""BatchedMDP: B independent MDP instances resident on one MI355X, advanced together.

The throughput-side counterpart of the reference's one-environment-per-process fan-out
(colosseum/experiment/experiment_instances.py:160-166): instances are (parameterisation x seed) pairs,
laid out back to back in HBM; every method is one call through the C ABI of libcmdp.so."""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L
from .mdp.builder import TabularModel


_ENV_FIELDS = (("sp_ptr", np.int64), ("sp_next", np.int32), ("sp_cum", np.float64), ("sp_reward", np.float64),
               ("sp_rkind", np.uint8), ("sp_rp0", np.float64), ("sp_rp1", np.float64), ("sp_seed", np.int32), ("start_off", np.int64), ("start_state", np.int32),
               ("start_cum", np.float64), ("start_seed", np.int32))
_DP_FIELDS = (("csr_ptr", np.int64), ("csr_col", np.int32), ("csr_val", np.float32), ("R", np.float32))


def tables_from_models(models: Sequence[TabularModel], with_env: bool = True, with_dp: bool = True) -> dict:
    """Concatenates per-instance TabularModels into the flat arrays of `cmdp_desc` (include/cmdp.h)."""
    A, H = models[0].n_actions, models[0].H
    rr = tuple(models[0].rewards_range)
    for m in models:
        if m.n_actions != A or m.H != H or tuple(m.rewards_range) != rr:
            raise ValueError("all instances of a batch share n_actions, the horizon H and the rewards range")
    n_states = np.array([m.n_states for m in models], np.int64)
    t = dict(B=len(models), A=A, H=H, rewards_range=rr,
             state_off=np.concatenate([[0], np.cumsum(n_states)]).astype(np.int64))
    if with_env:
        ent = np.array([len(m.sp_next) for m in models], np.int64)
        ent_off = np.concatenate([[0], np.cumsum(ent)])
        t["sp_ptr"] = np.concatenate([m.sp_ptr[:-1] + ent_off[i] for i, m in enumerate(models)] + [ent_off[-1:]])
        t["sp_next"] = np.concatenate([m.sp_next for m in models])
        t["sp_cum"] = np.concatenate([m.sp_cum for m in models])
        # the deterministic value (loc); for stochastic kinds the distribution mean (only accepted by the library
        # under CMDP_FLAG_REWARD_MEANS)
        t["sp_reward"] = np.concatenate([np.where(m.sp_rkind == 0, m.sp_rp0, m.sp_rmean) for m in models])
        t["sp_rkind"] = np.concatenate([m.sp_rkind for m in models])
        t["sp_rp0"] = np.concatenate([m.sp_rp0 for m in models])
        t["sp_rp1"] = np.concatenate([m.sp_rp1 for m in models])
        t["sp_seed"] = np.concatenate([m.sp_seed for m in models])
        ns = np.array([len(m.start_states) for m in models], np.int64)
        t["start_off"] = np.concatenate([[0], np.cumsum(ns)])
        t["start_state"] = np.concatenate([m.start_states for m in models])
        # itertools.accumulate == sequential float64 adds == np.cumsum
        t["start_cum"] = np.concatenate([np.cumsum(m.start_probs) for m in models])
        t["start_seed"] = np.array([max(m.start_seed, 0) for m in models], np.int32)
    if with_dp:
        csrs = [m.csr() for m in models]
        nz = np.array([len(c[1]) for c in csrs], np.int64)
        nz_off = np.concatenate([[0], np.cumsum(nz)])
        t["csr_ptr"] = np.concatenate([c[0][:-1].astype(np.int64) + nz_off[i] for i, c in enumerate(csrs)]
                                      + [nz_off[-1:]])
        t["csr_col"] = np.concatenate([c[1] for c in csrs])
        t["csr_val"] = np.concatenate([c[2] for c in csrs])
        t["R"] = np.concatenate([m.reward_matrix().ravel() for m in models])
    return t


class BatchedMDP:
    def __init__(self, models: Optional[Sequence[TabularModel]] = None, rng_mode: int = L.RNG_MT_COMPAT,
                 philox_keys: Optional[Sequence[int]] = None, with_env: bool = True, with_dp: bool = True,
                 tables: Optional[dict] = None, layout: int = L.LAYOUT_CSR, flags: int = 0):
        """Either `models` (TabularModel per instance) or pre-concatenated `tables` (see `tables_from_models`
        and colosseum_amd.mdp.fast_batch) describe the batch."""
        lib = L.load()
        if tables is None:
            models = list(models)
            assert len(models) > 0
            tables = tables_from_models(models, with_env, with_dp)
        self.models = models
        self.B, self.A, self.H = int(tables["B"]), int(tables["A"]), int(tables["H"])
        A = self.A
        rr = tables["rewards_range"]
        self.state_off = np.ascontiguousarray(tables["state_off"], np.int64)
        self.n_states = np.diff(self.state_off)
        self.row_off = self.state_off * A
        self.rng_mode = rng_mode
        keep = {}
        d = L.CmdpDesc()
        d.n_instances, d.n_actions, d.horizon, d.rng_mode, d.layout = self.B, A, self.H, rng_mode, layout
        d.flags = int(flags)
        d.reward_min, d.reward_max = float(rr[0]), float(rr[1])
        keep["state_off"] = self.state_off
        if "sp_ptr" in tables:
            for k, dt in _ENV_FIELDS:
                if k in ("sp_rp0", "sp_rp1") and k not in tables:
                    continue  # optional: only read for Beta entries
                keep[k] = np.ascontiguousarray(tables[k], dt)
            if philox_keys is None:
                philox_keys = np.arange(self.B, dtype=np.uint64)
            keep["philox_key"] = np.ascontiguousarray(philox_keys, np.uint64)
            assert len(keep["philox_key"]) == self.B
        if "csr_ptr" in tables:
            for k, dt in _DP_FIELDS:
                keep[k] = np.ascontiguousarray(tables[k], dt)
        for k, v in keep.items():
            setattr(d, k, L.ptr(v))
        self._keep = keep
        self._h = C.c_void_p()
        L.check(lib.cmdp_create(C.byref(self._h), C.byref(d)))
        self._lib = lib
        self.flags = int(flags)
        if flags & L.FLAG_REWARD_CACHE and models is not None:
            self.set_reward_streams([m.extra["rng_state"] for m in models])

    def set_reward_streams(self, states):
        """CMDP_FLAG_REWARD_CACHE: the numpy stream `BaseMDP._rng` of every instance as `RandomState.get_state()` gives it
        after the MDP's construction (`TabularModel.extra["rng_state"]`); the library continues these streams whenever it
        fills a cache of 5000 reward samples (colosseum/mdp/base.py:1196-1203)."""
        assert len(states) == self.B
        key = np.ascontiguousarray(np.stack([np.asarray(s[1], np.uint32) for s in states]))
        pos = np.ascontiguousarray([int(s[2]) for s in states], np.int32)
        has = np.ascontiguousarray([int(s[3]) for s in states], np.int32)
        cached = np.ascontiguousarray([float(s[4]) for s in states], np.float64)
        L.check(self._lib.cmdp_set_reward_streams(self._h, L.ptr(key), L.ptr(pos), L.ptr(has), L.ptr(cached)))

    def reward_cache_stats(self) -> dict:
        """Blocks of 5000 samples drawn so far and park / fill / relaunch rounds (CMDP_FLAG_REWARD_CACHE)."""
        v = C.c_double()
        out = {}
        for k, w in (("fills", L.STAT_REWARD_FILLS), ("rounds", L.STAT_REWARD_ROUNDS), ("fill_ms", L.STAT_REWARD_FILL_MS),
                     ("round_ms", L.STAT_REWARD_ROUND_MS)):
            L.check(self._lib.cmdp_stat(self._h, w, C.byref(v)))
            out[k] = int(v.value) if k in ("fills", "rounds") else float(v.value)
        return out

    @property
    def handle(self):
        """The cmdp_t* (for C-ABI calls this class has no method for, e.g. cmdp_stat)."""
        return self._h

    # -- life cycle --------------------------------------------------------------------------------------
    def _register_agent(self, agent):
        import weakref

        if not hasattr(self, "_agents"):
            self._agents = []
        self._agents.append(weakref.ref(agent))

    def close(self):
        # agents hold device state tied to this handle's stream: they go first
        for ref in getattr(self, "_agents", []):
            a = ref()
            if a is not None:
                a.close()
        self._agents = []
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.cmdp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- interaction ------------------------------------------------------------------------------------
    def reset(self, mask=None) -> np.ndarray:
        obs = np.zeros(self.B, np.int32)
        m = L.carr(mask, np.uint8)
        L.check(self._lib.cmdp_reset(self._h, L.ptr(m), L.ptr(obs)))
        return obs

    def step(self, actions, auto_reset: bool = False):
        a = L.carr(np.broadcast_to(np.asarray(actions), (self.B,)), np.int32)
        obs = np.zeros(self.B, np.int32)
        rew = np.zeros(self.B, np.float64)
        st = np.zeros(self.B, np.uint8)
        L.check(self._lib.cmdp_step(self._h, L.ptr(a), int(auto_reset), L.ptr(obs), L.ptr(rew), L.ptr(st)))
        return obs, rew, st

    def rollout(self, n_steps: int, actions=None, trace: bool = False, greedy_q=None):
        """n_steps transitions per instance (episodic terminations are followed by reset()).
        actions: None -> on-device uniform random policy; else int array [n_steps, B].  greedy_q: per-instance Q
        tables ([S, A], or [H, S, A] for episodic handles): the fixed policy "first maximiser of the current row"."""
        n_steps = int(n_steps)
        last = np.zeros(self.B, np.int32)
        rsum = np.zeros(self.B, np.float64)
        tr_obs = np.zeros((n_steps, self.B), np.int32) if trace else None
        tr_rew = np.zeros((n_steps, self.B), np.float64) if trace else None
        tr_ty = np.zeros((n_steps, self.B), np.uint8) if trace else None
        if greedy_q is not None:
            assert actions is None
            policy, arg = L.POLICY_GREEDY_Q, np.ascontiguousarray(self._flat_rows(greedy_q, lead=max(self.H, 1)), np.float32)
        elif actions is None:
            policy, arg = L.POLICY_RANDOM, None
        else:
            arg = L.carr(actions, np.int8)
            assert arg.shape == (n_steps, self.B), f"actions must be [n_steps, B], got {arg.shape}"
            policy = L.POLICY_HOST_ACTIONS
        L.check(self._lib.cmdp_rollout(self._h, policy, L.ptr(arg), n_steps, L.ptr(last), L.ptr(rsum), L.ptr(tr_obs),
                                       L.ptr(tr_rew), L.ptr(tr_ty)))
        out = dict(last_obs=last, reward_sum=rsum)
        if trace:
            out.update(obs=tr_obs, rew=tr_rew, stype=tr_ty)
        return out

    def rollout_async(self, n_steps: int):
        L.check(self._lib.cmdp_rollout_async(self._h, L.POLICY_RANDOM, int(n_steps)))

    def set_rollout_kernel(self, which: int):
        """L.ROLLOUT_AUTO | L.ROLLOUT_GLOBAL | L.ROLLOUT_LDS (tuning only; results are identical)."""
        L.check(self._lib.cmdp_set_option(self._h, L.OPT_ROLLOUT_KERNEL, int(which)))

    def lds_plan(self) -> dict:
        """The LDS-resident rollout chosen for this batch (cmdp_lds_plan): eligible, shared-table pipeline (K1T), pipeline (K1P),
        fused walker (K1L) or stochastic-dynamics kernel (K1S), instances per workgroup, transitions per chunk."""
        plan = np.zeros(4, np.int32)
        L.check(self._lib.cmdp_lds_plan(self._h, L.ptr(plan)))
        return dict(eligible=bool(plan[0]), kernel={0: "k_rollout_lds", 1: "k_rollout_pipe", 2: "k_rollout_stoch", 3: "k_rollout_tmpl",
                                                       4: "k_rollout_tmpl_stream", 5: "k_rollout_epi"}[int(plan[1])],
                    instances_per_workgroup=int(plan[2]), chunk=int(plan[3]))

    def set_option(self, option: int, value: int):
        L.check(self._lib.cmdp_set_option(self._h, int(option), int(value)))

    def set_dp_kernel(self, which: int):
        """L.DP_AUTO | L.DP_WORKGROUP | L.DP_REGISTER for the Jacobi sweeps (tuning only)."""
        L.check(self._lib.cmdp_set_option(self._h, L.OPT_DP_KERNEL, int(which)))

    def synchronize(self):
        L.check(self._lib.cmdp_synchronize(self._h))

    @property
    def stream(self) -> int:
        return int(self._lib.cmdp_stream(self._h) or 0)

    def visits(self, state: bool = True, sa: bool = True):
        """(state counts [sum S], state-action counts [sum S*A]); split with `split_states`/`split_rows`."""
        vs = np.zeros(int(self.state_off[-1]), np.int64) if state else None
        vsa = np.zeros(int(self.row_off[-1]), np.int64) if sa else None
        L.check(self._lib.cmdp_visits(self._h, L.ptr(vs), L.ptr(vsa)))
        return vs, vsa

    def reset_visits(self):
        L.check(self._lib.cmdp_reset_visits(self._h))

    def set_visits(self, state_counts=None, sa_counts=None):
        """Restores the visit counters (cmdp_set_visits): int64 arrays as `visits()` returns them; None leaves one as it is."""
        vs = None if state_counts is None else np.ascontiguousarray(state_counts, np.int64)
        vsa = None if sa_counts is None else np.ascontiguousarray(sa_counts, np.int64)
        L.check(self._lib.cmdp_set_visits(self._h, L.ptr(vs) if vs is not None else None, L.ptr(vsa) if vsa is not None else None))

    def state(self):
        cur = np.zeros(self.B, np.int32)
        h = np.zeros(self.B, np.int32)
        nr = np.zeros(self.B, np.uint8)
        L.check(self._lib.cmdp_state(self._h, L.ptr(cur), L.ptr(h), L.ptr(nr)))
        return cur, h, nr.astype(bool)

    def last_start(self) -> np.ndarray:
        """State index sampled by the latest reset() of every instance (BaseMDP.last_starting_node)."""
        out = np.zeros(self.B, np.int32)
        prev = np.zeros(self.B, np.int32)
        L.check(self._lib.cmdp_last_start(self._h, L.ptr(out), L.ptr(prev)))
        self.previous_start = prev
        return out

    def set_observation_table(self, tables):
        """Per-instance feature tables ([S_b, F], or [H, S_b, F] for episodic handles -- `EmissionMap.all_observations`,
        see colosseum_amd.emission_maps.observation_table)."""
        tables = [np.ascontiguousarray(t, np.float32) for t in tables]
        assert len(tables) == self.B
        timed = tables[0].ndim == 3
        F = tables[0].shape[-1]
        flat = np.ascontiguousarray(np.concatenate([t.reshape(-1) for t in tables]), np.float32)
        L.check(self._lib.cmdp_set_observation_table(self._h, L.ptr(flat), int(F), int(timed)))
        self._obs_F = F

    def observe(self, noise_scale: float = 0.0) -> np.ndarray:
        """Observations [B, F] of the current states (zeros after an episodic horizon); noise_scale > 0 adds Philox
        Gaussian noise on the device (throughput mode)."""
        out = np.zeros((self.B, self._obs_F), np.float32)
        L.check(self._lib.cmdp_observe(self._h, float(noise_scale), L.ptr(out)))
        return out

    def observe_noise(self, kind: str, scale: float = 0.1, df: float = 3.0, covariance=None) -> np.ndarray:
        """Observations with one of the reference's four noise classes sampled on the device (throughput mode):
        kind in "GaussianUncorrelated" (scale), "GaussianCorrelated" (covariance [F, F]), "StudentTUncorrelated" (df),
        "StudentTCorrelated" (covariance = shape matrix, df)."""
        code = {"GaussianUncorrelated": L.NOISE_GAUSSIAN, "GaussianCorrelated": L.NOISE_GAUSSIAN_CORRELATED,
                "StudentTUncorrelated": L.NOISE_STUDENT_T, "StudentTCorrelated": L.NOISE_STUDENT_T_CORRELATED}[kind]
        chol = None
        if covariance is not None:
            chol = np.ascontiguousarray(np.linalg.cholesky(np.asarray(covariance, np.float64)), np.float32)
            assert chol.shape == (self._obs_F, self._obs_F)
        out = np.zeros((self.B, self._obs_F), np.float32)
        L.check(self._lib.cmdp_observe_noise(self._h, code, float(scale), float(df), L.ptr(chol), L.ptr(out)))
        return out

    def average_reward(self, actions, start_states, mask=None):
        """`get_average_reward(T, R, one_hot(actions), [(start, 1.0)])` for every (continuous) instance on the device
        (kernel K9).  actions: per-instance arrays [S_b] (or one flat array); returns (values, n_recurrent_classes) with
        values a list of numpy scalars typed as the reference's results (np.float32 / np.float64)."""
        acts = np.ascontiguousarray(np.concatenate([np.asarray(a).ravel() for a in actions])
                                    if isinstance(actions, (list, tuple)) else actions, np.int32)
        assert acts.size == self.state_off[-1]
        st = np.ascontiguousarray(start_states, np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        avg = np.zeros(self.B, np.float64)
        kind = np.zeros(self.B, np.int32)
        ncls = np.zeros(self.B, np.int32)
        L.check(self._lib.cmdp_average_reward(self._h, L.ptr(acts), L.ptr(st), L.ptr(m), L.ptr(avg), L.ptr(kind), L.ptr(ncls)))
        return [np.float32(avg[b]) if kind[b] else np.float64(avg[b]) for b in range(self.B)], ncls

    # -- helpers ------------------------------------------------------------------------------------------
    def split_states(self, flat, lead: int = 1) -> List[np.ndarray]:
        return [flat[lead * self.state_off[b]: lead * self.state_off[b + 1]] for b in range(self.B)]

    def split_rows(self, flat, lead: int = 1) -> List[np.ndarray]:
        return [flat[lead * self.row_off[b]: lead * self.row_off[b + 1]] for b in range(self.B)]

    def _flat_rows(self, per_instance, lead: int = 1):
        if per_instance is None:
            return None
        if isinstance(per_instance, np.ndarray) and per_instance.ndim == 1 and per_instance.size == lead * self.row_off[-1]:
            return L.carr(per_instance, np.float32)
        return L.carr(np.concatenate([np.asarray(x, np.float32).ravel() for x in per_instance]), np.float32)

    # -- dynamic programming ----------------------------------------------------------------------------------
    def dp_buffers(self, pinned=True):
        """(Q, V, sweeps) result buffers to pass as `out=` to value_iteration / policy_evaluation when they are called
        repeatedly: page-locked by default (one DMA per array, no page faults inside the copy)."""
        mk = L.pinned_empty if pinned else np.empty
        return (mk(int(self.row_off[-1]), np.float32), mk(int(self.state_off[-1]), np.float32), mk(self.B, np.int64))

    def value_iteration(self, gamma=0.99, epsilon=1e-3, scheme=L.SCHEME_AUTO, max_sweeps=1_000_000,
                        max_abs_value=None, R=None, out=None):
        if out is not None:
            Q, V, sw = out
            assert Q.dtype == np.float32 and V.dtype == np.float32 and sw.dtype == np.int64
            assert len(Q) == int(self.row_off[-1]) and len(V) == int(self.state_off[-1]) and len(sw) == self.B
        else:
            Q = np.zeros(int(self.row_off[-1]), np.float32)
            V = np.zeros(int(self.state_off[-1]), np.float32)
            sw = np.zeros(self.B, np.int64)
        Rov = self._flat_rows(R)
        L.check(self._lib.cmdp_vi_discounted(self._h, gamma, epsilon, scheme, max_sweeps,
                                             0.0 if max_abs_value is None else float(max_abs_value), L.ptr(Rov),
                                             L.ptr(Q), L.ptr(V), L.ptr(sw)))
        return Q, V, sw

    def policy_evaluation(self, pi, gamma=0.99, epsilon=1e-7, scheme=L.SCHEME_AUTO, max_sweeps=1_000_000, R=None,
                          out=None):
        if out is not None:
            Q, V, sw = out
            assert Q.dtype == np.float32 and V.dtype == np.float32 and sw.dtype == np.int64
            assert len(Q) == int(self.row_off[-1]) and len(V) == int(self.state_off[-1]) and len(sw) == self.B
        else:
            Q = np.zeros(int(self.row_off[-1]), np.float32)
            V = np.zeros(int(self.state_off[-1]), np.float32)
            sw = np.zeros(self.B, np.int64)
        p = self._flat_rows(pi)
        Rov = self._flat_rows(R)
        L.check(self._lib.cmdp_pe_discounted(self._h, L.ptr(p), gamma, epsilon, scheme, max_sweeps, L.ptr(Rov),
                                             L.ptr(Q), L.ptr(V), L.ptr(sw)))
        return Q, V, sw

    def episodic_value_iteration(self, H=None, R=None):
        H = self.H if H is None else int(H)
        Q = np.zeros((H + 1) * int(self.row_off[-1]), np.float32)
        V = np.zeros((H + 1) * int(self.state_off[-1]), np.float32)
        Rov = self._flat_rows(R)
        L.check(self._lib.cmdp_vi_episodic(self._h, H, L.ptr(Rov), L.ptr(Q), L.ptr(V)))
        return Q, V

    def episodic_policy_evaluation(self, pi, H=None, R=None):
        H = self.H if H is None else int(H)
        Q = np.zeros((H + 1) * int(self.row_off[-1]), np.float32)
        V = np.zeros((H + 1) * int(self.state_off[-1]), np.float32)
        p = self._flat_rows(pi, lead=H)
        Rov = self._flat_rows(R)
        L.check(self._lib.cmdp_pe_episodic(self._h, H, L.ptr(p), L.ptr(Rov), L.ptr(Q), L.ptr(V)))
        return Q, V

    def greedy_policy_episodic(self, Q, q_layers: int, H=None):
        """argmax_3d with the reference's RandomState(42) tie-break on the device: Q per instance [q_layers, S, A]
        (flat concatenation or list) -> one-hot policies, flat [B: H*S*A]."""
        H = self.H if H is None else int(H)
        q = self._flat_rows(Q, lead=q_layers)
        pi = np.zeros(H * int(self.row_off[-1]), np.float32)
        L.check(self._lib.cmdp_greedy_policy_episodic(self._h, H, int(q_layers), L.ptr(q), L.ptr(pi)))
        return pi

    def diameter(self, epsilon=1e-3, scheme=L.SCHEME_AUTO, max_sweeps=1_000_000):
        per = np.zeros(int(self.state_off[-1]), np.float32)
        diam = np.zeros(self.B, np.float32)
        L.check(self._lib.cmdp_diameter(self._h, epsilon, scheme, max_sweeps, L.ptr(per), L.ptr(diam)))
        return diam, per

    def mixing_time(self, stationary, policy=None, threshold: float = 0.25, max_steps: int = 1_000_000):
        """Build-defined (the reference has none): smallest t with max_s TV(P^t(s, .), stationary) <= threshold for the
        chain of `policy` (per-instance [S_b, A] arrays; None = uniform).  stationary: per-instance float64 vectors.
        Returns (t_mix int64 [B] (-1: not reached within max_steps), total variation at that step)."""
        st = np.ascontiguousarray(np.concatenate([np.asarray(x, np.float64).ravel() for x in stationary]))
        assert st.size == self.state_off[-1]
        pi = None if policy is None else self._flat_rows(policy)
        t = np.zeros(self.B, np.int64)
        tv = np.zeros(self.B, np.float64)
        L.check(self._lib.cmdp_mixing_time(self._h, L.ptr(pi), L.ptr(st), float(threshold), int(max_steps), L.ptr(t), L.ptr(tv)))
        return t, tv

    def diameter_sparse_f64(self, epsilon=1e-3, max_sweeps=1_000_000):
        """`_get_sparse_diameter` (the single-core reference's path above 1000 states): float64, sequential targets with
        the running-maximum early exit.  Returns (diameter float64 [B], running maximum after every target, flat)."""
        run = np.zeros(int(self.state_off[-1]), np.float64)
        diam = np.zeros(self.B, np.float64)
        L.check(self._lib.cmdp_diameter_sparse_f64(self._h, float(epsilon), int(max_sweeps), L.ptr(run), L.ptr(diam)))
        return diam, run

    def diameter_range(self, target_lo: int, target_hi: int, epsilon=1e-3, max_sweeps=1_000_000) -> np.ndarray:
        """Optimal expected hitting times (max over start states) of the targets [target_lo, target_hi) of the flat
        state space, Jacobi scheme, 64 targets per workgroup (kernel K5S): the shard of `diameter()` one GPU takes
        when one large MDP is split over ranks (config C5)."""
        per = np.zeros(max(0, int(target_hi) - int(target_lo)), np.float32)  # a reversed range is refused by the library
        L.check(self._lib.cmdp_diameter_range(self._h, float(epsilon), int(max_sweeps), int(target_lo), int(target_hi),
                                              L.ptr(per)))
        return per

    def diameter_episodic(self, epsilon=1e-3, max_sweeps=1_000_000):
        """Episodic diameter of every instance (needs the batch to have been built from TabularModels or tables
        that carry the starting states)."""
        k = self._keep
        if "start_off" in k:
            soff, sst = k["start_off"], k["start_state"]
            cum = k["start_cum"]
            prob = np.diff(np.concatenate([[0.0], cum]))
            prob[soff[:-1]] = cum[soff[:-1]]  # first entry of every instance
        else:
            models = self.models
            ns = np.array([len(m.start_states) for m in models], np.int64)
            soff = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
            sst = np.concatenate([m.start_states for m in models]).astype(np.int32)
            prob = np.concatenate([m.start_probs for m in models])
        if self.models is not None:  # exact probabilities rather than differences of accumulated ones
            prob = np.concatenate([m.start_probs for m in self.models])
        soff = np.ascontiguousarray(soff, np.int64)
        sst = np.ascontiguousarray(sst, np.int32)
        prob = np.ascontiguousarray(prob, np.float32)
        per = np.zeros(int(self.state_off[-1]), np.float32)
        diam = np.zeros(self.B, np.float32)
        L.check(self._lib.cmdp_diameter_episodic(self._h, self.H, L.ptr(soff), L.ptr(sst), L.ptr(prob), epsilon,
                                                 max_sweeps, L.ptr(per), L.ptr(diam)))
        return diam, per

    def value_norm(self, V):
        v = L.carr(V, np.float32)
        assert v.size == self.state_off[-1]
        out = np.zeros(self.B, np.float32)
        L.check(self._lib.cmdp_value_norm(self._h, L.ptr(v), L.ptr(out)))
        return out

"""Stochastic (Beta) rewards: the host sampler that continues the MDP's numpy stream reproduces the reference's
`sample_reward` bit for bit (golden G8: 7000-step trajectories, 5000-sample caches per visited triple incl. refills,
the `r*(max-min) - min` rescale with a non-default rewards range)."""
import numpy as np
import pytest

from conftest import load_golden
from colosseum_amd.mdp import make_model
from colosseum_amd.mdp.reward_sampler import CompatRewardSampler


def _kwargs(c):
    kw = dict(c["kwargs"])
    if "rewards_range" in kw:
        kw["rewards_range"] = tuple(kw["rewards_range"])
    return kw


def test_structure_with_stochastic_rewards():
    z, cases = load_golden("G8_stochastic_rewards")
    for i, c in enumerate(cases):
        m = make_model(c["cls"], **_kwargs(c))
        k = f"c{i}_"
        assert not m.deterministic_rewards
        np.testing.assert_array_equal(m.nodes, z[k + "nodes"])  # other action-map draw order than with deterministic rewards
        np.testing.assert_array_equal(m.sp_next, z[k + "sp_next"])
        np.testing.assert_array_equal(m.sp_prob, z[k + "sp_prob"])
        np.testing.assert_array_equal(m.sp_rmean, z[k + "sp_rmean"])
        np.testing.assert_array_equal(m.reward_matrix(), z[k + "R"])


@pytest.mark.parametrize("name", ["G8_stochastic_rewards", "G12_families"])
def test_host_reward_sampler_matches_reference(name):
    z, cases = load_golden(name)
    n_checked = 0
    for i, c in enumerate(cases):
        m = make_model(c["cls"], **_kwargs(c))
        if m.deterministic_rewards:
            continue
        n_checked += 1
        k = f"c{i}_"
        rs = CompatRewardSampler(m)
        acts, states, stype, resets = z[k + "actions"], z[k + "state"], z[k + "stype"], z[k + "resets"]
        cur, ri = int(resets[0]), 1
        got = np.zeros(len(acts))
        for t in range(len(acts)):
            nxt = int(states[t])
            got[t] = rs.sample(cur, int(acts[t]), nxt)
            cur = nxt
            if stype[t] == 2:
                cur = int(resets[ri])
                ri += 1
        np.testing.assert_array_equal(got, z[k + "rew"], err_msg=str(c))
    assert n_checked >= 1


@pytest.mark.gpu
def test_gpu_mdp_with_stochastic_rewards(need_gpu):
    """End to end through GpuMDP.step: transitions (per-(s,a) MT19937 streams) on the device, Beta rewards on the
    host; observations, rewards and step types equal the reference's."""
    from colosseum_amd.mdp import gpu_mdp

    z, cases = load_golden("G8_stochastic_rewards")
    for i, c in enumerate(cases):
        k = f"c{i}_"
        mdp = getattr(gpu_mdp, c["cls"])(**_kwargs(c))
        ts = mdp.reset()
        assert ts.observation == z[k + "resets"][0]
        ri = 1
        n = 2500
        for t in range(n):
            ts = mdp.step(int(z[k + "actions"][t]))
            assert ts.observation == z[k + "obs"][t] and int(ts.step_type) == z[k + "stype"][t], (c, t)
            assert ts.reward == z[k + "rew"][t], (c, t)
            if ts.last():
                assert mdp.reset().observation == z[k + "resets"][ri]
                ri += 1
        mdp.close()


def _single_state_tables(B, a, b):
    """B copies of a one-state, one-action MDP whose only transition pays a Beta(a, b) reward."""
    return dict(
        B=B, A=1, H=0, rewards_range=(0.0, 1.0),
        state_off=np.arange(B + 1, dtype=np.int64),
        sp_ptr=np.arange(B + 1, dtype=np.int64), sp_next=np.zeros(B, np.int32), sp_cum=np.ones(B),
        sp_reward=np.full(B, a / (a + b)), sp_rkind=np.ones(B, np.uint8), sp_rp0=np.full(B, float(a)),
        sp_rp1=np.full(B, float(b)), sp_seed=np.zeros(B, np.int32),
        start_off=np.arange(B + 1, dtype=np.int64), start_state=np.zeros(B, np.int32), start_cum=np.ones(B),
        start_seed=np.zeros(B, np.int32),
    )


@pytest.mark.gpu
def test_device_beta_rewards_philox(need_gpu):
    """Throughput mode: Beta rewards sampled on the device (Philox domain 3).  (i) GPU == oracle recipe to rounding of
    the elementary functions, states bit-equal; (ii) the samples are Beta(a, b): Kolmogorov-Smirnov against scipy."""
    from scipy import stats

    from colosseum_amd import _lib as L
    from colosseum_amd.batched import BatchedMDP
    from oracle import oracle as O

    for cls, kw in [("DeepSeaEpisodic", dict(seed=2, size=6, p_rand=0.3, make_reward_stochastic=True)),
                    ("FrozenLakeContinuous", dict(seed=1, size=5, p_frozen=0.9, p_lazy=0.05, make_reward_stochastic=True,
                                                  reward_variance_multiplier=0.3))]:
        m = make_model(cls, **kw)
        # both recipes: inverse CDF for unit shapes (default), two gammas for every shape (CMDP_FLAG_BETA_GAMMAS, rounds 1-2)
        for gammas in (False, True):
            env = BatchedMDP([m, m], rng_mode=L.RNG_PHILOX, philox_keys=[7, 8], with_dp=False, flags=L.FLAG_BETA_GAMMAS if gammas else 0)
            env.reset()
            out = env.rollout(3000, None, trace=True)
            for i in range(2):
                e = O.OracleEnv(m, rng_mode=1, philox_key=7 + i, sample_beta=True, beta_gammas=gammas)
                e.reset()
                ref = e.rollout(3000)
                np.testing.assert_array_equal(out["obs"][:, i], ref["obs"])
                np.testing.assert_allclose(out["rew"][:, i], ref["rew"], rtol=1e-12, atol=1e-300)
                assert out["reward_sum"][i] == pytest.approx(ref["reward_sum"], rel=1e-12)
            assert 0.0 < out["rew"].min() and out["rew"].max() < 1.0 and np.unique(out["rew"]).size > 5000
            env.close()
        with pytest.raises(L.CmdpError):  # reference-exact Beta sampling is host side: MT_COMPAT refuses to fake it
            BatchedMDP([m], rng_mode=L.RNG_MT_COMPAT, with_dp=False)

    for a, b, fl in ((0.5, 0.11, 0), (2.0, 4.0, 0), (24.0, 1.0, 0), (1.0, 249.0, 0), (0.3, 7.5, 0), (1.0, 1.0, 0),
                     (24.0, 1.0, L.FLAG_BETA_GAMMAS), (1.0, 249.0, L.FLAG_BETA_GAMMAS)):
        B = 64
        env = BatchedMDP(tables=_single_state_tables(B, a, b), rng_mode=L.RNG_PHILOX,
                         philox_keys=np.arange(1000, 1000 + B, dtype=np.uint64), flags=fl)
        env.reset()
        x = env.rollout(2000, None, trace=True)["rew"].ravel()
        env.close()
        d = stats.beta(a, b)
        assert abs(x.mean() - d.mean()) < 5 * d.std() / np.sqrt(x.size), (a, b)
        assert abs(x.var() - d.var()) < 0.05 * d.var() + 1e-9, (a, b)
        # 20 equiprobable bins (chi-square): robust to the atoms at 1.0 / 0.0 that float64 rounding of
        # Ga / (Ga + Gb) creates for tiny shapes -- Beta(0.5, 0.11) puts 1.5 % of its mass within 1e-16 of 1
        edges = d.ppf(np.linspace(0, 1, 21))
        edges[0], edges[-1] = -1.0, 2.0
        counts = np.histogram(x, edges)[0]
        chi2 = stats.chisquare(counts)
        assert chi2.pvalue > 1e-4, (a, b, counts.tolist(), chi2)

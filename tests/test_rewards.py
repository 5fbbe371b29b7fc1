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


def test_host_reward_sampler_matches_reference():
    z, cases = load_golden("G8_stochastic_rewards")
    for i, c in enumerate(cases):
        m = make_model(c["cls"], **_kwargs(c))
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

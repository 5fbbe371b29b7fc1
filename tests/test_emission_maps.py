"""Emission maps and observation noise (golden G14, generated from the reference): feature tables, the reference-exact
observation stream through GpuMDP (noise = numpy's RandomState stream, incl. the samples `observation_spec()` draws at
episode ends), and the device gather / Philox noise of BatchedMDP.observe."""
import numpy as np
import pytest

from conftest import load_golden
from colosseum_amd.emission_maps import CompatNoise, observation_table
from colosseum_amd.mdp import make_model


def test_observation_tables_match_reference():
    z, cases = load_golden("G14_emission_maps")
    for i, c in enumerate(cases):
        m = make_model(c["cls"], **c["kwargs"])
        tab = observation_table(m, c["emission_map"])
        np.testing.assert_array_equal(tab, z[f"c{i}_all_observations"], err_msg=str(c))
    assert observation_table(m, "Tabular") is None
    with pytest.raises(NotImplementedError):
        observation_table(m, "ImageEncoding")


def test_compat_noise_is_numpys_stream():
    n = CompatNoise(5, (3,), scale=0.25)
    ref = np.random.RandomState(5).normal(0, 0.25, (5000, 3)).astype(np.float32)
    got = np.stack([next(n) for _ in range(40)])
    np.testing.assert_array_equal(got, ref[:40])


@pytest.mark.gpu
def test_gpu_mdp_observation_stream_matches_reference(need_gpu):
    from colosseum_amd.mdp import gpu_mdp

    z, cases = load_golden("G14_emission_maps")
    for i, c in enumerate(cases):
        k = f"c{i}_"
        extra = dict(emission_map=c["emission_map"])
        if c["noise_scale"] is not None:
            extra.update(noise="GaussianUncorrelated", noise_kwargs=dict(scale=c["noise_scale"]))
        mdp = getattr(gpu_mdp, c["cls"])(**c["kwargs"], **extra)
        ts = mdp.reset()
        np.testing.assert_array_equal(ts.observation, z[k + "reset_obs"][0])
        ri = 1
        for t, a in enumerate(z[k + "actions"]):
            ts = mdp.step(int(a))
            np.testing.assert_array_equal(ts.observation, z[k + "obs"][t], err_msg=f"{c} step {t}")
            assert int(ts.step_type) == z[k + "stype"][t]
            if ts.last():
                np.testing.assert_array_equal(mdp.reset().observation, z[k + "reset_obs"][ri])
                ri += 1
        mdp.close()


@pytest.mark.gpu
def test_device_observe_gathers_rows_and_adds_philox_noise(need_gpu):
    from colosseum_amd import _lib as L
    from colosseum_amd.batched import BatchedMDP
    from oracle import oracle as O

    ms = [make_model("MiniGridEmptyEpisodic", seed=s, size=4, n_starting_states=2) for s in range(3)]
    tabs = [observation_table(m, "StateInfo") for m in ms]
    keys = np.array([7, 8, 9], np.uint64)
    env = BatchedMDP(ms, rng_mode=L.RNG_PHILOX, philox_keys=keys, with_dp=False)
    env.set_observation_table(tabs)
    env.reset()
    for _ in range(5):
        env.rollout(3)
        cur, h, _ = env.state()
        clean = env.observe()
        for b in range(3):
            np.testing.assert_array_equal(clean[b], tabs[b][h[b], cur[b]])
    # noise: Box-Muller on the instance's Philox stream, domain 4, counter = observation number
    F = tabs[0].shape[-1]
    for n_obs in range(3):
        noisy = env.observe(0.5)
        for b in range(3):
            want = np.empty(F, np.float32)
            for j in range(F):
                w = O.philox((n_obs, 0, 4, j >> 1), (int(keys[b]) & 0xffffffff, int(keys[b]) >> 32))
                a, c = (w[2], w[3]) if j & 1 else (w[0], w[1])
                zz = np.sqrt(-2.0 * np.log((float(a) + 1.0) / 4294967296.0)) * np.cos(2 * np.pi * float(c) / 4294967296.0)
                want[j] = clean[b][j] + np.float32(0.5 * zz)
            np.testing.assert_allclose(noisy[b], want, rtol=1e-6, atol=1e-6)
    big = np.stack([env.observe(1.0) - clean for _ in range(300)])   # 300 x 3 x F samples of N(0, 1)
    assert abs(big.mean()) < 0.06 and abs(big.std() - 1.0) < 0.06
    env.close()

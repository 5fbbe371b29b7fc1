"""Parity sweep of the shared-table rollout K1T at scale: batch sizes around the workgroup capacity (ragged last groups, one
or both halves of a workgroup in use), DeepSea sizes 3..40, launch lengths around the chunk / Philox-block / flush
boundaries, composed launches -- every result against the HBM-table kernel K1 (itself pinned to the CPU oracle by the
test-suite and the fuzz sweep): last observation, float64 reward sum, all state and state-action visit counters.
    python tools/stress_k1t.py [seconds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colosseum_amd import _lib as L  # noqa: E402
from colosseum_amd.batched import BatchedMDP  # noqa: E402
from colosseum_amd.mdp.fast_batch import deepsea_episodic_tables  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
# second argument "k1u": the streamed-trace form K1U (k_rollout_tmpl_stream + k_trace_hist) instead of K1T;
# "k1e": the episode-parallel kernel K1E (k_rollout_epi + k_reward_scan; DeepSea-40 has too many states for it and is skipped;
# CMDP_K1E_OVERLAP as in the environment: the reward scan on the second stream or not)
K1E = len(sys.argv) > 2 and sys.argv[2] == "k1e"
K1U = len(sys.argv) > 2 and sys.argv[2] == "k1u"
WHICH = L.ROLLOUT_EPISODE_PARALLEL if K1E else L.ROLLOUT_LDS_TEMPLATE_STREAM if K1U else L.ROLLOUT_LDS_TEMPLATE
GENV = "CMDP_K1U_G" if K1U else "CMDP_K1T_G"
KNAME = "k_rollout_epi" if K1E else "k_rollout_tmpl_stream" if K1U else "k_rollout_tmpl"
rng = np.random.default_rng(7)
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    size = int(rng.choice([3, 4, 5, 8, 13, 21, 30, 31] if K1E else [3, 4, 5, 8, 13, 21, 30, 40]))
    B = int(rng.choice([1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000, 128 * 256 - 1, 128 * 256 + 1, 40000, 65536]))
    if size >= 30 and B > 40000 and rng.integers(0, 3):
        B = 40000
    g = int(rng.choice([0, 0, 17, 64, 65, 100, 128] + ([129, 200, 255, 256] if K1U else [])))   # 0: the library's own choice
    lens = [int(x) for x in rng.choice([1, 2, 3, 31, 32, 33, 63, 64, 65, 255, 257, 3551, 3553, 4097, 9001, 30001] + ([32767, 32769, 70001] if K1U else []) + ([29, 30, 59, 60, 61, 3839, 3840, 3841, 30000, 61439, 61441, 70001] if K1E else []),
                                       size=int(rng.integers(1, 4)))]
    seeds = rng.integers(0, 1 << 30, B)
    tables = deepsea_episodic_tables(seeds, size)
    keys = rng.integers(1, 1 << 62, B).astype(np.uint64)
    res = []
    for which in (WHICH, L.ROLLOUT_GLOBAL):
        if g and which == WHICH:
            os.environ[GENV] = str(g)
        try:
            env = BatchedMDP(tables=tables, rng_mode=L.RNG_PHILOX, philox_keys=keys)
        finally:
            os.environ.pop(GENV, None)
        env.set_rollout_kernel(which)
        env.reset()
        try:
            outs = [env.rollout(n) for n in lens]
        except L.CmdpError as ex:   # the handle was planned onto K1L (not the pipeline kernel): K1T is not built for it
            assert which == WHICH and ex.code == L.ERR_UNSUPPORTED, ex
            print("stress_k1t: size %d, %d instances: not planned as a pipeline batch, K1T refused" % (size, B), flush=True)
            env.close()
            res = None
            break
        vs, vsa = env.visits()
        if which == WHICH:
            assert env.lds_plan()["kernel"] == KNAME, env.lds_plan()
        res.append((outs, vs, vsa, env.state()))
        env.close()
    if res is None:
        continue
    (oa, vsa_s, vsa_sa, sta), (ob, vsb_s, vsb_sa, stb) = res
    for x, y in zip(oa, ob):
        assert np.array_equal(x["last_obs"], y["last_obs"]) and np.array_equal(x["reward_sum"], y["reward_sum"]), (size, B, g, lens)
    assert np.array_equal(vsa_s, vsb_s) and np.array_equal(vsa_sa, vsb_sa), (size, B, g, lens)
    for x, y in zip(sta, stb):
        assert np.array_equal(x, y), (size, B, g, lens)
    cases += 1
    print("stress_k1t: case %d ok (size %d, %d instances, G %s, launches %s)" % (cases, size, B, g or "auto", lens), flush=True)
print("stress_k1t: %d cases, %s == K1 in every counter" % (cases, "K1E" if K1E else "K1U" if K1U else "K1T"))

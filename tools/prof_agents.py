import json, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from colosseum_amd import _lib as L
from colosseum_amd import benchmark as bm
from colosseum_amd.mdp import make_model

allcfg = json.load(open("/root/repo/tests/golden/G11_benchmark_configs.json"))
def group(bench, cls, scope, n=20):
    kw = allcfg[bench]["mdp_configs"][cls][scope]
    return [make_model(cls, seed=s, **kw) for s in range(n)], list(range(n))
for bench, cls, scope, agent in (("benchmark_episodic_ergodic", "DeepSeaEpisodic", "prms_0", "QLearningEpisodic"),
                                 ("benchmark_continuous_ergodic", "FrozenLakeContinuous", "prms_0", "QLearningContinuous")):
    ms, seeds = group(bench, cls, scope)
    ms = [m for m in ms if m.H == ms[0].H]
    seeds = seeds[:len(ms)]
    for steps in (20000, 200000):
        t0 = time.time()
        rows = bm._run_group(ms, seeds, agent, bm.DEFAULT_AGENT_CONFIGS[agent], steps, 10020, L.RNG_MT_COMPAT, 0)
        print(cls, "S=%d" % ms[0].n_states, "steps", steps, "logs", len(rows[0]), "wall %.2f s" % (time.time() - t0), flush=True)

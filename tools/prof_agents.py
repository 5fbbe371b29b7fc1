"""Times single device batches of the benchmark runner at the reference's logging cadence (every 100 steps) and prints
where the host time goes.  usage: python tools/prof_agents.py [steps] [log_every]"""
import cProfile, io, json, pstats, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from colosseum_amd import _lib as L
from colosseum_amd import benchmark as bm
from colosseum_amd.mdp import make_model

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
log_every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
allcfg = json.load(open("/root/repo/tests/golden/G11_benchmark_configs.json"))
def group(bench, cls, scope, n):
    kw = allcfg[bench]["mdp_configs"][cls][scope]
    ms = [make_model(cls, seed=s, **kw) for s in range(n)]
    ms = [m for m in ms if m.H == ms[0].H]
    return ms, list(range(len(ms)))
for bench, cls, scope, agent, n in (("benchmark_episodic_ergodic", "DeepSeaEpisodic", "prms_0", "QLearningEpisodic", 20),
                                    ("benchmark_episodic_ergodic", "MiniGridEmptyEpisodic", "prms_0", "QLearningEpisodic", 40),
                                    ("benchmark_continuous_ergodic", "FrozenLakeContinuous", "prms_0", "QLearningContinuous", 20),
                                    ("benchmark_continuous_ergodic", "DeepSeaContinuous", "prms_0", "QLearningContinuous", 20)):
    ms, seeds = group(bench, cls, scope, n)
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    rows = bm._run_group(ms, seeds, agent, bm.DEFAULT_AGENT_CONFIGS[agent], steps, log_every, L.RNG_MT_COMPAT, 0)
    pr.disable()
    print(cls, "B=%d S=%d" % (len(ms), ms[0].n_states), "steps", steps, "logs", len(rows[0]), "wall %.2f s" % (time.time() - t0),
          "final", {k: float(v) for k, v in rows[0][-1].items() if k in ("normalized_cumulative_regret", "cumulative_reward")}, flush=True)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
    print("\n".join(l for l in s.getvalue().splitlines()[6:24]), flush=True)

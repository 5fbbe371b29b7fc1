"""Times the largest device batches of C4 alone (100 000 steps, a log row every 100): where does a big batch spend its time?"""
import json, os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from colosseum_amd import _lib as L
from colosseum_amd import benchmark as bm
from colosseum_amd.mdp import make_model
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
allcfg = json.load(open(R + "/tests/golden/G11_benchmark_configs.json"))
for bench, cls, scope, agent, n in (("benchmark_continuous_ergodic", "MiniGridEmptyContinuous", "prms_0", "QLearningContinuous", 20),
                                    ("benchmark_continuous_ergodic", "MiniGridEmptyContinuous", "prms_6", "QLearningContinuous", 20),
                                    ("benchmark_episodic_ergodic", "MiniGridEmptyEpisodic", "prms_0", "QLearningEpisodic", 20),
                                    ("benchmark_continuous_communicating", "MiniGridRoomsContinuous", "prms_0", "QLearningContinuous", 20)):
    kw = allcfg[bench]["mdp_configs"][cls][scope]
    ms = [make_model(cls, seed=s, **kw) for s in range(n)]
    ms = [m for m in ms if m.H == ms[0].H]
    t0 = time.time()
    rows = bm._run_group(ms, list(range(len(ms))), agent, bm.DEFAULT_AGENT_CONFIGS[agent], steps, 100, L.RNG_MT_COMPAT, 0)
    print(cls, scope, "B=%d S=%d H=%d" % (len(ms), ms[0].n_states, ms[0].H), "steps", steps, "wall %.2f s" % (time.time() - t0), flush=True)

"""One device batch of the benchmark runner, for rocprofv3.  usage: prof_one.py <bench> <cls> <scope> <agent> <n> <steps> <log_every>"""
import json, sys, time
sys.path.insert(0, "/root/repo")
from colosseum_amd import _lib as L
from colosseum_amd import benchmark as bm
from colosseum_amd.mdp import make_model

bench, cls, scope, agent, n, steps, log_every = sys.argv[1:8]
kw = json.load(open("/root/repo/tests/golden/G11_benchmark_configs.json"))[bench]["mdp_configs"][cls][scope]
ms = [make_model(cls, seed=s, **kw) for s in range(int(n))]
ms = [m for m in ms if m.H == ms[0].H]
t0 = time.time()
rows = bm._run_group(ms, list(range(len(ms))), agent, bm.DEFAULT_AGENT_CONFIGS[agent], int(steps), int(log_every), L.RNG_MT_COMPAT, 0)
print(cls, len(ms), ms[0].n_states, "wall %.2f" % (time.time() - t0))

import json, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from colosseum_amd import benchmark as bm
from colosseum_amd.experiment import MDPLoop, make_mdp_spec, batched_loop
from colosseum_amd.mdp import gpu_mdp
from helpers_agents import QLearningEpisodic
batched_loop._BatchedLoop.native = os.environ.get("NATIVE", "1") == "1"
cfg = json.load(open(R + "/tests/golden/G11_benchmark_configs.json"))["benchmark_episodic_ergodic"]["mdp_configs"]
inst = [i for i in bm.enumerate_instances(cfg, int(os.environ.get("NSEEDS", "20"))) if i.mdp_cls == "MiniGridEmptyEpisodic"]
n_steps, log_every = 1500, 500
models = bm.build_shard_models(inst)
groups = {}
for i, ins in enumerate(inst):
    m = models[i]
    groups.setdefault((m.H, m.n_actions), []).append((ins.mdp_scope, ins.seed, m.n_states))
for k, v in groups.items():
    print("group", k, "n", len(v), "scopes", sorted({(s, S) for s, _, S in v}))
res = bm.run_instances(inst, n_steps=n_steps, log_every=log_every, models=models)
seen = set()
for i, ins in enumerate(inst):
    if ins.mdp_scope in seen or ins.mdp_kwargs.get("make_reward_stochastic"):
        continue
    seen.add(ins.mdp_scope)
    mdp = getattr(gpu_mdp, ins.mdp_cls)(seed=ins.seed, **ins.mdp_kwargs)
    agent = QLearningEpisodic(seed=ins.seed, mdp_specs=make_mdp_spec(mdp), optimization_horizon=n_steps, **bm.DEFAULT_AGENT_CONFIGS[ins.agent_cls])
    loop = MDPLoop(mdp, agent)
    loop.run(T=n_steps, log_every=log_every)
    bad = [(got["steps"], k, float(got[k]), float(ref[k])) for got, ref in zip(res[i], loop.logger.data) for k in ref
           if k != "steps_per_second" and float(got[k]) != float(ref[k])]
    print(ins.mdp_scope, "seed", ins.seed, "S", mdp.n_states, "H", mdp.H, "mismatches", bad[:4])
    alone = bm.run_instances([ins], n_steps=n_steps, log_every=log_every)[0]
    bad2 = [(got["steps"], k, float(got[k]), float(ref[k])) for got, ref in zip(res[i], alone) for k in ref
            if k != "steps_per_second" and float(got[k]) != float(ref[k])]
    print("    vs alone:", bad2[:4])
    mdp.close()

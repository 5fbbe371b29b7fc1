"""Worker of tests/test_gpu_multirank.py::test_c4_at_its_real_length_reference_streams: ONE benchmark instance alone through
the per-instance path `GpuMDP` + `MDPLoop` + numpy agent (tests/helpers_agents.py) in a process of its own (the instances run
side by side).  The loop is set up for the FULL run (T = 500 000: the agent's optimisation horizon and MDPLoop's freeze rule
depend on it) and cut after `stop_after` steps -- it is causal, so the rows logged until then are the full run's first rows.
(The per-step Python loop takes 0.5-2.3 ms per step: all 500 000 steps of five instances do not fit a test suite.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_instance_alone(job):
    suite, idx, n_steps, log_every, golden, stop_after = job
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from colosseum_amd import benchmark as bm
    from colosseum_amd.experiment import MDPLoop, make_mdp_spec
    from colosseum_amd.mdp import gpu_mdp
    from helpers_agents import QLearningContinuous, QLearningEpisodic

    cfg = json.load(open(os.path.join(golden, "G11_benchmark_configs.json")))
    instances = bm.enumerate_instances(cfg[suite]["mdp_configs"], n_seeds=20)
    ins = instances[idx % len(instances)]
    mdp = getattr(gpu_mdp, ins.mdp_cls)(seed=ins.seed, **ins.mdp_kwargs)
    agent_cls = QLearningEpisodic if ins.agent_cls == "QLearningEpisodic" else QLearningContinuous
    agent = agent_cls(seed=ins.seed, mdp_specs=make_mdp_spec(mdp), optimization_horizon=n_steps,
                      **bm.DEFAULT_AGENT_CONFIGS[ins.agent_cls])
    loop = MDPLoop(mdp, agent)

    class _Enough(Exception):
        pass

    taken, real_step = [0], mdp.step

    def counted_step(action, *a, **k):
        if taken[0] >= stop_after:
            raise _Enough()
        taken[0] += 1
        return real_step(action, *a, **k)

    mdp.step = counted_step
    try:
        loop.run(T=n_steps, log_every=log_every)
    except _Enough:
        pass
    rows = [{k: float(v) for k, v in r.items() if k != "steps_per_second"} for r in loop.logger.data]
    mdp.close()
    return dict(label=ins.label, seed=ins.seed, beta=bool(ins.mdp_kwargs.get("make_reward_stochastic")), rows=rows)

"""Sharded benchmark runner (BASELINE config C4).

The reference enumerates (seed x MDP gin scope x agent gin scope) experiment instances, runs each in its own OS process
(colosseum/experiment/experiment_instances.py:117-223) and meets the results on the filesystem as one CSV per instance
(`<folder>/logs/<mdp_scope>-<MDP>____<agent_scope>-<Agent>/seed<n>_logs.csv`, header = sorted indicator names,
colosseum/utils/acme/csv_logger.py:99, colosseum/experiment/experiment_instance.py:53-82).

Here the same enumeration (seed-major, folder_structuring.py:76-104) is block-partitioned over the ranks (one process per
GPU, `colosseum_amd.sharding`), each rank groups its instances into device batches (same class, horizon and action
count), runs them with agents on the device (`colosseum_amd.agents`, `experiment.batched_loop`) and writes the same CSV
files.  The only communication is the final gather of one summary vector per instance.

Gin files of the benchmark folders only contain `prms_<i>/<Class>.<param> = <literal>` lines; they are parsed with a
regular expression (no gin dependency)."""
import ast
import csv
import os
import re
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import _lib as L
from .agents import BatchedQLearningContinuous, BatchedQLearningEpisodic
from .batched import BatchedMDP
from .experiment.batched_loop import BatchedContinuousLoop, BatchedEpisodicLoop
from .mdp import make_model
from .sharding import shard_range

_GIN_LINE = re.compile(r"^\s*(prms_\d+)/(\w+)\.(\w+)\s*=\s*(.+?)\s*$")

# tuned hyper-parameters shipped with the reference (benchmark/cached_hyperparameters/agent_configs/*.gin)
DEFAULT_AGENT_CONFIGS = {
    "QLearningEpisodic": dict(p=0.05, UCB_type="bernstein", c_1=0.9415278732894797, c_2=0.013873778519317169,
                              min_at=0.07263563483119442),
    "QLearningContinuous": dict(h_weight=0.942, span_approx_weight=0.014, min_at=0.073),
}


def parse_gin(text: str) -> Dict[str, Dict[str, Dict[str, Any]]]:
    """{class: {scope: {param: value}}} from `prms_0/DeepSeaEpisodic.size=10` lines."""
    out: Dict[str, Dict[str, Dict[str, Any]]] = {}
    for line in text.splitlines():
        m = _GIN_LINE.match(line)
        if not m:
            continue
        scope, cls, key, val = m.groups()
        out.setdefault(cls, {}).setdefault(scope, {})[key] = ast.literal_eval(val)
    return out


def load_mdp_configs(folder: str) -> Dict[str, Dict[str, Dict[str, Any]]]:
    """All `mdp_configs/*.gin` of a benchmark folder laid out like the reference's."""
    cfg: Dict[str, Dict[str, Dict[str, Any]]] = {}
    d = os.path.join(folder, "mdp_configs")
    for f in sorted(os.listdir(d)):
        if f.endswith(".gin"):
            for cls, scopes in parse_gin(open(os.path.join(d, f)).read()).items():
                cfg.setdefault(cls, {}).update(scopes)
    return cfg


@dataclass
class Instance:
    seed: int
    mdp_cls: str
    mdp_scope: str
    mdp_kwargs: Dict[str, Any]
    agent_cls: str
    agent_scope: str = "prms_0"

    @property
    def label(self) -> str:  # ExperimentInstance.experiment_label
        return f"{self.mdp_scope}-{self.mdp_cls}____{self.agent_scope}-{self.agent_cls}"


def enumerate_instances(mdp_configs: Dict[str, Dict[str, Dict[str, Any]]], n_seeds: int,
                        supported_only: bool = True) -> List[Instance]:
    """Seed-major enumeration of folder_structuring.py:76-104 with one tabular Q-learning agent per setting."""
    from .mdp.registry import FAMILIES

    out = []
    for seed in range(n_seeds):
        for cls, scopes in mdp_configs.items():
            fam = cls.replace("Episodic", "").replace("Continuous", "")
            if fam not in FAMILIES:
                if supported_only:
                    continue
                raise KeyError(cls)
            agent = "QLearningEpisodic" if cls.endswith("Episodic") else "QLearningContinuous"
            for scope in sorted(scopes, key=lambda s: int(s.split("_")[1])):
                out.append(Instance(seed, cls, scope, dict(scopes[scope]), agent))
    return out


def _build_one(ins: Instance):
    m = make_model(ins.mdp_cls, seed=ins.seed, **ins.mdp_kwargs)
    m.csr()             # the DP views every batch needs, computed where the model is built (a pool of processes), not under
    m.reward_matrix()   # the GIL of the threads that drive the device batches; they travel with the model (small arrays)
    return m


def _build_models(instances: Sequence[Instance], workers: int):
    """Host-side graph construction (Python) for many instances; a fork()ed pool must be used BEFORE the HIP runtime is
    initialised in this process, so `workers > 0` is only honoured then (the caller's responsibility)."""
    if workers and workers > 1 and len(instances) >= 4 * workers:
        import multiprocessing as mp

        with mp.get_context("fork").Pool(workers) as pool:
            return pool.map(_build_one, instances, chunksize=max(1, len(instances) // (workers * 8)))
    return [_build_one(i) for i in instances]


def build_shard_models(instances: Sequence[Instance], rank: int = 0, world: int = 1, workers: int = 0,
                       skip: Optional[Sequence[int]] = None) -> Dict[int, Any]:
    """{global instance index: TabularModel} of this rank's contiguous shard (fork()ed pool: call it before anything
    initialises the HIP runtime or RCCL in this process)."""
    lo, hi = shard_range(len(instances), rank, world)
    skip = set(skip or ())
    mine = [i for i in range(lo, hi) if i not in skip]
    return dict(zip(mine, _build_models([instances[i] for i in mine], workers)))


def _run_group(models, seeds, agent_cls, agent_kwargs, n_steps, log_every, rng_mode, device, max_time=np.inf,
               beta_rewards="reference"):
    L.check(L.load().cmdp_set_device(device))
    env, agent, loop, exact, t0, t1 = _setup_group(models, seeds, agent_cls, agent_kwargs, n_steps, rng_mode, beta_rewards)
    rows = loop.run(n_steps, log_every, max_time)
    t2 = time.time()
    for b, table in enumerate(rows):  # what MDPLoop.run returns first: where the time limit froze training (-1: it did not)
        table.last_training_step = int(loop.last_training_step[b])
    if exact and rows:
        rows[0].reward_cache_stats = env.reward_cache_stats()
    agent.close()
    env.close()
    if rows:
        rows[0].phase_seconds = (t1 - t0, t2 - t1, time.time() - t2)   # tables + baselines + agent | the interaction | release
    return rows


def _setup_group(models, seeds, agent_cls, agent_kwargs, n_steps, rng_mode, beta_rewards):
    stochastic = any(not m.deterministic_rewards for m in models)
    # Beta rewards, "reference": the reference's own per-triple caches of 5000 samples from the MDP's numpy stream
    # (CMDP_FLAG_REWARD_CACHE: blocks in HBM, drawn by the library on the host whenever an instance needs one), next to
    # the reference's transition streams -- rows equal the reference's for that seed.  "philox": sampled on the device
    # from counter-based streams (distribution-exact throughput mode; no host work).
    exact = stochastic and beta_rewards == "reference"
    t0 = time.time()
    env = BatchedMDP(models, rng_mode=L.RNG_PHILOX if stochastic and not exact else rng_mode,
                     philox_keys=np.asarray(seeds, np.uint64) * np.uint64(0x9E3779B1) + np.uint64(17),
                     flags=L.FLAG_REWARD_CACHE if exact else (L.FLAG_BETA_GAMMAS if beta_rewards == "philox-gammas" else 0))
    if agent_cls == "QLearningEpisodic":
        agent = BatchedQLearningEpisodic(env, seeds, optimization_horizon=n_steps, **agent_kwargs)
        loop = BatchedEpisodicLoop(env, agent)
    else:
        agent = BatchedQLearningContinuous(env, seeds, optimization_horizon=n_steps, **agent_kwargs)
        loop = BatchedContinuousLoop(env, agent)
    return env, agent, loop, exact, t0, time.time()


def run_instances(instances: Sequence[Instance], n_steps: int, log_every: int, rank: int = 0, world: int = 1,
                  agent_configs: Optional[Dict[str, Dict[str, Any]]] = None, rng_mode: int = L.RNG_MT_COMPAT,
                  device: int = 0, max_concurrent_groups: int = 6, build_workers: int = 0, progress=None,
                  max_batch: int = 128, models: Optional[Dict[int, Any]] = None, max_time: float = np.inf,
                  skip: Optional[Sequence[int]] = None, beta_rewards: str = "reference", on_group_done=None):
    """Runs this rank's contiguous shard; returns {global instance index: logger rows}.  `models` = the shard's
    models from `build_shard_models` (a caller that must initialise RCCL between the fork()ed build and the first HIP
    call builds them itself).  `max_time`: the experiment's `max_interaction_time_s` (training of an instance is frozen
    once its batch has run that long, agent_mdp_interaction.py:160-177).  `skip`: global indices not to run -- instances
    whose log file already exists (`unfinished_instances`), as the reference's resume does.  `beta_rewards`: "reference"
    (default: the reference's reward caches and streams, rows equal the reference's) or "philox" (device-sampled Beta
    rewards, distribution-exact).  `on_group_done(indices, rows)` is called from the worker thread as soon as a device
    batch has finished (the runner writes that batch's log files then, so an interrupted run resumes from them)."""
    assert beta_rewards in ("reference", "philox", "philox-gammas")
    agent_configs = agent_configs or DEFAULT_AGENT_CONFIGS
    lo, hi = shard_range(len(instances), rank, world)
    skip = set(skip or ())
    mine = [i for i in range(lo, hi) if i not in skip]
    if models is None:
        models = build_shard_models(instances, rank, world, build_workers, skip)
    groups: Dict[tuple, List[int]] = {}
    from .hardness import _vi_rule

    for i in mine:
        m = models[i]
        # continuous baselines use discounted VI under the reference's own scheme rule, which depends on the instance's
        # size and density: one scheme per device batch
        scheme = 0 if m.is_episodic else _vi_rule(m.n_states, m.n_actions, len(m.csr()[1]))
        # deterministic- and Beta-reward instances never share a device batch: in "philox" mode the two run different
        # transition streams (MT_COMPAT / Philox), and what an instance produces must not depend on which other
        # instances the benchmark holds
        groups.setdefault((instances[i].mdp_cls, m.H, m.n_actions, tuple(m.rewards_range), scheme, m.deterministic_rewards),
                          []).append(i)
    results: Dict[int, list] = {}

    def work(idx):

        ins = instances[idx[0]]
        t0 = time.time()
        rows = _run_group([models[i] for i in idx], [instances[i].seed for i in idx], ins.agent_cls,
                          agent_configs[ins.agent_cls], n_steps, log_every, rng_mode, device, max_time, beta_rewards)
        if on_group_done:
            on_group_done(idx, rows)
        if progress:
            rcs = getattr(rows[0], "reward_cache_stats", None) if rows else None
            ph = getattr(rows[0], "phase_seconds", None) if rows else None
            progress(f"{ins.label}: {len(idx)} instances, S={models[idx[0]].n_states}, H={models[idx[0]].H}, "
                     f"{time.time() - t0:.1f} s" + (f" (set-up and baselines {ph[0]:.1f} s, interaction {ph[1]:.1f} s, release {ph[2]:.1f} s)" if ph else "")
                     + (f", reward blocks {rcs['fills']} in {rcs['rounds']} rounds, {rcs['round_ms'] / 1e3:.1f} s "
                                                    f"in rounds of which {rcs['fill_ms'] / 1e3:.1f} s drawing" if rcs else ""))
        return idx, rows

    # Device batches are small next to the GPU (20-220 instances: a handful of wavefronts in latency-bound kernels), so a
    # few of them are driven concurrently, one host thread and one HIP stream each; the C calls release the GIL.  This
    # only pays because the per-log host work is short (vector_tracker): with per-instance Python trackers the threads
    # convoyed on the GIL (every one of the ~6 short C calls per log waited a 5 ms switch interval) and ran 10-100x
    # slower than one after the other.  Largest batches first, so that the tail is short.
    # very large batches are split: their per-log kernels are latency-bound per instance, so two halves in flight on two
    # streams finish sooner than one batch (and the longest batch is what bounds the wall time)
    parts = []
    for idx in groups.values():
        n_parts = -(-len(idx) // max(1, max_batch))
        size = -(-len(idx) // n_parts)
        parts += [idx[i:i + size] for i in range(0, len(idx), size)]
    # longest first.  A batch's time is rows x (dependent kernel time per row), which hardly depends on the number of
    # instances: per row the episodic loop evaluates a policy over H x S (state, time) pairs, the continuous one solves a
    # chain of S states (measured on C4: ~1.5 s per 1 000 for either measure)
    def expected(idx):
        m = models[idx[0]]
        return (m.H * m.n_states if m.is_episodic else 6 * m.n_states) * (1.0 + len(idx) / 400.0)

    order = sorted(parts, key=lambda idx: -expected(idx))
    import sys

    old_interval = sys.getswitchinterval()
    sys.setswitchinterval(2e-4)
    try:
        with ThreadPoolExecutor(max_workers=max(1, max_concurrent_groups)) as pool:
            for idx, rows in pool.map(work, order):
                for i, r in zip(idx, rows):
                    results[i] = r
    finally:
        sys.setswitchinterval(old_interval)
    return results


def _write_atomic(path: str, text: str):
    """The log file appears complete or not at all: a run killed mid-write must not leave a truncated file that the
    resume (`unfinished_instances`) would take for a finished instance."""
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w", newline="") as f:
        f.write(text)
    os.replace(tmp, path)


def _write_slice(task):
    """Worker of `submit_group_logs` (runs in a spawned process): formats the CSV texts of a slice of one device batch
    and writes them, each file atomically, with the `time_exceeded.txt` ledger lines."""
    from .experiment.vector_tracker import _csv_texts_of_slice

    steps, cols, n, paths, last_steps = task
    for text, path, last in zip(_csv_texts_of_slice((steps, cols, n)), paths, last_steps):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        _write_atomic(path, text)
        if last != -1:  # the ledger run_experiment_instance appends to (experiment_instances.py:218-222)
            with open(os.path.join(os.path.dirname(path), "time_exceeded.txt"), "a") as f:
                f.write(f"last training step at ({last}) for {path}\n")
    return len(paths)


def submit_group_logs(pool, folder: str, instances: Sequence[Instance], idx: Sequence[int], rows, chunk: int = 8):
    """Hands the log files of ONE finished device batch to `pool` (a concurrent.futures executor of spawned processes:
    the float -> text conversion is Python-level work that must not hold the GIL of the threads driving the other
    batches).  Returns the futures.  Files are written as their batch finishes, so an interrupted benchmark resumes from
    what is on disk (the reference writes every instance's file at its own end, experiment_instance.py:53-82)."""
    log = rows[0].log
    fin = log.finalize()
    futs = []
    for b0 in range(0, len(idx), chunk):
        b1 = min(len(idx), b0 + chunk)
        bs = [rows[j].b for j in range(b0, b1)]
        assert bs == list(range(bs[0], bs[0] + len(bs)))
        cols = {n: (v[:, bs[0]:bs[-1] + 1], k if isinstance(k, int) else k[:, bs[0]:bs[-1] + 1]) for n, (v, k) in fin.items()}
        paths = [log_file(folder, instances[idx[j]]) for j in range(b0, b1)]
        lasts = [int(getattr(rows[j], "last_training_step", -1)) for j in range(b0, b1)]
        futs.append(pool.submit(_write_slice, (list(log.steps), cols, b1 - b0, paths, lasts)))
    return futs


def write_csv_logs(folder: str, instances: Sequence[Instance], results: Dict[int, list], workers: int = 0):
    """The reference's on-disk wire format (CSVLogger with add_uid=False; header = sorted keys).  `workers` > 1
    formats the column stores of the device batches in a process pool.  Every file is written atomically."""
    from .experiment.vector_tracker import LogTable, csv_texts_parallel

    logs = {}
    for rows in results.values():
        if isinstance(rows, LogTable):
            logs.setdefault(id(rows.log), rows.log)
    texts = csv_texts_parallel(list(logs.values()), workers) if logs else {}
    for i, rows in results.items():
        ins = instances[i]
        d = os.path.join(folder, "logs", ins.label)
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"seed{ins.seed}_logs.csv")
        if isinstance(rows, LogTable):
            _write_atomic(path, texts[id(rows.log)][rows.b])
        else:
            import io

            buf = io.StringIO(newline="")
            fields = sorted(rows[0].keys())
            w = csv.DictWriter(buf, fieldnames=fields, extrasaction="ignore")
            w.writeheader()
            for r in rows:
                w.writerow({k: np.array(v) for k, v in r.items()})
            _write_atomic(path, buf.getvalue())
        last = getattr(rows, "last_training_step", -1)
        if last != -1:  # the ledger run_experiment_instance appends to (experiment_instances.py:218-222)
            with open(os.path.join(d, "time_exceeded.txt"), "a") as f:
                f.write(f"last training step at ({last}) for {os.path.join(d, f'seed{ins.seed}_logs.csv')}\n")


def log_file(folder: str, ins: Instance) -> str:
    """ExperimentInstance's log file (experiment_instance.py:67-82)."""
    return os.path.join(folder, "logs", ins.label, f"seed{ins.seed}_logs.csv")


def unfinished_instances(folder: str, instances: Sequence[Instance]) -> List[int]:
    """Indices of the instances whose log file does not exist yet: the reference only (re-)creates those
    (`does_log_file_exists`, folder_structuring.py:102), which is how an interrupted benchmark resumes."""
    return [i for i, ins in enumerate(instances) if not os.path.exists(log_file(folder, ins))]


def read_summary(folder: str, ins: Instance) -> np.ndarray:
    """`summary_vector` of an instance from its log file (instances skipped by a resumed run)."""
    with open(log_file(folder, ins), newline="") as f:
        last = list(csv.DictReader(f))[-1]
    return np.array([float(last["steps"]), float(last["normalized_cumulative_regret"]), float(last["cumulative_reward"])])


def summary_vector(rows) -> np.ndarray:
    """What is gathered across ranks at the end: last step, normalized cumulative regret, cumulative reward."""
    last = rows[-1]
    return np.array([last["steps"], last["normalized_cumulative_regret"], last["cumulative_reward"]], np.float64)

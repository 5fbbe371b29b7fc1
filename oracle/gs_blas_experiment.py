#!/usr/bin/env python3
"""How reproducible is the reference's Gauss-Seidel value iteration itself?  

`_discounted_value_iteration` (colosseum/dynamic_programming/infinite_horizon.py:121-142) computes `T[s] @ V` -- an
[A, S] x [S] float32 product that numpy hands to BLAS `sgemv`, whose accumulation order (vector width, number of partial
sums, FMA or not) belongs to the BLAS KERNEL, i.e. to the CPU the reference happens to run on: OpenBLAS selects a kernel
per micro-architecture at load time (DYNAMIC_ARCH), and OPENBLAS_CORETYPE overrides that choice.  This script runs the
reference's own function (imported from /root/reference, numba.njit = identity) on the G4 instances
(FrozenLakeContinuous 20x20, gamma .99, eps 1e-6) once per kernel family, each in its own process, and records how far
the reference's value functions are from each other -- and how far this repository's in-order float32 restatement
(oracle/cmdp_oracle.c, what the HIP kernel K3 reproduces bit for bit) is from each of them.

    cd /tmp/gs && python /root/repo/oracle/gs_blas_experiment.py            -> profiles/r03_gs_blas_kernels.json
TEST INFRASTRUCTURE ONLY (development container; the reference does not travel)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CORETYPES = ["Prescott", "Nehalem", "Sandybridge", "Haswell", "Zen", "SkylakeX"]
SEEDS = [0, 1, 2, 3]


def child(coretype, out_path):
    sys.path.insert(0, HERE)
    import ref_env

    np = ref_env.install()
    from colosseum import config as ref_config
    from colosseum.dynamic_programming import infinite_horizon as ref_ih
    from colosseum.mdp.frozen_lake import FrozenLakeContinuous

    ref_config.disable_multiprocessing()
    res = {}
    for seed in SEEDS:
        mdp = FrozenLakeContinuous(seed=seed, size=20, p_frozen=0.9, is_slippery=True, p_rand=0.1)
        T, R = mdp.transition_matrix_and_rewards
        T = np.asarray(T.todense() if hasattr(T, "todense") else T)
        Q, V = ref_ih._discounted_value_iteration(T, R, 0.99, 1e-6)
        res[f"V{seed}"] = np.asarray(V, np.float32)
        res[f"Q{seed}"] = np.asarray(Q, np.float32)
    try:
        from threadpoolctl import threadpool_info

        res["blas"] = np.array(json.dumps([{k: str(v) for k, v in d.items()} for d in threadpool_info()]))
    except Exception:
        pass
    np.savez(out_path, **res)


def main():
    import numpy as np

    sys.path.insert(0, ROOT)
    from colosseum_amd.mdp import make_model
    from oracle import oracle as O

    tmp = os.getcwd()
    runs = {}
    for ct in CORETYPES:
        out = os.path.join(tmp, f"gs_{ct}.npz")
        env = dict(os.environ, OPENBLAS_CORETYPE=ct, OPENBLAS_NUM_THREADS="1", PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", ct, out], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(ct, "failed:", r.stderr[-400:])
            continue
        runs[ct] = np.load(out)
        info = json.loads(str(runs[ct]["blas"])) if "blas" in runs[ct] else []
        arch = [d.get("architecture") for d in info if "openblas" in d.get("internal_api", "")]
        print(ct, "-> OpenBLAS reports architecture", arch, flush=True)
    ours = {}
    for seed in SEEDS:
        m = make_model("FrozenLakeContinuous", seed=seed, size=20, p_frozen=0.9, p_rand=0.1)
        _, V, it, _ = O.vi_discounted(m.n_states, m.n_actions, m.csr(), m.reward_matrix(), 0.99, 1e-6, 2)
        ours[seed] = np.asarray(V, np.float32)
    names = list(runs)
    report = {"what": __doc__.split("\n\n")[0], "instances": "FrozenLakeContinuous(seed=0..3, size=20, p_frozen=0.9, p_rand=0.1), gamma 0.99, eps 1e-6",
              "kernels": names, "reference_vs_reference": {}, "restatement_vs_reference": {}}
    worst_ref, worst_ours = 0.0, 0.0
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            d = max(float(np.abs(runs[a][f"V{s}"] - runs[b][f"V{s}"]).max()) for s in SEEDS)
            rel = max(float((np.abs(runs[a][f"V{s}"] - runs[b][f"V{s}"]) / np.maximum(np.abs(runs[b][f"V{s}"]), 1e-30)).max()) for s in SEEDS)
            report["reference_vs_reference"][f"{a} vs {b}"] = {"max_abs": d, "max_rel": rel}
            worst_ref = max(worst_ref, d)
    for a in names:
        d = max(float(np.abs(runs[a][f"V{s}"] - ours[s]).max()) for s in SEEDS)
        rel = max(float((np.abs(runs[a][f"V{s}"] - ours[s]) / np.maximum(np.abs(ours[s]), 1e-30)).max()) for s in SEEDS)
        report["restatement_vs_reference"][a] = {"max_abs": d, "max_rel": rel}
        worst_ours = max(worst_ours, d)
    report["largest_distance_between_two_runs_of_the_reference"] = worst_ref
    report["largest_distance_of_the_restatement_from_a_run_of_the_reference"] = worst_ours
    report["value_scale"] = float(max(np.abs(ours[s]).max() for s in SEEDS))
    json.dump(report, open(os.path.join(ROOT, "profiles", "r03_gs_blas_kernels.json"), "w"), indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
    else:
        main()

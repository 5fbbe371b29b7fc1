import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from colosseum_amd import _lib as L
from colosseum_amd.batched import BatchedMDP
from colosseum_amd.mdp import make_model
m = make_model("MiniGridEmptyEpisodic", seed=0, size=20, n_starting_states=2) if len(sys.argv) < 2 else make_model("DeepSeaEpisodic", seed=0, size=20)
B = 40
env = BatchedMDP([m] * B, rng_mode=L.RNG_PHILOX, with_env=False)
H, S, A = m.H, m.n_states, m.n_actions
print("H S A", H, S, A)
rng = np.random.default_rng(0)
for name, Q in (("all ties", np.zeros((B, H, S, A), np.float32)), ("no ties", rng.random((B, H, S, A)).astype(np.float32)),
                ("half", (rng.random((B, H, S, A)) > 0.5).astype(np.float32))):
    qs = [Q[b].ravel() for b in range(B)]
    env.greedy_policy_episodic(qs, H)
    t0 = time.time()
    for _ in range(5):
        env.greedy_policy_episodic(qs, H)
    print(name, "%.3f ms per call (incl. copies)" % ((time.time() - t0) / 5 * 1e3))

"""Randomised parity sweep: random family / size / noise parameters, GPU against the CPU oracle, bit for bit.
    python tools/fuzz_parity.py [seconds] [seed]
Covers: MT_COMPAT trajectories with host actions, Philox random-policy rollouts (LDS-resident kernel when eligible),
discounted VI/PE (both schemes), episodic VI, diameter (workgroup kernel and the 64-targets-per-workgroup kernel),
average reward of random deterministic policies (K9, exact-order mode against the host restatement)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from colosseum_amd import _lib as L
from colosseum_amd.batched import BatchedMDP
from colosseum_amd.markov_chain import get_average_reward
from colosseum_amd.mdp import make_model
from oracle import oracle as O
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests"))
from colosseum_amd import timestep as ts_
from colosseum_amd.agents import BatchedQLearningContinuous, BatchedQLearningEpisodic
from helpers_agents import QLearningContinuous, QLearningEpisodic   # numpy restatements of the reference agents (pinned to G7 / G10)

rng = np.random.default_rng(0)
counts = {}   # rollout kernel -> cases it ran


def random_model():
    fam = rng.choice(["DeepSea", "FrozenLake", "MiniGridEmpty", "MiniGridRooms", "RiverSwim", "SimpleGrid", "Taxi"])
    episodic = bool(rng.integers(0, 2))
    kw = dict(seed=int(rng.integers(0, 10_000)))
    p_rand = float(rng.choice([0, 0.05, 0.3])) or None
    p_lazy = float(rng.choice([0, 0.1])) or None
    if fam == "DeepSea":
        kw.update(size=int(rng.integers(3, 14)))
        p_lazy = None
    elif fam == "FrozenLake":
        kw.update(size=int(rng.integers(3, 9)), p_frozen=float(rng.choice([0.7, 0.9, 1.0])), is_slippery=bool(rng.integers(0, 2)))
    elif fam == "MiniGridEmpty":
        kw.update(size=int(rng.integers(3, 8)), n_starting_states=int(rng.integers(1, 4)))
    elif fam == "MiniGridRooms":
        kw.update(room_size=int(rng.integers(3, 5)), n_rooms=int(rng.choice([4, 9])), n_starting_states=int(rng.integers(1, 3)))
    elif fam == "RiverSwim":
        kw.update(size=int(rng.integers(3, 30)))
    elif fam == "SimpleGrid":
        kw.update(size=int(rng.integers(3, 8)), reward_type=int(rng.integers(0, 4)), n_starting_states=int(rng.integers(1, 4)))
    else:
        kw.update(size=int(rng.integers(5, 7)))
    if p_rand: kw["p_rand"] = p_rand
    if p_lazy: kw["p_lazy"] = p_lazy
    if rng.integers(0, 4) == 0 and fam != "Taxi": kw["randomize_actions"] = False
    cls = fam + ("Episodic" if episodic else "Continuous")
    return cls, kw, make_model(cls, **kw)


def check_one():
    cls, kw, m = random_model()
    S, A, H = m.n_states, m.n_actions, m.H
    tag = (cls, kw)
    # --- MT_COMPAT with host actions + Philox random policy -----------------------------------------------------------
    n = int(rng.integers(50, 3000))
    acts = rng.integers(0, A, n).astype(np.int8)
    env = BatchedMDP([m, m], rng_mode=L.RNG_MT_COMPAT)
    first = env.reset()
    out = env.rollout(n, np.stack([acts, acts], 1), trace=True)
    e = O.OracleEnv(m, rng_mode=0)
    assert e.reset() == first[0], tag
    ref = e.rollout(n, acts)
    for b in range(2):
        assert np.array_equal(out["obs"][:, b], ref["obs"]) and np.array_equal(out["rew"][:, b], ref["rew"]), tag
    assert np.array_equal(env.split_states(env.visits()[0])[1], e.visits()[0]), tag
    # --- DP on the same handle -----------------------------------------------------------------------------------------
    if H:
        Q, V = env.episodic_value_iteration()
        oQ, oV = O.episodic(S, A, H, m.csr(), m.reward_matrix())
        assert np.array_equal(env.split_states(V, H + 1)[0].reshape(H + 1, S), oV), tag
    else:
        for scheme in (L.SCHEME_JACOBI, L.SCHEME_GAUSS_SEIDEL):
            g = float(rng.choice([0.9, 0.99]))
            Q, V, sw = env.value_iteration(g, 1e-5, scheme)
            oQ, oV, oit, _ = O.vi_discounted(S, A, m.csr(), m.reward_matrix(), g, 1e-5, scheme)
            assert np.array_equal(env.split_states(V)[1], oV) and sw[1] == oit, (tag, scheme)
        if S <= 120:
            d0, per0 = env.diameter(1e-3, L.SCHEME_JACOBI)
            env.set_option(L.OPT_DP_KERNEL, 3)
            d1, per1 = env.diameter(1e-3, L.SCHEME_JACOBI)
            env.set_option(L.OPT_DIAMETER_RELABEL_MIN_STATES, 1)   # K5S with the rows in the locality order of the states
            d2, per2 = env.diameter(1e-3, L.SCHEME_JACOBI)
            assert np.array_equal(per2, per1) and np.array_equal(d2, d1), tag
            env.set_option(L.OPT_DIAMETER_RELABEL_MIN_STATES, 1 << 40)
            env.set_option(L.OPT_DP_KERNEL, 0)
            _, oper = O.diameter_continuous(S, A, m.csr(), scheme=1)
            assert np.array_equal(per0[:S], oper) and np.array_equal(per1, per0), tag
        pol = rng.integers(0, A, S).astype(np.int32)
        st = int(rng.integers(0, S))
        env.set_option(L.OPT_CHAIN_EXACT_ORDER, 1)
        vals, ncls = env.average_reward([pol, pol], [st, st])
        T, R = m.dense()
        oh = np.zeros((S, A), np.float32); oh[np.arange(S), pol] = 1
        want = get_average_reward(T, R, oh, [(st, 1.0)])
        assert type(vals[0]) is type(want) and vals[0] == want, (tag, vals[0], want, ncls)
    env.close()
    if m.deterministic_rewards:
        keys = rng.integers(1, 2**40, 3).astype(np.uint64)
        env = BatchedMDP([m] * 3, rng_mode=L.RNG_PHILOX, philox_keys=keys, with_dp=False)
        env.reset()
        n1 = int(rng.integers(0, 200))  # the second launch starts mid Philox block / mid episode
        env.rollout(n1)
        o2 = env.rollout(n)
        vs, _ = env.visits()
        for b in range(3):
            e = O.OracleEnv(m, rng_mode=1, philox_key=int(keys[b]))
            e.reset()
            e.rollout(n1, trace=False)
            r2 = e.rollout(n, trace=False)
            assert o2["last_obs"][b] == r2["last_obs"] and o2["reward_sum"][b] == r2["reward_sum"], tag
            assert np.array_equal(env.split_states(vs)[b], e.visits()[0]), tag
        env.close()
        # the other rollout kernels on the same streams: HBM tables (K1), the LDS kernels where the batch is eligible
        # (K1L / K1P, the shared-table K1T / K1U, the stochastic-dynamics K1S, the episode-parallel K1E)
        for which in (L.ROLLOUT_GLOBAL, L.ROLLOUT_LDS, L.ROLLOUT_LDS_TEMPLATE, L.ROLLOUT_LDS_TEMPLATE_STREAM, L.ROLLOUT_LDS_STOCHASTIC,
                      L.ROLLOUT_EPISODE_PARALLEL):
            env = BatchedMDP([m] * 3, rng_mode=L.RNG_PHILOX, philox_keys=keys, with_dp=False)
            env.set_rollout_kernel(which)
            env.reset()
            try:
                env.rollout(n1)
            except L.CmdpError as ex:
                assert ex.code == L.ERR_UNSUPPORTED and which != L.ROLLOUT_GLOBAL, (tag, which, ex)
                env.close()
                continue
            o3 = env.rollout(n)
            assert np.array_equal(o3["last_obs"], o2["last_obs"]) and np.array_equal(o3["reward_sum"], o2["reward_sum"]), (tag, which)
            assert np.array_equal(env.visits()[0], vs), (tag, which)
            env.close()
            counts[which] = counts.get(which, 0) + 1
        # dense-row layout (K1D; three instances: the second walker of the last wavefront idles)
        if S <= 400:
            try:
                env = BatchedMDP([m] * 3, rng_mode=L.RNG_PHILOX, philox_keys=keys, layout=L.LAYOUT_DENSE)
            except L.CmdpError as ex:   # the layout's documented precondition: probabilities in [2^-28, 1] (exact float64 scans)
                assert "2^-28" in str(ex), (tag, ex)
                return
            env.reset()
            env.rollout(n1)
            o4 = env.rollout(n)
            counts["dense"] = counts.get("dense", 0) + 1
            for b in range(3):
                e = O.OracleEnv(m, rng_mode=1, philox_key=int(keys[b]), dense=True)
                e.reset()
                e.rollout(n1, trace=False)
                r4 = e.rollout(n, trace=False)
                assert o4["last_obs"][b] == r4["last_obs"] and o4["reward_sum"][b] == r4["reward_sum"], (tag, "dense")
            env.close()


class _Spec:
    def __init__(self, m):
        self.time_horizon = m.H if m.H else np.inf
        self.observations = type("o", (), {"num_values": m.n_states})()
        self.actions = type("a", (), {"num_values": m.n_actions})()


def check_agent():
    """Device Q-learning (both settings, random hyper-parameters) against the numpy agent stepping the CPU oracle:
    action streams and final tables bit-equal."""
    cls, kw, m = random_model()
    if m.n_states * max(m.H, 1) > 3000 or not m.deterministic_rewards:
        return
    n = int(rng.integers(200, 1500))
    seeds = [int(rng.integers(0, 1000)) for _ in range(2)]
    keys = rng.integers(1, 2**40, 2).astype(np.uint64)
    env = BatchedMDP([m, m], rng_mode=L.RNG_PHILOX, philox_keys=keys, with_dp=False)
    env.reset()
    if m.H:
        ucb = str(rng.choice(["hoeffding", "bernstein"]))
        hp = dict(p=float(rng.choice([0.05, 0.2])), c_1=float(rng.choice([0.01, 0.5, 0.94])), min_at=float(rng.choice([0.0, 0.07, 0.3])),
                  UCB_type=ucb, optimization_horizon=n)
        if ucb == "bernstein":
            hp["c_2"] = float(rng.choice([0.014, 0.5]))
        ag = BatchedQLearningEpisodic(env, seeds, **hp)
        host_cls = QLearningEpisodic
    else:
        hp = dict(min_at=float(rng.choice([0.0, 0.05])), confidence=float(rng.choice([0.9, 0.95])),
                  span_approx_weight=float(rng.choice([0.5, 1.0])), h_weight=float(rng.choice([0.5, 1.0])), optimization_horizon=n)
        ag = BatchedQLearningContinuous(env, seeds, **hp)
        host_cls = QLearningContinuous
    out = ag.run(n, train=True, trace_actions=True)
    Q, N = ag.tables()
    for i in range(2):
        e = O.OracleEnv(m, rng_mode=1, philox_key=int(keys[i]))
        host = host_cls(mdp_specs=_Spec(m), seed=seeds[i], **hp)
        ts, h, acts = ts_.restart(e.reset()), 0, []
        for t in range(n):
            a = int(host.select_action(ts, h if m.H else t))
            acts.append(a)
            ty, o, r, _ = e.step(a)
            nts = ts_.termination(r, -1) if ty == 2 else ts_.transition(r, o)
            host.step_update(ts, a, nts, h if m.H else t)
            h, ts = h + 1, nts
            if ty == 2:
                ts, h = ts_.restart(e.reset()), 0
        assert np.array_equal(out["actions"][:, i], np.array(acts, np.int8)), (cls, kw, hp, i)
        assert np.array_equal(np.asarray(Q[i], np.float64), np.asarray(host.Q, np.float64)) and np.array_equal(N[i], host.N), (cls, kw, hp)
    ag.close()
    env.close()


def run(seconds=None, n_cases=None, seed=0):
    """Runs for `seconds` or for exactly `n_cases` random MDPs; raises AssertionError on the first mismatch."""
    global rng
    rng = np.random.default_rng(seed)
    t_end = time.time() + (seconds or 1e9)
    done = 0
    while time.time() < t_end and (n_cases is None or done < n_cases):
        check_one()
        if done % 4 == 0:
            check_agent()
        done += 1
        if done % 10 == 0:
            print("fuzz: %d MDPs checked" % done, flush=True)   # progress (a silent run looks hung to the GPU runner)
    return done


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("fuzz parity: %d random MDPs, all checks bit-equal" % run(seconds=budget, seed=seed))
    print("rollout kernels exercised (1 K1, 2 K1L/K1P, 3 K1S, 4 K1T):", counts)

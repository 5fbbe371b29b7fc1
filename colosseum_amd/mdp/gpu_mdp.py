"""GpuMDP: one MDP instance behind the reference's `BaseMDP` call surface.

What agents, `MDPLoop`, `make_mdp_spec` and `random_loop` touch in the reference
(colosseum/experiment/agent_mdp_interaction.py:140-147,219,224,243-245,295-297,326-386,519-575;
colosseum/utils/acme/specs.py:29-40) exists here with the same names, argument meaning and error behaviour; every
`reset`/`step` is one call through the C ABI (`cmdp_reset`, `cmdp_step`) and every value function comes from the HIP
dynamic-programming kernels.  The per-(s,a) MT19937 sampler streams of the reference live on the device
(`CMDP_RNG_MT_COMPAT`), so trajectories are bit-identical to the reference for the same constructor arguments.

Class factories with the reference's class names are generated at the bottom (`DeepSeaEpisodic`, ...)."""
from typing import Dict, Tuple

import numpy as np

from .. import _lib as L
from .. import timestep as ts_
from ..batched import BatchedMDP
from ..dynamic_programming import get_policy_from_q_values
from .registry import make_model, split_class_name
from ..emission_maps import CompatNoise, observation_table
from .reward_sampler import CompatRewardSampler


class GpuMDP:
    _cls_name = None

    def __init__(self, cls_name: str = None, **kwargs):
        cls_name = cls_name or self._cls_name
        # emission map / noise: class objects or names ("StateInfo", "OneHotEncoding"; noise "GaussianUncorrelated"),
        # as BaseMDP.__init__ takes them (mdp/base.py:336-339,450-461); the noise seed is the MDP's seed
        em = kwargs.pop("emission_map", None)
        em_kwargs = kwargs.pop("emission_map_kwargs", {}) or {}
        noise = kwargs.pop("noise", None)
        noise_kwargs = dict(kwargs.pop("noise_kwargs", {}) or {})
        em_name = em if isinstance(em, str) or em is None else em.__name__
        noise_name = noise if isinstance(noise, str) or noise is None else noise.__name__
        self._model = make_model(cls_name, **kwargs)
        self._family_name, self._episodic = split_class_name(cls_name)
        m = self._model
        # Beta rewards are sampled on the host exactly as the reference does (reward_sampler.py); the device then
        # only reports the mean, which `step` replaces
        self._reward_sampler = None if m.deterministic_rewards else CompatRewardSampler(m)
        self._env = BatchedMDP([m], rng_mode=L.RNG_MT_COMPAT,
                               flags=0 if m.deterministic_rewards else L.FLAG_REWARD_MEANS)
        self.n_states, self.n_actions = m.n_states, m.n_actions
        self.rewards_range = self._rewards_range = tuple(m.rewards_range)
        self.r_min, self.r_max = self.rewards_range
        self.emission_map = None
        self.is_tabular = True
        # the observation table is built when the first observation is asked for, as in the reference
        # (`all_observations` is lazy, emission_maps/base.py:56-83): the StateLinear maps draw their features from the
        # global numpy stream at that moment, and the MiniGrid drawings show the state the MDP is in at that moment
        self._em_name = None if em_name in (None, "Tabular") else em_name
        self._em_kwargs = dict(em_kwargs)
        self._obs_table_cache = None
        self._noise = None
        self._noise_spec = None
        if self._em_name is not None:
            if self._em_name not in ("StateInfo", "OneHotEncoding", "TensorEncoding", "ImageEncoding", "StateLinearOptimal",
                                     "StateLinearRandom"):
                raise NotImplementedError(f"emission map {em_name!r} is not one of the reference's")
            self.emission_map = em_name
            self.is_tabular = False
            if noise_name is not None:
                if noise_name not in CompatNoise.KINDS:
                    raise NotImplementedError(f"noise {noise_name!r} is not one of the reference's")
                noise_kwargs.pop("seed", None)
                self._noise_spec = (noise_name, noise_kwargs)
        self._seed = kwargs.get("seed")
        self.parameters = dict(kwargs)
        nodes = [tuple(int(x) for x in n) for n in m.nodes]
        self.index_to_node = dict(enumerate(nodes))
        self.node_to_index = {n: i for i, n in enumerate(nodes)}
        self.starting_nodes = [self.index_to_node[int(s)] for s in m.start_states]
        self.starting_states = [int(s) for s in m.start_states]
        self.starting_state_distribution = np.zeros(self.n_states)
        self.starting_state_distribution[m.start_states] = m.start_probs
        self.starting_states_and_probs = list(zip(self.starting_states, m.start_probs.tolist()))
        self.h = 0
        self.cur_node = None
        self.last_starting_node = None
        self.last_edge = None
        self._cache: Dict[str, object] = {}
        if not self._episodic:
            self.random_policy = np.ones((self.n_states, self.n_actions), np.float32) / self.n_actions
        # `necessary_reset` is created by the first reset(), exactly like the reference (mdp/base.py:1272): step()
        # before reset() raises AttributeError there and here

    # -- static facts ---------------------------------------------------------------------------------------
    def is_episodic(self) -> bool:
        return self._episodic

    @property
    def H(self) -> int:
        if not self._episodic:
            raise AttributeError("H is defined for episodic MDPs only")
        return self._model.H

    @property
    def _obs_table(self):
        """`EmissionMap.all_observations`, built at the first observation (None for the tabular map)."""
        if self._em_name is None:
            return None
        if self._obs_table_cache is None:
            values = None
            if self._em_name == "StateLinearOptimal":
                values = self.optimal_value_functions[1]
            elif self._em_name == "StateLinearRandom":
                values = self.random_value_functions[1]
            cur = None if self.cur_node is None else self.node_to_index[self.cur_node]
            self._obs_table_cache = observation_table(self._model, self._em_name, cur_state=cur, h_now=self.h, values=values,
                                                      **self._em_kwargs)
            if self._noise_spec is not None:
                kind, kw = self._noise_spec
                shape = self._obs_table_cache.shape[2:] if self._episodic else self._obs_table_cache.shape[1:]
                self._noise = CompatNoise(self._seed, shape, kind=kind, **kw)
        return self._obs_table_cache

    def action_spec(self):
        return ts_.DiscreteArray(self.n_actions, name="action")

    def observation_spec(self):
        if self._obs_table is None:
            return ts_.DiscreteArray(self.n_states, name="observation")
        # the reference builds the spec from an actual observation of the first starting node (mdp/base.py:1249-1252),
        # which draws one noise sample every time it is called
        obs = self._observation(self.starting_states[0], 0)
        return ts_.BoundedArray(shape=obs.shape, dtype=obs.dtype, minimum=-np.inf, maximum=np.inf, name="observation")

    def _observation(self, state: int, h: int):
        """EmissionMap.get_observation (emission_maps/base.py:110-141)."""
        table = self._obs_table
        if self._episodic and h >= self._model.H:
            return np.zeros(table.shape[2:], np.float32)
        # continuous setting: the reference indexes `all_observations[None, state]` (in_episode_time = None is numpy's
        # newaxis, emission_maps/base.py:135-137), so its observations carry a leading axis of length 1
        obs = table[h, state] if self._episodic else table[None, state]
        return obs + next(self._noise) if self._noise is not None else obs

    def reward_spec(self):
        return ts_.Array(shape=(), dtype=float, name="reward")

    def discount_spec(self):
        return ts_.BoundedArray(shape=(), dtype=float, minimum=0.0, maximum=1.0, name="discount")

    # -- interaction (colosseum/mdp/base.py:1268-1355) -----------------------------------------------------------
    def reset(self):
        obs = int(self._env.reset()[0])
        self.necessary_reset = False
        self.h = 0
        self.cur_node = self.last_starting_node = self.index_to_node[obs]
        return ts_.restart(obs if self._obs_table is None else self._observation(obs, 0))

    def step(self, action, auto_reset=False):
        if auto_reset and self.necessary_reset:
            return self.reset()
        assert not self.necessary_reset  # AttributeError before the first reset(), as in the reference
        action = int(action)
        obs, rew, st = self._env.step([action], False)
        self.h += 1
        cur, _, _ = self._env.state()
        old = self.cur_node
        self.cur_node = self.index_to_node[int(cur[0])]
        self.last_edge = old, self.cur_node
        reward = float(rew[0])
        if self._reward_sampler is not None:
            reward = self._reward_sampler.sample(self.node_to_index[old], action, int(cur[0]))
        if self._obs_table is None:
            if st[0] == 2:
                self.necessary_reset = True
                return ts_.termination(reward=reward, observation=-1)
            return ts_.transition(reward=reward, observation=int(obs[0]))
        observation = self._observation(int(cur[0]), self.h)  # mdp/base.py:1308-1320
        if st[0] == 2:
            self.necessary_reset = True
            return ts_.termination(reward=reward, observation=np.zeros_like(self._spec_value()))
        return ts_.transition(reward=reward, observation=observation)

    def _spec_value(self):
        # `np.zeros_like(self.observation_spec().generate_value())`: building the spec draws a noise sample
        self.observation_spec()
        return np.zeros(self._obs_table.shape[2:] if self._episodic else (1,) + self._obs_table.shape[1:], np.float32)

    def random_step(self, auto_reset=False):
        """mdp/base.py:1341-1355: the action comes from the MDP's own numpy stream (shared with the reward caches),
        which the builder hands over positioned exactly as the reference's after construction."""
        action = int(self._model.extra["rng"].randint(self.n_actions))
        return self.step(action, auto_reset), action

    def get_visitation_counts(self, state_only=True):
        vs, vsa = self._env.visits()
        if state_only:
            return {self.index_to_node[i]: int(vs[i]) for i in range(self.n_states)}
        vsa = vsa.reshape(self.n_states, self.n_actions)
        return {(self.index_to_node[i], a): int(vsa[i, a]) for i in range(self.n_states) for a in range(self.n_actions)}

    def reset_visitation_counts(self):
        self._env.reset_visits()

    # -- tables and value functions ------------------------------------------------------------------------------
    @property
    def transition_matrix_and_rewards(self) -> Tuple[np.ndarray, np.ndarray]:
        return self._model.dense()

    @property
    def T(self):
        return self._model.dense()[0]

    @property
    def R(self):
        return self._model.dense()[1]

    def _memo(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    def _vi(self, R=None):
        if self._episodic:
            H, S, A = self.H, self.n_states, self.n_actions
            Q, V = self._env.episodic_value_iteration(R=None if R is None else [R])
            return Q.reshape(H + 1, S, A), V.reshape(H + 1, S)
        Q, V, _ = self._env.value_iteration(R=None if R is None else [R])  # gamma .99, eps 1e-3: reference defaults
        return Q.reshape(self.n_states, self.n_actions), V

    def _pe(self, policy):
        if self._episodic:
            H, S, A = self.H, self.n_states, self.n_actions
            Q, V = self._env.episodic_policy_evaluation([np.asarray(policy, np.float32)])
            return Q.reshape(H + 1, S, A), V.reshape(H + 1, S)
        Q, V, _ = self._env.policy_evaluation([np.asarray(policy, np.float32)])
        return Q.reshape(self.n_states, self.n_actions), V

    def get_value_functions(self, policy):
        return self._pe(policy)

    @property
    def optimal_value_functions(self):
        return self._memo("opt", self._vi)

    def get_optimal_policy(self, stochastic_form: bool):
        return self._memo(("opt_pi", stochastic_form),
                          lambda: get_policy_from_q_values(self.optimal_value_functions[0], stochastic_form))

    def get_worst_policy(self, stochastic_form: bool):
        return self._memo(("worst_pi", stochastic_form),
                          lambda: get_policy_from_q_values(self._vi(-self.R)[0], stochastic_form))

    @property
    def worst_value_functions(self):
        def f():
            pol = self.get_worst_policy(True)
            return self._pe(pol[: self.H] if self._episodic else pol)

        return self._memo("worst", f)

    @property
    def random_value_functions(self):
        def f():
            if self._episodic:
                pol = np.ones((self.H, self.n_states, self.n_actions), np.float32) / self.n_actions
            else:
                pol = self.random_policy
            return self._pe(pol)

        return self._memo("rand", f)

    # -- episodic baselines (colosseum/mdp/base_finite.py:210-253,339-375) ------------------------------------------
    def get_optimal_policy_starting_value(self, node):
        return self.optimal_value_functions[1][0, self.node_to_index[node]]

    def get_worst_policy_starting_value(self, node):
        return self.worst_value_functions[1][0, self.node_to_index[node]]

    def get_random_policy_starting_value(self, node):
        return self.random_value_functions[1][0, self.node_to_index[node]]

    def get_minimal_regret_for_starting_node(self, node):
        return self.get_optimal_policy_starting_value(node) - self.get_worst_policy_starting_value(node)

    def _episodic_average(self, getter):
        acc = 0.0
        for sn, p in zip(self.starting_nodes, self._model.start_probs.tolist()):
            acc += p * getter(sn)
        return acc / self.H

    @property
    def episodic_optimal_average_reward(self):
        return self._memo("eoar", lambda: self._episodic_average(self.get_optimal_policy_starting_value))

    @property
    def episodic_worst_average_reward(self):
        return self._memo("ewar", lambda: self._episodic_average(self.get_worst_policy_starting_value))

    @property
    def episodic_random_average_reward(self):
        return self._memo("erar", lambda: self._episodic_average(self.get_random_policy_starting_value))

    # -- Markov chains of the baseline policies, average rewards (colosseum/mdp/base.py:681-941) -------------------
    def get_stationary_distribution(self, policy):
        from .. import markov_chain as mc

        return mc.get_stationary_distribution(mc.get_transition_probabilities(self.T, policy),
                                              self.starting_states_and_probs)

    def get_average_reward(self, policy):
        from .. import markov_chain as mc

        return sum(self.get_stationary_distribution(policy) * mc.get_average_rewards(self.R, policy))

    def _baseline(self, which):
        """(transition probabilities, stationary distribution, per-state average rewards, average reward) of the
        optimal / worst / uniform policy; for episodic MDPs the reference evaluates them on the continuous form,
        which is not built (not used by the episodic MDPLoop)."""
        from .. import markov_chain as mc

        if self._episodic:
            raise NotImplementedError("average rewards of episodic MDPs (continuous form chains) are not built")

        def f():
            pi = {"optimal": lambda: self.get_optimal_policy(True), "worst": lambda: self.get_worst_policy(True),
                  "random": lambda: self.random_policy}[which]()
            tps = mc.get_transition_probabilities(self.T, pi)
            sd = mc.get_stationary_distribution(tps, None if which == "random" else self.starting_states_and_probs)
            ars = mc.get_average_rewards(self.R, pi)
            return tps, sd, ars, sum(sd * ars)

        return self._memo(("baseline", which), f)

    optimal_transition_probabilities = property(lambda self: self._baseline("optimal")[0])
    worst_transition_probabilities = property(lambda self: self._baseline("worst")[0])
    random_transition_probabilities = property(lambda self: self._baseline("random")[0])
    optimal_stationary_distribution = property(lambda self: self._baseline("optimal")[1])
    worst_stationary_distribution = property(lambda self: self._baseline("worst")[1])
    random_stationary_distribution = property(lambda self: self._baseline("random")[1])
    optimal_average_rewards = property(lambda self: self._baseline("optimal")[2])
    worst_average_rewards = property(lambda self: self._baseline("worst")[2])
    random_average_rewards = property(lambda self: self._baseline("random")[2])
    optimal_average_reward = property(lambda self: self._baseline("optimal")[3])
    worst_average_reward = property(lambda self: self._baseline("worst")[3])
    random_average_reward = property(lambda self: self._baseline("random")[3])

    # -- hardness (continuous setting; colosseum/mdp/base.py:996-1016,1060-1081) -----------------------------------
    @property
    def diameter(self):
        if self._episodic:
            return self._memo("diam", lambda: float(self._env.diameter_episodic()[0][0]))
        return self._memo("diam", lambda: float(self._env.diameter()[0][0]))

    @property
    def value_norm(self):
        from ..hardness import value_norm

        return self._memo("vnorm", lambda: float(value_norm([self._model])[0]))

    def close(self):
        self._env.close()


def _make_class(name):
    return type(name, (GpuMDP,), {"_cls_name": name, "__doc__": f"`{name}` with the reference's constructor keywords."})


DeepSeaEpisodic = _make_class("DeepSeaEpisodic")
DeepSeaContinuous = _make_class("DeepSeaContinuous")
FrozenLakeEpisodic = _make_class("FrozenLakeEpisodic")
FrozenLakeContinuous = _make_class("FrozenLakeContinuous")
MiniGridEmptyEpisodic = _make_class("MiniGridEmptyEpisodic")
MiniGridEmptyContinuous = _make_class("MiniGridEmptyContinuous")
MiniGridRoomsEpisodic = _make_class("MiniGridRoomsEpisodic")
MiniGridRoomsContinuous = _make_class("MiniGridRoomsContinuous")
RiverSwimEpisodic = _make_class("RiverSwimEpisodic")
RiverSwimContinuous = _make_class("RiverSwimContinuous")
SimpleGridEpisodic = _make_class("SimpleGridEpisodic")
SimpleGridContinuous = _make_class("SimpleGridContinuous")
TaxiEpisodic = _make_class("TaxiEpisodic")
TaxiContinuous = _make_class("TaxiContinuous")
CustomEpisodic = _make_class("CustomEpisodic")
CustomContinuous = _make_class("CustomContinuous")

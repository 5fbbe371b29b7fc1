"""Family definitions for the four in-scope Colosseum MDP families.

Each family states, over plain integer tuples, exactly what the reference's
family class feeds into the generic graph construction: the start-node layout
(and the random draws it consumes), the successor rule of one *raw* action, and
the reward-distribution rule.  Reference anchors:

  DeepSea        colosseum/mdp/deep_sea/base.py:249-287
  FrozenLake     colosseum/mdp/frozen_lake/base.py:145-182,285-311
  MiniGridEmpty  colosseum/mdp/minigrid_empty/base.py:153-235
  MiniGridRooms  colosseum/mdp/minigrid_rooms/base.py:165-269
  RiverSwim      colosseum/mdp/river_swim/base.py:152-284          
  SimpleGrid     colosseum/mdp/simple_grid/base.py:172-268,300-414
  Taxi           colosseum/mdp/taxi/base.py:124-199,229-330,380-494

Reward distributions are ("deterministic", loc) or ("beta", a, b) tuples, the two
kinds the reference constructs (colosseum/utils/miscellanea.py:253-270).
"""
from itertools import product
from typing import List, Optional, Sequence, Tuple

import numpy as np

Node = Tuple[int, ...]
Dist = Tuple  # ("deterministic", loc) | ("beta", a, b)


def _dist(spec) -> Dist:
    name, args = spec
    if name == "deterministic":
        return ("deterministic", float(args[0]))
    if name == "beta":
        return ("beta", float(args[0]), float(args[1]))
    raise ValueError(f"unsupported reward distribution {name!r}")


def dist_mean(d: Dist) -> float:
    """Mean as scipy's frozen distribution reports it (deterministic: loc; beta: a/(a+b))."""
    if d[0] == "deterministic":
        return d[1]
    a, b = d[1], d[2]
    return a * 1.0 / (a + b)


class StartSpec:
    """What the reference's `_get_starting_node_sampler` returns: nodes, probabilities and whether the
    sampler is handed a seed drawn from `_fast_rng` (only then is that stream advanced)."""

    def __init__(self, nodes: List[Node], probs: Optional[List[float]], wants_seed: bool):
        self.nodes, self.probs, self.wants_seed = nodes, probs, wants_seed


class Family:
    name = ""
    n_actions = 0
    episodic_H_is_size = False

    def check(self, p_lazy, p_rand):
        pass

    def start(self, rng: np.random.RandomState, fast_rng) -> StartSpec:
        raise NotImplementedError

    def next_nodes(self, node: Node, action: int) -> Sequence[Tuple[Node, float]]:
        raise NotImplementedError

    def reward_dist(self, node: Node, action: int, next_node: Node) -> Dist:
        raise NotImplementedError

    def possible_starting_nodes(self) -> List[Node]:
        raise NotImplementedError


# ------------------------------------------------------------------------------------------------------
class DeepSea(Family):
    """deep_sea/base.py.  Node = (X, Y); actions LEFT=0, RIGHT=1."""

    name = "DeepSea"
    n_actions = 2

    def __init__(self, size, optimal_return=1.0, suboptimal_return=0.5, optimal_distribution=None,
                 sub_optimal_distribution=None, other_distribution=None, make_reward_stochastic=False,
                 reward_variance_multiplier=1.0):
        assert size > 1
        assert suboptimal_return < optimal_return - 0.1
        self.size = size
        given = [sub_optimal_distribution, optimal_distribution, other_distribution]
        assert given.count(None) in (0, 3)
        if given.count(None) == 0:
            self.sub, self.opt, self.other = (_dist(d) for d in given)
        elif make_reward_stochastic:
            m = reward_variance_multiplier
            self.sub = ("beta", m, m * (size / suboptimal_return - 1))
            self.opt = ("beta", m * (size / optimal_return - 1), m)
            self.other = ("beta", m, m * 10 * (size / suboptimal_return - 1))
        else:
            self.sub = ("deterministic", 1.0 / (size ** 2))
            self.opt = ("deterministic", 1.0)
            self.other = ("deterministic", 0.0)

    def check(self, p_lazy, p_rand):
        assert p_lazy is None, "No lazy mechanic for DeepSea"  # deep_sea/base.py:294-295

    def possible_starting_nodes(self):
        return [(0, self.size - 1)]

    def start(self, rng, fast_rng):
        return StartSpec(self.possible_starting_nodes(), None, False)

    def next_nodes(self, node, action):
        x, y = node
        if y == 0:
            return (((0, self.size - 1), 1.0),)
        nx = min(x + 1, self.size - 1) if action == 1 else max(x - 1, 0)
        return (((nx, max(0, y - 1)), 1.0),)

    def reward_dist(self, node, action, next_node):
        if node[0] == self.size - 1 and node[1] == 0 and action == 1:
            return self.opt
        return self.sub if action == 0 else self.other


# ------------------------------------------------------------------------------------------------------
def generate_random_lake(seed: int, size: int, p: float) -> np.ndarray:
    """gym (<=0.25, un-vendored, unpinned: reference setup.py:27) `generate_random_map` as the reference
    calls it after `np.random.seed(seed)` (frozen_lake/base.py:285-293): tiles 'F'/'H' drawn with
    probabilities (p, 1-p) from the global legacy generator, corners forced to 'S'/'G', redrawn until a
    DFS over non-hole tiles reaches 'G'.  Returns a (size, size) array of single characters."""
    rs = np.random.RandomState(seed)  # == np.random.seed(seed) + global draws
    p = min(1, p)
    while True:
        grid = rs.choice(["F", "H"], (size, size), p=[p, 1 - p])
        grid[0][0] = "S"
        grid[-1][-1] = "G"
        stack, seen, ok = [(0, 0)], set(), False
        while stack and not ok:
            r, c = stack.pop()
            if (r, c) in seen:
                continue
            seen.add((r, c))
            for dr, dc in ((1, 0), (0, 1), (-1, 0), (0, -1)):
                rr, cc = r + dr, c + dc
                if rr < 0 or rr >= size or cc < 0 or cc >= size:
                    continue
                if grid[rr][cc] == "G":
                    ok = True
                    break
                if grid[rr][cc] != "H":
                    stack.append((rr, cc))
        if ok:
            return np.array([list("".join(row)) for row in grid])


class FrozenLake(Family):
    """frozen_lake/base.py.  Node = (X, Y); actions UP=0, RIGHT=1, DOWN=2, LEFT=3."""

    name = "FrozenLake"
    n_actions = 4

    def __init__(self, seed, size, p_frozen, optimal_return=1.0, suboptimal_return=0.1, is_slippery=True,
                 goal_r=None, default_r=None, make_reward_stochastic=False, reward_variance_multiplier=1.0):
        assert p_frozen >= 0.1 and size > 2
        assert suboptimal_return + 0.2 < optimal_return
        self.size, self.is_slippery = size, is_slippery
        self.lake = generate_random_lake(seed, size, p_frozen)
        given = [default_r, goal_r]
        assert given.count(None) in (0, 2)
        if given.count(None) == 0:
            self.default_r, self.goal_r = _dist(default_r), _dist(goal_r)
        elif make_reward_stochastic:
            m = reward_variance_multiplier
            self.default_r = ("beta", m, m * (size ** 2 / suboptimal_return - 1))
            self.goal_r = ("beta", m * (size ** 2 / optimal_return - 1), m)
        else:
            self.default_r = ("deterministic", 0.0)
            self.goal_r = ("deterministic", 1.0)

    def possible_starting_nodes(self):
        return [(0, 0)]

    def start(self, rng, fast_rng):
        return StartSpec([(0, 0)], None, False)

    def _next_position(self, x, y, a):
        lake, n = self.lake, self.size
        if lake[x, y] == "G":
            return (0, 0)
        if a == 3:
            nx, ny = x, min(y + 1, n - 1)
        elif a == 2:
            nx, ny = min(x + 1, n - 1), y
        elif a == 1:
            nx, ny = x, max(y - 1, 0)
        else:
            nx, ny = max(x - 1, 0), y
        if lake[nx, ny] == "H":
            return (0, 0)
        return (nx, ny)

    def next_nodes(self, node, action):
        x, y = node
        p = 0.5 if self.is_slippery else 1.0
        out = [(self._next_position(x, y, action), p)]
        if self.is_slippery:
            for a in ((action - 1) % 4, (action + 1) % 4):
                out.append((self._next_position(x, y, a), p / 2))
        return out

    def reward_dist(self, node, action, next_node):
        return self.goal_r if self.lake[next_node[0], next_node[1]] == "G" else self.default_r


# ------------------------------------------------------------------------------------------------------
class _MiniGrid(Family):
    n_actions = 3  # MoveForward=0, TurnRight=1, TurnLeft=2; Dir UP=0, RIGHT=1, DOWN=2, LEFT=3

    def _set_reward_dists(self, n_cells_minus_1, optimal_distribution, other_distribution,
                          make_reward_stochastic, m):
        given = [optimal_distribution, other_distribution]
        assert given.count(None) in (0, 2)
        if given.count(None) == 0:
            self.opt, self.other = _dist(optimal_distribution), _dist(other_distribution)
        elif make_reward_stochastic:
            self.other = ("beta", m, m * n_cells_minus_1)
            self.opt = ("beta", m * n_cells_minus_1, m)
        else:
            self.opt = ("deterministic", 1.0)
            self.other = ("deterministic", 0.0)

    def reward_dist(self, node, action, next_node):
        g = self.goal_position
        return self.opt if (next_node[0] == g[0] and next_node[1] == g[1]) else self.other


class MiniGridEmpty(_MiniGrid):
    """minigrid_empty/base.py.  Node = (X, Y, Dir)."""

    name = "MiniGridEmpty"

    def __init__(self, size, n_starting_states=1, optimal_distribution=None, other_distribution=None,
                 make_reward_stochastic=False, reward_variance_multiplier=1.0):
        assert size > 2 and n_starting_states > 0
        self.size, self.n_starting_states = size, n_starting_states
        self._set_reward_dists(size ** 2 - 1, optimal_distribution, other_distribution,
                               make_reward_stochastic, reward_variance_multiplier)

    def positions_on_side(self, side):
        n, out = self.size, []
        for i in range(n):
            for j in range(n):
                if side == 0:
                    out.append((i, j))
                elif side == 1:
                    out.append((j, i))
                elif side == 2:
                    out.append((n - 1 - i, n - 1 - j))
                else:
                    out.append((n - 1 - j, n - 1 - i))
        return out

    def start(self, rng, fast_rng):
        n = self.size
        self.side_start = int(rng.randint(4))
        self.goal_position = self.positions_on_side((self.side_start + 2) % 4)[:n][int(rng.randint(n))]
        self._start_positions = self.positions_on_side(self.side_start)[:n]
        rng.shuffle(self._start_positions)
        chosen = self._start_positions[: self.n_starting_states]
        nodes = [(x, y, int(rng.randint(4))) for x, y in chosen]
        return StartSpec(nodes, [1 / len(chosen) for _ in chosen], True)

    def possible_starting_nodes(self):
        return [(x, y, d) for (x, y), d in product(self._start_positions, range(4))]

    def next_nodes(self, node, action):
        x, y, d = node
        n = self.size
        if action == 1:
            return (((x, y, (d + 1) % 4), 1.0),)
        if action == 2:
            return (((x, y, (d - 1) % 4), 1.0),)
        if d == 0:
            return (((x, min(y + 1, n - 1), d), 1.0),)
        if d == 1:
            return (((min(n - 1, x + 1), y, d), 1.0),)
        if d == 2:
            return (((x, max(y - 1, 0), d), 1.0),)
        return (((max(0, x - 1), y, d), 1.0),)


class MiniGridRooms(_MiniGrid):
    """minigrid_rooms/base.py.  Node = (X, Y, Dir)."""

    name = "MiniGridRooms"

    def __init__(self, room_size, n_rooms=4, n_starting_states=2, optimal_distribution=None,
                 other_distribution=None, make_reward_stochastic=False, reward_variance_multiplier=1.0):
        assert n_rooms >= 4 and room_size >= 2 and n_starting_states > 0
        assert int(np.sqrt(n_rooms)) == np.sqrt(n_rooms), "Please provide a number of rooms with perfect square."
        self.room_size, self.n_rooms, self.n_starting_states = room_size, n_rooms, n_starting_states
        size = int(room_size * n_rooms ** 0.5)  # minigrid_rooms/base.py:411
        self._set_reward_dists(size ** 2 - 1, optimal_distribution, other_distribution,
                               make_reward_stochastic, reward_variance_multiplier)
        # the reference rebuilds this list on every forward move (base.py:218); it is a pure function of
        # (room_size, n_rooms), so a set built once is equivalent
        rpr = int(np.sqrt(n_rooms))
        vertical = [j * room_size + j + int(np.floor(room_size / 2)) for j in range(rpr)]
        horizontal = [j * room_size + j - 1 for j in range(1, rpr)]
        cells = set(product(horizontal, vertical)) | set(product(vertical, horizontal))
        for rc in product(range(rpr), range(rpr)):
            cells.update(self.room_cells(rc))
        self.admissible = cells

    def room_cells(self, room_coord):
        """`get_positions_coords_in_room(...).ravel().tolist()` order: rows of decreasing j, increasing i."""
        rs = self.room_size
        xr, yr = room_coord
        return [(i + (rs + 1) * xr, j + (rs + 1) * yr) for j in range(rs - 1, -1, -1) for i in range(rs)]

    def start(self, rng, fast_rng):
        corners = list(product((0, int(self.n_rooms ** 0.5) - 1), repeat=2))
        sr = fast_rng.randint(0, len(corners) - 1)
        self.starting_room = corners.pop(sr)
        self.goal_room = corners[fast_rng.randint(0, len(corners) - 1)]
        goal_positions = self.room_cells(self.goal_room)
        rng.shuffle(goal_positions)
        self.goal_position = goal_positions[0]
        starting = [(x, y, d) for x, y in self.room_cells(self.starting_room) for d in range(4)]
        rng.shuffle(starting)
        self._possible = starting
        k = self.n_starting_states
        return StartSpec(starting[:k], [1 / k for _ in range(k)], True)

    def possible_starting_nodes(self):
        return self._possible

    def next_nodes(self, node, action):
        x, y, d = node
        if action == 1:
            return (((x, y, (d + 1) % 4), 1.0),)
        if action == 2:
            return (((x, y, (d - 1) % 4), 1.0),)
        if d == 0:
            nc = (x, y + 1)
        elif d == 1:
            nc = (x + 1, y)
        elif d == 2:
            nc = (x, y - 1)
        else:
            nc = (x - 1, y)
        if nc in self.admissible:
            return (((nc[0], nc[1], d), 1.0),)
        return ((node, 1.0),)


# ------------------------------------------------------------------------------------------------------
def _three_dists(given, stochastic, betas, deterministic):
    """The families' common rule: all three distributions given -> taken as they are; otherwise Beta parameters
    when make_reward_stochastic, else the family's deterministic defaults."""
    assert given.count(None) in (0, 3)
    if given.count(None) == 0:
        return tuple(_dist(d) for d in given)
    if stochastic:
        return tuple(("beta", float(a), float(b)) for a, b in betas)
    return tuple(("deterministic", float(v)) for v in deterministic)


class RiverSwim(Family):
    """river_swim/base.py.  Node = (X,); actions LEFT=0, RIGHT=1; the chain is deterministic, its stochasticity is the
    base class's p_rand / p_lazy."""

    name = "RiverSwim"
    n_actions = 2

    def __init__(self, size, optimal_mean_reward=0.9, sub_optimal_mean_reward=0.2, sub_optimal_distribution=None,
                 optimal_distribution=None, other_distribution=None, make_reward_stochastic=False,
                 reward_variance_multiplier=1.0, episodic=False):
        assert size > 1 and optimal_mean_reward - 0.1 > sub_optimal_mean_reward  # :262-263 (on the given means)
        self.size = size
        m = reward_variance_multiplier
        sub_mean = sub_optimal_mean_reward / size if episodic else sub_optimal_mean_reward  # :214-215
        self.sub, self.opt, self.other = _three_dists(
            [sub_optimal_distribution, optimal_distribution, other_distribution], make_reward_stochastic,
            [(m, m * (1 / sub_mean - 1)), (m, m * (1 / optimal_mean_reward - 1)), (m, m * (10 / sub_mean - 1))],
            [5 / 1000, 1.0, 0.0])

    def possible_starting_nodes(self):
        return [(0,)]

    def start(self, rng, fast_rng):
        return StartSpec(self.possible_starting_nodes(), None, False)

    def next_nodes(self, node, action):
        x = node[0]
        return (((min(x + 1, self.size - 1) if action == 1 else max(x - 1, 0),), 1.0),)

    def reward_dist(self, node, action, next_node):
        if node[0] == self.size - 1 and action == 1:
            return self.opt
        return self.sub if node[0] == 0 and action == 0 else self.other


class SimpleGrid(Family):
    """simple_grid/base.py.  Node = (X, Y); actions UP, RIGHT, DOWN, LEFT, NO_OP; the corners' self-loops carry the
    AND / NAND / OR / XOR reward."""

    name = "SimpleGrid"
    n_actions = 5
    AND, NAND, OR, XOR = 0, 1, 2, 3

    def __init__(self, size, reward_type=3, n_starting_states=1, optimal_mean_reward=0.9, sub_optimal_mean_reward=0.2,
                 optimal_distribution=None, sub_optimal_distribution=None, other_distribution=None,
                 make_reward_stochastic=False, reward_variance_multiplier=1.0):
        assert n_starting_states <= (size - 1) ** 2 and optimal_mean_reward - 0.1 > sub_optimal_mean_reward
        if isinstance(reward_type, str):  # the enum member's name, as the parameter hash / gin files spell it
            reward_type = {"AND": 0, "NAND": 1, "OR": 2, "XOR": 3}[reward_type.split(".")[-1]]
        self.size, self.reward_type, self.n_starting_states = size, int(reward_type), n_starting_states
        m = reward_variance_multiplier
        self.sub, self.opt, self.other = _three_dists(
            [sub_optimal_distribution, optimal_distribution, other_distribution], make_reward_stochastic,
            [(m, m * (10 / sub_optimal_mean_reward - 1)), (m, m * (1 / optimal_mean_reward - 1)),
             (m, m * (1 / sub_optimal_mean_reward - 1))],
            [0.0, 1.0, 0.5])

    def _starting_nodes_by_distance(self, rng):
        """`_calculate_starting_nodes` (:225-241): cells by distance from the centre, ties in np.where order, only the
        innermost batch shuffled."""
        n = self.size
        center = np.array(((n - 1) / 2, (n - 1) / 2))
        distances = np.empty((n, n))
        for x in range(n):
            for y in range(n):
                distances[x, y] = ((np.array((x, y)) - center) ** 2).sum()
        batch = np.array(np.where(distances == distances.min())).T.tolist()
        rng.shuffle(batch)
        while not np.all(distances == np.inf):
            distances[batch[0][0], batch[0][1]] = np.inf
            yield batch[0]
            batch.pop(0)
            if len(batch) == 0:
                batch = np.array(np.where(distances == distances.min())).T.tolist()

    def start(self, rng, fast_rng):
        it = self._starting_nodes_by_distance(rng)
        self._possible = [tuple(int(v) for v in next(it)) for _ in range((self.size - 1) ** 2)]
        chosen = self._possible[: self.n_starting_states]
        rng.shuffle(chosen)
        if len(chosen) == 1:
            return StartSpec(chosen, None, False)
        return StartSpec(chosen, [1 / self.n_starting_states for _ in range(self.n_starting_states)], True)

    def possible_starting_nodes(self):
        return list(self._possible)

    def next_nodes(self, node, action):
        x, y = node
        n = self.size
        if action == 0:
            return (((x, min(y + 1, n - 1)), 1.0),)
        if action == 1:
            return (((min(x + 1, n - 1), y), 1.0),)
        if action == 2:
            return (((x, max(y - 1, 0)), 1.0),)
        if action == 3:
            return (((max(x - 1, 0), y), 1.0),)
        return (((x, y), 1.0),)

    def reward_dist(self, node, action, next_node):
        x, y = node
        corner = node == next_node and x in (0, self.size - 1) and y in (0, self.size - 1)
        if not corner:
            return self.other
        t = self.reward_type
        good = ((t == self.AND and (x and y)) or (t == self.NAND and not (x and y)) or (t == self.OR and (x | y))
                or (t == self.XOR and (x ^ y)))
        return self.opt if good else self.sub


class Taxi(Family):
    """taxi/base.py.  Node = (X, Y, XPass, YPass, XDest, YDest); actions South, North, East, West, PickUp, DropOff.
    The reference forces randomize_actions=False for this family (:488-490)."""

    name = "Taxi"
    n_actions = 6
    force_randomize_actions = False

    def __init__(self, size, length=2, width=1, space=1, n_locations=2 ** 2, optimal_mean_reward=0.9,
                 sub_optimal_mean_reward=0.2, default_r=None, successfully_delivery_r=None, failure_delivery_r=None,
                 make_reward_stochastic=False, reward_variance_multiplier=1.0, episodic=False):
        self.size, self.length, self.width, self.space = size, length, width, space
        self.n_locations_arg = n_locations
        self.n_locations = int(np.ceil(n_locations ** 0.5) ** 2)
        m = reward_variance_multiplier
        self.default_r, self.success_r, self.failure_r = _three_dists(
            [default_r, successfully_delivery_r, failure_delivery_r], make_reward_stochastic,
            [(m, m * (1 / sub_optimal_mean_reward - 1)), (m, m * (1 / optimal_mean_reward - 1)),
             (m, m * (10 / sub_optimal_mean_reward - 1))],
            [0.1, 1, 0])
        assert dist_mean(self.failure_r) < dist_mean(self.default_r) < dist_mean(self.success_r)  # :349-353
        assert size > 3 and n_locations > (1 if episodic else 2)
        assert size > length and size > width and size > space / 2 and size > 2 * n_locations ** 0.5
        assert optimal_mean_reward - 0.1 > sub_optimal_mean_reward
        self.admissible = self._admissible_coordinate()
        self._adm_set = {tuple(c) for c in self.admissible}
        self._locations = None

    def _admissible_coordinate(self):
        """:128-155, the wall pattern; cells with 0 are free."""
        n, rows, j = self.size, [], 0
        while len(rows) < n:
            row = [] if j % 2 != 0 else [0] * int((self.width + self.space) // 2)
            i = 0
            while len(row) < n:
                row.append(int(i % (1 + self.space) == 0))
                if row[-1] == 1:
                    for _ in range(self.width - 1):
                        if len(row) == n:
                            break
                        row.append(1)
                i += 1
            for _ in range(self.length):
                if len(rows) == n:
                    break
                rows.append(row)
            if len(rows) < n:
                rows.append([0] * n)
            j += 1
        return np.vstack(np.where(np.array(rows) == 0)).T.tolist()

    def _quadrants(self):
        n, k = self.size, int(self.n_locations ** 0.5)
        quadrants = np.zeros((n, n))
        split = np.array_split(range(n), k)
        for i, (xs, ys) in enumerate(product(split, split)):
            for qx, qy in product(xs, ys):
                quadrants[qx, qy] = i
        out = [[c for c in np.vstack(np.where(quadrants == i)).T.tolist() if tuple(c) in self._adm_set]
               for i in range(self.n_locations)]
        assert all(len(q) != 0 for q in out)
        return out

    def locations(self, rng):
        """:181-199: one location per quadrant, re-drawn until all are further apart than the quadrant width, then
        shuffled and cut to the requested number."""
        if self._locations is None:
            quadrants = self._quadrants()
            min_distance = max(self.size / int(self.n_locations ** 0.5) / 2, 2)
            while True:
                locs = [quadrants[i][rng.randint(len(quadrants[i]))] for i in range(self.n_locations)]
                npl = np.array(locs)
                again = False
                for i in range(self.n_locations):
                    for j in range(1 + i, self.n_locations):
                        if np.sqrt(((npl[i] - npl[j]) ** 2).sum()) <= min_distance:
                            again = True
                            break
                    if again:
                        break
                if not again:
                    break
            rng.shuffle(locs)
            self._locations = [tuple(int(v) for v in c) for c in locs[: self.n_locations_arg]]
        return self._locations

    def start(self, rng, fast_rng):
        locs = self.locations(rng)
        nodes = []
        for (px, py), (dx, dy), (tx, ty) in product(locs, locs, self.admissible):
            if (px, py) == (dx, dy):
                continue
            nodes.append((int(tx), int(ty), px, py, dx, dy))
        rng.shuffle(nodes)
        self._start_nodes = nodes
        return StartSpec(nodes, [1 / len(nodes) for _ in nodes], True)

    def possible_starting_nodes(self):
        return list(self._start_nodes)

    def next_nodes(self, node, action):
        x, y, xp, yp, xd, yd = node
        locs = self._locations
        if action == 5 and xp == -1 and x == xd and y == yd:
            # successful drop-off: a new passenger anywhere but here, a destination anywhere but at the passenger
            pairs = [(pl, de) for pl in locs if pl != (x, y) for de in locs if de != pl]
            p = 1.0 / len(pairs)
            return tuple(((x, y, pl[0], pl[1], de[0], de[1]), p) for pl, de in pairs)
        if action == 4 and xp != -1 and x == xp and y == yp:
            xp, yp = -1, -1
        if action == 1:
            nc = (x, y + 1)
        elif action == 2:
            nc = (x + 1, y)
        elif action == 0:
            nc = (x, y - 1)
        elif action == 3:
            nc = (x - 1, y)
        else:
            nc = (x, y)
        if nc in self._adm_set:
            x, y = nc
        return (((x, y, xp, yp, xd, yd), 1.0),)

    def reward_dist(self, node, action, next_node):
        if action == 4 and (next_node[2] != -1 or node[2] == -1):
            return self.failure_r
        if action == 5:
            if next_node[2] == -1 or node[2] != -1:
                return self.failure_r
            if node[2] == -1 and next_node[2] != -1:
                return self.success_r
        return self.default_r


class Custom(Family):
    """mdp/custom_mdp.py:80-104,184-232.  Node = (ID,); the user gives the start distribution T_0 (dict state -> prob,
    or an array), the transition array T [S, A, S] and the rewards R: a dict (state, action) -> distribution (scipy
    frozen distribution or ("deterministic", (v,)) / ("beta", (a, b)) tuple) or an [S, A] array.

    As in the reference every reward is DETERMINISTIC and equal to the distribution's mean (its `R` property returns
    the mean matrix, so the `type(self.R) == dict` branch of `_get_reward_distribution` never runs, :96-100).  Deviation:
    the reference's array-R constructor raises (`_R` is unbound, :207-211); here an array R means what the docstring
    says.  T is taken as float64 (a float32 T would make the reference's samplers accumulate in float32)."""

    name = "Custom"

    def __init__(self, T_0, T, R):
        self.T = np.asarray(T, np.float64)
        assert self.T.ndim == 3 and self.T.shape[0] == self.T.shape[2]
        S, A = self.T.shape[:2]
        self.n_actions = A
        if isinstance(R, dict):
            Rm = np.zeros((S, A), np.float32)
            for (s_, a_), d in R.items():
                if isinstance(d, tuple):
                    Rm[s_, a_] = dist_mean(_dist(d))
                else:
                    Rm[s_, a_] = d.mean()
        else:
            Rm = np.asarray(R, np.float32)
            assert Rm.shape == (S, A)
        self.R = Rm
        if isinstance(T_0, dict):
            self.T_0 = {int(k): float(v) for k, v in T_0.items()}
        else:
            self.T_0 = {i: float(p) for i, p in enumerate(np.asarray(T_0)) if p > 0}
        assert np.isclose(sum(self.T_0.values()), 1)
        for s_ in range(S):
            for a_ in range(A):
                assert np.isclose(self.T[s_, a_].sum(), 1), (
                    f"The transition kernel associated with state {s_} and action {a_} is not a well defined "
                    f"probability distribution.")

    def possible_starting_nodes(self):
        return [(k,) for k in self.T_0]

    def start(self, rng, fast_rng):
        return StartSpec(self.possible_starting_nodes(), list(self.T_0.values()), True)

    def next_nodes(self, node, action):
        row = self.T[node[0], action]
        return tuple(((int(j),), float(row[j])) for j in range(len(row)) if row[j] > 0.0)

    def reward_dist(self, node, action, next_node):
        return ("deterministic", float(self.R[node[0], action]))

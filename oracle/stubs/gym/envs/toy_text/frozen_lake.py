"""Restatement of gym's `generate_random_map` (gym <= 0.25, the un-vendored and
unpinned dependency at reference colosseum/mdp/frozen_lake/base.py:7,285-293).
Published algorithm: draw a size x size grid of 'F'/'H' tiles from the GLOBAL
numpy generator with probabilities (p, 1-p), force 'S' at [0][0] and 'G' at
[-1][-1], accept the grid iff a depth-first search from (0,0) over non-hole
tiles reaches 'G', else redraw.  Test infrastructure only."""
import numpy as np


def _goal_reachable(grid, size):
    stack, seen = [(0, 0)], set()
    while stack:
        r, c = stack.pop()
        if (r, c) in seen:
            continue
        seen.add((r, c))
        for dr, dc in ((1, 0), (0, 1), (-1, 0), (0, -1)):
            rr, cc = r + dr, c + dc
            if rr < 0 or rr >= size or cc < 0 or cc >= size:
                continue
            if grid[rr][cc] == "G":
                return True
            if grid[rr][cc] != "H":
                stack.append((rr, cc))
    return False


def generate_random_map(size=8, p=0.8):
    while True:
        p = min(1, p)
        grid = np.random.choice(["F", "H"], (size, size), p=[p, 1 - p])
        grid[0][0] = "S"
        grid[-1][-1] = "G"
        if _goal_reachable(grid, size):
            return ["".join(row) for row in grid]

"""Continuous form of an episodic MDP (state space augmented with the in-episode time step).

Restates, on indices, `get_episodic_graph` (colosseum/mdp/utils/mdp_creation.py:179-209) and
`get_continuous_form_episodic_transition_matrix_and_rewards` (:131-176), which the reference uses for the value
norm of episodic MDPs (colosseum/mdp/base.py:1042-1059: `T_cf`, `R_cf`, `optimal_value_continuous_form`).
The augmented node order is the insertion order of the reference's recursive graph walk (iterative here)."""
from typing import Dict, List, Tuple

import numpy as np

from .builder import TabularModel


def episodic_graph_nodes(model: TabularModel) -> Tuple[List[Tuple[int, int]], Dict[Tuple[int, int], List[Tuple[int, int]]]]:
    """(nodes in insertion order, successors per node in insertion order) of `get_episodic_graph(..., True)`."""
    H = model.H
    succ_of = model.extra["successors"]
    starts = [int(s) for s in model.start_states]
    nodes: Dict[Tuple[int, int], None] = {}
    adj: Dict[Tuple[int, int], Dict[Tuple[int, int], None]] = {}

    for sn in starts:
        # frames: [state, h, successor list, position] -- iterative form of add_successors(n, h)
        stack = [[sn, 0, None, 0]]
        while stack:
            fr = stack[-1]
            n, h = fr[0], fr[1]
            if fr[2] is None:
                fr[2] = succ_of[n] if h < H - 1 else starts
            if fr[3] == len(fr[2]):
                stack.pop()
                continue
            succ = fr[2][fr[3]]
            fr[3] += 1
            next_h = h + 1 if h + 1 != H else 0
            u, v = (h, n), (next_h, succ)
            nodes.setdefault(u, None)
            nodes.setdefault(v, None)
            adj.setdefault(u, {}).setdefault(v, None)
            if h < H - 1 and len(adj.get(v, ())) == 0:
                stack.append([succ, next_h, None, 0])
    return list(nodes), {k: list(v) for k, v in adj.items()}


def continuous_form(model: TabularModel):
    """(N, A, (ptr, col, val), R_cf): CSR of the reference's `T_cf` (non-zeros, ascending column) and `R_cf`."""
    assert model.is_episodic
    H, A = model.H, model.n_actions
    T, R = model.dense()
    nodes, adj = episodic_graph_nodes(model)
    idx = {n: i for i, n in enumerate(nodes)}
    N = len(nodes)
    ptr = np.zeros(N * A + 1, np.int64)
    cols, vals = [], []
    R_cf = np.zeros((N, A), np.float32)
    start = [(int(s), np.float32(p)) for s, p in zip(model.start_states, model.start_probs)]
    for i, (h, n) in enumerate(nodes):
        if h == H - 1:
            # `T_epi[nodes.index((h, n)), :, node_to_index[sn]] = p` (mdp_creation.py:168): the column is the ORIGINAL
            # index of the starting state, used as an index into the augmented space -- reproduced as written
            row = {sn: p for sn, p in start}
        else:
            row = None
        R_cf[i] = R[n]
        for a in range(A):
            if row is not None:
                ent = sorted(row.items())
            else:
                ent = sorted((idx[v], T[n, a, v[1]]) for v in adj.get((h, n), ()))
            ent = [(c, v) for c, v in ent if v != 0]
            cols.extend(c for c, _ in ent)
            vals.extend(v for _, v in ent)
            ptr[i * A + a + 1] = len(cols)
    return N, A, (ptr, np.array(cols, np.int32), np.array(vals, np.float32)), R_cf

"""Experiment: how much does the state numbering matter for K5S?  Relabels the C5 MDP (identity / random / RCM / BFS
clusters) and times the full diameter.  (Relabelling changes the float32 accumulation order inside rows, so this is a
timing experiment only.)"""
import sys, time
import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import reverse_cuthill_mckee, breadth_first_order
sys.path.insert(0, "/root/repo")
from colosseum_amd import _lib as L
from colosseum_amd.batched import BatchedMDP, tables_from_models
from colosseum_amd.mdp import make_model

m = make_model("MiniGridRoomsContinuous", seed=0, room_size=28, n_rooms=16, n_starting_states=2, p_lazy=0.1)
S, A = m.n_states, m.n_actions
t0 = tables_from_models([m], with_env=False)
ptr, col, val = t0["csr_ptr"], t0["csr_col"], t0["csr_val"]
rows = np.repeat(np.arange(S * A) // A, np.diff(ptr))
G = csr_matrix((np.ones(len(col)), (rows, col)), shape=(S, S))
G = ((G + G.T) > 0).astype(np.int8).tocsr()

def clusters(C):
    """greedy region growing: BFS from the first unassigned state until C states are collected"""
    order, seen = [], np.zeros(S, bool)
    indptr, indices = G.indptr, G.indices
    frontier_seed = 0
    from collections import deque
    pending = deque([0])
    while len(order) < S:
        while pending and seen[pending[0]]:
            pending.popleft()
        if not pending:
            pending.append(int(np.flatnonzero(~seen)[0]))
        q = deque([pending.popleft()])
        seen[q[0]] = True
        cnt = 0
        while q and cnt < C:
            v = q.popleft(); order.append(v); cnt += 1
            for w in indices[indptr[v]:indptr[v + 1]]:
                if not seen[w]:
                    seen[w] = True; q.append(w)
        for w in q:  # leftovers become seeds of the next clusters
            seen[w] = False; pending.append(w)
    return np.array(order)

def run(name, order):
    inv = np.empty(S, np.int64); inv[order] = np.arange(S)   # old -> new
    new_ptr, new_col, new_val = [0], [], []
    for s_new in range(S):
        s_old = order[s_new]
        for a in range(A):
            r = s_old * A + a
            c = inv[col[ptr[r]:ptr[r + 1]]]; v = val[ptr[r]:ptr[r + 1]]
            o = np.argsort(c, kind="stable")
            new_col.append(c[o]); new_val.append(v[o]); new_ptr.append(new_ptr[-1] + len(c))
    t = dict(t0)
    t["csr_ptr"] = np.array(new_ptr, np.int64); t["csr_col"] = np.concatenate(new_col).astype(np.int32)
    t["csr_val"] = np.concatenate(new_val).astype(np.float32)
    dp = BatchedMDP(tables=t, with_env=False)
    dp.diameter_range(0, 64)  # allocates the workspace, builds the fixed-width rows
    t1 = time.time(); per = dp.diameter_range(0, S); dt = time.time() - t1
    print(f"{name:12s} solve {dt:.3f} s  diameter {per.max():.4f}", flush=True)
    dp.close()

run("identity", np.arange(S))
run("rcm", np.asarray(reverse_cuthill_mckee(G, symmetric_mode=True)))
run("bfs", np.asarray(breadth_first_order(G, 0, directed=False, return_predecessors=False)))
for C in (32, 80, 256):
    run(f"clusters{C}", clusters(C))
run("random", np.random.default_rng(0).permutation(S))

"""Synthetic problem generators for the BASELINE.json configs.

Every instance is derived from ONE documented splitmix64 counter stream
(`stream(seed, n)[i] = splitmix64(seed + i * 0x9E3779B97F4A7C15)`) so the C++
oracle, the HIP path and Python agree bit for bit on the inputs.  Data only:
no solver logic lives here.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
MASK = (1 << 64) - 1


def splitmix64_np(v):
    """Vectorised splitmix64 finaliser (heuristic/selector/move_selector/iter.rs:193-198)."""
    with np.errstate(over="ignore"):
        v = (v + GOLDEN).astype(np.uint64)
        v = ((v ^ (v >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
        v = ((v ^ (v >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
        return v ^ (v >> np.uint64(31))


def stream(seed, n, offset=0):
    with np.errstate(over="ignore"):
        idx = (np.arange(offset, offset + n, dtype=np.uint64) * GOLDEN).astype(np.uint64)
        return splitmix64_np((np.uint64(seed & MASK) + idx).astype(np.uint64))


def step_seed(random_seed, draw_index):
    """Build-defined step-seed stream (the reference's StdRng is unpinned)."""
    return int(stream(random_seed, 1, draw_index)[0])


def make_cvrp(n_customers=1000, n_vehicles=100, capacity=55, seed=0, coord_range=1000):
    """C3/C5: depot + customers uniform in [0,coord_range)^2, dist = llround(euclid) i64,
    demand uniform 1..9, round-robin start routes in customer order."""
    dim = n_customers + 1
    r = stream(seed, 3 * dim)
    xs = (r[0:dim] % np.uint64(coord_range)).astype(np.int64)
    ys = (r[dim:2 * dim] % np.uint64(coord_range)).astype(np.int64)
    demands = (r[2 * dim:3 * dim] % np.uint64(9)).astype(np.int32) + 1
    demands[0] = 0
    dx = (xs[:, None] - xs[None, :]).astype(np.float64)
    dy = (ys[:, None] - ys[None, :]).astype(np.float64)
    matrix = np.floor(np.sqrt(dx * dx + dy * dy) + 0.5).astype(np.int64)  # llround for non-negative values
    routes = [[] for _ in range(n_vehicles)]
    for c in range(1, dim):
        routes[(c - 1) % n_vehicles].append(c)
    return {
        "capacity": capacity,
        "depot": 0,
        "demands": demands,
        "matrix": np.ascontiguousarray(matrix),
        "customers": np.arange(1, dim, dtype=np.uint32),
        "routes": routes,
        "n_customers": n_customers,
        "n_vehicles": n_vehicles,
    }


def make_graph(n=10000, n_edges=100000, n_colors=16, seed=0):
    """C2: n nodes, n_edges distinct undirected edges (i<j uniform, reject dup/self),
    neighbours symmetric + sorted; colours start unassigned."""
    edges = set()
    offset = 0
    while len(edges) < n_edges:
        need = n_edges - len(edges)
        r = stream(seed, 2 * (need + need // 8 + 16), offset)
        offset += len(r)
        a = (r[0::2] % np.uint64(n)).astype(np.int64)
        b = (r[1::2] % np.uint64(n)).astype(np.int64)
        for i, j in zip(a.tolist(), b.tolist()):
            if i == j:
                continue
            e = (i, j) if i < j else (j, i)
            if e not in edges:
                edges.add(e)
                if len(edges) == n_edges:
                    break
    e = np.array(sorted(edges), dtype=np.int64)
    src = np.concatenate([e[:, 0], e[:, 1]])
    dst = np.concatenate([e[:, 1], e[:, 0]])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    adj_off = np.zeros(n + 1, dtype=np.uint32)
    np.add.at(adj_off, src + 1, 1)
    adj_off = np.cumsum(adj_off).astype(np.uint32)
    return {
        "n": n,
        "n_colors": n_colors,
        "adj_off": adj_off,
        "adj": dst.astype(np.uint32),
        "colors": np.full(n, -1, dtype=np.int64),
    }


def construct_graph(g):
    """The post-construction start state of C2 ("start = first-fit construction"): the reference's
    first-fit construction (phase/construction/forager_step.rs:149-226) visits the vertices in index order and takes
    the first colour whose trial score is strictly better than keeping the vertex unassigned, i.e. the first colour no
    already-coloured neighbour holds; a vertex with no such colour stays unassigned.  Checked against the oracle's
    construct_first_fit in tests/test_oracle_golden.py."""
    n, k = g["n"], g["n_colors"]
    off, adj = g["adj_off"], g["adj"]
    colors = np.full(n, -1, dtype=np.int64)
    for v in range(n):
        used = set(colors[adj[off[v]:off[v + 1]]].tolist())
        for c in range(k):
            if c not in used:
                colors[v] = c
                break
    out = dict(g)
    out["colors"] = colors
    return out


def make_jobshop(n_jobs=500, n_machines=20):
    """C4: operations id -> (job=id//n_machines, step=id%n_machines); all unassigned / unscheduled."""
    n_ops = n_jobs * n_machines
    ids = np.arange(n_ops, dtype=np.int64)
    return {
        "n_ops": n_ops,
        "n_machines": n_machines,
        "job": ids // n_machines,
        "step": ids % n_machines,
        "machine_idx": np.full(n_ops, -1, dtype=np.int64),
        "sequences": [[] for _ in range(n_machines)],
    }


def construct_jobshop(p, seed=0):
    """A post-construction start state for C4 (the reference runs a construction phase before local search;
    construction itself is out of scope): every operation gets a machine drawn from the
    documented splitmix64 stream and is appended to a machine sequence drawn from the same stream."""
    n, m = p["n_ops"], p["n_machines"]
    r = stream(seed + 4242, 2 * n)
    q = dict(p)
    q["machine_idx"] = (r[:n] % np.uint64(m)).astype(np.int64)
    seqs = [[] for _ in range(m)]
    for op in range(n):
        seqs[int(r[n + op] % np.uint64(m))].append(op)
    q["sequences"] = seqs
    return q


def make_precedence_shop(n_jobs=10, n_machines=5, seed=0, scheduled=True, max_duration=9):
    """Classic job shop for the ListPrecedenceMakespanConstraint: operation id = job * n_machines + step; the job order is the
    fixed successor relation, every operation has a duration in 1..max_duration and an expected machine (a seeded permutation of
    the machines per job).  scheduled: every machine's sequence holds its operations in job order (acyclic: an operation's
    sequence position follows the job index, its job predecessors sit on other machines ... not necessarily acyclic -- the
    constraint scores cycles too); otherwise the sequences are empty."""
    n = n_jobs * n_machines
    r = stream(seed + 777, 3 * n + 1)
    dur = (r[:n] % np.uint64(max_duration)).astype(np.int64) + 1
    owner = np.zeros(n, dtype=np.int64)
    for j in range(n_jobs):  # Fisher-Yates with the documented stream
        perm = list(range(n_machines))
        for k in range(n_machines - 1, 0, -1):
            q = int(r[n + j * n_machines + k] % np.uint64(k + 1))
            perm[k], perm[q] = perm[q], perm[k]
        owner[j * n_machines:(j + 1) * n_machines] = perm
    succ = [[op + 1] if (op % n_machines) + 1 < n_machines else [] for op in range(n)]
    seqs = [[] for _ in range(n_machines)]
    if scheduled:  # step-major order: every machine sees its operations by (step, job) -- a feasible (acyclic) schedule
        for step in range(n_machines):
            for j in range(n_jobs):
                op = j * n_machines + step
                seqs[int(owner[op])].append(op)
    return {"durations": dur, "successors": succ, "expected_owner": owner, "sequences": seqs, "n_jobs": n_jobs, "n_machines": n_machines}

"""Synthetic graph structures for the benchmark and the smoke run (data generation only, host side, numpy).

The reference's timing harness draws Barabasi-Albert graphs through networkx (run_perf_tests.py:11-51,
`nx.barabasi_albert_graph(n, 5, seed)`); networkx is not guaranteed on the GPU box, so this is an own seeded
generator of the same attachment process (not bit-identical with networkx)."""
from typing import List

import numpy as np


def ba_edges(n: int, m: int, rng: np.random.Generator) -> np.ndarray:
    """Symmetric directed edge list [2, 2*(n-m)*m] (local node ids, row-major COO order, no self loops, unique)
    of one preferential-attachment graph: node `src` = m..n-1 attaches to m distinct earlier nodes drawn from the
    degree-weighted pool."""
    targets = list(range(m))
    pool: List[int] = []
    pairs = []
    for src in range(m, n):
        for t in targets:
            pairs.append((src, t))
        pool.extend(targets)
        pool.extend([src] * m)
        picked = set()
        while len(picked) < m:
            picked.add(pool[int(rng.integers(len(pool)))])
        targets = list(picked)
    und = np.asarray(pairs, dtype=np.int64)
    both = np.concatenate([und, und[:, ::-1]], 0)
    both = both[np.lexsort((both[:, 1], both[:, 0]))]
    return np.ascontiguousarray(both.T)

"""Generate tests/golden/merwgen_*.npz by calling the reference's own compute_merw (imported from
/root/reference/preprocess/compute_merw.py) on synthetic symmetric graphs, the way init_rw.py:76 calls it.
Run in the build container only:  python tests/golden/make_golden_merwgen.py
"""
import importlib.util
import os

import numpy as np
import scipy.sparse as sp

OUT = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("compute_merw", "/root/reference/preprocess/compute_merw.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def graph(n, m_und, seed, repeats=0):
    """connected, non-bipartite, symmetric; edge_index holds both directions (like from_scipy_sparse_matrix(adj))"""
    rng = np.random.default_rng(seed)
    und = set((a, (a + 1) % n) if a < (a + 1) % n else ((a + 1) % n, a) for a in range(n))
    und.add((0, 2))
    while len(und) < m_und:
        a, b = rng.integers(0, n, 2)
        if a != b:
            und.add((min(a, b), max(a, b)))
    und = sorted(und)
    und += und[:repeats]                                      # repeated edges add up in the adjacency matrix
    row = np.array([a for a, b in und] + [b for a, b in und], np.int64)
    col = np.array([b for a, b in und] + [a for a, b in und], np.int64)
    return np.stack([row, col])


def one(tag, n, m_und, seed, repeats=0):
    ei = graph(n, m_und, seed, repeats)
    A = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(n, n))       # init_rw.py:66-68
    P, psi, lam, _ = ref.compute_merw(A)                                           # init_rw.py:76
    P = P.tocsr()
    p_edge = np.asarray(P[ei[0], ei[1]]).reshape(-1)
    np.savez_compressed(os.path.join(OUT, "merwgen_%s.npz" % tag), n=n, edge_index=ei, p_edge=p_edge,
                        psi=np.abs(psi), lam=lam)
    print(tag, "n", n, "columns", ei.shape[1], "lambda", lam, "row sums", P.sum(axis=1).min(), P.sum(axis=1).max())


if __name__ == "__main__":
    one("g60", 60, 150, 1)
    one("g400", 400, 1400, 2, repeats=25)
    one("ring41", 41, 41, 3)                    # odd ring + one chord: small spectral gap, wide range of psi

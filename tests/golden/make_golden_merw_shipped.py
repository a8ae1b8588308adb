"""Generate tests/golden/merwfile_*.npz from the edge files the reference SHIPS (edge_input/cornell.in, Nba.in, cora.in):
the only outputs of preprocess/init_rw.py + compute_merw.py that the reference itself holds.
The weighted adjacency matrix the generator was run on is not shipped; it is recovered from the file:
    P[u,v] P[v,u] = A[u,v]^2 / lambda^2   (compute_merw.py:116-120: P = A psi_v / (lambda psi_u), A symmetric)
so  A[u,v] = lambda sqrt(P[u,v] P[v,u]),  with lambda fixed by requiring integer weights (self loops weigh 1 or 2: the
old_datasets branch of init_rw.py adds the identity to an adjacency that may already hold self loops).
Run in the build container only:  python tests/golden/make_golden_merw_shipped.py"""
import os

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/edge_input"


def components(n, u, v):
    lab = np.arange(n)
    while True:
        new = lab.copy()
        np.minimum.at(new, u, lab[v])
        np.minimum.at(new, v, lab[u])
        new = new[new]
        if (new == lab).all():
            return lab
        lab = new


def one(name, tag):
    rows = [ln.split() for ln in open(os.path.join(REF, name + ".in")).read().strip().split("\n")]
    n, m = int(rows[0][0]), int(rows[0][1])
    u = np.array([int(r[0]) for r in rows[1:]])
    v = np.array([int(r[1]) for r in rows[1:]])
    p = np.array([float(r[2]) for r in rows[1:]])
    assert len(u) == m and (u[0::2] == v[1::2]).all() and (v[0::2] == u[1::2]).all()     # init_rw.py:80-86: rows in pairs
    ei = np.stack([u[0::2], v[0::2]])
    p_uv, p_vu = p[0::2], p[1::2]
    lab = components(n, ei[0], ei[1])
    sizes = np.bincount(lab, minlength=n)
    giant = lab == np.argmax(sizes)
    in_giant = giant[ei[0]]
    # lambda: the smallest multiple of 1 / P[u,u] (u a self loop of the giant component) that makes every weight an integer
    k = np.flatnonzero((ei[0] == ei[1]) & in_giant)
    base = (1.0 / p_uv[k]).max()
    for mult in (1, 2, 3, 4):
        lam = base * mult
        w = lam * np.sqrt(np.maximum(p_uv * p_vu, 0.0))
        if np.abs(w[in_giant] - np.round(w[in_giant])).max() < 1e-6 and np.round(w[in_giant]).min() >= 1:
            break
    else:
        raise SystemExit("no integer weights for " + name)
    weights = np.where(in_giant, np.round(w), 0.0)
    # columns outside the giant component: self loops of single nodes keep the value the file implies (P[u,u] = A[u,u] /
    # lambda whatever psi is); larger minor components carry eigensolver noise in the file -- unit weights, not compared
    single = (~in_giant) & (ei[0] == ei[1]) & (sizes[lab[ei[0]]] == 1)
    weights = np.where(single, np.round(p_uv * lam), weights)
    weights = np.where((~in_giant) & ~single, 1.0, weights)
    assert weights.min() >= 1
    np.savez_compressed(os.path.join(OUT, "merwfile_%s.npz" % tag), n=n, edge_index=ei.astype(np.int32), weights=weights,
                        p_uv=p_uv, p_vu=p_vu, lam=lam, in_giant=in_giant, single=single)
    print(name, "n", n, "columns", ei.shape[1], "lambda", lam, "weights", np.unique(weights), "giant columns", int(in_giant.sum()),
          "single-node self loops outside", int(single.sum()))


if __name__ == "__main__":
    one("cornell", "cornell")
    one("Nba", "nba")
    one("cora", "cora")

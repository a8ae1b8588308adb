"""MERW transition-probability generator: oracle vs the goldens the reference's own compute_merw
produced (CPU), and the HIP power iteration vs both (GPU)."""
import os

import numpy as np
import pytest

from conftest import golden, golden_files
from oracle import merw_gen as mg

GOLDENS = golden_files("merwgen_*.npz")
TOL = 1e-9          # two different eigensolvers (ARPACK in the reference); probabilities are O(0.01 .. 1)


@pytest.mark.parametrize("name", GOLDENS)
def test_oracle_reproduces_reference_compute_merw(name):
    g = golden(name)
    n, ei = int(g["n"]), g["edge_index"]
    P, psi, lam = mg.merw_matrix(mg.adjacency_dense(n, ei))
    assert abs(lam - float(g["lam"])) < 1e-11
    assert np.abs(psi - g["psi"]).max() < 1e-11
    assert np.abs(P[ei[0], ei[1]] - g["p_edge"]).max() < TOL
    assert np.abs(P.sum(1) - 1.0).max() < 1e-9                       # a stochastic matrix


def test_host_adjacency_matches_scipy_semantics_of_the_reference():
    from pathnet_amd import merw_init as mi
    g = golden("merwgen_g400.npz")                                   # holds repeated edges: they add up (init_rw.py:66-68)
    n, ei = int(g["n"]), g["edge_index"]
    ro, col, val, k_uv, k_vu = mi.adjacency_csr(n, ei)
    A = mg.adjacency_dense(n, ei)
    B = np.zeros_like(A)
    for i in range(n):
        B[i, col[ro[i]:ro[i + 1]]] = val[ro[i]:ro[i + 1]]
    assert (A == B).all() and val.max() == 2.0
    assert (col[k_uv] == ei[1]).all() and (col[k_vu] == ei[0]).all()
    with pytest.raises(ValueError):
        mi.merw_probabilities(3, np.array([[0, 1], [1, 2]]))          # (0, 1) without (1, 0): refused on the host


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDENS)
def test_hip_power_iteration_matches_reference_golden(name):
    from pathnet_amd import merw_init as mi
    g = golden(name)
    n, ei = int(g["n"]), g["edge_index"]
    r = mi.merw_probabilities(n, ei)
    assert abs(r["lam"] - float(g["lam"])) < 1e-10
    assert np.abs(r["psi"] - g["psi"]).max() < 1e-9
    assert np.abs(r["p_uv"] - g["p_edge"]).max() < TOL
    P, _, _ = mg.merw_matrix(mg.adjacency_dense(n, ei))
    assert np.abs(r["p_vu"] - P[ei[1], ei[0]]).max() < TOL


@pytest.mark.gpu
def test_generated_edge_file_feeds_the_sampler(tmp_path):
    """init_rw.py's file format end to end: generator -> edge_input/<name>.in -> MerwSampler.  (Every row of such a file
    appears twice, so a node's listed probabilities sum to 2 and the alias tables built from it -- the reference's and
    ours, bit for bit -- do not sample exactly P; that is the reference's behaviour, not checked against P here.)"""
    import pathnet_amd
    from pathnet_amd import merw_init as mi
    g = golden("merwgen_g60.npz")
    n, ei = int(g["n"]), g["edge_index"]
    r = mi.merw_probabilities(n, ei)
    f = os.path.join(tmp_path, "g60.in")
    mi.write_edge_input(f, n, ei, r["p_uv"], r["p_vu"])
    head = open(f).readline().split()
    assert head == [str(n), str(2 * ei.shape[1])]
    want = mg.format_edge_file(n, *mg.edge_rows(n, ei, mg.merw_matrix(mg.adjacency_dense(n, ei))[0]))
    got_rows = [ln.split() for ln in open(f).read().strip().split("\n")[1:]]
    want_rows = [ln.split() for ln in want.strip().split("\n")[1:]]
    assert [r_[:2] for r_ in got_rows] == [r_[:2] for r_ in want_rows]
    assert max(abs(float(a[2]) - float(b[2])) for a, b in zip(got_rows, want_rows)) < TOL
    # the file is what the sampler (and the reference's gen_merw) reads: same walks as the pinned oracle on it
    from oracle import merw
    smp = pathnet_amd.MerwSampler.from_edge_file(f, 4)
    ids, codes = smp.sample(40, 5, epoch_count=2)
    n2, u2, v2, p2 = merw.read_edge_file(f)
    oi, oc = merw.sample_full(n2, u2, v2, p2, 40, 4, merw.DRAW_PHILOX, 5, epoch_count=2)
    assert n2 == n and (ids.cpu().numpy() == oi).all() and (codes.cpu().numpy() == oc).all()


def test_edge_file_writer_prints_what_the_reference_prints(tmp_path):
    """init_rw.py:78-86: header "n 2M", per edge_index column the rows "u v P[u,v]" and "v u P[v,u]", floats as Python
    prints numpy float64 (the writer is host code: no GPU needed)."""
    from pathnet_amd import merw_init as mi
    g = golden("merwgen_g60.npz")
    n, ei = int(g["n"]), g["edge_index"]
    P, _, _ = mg.merw_matrix(mg.adjacency_dense(n, ei))
    f = os.path.join(tmp_path, "g60.in")
    mi.write_edge_input(f, n, ei, P[ei[0], ei[1]], P[ei[1], ei[0]])
    assert open(f).read() == mg.format_edge_file(n, *mg.edge_rows(n, ei, P))
    # and it parses back through the sampler's own reader
    from pathnet_amd import sampler
    n2, u2, v2, p2 = sampler.read_edge_file(f)
    ru, rv, rp = mg.edge_rows(n, ei, P)
    assert n2 == n and (u2 == ru).all() and (v2 == rv).all() and (p2 == rp).all()


# ---- the edge files the reference ships: f-2 pinned on reference-held artefacts --------------------------
SHIPPED = golden_files("merwfile_*.npz")


def _giant_oracle(g):
    """P of the dominant component from the oracle's dense eigensolver, per edge column (NaN elsewhere)"""
    n, ei, w = int(g["n"]), g["edge_index"], g["weights"]
    big = g["in_giant"]
    nodes = np.unique(ei[:, big])
    pos = np.full(n, -1)
    pos[nodes] = np.arange(len(nodes))
    A = mg.adjacency_dense(len(nodes), pos[ei[:, big]], w[big])
    P, psi, lam = mg.merw_matrix(A)
    a, b = pos[ei[0, big]], pos[ei[1, big]]
    return P[a, b], P[b, a], lam


@pytest.mark.parametrize("name", [f for f in SHIPPED if "cora" not in f])
def test_oracle_reproduces_the_shipped_edge_files(name):
    """edge_input/cornell.in and Nba.in, every row: the oracle on the adjacency recovered from the file itself"""
    g = golden(name)
    p_uv, p_vu, lam = _giant_oracle(g)
    big = g["in_giant"]
    assert abs(lam - float(g["lam"])) < 1e-9
    assert np.abs(p_uv - g["p_uv"][big]).max() < TOL and np.abs(p_vu - g["p_vu"][big]).max() < TOL
    # single nodes outside the dominant component: A[u,u] / lambda (psi cancels in compute_merw.py:118)
    s_ = g["single"]
    assert np.abs(g["weights"][s_] / lam - g["p_uv"][s_]).max(initial=0.0) < TOL
    assert (big | s_).all()                                      # i.e. every row of these two files is reproduced
    if "cornell" in name:                                        # connected: the file itself, where repr(float) agrees
        n, ei = int(g["n"]), g["edge_index"]
        P = mg.merw_matrix(mg.adjacency_dense(n, ei, g["weights"]))[0]
        got = mg.format_edge_file(n, *mg.edge_rows(n, ei, P)).split("\n")
        want = mg.format_edge_file(n, *mg.edge_rows_from(ei, g["p_uv"], g["p_vu"])).split("\n")
        assert got[0] == want[0] and len(got) == len(want)
        # same rows in the same order; the printed probabilities agree to 12 significant digits (two eigensolvers: the
        # 17-digit repr itself coincides on 28 of the 1474 rows)
        assert all(a.split()[:2] == b.split()[:2] for a, b in zip(got[1:-1], want[1:-1]))
        assert all(abs(float(a.split()[2]) - float(b.split()[2])) <= 1e-12 * abs(float(b.split()[2]))
                   for a, b in zip(got[1:-1], want[1:-1]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", SHIPPED)
def test_hip_generator_reproduces_the_shipped_edge_files(name):
    from pathnet_amd import merw_init as mi
    g = golden(name)
    n, ei = int(g["n"]), g["edge_index"].astype(np.int64)
    r = mi.merw_probabilities(n, ei, weights=g["weights"])
    big, s_ = g["in_giant"], g["single"]
    # cora.in: the reference's own eigensolver (ARPACK, default tolerance) is 1.6e-5 away from the exact dominant pair
    tol = 5e-5 if "cora" in name else TOL
    assert abs(r["lam"] - float(g["lam"])) < (1e-4 if "cora" in name else 1e-9)
    assert np.abs(r["p_uv"][big] - g["p_uv"][big]).max() < tol and np.abs(r["p_vu"][big] - g["p_vu"][big]).max() < tol
    assert np.abs(r["p_uv"][s_] - g["p_uv"][s_]).max(initial=0.0) < tol
    assert (r["reference_defined"] == (big | s_)).all()
    assert r["components"] == len(np.unique(mi.component_labels(n, ei[0], ei[1])))
    # whatever the component, a node's probabilities sum to one -- except single nodes outside the dominant component,
    # where the reference's A[u,u] / lambda is kept
    ro, col, val, k_uv, _ = mi.adjacency_csr(n, ei, g["weights"])
    tot = np.zeros(n)
    first = np.unique(k_uv, return_index=True)[1]                # one column per stored entry
    np.add.at(tot, ei[0][first], r["p_uv"][first])
    lone = np.zeros(n, bool)
    lone[ei[0][s_]] = True
    assert np.abs(tot[~lone] - 1.0).max() < 1e-9


@pytest.mark.gpu
def test_disconnected_graph_policy(tmp_path):
    """cora.in's graph has 78 components: the default computes every component (the dominant one as the reference does),
    disconnected='error' and the CLI's --strict refuse it with a clear message"""
    from pathnet_amd import merw_init as mi
    g = golden("merwfile_cora.npz")
    n, ei = int(g["n"]), g["edge_index"].astype(np.int64)
    with pytest.raises(ValueError, match="connected components"):
        mi.merw_probabilities(n, ei, disconnected="error")
    np.save(os.path.join(tmp_path, "ei.npy"), ei)
    assert mi.main([os.path.join(tmp_path, "ei.npy"), str(n), "-o", os.path.join(tmp_path, "cora.in"), "--strict"]) == 1
    assert mi.main([os.path.join(tmp_path, "ei.npy"), str(n), "-o", os.path.join(tmp_path, "cora.in")]) == 0
    head = open(os.path.join(tmp_path, "cora.in")).readline().split()
    assert head == [str(n), str(2 * ei.shape[1])]


@pytest.mark.gpu
def test_many_isolated_nodes_cost_their_own_size_not_the_graph():
    """200 000 single-node components (half of them with a self loop) beside a ring and a path: the components are grouped
    once and only those two run a power iteration (a flatnonzero over all nodes per component made this quadratic)."""
    import time
    from pathnet_amd import merw_init as mi
    n_iso, ring_a, ring_b = 200_000, 41, 9            # odd rings: not bipartite
    n = n_iso + ring_a + ring_b
    loops = np.arange(0, n_iso, 2)                                         # self loops on the even isolated nodes
    a0, b0 = n_iso, n_iso + ring_a
    ra = np.arange(ring_a)
    rb = np.arange(ring_b)
    pb = np.arange(ring_b - 1)                                             # the second component: a path (lambda = 2 cos(pi / 10) < 2)
    u = np.concatenate([loops, a0 + ra, a0 + (ra + 1) % ring_a, b0 + pb, b0 + pb + 1])
    v = np.concatenate([loops, a0 + (ra + 1) % ring_a, a0 + ra, b0 + pb + 1, b0 + pb])
    t0 = time.time()
    r = mi.merw_probabilities(n, np.stack([u, v]))
    dt = time.time() - t0
    assert dt < 20.0, dt
    assert r["components"] == n_iso + 2
    # a ring's adjacency has lambda = 2 and a uniform eigenvector: P = 1/2 on every ring edge; it dominates the path
    assert abs(r["lam"] - 2.0) < 1e-9
    ring_cols = np.flatnonzero(u >= n_iso)
    on_ring = ring_cols[:2 * ring_a]
    assert np.abs(r["p_uv"][on_ring] - 0.5).max() < 1e-9 and np.abs(r["p_vu"][on_ring] - 0.5).max() < 1e-9
    tot = np.zeros(n)                                                      # the path: its own maximal-entropy walk, rows sum to one
    np.add.at(tot, u[ring_cols[2 * ring_a:]], r["p_uv"][ring_cols[2 * ring_a:]])
    assert np.abs(tot[b0:b0 + ring_b] - 1.0).max() < 1e-9
    psi_ring = r["psi"][a0:a0 + ring_a]                                    # the dominant eigenvector: uniform on the ring, zero elsewhere
    assert psi_ring.min() > 0 and np.ptp(psi_ring) < 1e-9 * psi_ring.max() and (r["psi"][:n_iso] == 0).all()
    # a single node with a self loop (weight 1) outside the dominant component: the reference's A[u,u] / lambda = 1/2
    assert np.abs(r["p_uv"][:len(loops)] - 0.5).max() < 1e-12
    assert r["reference_defined"][:len(loops)].all() and r["reference_defined"][ring_cols[:2 * ring_a]].all()
    assert not r["reference_defined"][ring_cols[2 * ring_a:]].any()        # the path: its own walk, not the reference's noise

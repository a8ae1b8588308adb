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

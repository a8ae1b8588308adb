"""CPU restatement of the MERW transition-probability generator.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Reference behaviour restated:
  /root/reference/preprocess/compute_merw.py:107-121  compute_merw(A): dominant eigenpair (lambda, psi) of the symmetric
      adjacency matrix (scipy eigsh, k=1), P[i, j] = A[i, j] * psi[j] / (lambda * psi[i]) on the non-zeros of A;
  /root/reference/preprocess/init_rw.py:63-86  adjacency = csr_matrix((ones, (row, col))) of the edge_index columns
      (repeated columns ADD UP), then the edge file: header "n 2M", and for every column i of edge_index the two rows
      "u v P[u,v]" and "v u P[v,u]" (edge_index of a symmetric graph already holds both directions, so every row of
      the file appears twice -- the shipped .in files do).
Pinned against the reference function itself, imported from /root/reference on synthetic graphs
(tests/golden/make_golden_merwgen.py -> tests/golden/merwgen_*.npz).  Also pinned on what the reference SHIPS: edge_input/cornell.in and Nba.in are
reproduced row for row (2e-13) from the adjacency recovered from them (tests/golden/merwfile_*.npz,
tests/test_merw_gen.py).  Parity is to 1e-9 (two different eigensolvers),
on connected non-bipartite graphs -- elsewhere the reference's own output is not well defined (eigsh's "largest
magnitude" may return -lambda on a bipartite graph, and the eigenvector is noise on the smaller components, which is
where the negative and > 1 "probabilities" of the shipped cora/citeseer files come from).
"""
import numpy as np


def adjacency_dense(n, edge_index, weights=None):
    """init_rw.py:63-68: csr_matrix((ones, (row, col)), shape=(n, n)) -- duplicates accumulate.  weights: per column
    instead of ones (the shipped cornell.in / Nba.in come from matrices whose self loops weigh 2,
    tests/golden/make_golden_merw_shipped.py)."""
    A = np.zeros((n, n), np.float64)
    np.add.at(A, (np.asarray(edge_index[0]), np.asarray(edge_index[1])), 1.0 if weights is None else np.asarray(weights, np.float64))
    return A


def merw_matrix(A):
    """compute_merw.py:107-121 with a dense symmetric eigensolver: -> P (dense), psi (unit norm, psi[0] > 0), lambda."""
    w, v = np.linalg.eigh(A)
    lam, psi = w[-1], v[:, -1]
    if psi[np.argmax(np.abs(psi))] < 0:
        psi = -psi
    with np.errstate(divide="ignore", invalid="ignore"):
        P = np.where(A != 0, A * psi[None, :] / (lam * psi[:, None]), 0.0)
    return P, psi, lam


def edge_rows(n, edge_index, P):
    """init_rw.py:80-86: the (u, v, p) rows of the edge file, in file order."""
    u, v = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    ru = np.stack([u, v], 1).reshape(-1)
    rv = np.stack([v, u], 1).reshape(-1)
    return ru.astype(np.int32), rv.astype(np.int32), P[ru, rv]


def edge_rows_from(edge_index, p_uv, p_vu):
    """the same rows from per-column probabilities"""
    u, v = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    ru = np.stack([u, v], 1).reshape(-1)
    rv = np.stack([v, u], 1).reshape(-1)
    return ru.astype(np.int32), rv.astype(np.int32), np.stack([p_uv, p_vu], 1).reshape(-1)


def format_edge_file(n, ru, rv, rp):
    """header + rows the way init_rw.py prints them (Python repr of numpy float64)."""
    lines = ["%d %d" % (n, len(ru))]
    lines += ["%d %d %s" % (a, b, repr(float(p))) for a, b, p in zip(ru, rv, rp)]
    return "\n".join(lines) + "\n"

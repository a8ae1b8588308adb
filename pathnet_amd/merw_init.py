"""MERW transition probabilities on the GPU: the generator of the sampler's ``edge_input/<name>.in``.

Replaces /root/reference/preprocess/init_rw.py:63-86 + compute_merw.py:107-121:

    adjacency  A = csr_matrix((ones, (row, col)))            of the edge_index columns (repeats add up)     init_rw.py:63-68
    (lambda, psi) = dominant eigenpair of A                                                                compute_merw.py:109-112
    P[u, v] = A[u, v] * psi[v] / (lambda * psi[u])                                                           compute_merw.py:116-120
    file: "n 2M", then per edge_index column  "u v P[u,v]"  and  "v u P[v,u]"                                init_rw.py:80-86

``python -m pathnet_amd.merw_init <edge_index.npy | pairs.txt> <n> -o edge_input/<name>.in [--strict]``
"""
import ctypes
import sys

import numpy as np
import torch

from . import _lib


def adjacency_csr(n, edge_index, weights=None):
    """-> row_off int64 [n+1], col int32 [nnz], val float64 [nnz] (sorted; repeated columns of edge_index add up, each
    counting 1 or its entry of `weights`), and for every edge_index column i the positions k_uv[i], k_vu[i] of entries
    (u, v) / (v, u) (-1 if absent)."""
    u = np.asarray(edge_index[0], np.int64)
    v = np.asarray(edge_index[1], np.int64)
    if u.size and (min(u.min(), v.min()) < 0 or max(u.max(), v.max()) >= n):
        raise ValueError("edge_index holds node ids outside [0, n)")
    key = u * n + v
    uniq, inverse, counts = np.unique(key, return_inverse=True, return_counts=True)
    if weights is None:
        val = counts.astype(np.float64)
    else:
        val = np.zeros(len(uniq), np.float64)
        np.add.at(val, inverse, np.asarray(weights, np.float64))
    rows, cols = uniq // n, uniq % n
    row_off = np.zeros(n + 1, np.int64)
    np.add.at(row_off, rows + 1, 1)
    row_off = np.cumsum(row_off)
    k_uv = np.searchsorted(uniq, key)
    rkey = v * n + u
    k_vu = np.searchsorted(uniq, rkey)
    k_vu = np.where((k_vu < len(uniq)) & (uniq[np.minimum(k_vu, len(uniq) - 1)] == rkey), k_vu, -1)
    return row_off, cols.astype(np.int32), val, k_uv, k_vu


def component_labels(n, u, v):
    """label of every node = the smallest node id of its connected component (min-label propagation with pointer jumping)"""
    u, v = np.asarray(u, np.int64), np.asarray(v, np.int64)
    lab = np.arange(n, dtype=np.int64)
    while True:
        new = lab.copy()
        np.minimum.at(new, u, lab[v])
        np.minimum.at(new, v, lab[u])
        new = new[new]
        if (new == lab).all():
            return lab
        lab = new


def _power_iteration(lib, dev, row_off, col, val, tol, max_iter):
    n = len(row_off) - 1
    d_off, d_col = torch.from_numpy(row_off).to(dev), torch.from_numpy(col).to(dev)
    d_val = torch.from_numpy(val).to(dev)
    nnz = len(col)
    p = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
    psi = torch.empty(n, dtype=torch.float64, device=dev)
    need = ctypes.c_int64(0)
    _lib.check(lib.pn_merw_workspace_bytes(n, ctypes.byref(need)))
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    lam, iters = ctypes.c_double(0.0), ctypes.c_int32(0)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.pn_merw_probabilities(n, nnz, _lib.ptr(d_off), _lib.ptr(d_col), _lib.ptr(d_val), _lib.ptr(p),
                                             _lib.ptr(psi), ctypes.byref(lam), max_iter, tol, ctypes.byref(iters),
                                             _lib.ptr(ws), need.value, stream))
    return p.cpu().numpy()[:nnz], psi.cpu().numpy(), lam.value, iters.value


def merw_probabilities(n, edge_index, device="cuda", tol=1e-13, max_iter=200000, weights=None, disconnected="components"):
    """-> dict(p_uv, p_vu: float64 per edge_index column; psi [n]; lam; iters; components; reference_defined).
    The adjacency must be symmetric (the reference feeds an undirected graph's edge_index, which lists both directions).
    weights: one entry per edge_index column instead of 1 (the shipped cornell.in / Nba.in were made from adjacency matrices
    with self loops of weight 2, tests/golden/make_golden_merw_shipped.py).

    Disconnected graphs.  compute_merw.py:109-120 takes ONE eigenpair of the whole matrix: psi is that of the component with
    the largest eigenvalue and numerically zero -- eigensolver noise -- elsewhere, so the reference's P is exact on that
    component, equals A[u,u] / lambda on single nodes with a self loop (psi cancels), and is noise on every other component
    (the negative and > 1 "probabilities" of the shipped cora.in / citeseer.in).  disconnected="components" (default):
    the dominant component and the single nodes get the reference's values; every other component gets ITS OWN eigenpair's
    maximal-entropy walk -- a stochastic matrix where the reference prints noise; `reference_defined` marks the columns that
    reproduce the reference.  disconnected="error": refuse such a graph."""
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("pathnet_amd.merw_init: GPU only (there is no CPU fallback)")
    if disconnected not in ("components", "error"):
        raise ValueError("disconnected: 'components' or 'error'")
    row_off, col, val, k_uv, k_vu = adjacency_csr(n, edge_index, weights)
    if (k_vu < 0).any():
        raise ValueError("the adjacency matrix is not symmetric: edge (v, u) is missing for some (u, v)")
    u = np.asarray(edge_index[0], np.int64)
    lab = component_labels(n, u, np.asarray(edge_index[1], np.int64))
    roots = np.unique(lab)
    if len(roots) == 1:
        ph, psi, lam, iters = _power_iteration(lib, dev, row_off, col, val, tol, max_iter)
        return dict(p_uv=ph[k_uv], p_vu=ph[k_vu], psi=psi, lam=lam, iters=iters, components=1,
                    reference_defined=np.ones(len(u), bool))
    if disconnected == "error":
        raise ValueError("the graph has %d connected components: the reference's output is defined on the dominant one only "
                         "(pass disconnected='components')" % len(roots))
    rows_of = np.repeat(np.arange(n), np.diff(row_off))             # row of every stored entry
    ph = np.zeros(len(col), np.float64)
    psi = np.zeros(n, np.float64)
    sizes = np.bincount(lab, minlength=n)
    # nodes and stored entries grouped by component ONCE (argsort + segment offsets): the work per component is then its own
    # size -- citeseer has ~440 components, a large graph with many isolated nodes has as many as it has such nodes, and a
    # flatnonzero over all nodes / entries per component would be quadratic
    node_order = np.argsort(lab, kind="stable")
    node_seg = np.concatenate([[0], np.cumsum(sizes[roots])])
    ent_lab = lab[rows_of]
    ent_order = np.argsort(ent_lab, kind="stable")
    ent_seg = np.concatenate([[0], np.cumsum(np.bincount(ent_lab, minlength=n)[roots])])
    # single nodes, all at once: lambda = A[u,u] (0 without a self loop), P[u,u] = 1 on its own
    single = sizes[roots] == 1
    diag = np.zeros(n, np.float64)
    np.add.at(diag, rows_of, np.where(sizes[ent_lab] == 1, val, 0.0))
    single_ent = np.flatnonzero(sizes[ent_lab] == 1)
    ph[single_ent] = 1.0
    best, total_iters, dom, psi_dom, nodes_dom = (-1.0, -1), 0, None, None, None
    if single.any():
        r_s = roots[single]
        k = int(np.argmax(diag[r_s]))       # (a component's label is its smallest node: the node itself here)
        best, dom, psi_dom, nodes_dom = (float(diag[r_s[k]]), 1), int(r_s[k]), np.ones(1), np.array([int(r_s[k])])
        # (ties between single nodes: the first one, as the sequential loop did -- np.argmax returns the first maximum)
    lam_of = {}
    pos = np.full(n, -1, np.int64)
    for i in np.flatnonzero(~single).tolist():                      # power iteration only where there is something to iterate
        r = int(roots[i])
        nodes = np.sort(node_order[node_seg[i]:node_seg[i + 1]])
        ent = np.sort(ent_order[ent_seg[i]:ent_seg[i + 1]])         # its stored entries (rows and columns stay inside it)
        pos[nodes] = np.arange(len(nodes))
        ro = np.concatenate([[0], np.cumsum(np.diff(row_off)[nodes])]).astype(np.int64)
        p_c, psi_c, lam_c, it = _power_iteration(lib, dev, ro, pos[col[ent]].astype(np.int32), val[ent], tol, max_iter)
        ph[ent] = p_c
        total_iters += it
        lam_of[r] = lam_c
        if (lam_c, len(nodes)) > best or ((lam_c, len(nodes)) == best and r < dom):
            best, dom, psi_dom, nodes_dom = (lam_c, len(nodes)), r, psi_c, nodes
    lam = best[0]
    psi[nodes_dom] = psi_dom
    # the reference's value on a single node outside the dominant component: A[u,u] psi_u / (lambda psi_u)
    off_dom = single_ent[ent_lab[single_ent] != dom]
    ph[off_dom] = val[off_dom] / lam
    defined = (lab[u] == dom) | (sizes[lab[u]] == 1)
    return dict(p_uv=ph[k_uv], p_vu=ph[k_vu], psi=psi, lam=lam, iters=total_iters, components=len(roots),
                reference_defined=defined)


def write_edge_input(path, n, edge_index, p_uv, p_vu):
    """The edge file init_rw.py:78-86 writes (rows in its order, floats printed like Python prints numpy float64)."""
    u, v = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    with open(path, "w") as f:
        f.write("%d %d\n" % (n, 2 * len(u)))
        for a, b, x, y in zip(u.tolist(), v.tolist(), p_uv.tolist(), p_vu.tolist()):
            f.write("%d %d %r\n%d %d %r\n" % (a, b, x, b, a, y))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    out = None
    strict = "--strict" in argv
    if strict:
        argv.remove("--strict")
    if "-o" in argv:
        i = argv.index("-o")
        out = argv[i + 1]
        del argv[i:i + 2]
    if len(argv) != 2 or out is None:
        print("usage: python -m pathnet_amd.merw_init <edge_index.npy | pairs.txt> <n> -o <edge_input/name.in> [--strict]",
              file=sys.stderr)
        return 2
    src, n = argv[0], int(argv[1])
    ei = np.load(src) if src.endswith(".npy") else np.loadtxt(src, dtype=np.int64).reshape(-1, 2).T
    try:
        r = merw_probabilities(n, ei, disconnected="error" if strict else "components")
    except ValueError as e:
        print("pathnet_amd.merw_init: %s" % e, file=sys.stderr)
        return 1
    write_edge_input(out, n, ei, r["p_uv"], r["p_vu"])
    print("lambda %.12g after %d iterations; %d rows -> %s" % (r["lam"], r["iters"], 2 * ei.shape[1], out))
    if r["components"] > 1:
        bad = int((~r["reference_defined"]).sum())
        print("pathnet_amd.merw_init: %d connected components; %d of %d edge columns lie in minor components of more than one "
              "node, where preprocess/compute_merw.py prints eigensolver noise and this file holds each component's own "
              "maximal-entropy walk (--strict refuses such graphs)" % (r["components"], bad, ei.shape[1]), file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())

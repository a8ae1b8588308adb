"""MERW transition probabilities on the GPU: the generator of the sampler's ``edge_input/<name>.in``.

Replaces /root/reference/preprocess/init_rw.py:63-86 + compute_merw.py:107-121:

    adjacency  A = csr_matrix((ones, (row, col)))            of the edge_index columns (repeats add up)     init_rw.py:63-68
    (lambda, psi) = dominant eigenpair of A                                                                compute_merw.py:109-112
    P[u, v] = A[u, v] * psi[v] / (lambda * psi[u])                                                           compute_merw.py:116-120
    file: "n 2M", then per edge_index column  "u v P[u,v]"  and  "v u P[v,u]"                                init_rw.py:80-86

``python -m pathnet_amd.merw_init <edge_index.npy | pairs.txt> <n> -o edge_input/<name>.in``
"""
import ctypes
import sys

import numpy as np
import torch

from . import _lib


def adjacency_csr(n, edge_index):
    """-> row_off int64 [n+1], col int32 [nnz], val float64 [nnz] (sorted, repeated columns of edge_index summed), and
    for every edge_index column i the positions k_uv[i], k_vu[i] of entries (u, v) / (v, u) (-1 if absent)."""
    u = np.asarray(edge_index[0], np.int64)
    v = np.asarray(edge_index[1], np.int64)
    if u.size and (min(u.min(), v.min()) < 0 or max(u.max(), v.max()) >= n):
        raise ValueError("edge_index holds node ids outside [0, n)")
    key = u * n + v
    uniq, counts = np.unique(key, return_counts=True)
    rows, cols = uniq // n, uniq % n
    row_off = np.zeros(n + 1, np.int64)
    np.add.at(row_off, rows + 1, 1)
    row_off = np.cumsum(row_off)
    k_uv = np.searchsorted(uniq, key)
    rkey = v * n + u
    k_vu = np.searchsorted(uniq, rkey)
    k_vu = np.where((k_vu < len(uniq)) & (uniq[np.minimum(k_vu, len(uniq) - 1)] == rkey), k_vu, -1)
    return row_off, cols.astype(np.int32), counts.astype(np.float64), k_uv, k_vu


def merw_probabilities(n, edge_index, device="cuda", tol=1e-13, max_iter=200000):
    """-> dict(p_uv, p_vu: float64 per edge_index column; psi [n]; lam; iters).  The adjacency must be symmetric (the
    reference feeds an undirected graph's edge_index, which lists both directions)."""
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("pathnet_amd.merw_init: GPU only (there is no CPU fallback)")
    row_off, col, val, k_uv, k_vu = adjacency_csr(n, edge_index)
    if (k_vu < 0).any():
        raise ValueError("the adjacency matrix is not symmetric: edge (v, u) is missing for some (u, v)")
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
    ph = p.cpu().numpy()
    return dict(p_uv=ph[k_uv], p_vu=ph[k_vu], psi=psi.cpu().numpy(), lam=lam.value, iters=iters.value)


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
    if "-o" in argv:
        i = argv.index("-o")
        out = argv[i + 1]
        del argv[i:i + 2]
    if len(argv) != 2 or out is None:
        print("usage: python -m pathnet_amd.merw_init <edge_index.npy | pairs.txt> <n> -o <edge_input/name.in>",
              file=sys.stderr)
        return 2
    src, n = argv[0], int(argv[1])
    ei = np.load(src) if src.endswith(".npy") else np.loadtxt(src, dtype=np.int64).reshape(-1, 2).T
    r = merw_probabilities(n, ei)
    write_edge_input(out, n, ei, r["p_uv"], r["p_vu"])
    print("lambda %.12g after %d iterations; %d rows -> %s" % (r["lam"], r["iters"], 2 * ei.shape[1], out))
    return 0


if __name__ == "__main__":
    sys.exit(main())

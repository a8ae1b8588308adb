"""bench.py's synthetic workload (CPU-only checks): the graph mirrors the shipped edge_input files (SURVEY.md §8a-1:
self loops, every row written twice), shapes are BASELINE.json's configs[1], ranks of a multi-GPU run own equal node
blocks of one common graph, and the oracle accepts the graph."""
import numpy as np

import bench
from oracle import merw


def test_synthetic_graph_mirrors_the_shipped_edge_files():
    n, u, v, p = bench.synthetic_graph(500, 3)
    assert u.dtype == np.int32 and v.dtype == np.int32 and len(u) == len(v) == len(p)
    assert (u[0::2] == u[1::2]).all() and (v[0::2] == v[1::2]).all() and (p[0::2] == p[1::2]).all()   # rows twice
    pairs = set(zip(u.tolist(), v.tolist()))
    assert all((i, i) in pairs for i in range(n))                                                     # self loops
    assert all((b, a) in pairs for a, b in pairs)                                                     # symmetric
    tot = np.zeros(n)
    np.add.at(tot, u[0::2], p[0::2])
    assert np.allclose(tot, 1.0)                       # a transition row sums to one (before the duplication)
    ids, codes = merw.sample_full(n, u, v, p, 5, 4, merw.DRAW_PHILOX, 1, epoch_count=1)
    assert ids.shape == (1, n, 5, 4) and (ids[0, :, :, 0] == np.arange(n)[:, None]).all()
    assert codes.shape == ids.shape and int(codes.max()) <= 4          # hop labels (their values are checked elsewhere)


def test_workload_is_the_cora_configuration_and_shards_evenly():
    w1 = bench.workload(0, 1)
    assert (w1["n"], w1["F"], w1["C"], w1["H"], w1["W"], w1["L"]) == (2708, 1433, 7, 128, 40, 4)
    assert w1["X"].shape == (2708, 1433) and w1["X"].dtype == np.float32
    assert abs(int(w1["mask"].sum()) - int(0.48 * 2708)) <= 1
    wa, wb = bench.workload(0, 2), bench.workload(1, 2)
    assert wa["n"] == wb["n"] == 2 * 2708 and wa["n_loc"] == 2708
    assert (wa["graph"][1] == wb["graph"][1]).all() and (wa["X"] == wb["X"]).all() and (wa["mask"] == wb["mask"]).all()

"""bench.py's synthetic workload (CPU-only checks): the graph mirrors the shipped edge_input files (SURVEY.md §8a-1:
self loops, every row written twice), shapes are BASELINE.json's configs[1], ranks of a multi-GPU run own equal node
blocks of one common graph, and the oracle accepts the graph."""
import os

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


def test_roofline_bound_follows_the_evidence():
    """roofline.bound is the larger of the matrix-pipe fraction (algorithmic flops) and the HBM fraction (PMC bytes of the
    same build); without a PMC stamp only the matrix side is known.  Both sides are always printed."""
    P, L, H = 51960, 4, 128
    no_pmc = bench.roofline_block("seq_bwd", 0.3563, 20, P, L, H, None)
    assert no_pmc["bound"] == "mfma" and no_pmc["unit"] == "TFLOP/s" and no_pmc["bound_evidence"]["hbm_frac"] is None
    assert abs(no_pmc["frac"] - 47.67e9 / 0.3563e-3 / 1e12 / (2500.0 / 3)) < 2e-3            # round 4's 0.16
    r4 = bench.roofline_block("seq_bwd", 0.3563, 20, P, L, H, 1.139e9, l2_requests=3.0e7)
    ev = r4["bound_evidence"]
    assert r4["bound"] == "hbm" and ev["largest"] == "hbm" and 0.39 < ev["hbm_frac"] < 0.41 and 0.15 < ev["mfma_frac"] < 0.17
    assert r4["unit"] == "GB/s" and r4["peak"] == 8000.0 and r4["traffic"] == 1.139e9
    # achieved = ALGORITHMIC bytes (gather backward, 2 * L * H * 4 per path) / duration
    assert abs(r4["achieved"] - P * 4096 / 0.3563e-3 / 1e9) < 1.0 and abs(r4["frac"] - r4["achieved"] / 8000.0) < 1e-3
    assert 5.2 < r4["as_hbm"]["traffic_over_algorithmic"] < 5.5 and r4["as_mfma"]["frac"] == ev["mfma_frac"]
    assert bench.algorithmic_bytes("seq_fwd", 1, 4, 128) == 2068 and bench.algorithmic_bytes("seq_bwd", 1, 4, 128) == 4096
    fast = bench.roofline_block("seq_fwd", 0.05, 20, P, L, H, 1.0e8)          # a launch that is far from its bytes
    assert fast["bound"] == "mfma"


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus N` without WORLD_SIZE re-executes itself under torch.distributed.run"""
    import argparse
    import sys
    old = sys.argv
    sys.argv = ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"]
    try:
        cmd, env = bench.spawn_ranks(argparse.Namespace(gpus=4), dry_run=True)
    finally:
        sys.argv = old
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    import torch
    if torch.cuda.device_count() < 4:                    # fewer devices than ranks -> host-staged collectives through gloo
        assert env.get("PN_DIST_BACKEND") == "gloo"
    else:
        assert "PN_DIST_BACKEND" not in env or env["PN_DIST_BACKEND"] == os.environ.get("PN_DIST_BACKEND")


def _fail_constant(name):
    raise ValueError("non-strict JSON constant %s" % name)


REQUIRED_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                      "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def test_line_is_small_and_strict_json():
    """the driver could not parse round 5's 20 KB stdout line.  The stdout line is a slim, numbers-only
    projection (< 6 KB, strict JSON, every contract key); the full record goes to bench_extras.json / stderr.  Checked on
    round 5's real full record (profiles/r05_bench_fresh_box.json) and on the two-rank one."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("r05_bench_fresh_box.json", "r05_bench_two_ranks_one_gpu.json", "r05_bench_four_ranks_one_gpu.json"):
        full = json.load(open(os.path.join(root, "profiles", name)))
        assert len(json.dumps(full)) > 8000                   # (the record that broke the parse)
        line = bench.slim_line(full)
        assert "\n" not in line and len(line.encode()) < 6144, len(line)
        got = json.loads(line, parse_constant=_fail_constant)
        need = [k for k in REQUIRED_LINE_KEYS if k != "cpu_baseline" or full["n_gpus"] == 1]
        assert not [k for k in need if k not in got], [k for k in need if k not in got]
        assert got["value"] == float("%.6g" % full["value"]) and got["config"]["workload"] == full["config"]["workload"]
        r = got["roofline"]
        assert {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r)
        assert r["frac"] == full["roofline"]["frac"]
        if r["kernel"] in ("seq_fwd", "seq_bwd", "wgrad"):
            assert "frac" in r["as_mfma"] and "frac" in r["as_hbm"]
        if full["n_gpus"] == 1:
            cb = got["cpu_baseline"]
            assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and len(cb["sample"]) <= 160
            assert {"min", "median", "max"} <= set(got["dispersion"]["block_ms_per_step"])
            assert got["gather"]["standalone_frac"] and got["gather"]["fused_frac"]
        assert not any(k in got for k in ("graph_replay", "dtype_note", "device", "hid512_step"))   # extras stay out
    # a pathological record still yields a line under the cap (stages dropped before anything the contract names)
    fat = dict(full, stages_ms={"s%d" % i: 1.0 / 3 for i in range(600)})
    assert len(bench.slim_line(fat)) < 6144 and "roofline" in json.loads(bench.slim_line(fat))


def test_emit_prints_the_slim_line_last_on_stdout(tmp_path, capsys, monkeypatch):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, "profiles", "r05_bench_fresh_box.json")))
    monkeypatch.setenv("PN_BENCH_EXTRAS", str(tmp_path / "extras.json"))
    bench.emit(full)
    cap = capsys.readouterr()
    lines = cap.out.strip().split("\n")
    assert len(lines) == 1 and json.loads(lines[0])["metric"] == full["metric"]
    assert json.loads(cap.err.strip().split("\n")[-1]) == full                       # everything else: stderr ...
    assert json.load(open(tmp_path / "extras.json")) == full                          # ... and the extras file

"""Sampler throughput on a synthetic Erdos-Renyi-like graph too large for the dense hop table
(on-the-fly hop codes), BASELINE.json configs[4] shape: deg ~16, path_num 40, path_len 6.
    python tools/bench_sampler_large.py [n_nodes=2000000] [L=6]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathnet_amd  # noqa: E402


def er_graph(n, deg, seed):
    rng = np.random.default_rng(seed)
    m = n * deg // 2
    a = rng.integers(0, n, m, dtype=np.int64)
    b = rng.integers(0, n, m, dtype=np.int64)
    keep = a != b
    a, b = a[keep], b[keep]
    src = np.concatenate([a, b, np.arange(n)])
    dst = np.concatenate([b, a, np.arange(n)])
    key = np.unique(src * n + dst)
    src, dst = (key // n).astype(np.int32), (key % n).astype(np.int32)
    degs = np.bincount(src, minlength=n)
    return n, src, dst, 1.0 / degs[src]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    W = 40
    t0 = time.time()
    g = er_graph(n, 16, 0)
    t1 = time.time()
    smp = pathnet_amd.MerwSampler(*g, L, hops="otf")
    t2 = time.time()
    chunk = min(n, 500000)
    ids = torch.empty((1, chunk, W, L), dtype=torch.int32, device="cuda")
    codes = torch.empty((1, chunk, W, L), dtype=torch.uint8, device="cuda")
    smp.sample(W, 1, node_begin=0, node_count=chunk, out=(ids, codes))
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    done = 0
    for lo in range(0, n, chunk):
        cnt = min(chunk, n - lo)
        smp.sample(W, 1, node_begin=lo, node_count=cnt, check=False, out=(ids[:, :cnt], codes[:, :cnt]))
        done += cnt * W
    ev1.record()
    torch.cuda.synchronize()
    dt = ev0.elapsed_time(ev1) * 1e-3
    hist = torch.bincount(codes.flatten().to(torch.int64), minlength=L).tolist()
    print(json.dumps({"nodes": n, "edge_rows": int(len(g[1])), "W": W, "L": L, "hops": smp.hops,
                      "sampled_paths_per_s": done / dt, "seconds_per_epoch": dt,
                      "host_graph_s": t1 - t0, "host_tables_s": t2 - t1, "code_histogram_last_chunk": hist}))


if __name__ == "__main__":
    main()

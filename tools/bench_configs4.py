"""BASELINE.json configs[4] on ONE GPU: synthetic Erdos-Renyi graph, 10 M nodes, degree ~16, feat = 128, path_num = 40,
path_len = 6 -- the exact on-the-fly sampler + one full training step of the aggregator over a batch of masked nodes
(micro-batched inside the library; the node tables Z / dZ [N, 6, 128] are 30.7 GB each).
    python tools/bench_configs4.py [n_nodes=10000000] [masked=400000] [steps=2]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pathnet_amd  # noqa: E402
from pathnet_amd import _lib, modules  # noqa: E402
from bench_sampler_large import er_graph  # noqa: E402


def run(n=10_000_000, S=400_000, steps=2):
    F, H, C, W, L = 128, 128, 8, 40, 6
    dev = torch.device("cuda")
    t0 = time.time()
    g = er_graph(n, 16, 0)
    t1 = time.time()
    smp = pathnet_amd.MerwSampler(*g, L, hops="otf")
    t2 = time.time()
    torch.manual_seed(0)
    model = pathnet_amd.PathNet_homo(F, H, C, L, dropout=0.7).to(dev).train()
    opt = pathnet_amd.Adam(model.parameters(), lr=0.005, weight_decay=0.0005)
    lossf = pathnet_amd.CrossEntropyLoss()
    X = torch.rand((n, F), device=dev)
    rng = np.random.default_rng(1)
    sel = torch.from_numpy(np.sort(rng.choice(n, S, replace=False)).astype(np.int32)).to(dev)
    Y = torch.randint(0, C, (S,), device=dev)
    ids = torch.empty((1, S, W, L), dtype=torch.int32, device=dev)
    codes = torch.empty((1, S, W, L), dtype=torch.uint8, device=dev)
    lib = _lib.load()
    ctx = _lib.context(dev)
    import bench
    names = bench.stage_names(lib)

    fused = os.environ.get("PN_BENCH_UNFUSED", "0") in ("", "0")

    def step(e):
        smp.sample(W, 77, epoch_begin=e, epoch_count=1, nodes=sel, check=False, out=(ids, codes))
        if fused:       # pn_pagg_train_step: a micro-batch's forward, loss gradient and backward follow each other
            loss, _ = model.forward_loss(X, ids[0], W, L, sel, codes[0], Y, fused=True)
        else:           # forward (keeps nothing), loss, backward (re-runs every micro-batch's recurrence)
            out = model(X, ids[0], W, L, sel, codes[0], None)
            loss = lossf(out, Y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return float(loss.detach())     # (a live loss would keep the step's graph -- and its 74 GB workspace -- alive)

    step(0)
    torch.cuda.synchronize()
    _lib.check(lib.pn_profile_configure(ctx, 1, -1))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for e in range(steps):
        loss = step(1 + e)
    ev1.record()
    torch.cuda.synchronize()
    prof = bench.read_profile(lib, names, ctx)
    _lib.check(lib.pn_profile_configure(ctx, 0, -1))
    dt = ev0.elapsed_time(ev1) * 1e-3 / steps
    bg = modules.pick_batch_groups("homo", n, F, H, C, S, W, L, budget=model.workspace_budget)
    return {
        "config": "configs[4] on one GPU: Erdos-Renyi n=%d deg~16 (%d edge rows), feat=%d hid=%d path_num=%d path_len=%d, "
                  "%d masked nodes = %d paths/step, PathNet_homo, dropout 0.7, Adam; exact on-the-fly hop codes" %
                  (n, len(g[1]), F, H, W, L, S, S * W),
        "seconds_per_step": dt, "paths_per_s": S * W / dt, "loss": loss,
        "compact_rows": (os.environ["PN_COMPACT"] != "0") if "PN_COMPACT" in os.environ else 2 * S * W * L < n * L,
        "step_calls": "pn_pagg_train_step (forward + loss + backward per micro-batch)" if fused else
                      "pn_pagg_forward, pn_cross_entropy, pn_pagg_backward (the backward re-runs each micro-batch's forward)",
        "micro_batch_nodes": bg, "micro_batches": (S + bg - 1) // bg if bg else 1,
        "workspace_GB": round(modules.workspace_bytes("homo", n, F, H, C, S, W, L, batch_groups=bg) / 2 ** 30, 1),
        "torch_max_allocated_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
        "stage_ms_per_step": {k: round(v[0] / steps, 2) for k, v in prof.items()},
        "host_graph_s": round(t1 - t0, 1), "host_sampler_tables_s": round(t2 - t1, 1)}


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*a)))

"""Does running the training step in micro-batches that fit the 256 MB Infinity Cache pay?  (pn_pagg_train_step: forward, loss
and backward of a micro-batch back to back, so the BPTT finds the forward's saved tensors and the weight-gradient GEMM the
BPTT's gate gradients in the memory-side cache instead of HBM.)   python tools/ab_chunks.py [pubmed|cora] [groups ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pathnet_amd
from pathnet_amd import modules


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "pubmed"
    groups = [int(a) for a in sys.argv[2:]] or [0, 4800, 2400, 1200, 600, 300]
    wl = bench.workload(0, 1) if which == "cora" else bench.pubmed_workload()
    dev = torch.device("cuda")
    gn, u, v, p = wl["graph"]
    smp = pathnet_amd.MerwSampler(gn, u, v, p, wl["L"], device=dev)
    torch.manual_seed(0)
    model = pathnet_amd.PathNet_homo(wl["F"], wl["H"], wl["C"], wl["L"], dropout=0.7).to(dev).train()
    X = torch.from_numpy(wl["X"]).to(dev)
    sel = torch.from_numpy(np.flatnonzero(wl["mask"]).astype(np.int64)).to(dev)
    ids, codes = smp.sample(wl["W"], 1, epoch_count=1)
    ids, codes = ids[0].index_select(0, sel), codes[0].index_select(0, sel)
    y = torch.from_numpy(wl["Y"]).to(dev)[sel]
    sel32 = sel.to(torch.int32)
    real = modules.pick_batch_groups
    for bg in groups:
        modules.pick_batch_groups = (lambda *a, **k: bg) if bg else real

        def step():
            loss, _ = model.forward_loss(X, ids, wl["W"], wl["L"], sel32, codes, y, fused=True)
            model.zero_grad(set_to_none=True)
            loss.backward()
        for _ in range(3):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        S = int(sel.numel())
        print("%s: micro-batch %5d nodes (%3d micro-batches, %.0f MB of saved tensors + gate gradients each): %.3f ms per step"
              % (which, bg or S, -(-S // (bg or S)), (bg or S) * wl["W"] * wl["L"] * 11 * wl["H"] * 4 / 1e6,
                 e0.elapsed_time(e1) / 10))


if __name__ == "__main__":
    main()

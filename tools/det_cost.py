"""What the fixed-order backward (pn_pagg_shape.deterministic) costs: forward + backward of the aggregator at the headline
shape (configs[1]: 2708 nodes, 1433 features, 1299... masked nodes x 40 paths x 4 steps, hidden 128) with the default
(atomic) backward and with the deterministic one, plus how many gradient elements differ between two runs of each.

    python tools/det_cost.py [--S 1299] [--reps 30]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathnet_amd  # noqa: E402
from pathnet_amd import modules as M  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=2708)
    ap.add_argument("--F", type=int, default=1433)
    ap.add_argument("--S", type=int, default=1299)
    ap.add_argument("--W", type=int, default=40)
    ap.add_argument("--L", type=int, default=4)
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    torch.manual_seed(0)
    m = pathnet_amd.PathNet_homo(a.F, 128, 7, a.L, dropout=0.5).cuda().train()
    X = torch.rand(a.N, a.F, generator=g).cuda()
    sel = torch.randperm(a.N, generator=g)[:a.S].sort().values.to(torch.int32)
    ids = torch.randint(0, a.N, (a.S, a.W, a.L), generator=g).to(torch.int32)
    ids[:, :, 0] = sel[:, None]
    codes = torch.randint(0, a.L, (a.S, a.W, a.L), generator=g).to(torch.uint8)
    G = torch.randn(a.S, 7, generator=g).cuda()
    ids, codes, sel = ids.cuda(), codes.cuda(), sel.cuda()

    def step(seed=7):
        torch.manual_seed(seed)
        m.zero_grad(set_to_none=True)
        out = m(X, ids, a.W, a.L, sel, codes, None)
        out.backward(G)

    res = {"shape": vars(a)}
    for det in (False, True):
        m.deterministic = det
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            step()
        e1.record()
        torch.cuda.synchronize()
        step()
        g0 = {k: v.grad.clone() for k, v in m.named_parameters()}
        step()
        differing = sum(int((v.grad != g0[k]).sum().item()) for k, v in m.named_parameters())
        total = sum(v.numel() for v in m.parameters())
        res["deterministic" if det else "default"] = {
            "ms_per_forward_backward": e0.elapsed_time(e1) / a.reps,
            "gradient_elements_differing_between_two_runs": differing, "gradient_elements": total,
            "workspace_MB": M.workspace_bytes("homo", a.N, a.F, 128, 7, a.S, a.W, a.L, deterministic=det) / 1e6}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

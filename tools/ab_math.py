"""Stage times of the aggregator step with the recurrent GEMMs as fp16 x 2 planes vs bf16 x 3 planes (pn_pagg_shape.seq_math),
same process, same inputs:   python tools/ab_math.py [cora|pubmed] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pathnet_amd
from pathnet_amd import _lib


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "cora"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    lib = _lib.load()
    names = bench.stage_names(lib)
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
    G = torch.randn(sel.numel(), wl["C"], device=dev)
    sel32 = sel.to(torch.int32)

    def step():
        out = model(X, ids, wl["W"], wl["L"], sel32, codes, None)
        model.zero_grad(set_to_none=True)
        out.backward(G)

    def fwd_only():
        with torch.no_grad():
            model(X, ids, wl["W"], wl["L"], sel32, codes, None)

    res = {}
    for rnd in range(2):
        for math in ("bf16x3", "f16x2"):
            model.seq_math = math
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            _lib.check(lib.pn_profile_configure(_lib.context("cuda"), 1, -1))
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            prof = bench.read_profile(lib, names)
            _lib.check(lib.pn_profile_configure(_lib.context("cuda"), 0, -1))
            for _ in range(5):
                step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                step()
            e1.record()
            torch.cuda.synchronize()
            d = {k: round(v[0] / v[1], 4) for k, v in prof.items()}
            d["_wall_fwd_bwd"] = round(e0.elapsed_time(e1) / 50, 4)
            model.eval()
            for _ in range(3):
                fwd_only()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                fwd_only()
            e1.record()
            torch.cuda.synchronize()
            d["_wall_eval_fwd"] = round(e0.elapsed_time(e1) / 50, 4)
            model.train()
            res["%s#%d" % (math, rnd)] = d
            print("%-8s round %d: fwd %.3f bwd %.3f wgrad %.3f plan_pack %.3f bank %.3f | stages %.3f | wall fwd+bwd %.3f | eval fwd %.3f"
                  % (math, rnd, d.get("seq_fwd", -1), d.get("seq_bwd", -1), d.get("wgrad", -1), d.get("plan_pack", -1),
                     d.get("bank", -1), sum(v for k, v in d.items() if k[0] != "_"), d["_wall_fwd_bwd"], d["_wall_eval_fwd"]))
    print("RESULT " + json.dumps({"workload": which, "paths": int(sel.numel()) * wl["W"], "stages_ms": res}))


if __name__ == "__main__":
    main()

"""Stage times of the aggregator step with the default (atomic) backward and the fixed-order one (pn_pagg_shape.deterministic),
same process, same inputs:   python tools/ab_det.py [cora|pubmed|bgp] [steps]"""
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
    wl = bench.workload(0, 1) if which == "cora" else bench.pubmed_workload() if which == "pubmed" else bench.bgp_workload()
    dev = torch.device("cuda")
    gn, u, v, p = wl["graph"]
    smp = pathnet_amd.MerwSampler(gn, u, v, p, wl["L"], device=dev)
    torch.manual_seed(0)
    cls = getattr(pathnet_amd, wl.get("cls", "PathNet_homo"))
    model = cls(wl["F"], wl["H"], wl["C"], wl["L"], dropout=0.7).to(dev).train()
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

    res = {}
    for rnd in range(2):
        for det in (False, True):
            model.deterministic = det
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            _lib.check(lib.pn_profile_configure(_lib.context("cuda"), 1, -1))
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            prof = bench.read_profile(lib, names)
            _lib.check(lib.pn_profile_configure(_lib.context("cuda"), 0, -1))
            for _ in range(3):
                step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                step()
            e1.record()
            torch.cuda.synchronize()
            d = {k: round(v[0] / v[1], 4) for k, v in prof.items()}
            d["_wall_fwd_bwd"] = round(e0.elapsed_time(e1) / 20, 4)
            res["%s#%d" % ("det" if det else "default", rnd)] = d
            print("%-8s round %d: %s | wall %.3f" % ("det" if det else "default", rnd,
                  " ".join("%s %.3f" % (k, d[k]) for k in ("pool_bwd", "seq_bwd", "wgrad", "bank_bwd", "fc0_bwd", "fc2_grad") if k in d),
                  d["_wall_fwd_bwd"]))
    print("RESULT " + json.dumps({"workload": which, "stages_ms": res}))


if __name__ == "__main__":
    main()

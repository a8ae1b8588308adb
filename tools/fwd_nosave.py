"""Forward stage times with and without the tensors saved for the backward (how much of seq_fwd is its stores?)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, pathnet_amd
from pathnet_amd import _lib
lib = _lib.load()
names = bench.stage_names(lib)
wl = bench.workload(0, 1)
dev = torch.device("cuda")
gn, u, v, p = wl["graph"]
smp = pathnet_amd.MerwSampler(gn, u, v, p, wl["L"], device=dev)
torch.manual_seed(0)
model = pathnet_amd.PathNet_homo(wl["F"], wl["H"], wl["C"], wl["L"], dropout=0.7).to(dev).train()
X = torch.from_numpy(wl["X"]).to(dev)
sel = torch.from_numpy(np.flatnonzero(wl["mask"]).astype(np.int64)).to(dev)
ids, codes = smp.sample(wl["W"], 1, epoch_count=1)
ids, codes = ids[0].index_select(0, sel), codes[0].index_select(0, sel)
for grad in (True, False):
    def step():
        with torch.set_grad_enabled(grad):
            return model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
    for _ in range(3): step()
    torch.cuda.synchronize()
    _lib.check(lib.pn_profile_configure(_lib.context("cuda"), 1, -1))
    for _ in range(20): step()
    torch.cuda.synchronize()
    prof = bench.read_profile(lib, names)
    _lib.check(lib.pn_profile_configure(_lib.context("cuda"), 0, -1))
    print("saved tensors written" if grad else "no_save", json.dumps({k: round(v[0] / v[1], 4) for k, v in prof.items()}))

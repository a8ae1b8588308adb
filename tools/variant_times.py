"""Stage times of the three module variants on the bench workload (sanity check that none is pathologically slow)."""
import json, os, sys
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
X = torch.from_numpy(wl["X"]).to(dev)
sel = torch.from_numpy(np.flatnonzero(wl["mask"]).astype(np.int64)).to(dev)
ids, codes = smp.sample(wl["W"], 1, epoch_count=1)
ids, codes = ids[0].index_select(0, sel), codes[0].index_select(0, sel)
G = torch.randn(sel.numel(), wl["C"], device=dev)
for name, cls, arg in (("homo", pathnet_amd.PathNet_homo, wl["L"]), ("hetero", pathnet_amd.PathNet, wl["L"]),
                       ("pagg", pathnet_amd.PAGG, wl["n"])):
    torch.manual_seed(0)
    model = cls(wl["F"], wl["H"], wl["C"], arg, dropout=0.7).to(dev).train() if name != "pagg" else \
        cls(wl["F"], wl["H"], wl["C"], arg).to(dev).train()
    def step():
        out = model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
        model.zero_grad(set_to_none=True)
        out.backward(G)
    for _ in range(3): step()
    torch.cuda.synchronize()
    _lib.check(lib.pn_profile_configure(_lib.context("cuda"), 1, -1))
    for _ in range(10): step()
    torch.cuda.synchronize()
    prof = bench.read_profile(lib, names)
    _lib.check(lib.pn_profile_configure(_lib.context("cuda"), 0, -1))
    d = {k: round(v[0] / v[1], 4) for k, v in prof.items()}
    print(name, "total %.3f" % sum(d.values()), json.dumps(d))

import os, sys, json
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import numpy as np, torch
import bench, pathnet_amd
from pathnet_amd import _lib
lib = _lib.load(); names = bench.stage_names(lib); ctx = _lib.context("cuda")
wl = bench.workload(0, 1); dev = torch.device("cuda")
gn, u, v, p = wl["graph"]
smp = pathnet_amd.MerwSampler(gn, u, v, p, wl["L"], device=dev)
torch.manual_seed(0)
model = pathnet_amd.PathNet_homo(wl["F"], wl["H"], wl["C"], wl["L"], dropout=0.7).to(dev).train()
X = torch.from_numpy(wl["X"]).to(dev)
allsel = torch.arange(wl["n"], device=dev)
ids_all, codes_all = smp.sample(wl["W"], 1, epoch_count=1)
for S in (614, 1228, 1299, 1536, 1843, 2457, 2708):
    sel = allsel[:S]
    ids, codes = ids_all[0].index_select(0, sel), codes_all[0].index_select(0, sel)
    G = torch.randn(S, wl["C"], device=dev)
    def step():
        out = model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
        model.zero_grad(set_to_none=True); out.backward(G)
    for _ in range(3): step()
    torch.cuda.synchronize()
    _lib.check(lib.pn_profile_configure(ctx, 1, -1))
    for _ in range(10): step()
    torch.cuda.synchronize()
    prof = bench.read_profile(lib, names)
    _lib.check(lib.pn_profile_configure(ctx, 0, -1))
    d = {k: v[0] / v[1] for k, v in prof.items()}
    P = S * wl["W"]
    print("S %5d P %6d tiles32 %5d rounds768 %.2f rounds512 %.2f | fwd %.3f bwd %.3f wgrad %.3f | per 1k paths: fwd %.4f bwd %.4f wgrad %.4f" % (
        S, P, (P + 31) // 32, P / 32 / 768, P / 32 / 512, d["seq_fwd"], d["seq_bwd"], d["wgrad"], d["seq_fwd"] / P * 1e3, d["seq_bwd"] / P * 1e3, d["wgrad"] / P * 1e3), flush=True)

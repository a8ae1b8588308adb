"""Phase timeline of the recurrent kernels (needs a -DPN_TRACE_PHASES=1 build given by PN_LIB_PATH).
Prints, per kernel, mean cycles of [gather/cell-bwd | MFMA | cell/scatter] per step for the workgroups."""
import ctypes
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import pathnet_amd
from pathnet_amd import _lib

lib = ctypes.CDLL(_lib.LIB_PATH)
lib.pn_debug_set_trace.argtypes = [ctypes.c_void_p]
wl = bench.workload(0, 1)
dev = torch.device("cuda")
gn, u, v, p = wl["graph"]
smp = pathnet_amd.MerwSampler(gn, u, v, p, wl["L"], device=dev)
torch.manual_seed(0)
model = pathnet_amd.PathNet_homo(wl["F"], wl["H"], wl["C"], wl["L"], dropout=0.7).to(dev).train()
X = torch.from_numpy(wl["X"]).to(dev)
sel = torch.from_numpy(np.flatnonzero(wl["mask"]).astype(np.int64)).to(dev)
if os.environ.get("PN_TRACE_NODES"):        # fewer workgroups: how much of a phase is contention?
    sel = sel[:int(os.environ["PN_TRACE_NODES"])]
ids, codes = smp.sample(wl["W"], 1, epoch_count=1)
ids, codes = ids[0].index_select(0, sel), codes[0].index_select(0, sel)
G = torch.randn(sel.numel(), wl["C"], device=dev)
nblk = (sel.numel() * wl["W"] + 31) // 32
for which in ("fwd", "bwd"):
    buf = torch.zeros((nblk, 64), dtype=torch.int64, device=dev)
    for it in range(3):
        lib.pn_debug_set_trace(None)
        out = model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
        model.zero_grad(set_to_none=True)
        if which == "bwd":
            torch.cuda.synchronize()
            lib.pn_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
        out.backward(G)
        torch.cuda.synchronize()
        if which == "fwd":
            lib.pn_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
            with torch.no_grad():
                pass
    if which == "fwd":
        buf.zero_()
        lib.pn_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
        out = model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
        torch.cuda.synchronize()
        lib.pn_debug_set_trace(None)
    t = buf.cpu().numpy().astype(np.float64)[:, :16].reshape(nblk, 4, 4)
    ok = (t > 0).all(axis=(1, 2))
    t = t[ok]
    d_pre = t[:, :, 1] - t[:, :, 0]
    d_mfma = t[:, :, 2] - t[:, :, 1]
    d_post = t[:, :, 3] - t[:, :, 2]
    life = t[:, 3, 3] - t[:, 0, 0]
    print(which, "workgroups", len(t), "life cycles mean %.0f" % life.mean())
    for s in range(4):
        print("  step %d: pre %.0f  mfma %.0f  post %.0f" % (s, d_pre[:, s].mean(), d_mfma[:, s].mean(), d_post[:, s].mean()))
    span = t.max() - t.min()
    print("  kernel span %.0f cycles" % span)

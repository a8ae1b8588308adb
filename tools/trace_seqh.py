"""Phase timeline of the fp16 recurrent kernels (needs a -DPN_TRACE_H=1 build given by PN_LIB_PATH, tools/seqh_variants.sh).
Mean cycles per phase and step over the workgroups, and how many workgroups a CU held at a time."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pathnet_amd
from pathnet_amd import _lib

lib = ctypes.CDLL(_lib.LIB_PATH)
lib.pn_debug_set_trace_h.argtypes = [ctypes.c_void_p]
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
L = wl["L"]
nblk = (sel.numel() * wl["W"] + 31) // 32
sel32 = sel.to(torch.int32)
for _ in range(3):
    out = model(X, ids, wl["W"], L, sel32, codes, None)
    model.zero_grad(set_to_none=True)
    out.backward(G)
torch.cuda.synchronize()
for which, per, names in (("fwd", 4, ["k loop", "barrier", "cell + commit"]),
                          ("bwd", 6, ["loads + cell bwd", "barrier A", "split + LDS + barrier B", "k loop", "scatter"])):
    buf = torch.zeros((nblk, 64), dtype=torch.int64, device=dev)
    if which == "fwd":
        lib.pn_debug_set_trace_h(ctypes.c_void_p(buf.data_ptr()))
        out = model(X, ids, wl["W"], L, sel32, codes, None)
        torch.cuda.synchronize()
        lib.pn_debug_set_trace_h(None)
    else:
        out = model(X, ids, wl["W"], L, sel32, codes, None)
        model.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        lib.pn_debug_set_trace_h(ctypes.c_void_p(buf.data_ptr()))
        out.backward(G)
        torch.cuda.synchronize()
        lib.pn_debug_set_trace_h(None)
    t = buf.cpu().numpy().astype(np.float64)[:, :per * L].reshape(nblk, L, per)
    ok = (t > 0).all(axis=(1, 2))
    t = t[ok]
    life = t[:, L - 1, per - 1] - t[:, 0, 0]
    span = t.max() - t.min()
    print("%s: %d workgroups, life %.0f cycles mean, kernel span %.0f cycles, mean concurrency %.1f workgroups"
          % (which, len(t), life.mean(), span, life.sum() / span))
    # dispatch timeline from the device-wide 100 MHz counter (slots 62 / 63; the cycle counters are per CU).  "drain" = the
    # moment the last workgroup started: from there on the launch only empties -- the tail a launch cannot avoid when its
    # workgroup count is not a multiple of the resident slots
    rt = buf.cpu().numpy().astype(np.float64)[ok][:, 62:64] * 10.0e-3        # us
    st, en = rt[:, 0], rt[:, 1]
    t0, t1, drain = st.min(), en.max(), st.max()
    busy_tail = np.clip(en - np.maximum(st, drain), 0, None).sum()
    busy_head = (en - st).sum() - busy_tail
    print("  timeline: first start -> last end %.1f us; last dispatch at %.1f us, %.1f us (%.0f %%) before the end; workgroups "
          "in flight: %.0f before, %.0f after; workgroup life %.1f us mean, %.1f us for the last 5 %% dispatched"
          % (t1 - t0, drain - t0, t1 - drain, 100 * (t1 - drain) / (t1 - t0), busy_head / max(drain - t0, 1e-9),
             busy_tail / max(t1 - drain, 1e-9), (en - st).mean(), (en - st)[np.argsort(st)[-len(st) // 20:]].mean()))
    edges = np.linspace(t0, t1, 11)
    print("  in flight per tenth of the launch: " + " ".join(
        "%.0f" % (np.clip(np.minimum(en, edges[i + 1]) - np.maximum(st, edges[i]), 0, None).sum() / (edges[i + 1] - edges[i]))
        for i in range(10)))
    for s in range(L):
        parts = ["%s %.0f" % (names[i], (t[:, s, i + 1] - t[:, s, i]).mean()) for i in range(per - 1)]
        gap = (t[:, s + 1, 0] - t[:, s, per - 1]).mean() if s + 1 < L else 0.0
        print("  step %d: %s | to next step %.0f" % (s, " | ".join(parts), gap))

"""Phase timeline of seq_fwd4_kernel (needs a -DPN_TRACE4=1 build given by PN_LIB_PATH; PN_SEQ4=1).
Stamps of waves 0 (row group 0) and 4 (row group 1) of every workgroup: per k-step [top | before products | after
products | after commit | after barrier], per step [k-loop end | cell update end]."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import pathnet_amd
from pathnet_amd import _lib

os.environ.setdefault("PN_SEQ4", "1")
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.pn_debug_set_trace4.argtypes = [ctypes.c_void_p]
wl = bench.workload(0, 1)
dev = torch.device("cuda")
gn, u, v, p = wl["graph"]
smp = pathnet_amd.MerwSampler(gn, u, v, p, wl["L"], device=dev)
torch.manual_seed(0)
model = pathnet_amd.PathNet_homo(wl["F"], wl["H"], wl["C"], wl["L"], dropout=0.7).to(dev).train()
X = torch.from_numpy(wl["X"]).to(dev)
sel = torch.from_numpy(np.flatnonzero(wl["mask"]).astype(np.int64)).to(dev)
if os.environ.get("PN_TRACE_NODES"):
    sel = sel[:int(os.environ["PN_TRACE_NODES"])]
ids, codes = smp.sample(wl["W"], 1, epoch_count=1)
ids, codes = ids[0].index_select(0, sel), codes[0].index_select(0, sel)
nblk = (sel.numel() * wl["W"] + 127) // 128
SL = 512
buf = torch.zeros((nblk, 2, SL), dtype=torch.int64, device=dev)
for it in range(3):
    out = model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
torch.cuda.synchronize()
lib.pn_debug_set_trace4(ctypes.c_void_p(buf.data_ptr()))
out = model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
torch.cuda.synchronize()
lib.pn_debug_set_trace4(None)
t = buf.cpu().numpy().astype(np.float64)
L, KX, KS = wl["L"], 8, 16
nk = KX + (L - 1) * KS
first = t[:, 0, 8]          # first stamp of wave 0
order = np.argsort(first)
early = order[:min(256, nblk)]      # the first round of workgroups
for name, blocks in (("first-round workgroups", early), ("later workgroups", order[min(256, nblk):])):
    if len(blocks) == 0:
        continue
    print(name, len(blocks))
    for w in (0, 1):
        tt = t[blocks, w]
        ks = tt[:, 8:8 + 5 * nk].reshape(len(blocks), nk, 5)
        d = np.diff(ks, axis=2)                                   # top->pre, pre->post products, post->commit, commit->barrier
        nxt = ks[:, 1:, 0] - ks[:, :-1, 4]                        # barrier -> next top (cell update in between at step ends)
        print("  wave %d (row group %d): per k-step mean cycles  aux/top %.0f  products %.0f  aux/commit %.0f  barrier wait %.0f   total %.0f" % (
            4 * w, w, d[:, :, 0].mean(), d[:, :, 1].mean(), d[:, :, 2].mean(), d[:, :, 3].mean(), (ks[:, :, 4] - ks[:, :, 0]).mean()))
        xs = np.r_[0:KX]
        hs = np.r_[KX + KX:KX + KS]
        for nm, idx in (("x k-steps of step 0", xs), ("h k-steps of step 1", hs)):
            print("     %-22s aux/top %.0f  products %.0f  aux/commit %.0f  barrier wait %.0f" % (
                nm, d[:, idx, 0].mean(), d[:, idx, 1].mean(), d[:, idx, 2].mean(), d[:, idx, 3].mean()))
        cell = tt[:, 1:2 * L:2] - tt[:, 0:2 * L:2]
        print("     cell update per step:", " ".join("%.0f" % c for c in cell.mean(axis=0)),
              "  kernel life %.0f" % (tt[:, 2 * L - 1] - tt[:, 8]).mean())
span = t[:, :, :8 + 5 * nk][t[:, :, :8 + 5 * nk] > 0]
print("span of all stamps %.0f cycles (100 MHz timer ticks if s_memtime is the constant clock)" % (span.max() - span.min()))

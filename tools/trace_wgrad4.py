"""Phase timeline of wgrad4_kernel (needs a -DPN_TRACE4=1 build given by PN_LIB_PATH): per K tile of 16 rows and workgroup,
waves 0 and 4 stamp  0 top of the step | 1 rows of the next tile have arrived | 2 arrives at the barrier (four MFMA groups +
the commit issued) | 3 leaves the barrier | 4 end of the step (two more groups + the next tile's first fragment reads issued)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, pathnet_amd
from pathnet_amd import _lib
os.environ.setdefault("PN_SEQ4", "4")
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.pn_debug_set_trace4.argtypes = [ctypes.c_void_p]
wl = bench.workload(0, 1); dev = torch.device("cuda")
gn, u, v, p = wl["graph"]
smp = pathnet_amd.MerwSampler(gn, u, v, p, wl["L"], device=dev)
torch.manual_seed(0)
model = pathnet_amd.PathNet_homo(wl["F"], wl["H"], wl["C"], wl["L"], dropout=0.7).to(dev).train()
X = torch.from_numpy(wl["X"]).to(dev)
sel = torch.from_numpy(np.flatnonzero(wl["mask"]).astype(np.int64)).to(dev)
ids, codes = smp.sample(wl["W"], 1, epoch_count=1)
ids, codes = ids[0].index_select(0, sel), codes[0].index_select(0, sel)
G = torch.randn(sel.numel(), wl["C"], device=dev)
nblk = 4096
buf = torch.zeros((nblk, 2, 512), dtype=torch.int64, device=dev)
for it in range(3):
    out = model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
    model.zero_grad(set_to_none=True); out.backward(G)
out = model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
model.zero_grad(set_to_none=True)
torch.cuda.synchronize()
lib.pn_debug_set_trace4(ctypes.c_void_p(buf.data_ptr()))
out.backward(G)
torch.cuda.synchronize()
lib.pn_debug_set_trace4(None)
t = buf.cpu().numpy().astype(np.float64)
used = np.flatnonzero(t[:, 0, 0] > 0)
print("workgroups with stamps:", len(used))
NT = 30
for w in (0, 1):
    tt = t[used, w, :NT * 5 + 1].reshape(len(used), -1)
    st = tt[:, :NT * 5].reshape(len(used), NT, 5)
    nxt = np.concatenate([st[:, 1:, 0], tt[:, NT * 5:NT * 5 + 1]], axis=1)
    ok = (st > 0).all(axis=2) & (nxt > 0)
    d = np.diff(st, axis=2)
    names = ("wait rows", "4 groups + commit", "barrier", "2 groups + reads")
    print("wave %d:" % (4 * w), "  ".join("%s %.0f" % (n, d[:, :, k][ok].mean()) for k, n in enumerate(names)),
          " | period %.0f cycles (100 MHz ticks x shader/100: see clock probe)" % (nxt - st[:, :, 0])[ok].mean())

"""Phase timeline of wgrad3_kernel (needs a -DPN_TRACE_PHASES=1 build given by PN_LIB_PATH): per K tile
[wait loads | split + LDS write | barrier | MFMA (+ next loads issued) | barrier]."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, pathnet_amd
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
ids, codes = smp.sample(wl["W"], 1, epoch_count=1)
ids, codes = ids[0].index_select(0, sel), codes[0].index_select(0, sel)
G = torch.randn(sel.numel(), wl["C"], device=dev)
buf = torch.zeros((4096, 64), dtype=torch.int64, device=dev)
for it in range(3):
    out = model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
    model.zero_grad(set_to_none=True)
    out.backward(G)
torch.cuda.synchronize()
out = model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
model.zero_grad(set_to_none=True)
torch.cuda.synchronize()
buf.zero_()
lib.pn_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
out.backward(G)
torch.cuda.synchronize()
lib.pn_debug_set_trace(None)
t = buf.cpu().numpy().astype(np.float64)
# the recurrent kernels stamp slots < 16 of rows < n_tiles too; wgrad rows are the first 256 and use slots 0..59
t = t[:256, :60].reshape(256, 12, 5)
ok = (t > 0).all(axis=(1, 2))
t = t[ok]
print("workgroups with complete stamps:", len(t))
names = ["wait loads", "split+LDS write", "barrier 1", "MFMA (+issue)", "barrier 2 (to next tile top)"]
d = np.diff(t, axis=2)                       # 4 intervals inside a tile
nxt = t[:, 1:, 0] - t[:, :-1, 4]             # barrier 2
for i in range(4):
    print("  %-28s %8.0f cycles" % (names[i], d[:, :, i].mean()))
print("  %-28s %8.0f cycles" % (names[4], nxt.mean()))
print("  tile period %.0f" % (t[:, 1:, 0] - t[:, :-1, 0]).mean())

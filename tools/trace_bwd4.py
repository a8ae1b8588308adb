"""Phase timeline of seq_bwd4_kernel (needs a -DPN_TRACE4=1 build given by PN_LIB_PATH; PN_SEQ4 with bit 1 set)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, pathnet_amd
from pathnet_amd import _lib
os.environ.setdefault("PN_SEQ4", "2")
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
nblk = (sel.numel() * wl["W"] + 127) // 128
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
L, NS = wl["L"], 16
order = np.argsort(t[:, 0, 0])
for name, blocks in (("first-round workgroups", order[:256]), ("later workgroups", order[256:])):
    if len(blocks) == 0: continue
    print(name, len(blocks))
    for w in (0, 1):
        tt = t[blocks, w]
        for ti in range(L):
            st = tt[:, 8 * ti: 8 * ti + 6]
            d = np.diff(st, axis=1).mean(axis=0)
            ks = tt[:, 64 + ti * NS * 4: 64 + (ti + 1) * NS * 4].reshape(len(blocks), NS, 4)
            kd = np.diff(ks, axis=2).mean(axis=(0, 1))
            per = (ks[:, 1:, 0] - ks[:, :-1, 0]).mean()
            print("  wave %d step %d: cell %.0f  drain %.0f  sync+A0 %.0f  gemm %.0f  scatter %.0f | per stage: mfma %.0f  wait+commit %.0f  barrier %.0f  period %.0f" % (
                4 * w, ti, d[0], d[1], d[2], d[3], d[4], kd[0], kd[1], kd[2], per))
        print("  wave %d life %.0f" % (4 * w, (tt[:, 8 * (L - 1) + 5] - tt[:, 0]).mean()))

"""Time the aggregator stages of every variant library in pathnet_amd/csrc/_variants (GPU box).
Each variant runs in its own process (PN_LIB_PATH).  python tools/tune_run.py [steps]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(ROOT, "pathnet_amd", "csrc", "_variants")

CHILD = r'''
import sys, json, ctypes
sys.path.insert(0, %r)
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
G = torch.randn(sel.numel(), wl["C"], device=dev)
def step():
    out = model(X, ids, wl["W"], wl["L"], sel.to(torch.int32), codes, None)
    model.zero_grad(set_to_none=True)
    out.backward(G)
for _ in range(3): step()
torch.cuda.synchronize()
_lib.check(lib.pn_profile_configure(_lib.context("cuda"), 1, -1))
for _ in range(%d): step()
torch.cuda.synchronize()
prof = bench.read_profile(lib, names)
_lib.check(lib.pn_profile_configure(_lib.context("cuda"), 0, -1))
for _ in range(5): step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(50): step()
e1.record()
torch.cuda.synchronize()
res = {k: round(v[0] / v[1], 4) for k, v in prof.items()}
res["_wall_fwd_bwd"] = round(e0.elapsed_time(e1) / 50, 4)
print("RESULT " + json.dumps(res))
'''


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    specs = [l.strip().split(" ", 1) for l in open(os.path.join(VAR, "specs.txt")) if l.strip()]
    for n, spec in [(s[0], s[1] if len(s) > 1 else "") for s in specs]:
        env = dict(os.environ, PN_LIB_PATH=os.path.join(VAR, "lib_%s.so" % n))
        env.update(tok[4:].split("=", 1) for tok in spec.split() if tok.startswith("env:"))     # run-time knobs of the variant
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, steps)], env=env, capture_output=True, text=True)
        res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if res:
            d = json.loads(res[0][7:])
            print("%-60s fwd %.3f bwd %.3f wgrad %.3f | total %.3f" % (spec, d.get("seq_fwd", -1), d.get("seq_bwd", -1),
                                                                      d.get("wgrad", -1),
                                                                      sum(v for k, v in d.items() if k[0] != "_")) +
                  " | wall fwd+bwd %.3f" % d.get("_wall_fwd_bwd", -1))
            if os.environ.get("PN_TUNE_ALL"):
                print("    " + " ".join("%s %.3f" % kv for kv in d.items()))
        else:
            print("%-60s FAILED %s" % (spec, r.stderr[-300:]))


if __name__ == "__main__":
    main()

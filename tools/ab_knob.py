"""A/B of a context knob INSIDE one process (GPU box): the bench's full training step (bench.StepRunner: sampler + forward +
loss + backward + Adam) on a workload, blocks of steps alternating between the knob's values so that box-to-box and
minute-to-minute drift cancel.  Prints the blocks' wall ms/step per value and their medians.

    python tools/ab_knob.py PN_POOL_STEP 0 1 [workload=cora|pubmed|bgp] [blocks=6] [steps=30] [fused=0|1]
    python tools/ab_knob.py FUSED 0 1            (pseudo-knob: three library calls vs pn_pagg_train_step)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pathnet_amd import _lib  # noqa: E402


def main():
    name, values, opts = sys.argv[1], [], {"workload": "cora", "blocks": "6", "steps": "30", "fused": "0"}
    for a in sys.argv[2:]:
        if "=" in a:
            k, v = a.split("=", 1)
            opts[k] = v
        else:
            values.append(int(a))
    os.environ["PN_BENCH_FUSED"] = "0" if opts["fused"] == "0" else "1"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    wl = {"cora": lambda: bench.workload(0, 1), "pubmed": bench.pubmed_workload, "bgp": bench.bgp_workload}[opts["workload"]]()
    sr = bench.StepRunner(wl, dev, 0, 1, sharded=False)
    steps, blocks = int(opts["steps"]), int(opts["blocks"])
    for e in range(10):
        sr.step(e)
    torch.cuda.synchronize()
    res = {v: [] for v in values}
    e = 100
    for b in range(blocks):
        for v in values:
            if name == "FUSED":         # pseudo-knob: the step as three library calls (0) or through pn_pagg_train_step (1)
                sr.fused = bool(v)
            else:
                _lib.set_knob(name, v, dev)
            for _ in range(5):
                sr.step(e)
                e += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                sr.step(e)
                e += 1
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / steps * 1e3)
    out = {"knob": name, "workload": opts["workload"], "fused": opts["fused"], "steps_per_block": steps,
           "ms_per_step": {str(v): {"median": round(float(np.median(r)), 4), "min": round(min(r), 4), "max": round(max(r), 4),
                                    "blocks": [round(x, 4) for x in r]} for v, r in res.items()}}
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()

"""What ONE rank of an N-GPU weak-scaling run computes per step, measured on one GPU: the bench graph at world*2708 nodes,
the masked nodes of block 0 only.  The node-level stages that are replicated over all rows (bank, bank_bwd) show their
world-size cost; fc0 / fc0_bwd run over all rows here but only over the rank's 2708 rows in the sharded run (ignore them).
    python tools/emulate_rank_work.py [world=8] [steps=10]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pathnet_amd import _lib  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    lib = _lib.load()
    dev = torch.device("cuda")
    ctx = _lib.context(dev)
    names = bench.stage_names(lib)
    wl = bench.workload(0, world)
    wl["mask"][wl["n_loc"]:] = False
    sr = bench.StepRunner(wl, dev, 0, 1, sharded=False)
    m = bench.measure(sr, lib, ctx, names, steps, 3, torch.cuda.synchronize)
    print(json.dumps({"world_emulated": world, "nodes": wl["n"], "masked": sr.S, "ms_per_step": m["elapsed"] / steps * 1e3,
                      "stages_ms": m["stages_ms"]}))


if __name__ == "__main__":
    main()

"""Wall time of one full epoch of the reference recipe (train step + validation forward + test forward and
checkpoint on improvement, PathNet_run.py:315-394) with pathnet_amd.trainer on the bench workload."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pathnet_amd  # noqa: E402
from pathnet_amd import trainer  # noqa: E402

wl = bench.workload(0, 1)
rng = np.random.default_rng(5)
n = wl["n"]
perm = rng.permutation(n)
tr, va, te = np.zeros(n, bool), np.zeros(n, bool), np.zeros(n, bool)
tr[perm[:int(0.48 * n)]], va[perm[int(0.48 * n):int(0.8 * n)]], te[perm[int(0.8 * n):]] = True, True, True
smp = pathnet_amd.MerwSampler(*wl["graph"], wl["L"])
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
with tempfile.TemporaryDirectory() as d:
    trainer.train_fixed_indices(wl["X"], wl["Y"], wl["C"], "cora", tr, va, te, wl["W"], wl["H"], wl["L"], smp, epochs=5,
                                save_dir=d)
    torch.cuda.synchronize()
    t0 = time.time()
    res = trainer.train_fixed_indices(wl["X"], wl["Y"], wl["C"], "cora", tr, va, te, wl["W"], wl["H"], wl["L"], smp,
                                      epochs=epochs, save_dir=d)
    torch.cuda.synchronize()
    dt = time.time() - t0
print(json.dumps({"epochs": epochs, "ms_per_epoch": dt / epochs * 1e3, "train_nodes": int(tr.sum()),
                  "val_nodes": int(va.sum()), "test_nodes": int(te.sum()),
                  "paths_aggregated_per_epoch": int((tr.sum() + va.sum()) * wl["W"]), "result": res}))

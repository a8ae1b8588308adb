"""Phase times inside pool_step2_kernel (the fused pooling step of a node), from a library built with -DPN_POOL_TRACE
(pn_pagg.hip stamps wall_clock64 at eleven points of workgroup g's thread 0 into a device array, pn_debug_pool_trace copies it out):
    hipcc ... -DPN_POOL_TRACE -c pn_pagg.hip ; link as pathnet_amd/csrc/_variants/lib_trace.so
    PN_LIB_PATH=pathnet_amd/csrc/_variants/lib_trace.so python tools/trace_pool_step.py [workload]
Prints, over the launch's workgroups: the median / p90 duration of every phase (100 MHz ticks -> us), the median start offset of
a workgroup from the launch's first stamp, and the launch's span."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pathnet_amd import _lib  # noqa: E402

PHASES = ["loads issued -> rows arrived (+ same-row vote)", "attention loads + scores + coefficients", "pooled partials + barrier",
          "layer1 (Xh row, dropout) + barrier", "logits (fc2) + barrier", "cross entropy (one thread) + barrier",
          "d layer1 (fc2^T), atomics on d Xh + barrier", "d coef (wave sums)", "member loop: d h_n stores", "ego / attention partials + stores"]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cora"
    os.environ["PN_BENCH_FUSED"] = "1"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w = {"cora": lambda: bench.workload(0, 1), "pubmed": bench.pubmed_workload, "bgp": bench.bgp_workload}[wl]()
    sr = bench.StepRunner(w, dev, 0, 1, sharded=False)
    for e in range(8):
        sr.step(e)
    torch.cuda.synchronize()
    lib = _lib.load()
    out = np.zeros(2048 * 12, dtype=np.int64)
    assert lib.pn_debug_pool_trace(out.ctypes.data_as(ctypes.c_void_p)) == 0
    t = out.reshape(2048, 12)[:, :11]
    n = int((t[:, 0] > 0).sum())
    t = t[:n].astype(np.float64) / 100.0
    t0 = t[:, 0].min()
    print(f"{wl}: {n} workgroups traced; launch span {t[:, 10].max() - t0:.1f} us; a workgroup's own time median {np.median(t[:, 10] - t[:, 0]):.1f} us, "
          f"p90 {np.percentile(t[:, 10] - t[:, 0], 90):.1f}; start offset median {np.median(t[:, 0] - t0):.1f}, max {(t[:, 0] - t0).max():.1f} us")
    for k, name in enumerate(PHASES):
        d = t[:, k + 1] - t[:, k]
        print(f"  {k:2d} {name:58s} median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f} us")


if __name__ == "__main__":
    main()

"""How host-bound is the bench step?  (GPU box)  python tools/host_overhead.py [steps]
 (a) host time to ENQUEUE a step (no synchronisation) vs wall time per step,
 (b) the same step captured into a hipGraph (torch.cuda.CUDAGraph) and replayed: what the launches cost without Python."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sr = bench.StepRunner(bench.workload(0, 1), dev, 0, 1, sharded=False)
    for e in range(5):
        sr.step(e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in range(steps):
        sr.step(100 + e)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    res = {"steps": steps, "host_enqueue_ms_per_step": t_enq / steps * 1e3, "wall_ms_per_step": t_all / steps * 1e3}
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for e in range(3):
                sr.step(200 + e)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            sr.step(300)
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        res["graph_replay_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
    except Exception as e:      # noqa: BLE001
        res["graph_error"] = repr(e)[:500]
    print(json.dumps(res))


if __name__ == "__main__":
    main()

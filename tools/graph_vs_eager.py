"""Why is the captured step not faster than the eager one?  

    rocprofv3 --kernel-trace -d <dir>/eager -o t -- python tools/graph_vs_eager.py run eager
    rocprofv3 --kernel-trace -d <dir>/graph -o t -- python tools/graph_vs_eager.py run graph
    python tools/graph_vs_eager.py analyse <dir>/eager <dir>/graph  > profiles/r05_graph_vs_eager.txt

(`run det`: the eager step with the deterministic backward -- where its extra time goes, kernel by kernel.)
`run` drives the headline training step (bench.StepRunner with its per-step values in device memory, so that both modes
launch the very same kernels) 40 times, eagerly or as one captured hipGraph replayed; the first launch of every step is
step_state_advance_kernel, which is how `analyse` cuts the kernel trace into steps.  Per step it reports: kernels, the span
from the first kernel's start to the last one's end, the time at least one kernel was running (union of the intervals), the
idle gaps inside the span (span - union), the time two kernels ran at once (sum - union: the second stream at work), and the
largest gaps by the kernel that precedes them.  Wall time per step comes from the run itself (events around the 40 steps)."""
import glob
import json
import os
import sqlite3
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEPS = 40


def run(mode):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    wl = bench.workload(0, 1)
    # "headline": the step exactly as bench.py times it (host-side seeds, the sampler walk on the stream that reads the paths);
    # the other modes keep the step state in device memory, whose advance kernel marks the step boundaries
    sr = bench.StepRunner(wl, dev, 0, 1, sharded=False, device_state=(mode != "headline"))
    if mode == "det":           # the eager step with the fixed-order backward (pn_pagg_shape.deterministic)
        sr.model.deterministic = True
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for e in range(5):
            sr.step(e)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = None
    if mode == "graph":
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            sr.step(0)
        for _ in range(3):
            g.replay()
    else:
        for e in range(3):
            sr.step(10 + e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in range(STEPS):
        if g is not None:
            g.replay()
        else:
            sr.step(100 + e)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS * 1e3
    print("RESULT " + json.dumps({"mode": mode, "wall_ms_per_step": dt, "steps": STEPS}))


def load(d):
    db = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    if not db:
        raise SystemExit("no rocpd database under %s" % d)
    con = sqlite3.connect(db[0])
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    extra = [c for c in ("stream_id", "queue_id") if c in cols]
    q = "select %s, start, end%s from kernels order by start" % (name_col, "".join(", " + c for c in extra))
    return [dict(name=r[0], start=r[1], end=r[2], where=tuple(r[3:])) for r in cur.execute(q)]


def cut_steps(rows):
    """the timed steps of a trace, minus their first few: a step begins at its step_state_advance_kernel, or -- the headline
    mode has none -- at the first kernel that STARTS after the previous step's adam_kernel has ended (its own sampler walk
    runs earlier, under the previous step: it is listed with the step it belongs to by start time, i.e. with the previous one)"""
    marks = [i for i, r in enumerate(rows) if "step_state_advance_kernel" in r["name"]]
    if len(marks) < 3:
        ends = [r["end"] for r in rows if "adam_kernel" in r["name"]]
        marks, k = [], 0
        for i, r in enumerate(rows):
            if k < len(ends) and r["start"] >= ends[k]:
                marks.append(i)
                k += 1
    return [rows[a:b] for a, b in zip(marks, marks[1:])][-(STEPS - 5):]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]


def analyse_one(rows):
    steps = cut_steps(rows)
    out = dict(steps=len(steps))
    if not steps:
        return out
    agg = dict(kernels=0, span=0.0, union=0.0, total=0.0, period=0.0)
    gaps = {}
    streams = set()
    for k, st in enumerate(steps):
        agg["kernels"] += len(st)
        s0, s1 = st[0]["start"], max(r["end"] for r in st)
        agg["span"] += (s1 - s0) / 1e3
        agg["total"] += sum(r["end"] - r["start"] for r in st) / 1e3
        cur_end, un = s0, 0
        for r in st:
            streams.add(r["where"])
            if r["start"] > cur_end:
                g = r["start"] - cur_end
                prev = max((q for q in st if q["end"] <= r["start"]), key=lambda q: q["end"], default=None)
                key = (short(prev["name"]) if prev else "?") + " -> " + short(r["name"])
                gaps.setdefault(key, []).append(g / 1e3)
                cur_end = r["start"]
            if r["end"] > cur_end:
                un += r["end"] - max(cur_end, r["start"])
                cur_end = r["end"]
        agg["union"] += un / 1e3
        if k + 1 < len(steps):
            agg["period"] += (steps[k + 1][0]["start"] - s0) / 1e3
    n = len(steps)
    out.update(kernels_per_step=agg["kernels"] / n, span_us=agg["span"] / n, busy_us=agg["union"] / n,
               idle_in_span_us=(agg["span"] - agg["union"]) / n, overlapped_us=(agg["total"] - agg["union"]) / n,
               period_us=agg["period"] / max(n - 1, 1), queues=len(streams))
    out["gaps"] = sorted(((sum(v) / n, len(v) / n, k) for k, v in gaps.items()), reverse=True)[:14]
    per = {}
    for st in steps:
        for r in st:
            k = short(r["name"])
            a = per.setdefault(k, [0.0, 0])
            a[0] += (r["end"] - r["start"]) / 1e3
            a[1] += 1
    out["kernels"] = sorted(((v[0] / n, v[1] / n, k) for k, v in per.items()), reverse=True)
    return out


def analyse(dirs):
    for d in dirs:
        a = analyse_one(load(d))
        print("== %s: %d steps analysed, %.1f kernels per step on %d queue(s)" % (d, a["steps"], a.get("kernels_per_step", 0),
                                                                                   a.get("queues", 0)))
        if not a["steps"]:
            continue
        print("   step period (start to start) %.1f us | span first start -> last end %.1f us | some kernel running %.1f us | "
              "idle inside the span %.1f us | two kernels at once %.1f us" % (a["period_us"], a["span_us"], a["busy_us"],
                                                                                a["idle_in_span_us"], a["overlapped_us"]))
        print("   largest idle gaps (us per step, occurrences per step, predecessor -> successor):")
        for us, cnt, k in a["gaps"]:
            print("     %7.2f  x%.1f  %s" % (us, cnt, k))
        print("   kernel time per step (us, launches per step):")
        for us, cnt, k in a["kernels"]:
            print("     %8.2f  x%.1f  %s" % (us, cnt, k))


def timeline(d):
    """one step (the one of median length) kernel by kernel: start offset, duration, queue"""
    rows = load(d)
    steps = cut_steps(rows)
    if not steps:
        raise SystemExit("no steps in %s" % d)
    steps.sort(key=lambda st: max(r["end"] for r in st) - st[0]["start"])
    st = steps[len(steps) // 2]
    s0 = st[0]["start"]
    queues = {}
    print("== %s: the step of median span, %d kernels" % (d, len(st)))
    print("   start us |  dur us | end us  | queue | kernel")
    for r in st:
        q = queues.setdefault(r["where"], len(queues))
        print("   %8.1f | %7.1f | %7.1f | %5d | %s" % ((r["start"] - s0) / 1e3, (r["end"] - r["start"]) / 1e3, (r["end"] - s0) / 1e3, q,
                                                    short(r["name"])))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "timeline":
        for d in sys.argv[2:]:
            timeline(d)
    elif len(sys.argv) >= 3 and sys.argv[1] == "run":
        run(sys.argv[2])
    elif len(sys.argv) >= 3 and sys.argv[1] == "analyse":
        analyse(sys.argv[2:])
    else:
        raise SystemExit(__doc__)

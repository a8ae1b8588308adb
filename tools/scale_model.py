"""Scaling model of the node-sharded step (pathnet_amd/dist.py) for 2 / 4 / 8 MI355X of one box, from SINGLE-GPU
measurements -- there is no multi-GPU box to measure on, so the arithmetic is written down where it can be checked.

    python tools/scale_model.py [bench.json] [--link-GBs 50] [--latency-us 25] [--md]

Inputs: a bench.py JSON line (stages_ms of the headline configs[1] step and of the configs[3] / configs[4] extras).
Model, per rank and step, R ranks:

  sharded work   every stage that runs over the rank's masked nodes (sampler, plan, recurrence forward / BPTT, weight
                 gradient, pooling, classifier gradient): proportional to the rank's paths.
  own rows       fc0 forward / backward: the rank's N / R rows of X.
  replicated     the distance bank Z = bank(Xh) and its backward over ALL N rows of the graph, on every rank
                 (sharding Z instead would all-gather L x the bytes).  With touched-row compaction
                 (pn_pagg_shape.compact_rows, this round) only the (node, code) rows the rank's paths touch are computed.
  collectives    all-gather of Xh and reduce-scatter of dXh: every rank sends / receives its N/R x H x 4-byte block
                 to / from each of the R - 1 peers over its own xGMI link to that peer (the links run concurrently, one
                 block per link): t = latency + block_bytes / link_rate.  All-reduce of the flat gradient buffer
                 (G bytes): 2 (R - 1) / R x G / (links x link_rate) + latency.  The hetero class adds the all-gather of
                 the batch's index arrays (5 bytes per path step).
  overlap        (round 4, dist.py) the all-gather of Xh runs on a communication stream from begin_step on: under the sampler
                 and the aggregator's preamble (plan_pack); the reduce-scatter of dXh + fc0's backward on the rank's rows
                 run under the weight gradients that follow d Xh in the library's backward -- the recurrent one on its
                 own stream and half of the bank backward stage (its weight-gradient GEMMs; the other half, d Xh itself,
                 precedes the event).  What a collective costs the step is max(0, its time - the stage time it hides
                 under); the gradient all-reduce stays exposed.  overlap=False charges every collective in full (round 3).
  zero fill      (round 4: its own stage) the backward clears d Z and d Xh every step -- rows of the graph, not paths: d Z
                 (fraction zero_dz of the stage) follows the bank's rows (all N x L, or the rows the rank's paths touch when
                 the bank is compact), d Xh follows the nodes (all N; the touched ones in the restricted replicated mode).
  replicated     dist.ReplicatedAggregator: every rank holds all of X; the only collective is the gradient all-reduce (plus
                 the hetero class's index arrays).  Round 4: the call is restricted to the rows of X the rank's paths touch
                 (touched_nodes: their expected fraction), so fc0 and its backward run over those rows only -- before,
                 every rank projected all N rows each step, the term that capped configs[4] at 5.4 x on 8 ranks.

  sparse         (round 5, dist.ShardedAggregator(exchange="sparse")) a rank asks the owners for the rows its paths name instead of
                 gathering all of Xh: per peer link frac(R) x N / R rows of H x 4 bytes each way (frac = the fraction of the
                 graph's nodes a rank's paths touch) plus 4 bytes per row for the ids; the aggregator runs on the compact
                 table, so the zero fill of d Xh follows the touched nodes as well.  The exchange cannot start before the
                 step's paths exist: forward it hides under the index plan / weight packing only.

weak scaling (configs[1]: every rank owns a 2708-node block of an R x 2708-node graph, bench.py --gpus R):
  efficiency = t(1) / t(R).  strong scaling (configs[3], configs[4]: one graph): speed-up = t(1) / t(R).
"""
import argparse
import json
import os
import sys

SHARDED = ("sampler_walk", "sampler_fill", "plan_pack", "seq_fwd", "pool_fwd", "fc2_grad", "pool_bwd", "seq_bwd", "wgrad", "gather")
OWN_ROWS = ("fc0", "fc0_bwd")
REPLICATED = ("bank", "bank_bwd")


def split(stages):
    s = sum(v for k, v in stages.items() if k in SHARDED)
    o = sum(v for k, v in stages.items() if k in OWN_ROWS)
    r = sum(v for k, v in stages.items() if k in REPLICATED)
    rest = sum(v for k, v in stages.items() if k not in SHARDED + OWN_ROWS + REPLICATED + ("zero_fill",))
    return s, o, r, rest


DEFAULT_LINK_GBS, DEFAULT_LATENCY_US = 50.0, 25.0


def collectives_ms(R, n_total, H, grad_bytes, link_GBs, lat_us, idx_bytes=0, row_frac=1.0):
    """row_frac < 1: the sparse exchange -- that fraction of a block's rows travels per link (+ their 4-byte ids, once)"""
    if R == 1:
        return 0.0, {}
    block = n_total / R * H * 4 * row_frac
    ag = lat_us * 1e-3 + block / (link_GBs * 1e9) * 1e3
    rs = ag
    ar = lat_us * 1e-3 + 2 * (R - 1) / R * grad_bytes / (min(R - 1, 7) * link_GBs * 1e9) * 1e3
    ix = (lat_us * 1e-3 + idx_bytes / R / (link_GBs * 1e9) * 1e3) if idx_bytes else 0.0
    sx = 0.0
    if row_frac < 1.0:      # the row ids of the sparse exchange: one more small all-to-all
        sx = lat_us * 1e-3 + n_total / R * row_frac * 4 / (link_GBs * 1e9) * 1e3
    return ag + rs + ar + ix + sx, {"all_gather_Xh": ag, "reduce_scatter_dXh": rs, "all_reduce_grads": ar, "all_gather_indices": ix,
                                    "sparse_row_ids": sx}


BANK_MS_PER_NODE = 6.2e-6    # measured slope of bank + bank backward over the node count at L = 4, hid = 128: 0.058 ms at 2708
                             # nodes, 0.174 ms at 8 x 2708 (tools/emulate_rank_work.py): small GEMMs are
                             # latency-bound, the replicated bank does NOT cost R times the single-block time


def model(stages, total_ms, n_total_1, H, grad_bytes, weak, link_GBs, lat_us, idx_bytes=0, touched_frac=None, replicated=False,
          overlap=False, touched_nodes=None, zero_dz=0.8, sparse_nodes=None):
    """-> rows (R, t_ms, speedup_or_efficiency, parts).  stages: single-GPU per-stage ms; total_ms: single-GPU wall per
    step (the part not covered by the stage timers -- loss, Adam, torch glue -- is carried as `other`, per rank).
    overlap: collectives hide under the stages named in the module docstring.  touched_nodes(R): fraction of the graph's
    nodes a rank's paths name (replicated mode: fc0 over those rows only)."""
    s, o, r, rest = split(stages)
    z = stages.get("zero_fill", 0.0)
    other = max(total_ms - (s + o + r + rest + z), 0.0) + rest
    rows = []
    for R in (1, 2, 4, 8):
        if weak:        # per-rank paths and rows stay, the graph grows: the replicated bank grows with it
            n_total = n_total_1 * R
            sh, own, rep = s, o, r + BANK_MS_PER_NODE * n_total_1 * (R - 1)
            scale = 1.0
        else:           # one graph: paths and rows shrink, the replicated bank does not
            n_total = n_total_1
            sh, own, rep = s / R, o / R, r
            scale = 1.0 / R
        if touched_frac is not None:    # compaction: the bank runs over the rows this rank's paths touch
            rep = min(rep, rep * min(1.0, touched_frac(R)))
        sparse = sparse_nodes is not None and not replicated and R > 1
        coll, parts = collectives_ms(R, n_total, H, grad_bytes, link_GBs, lat_us, idx_bytes,
                                     row_frac=min(1.0, sparse_nodes(R)) if sparse else 1.0)
        if replicated:      # dist.ReplicatedAggregator: all of X on every rank, no exchange of Xh / d Xh
            own = o * (R if weak else 1)
            if touched_nodes is not None:
                own *= min(1.0, touched_nodes(R)) / min(1.0, touched_nodes(1))
            if R > 1:
                coll = parts["all_reduce_grads"] + parts["all_gather_indices"]
                parts = dict(parts, all_gather_Xh=0.0, reduce_scatter_dXh=0.0, sparse_row_ids=0.0)
        elif overlap and R > 1:
            hide_ag = (stages.get("sampler_walk", 0.0) + stages.get("sampler_fill", 0.0) + stages.get("plan_pack", 0.0)) * scale
            if sparse:      # the rows can only be asked for once the step's paths exist
                hide_ag = stages.get("plan_pack", 0.0) * scale
            hide_rs = max(stages.get("wgrad", 0.0) * scale, 0.5 * stages.get("bank_bwd", 0.0) * (rep / r if r > 0 else 1.0))
            own_bwd = stages.get("fc0_bwd", 0.0) * (scale if not weak else 1.0)      # fc0's backward rides on the communication stream
            ag = max(0.0, parts["all_gather_Xh"] - hide_ag)
            rs = max(0.0, parts["reduce_scatter_dXh"] + own_bwd - hide_rs)
            own = own - own_bwd
            # (round 6, dist.py _plan_sparse) the row ids of the sparse exchange travel on the communication stream under
            # fc0's projection of the rank's own rows
            sx = max(0.0, parts["sparse_row_ids"] - stages.get("fc0", 0.0) * (scale if not weak else 1.0))
            coll = ag + rs + parts["all_reduce_grads"] + parts["all_gather_indices"] + sx
            parts = dict(parts, all_gather_Xh=ag, reduce_scatter_dXh=rs, sparse_row_ids=sx)
        # zero fill of d Z (with the bank's rows) and d Xh (with the nodes)
        grow = R if weak else 1
        z_dz = z * zero_dz * grow * (min(1.0, touched_frac(R)) if touched_frac is not None else 1.0)
        z_dx = z * (1.0 - zero_dz) * grow
        if replicated and touched_nodes is not None:
            z_dx *= min(1.0, touched_nodes(R)) / min(1.0, touched_nodes(1))
        if sparse:          # d Xh is the compact table of touched rows
            z_dx *= min(1.0, sparse_nodes(R))
        t = sh + own + rep + other + z_dz + z_dx + coll
        rows.append((R, t, dict(sharded=sh, own_rows=own, replicated_bank=rep, other=other + z_dz + z_dx, collectives=coll, **parts)))
    t1 = rows[0][1]
    return [(R, t, (t1 / t) if weak else (t1 / t), p) for R, t, p in rows]


def newest_bench_json():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = ("r06_bench_final_extras.json", "r05_bench_final.json", "r04_bench_final.json", "r04_bench_f16_v1.json", "r03_bench_final.json")
    return next(p for p in (os.path.join(root, "profiles", n) for n in names) if os.path.exists(p))


def load_bench(path):
    b = json.load(open(path))
    b = b.get("bench", b) if "stages_ms" not in b else b
    if "stages_ms" not in b:            # the driver's record wraps the line
        b = json.loads([l for l in b.get("run", {}).get("stdout_tail", "").splitlines() if l.startswith("{")][-1])
    return b


def all_rows(b, link_GBs=DEFAULT_LINK_GBS, lat_us=DEFAULT_LATENCY_US):
    """-> [(key, title, what, rows)] for every configuration the bench line carries"""
    import math
    H = 128
    out = []
    # configs[1]: weak scaling of the Cora-shaped block (bench.py --gpus R)
    F, C, L = 1433, 7, 4
    grad1 = (F * H + H + L * (H * H + H) + 2 * (4 * H * H + 4 * H) + 2 * H + 1 + 2 * H * C + C) * 4
    out.append(("cora_weak", "configs[1] Cora-shaped block per rank (weak), collectives charged in full (no overlap)", "efficiency",
                model(b["stages_ms"], b["ms_per_step"], 2708, H, grad1, True, link_GBs, lat_us)))
    out.append(("cora_weak_overlap", "configs[1] Cora-shaped block per rank (weak), collectives overlapped (dist.py, round 4)", "efficiency",
                model(b["stages_ms"], b["ms_per_step"], 2708, H, grad1, True, link_GBs, lat_us, overlap=True)))
    # a rank's 51 960 paths x 4 steps land on the whole R x 2708-node graph (bench.workload draws ONE graph over all
    # nodes): expected distinct (node, code) rows = rows x (1 - exp(-steps / rows)) -- nearly all of them up to 8 ranks
    steps1 = 1299 * 40 * L
    out.append(("cora_weak_touched", "configs[1], bank over the rows a rank's paths touch (uniform estimate)", "efficiency",
                model(b["stages_ms"], b["ms_per_step"], 2708, H, grad1, True, link_GBs, lat_us,
                      touched_frac=lambda R: 1.0 - math.exp(-steps1 / (R * 2708.0 * L)))))
    if "bgp_scale_step" in b:
        g = b["bgp_scale_step"]
        F3, C3 = 287, 8
        grad3 = (F3 * H + H + L * (H * H + H) + 2 * (4 * H * H + 4 * H) + 2 * H + 1 + 2 * H * C3 + C3) * 4
        idx = 30708 * 40 * 4 * 5 + 30708 * 4
        out.append(("bgp", "configs[3] BGP-sized, hetero class (strong), no overlap", "speed-up",
                    model(g["stages_ms"], g["ms_per_step"], 63977, H, grad3, False, link_GBs, lat_us, idx_bytes=idx)))
        out.append(("bgp_overlap", "configs[3] BGP-sized, hetero class (strong), collectives overlapped", "speed-up",
                    model(g["stages_ms"], g["ms_per_step"], 63977, H, grad3, False, link_GBs, lat_us, idx_bytes=idx,
                          overlap=True)))
    if "configs4_one_gpu_step" in b and b["configs4_one_gpu_step"].get("stage_ms_per_step"):
        g = b["configs4_one_gpu_step"]
        st = g["stage_ms_per_step"]
        F4, C4, L4 = 128, 8, 6
        grad4 = (F4 * H + H + L4 * (H * H + H) + 2 * (4 * H * H + 4 * H) + 2 * H + 1 + 2 * H * C4 + C4) * 4
        tot = g["seconds_per_step"] * 1e3
        rows, steps = 60e6, 100_000 * 40 * 6
        ZDZ4 = 24e6 / (24e6 + 10e6)      # d Z holds min(rows, path steps) = 24 M rows of the zero fill, d Xh 10 M
        uniq = lambda R: rows * (1.0 - math.exp(-steps / (R * rows)))      # expected distinct (node, code) rows of a rank's steps
        nodes, nsteps = 10e6, 100_000 * (40 * 6 + 1)
        tn = lambda R: 1.0 - math.exp(-nsteps / (R * nodes))       # expected fraction of the nodes a rank's paths name
        if g.get("compact_rows", True):
            # the measured step already runs the bank over the rows its 24 M path steps touch (19.8 M of 60 M expected);
            # a rank's share of the paths touches uniq(R) of them
            kw = dict(touched_frac=lambda R: uniq(R) / uniq(1), zero_dz=ZDZ4)
            out.append(("configs4_sharded_dense", "configs[4] 10 M nodes, L = 6, 100 000 masked nodes (strong), node-sharded, dense "
                        "all-gather / reduce-scatter, bank over the touched rows (as measured)",
                        "speed-up", model(st, tot, 10_000_000, H, grad4, False, link_GBs, lat_us, **kw)))
            out.append(("configs4_sharded_dense_overlap", "configs[4] node-sharded, dense exchange, collectives overlapped (the exchange of "
                        "Xh / dXh does not fit under anything)",
                        "speed-up", model(st, tot, 10_000_000, H, grad4, False, link_GBs, lat_us, overlap=True, **kw)))
            out.append(("configs4_sharded_sparse", "configs[4] node-sharded, SPARSE exchange of the touched rows (round 5: %.0f %% of the "
                        "nodes per rank at 2 ranks, %.0f %% at 8), overlapped" % (100 * tn(2), 100 * tn(8)),
                        "speed-up", model(st, tot, 10_000_000, H, grad4, False, link_GBs, lat_us, overlap=True,
                                          sparse_nodes=tn, **kw)))
            out.append(("configs4_replicated", "configs[4], all of X on every rank (dist.ReplicatedAggregator): fc0 over all rows (round 3), "
                        "gradient all-reduce only",
                        "speed-up", model(st, tot, 10_000_000, H, grad4, False, link_GBs, lat_us, replicated=True, **kw)))
            out.append(("configs4_replicated_touched", "configs[4], all of X on every rank, the call restricted to the rows its paths touch "
                        "(round 4: fc0 over %.0f %% of the nodes on one rank, %.0f %% on each of 8)" % (100 * tn(1), 100 * tn(8)),
                        "speed-up", model(st, tot, 10_000_000, H, grad4, False, link_GBs, lat_us, replicated=True,
                                          touched_nodes=lambda R: tn(R) / 1.0, **kw)))
        else:
            out.append(("configs4_sharded_dense", "configs[4] 10 M nodes, L = 6, 100 000 masked nodes (strong), bank over all 60 M rows",
                        "speed-up", model(st, tot, 10_000_000, H, grad4, False, link_GBs, lat_us)))
    return out


def rows_for(b, key, link_GBs=DEFAULT_LINK_GBS, lat_us=DEFAULT_LATENCY_US):
    """the model's rows [(R, ms, speed-up or efficiency, parts)] of one configuration (bench.py prints the row of its own
    world size next to what it measured)"""
    b = b if "stages_ms" in b else load_bench_dict(b)
    for k, _, _, rows in all_rows(b, link_GBs, lat_us):
        if k == key:
            return rows
    raise KeyError(key)


def load_bench_dict(b):
    b = b.get("bench", b) if "stages_ms" not in b else b
    if "stages_ms" not in b:
        b = json.loads([l for l in b.get("run", {}).get("stdout_tail", "").splitlines() if l.startswith("{")][-1])
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bench", nargs="?", default=None)
    ap.add_argument("--link-GBs", type=float, default=DEFAULT_LINK_GBS, help="sustained rate of one xGMI link, one direction")
    ap.add_argument("--latency-us", type=float, default=DEFAULT_LATENCY_US, help="per collective")
    ap.add_argument("--md", action="store_true")
    a = ap.parse_args()
    path = a.bench or newest_bench_json()
    out = all_rows(load_bench(path), a.link_GBs, a.latency_us)
    for _, title, what, rows in out:
        print(("### " if a.md else "") + title + "   (xGMI link %.0f GB/s, %.0f us per collective; source %s)" % (
            a.link_GBs, a.latency_us, os.path.basename(path)))
        if a.md:
            print("| ranks | ms/step | %s | sharded | own rows (fc0) | bank (replicated) | other | collectives |" % what)
            print("|---|---|---|---|---|---|---|---|")
        for R, t, sp, p in rows:
            fmt = "| %d | %.3f | %.2f | %.3f | %.3f | %.3f | %.3f | %.3f |" if a.md else \
                "  R=%d  %9.3f ms  %s %.2f   sharded %.3f  fc0 %.3f  bank %.3f  other %.3f  collectives %.3f"
            args = (R, t, sp, p["sharded"], p["own_rows"], p["replicated_bank"], p["other"], p["collectives"]) if a.md else \
                (R, t, what, sp, p["sharded"], p["own_rows"], p["replicated_bank"], p["other"], p["collectives"])
            print(fmt % args)
        print()


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python
"""bench.py -- throughput of the path-aggregation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch, i.e. one training step of the reference loop
(/root/reference/PathNet_run.py:336-352) on the workload BASELINE.json's metric is quoted on
(configs[1]: Cora, path_num=40, path_len=4, hid=128):
    sample this epoch's W paths for every node on the GPU (MERW walker, Philox draws)
 -> select the paths of the masked (train) nodes
 -> PAGG forward (PathNet_homo) -> cross-entropy -> PAGG backward -> Adam step (lr 0.005, wd 5e-4)
Inputs are synthetic (no dataset ships with the reference mount) but Cora-shaped: N=2708 nodes,
F=1433 bag-of-words-like features, C=7, 48% of the nodes masked, MERW-like transition rows with every
row duplicated and self loops, as in the shipped edge_input files.
Everything is resident in HBM before the timed region.  value = paths aggregated per second
(S*W per step, whole job).  With N > 1 each rank owns an N-th of a graph that is N times larger
(weak scaling): node-sharded fc0, one all-gather of the projected feature matrix per step, one
reduce-scatter of its gradient and one flat all-reduce of the parameter gradients (RCCL).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
BF16_MFMA_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16 dense peak, same guide (fp32-input MFMA: 157.3)
# The recurrent GEMMs evaluate every fp32 product on the 16-bit matrix pipe (pn_kernels.h; fp16 and bf16 MFMAs run at the
# same rate): THREE fp16 MFMAs over scaled two-plane splits (pn_pagg_shape.seq_math = f16x2, the default since round 4) or
# SIX bf16 MFMAs over three-plane splits (bf16x3, rounds 1-3).  The ceiling for fp32-accurate flops is the pipe's dense
# peak divided by the MFMAs one product takes.
MFMAS_PER_PRODUCT = {"f16x2": 3, "bf16x3": 6}
F32_ON_BF16_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0


def seq_math_name():
    return os.environ.get("PN_SEQ_MATH", "").strip().lower() or "f16x2"


def synthetic_graph(n, seed, avg_und_deg=3.9):
    """Symmetric sparse graph + self loops, MERW-like probabilities (p ~ psi_v / psi_u-normalised from a few
    power iterations), every row written twice like init_rw.py:83-86 does."""
    rng = np.random.default_rng(seed)
    m_und = int(n * avg_und_deg / 2)
    a = rng.integers(0, n, m_und * 2)
    b = rng.integers(0, n, m_und * 2)
    keep = a != b
    und = np.unique(np.stack([np.minimum(a, b)[keep], np.maximum(a, b)[keep]], 1), axis=0)[:m_und]
    src = np.concatenate([und[:, 0], und[:, 1], np.arange(n)])
    dst = np.concatenate([und[:, 1], und[:, 0], np.arange(n)])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    psi = np.ones(n)
    for _ in range(20):
        nxt = np.zeros(n)
        np.add.at(nxt, src, psi[dst])
        psi = nxt / np.linalg.norm(nxt)
    w = psi[dst]
    tot = np.zeros(n)
    np.add.at(tot, src, w)
    p = w / tot[src]
    return n, np.repeat(src, 2).astype(np.int32), np.repeat(dst, 2).astype(np.int32), np.repeat(p, 2)


def workload(rank, world, seed=0):
    """Cora-shaped workload; with world > 1 every rank owns a 2708-node block of a world*2708-node graph."""
    n_loc, F, C, H, W, L = 2708, 1433, 7, 128, 40, 4
    n = n_loc * world
    g = synthetic_graph(n, seed)
    rng = np.random.default_rng(seed + 1)
    X = (rng.random((n, F)) < 0.0127).astype(np.float32)          # Cora's bag-of-words density
    X /= np.maximum(X.sum(1, keepdims=True), 1.0)                 # row-normalised like dataset.py's preprocess
    Y = rng.integers(0, C, n)
    perm = rng.permutation(n)
    mask = np.zeros(n, bool)
    mask[perm[: int(0.48 * n)]] = True
    return dict(n=n, n_loc=n_loc, F=F, C=C, H=H, W=W, L=L, graph=g, X=X, Y=Y, mask=mask)


def stage_names(lib):
    return [lib.pn_profile_stage_name(i).decode() for i in range(lib.pn_profile_stage_count())]


def read_profile(lib, names, ctx=None):
    from pathnet_amd import _lib
    ms = (ctypes.c_double * len(names))()
    cnt = (ctypes.c_int64 * len(names))()
    _lib.check(lib.pn_profile_read(ctx if ctx is not None else _lib.context("cuda"), ms, cnt))
    return {names[i]: (ms[i], cnt[i]) for i in range(len(names)) if cnt[i]}


def source_hash():
    """Hash of the library's sources: PMC traffic figures under profiles/ are only quoted for the build they were
    taken on (tools/pmc_passes.sh stamps them with the same hash)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pathnet_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def clock_mhz(lib, dev):
    from pathnet_amd import _lib
    v = ctypes.c_double(0.0)
    with torch.cuda.device(dev):
        _lib.check(lib.pn_clock_probe(ctypes.byref(v), _lib.stream_ptr(dev)))
    return round(v.value, 1)


def seq_flops(P, L, H, G=4):
    """Flops of one recurrent GEMM kernel over P paths.  SURVEY.md §8d charges L*16*H^2 per path ([x;h] (2H) x 4H gate
    columns per step); the math needs (2L-1)/(2L) of it: h_{-1} = 0, so the W_hh product of step 0 (seq_fwd), the
    dh_{-1} product (seq_bwd) and the W_hh gradient of the t = 0 rows (wgrad) do not exist."""
    survey = 2.0 * P * L * (2 * H) * (G * H)
    return survey * (2 * L - 1) / (2 * L), survey


L2_PEAK_GBS = 34500.0        # aggregate L2 -> CU, same guide ("L2 (per XCD)": ~34.5 TB/s)


def algorithmic_bytes(kernel, P, L, H, G=4):
    """HBM bytes one launch of a recurrent kernel has to move by SURVEY.md 8(d)'s per-path figures: the gather forward
    L*H*4 + L*5 (2068 B at L = 4, H = 128), the gather backward 2*L*H*4 (upstream gradient read + accumulated row written);
    8(d) has no figure for the weight-gradient GEMM -- a GEMM in a launch of its own reads its two operands once,
    L*(G*H + 2*H)*4 per path.  Everything else these kernels move (saved gates, cell states, [x|h] rows, gate gradients)
    is the DESIGN's traffic, which is what `traffic` (PMC) shows beside it."""
    per_path = {"seq_fwd": L * H * 4 + L * 5, "seq_bwd": 2 * L * H * 4, "wgrad": L * (G * H + 2 * H) * 4}[kernel]
    return float(P) * per_path


def roofline_block(dominant, dom_ms, launches, P, L, H, traffic, math=None, l2_requests=None):
    alg, survey = seq_flops(P, L, H)
    math = math or seq_math_name()
    if dominant in ("seq_fwd", "seq_bwd", "wgrad"):
        per = MFMAS_PER_PRODUCT[math]
        peak = BF16_MFMA_PEAK_TFLOPS / per
        achieved = alg / (dom_ms * 1e-3) / 1e12
        # ---- which resource the launch leans on hardest: matrix pipe by its algorithmic flops, HBM by the bytes the PMC
        #      passes counted for this very build, L2 -> CU by the counted requests (128 B each).  `bound` is the larger of
        #      the first two (the contract knows "mfma" and "hbm"); all three are printed.  Without a PMC stamp of this build
        #      only the matrix side is known and `bound` says so.
        alg_b = algorithmic_bytes(dominant, P, L, H)
        ev = {"mfma_frac": round(achieved / peak, 4),
              "hbm_frac": round(traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
              "l2_frac": round(l2_requests * 128.0 / (dom_ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4) if l2_requests else None,
              "note": "mfma_frac: algorithmic fp32 flops / (2.5 PFLOP/s / MFMAs per product); hbm_frac: PMC HBM bytes of the launch / "
                      "its duration / 8 TB/s; l2_frac: TCC_REQ x 128 B / duration / 34.5 TB/s.  None = no PMC pass of this build "
                      "(profiles/pmc_traffic.json carries another source hash)"}
        ev["largest"] = max((k for k in ("mfma_frac", "hbm_frac", "l2_frac") if ev[k] is not None), key=lambda k: ev[k])[:-5]
        hbm_bound = ev["hbm_frac"] is not None and ev["hbm_frac"] > ev["mfma_frac"]
        as_mfma = {"achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4)}
        hbm_ach = alg_b / (dom_ms * 1e-3) / 1e9
        as_hbm = {"achieved": round(hbm_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_ach / HBM_PEAK_GBS, 4),
                  "algorithmic_bytes_per_launch": alg_b,
                  "traffic_over_algorithmic": round(traffic / alg_b, 2) if traffic else None}
        top = as_hbm if hbm_bound else as_mfma
        blk = {"kernel": dominant, "bound": "hbm" if hbm_bound else "mfma", "achieved": top["achieved"],
               "peak": top["peak"], "unit": top["unit"],
               "frac": top["frac"], "traffic": traffic,
               "bound_evidence": ev, "as_mfma": as_mfma, "as_hbm": as_hbm,
               "avg_launch_ms": round(dom_ms, 4), "launches_timed": int(launches),
               "seq_math": math,
               "algorithmic_flops_per_launch": alg,
               "survey_8d_flops_per_launch": survey,
               "frac_with_survey_8d_flops": round(survey / (dom_ms * 1e-3) / 1e12 / peak, 4),
               "mfma_flops_issued_per_launch": per * alg,
               "ceilings_TFLOPs": {"f16_pipe_over_3": round(BF16_MFMA_PEAK_TFLOPS / 3, 1),
                                   "bf16_pipe_over_6": round(F32_ON_BF16_PEAK_TFLOPS, 1), "f32_input_mfma": 157.3},
               "frac_of_bf16_pipe_over_6": round(achieved / F32_ON_BF16_PEAK_TFLOPS, 4),
               "frac_of_f32_input_mfma_peak": round(achieved / 157.3, 4),
               "note": "fp32 results (1e-5 parity; measured ~3e-7 against float64) from the 16-bit matrix pipe: %s; peak = "
                       "2.5 PFLOP/s dense / %d.  achieved counts ALGORITHMIC fp32 flops = (2L-1)*8*H^2 per path, i.e. 7/8 "
                       "(L=4) of SURVEY.md 8d's L*16*H^2 (the step-0 products with h_{-1} = 0 are not charged); "
                       "frac_with_survey_8d_flops charges all of it.  With three MFMAs per product the kernel's saved-tensor "
                       "traffic is the co-bound: see hbm_side" %
                       ("fp32 = 2 scaled fp16 planes, 3 MFMAs per product, fp32 accumulate" if per == 3 else
                        "fp32 = 3 bf16 planes, 6 MFMAs per product, fp32 accumulate", per)}
        if traffic:
            tbs = traffic / (dom_ms * 1e-3) / 1e12
            blk["hbm_side"] = {"traffic_bytes_per_launch": traffic, "TB_per_s": round(tbs, 3),
                               "frac_of_hbm_peak": round(tbs * 1e3 / HBM_PEAK_GBS, 4),
                               "note": "PMC HBM bytes of the same launch / its duration: what the design's saved tensors (gates, "
                                       "cell states, [x|h] rows, gate gradients) cost, beside the matrix-pipe fraction above"}
        return blk
    return {"kernel": dominant, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": None, "traffic": traffic, "avg_launch_ms": round(dom_ms, 4)}


def pmc_traffic(kernel, key):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes committed under profiles/ -- only when they were
    taken on this very build (source hash) and workload (key); otherwise None: no stale figure is ever quoted."""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if tr.get("source_hash") == source_hash() and kernel in tr.get(key, {}):
            return tr[key][kernel], tr.get("source")
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def pmc_l2_requests(kernel, key):
    """TCC_REQ per launch from the same stamped passes (key: "l2_requests_per_launch" / "pubmed_l2_requests_per_launch")"""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if tr.get("source_hash") == source_hash():
            return tr.get(key, {}).get(kernel)
    except (OSError, ValueError, KeyError):
        pass
    return None


SLIM_LINE_LIMIT = 6144      # bytes: the driver keeps ~8 KB of stdout; the r05 line (20 KB) came back unparsed


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _num(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def slim_line(full):
    """The ONE stdout line: numbers only, every key the bench contract names, nothing else.  Everything bench.py measures
    beyond it (other configurations, stage tables, notes) is in `full`, which goes to bench_extras.json and to stderr."""
    out = _pick(full, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "seq_math", "step_api", "data")
    cfg = full.get("config", {})
    out["config"] = _pick(cfg, "workload", "nodes", "paths_per_step", "parallelism")
    r = full.get("roofline") or {}
    ro = _pick(r, "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_timed")
    ro["as_mfma"] = _pick(r.get("as_mfma", {}), "achieved", "peak", "frac")
    ro["as_hbm"] = _pick(r.get("as_hbm", {}), "achieved", "peak", "frac", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")
    if r.get("hbm_side"):
        ro["hbm_side"] = _pick(r["hbm_side"], "TB_per_s", "frac_of_hbm_peak")
    out["roofline"] = ro
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "cores_available", "kind", "detail", "ms_per_step")
        out["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
    sb = full.get("cpu_baseline_sampler")
    if sb:
        out["cpu_baseline_sampler"] = _pick(sb, "value", "unit", "cores", "kind")
    d = full.get("dispersion", {}).get("block_ms_per_step", {})
    out["dispersion"] = {"block_ms_per_step": _pick(d, "min", "median", "max")}
    if "stages_ms" in full:
        out["stages_ms"] = full["stages_ms"]
    if "sampler" in full:
        out["sampler"] = _pick(full["sampler"], "value", "unit")
    g, fg = full.get("gather_hbm_table"), full.get("fused_gather_hbm_table")
    if g or fg:
        out["gather"] = {"standalone_frac": _num((g or {}).get("read_frac_of_hbm_peak")),
                         "fused_frac": _num((fg or {}).get("gather_read_frac_of_hbm_peak")),
                         "table_MB": (g or fg).get("table_MB"), "pubmed_table": "cache-resident"}
    for k in ("pubmed_scale_step", "bgp_scale_step"):
        if k in full and "ms_per_step" in full[k]:
            out[k] = {"ms_per_step": _num(full[k]["ms_per_step"]), "value": _num(full[k]["value"], 1)}
    out["accuracy_cornell"] = "blocked: splits.zip absent from the reference mount"
    c = full.get("collectives")
    if c:
        ex = c.get("exposed_ms_per_step_by_rank") or []
        out["collectives"] = dict(_pick(c, "rccl_ranks_seen", "backend", "distinct_devices", "overlap", "bytes_per_step"),
                                  exposed_ms_per_step_max={k: max(e.get(k, 0.0) for e in ex) for k in (ex[0] if ex else {})})
    for k in ("bgp_strong", "configs4_replicated", "configs4_sharded"):     # the N > 1 blocks (configs[3] / [4]): numbers only
        if isinstance(full.get(k), dict):
            out[k] = _pick(full[k], "value", "unit", "ms_per_step", "scaling")
    out["library_source_hash"] = full.get("library_source_hash")
    out["extras"] = "bench_extras.json"

    def rnd(o):
        if isinstance(o, float):
            return float("%.6g" % o)
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, list):
            return [rnd(v) for v in o]
        return o
    out = rnd(out)
    line = json.dumps(out, allow_nan=False, separators=(", ", ": "))
    for drop in ("stages_ms", "collectives", "pubmed_scale_step", "bgp_scale_step", "sampler"):   # (never needed so far)
        if len(line) <= SLIM_LINE_LIMIT:
            break
        out.pop(drop, None)
        line = json.dumps(out, allow_nan=False)
    assert len(line) <= SLIM_LINE_LIMIT, len(line)
    return line


def emit(full):
    """extras file + stderr first, the slim stdout line LAST (so it is the last line of either stream)"""
    path = os.environ.get("PN_BENCH_EXTRAS", os.path.join(ROOT, "bench_extras.json"))
    text = json.dumps(full)
    try:
        with open(path, "w") as f:
            f.write(text + "\n")
    except OSError as e:
        sys.stderr.write("bench_extras.json not written: %r\n" % (e,))
    sys.stderr.write(text + "\n")
    sys.stderr.flush()
    sys.stdout.write(slim_line(full) + "\n")
    sys.stdout.flush()


def cpu_baseline(wl, seconds_budget=20.0):
    """The oracle's port of the reference PAGG step (oracle/pagg_oracle.py: same torch CPU arithmetic as
    the reference classes) timed on this host: forward + CE + backward + Adam on a bounded node sample."""
    from oracle import pagg_oracle as po
    from oracle import merw
    n, F, C, H, W, L = wl["n"], wl["F"], wl["C"], wl["H"], wl["W"], wl["L"]
    torch.manual_seed(0)
    sel_all = np.flatnonzero(wl["mask"])
    S = len(sel_all)            # the same batch the GPU step aggregates
    sel = sel_all[:S]
    gn, u, v, p = wl["graph"]
    ids, codes = merw.sample_full(gn, u, v, p, W, L, merw.DRAW_PHILOX, 1, epoch_count=1)
    ids, codes = ids[0][sel], codes[0][sel]
    lin = torch.nn.Linear
    mods = {"fc0": lin(F, H), "fc2": lin(2 * H, C), "attw": lin(2 * H, 1), "LSTM": torch.nn.LSTM(H, H)}
    params = {}
    for k, m in mods.items():
        for pn_, t in m.named_parameters():
            params["%s.%s" % (k, pn_)] = t.detach().clone().requires_grad_(True)
    for d in range(L):
        m = lin(H, H)
        params["nets.%d.weight" % d] = m.weight.detach().clone().requires_grad_(True)
        params["nets.%d.bias" % d] = m.bias.detach().clone().requires_grad_(True)
    opt = torch.optim.Adam(list(params.values()), lr=0.005, weight_decay=0.0005)
    X = torch.from_numpy(wl["X"])
    Y = torch.from_numpy(wl["Y"][sel])
    lossf = torch.nn.CrossEntropyLoss()
    times = []
    t_begin = time.time()
    for it in range(12):
        t0 = time.time()
        out = po.forward("homo", params, X, ids, codes, sel, W, L)
        loss = lossf(out, Y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.time() - t0)
        if it >= 3 and time.time() - t_begin > seconds_budget:
            break
    med = float(np.median(times[1:]))
    return {"value": S * W / med, "unit": "paths/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle/pagg_oracle.py (torch CPU, reference arithmetic) fwd+CE+bwd+Adam on %d of the %d "
                      "masked nodes (%d paths), all %d nodes projected by fc0 as in the reference; median of %d "
                      "steps" % (S, len(sel_all), S * W, n, len(times) - 1), "ms_per_step": med * 1e3}


class _ReferenceOpsHomo(torch.nn.Module):
    """The reference's op sequence for the homophilous class (/root/reference/PathNet_run.py:220-278) written with stock
    torch modules -- what BASELINE.md section 2 asks to be timed on the host ("the reference classes, CPU PyTorch"):
    fc0 over all N nodes + ReLU, row gather, ALL L distance layers on every gathered row stacked and one selected per
    row by its code, ReLU, dropout, nn.LSTM over the L steps, attention against the ego row, mean over the paths,
    concat with the node's own row, dropout, fc2.  (The reference classes themselves cannot travel to the GPU box;
    this module is measurement infrastructure, used by nothing else.)"""

    def __init__(self, F, H, C, L, p_drop):
        super().__init__()
        nn = torch.nn
        self.fc0, self.fc2, self.attw = nn.Linear(F, H), nn.Linear(2 * H, C), nn.Linear(2 * H, 1)
        self.nets = nn.ModuleList([nn.Linear(H, H) for _ in range(L)])
        self.LSTM = nn.LSTM(H, H)
        self.H, self.p = H, p_drop

    def forward(self, X, neis, W, L, sel, codes):
        Fn = torch.nn.functional
        S, H = sel.numel(), self.H
        Xh = torch.relu(self.fc0(X))
        nei = Xh[neis.reshape(-1)]                                             # [S*W*L, H]
        nei = torch.stack([layer(nei) for layer in self.nets], dim=1)           # [S*W*L, L, H]
        nei = torch.relu(nei[torch.arange(S * W * L), codes.reshape(-1)].view(S * W, L, H))
        ego_full = nei.reshape(S, W, L, H)[:, :, 0, :]
        seq = Fn.dropout(nei.transpose(0, 1), p=self.p, training=self.training)
        _, (h_n, _) = self.LSTM(seq)
        h_n = h_n.transpose(0, 1).reshape(S, W, H)
        h_n = ((1 + self.attw(torch.cat((h_n, ego_full), dim=-1))) * h_n).mean(dim=1)
        layer1 = Fn.dropout(torch.cat((Xh[sel], h_n), dim=1), p=self.p, training=self.training)
        return self.fc2(layer1)


def cpu_baseline_reference_ops(X, Y, ids, codes, sel, F, H, C, W, L, threads=(16, 32, 64, 128), seconds_budget=25.0,
                               what=""):
    """fwd + CE + bwd + torch.optim.Adam of _ReferenceOpsHomo on the host, best of a thread sweep (median of the steps
    after a warm-up for every thread count); the sweep stops when the budget is used up."""
    torch.manual_seed(0)
    model = _ReferenceOpsHomo(F, H, C, L, 0.7).train()
    opt = torch.optim.Adam(model.parameters(), lr=0.005, weight_decay=0.0005)
    lossf = torch.nn.CrossEntropyLoss()
    Xt, Yt = torch.as_tensor(X), torch.as_tensor(Y).long()
    neis, cd, st = torch.as_tensor(ids).long(), torch.as_tensor(codes).long(), torch.as_tensor(sel).long()
    S = st.numel()
    old = torch.get_num_threads()
    t_begin, sweep = time.time(), {}
    try:
        for k in [t for t in threads if t <= (os.cpu_count() or 1)] or [old]:
            torch.set_num_threads(k)
            times = []
            for it in range(4):
                t0 = time.time()
                loss = lossf(model(Xt, neis, W, L, st, cd), Yt)
                opt.zero_grad()
                loss.backward()
                opt.step()
                times.append(time.time() - t0)
            sweep[k] = float(np.median(times[1:]))
            if time.time() - t_begin > seconds_budget:
                break
    finally:
        torch.set_num_threads(old)
    best = min(sweep, key=sweep.get)
    return {"value": S * W / sweep[best], "unit": "paths/s", "cores": best, "cores_available": os.cpu_count(),
            "kind": "port", "detail": "reference-ops: best of a thread sweep, `cores` = the winning thread count",
            "ms_per_step": sweep[best] * 1e3, "thread_sweep_ms": {str(k): round(v * 1e3, 1) for k, v in sweep.items()},
            "sample": "the reference's op sequence (PathNet_run.py:239-278, :345-352) with stock torch modules -- nn.Linear "
                      "fc0 over all nodes, L stacked nn.Linear + select by code, nn.LSTM, attention, fc2, CrossEntropyLoss, "
                      "torch.optim.Adam -- %s: fwd + loss + bwd + Adam step over %d masked nodes (%d paths), median of 3 "
                      "steps after a warm-up per thread count, best thread count quoted" % (what, S, S * W)}


def cpu_baseline_cornell():
    """configs[0] (Cornell, path_num 40, path_len 4, hid 128: the reference's own CPU-runnable case): the unmodified
    sampler on the shipped cornell edge list (carried by tests/golden/sampler_cornell_40_4.npz) and the reference-ops
    aggregator step at Cornell's shape (N = 183, F = 1703, C = 5, 87 train nodes; features synthetic: splits.zip is
    absent from the reference mount)."""
    from oracle import merw
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "sampler_cornell_40_4.npz")))
    n, u, v, p = int(g["n"]), g["u"], g["v"], g["p"]
    res = {}
    if merw.have_ref():
        path = "/tmp/pn_bench_cornell_%d.in" % os.getpid()
        merw.write_edge_file(path, n, u, v, p)
        t0 = time.time()
        merw.run_ref(path, 40, 4, 1, to_devnull=True, timeout=600)
        dt = time.time() - t0
        os.remove(path)
        res["sampler"] = {"value": 1000 * n * 40 / dt, "unit": "sampled paths/s", "cores": 1, "kind": "reference",
                          "sample": "oracle/_ref/gen_merw (unmodified gen_merw.cpp) on the shipped cornell edge list, "
                                    "W=40 L=4, its fixed 1000 epochs = %d paths, output to /dev/null, wall %.1f s"
                                    % (1000 * n * 40, dt)}
    rng = np.random.default_rng(0)
    F, C, H, W, L, S = 1703, 5, 128, 40, 4, 87
    X = (rng.random((n, F)) < 0.05).astype(np.float32)
    sel = np.sort(rng.permutation(n)[:S])
    ids, codes = merw.sample_full(n, u, v, p, W, L, merw.DRAW_PHILOX, 1, epoch_count=1)
    res["pagg"] = cpu_baseline_reference_ops(X, rng.integers(0, C, S), ids[0][sel], codes[0][sel], sel, F, H, C, W, L,
                                             threads=(1, 4, 16, 64), seconds_budget=8.0, what="Cornell shape")
    return res


def cpu_baseline_sampler(seconds_budget=20.0):
    """The unmodified reference sampler (oracle/_ref/gen_merw, compiled from gen_merw.cpp) on a graph sized
    so that its fixed 1000 epochs take ~10 s; single-threaded like the reference; output -> /dev/null."""
    from oracle import merw
    if not merw.have_ref():
        return None
    n = 280
    g = synthetic_graph(n, 7)
    path = "/tmp/pn_bench_edges_%d.in" % os.getpid()
    merw.write_edge_file(path, *g)
    t0 = time.time()
    merw.run_ref(path, 40, 4, 1, to_devnull=True, timeout=600)
    dt = time.time() - t0
    os.remove(path)
    return {"value": 1000 * n * 40 / dt, "unit": "sampled paths/s", "cores": 1, "kind": "reference",
            "sample": "oracle/_ref/gen_merw (unmodified gen_merw.cpp, -O2 -mcmodel=medium) on a %d-node synthetic graph, "
                      "W=40 L=4, its fixed 1000 epochs = %d paths, output to /dev/null, whole-process wall %.1f s"
                      % (n, 1000 * n * 40, dt)}


def pubmed_workload(seed=3):
    """BASELINE.json configs[2] shapes (Pubmed N=19717, F=500, C=3, 9464 masked nodes, W=40, L=4, hid=128);
    pubmed.in is absent from the reference mount, so graph, features and labels are synthetic of that size."""
    n, F, C, H, W, L = 19717, 500, 3, 128, 40, 4
    g = synthetic_graph(n, seed, avg_und_deg=4.5)             # Pubmed: 44 324 undirected edges
    rng = np.random.default_rng(seed + 1)
    X = (rng.random((n, F)) < 0.1).astype(np.float32) * rng.random((n, F)).astype(np.float32)    # TF-IDF-like
    X /= np.maximum(X.sum(1, keepdims=True), 1e-6)
    Y = rng.integers(0, C, n)
    mask = np.zeros(n, bool)
    mask[rng.permutation(n)[:9464]] = True
    return dict(n=n, n_loc=n, F=F, C=C, H=H, W=W, L=L, graph=g, X=X, Y=Y, mask=mask)


def bgp_workload(world=1, seed=5):
    """BASELINE.json configs[3] (BGP, other_data: bgp.in and other_data.zip are absent from the reference mount): a
    synthetic stand-in of the published size of that dataset -- 63 977 nodes, 287 features, 8 classes -- with the class
    the reference uses for it, PathNet (hetero, PathNet_run.py:286-291).  The node count is what it is: the sharded runner's
    node blocks are ceil(n / world) rows, the last one shorter (dist.node_block)."""
    n0, F, C, H, W, L = 63977, 287, 8, 128, 40, 4
    n = n0
    g = synthetic_graph(n, seed, avg_und_deg=3.2)
    rng = np.random.default_rng(seed + 1)
    X = rng.random((n, F)).astype(np.float32)
    Y = rng.integers(0, C, n)
    mask = np.zeros(n, bool)
    mask[rng.permutation(n0)[: int(0.48 * n0)]] = True
    return dict(n=n, n_loc=-(-n // world), F=F, C=C, H=H, W=W, L=L, graph=g, X=X, Y=Y, mask=mask, cls="PathNet")


class StepRunner:
    """One training step of the reference loop on a workload, everything resident on the GPU."""

    def __init__(self, wl, dev, rank, world, sharded, timing_comm=False, hops="auto", device_state=False, replicated=False,
                 exchange="auto"):
        """device_state: epoch, dropout seed and Adam's step count live in device memory (pathnet_amd.StepState), so that
        a step captured into a hipGraph replays as the next step (measure_graph)."""
        import pathnet_amd
        self.wl, self.dev, self.world = wl, dev, world
        self.state = pathnet_amd.StepState(dev, seed=1234, first_epoch=0) if device_state else None
        n, F, C, H, W, L = wl["n"], wl["F"], wl["C"], wl["H"], wl["W"], wl["L"]
        gn, u, v, p = wl["graph"]
        key = ("_sampler", hops, str(dev))
        self.smp = wl.get(key)
        if self.smp is None:        # (the tables of a 10 M-node graph take ~10 s of host time: built once per workload and device)
            self.smp = wl[key] = pathnet_amd.MerwSampler(gn, u, v, p, L, device=dev, hops=hops)
        torch.manual_seed(0)
        self.model = getattr(pathnet_amd, wl.get("cls", "PathNet_homo"))(F, H, C, L, dropout=0.7).to(dev)
        self.model.step_state = self.state
        self.opt = pathnet_amd.Adam(self.model.parameters(), lr=0.005, weight_decay=0.0005,   # torch.optim.Adam's update, one launch
                                    **({"step_state": self.state} if self.state is not None else {}))
        self.lossf = pathnet_amd.CrossEntropyLoss()                                          # torch.nn.CrossEntropyLoss(), one launch
        # one GPU: forward, loss and backward of the step in ONE library call (pn_pagg_train_step: module.forward_loss(fused=True)),
        # since round 6 the quicker form at every size (pooling forward + loss + pooling backward in one launch, the small
        # launches under the BPTT: profiles/r06_glue.txt; same kernels and values as the three calls, tests/test_gpu_fused_step.py);
        # PN_BENCH_FUSED=0 runs forward / loss / backward as the reference's loop does, three calls.  Several ranks: the
        # node-sharded runner (three calls with the collectives between them)
        self.fused = os.environ.get("PN_BENCH_FUSED", "1") not in ("", "0")
        self.sample_beside = os.environ.get("PN_BENCH_SAMPLE_BESIDE", "1") not in ("", "0")      # (module.paths_stream)
        Y = torch.from_numpy(wl["Y"]).to(dev)
        self.runner = None
        self.overlap = os.environ.get("PN_BENCH_OVERLAP", "1") not in ("", "0")    # collectives on their own stream (dist.py)
        self.seed_backward = os.environ.get("PN_BENCH_PLAIN_BACKWARD", "0") in ("", "0")
        to_dev = lambda a: a.to(dev) if torch.is_tensor(a) else torch.from_numpy(a).to(dev)     # noqa: E731
        if not sharded:
            self.X = to_dev(wl["X"])
            self.sel = torch.from_numpy(np.flatnonzero(wl["mask"]).astype(np.int64)).to(dev)
            self.sel32 = self.sel.to(torch.int32)
            self.node_begin, self.node_count = 0, n
            self.loss_scale = 1.0
        else:
            from pathnet_amd import dist as pdist
            self.node_begin, n_loc = pdist.node_block(n, world, rank)       # ceil(n / world) rows, the last block shorter
            self.node_count = n_loc
            loc_mask = wl["mask"][self.node_begin:self.node_begin + n_loc]
            self.sel = torch.from_numpy(np.flatnonzero(loc_mask).astype(np.int64)).to(dev)        # local row ids
            self.sel32 = (self.sel + self.node_begin).to(torch.int32)                             # global node ids
            self.comm = pdist.Comm(timing=timing_comm)
            self.comm.overlap = self.overlap
            if replicated:      # all of X on every rank, each rank aggregates the masked nodes of its block (dist.py)
                self.X = to_dev(wl["X"])
                self.runner = pdist.ReplicatedAggregator(self.model, n_total=n, comm=self.comm)
            else:
                self.X = to_dev(wl["X"][self.node_begin:self.node_begin + n_loc]).contiguous()        # this rank's rows only
                self.runner = pdist.ShardedAggregator(self.model, n_total=n, row_begin=self.node_begin, row_count=n_loc,
                                                      comm=self.comm, exchange=exchange)
            # the mask is fixed: every rank knows every rank's count (no per-step exchange of counts)
            counts = [int(wl["mask"][b:b + c].sum()) for b, c in (pdist.node_block(n, world, r) for r in range(world))]
            self.runner.set_batch_counts(counts)
            self.loss_scale = world * counts[rank] / max(sum(counts), 1)      # mean over the WHOLE batch (dist.py)
        self.S = int(self.sel.numel())
        self.Ysel = Y[self.sel + (self.node_begin if sharded else 0)]
        self.sharded = sharded
        # the step samples the paths of its masked nodes only (PathNet_run.py:345 picks them out of the epoch's file)
        self.ids_buf = torch.empty((1, self.S, W, L), dtype=torch.int32, device=dev)
        self.codes_buf = torch.empty((1, self.S, W, L), dtype=torch.uint8, device=dev)

    def step(self, epoch):
        import pathnet_amd
        wl = self.wl
        W, L = wl["W"], wl["L"]
        if self.runner is not None and self.overlap and getattr(self.runner, "exchange", None) != "sparse":
            self.runner.begin_step(self.X)      # fc0 of the rank's rows + the all-gather of Xh start now, under the sampler
        if self.state is not None:
            self.state.advance()        # (one tiny launch: epoch + 1, Adam step + 1, the step's dropout seed)
            self.smp.sample(W, 0, nodes=self.sel32, draw_source=pathnet_amd.DRAW_PHILOX, check=False,
                            out=(self.ids_buf, self.codes_buf), step_state=self.state)
        elif self.runner is None and self.sample_beside:
            # this epoch's walk on the stream where the aggregator reads the paths (the library's second stream: behind the
            # previous step's recurrent weight gradient, in front of this step's index plan -- no event, nothing on the main stream)
            with torch.cuda.stream(self.model.paths_stream()):
                self.smp.sample(W, 1234, epoch_begin=epoch, epoch_count=1, nodes=self.sel32,
                                draw_source=pathnet_amd.DRAW_PHILOX, check=False, out=(self.ids_buf, self.codes_buf))
        else:
            self.smp.sample(W, 1234, epoch_begin=epoch, epoch_count=1, nodes=self.sel32,
                            draw_source=pathnet_amd.DRAW_PHILOX, check=False, out=(self.ids_buf, self.codes_buf))
        ids, codes = self.ids_buf[0], self.codes_buf[0]
        self.model.train()
        if self.runner is None and self.fused:
            # forward, CrossEntropyLoss and backward in one library call (pn_pagg_train_step): same kernels, same values
            loss, _ = self.model.forward_loss(self.X, ids, W, L, self.sel32, codes, self.Ysel, fused=True)
        else:
            if self.runner is None:
                out = self.model(self.X, ids, W, L, self.sel32, codes, None)
            else:
                out = self.runner(self.X, ids, W, L, self.sel32, codes)
            loss = self.lossf(out, self.Ysel)
            if self.loss_scale != 1.0:
                loss = loss * self.loss_scale
        self.opt.zero_grad(set_to_none=True)
        if self.seed_backward:
            pathnet_amd.backward(loss)      # loss.backward() without autograd's ones_like fill and the multiply by it (optim.py)
        else:
            loss.backward()
        if self.runner is not None:
            self.runner.allreduce_grads(average=True)
        self.opt.step()
        return loss


SETTLE_STEPS = 20
DOM_EVERY = 4       # the dominant kernel is bracketed by HIP events on every 4th step of the timed region


def measure(sr, lib, ctx, names, steps, warmup, barrier, repeats=0):
    """warm-up with every stage bracketed (finds the dominant kernel) -> exactly `steps` timed steps with only the
    dominant kernel carrying an event pair -> per-stage breakdown.  Returns a dict."""
    from pathnet_amd import _lib
    _lib.check(lib.pn_profile_configure(ctx, 1, -1))
    for e in range(max(1, warmup)):
        sr.step(e)
        if e == 0 and warmup > 1:       # the very first step loads code objects and grants LDS attributes: its stage
            torch.cuda.synchronize()    # times (a 5 ms plan_pack) must not pick the dominant kernel
            read_profile(lib, names, ctx)
    torch.cuda.synchronize()
    prof = read_profile(lib, names, ctx)
    kernel_stages = {k: v[0] / v[1] for k, v in prof.items()}
    dominant = max(kernel_stages, key=kernel_stages.get)
    _lib.check(lib.pn_profile_configure(ctx, 2, names.index(dominant)))
    # untimed steps in exactly the timed configuration: the warm-up above ran with every stage bracketed (serial, no second
    # stream) and is followed by synchronisations and a profile read-back -- the first steps after that run ~2 % slower than
    # the steady state every later block of the same length shows (dispersion.block_ms_per_step, profiles/r05_bench_final.json)
    for e in range(2):
        sr.step(480 + e)
    barrier()
    read_profile(lib, names, ctx)    # (drop the event pairs of the switch-over)
    # how stable the number is: an event at every step boundary of the timed region (a host-side record, nothing on the
    # GPU's critical path) gives the steps' own durations; after the region, `repeats` more regions of `steps` steps (each
    # between two device synchronisations, outside the timed one) say how much a K-step mean moves from block to block
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    # ... and only now the settling steps: between them and the timed region lies nothing but the barrier (a read-back or a
    # burst of event creations here lets the device idle for a millisecond, and the first steps after that run ~3 % slower)
    _lib.check(lib.pn_profile_configure(ctx, 0, -1))     # (no event pairs to read back in front of the region)
    # (one process: ~0.1 s of them, at least SETTLE_STEPS -- the count comes from the warm-up's stage times, not from a clock
    #  read here, which would need a synchronisation.  In profiles/r06 sessions the first regions after 20 steps still drifted:
    #  0.898 / 0.887 / 0.883 / 0.878 / 0.869 ms.  Several ranks: the fixed count, the same on every rank.)
    settle = SETTLE_STEPS
    if sr.world == 1:
        settle = int(min(200, max(SETTLE_STEPS, 100.0 / max(1e-3, sum(kernel_stages.values())))))
    for e in range(settle):
        sr.step(500 + e)
    barrier()
    # the dominant kernel's event pair on every DOM_EVERY-th step of the timed region (a host-side switch per step, nothing on
    # the GPU): an event between two kernels of the queue costs ~6 us of idle time there, two per bracketed launch
    dom_stage = names.index(dominant)
    t0 = time.perf_counter()
    for e in range(steps):
        _lib.check(lib.pn_profile_configure(ctx, 2 if e % DOM_EVERY == 0 else 0, dom_stage))
        sr.step(1000 + e)
    barrier()
    elapsed = time.perf_counter() - t0
    dom = read_profile(lib, names, ctx)[dominant]
    blocks = [elapsed / steps * 1e3]
    for r in range(max(0, repeats)):
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for e in range(steps):
            _lib.check(lib.pn_profile_configure(ctx, 2 if e % DOM_EVERY == 0 else 0, dom_stage))
            sr.step(2000 + r * steps + e)
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - tb) / steps * 1e3)
    read_profile(lib, names, ctx)
    _lib.check(lib.pn_profile_configure(ctx, 0, -1))
    # the steps' own durations: one more region of the same length with an event at every step boundary -- outside the timed
    # region, because an event between two kernels of the queue costs ~6 us of idle time there (profiles/r06_glue.txt: round
    # 5's timed region carried these marks and was the slowest of its five blocks every time)
    marks[0].record()
    for e in range(steps):
        sr.step(3000 + e)
        marks[e + 1].record()
    torch.cuda.synchronize()
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    sb = sorted(blocks)
    dispersion = {"step_ms": {"min": round(per_step[0], 4), "median": round(per_step[len(per_step) // 2], 4),
                              "max": round(per_step[-1], 4)},
                  "block_ms_per_step": {"min": round(sb[0], 4), "median": round(sb[len(sb) // 2], 4), "max": round(sb[-1], 4),
                                        "blocks": [round(b, 4) for b in blocks], "steps_per_block": steps},
                  "note": "step_ms: GPU time between consecutive step boundaries (events) of one more region run after the timed one; "
                          "block_ms_per_step: wall time per step of the timed region (first entry) and of %d more regions of the "
                          "same length run right after it -- a change smaller than max - min of these is not a result" % max(0, repeats)}
    _lib.check(lib.pn_profile_configure(ctx, 1, -1))
    for e in range(min(10, steps)):
        sr.step(5000 + e)
    torch.cuda.synchronize()
    prof = read_profile(lib, names, ctx)
    _lib.check(lib.pn_profile_configure(ctx, 0, -1))
    return {"elapsed": elapsed, "dominant": dominant, "dom_ms": dom[0] / dom[1], "dom_launches": dom[1], "settle_steps": settle,
            "stages_ms": {k: round(v[0] / v[1], 4) for k, v in prof.items()}, "dispersion": dispersion}


def measure_graph(wl, dev, steps, warmup, barrier):
    """The same training step captured once into a hipGraph (torch.cuda.CUDAGraph; the library's second stream joins the
    capture through its fork / join events) and replayed: exactly `steps` replays between two barriers.  Every replay
    is a NEW step -- epoch, dropout seed and Adam's step count are read from device memory (pn_step_state)."""
    sr = StepRunner(wl, dev, 0, 1, sharded=False, device_state=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for e in range(max(2, warmup)):
            sr.step(e)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sr.step(0)
    for _ in range(3):
        g.replay()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    barrier()
    elapsed = time.perf_counter() - t0
    st = sr.state.values()
    return {"ms_per_step": elapsed / steps * 1e3, "value": sr.S * wl["W"] * steps / elapsed, "unit": "paths/s", "steps": steps,
            "epochs_sampled": int(st["epoch"]) + 1, "adam_steps": int(st["adam_step"]),
            "note": "one captured step replayed: same kernels as the eager step (no per-kernel events, no Python between "
                    "the launches); the sampler's epoch, the dropout seed and Adam's step count advance in device memory"}


def time_launches(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e-3 / reps


def deterministic_line(sr, lib, ctx, names, steps, W):
    """the same step with module.deterministic = True: ms per step and the stages that change"""
    old = sr.model.deterministic
    sr.model.deterministic = True
    try:
        m = measure(sr, lib, ctx, names, steps, 3, torch.cuda.synchronize)
        return {"deterministic_ms_per_step": m["elapsed"] / steps * 1e3, "value": sr.S * W / (m["elapsed"] / steps),
                "unit": "paths/s", "steps": steps,
                "stages_ms": {k: v for k, v in m["stages_ms"].items() if k in ("pool_bwd", "seq_bwd", "wgrad", "bank_bwd", "fc0_bwd", "fc2_grad")},
                "note": "seq_bwd is bracketed twice in this mode (the BPTT, then the sorted segmented reduce of its rows): the "
                        "stage figure is the mean of the two brackets"}
    finally:
        sr.model.deterministic = old


def extras_single_gpu(lib, ctx, names, dev, sr, args):
    """Untimed extras of the N = 1 report: the other single-GPU configurations and the HBM-side microbenchmarks."""
    import pathnet_amd
    from pathnet_amd import _lib
    out = {}
    wl = sr.wl
    W, L, H = wl["W"], wl["L"], wl["H"]
    # ---- sampler alone, bench graph (7 MB hop table: cache resident -- a rate, not a roofline) ----------------------
    big_e = 16
    ids_big = torch.empty((big_e, wl["n"], W, L), dtype=torch.int32, device=dev)
    codes_big = torch.empty((big_e, wl["n"], W, L), dtype=torch.uint8, device=dev)
    cnt = [0]

    def smp_cora():
        cnt[0] += 1
        sr.smp.sample(W, 99, epoch_begin=cnt[0] * big_e, epoch_count=big_e, check=False, out=(ids_big, codes_big))
    dt = time_launches(smp_cora, 50)
    paths = big_e * wl["n"] * W
    out["sampler"] = {"value": paths / dt, "unit": "sampled paths/s", "draws": "philox", "paths_per_launch": paths,
                      "ms_per_launch": dt * 1e3, "hop_table": "dense %d^2 B (cache resident)" % wl["n"]}

    # the bit-exact mode: glibc's rand() stream regenerated on the device (srand(seed), two draws per step, the reference's
    # draw order -- gen_merw.cpp:81-91, :182-209); same graph, same launch size
    def smp_glibc():
        cnt[0] += 1
        sr.smp.sample(W, 99, epoch_begin=cnt[0] * big_e, epoch_count=big_e, check=False, out=(ids_big, codes_big),
                      draw_source=pathnet_amd.DRAW_GLIBC_REPLAY)
    dt_g = time_launches(smp_glibc, 20)
    out["sampler_glibc_replay"] = {"value": paths / dt_g, "unit": "sampled paths/s", "draws": "glibc rand() replay (bit-exact to "
                                   "the reference binary for the same seed)", "paths_per_launch": paths, "ms_per_launch": dt_g * 1e3,
                                   "vs_philox": round(dt / dt_g, 3)}
    del ids_big, codes_big

    # ---- forward only (metric (i): "report forward-only too"): the inference forward of the headline
    #      workload -- eval mode, nothing saved for a backward -- and the training forward (dropout, saved tensors) ----------
    ids_f, codes_f = sr.ids_buf[0], sr.codes_buf[0]

    def fwd_eval():
        with torch.no_grad():
            sr.model(sr.X, ids_f, W, L, sr.sel32, codes_f, None)

    def fwd_train():
        sr.model(sr.X, ids_f, W, L, sr.sel32, codes_f, None)
    sr.model.eval()
    dt_e = time_launches(fwd_eval, 50)
    sr.model.train()
    dt_t = time_launches(fwd_train, 50)
    out["forward_only"] = {"inference": {"value": sr.S * W / dt_e, "unit": "paths/s", "ms": dt_e * 1e3},
                           "training_forward": {"value": sr.S * W / dt_t, "unit": "paths/s", "ms": dt_t * 1e3},
                           "note": "PAGG forward alone on the headline workload (paths already sampled and resident): fc0, "
                                   "bank, recurrence, pooling, classifier; the training forward also draws the dropout masks "
                                   "and writes the tensors the backward reads"}

    # ---- the same headline step with the other arithmetic of the recurrent GEMMs (report both) -------------------
    other = "bf16x3" if seq_math_name() == "f16x2" else "f16x2"
    old_math = sr.model.seq_math
    sr.model.seq_math = other
    try:
        mo = measure(sr, lib, ctx, names, max(5, args.steps // 2), 3, torch.cuda.synchronize)
        steps_o = max(5, args.steps // 2)
        out["headline_step_" + other] = {
            "seq_math": other, "ms_per_step": mo["elapsed"] / steps_o * 1e3, "value": sr.S * W / (mo["elapsed"] / steps_o),
            "unit": "paths/s", "steps": steps_o, "stages_ms": mo["stages_ms"],
            "roofline": roofline_block(mo["dominant"], mo["dom_ms"], mo["dom_launches"], sr.S * W, L, H, None, math=other)}
    finally:
        sr.model.seq_math = old_math
    # ---- and with the fixed-order backward (pn_pagg_shape.deterministic: bitwise reproducible gradients, no fp32 atomics) ------
    out["headline_step_deterministic"] = deterministic_line(sr, lib, ctx, names, max(5, args.steps // 2), W)

    # ---- configs[2]: Pubmed-scale full training step (on-GPU MERW sampler + PAGG fwd/bwd + Adam) ---------------------
    pw = pubmed_workload()
    t0 = time.time()
    psr = StepRunner(pw, dev, 0, 1, sharded=False, hops="dense")
    setup_s = time.time() - t0
    m = measure(psr, lib, ctx, names, max(5, args.steps // 2), 3, torch.cuda.synchronize)
    steps_p = max(5, args.steps // 2)
    Pp = psr.S * W
    tr, tr_src = pmc_traffic(m["dominant"], "pubmed_hbm_bytes_per_launch")
    rb = roofline_block(m["dominant"], m["dom_ms"], m["dom_launches"], Pp, L, H, tr,
                        l2_requests=pmc_l2_requests(m["dominant"], "pubmed_l2_requests_per_launch"))
    if tr_src:
        rb["traffic_source"] = tr_src
    out["pubmed_scale_step"] = {
        "config": "configs[2] shapes, synthetic: N=%d F=%d C=%d hid=%d W=%d L=%d, %d masked nodes = %d paths/step, "
                  "PathNet_homo, dropout 0.7, Adam; dense hop table %d MB in HBM" %
                  (pw["n"], pw["F"], pw["C"], H, W, L, psr.S, Pp, pw["n"] ** 2 >> 20),
        "value": Pp / (m["elapsed"] / steps_p), "unit": "paths/s", "ms_per_step": m["elapsed"] / steps_p * 1e3,
        "steps": steps_p, "roofline": rb, "stages_ms": m["stages_ms"], "sampler_setup_s": round(setup_s, 2)}
    out["pubmed_scale_step"]["deterministic"] = deterministic_line(psr, lib, ctx, names, max(3, args.steps // 4), W)
    # sampler at Pubmed size: dense 389 MB table (beyond the Infinity Cache) and exact on-the-fly hop codes
    ids_p = torch.empty((4, pw["n"], W, L), dtype=torch.int32, device=dev)
    codes_p = torch.empty((4, pw["n"], W, L), dtype=torch.uint8, device=dev)

    def smp_pub(s):
        def f():
            cnt[0] += 1
            s.sample(W, 7, epoch_begin=cnt[0] * 4, epoch_count=4, check=False, out=(ids_p, codes_p))
        return f
    dt = time_launches(smp_pub(psr.smp), 20)
    paths = 4 * pw["n"] * W
    # per path L x (16 B triple + 1 B code) random reads + L x 5 B written; a random read costs a
    # 64 B sector at least
    out["sampler_pubmed_dense"] = {"value": paths / dt, "unit": "sampled paths/s", "ms_per_launch": dt * 1e3,
                                   "hop_table_MB": pw["n"] ** 2 >> 20,
                                   "algorithmic_GBs": paths * (L * 17 + L * 5) / dt / 1e9,
                                   "sector_GBs": paths * (L * 2 * 64 + L * 5) / dt / 1e9,
                                   "note": "random 16 B triple + 1 B hop-code reads: latency / cache bound (L2 + Infinity "
                                           "Cache serve most of the 64 B sectors), not an HBM streaming roofline"}
    gn, u, v, p = pw["graph"]
    otf = pathnet_amd.MerwSampler(gn, u, v, p, L, device=dev, hops="otf")
    dt = time_launches(smp_pub(otf), 10)
    out["sampler_pubmed_otf"] = {"value": paths / dt, "unit": "sampled paths/s", "ms_per_launch": dt * 1e3,
                                 "note": "exact hop codes from CSR lists (no n^2 table): what graphs beyond the "
                                         "reference's n = 100050 cap use"}
    del ids_p, codes_p, otf, psr

    # ---- configs[3] on one GPU: BGP-sized graph, the hetero class (the 8-GPU run shards exactly this step) ------------
    bw = bgp_workload()
    bsr = StepRunner(bw, dev, 0, 1, sharded=False)
    steps_b = max(3, args.steps // 5)
    mb = measure(bsr, lib, ctx, names, steps_b, 2, torch.cuda.synchronize)
    Pb = bsr.S * W
    out["bgp_scale_step"] = {
        "config": "configs[3] stand-in, synthetic: N=%d F=%d C=%d hid=%d W=%d L=%d, %d masked nodes = %d paths/step, "
                  "PathNet (hetero), dropout 0.7, Adam; dense hop table %d MB" %
                  (bw["n"], bw["F"], bw["C"], H, W, L, bsr.S, Pb, bw["n"] ** 2 >> 20),
        "value": Pb / (mb["elapsed"] / steps_b), "unit": "paths/s", "ms_per_step": mb["elapsed"] / steps_b * 1e3,
        "steps": steps_b, "roofline": roofline_block(mb["dominant"], mb["dom_ms"], mb["dom_launches"], Pb, L, H, None),
        "stages_ms": mb["stages_ms"]}
    del bsr

    # ---- hid = 512 on the bench graph: the step-by-step recurrence that serves hidden sizes beyond the fused kernels ---
    hw = dict(wl, H=512)
    hsr = StepRunner(hw, dev, 0, 1, sharded=False)
    steps_h = max(3, args.steps // 5)
    mh = measure(hsr, lib, ctx, names, steps_h, 2, torch.cuda.synchronize)
    Ph = hsr.S * W
    out["hid512_step"] = {
        "config": "bench graph, hid=512 (generic recurrence: one bf16x3 MFMA GEMM per step + cell kernels), %d paths/step" % Ph,
        "value": Ph / (mh["elapsed"] / steps_h), "unit": "paths/s", "ms_per_step": mh["elapsed"] / steps_h * 1e3,
        "steps": steps_h, "roofline": roofline_block(mh["dominant"], mh["dom_ms"], mh["dom_launches"], Ph, L, 512, None, math="bf16x3"),
        "stages_ms": mh["stages_ms"]}
    del hsr

    # ---- configs[4] on one GPU: 10 M-node Erdos-Renyi graph, path_len 6, exact on-the-fly sampler + a micro-batched step -
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs4
        torch.cuda.empty_cache()
        out["configs4_one_gpu_step"] = bench_configs4.run(10_000_000, 100_000, 2)
    except Exception as e:      # (an untimed extra must not take the headline line down with it)
        out["configs4_one_gpu_step"] = {"error": repr(e)[:300]}
    torch.cuda.empty_cache()

    # ---- the path-feature gather against HBM: a table that cannot sit in the 256 MB Infinity Cache -------------------
    Ng, Sg = 1 << 20, 9464                      # Z table [2^20, L, H] fp32 = 2 GB; Pubmed's path count
    table = torch.randn(Ng, L, H, device=dev)
    gi = torch.randint(0, Ng, (Sg, W, L), dtype=torch.int32, device=dev)
    gc = torch.randint(0, L, (Sg, W, L), dtype=torch.uint8, device=dev)
    rows = torch.empty((Sg * W, L, H), device=dev)
    sh = _lib.PaggShape(_lib.VARIANT_HOMO, Ng, 1, H, 1, Sg, W, L, 0, 0, 0)

    def gather():
        _lib.check(lib.pn_pagg_gather(ctx, ctypes.byref(sh), table.data_ptr(), gi.data_ptr(), gc.data_ptr(),
                                      rows.data_ptr(), _lib.stream_ptr(dev)))
    dt = time_launches(gather, 20)
    read_b = Sg * W * (L * H * 4 + L * 5)                  # 2068 B/path at L=4, H=128
    out["gather_hbm_table"] = {"paths": Sg * W, "table_MB": Ng * L * H * 4 >> 20, "ms": dt * 1e3,
                               "read_GBs": read_b / dt / 1e9, "read_frac_of_hbm_peak": read_b / dt / 1e9 / HBM_PEAK_GBS,
                               "read_plus_write_GBs": (read_b + Sg * W * L * H * 4) / dt / 1e9,
                               "read_plus_write_frac_of_hbm_peak": (read_b + Sg * W * L * H * 4) / dt / 1e9 / HBM_PEAK_GBS,
                               "algorithmic_read_bytes_per_path": L * H * 4 + L * 5,
                               "note": "stand-alone pn_pagg_gather stage: 512-byte rows at random from a 2 GB table (HBM, "
                                       "not the Infinity Cache), gathered rows written back to HBM"}
    del rows
    # the same gather where the product does it: fused into the recurrent forward (rows go HBM -> LDS tile, never back)
    Nf, Ff = Ng, 16
    mdl = pathnet_amd.PathNet_homo(Ff, H, 3, L).to(dev).eval()
    Xf = torch.rand(Nf, Ff, device=dev)
    self32 = gi[:, 0, 0].contiguous()
    _lib.check(lib.pn_profile_configure(ctx, 1, -1))
    with torch.no_grad():
        for _ in range(6):
            mdl(Xf, gi, W, L, self32, gc, None)
    torch.cuda.synchronize()
    prof = read_profile(lib, names, ctx)
    _lib.check(lib.pn_profile_configure(ctx, 0, -1))
    ms = prof["seq_fwd"][0] / prof["seq_fwd"][1]
    out["fused_gather_hbm_table"] = {"paths": Sg * W, "table_MB": Nf * L * H * 4 >> 20, "seq_fwd_ms": ms,
                                     "gather_read_GBs": read_b / (ms * 1e-3) / 1e9,
                                     "gather_read_frac_of_hbm_peak": read_b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "note": "inference forward of PathNet_homo on 2^20 nodes (nothing saved for a backward): the "
                                             "recurrent forward gathers the same rows straight into its LDS tile; the launch is "
                                             "bound by its weight-fragment stream and MFMAs, not by this read (the training "
                                             "forward, which also writes six times what it gathers: profiles/r06_gather.md)"}
    return out


def configs4_workload(dev, n=10_000_000, masked=100_000, seed=0):
    """BASELINE.json configs[4]: Erdos-Renyi graph, n nodes, degree ~16, feat = 128, path_num = 40, path_len = 6; `masked`
    masked nodes drawn uniformly.  X is generated on the device (5.1 GB at 10 M nodes: the same values on every rank)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_sampler_large import er_graph
    F, C, H, W, L = 128, 8, 128, 40, 6
    g = er_graph(n, 16, seed)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 1)
    X = torch.rand((n, F), device=dev, generator=gen)
    rng = np.random.default_rng(seed + 2)
    mask = np.zeros(n, bool)
    mask[rng.choice(n, masked, replace=False)] = True
    return dict(n=n, n_loc=-(-n // 1), F=F, C=C, H=H, W=W, L=L, graph=g, X=X, Y=rng.integers(0, C, n), mask=mask)


def _model_prediction(block, world):
    """tools/scale_model.py's row for `block` at `world` ranks, from the newest single-GPU bench line under profiles/ --
    printed NEXT TO the measurement so that a SCALE record reads against the model without further arithmetic."""
    try:
        import importlib.util
        import contextlib
        import io
        spec = importlib.util.spec_from_file_location("scale_model", os.path.join(ROOT, "tools", "scale_model.py"))
        sm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sm)
        src = sm.newest_bench_json()
        rows = sm.rows_for(json.load(open(src)), block)
        for R, t, sp, parts in rows:
            if R == world:
                return {"source": os.path.basename(src), "ms_per_step": round(t, 3), "speed_up_or_efficiency": round(sp, 3),
                        "collectives_ms": round(parts["collectives"], 3), "link_GBs": sm.DEFAULT_LINK_GBS,
                        "latency_us": sm.DEFAULT_LATENCY_US}
    except Exception as e:      # noqa: BLE001  (a prediction is an annotation: never fatal)
        return {"error": repr(e)[:200]}
    return None


def multi_gpu_blocks(lib, ctx, names, dev, rank, world, red_dev, barrier, args):
    """N > 1 only, after the timed headline: the two configurations north_star names for the 8-GPU box, each a handful of
    steps with every collective's EXPOSED time (what the compute stream actually waits for) per rank:
      bgp_strong           configs[3] stand-in, hetero class, ONE 63 977-node graph node-sharded over the ranks
                           (dist.ShardedAggregator; strong scaling)
      configs4_replicated  configs[4]: 10 M-node Erdos-Renyi graph, path_len 6, 100 000 masked nodes split over the ranks,
                           all of X on every rank, the call restricted to the rows its paths touch (dist.ReplicatedAggregator)
      configs4_sharded     the same step node-sharded with the SPARSE exchange of touched rows
    Sizes shrink with PN_BENCH_MULTI_SMALL=1 (the one-GPU test of this code path)."""
    import torch.distributed as dist
    small = os.environ.get("PN_BENCH_MULTI_SMALL", "0") not in ("", "0")
    out = {}

    def all_ok(ok):
        """every rank's flag: a block runs only where ALL ranks got through its set-up, and counts only where all finished it"""
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def run_block(name, make_wl, steps, model_key, **kw):
        # set-up (host graph, sampler tables, device tensors: no collective in it) may fail on one rank alone -- out of host
        # or device memory on a box smaller than planned for: the ranks vote, and a block that cannot run everywhere is
        # reported as skipped instead of hanging the others.  Nothing here may take the headline line down with it.
        t0 = time.time()
        sr, err = None, None
        try:
            wl = make_wl()
            sr = StepRunner(wl, dev, rank, world, sharded=True, **kw)
        except Exception as e:      # noqa: BLE001
            err = repr(e)[:300]
        if not all_ok(sr is not None):
            out[name] = {"skipped": "set-up failed on at least one rank", "this_rank_error": err}
            return
        setup = time.time() - t0
        try:
            measure_block(name, wl, sr, steps, model_key, setup)
            done = True
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": repr(e)[:300]}
            done = False
        if not all_ok(done) and "error" not in out.get(name, {}):
            out[name] = {"error": "another rank failed inside this block"}
        del sr
        torch.cuda.empty_cache()

    def measure_block(name, wl, sr, steps, model_key, setup):
        m = measure(sr, lib, ctx, names, steps, 2, barrier)
        t = torch.tensor([m["elapsed"]], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        s_tot = torch.tensor([sr.S], dtype=torch.int64, device=red_dev)
        dist.all_reduce(s_tot)
        sr.comm.measure_exposed = True
        k = min(3, steps)
        for e in range(k):
            sr.step(8000 + e)
        torch.cuda.synchronize()
        exposed = {kk: round(v / k, 4) for kk, v in sr.comm.exposed_ms().items()}
        sr.comm.measure_exposed = False
        gathered = [None] * world
        dist.all_gather_object(gathered, {"exposed_ms": exposed, "masked_nodes": sr.S, "stages_ms": m["stages_ms"],
                                          "exchange": getattr(sr.runner, "last_exchange", None)})
        ms = float(t.item()) / steps * 1e3
        out[name] = {"config": "%s: N=%d F=%d hid=%d path_num=%d path_len=%d, %d masked nodes = %d paths/step over %d ranks, %s"
                               % (name, wl["n"], wl["F"], wl["H"], wl["W"], wl["L"], int(s_tot.item()), int(s_tot.item()) * wl["W"],
                                  world, wl.get("cls", "PathNet_homo")),
                     "ms_per_step": ms, "value": int(s_tot.item()) * wl["W"] / (ms * 1e-3), "unit": "paths/s", "steps": steps,
                     "scaling": "strong", "by_rank": gathered, "setup_s": round(setup, 1),
                     "model_prediction": _model_prediction(model_key, world)}

    run_block("bgp_strong", lambda: bgp_workload(world) if not small else dict(F=287, C=8, H=128, W=40, L=4, cls="PathNet", **_shrunk_bgp()),
              max(3, args.steps // 5), "bgp_overlap")
    n4, m4 = (10_000_000, 100_000) if not small else (200_000, 4_000)
    wl4 = {}

    def c4():       # built once (the host graph and the sampler's tables take a minute at 10 M nodes), used by both blocks
        if not wl4:
            wl4.update(configs4_workload(dev, n4, m4))
        return wl4
    run_block("configs4_replicated", c4, 2, "configs4_replicated_touched", hops="otf", replicated=True)
    run_block("configs4_sharded", c4, 2, "configs4_sharded_sparse", hops="otf", exchange="sparse")
    return out


def _shrunk_bgp():
    """a 4 001-node stand-in of the stand-in for the one-GPU test of the N > 1 code path (odd size: blocks are unequal)"""
    n = 4001
    g = synthetic_graph(n, 5, avg_und_deg=3.2)
    rng = np.random.default_rng(6)
    mask = np.zeros(n, bool)
    mask[rng.permutation(n)[: int(0.48 * n)]] = True
    return dict(n=n, n_loc=n, graph=g, X=rng.random((n, 287)).astype(np.float32), Y=rng.integers(0, 8, n), mask=mask)


def spawn_ranks(args, dry_run=False):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (the form the
    driver uses for N > 1; a bare invocation must not fail for launch reasons).  Fewer visible GPUs than ranks (the
    one-GPU test box): the ranks share devices, which RCCL refuses -- the collectives then go through gloo, staged
    through the host, and the line says so (collectives.backend)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if torch.cuda.device_count() < args.gpus and "PN_DIST_BACKEND" not in env:
        env["PN_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    if dry_run:
        return cmd, env
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline configuration only")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph replay of the step (N = 1 only)")
    ap.add_argument("--workload", choices=["cora", "pubmed", "bgp"], default="cora",
                    help="cora = BASELINE.json configs[1], what `value` is quoted on; pubmed = configs[2] as the timed "
                         "workload (profiling runs: tools/pmc_passes.sh); bgp = configs[3] stand-in (hetero class, "
                         "node-sharded with --gpus N: strong scaling of one 63 977-node graph)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)           # (does not return)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or without a "
                         "launcher at all (bench.py then starts its own ranks)" % (args.gpus, world, args.gpus))
    local_rank %= max(1, torch.cuda.device_count())     # (ranks share GPUs only in the one-GPU tests of the N > 1 path)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # PN_DIST_BACKEND=gloo: the N > 1 path with the ranks sharing one GPU (tests; RCCL needs a GPU per rank) -- dist.Comm
    # stages the device tensors through the host, the scalars below travel as host tensors
    backend = os.environ.get("PN_DIST_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        # (a rank that dies inside a collective must not hang the others for the default ten minutes to half an hour)
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("PN_BENCH_PG_TIMEOUT_S", "600")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    import pathnet_amd      # noqa: F401
    from pathnet_amd import _lib
    lib = _lib.load()
    ctx = _lib.context(dev)
    names = stage_names(lib)
    info = _lib.DeviceInfo()
    _lib.check(lib.pn_device_query(ctypes.byref(info)))

    if args.workload == "pubmed" and world > 1:
        raise SystemExit("--workload pubmed is a single-GPU profiling aid")
    wl = (workload(rank, world) if args.workload == "cora" else pubmed_workload() if args.workload == "pubmed"
          else bgp_workload(world))
    n, F, C, H, W, L = wl["n"], wl["F"], wl["C"], wl["H"], wl["W"], wl["L"]
    sharded = world > 1 or os.environ.get("PN_BENCH_FORCE_SHARDED") == "1"   # the env hook exercises the N>1 code path on one GPU
    sr = StepRunner(wl, dev, rank, world, sharded)
    S = sr.S

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    clk0 = clock_mhz(lib, dev)
    m = measure(sr, lib, ctx, names, args.steps, args.warmup, barrier, repeats=4 if world == 1 else 0)
    clk1 = clock_mhz(lib, dev)
    elapsed, dominant, dom_ms = m["elapsed"], m["dominant"], m["dom_ms"]
    collectives = None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s_tot = torch.tensor([S], dtype=torch.int64, device=red_dev)
        dist.all_reduce(s_tot)
        S_total = int(s_tot.item())
        # untimed: a few steps with every collective bracketed by device synchronisation, per rank
        sr.comm.timing = True
        k = min(10, args.steps)
        for e in range(k):
            sr.step(7000 + e)
        sr.comm.timing = False
        mine = {kk: round(v / k * 1e3, 4) for kk, v in sr.comm.seconds.items()}
        # and what the overlapped step actually waits for: per collective, from the moment the compute stream needs the
        # result to the collective's end (events on both streams; zero when it is hidden)
        sr.comm.measure_exposed = True
        for e in range(k):
            sr.step(8000 + e)
        torch.cuda.synchronize()
        exposed = {kk: round(v / k, 4) for kk, v in sr.comm.exposed_ms().items()}
        sr.comm.measure_exposed = False
        gathered_exposed = [None] * world
        dist.all_gather_object(gathered_exposed, exposed)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        # every rank contributes a one through the backend the step used (backend "nccl" IS RCCL on ROCm) and names the
        # device it ran on: ranks_seen == n_gpus and as many distinct devices says the N ranks really were N GPUs
        one = torch.ones(1, dtype=torch.int64, device=red_dev)
        dist.all_reduce(one)
        prop = torch.cuda.get_device_properties(dev)
        devs = [None] * world
        dist.all_gather_object(devs, "%s#%d/%s" % (os.uname().nodename, dev.index, getattr(prop, "uuid", "")))
        collectives = {"rccl_ranks_seen": int(one.item()), "backend": "rccl" if backend == "nccl" else backend,
                       "distinct_devices": len(set(devs)),
                       "ms_per_step_by_rank": gathered,
                       "exposed_ms_per_step_by_rank": gathered_exposed,
                       "overlap": bool(sr.overlap),
                       "note": "ms_per_step_by_rank: each collective bracketed by torch.cuda.synchronize (serialised: an upper "
                               "bound).  exposed_ms_per_step_by_rank: the timed step runs the all-gather of Xh on a communication "
                               "stream under the sampler + index plan + weight packing and the reduce-scatter of dXh (+ fc0's "
                               "backward) under the recurrent / bank weight-gradient GEMMs; exposed = time from the compute "
                               "stream needing the result to the collective's end, 0 when hidden (PN_BENCH_OVERLAP=0 disables)",
                       "bytes_per_step": {"all_gather_Xh": n * H * 4, "reduce_scatter_dXh": n * H * 4,
                                          "all_reduce_grads": sum(q.numel() for q in sr.model.parameters()) * 4}}
    else:
        S_total = S
    ms_per_step = elapsed / args.steps * 1e3
    value = S_total * W / (elapsed / args.steps)

    P = S * W
    pmc_key = {"cora": "", "pubmed": "pubmed_"}.get(args.workload) if world == 1 and not sharded else None
    tr, tr_src = pmc_traffic(dominant, pmc_key + "hbm_bytes_per_launch") if pmc_key is not None else (None, None)
    l2r = pmc_l2_requests(dominant, pmc_key + "l2_requests_per_launch") if pmc_key is not None else None
    roofline = roofline_block(dominant, dom_ms, m["dom_launches"], P, L, H, tr, l2_requests=l2r)
    if tr_src:
        roofline["traffic_source"] = tr_src
    multi = {}
    if world > 1 and not args.no_extras and args.workload == "cora" and os.environ.get("PN_BENCH_MULTI_EXTRAS", "1") != "0":
        try:
            multi = multi_gpu_blocks(lib, ctx, names, dev, rank, world, red_dev, barrier, args)
        except Exception as e:      # noqa: BLE001  (annotations of the N > 1 line: never fatal to it)
            multi = {"multi_gpu_blocks_error": repr(e)[:300]}

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        extras = extras_single_gpu(lib, ctx, names, dev, sr, args)

    result = {
        "metric": "paths aggregated/sec (PAGG fwd+bwd, one training step incl. on-GPU MERW sampling + Adam)",
        "value": value, "unit": "paths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "untimed_steps_before_timed_region": max(1, args.warmup) + 2 + m["settle_steps"],
        "scaling": "strong" if args.workload == "bgp" else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "dtype_note": "fp32 in, fp32 out, fp32 accumulation; the recurrent GEMM products run as %s, everything else in fp32"
                      % ("scaled 2-plane fp16 splits (3 fp16 MFMAs per fp32 product)" if seq_math_name() == "f16x2" else
                         "3-plane bf16 splits (6 bf16 MFMAs per fp32 product)"),
        "seq_math": seq_math_name(),
        "step_api": "pn_pagg_train_step" if (sr.runner is None and sr.fused) else "forward + loss + backward (three calls)",
        "config": {"workload": {"cora": "Cora-shaped synthetic (configs[1])", "pubmed": "Pubmed-shaped synthetic (configs[2])",
                                "bgp": "BGP-sized synthetic, hetero class (configs[3])"}[args.workload] + ": N=%d F=%d C=%d hid=%d path_num=%d path_len=%d, "
                               "%d masked nodes = %d paths/step, %s, dropout 0.7, Adam" %
                               (n, F, C, H, W, L, S_total, S_total * W, wl.get("cls", "PathNet_homo")),
                   "nodes": n, "paths_per_step": S_total * W, "parallelism": "node-shard x%d" % world},
        "roofline": roofline,
        "stages_ms": m["stages_ms"],
        "dispersion": m["dispersion"],
        "device": {"name": info.name.decode(errors="replace"), "arch": info.arch.decode(errors="replace"),
                   "compute_units": info.compute_units, "max_clock_mhz": info.clock_khz / 1e3,
                   "shader_clock_mhz_probe": {"before": clk0, "after": clk1},
                   "note": "probe = shader cycles per 100 MHz wall tick of a one-wave kernel right before / after the "
                           "timed region (light load: the clock under the MFMA kernels is lower)"},
        "library_source_hash": source_hash(),
    }
    if collectives:
        result["collectives"] = collectives
        result["collectives"]["model_prediction"] = _model_prediction("cora_weak_overlap", world)
    result.update(multi)
    if world == 1 and not sharded and not args.no_graph and args.workload == "cora":
        # the same K steps as one captured hipGraph replayed K times (`value` / `ms_per_step` above stay the eager run)
        try:
            result["graph_replay"] = measure_graph(wl, dev, args.steps, args.warmup, barrier)
            result["graph_replay"]["eager_ms_per_step"] = ms_per_step
        except Exception as e:      # noqa: BLE001  (an extra must not take the headline line down with it)
            result["graph_replay"] = {"error": repr(e)[:300]}
    result.update(extras)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the stated baseline: the reference's op sequence on stock torch modules, best thread
        # count; the oracle's port (explicit index plan, cell loop) stays beside it as a second line
        from oracle import merw as _merw
        gn, gu, gv, gp = wl["graph"]
        sel_b = np.flatnonzero(wl["mask"])
        ids_b, codes_b = _merw.sample_full(gn, gu, gv, gp, wl["W"], wl["L"], _merw.DRAW_PHILOX, 1, epoch_count=1)
        result["cpu_baseline"] = cpu_baseline_reference_ops(wl["X"], wl["Y"][sel_b], ids_b[0][sel_b], codes_b[0][sel_b], sel_b,
                                                            wl["F"], wl["H"], wl["C"], wl["W"], wl["L"],
                                                            what="the bench workload (configs[1] shape)")
        result["cpu_baseline_port"] = cpu_baseline(wl, seconds_budget=8.0)
        result["cpu_baseline_configs0"] = cpu_baseline_cornell()
        sb = cpu_baseline_sampler()
        if sb:
            result["cpu_baseline_sampler"] = sb
    if rank == 0:
        emit(result)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()              # rank 0's untimed extras run a little longer: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

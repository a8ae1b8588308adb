"""A/B of the 128-path recurrent kernels (pn_seq4.hip) against the fused ones (pn_pagg.hip) on one GPU, in one process:
PN_SEQ4 is a context knob (pn_context_set_knob), so the same module / inputs / dropout seed run through either set of kernels.

  python tools/ab_seq4.py [--masks 0,1,3,7] [--steps 10] [--out gpurun_out/ab_seq4.json] [--skip-parity]

Parity part: logits, every gradient and the intermediate per-path tensors in the workspace (h_n, saved gates, [x|h]
rows, keep bits, gate gradients) of each mask against mask 0, on the bench workload and on small / ragged / GRU /
hetero / L = 6 / eval shapes; where a tensor differs, the error is broken down by row block, column block and step.
Timing part: per-stage times (library HIP events) and wall time of forward + backward per mask.
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pathnet_amd  # noqa: E402
from pathnet_amd import _lib, modules  # noqa: E402


def a256(x):
    return (x + 255) // 256 * 256


def ws_views(ws, cfgshape, S, W, L, H, C, G=4, SV=5):
    """per-path tensors inside the workspace (ws_layout in pn_pagg.hip, single micro-batch)"""
    lib = _lib.load()
    off = (ctypes.c_int64 * 4)()
    _lib.check(lib.pn_pagg_debug_offsets(ctypes.byref(cfgshape), off))
    P = S * W
    hn = off[2]
    saved = a256(hn + P * H * 4)
    layer1 = off[3]
    outb = a256(layer1 + S * 2 * H * 4)
    xh = a256(outb + S * C * 4)
    keep = a256(xh + P * L * 2 * H * 4)
    dG = a256(keep + P * L * (H // 4))
    dhn = a256(dG + P * L * G * H * 4)

    def f32(o, *shape):
        n = int(np.prod(shape))
        return ws[o:o + 4 * n].view(torch.float32).view(*shape)
    return {"hn": f32(hn, P, H), "saved": f32(saved, P, L, SV, H), "xh": f32(xh, P, L, 2 * H),
            "keep": ws[keep:keep + P * L * (H // 4)].view(P, L, H // 4), "dG": f32(dG, P, L, G * H),
            "dhn": f32(dhn, P, H)}


def breakdown(name, a, b):
    """where two per-path tensors differ: by row mod 128 (blocks of 32), by column block of 32, by step"""
    d = (a.double() - b.double()).abs()
    if a.dtype == torch.uint8:
        d = (a != b).double()
    res = {"max": float(d.max()), "ref_max": float(b.double().abs().max()), "nbad": int((d > 1e-4).sum())}
    if res["max"] > 1e-5:
        P = a.shape[0]
        rows = torch.arange(P, device=a.device)
        rb = (rows % 128) // 32
        dd = d.reshape(P, -1)
        res["by_rowblock"] = [float(dd[rb == i].max()) if (rb == i).any() else 0.0 for i in range(4)]
        last = d.reshape(-1, d.shape[-1])
        nb = max(1, d.shape[-1] // 32)
        res["by_colblock32"] = [float(last[:, 32 * i:32 * (i + 1)].max()) for i in range(min(nb, 16))]
        if d.dim() >= 3:
            res["by_step"] = [float(d[:, t].max()) for t in range(d.shape[1])]
        res["tile_of_first_bad"] = int((dd.max(1).values > 1e-5).nonzero()[0]) // 128
        bad = (d > 1e-5).nonzero()[:6].tolist()
        res["first_bad"] = [(ix, float(a[tuple(ix)]), float(b[tuple(ix)])) for ix in bad]
    return res


def make_case(name, dev, variant="homo", N=600, F=64, H=128, C=5, S=97, W=7, L=4, cell=None, train=True, drop=0.5, seed=0):
    g = torch.Generator().manual_seed(seed)
    cls = {"homo": pathnet_amd.PathNet_homo, "hetero": pathnet_amd.PathNet}[variant]
    torch.manual_seed(seed)
    model = cls(F, H, C, L, dropout=drop, cell=cell).to(dev)
    model.train(train)
    X = torch.rand(N, F, generator=g).to(dev)
    sel = torch.randperm(N, generator=g)[:S].sort().values.to(torch.int32).to(dev)
    ids = torch.randint(0, N, (S, W, L), generator=g).to(torch.int32)
    ids[:, :, 0] = sel.cpu()[:, None]
    codes = torch.randint(0, L, (S, W, L), generator=g).to(torch.uint8)
    Gout = torch.randn(S, C, generator=g).to(dev)
    return dict(name=name, model=model, X=X, ids=ids.to(dev), codes=codes.to(dev), sel=sel, G=Gout, W=W, L=L, S=S, H=H,
                C=C, N=N, F=F, variant=variant, cell=cell, train=train)


def bench_case(dev):
    wl = bench.workload(0, 1)
    gn, u, v, p = wl["graph"]
    smp = pathnet_amd.MerwSampler(gn, u, v, p, wl["L"], device=dev)
    torch.manual_seed(0)
    model = pathnet_amd.PathNet_homo(wl["F"], wl["H"], wl["C"], wl["L"], dropout=0.7).to(dev).train()
    X = torch.from_numpy(wl["X"]).to(dev)
    sel = torch.from_numpy(np.flatnonzero(wl["mask"]).astype(np.int64)).to(dev)
    ids, codes = smp.sample(wl["W"], 1, epoch_count=1)
    ids, codes = ids[0].index_select(0, sel), codes[0].index_select(0, sel)
    Gout = torch.randn(sel.numel(), wl["C"], device=dev)
    return dict(name="bench_cora", model=model, X=X, ids=ids, codes=codes, sel=sel.to(torch.int32), G=Gout, W=wl["W"],
                L=wl["L"], S=int(sel.numel()), H=wl["H"], C=wl["C"], N=wl["n"], F=wl["F"], variant="homo", cell=None,
                train=True)


def run_once(case, mask, seed=123):
    _lib.set_knob("PN_SEQ4", mask)
    m = case["model"]
    torch.manual_seed(seed)
    if case["train"]:
        m.zero_grad(set_to_none=True)
        out = m(case["X"], case["ids"], case["W"], case["L"], case["sel"], case["codes"], None)
        out.backward(case["G"])
        torch.cuda.synchronize()
        ws = out.grad_fn.ws
        grads = {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}
        sh = modules._shape(case["variant"], case["N"], case["F"], case["H"], case["C"], case["S"], case["W"], case["L"],
                            cell=m._cell_kind)
        views = {k: v.clone() for k, v in ws_views(ws, sh, case["S"], case["W"], case["L"], case["H"], case["C"]).items()}
        return out.detach().clone(), grads, views
    with torch.no_grad():
        out = m(case["X"], case["ids"], case["W"], case["L"], case["sel"], case["codes"], None)
    torch.cuda.synchronize()
    return out.detach().clone(), {}, {}


def parity(cases, masks):
    report = {}
    for case in cases:
        ref = run_once(case, 0)
        again = run_once(case, 0)
        rep = {"self_out": float((ref[0] - again[0]).abs().max()),
               "self_grad": max([float((ref[1][k] - again[1][k]).abs().max()) for k in ref[1]] or [0.0])}
        for mask in masks:
            if mask == 0:
                continue
            try:
                got = run_once(case, mask)
            except Exception as e:      # noqa: BLE001
                rep["mask%d" % mask] = {"error": repr(e)[:300]}
                continue
            r = {"out": float((got[0] - ref[0]).abs().max()), "out_ref_max": float(ref[0].abs().max()),
                 "out_nan": bool(torch.isnan(got[0]).any())}
            r["grads"] = {k: [float((got[1][k] - ref[1][k]).abs().max()), float(ref[1][k].abs().max())] for k in ref[1]}
            for k in ref[2]:
                if k == "dhn":
                    continue
                r[k] = breakdown(k, got[2][k], ref[2][k])
            rep["mask%d" % mask] = r
        report[case["name"]] = rep
        print("PARITY", case["name"], json.dumps(rep), flush=True)
    return report


def timing(case, masks, steps):
    lib = _lib.load()
    names = bench.stage_names(lib)
    ctx = _lib.context("cuda")
    m = case["model"]

    def step():
        out = m(case["X"], case["ids"], case["W"], case["L"], case["sel"], case["codes"], None)
        m.zero_grad(set_to_none=True)
        out.backward(case["G"])
    res = {}
    for rnd in range(2):            # two interleaved rounds: box drift shows up as a difference between them
        for mask in masks:
            _lib.set_knob("PN_SEQ4", mask)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            _lib.check(lib.pn_profile_configure(ctx, 1, -1))
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            prof = bench.read_profile(lib, names)
            _lib.check(lib.pn_profile_configure(ctx, 0, -1))
            for _ in range(3):
                step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(30):
                step()
            e1.record()
            torch.cuda.synchronize()
            d = {k: round(v[0] / v[1], 4) for k, v in prof.items()}
            d["_wall_fwd_bwd"] = round(e0.elapsed_time(e1) / 30, 4)
            res["mask%d_round%d" % (mask, rnd)] = d
            print("TIMING %s mask %d round %d: fwd %.3f bwd %.3f wgrad %.3f wall %.3f" % (
                case["name"], mask, rnd, d.get("seq_fwd", -1), d.get("seq_bwd", -1), d.get("wgrad", -1), d["_wall_fwd_bwd"]),
                flush=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--masks", default="0,1")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ab_seq4.json"))
    ap.add_argument("--skip-parity", action="store_true")
    ap.add_argument("--skip-timing", action="store_true")
    ap.add_argument("--pubmed", action="store_true", help="also time a Pubmed-sized batch (378 560 paths)")
    a = ap.parse_args()
    masks = [int(x) for x in a.masks.split(",")]
    dev = torch.device("cuda")
    report = {"masks": masks}
    big = bench_case(dev)
    if not a.skip_parity:
        cases = [make_case("small_ragged", dev, S=97, W=7),                     # 679 paths: 5 full tiles + 39 rows
                 make_case("one_row", dev, S=1, W=1),
                 make_case("exact_tiles", dev, S=64, W=8),                      # 512 paths
                 make_case("eval_nograd", dev, S=97, W=7, train=False),
                 make_case("nodrop", dev, S=50, W=10, drop=0.0),
                 make_case("gru", dev, S=97, W=7, cell="gru"),
                 make_case("hetero", dev, variant="hetero", S=97, W=7),
                 make_case("L6", dev, S=40, W=9, L=6),
                 make_case("L1", dev, S=40, W=9, L=1),
                 big]
        report["parity"] = parity(cases, masks)
    if not a.skip_timing:
        report["timing"] = {"bench_cora": timing(big, masks, a.steps)}
        if a.pubmed:
            pm = make_case("pubmed_sized", dev, N=19717, F=500, C=3, S=9464, W=40, L=4, drop=0.7)
            report["timing"]["pubmed_sized"] = timing(pm, masks, max(3, a.steps // 2))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(report, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

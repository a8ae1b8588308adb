"""Stand-alone times of the node-level GEMMs of a training step (fc0, the distance bank, their backward) through the C ABI:
pn_linear_forward / pn_gemm_f32 / pn_linear_backward at the headline (Cora) and Pubmed shapes, HIP events around blocks of
launches.  Used for the A/Bs of gemm_kernel variants in round 6 (the 32 x 32-tile bf16 x 3 kernel, the split-K target, two K
tiles in flight: profiles/r06_node_gemm.txt, r06_glue.txt sections 16-17), run against two builds of the library
(PN_LIB_PATH).   python tools/bench_node_gemm.py [reps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathnet_amd import _lib  # noqa: E402


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) * 1000.0 / reps)
    best.sort()
    return round(best[len(best) // 2], 2)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    lib = _lib.load()
    ctx = _lib.context("cuda")
    out = {}
    for name, (N, F, H, L) in {"cora": (2708, 1433, 128, 4), "pubmed": (19717, 500, 128, 4)}.items():
        torch.manual_seed(0)
        X = torch.randn(N, F, device="cuda")
        W0 = torch.randn(H, F, device="cuda") / F ** 0.5
        b0 = torch.randn(H, device="cuda")
        Wb = torch.randn(L * H, H, device="cuda") / H ** 0.5
        bb = torch.randn(L * H, device="cuda")
        Xh = torch.empty(N, H, device="cuda")
        Z = torch.empty(N, L * H, device="cuda")
        dZ = torch.randn(N, L * H, device="cuda")
        dXh = torch.empty(N, H, device="cuda")
        gW0, gb0 = torch.empty_like(W0), torch.empty_like(b0)
        gWb, gbb = torch.empty_like(Wb), torch.empty_like(bb)
        ws = torch.empty(_lib.LINEAR_SPLIT_MAX * N * max(H, 1), device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        r = {}
        r["fc0_fwd"] = timed(lambda: _lib.check(lib.pn_linear_forward(ctx, X.data_ptr(), W0.data_ptr(), b0.data_ptr(), N, F, H, 1,
                                                                       Xh.data_ptr(), ws.data_ptr(), ws.numel() * 4, s)), reps)
        r["bank_fwd"] = timed(lambda: _lib.check(lib.pn_gemm_f32(Xh.data_ptr(), H, 1, Wb.data_ptr(), H, 1, Z.data_ptr(), L * H,
                                                                  bb.data_ptr(), N, L * H, H, 1, s)), reps)
        # bank backward as a linear layer in = H, out = L H: g_X = dZ' . Wb, g_W = dZ'^T . Xh, g_b
        r["bank_bwd_dx"] = timed(lambda: _lib.check(lib.pn_linear_backward(ctx, dZ.data_ptr(), Z.data_ptr(), None, Wb.data_ptr(), N, H, L * H,
                                                                            None, None, dXh.data_ptr(), None, 0, s)), reps)
        r["bank_bwd_dw"] = timed(lambda: _lib.check(lib.pn_linear_backward(ctx, dZ.data_ptr(), Z.data_ptr(), Xh.data_ptr(), None, N, H, L * H,
                                                                            gWb.data_ptr(), gbb.data_ptr(), None, None, 0, s)), reps)
        r["fc0_bwd_dw"] = timed(lambda: _lib.check(lib.pn_linear_backward(ctx, dXh.data_ptr(), Xh.data_ptr(), X.data_ptr(), None, N, F, H,
                                                                           gW0.data_ptr(), gb0.data_ptr(), None, None, 0, s)), reps)
        # values, against float64
        ref = torch.relu(X.double() @ W0.double().t() + b0.double())
        r["fc0_fwd_err"] = float((Xh.double() - ref).abs().max())
        d = dXh.double() * (Xh > 0)
        r["fc0_bwd_dw_relerr"] = float((gW0.double() - d.t() @ X.double()).abs().max() / (d.t() @ X.double()).abs().max())
        out[name] = r
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()

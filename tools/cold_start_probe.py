"""Why does the distance-bank GEMM take 22 us inside a training step and 11 us when launched back to back?  The same pn_gemm_f32
call (headline shape: 2708 x 512 x 128) timed with an event pair around EVERY launch, in three settings:
    alone        the launch repeated (operands, result and instructions stay where they were)
    other code   a different kernel of the library between two launches (fc0's split GEMM + its finish: other instructions,
                 operands that fit the caches beside the bank's)
    cold data    256 MB written between two launches (the bank's operands and result lines are gone from L2 / Infinity Cache)
python tools/cold_start_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathnet_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    ctx = _lib.context("cuda")
    N, F, H, L = 2708, 1433, 128, 4
    torch.manual_seed(0)
    X = torch.randn(N, F, device="cuda")
    W0 = torch.randn(H, F, device="cuda") / F ** 0.5
    b0 = torch.randn(H, device="cuda")
    Xh = torch.randn(N, H, device="cuda")
    Xh2 = torch.empty(N, H, device="cuda")
    Wb = torch.randn(L * H, H, device="cuda") / H ** 0.5
    bb = torch.randn(L * H, device="cuda")
    Z = torch.empty(N, L * H, device="cuda")
    ws = torch.empty(_lib.LINEAR_SPLIT_MAX * N * H, device="cuda")
    junk = torch.empty(64 << 20, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def bank():
        _lib.check(lib.pn_gemm_f32(Xh.data_ptr(), H, 1, Wb.data_ptr(), H, 1, Z.data_ptr(), L * H, bb.data_ptr(), N, L * H, H, 1, s))

    def fc0():
        _lib.check(lib.pn_linear_forward(ctx, X.data_ptr(), W0.data_ptr(), b0.data_ptr(), N, F, H, 1, Xh2.data_ptr(), ws.data_ptr(),
                                         ws.numel() * 4, s))

    def timed(between, reps=60):
        ts = []
        for _ in range(reps):
            if between:
                between()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            bank()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1000.0)
        ts.sort()
        return round(ts[len(ts) // 2], 2)

    for _ in range(5):
        bank(), fc0()
    torch.cuda.synchronize()
    out = {"alone": timed(None), "other code between": timed(fc0), "cold data between": timed(lambda: junk.fill_(1.0)),
           "an empty event pair": None}
    ts = []
    for _ in range(60):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1000.0)
    ts.sort()
    out["an empty event pair"] = round(ts[30], 2)
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()

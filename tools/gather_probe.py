"""The path-feature gather against HBM (north_star: ">= 30 % of the HBM-read roofline on the PAGG gather at Pubmed scale"), as
one script that rocprofv3 can wrap (tools/gather_passes.sh: kernel trace + the FETCH_SIZE / WRITE_SIZE counter passes):

  * the stand-alone stage `pn_pagg_gather` (gather_kernel): Pubmed's path count (9464 masked nodes x 40 paths x 4 steps), rows
    of 512 B taken at random from a [2^20, L, H] fp32 table = 2 GB -- HBM, not the 256 MB Infinity Cache;
  * the same rows where the product gathers them: fused into the recurrent forward (seq_fwdh_kernel, training mode: dropout,
    saved tensors) of PathNet_homo over 2^20 nodes.
At Pubmed's REAL size the table is 19717 x 4 x 128 x 4 B = 40 MB: cache-resident, no HBM roofline to speak of; this probe is
the HBM-bound version of the question.      python tools/gather_probe.py [reps]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import pathnet_amd  # noqa: E402
from pathnet_amd import _lib  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lib, ctx = _lib.load(), _lib.context(dev)
    names = bench.stage_names(lib)
    W, L, H = 40, 4, 128
    Ng, Sg = 1 << 20, 9464
    g = torch.Generator(device="cpu").manual_seed(0)
    gi = torch.randint(0, Ng, (Sg, W, L), dtype=torch.int32, generator=g).to(dev)
    gc = torch.randint(0, L, (Sg, W, L), dtype=torch.uint8, generator=g).to(dev)
    read_b = Sg * W * (L * H * 4 + L * 5)                       # 2068 B / path at L = 4, H = 128
    out = {"paths": Sg * W, "table_MB": Ng * L * H * 4 >> 20, "algorithmic_read_bytes_per_path": L * H * 4 + L * 5}
    # ---- stand-alone stage
    table = torch.randn(Ng, L, H, device=dev)
    rows = torch.empty((Sg * W, L, H), device=dev)
    sh = _lib.PaggShape(_lib.VARIANT_HOMO, Ng, 1, H, 1, Sg, W, L, 0, 0, 0)

    def gather():
        _lib.check(lib.pn_pagg_gather(ctx, ctypes.byref(sh), table.data_ptr(), gi.data_ptr(), gc.data_ptr(), rows.data_ptr(),
                                      _lib.stream_ptr(dev)))
    dt = bench.time_launches(gather, reps)
    out["standalone"] = {"kernel": "gather_kernel", "ms": dt * 1e3, "read_GBs": read_b / dt / 1e9,
                         "read_frac_of_hbm_peak": read_b / dt / 1e9 / bench.HBM_PEAK_GBS,
                         "read_plus_write_GBs": (read_b + Sg * W * L * H * 4) / dt / 1e9}
    del table, rows
    # ---- fused into the product's recurrent forward (training mode)
    mdl = pathnet_amd.PathNet_homo(16, H, 3, L, dropout=0.7).to(dev).train()
    Xf = torch.rand(Ng, 16, device=dev)
    self32 = gi[:, 0, 0].contiguous()
    _lib.check(lib.pn_profile_configure(ctx, 1, -1))
    for _ in range(max(3, reps // 2)):
        mdl(Xf, gi, W, L, self32, gc, None)
    torch.cuda.synchronize()
    prof = bench.read_profile(lib, names, ctx)
    _lib.check(lib.pn_profile_configure(ctx, 0, -1))
    ms = prof["seq_fwd"][0] / prof["seq_fwd"][1]
    out["fused"] = {"kernel": "seq_fwdh_kernel (training forward)", "seq_fwd_ms": ms, "gather_read_GBs": read_b / (ms * 1e-3) / 1e9,
                    "gather_read_frac_of_hbm_peak": read_b / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS,
                    "saved_bytes_per_path_written": L * (16 * H + 8 * H + H // 4),
                    "note": "the kernel writes ~6x what it gathers (saved gates 16 B, [x|h] 8 B, keep bits per unit and step) and runs "
                            "(2L-1) * 8 H^2 flops per path at three fp16 MFMAs per product: the gather is a seventh of its traffic"}
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()

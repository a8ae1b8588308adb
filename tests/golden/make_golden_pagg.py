"""Generate tests/golden/pagg_*.npz by running the reference aggregator classes themselves
(ast-loaded from /root/reference, see tests/ref_extract.py) on seeded inputs, in eval() mode
(dropout inactive), forward and backward.  Run in the build container only.

Each fixture: shapes, X, ids [S,W,L], codes [S,W,L], mask [N], every state_dict tensor
("param/<key>"), the reference logits ("out"), the upstream gradient G used for the backward
(loss = sum(out * G)), and the reference gradient of every parameter ("grad/<key>") and of X.
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ref_extract import reference_classes  # noqa: E402

warnings.filterwarnings("ignore")
OUT = os.path.dirname(os.path.abspath(__file__))
CLASS = {"hetero": "PathNet", "homo": "PathNet_homo", "pagg": "PAGG"}


def make(variant, tag, N, F, H, C, W, L, S, seed):
    cls = reference_classes(0.0)
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    model = cls[CLASS[variant]](F, H, C, L if variant != "pagg" else N)
    with torch.no_grad():                       # default-init biases are tiny: make every term matter
        for k, v in model.named_parameters():
            if k.endswith("bias"):
                v.uniform_(-0.3, 0.3)
        if hasattr(model, "attw"):
            model.attw.weight.mul_(4.0)
    X = torch.rand(N, F).requires_grad_(True)
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    sel = np.nonzero(mask)[0]
    ids = rng.integers(0, N, size=(S, W, L))
    ids[:, :, 0] = sel[:, None]
    codes = np.minimum(rng.integers(0, L, size=(S, W, L)), np.arange(L)[None, None, :])
    model.eval()
    out = model(X, torch.tensor(ids.reshape(S, W * L)), W, L, torch.tensor(mask), torch.tensor(codes),
                torch.arange(S * W * L))
    G = torch.randn(S, C)
    (out * G).sum().backward()
    d = dict(variant=variant, N=N, F=F, H=H, C=C, W=W, L=L, S=S, X=X.detach().numpy(), ids=ids.astype(np.int32),
             codes=codes.astype(np.uint8), mask=mask, out=out.detach().numpy(), G=G.numpy(),
             grad_X=X.grad.numpy())
    for k, v in model.state_dict().items():
        d["param/" + k] = v.numpy()
    for k, v in model.named_parameters():
        d["grad/" + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "pagg_%s_%s.npz" % (variant, tag)), **d)
    print(variant, tag, "out absmax %.3f" % out.abs().max().item())


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for variant in ("hetero", "homo", "pagg"):
        make(variant, "h32", N=37, F=19, H=32, C=3, W=5, L=4, S=11, seed=11)
        make(variant, "h64w40", N=61, F=33, H=64, C=5, W=40, L=4, S=23, seed=12)
    for variant in ("hetero", "homo"):
        make(variant, "h32l6", N=45, F=12, H=32, C=4, W=8, L=6, S=17, seed=13)
    # the headline shape: hid = 128, 40 paths, Cornell's N = 183 / S_train = 87 (PathNet_run.py:178); F reduced from
    # 1703 to 160 to keep the fixture small (fc0 is a plain Linear; its width is covered by the oracle tests)
    for variant in ("hetero", "homo", "pagg"):
        make(variant, "h128w40cornell", N=183, F=160, H=128, C=5, W=40, L=4, S=87, seed=14)

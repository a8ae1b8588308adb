"""Generate tests/golden/paggtrain_*.npz: the reference aggregator classes themselves (ast-loaded from /root/reference,
tests/ref_extract.py) in TRAINING mode -- the mode PathNet_run.py:340-352 trains in -- forward and backward, with the
dropout masks they drew recorded.  Run in the build container only.

The classes call ``F.dropout(x, p=dropout, training=self.training)`` twice per forward (PathNet_run.py:194 / :264 on the
[L, P, H] sequence, :209 / :276 on the [S, 2H] classifier input; copy.py:348, :357).  ``F`` in the namespace the classes are
exec'd in is replaced by a thin stand-in whose ``dropout`` draws the same kind of mask -- Bernoulli(1 - p) keep bits scaled by
1 / (1 - p) -- from torch's generator, RECORDS it and applies it; everything else of ``F`` is torch.nn.functional itself.  The
class code is untouched; the product and the oracle are then handed the recorded masks (their explicit-mask hooks), so the
comparison pins the whole training-mode arithmetic: mask position conventions ([L, P, H] time-major rows of the scrambled
hetero sequence included), the 1 / (1 - p) scale, both dropout sites.

Each fixture: shapes, p, X, ids, codes, mask, every state_dict tensor, the two dropout masks, the reference logits, the upstream
gradient G (loss = sum(out * G)), the reference gradient of every parameter and of X."""
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ref_extract import reference_classes  # noqa: E402

warnings.filterwarnings("ignore")
torch.set_num_threads(1)        # index_select's backward sums in thread order: one thread = the same bits on every run
OUT = os.path.dirname(os.path.abspath(__file__))
CLASS = {"hetero": "PathNet", "homo": "PathNet_homo", "pagg": "PAGG"}


class RecordingF(types.ModuleType):
    """torch.nn.functional with a dropout that keeps the masks it applied"""

    def __init__(self):
        super().__init__("F")
        self.masks = []

    def __getattr__(self, name):
        return getattr(TF, name)

    def dropout(self, x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        m = (torch.rand_like(x) >= p).to(x.dtype) / (1.0 - p)
        self.masks.append(m.detach().clone())
        return x * m


def make(variant, tag, N, F, H, C, W, L, S, seed, p):
    cls = reference_classes(p)
    rec = RecordingF()
    for ns in cls["_ns"]:
        ns["F"] = rec                   # the classes look F up in their globals at call time
    if variant == "pagg":               # copy.py's PAGG hard-codes its own rate (0.9, copy.py:348,357) -- recorded as it is
        pass
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    model = cls[CLASS[variant]](F, H, C, L if variant != "pagg" else N)
    with torch.no_grad():
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
    model.train()
    out = model(X, torch.tensor(ids.reshape(S, W * L)), W, L, torch.tensor(mask), torch.tensor(codes),
                torch.arange(S * W * L))
    assert len(rec.masks) == 2 and tuple(rec.masks[0].shape) == (L, S * W, H) and tuple(rec.masks[1].shape) == (S, 2 * H), \
        [tuple(m.shape) for m in rec.masks]
    G = torch.randn(S, C)
    (out * G).sum().backward()
    p_used = 1.0 - 1.0 / float(rec.masks[0].max())
    d = dict(variant=variant, N=N, F=F, H=H, C=C, W=W, L=L, S=S, p=p_used, X=X.detach().numpy(), ids=ids.astype(np.int32),
             codes=codes.astype(np.uint8), mask=mask, out=out.detach().numpy(), G=G.numpy(), grad_X=X.grad.numpy(),
             mask_seq=rec.masks[0].numpy(), mask_cls=rec.masks[1].numpy())
    for k, v in model.state_dict().items():
        d["param/" + k] = v.numpy()
    for k, v in model.named_parameters():
        d["grad/" + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "paggtrain_%s_%s.npz" % (variant, tag)), **d)
    print(variant, tag, "p %.2f out absmax %.3f kept %.3f" % (p_used, out.abs().max().item(), float((rec.masks[0] > 0).float().mean())))


if __name__ == "__main__":
    for variant in ("hetero", "homo", "pagg"):
        make(variant, "h32p7", N=37, F=19, H=32, C=3, W=5, L=4, S=11, seed=21, p=0.7)        # the reference's default rate (PathNet_run.py:49)
        make(variant, "h64w40p5", N=61, F=33, H=64, C=5, W=40, L=4, S=23, seed=22, p=0.5)
    for variant in ("hetero", "homo"):
        make(variant, "h32l6p7", N=45, F=12, H=32, C=4, W=8, L=6, S=17, seed=23, p=0.7)

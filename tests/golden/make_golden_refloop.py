"""Generate tests/golden/refloop_*.npz: the reference's OWN training function, run as written.

Run in the build container only (needs /root/reference).  `train_fixed_indices` (PathNet_run.py:281-403) is taken out of
the reference file with ``ast`` -- the FunctionDef node itself, nothing retyped -- and executed in a namespace that holds
what the script's top level would have put there: ``device = 'cpu'``, ``lr``, ``weight_decay``, ``epochs``, ``new_data``,
``paths_root``, ``marker``, ``tqdm``, sklearn's metrics, and the reference's own ``PathNet`` / ``PathNet_homo`` classes
(ast-loaded as well, tests/ref_extract.py).  Two things are observed without touching the function:
  * the classes are wrapped in a subclass whose ``forward`` logs every call the loop makes -- the Python type, dtype,
    device and shape of each argument (CPU int64 ``neis[train_indices]``, numpy-bool masks, ``indxx`` ...) and the logits --
    and that keeps the freshly initialised ``state_dict``;
  * ``F`` in the classes' namespace is the recording stand-in of make_golden_pagg_train.py, so the dropout masks each
    training forward drew (p = 0.7, the script's default) are kept.
The function trains for ``epochs`` epochs on a tiny synthetic dataset, calls sklearn on ``.cpu()`` logits, ``torch.save``s
the best ``state_dict`` and renames the file (PathNet_run.py:372-373, :398-399) -- all of it as written.

A fixture holds the inputs (X, Y, the three numpy-bool masks, the paths of the epochs that are used, the initial
state_dict, the recorded masks) and what the reference produced: the logits of every forward call in call order with its
mode (train / eval) and mask, the returned metrics, the saved state_dict, the per-call argument log.  tests/
test_reference_loop.py drives pathnet_amd's classes through the same sequence of calls, argument forms included, with
torch.optim.Adam / CrossEntropyLoss / sklearn exactly as the loop uses them, and compares call by call."""
import ast
import json
import os
import sys
import tempfile
import time
import warnings

import numpy as np
import torch
import torch.nn.functional as TF  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden_pagg_train import RecordingF  # noqa: E402
from ref_extract import REF_ROOT, reference_classes  # noqa: E402

warnings.filterwarnings("ignore")
torch.set_num_threads(1)
OUT = os.path.dirname(os.path.abspath(__file__))


def describe(a):
    if torch.is_tensor(a):
        return {"type": "torch.Tensor", "dtype": str(a.dtype), "device": str(a.device), "shape": list(a.shape)}
    if isinstance(a, np.ndarray):
        return {"type": "numpy.ndarray", "dtype": str(a.dtype), "shape": list(a.shape)}
    return {"type": type(a).__name__, "value": a if isinstance(a, (int, float, str)) else None}


def recording(cls, log):
    class Recorded(cls):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            log["ctor"] = [describe(x) for x in a]
            log["init_state"] = {n: v.detach().clone() for n, v in self.state_dict().items()}

        def forward(self, *a):
            out = super().forward(*a)
            log["calls"].append({"training": bool(self.training), "args": [describe(x) for x in a],
                                 "mask": np.asarray(a[4]).copy(), "logits": out.detach().clone()})
            return out
    Recorded.__name__ = cls.__name__
    return Recorded


def make(data_name, tag, N, Fdim, H, C, W, L, epochs, seed, p=0.7):
    cls = reference_classes(p)
    rec = RecordingF()
    for ns in cls["_ns"]:
        ns["F"] = rec
    log = {"calls": []}
    with open(os.path.join(REF_ROOT, "PathNet_run.py")) as f:
        tree = ast.parse(f.read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "train_fixed_indices"]
    assert len(fn) == 1
    import tqdm
    from sklearn.metrics import accuracy_score, f1_score, precision_score, recall_score
    ns = {"torch": torch, "F": TF, "np": np, "os": os, "time": time, "tqdm": tqdm, "device": "cpu", "lr": 0.005,
          "weight_decay": 0.0005, "epochs": epochs, "new_data": "__none__", "paths_root": "./", "marker": "merw",
          "accuracy_score": accuracy_score, "f1_score": f1_score, "precision_score": precision_score, "recall_score": recall_score,
          "PathNet": recording(cls["PathNet"], log), "PathNet_homo": recording(cls["PathNet_homo"], log)}
    exec(compile(ast.Module(body=fn, type_ignores=[]), os.path.join(REF_ROOT, "PathNet_run.py"), "exec"), ns)

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    X = torch.rand(N, Fdim)
    Wtrue = rng.normal(size=(Fdim, C))
    Y = torch.from_numpy(((X.numpy() - 0.5) @ Wtrue).argmax(1).astype(np.int64))     # learnable, balanced labels
    perm = rng.permutation(N)
    masks = [np.zeros(N, bool) for _ in range(3)]                                # numpy bool, as PlanetoidData gives them
    masks[0][perm[:int(0.5 * N)]] = True
    masks[1][perm[int(0.5 * N):int(0.75 * N)]] = True
    masks[2][perm[int(0.75 * N):]] = True
    # the loop wants the whole 1000-epoch path file as Python lists (PathNet_run.py:310-313 .view(1000, N, ...)); only the
    # first `epochs` epochs are read -- the rest is filler that the fixture does not carry
    ids = rng.integers(0, N, size=(epochs, N, W, L))
    ids[:, :, :, 0] = np.arange(N)[None, :, None]
    codes = np.minimum(rng.integers(0, L, size=(epochs, N, W, L)), np.arange(L)[None, None, None, :])
    walks = ids.reshape(-1, L).tolist() + [[0] * L] * ((1000 - epochs) * N * W)
    ptype = codes.reshape(-1, L).tolist() + [[0] * L] * ((1000 - epochs) * N * W)

    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "saved_models"))
    os.chdir(tmp)
    try:
        ret = ns["train_fixed_indices"](X, Y, C, "pathnet", data_name, masks[0], masks[1], masks[2], W, H, L, walks, ptype, 0)
        saved = [f for f in os.listdir("saved_models") if f.startswith(data_name)]
        assert len(saved) == 1, saved
        saved_state = torch.load(os.path.join("saved_models", saved[0]))
    finally:
        os.chdir(cwd)
    n_train = sum(1 for c in log["calls"] if c["training"])
    assert n_train == epochs and len(rec.masks) == 2 * epochs, (n_train, len(rec.masks))
    out = {"data_name": data_name, "N": N, "F": Fdim, "H": H, "C": C, "W": W, "L": L, "epochs": epochs, "p": p,
           "lr": 0.005, "weight_decay": 0.0005,
           "X": X.numpy(), "Y": Y.numpy(), "train_mask": masks[0], "val_mask": masks[1], "test_mask": masks[2],
           "ids": ids.astype(np.int32), "codes": codes.astype(np.uint8), "returned": np.asarray(ret, dtype=np.float64),
           "n_calls": len(log["calls"]),
           "call_training": np.asarray([c["training"] for c in log["calls"]]),
           "call_log": json.dumps({"ctor": log["ctor"], "calls": [c["args"] for c in log["calls"]]})}
    for i, c in enumerate(log["calls"]):
        out["call%d_mask" % i] = c["mask"]
        out["call%d_logits" % i] = c["logits"].numpy()
    for e in range(epochs):
        out["mask_seq%d" % e] = rec.masks[2 * e].numpy()
        out["mask_cls%d" % e] = rec.masks[2 * e + 1].numpy()
    for k, v in log["init_state"].items():
        out["init." + k] = v.numpy()
    for k, v in saved_state.items():
        out["saved." + k] = v.numpy()
    path = os.path.join(OUT, "refloop_%s.npz" % tag)
    np.savez_compressed(path, **out)
    print(tag, "calls", len(log["calls"]), "returned", [round(float(r), 4) for r in ret], os.path.getsize(path), "bytes")


if __name__ == "__main__":
    make("cora", "homo_cora", N=64, Fdim=24, H=32, C=4, W=5, L=4, epochs=6, seed=11)        # -> PathNet_homo (:286-288)
    make("cornell", "hetero_cornell", N=60, Fdim=20, H=32, C=3, W=6, L=4, epochs=6, seed=12)  # -> PathNet (:289-291)

"""Training-loop glue: the reference's ``train_fixed_indices``
(/root/reference/PathNet_run.py:281-403) with everything resident on the GPU.

Same recipe: model by dataset name (:286-291), Adam(lr, weight_decay) + CrossEntropyLoss (:295-297), one training
step per epoch on that epoch's paths (:336-352), validation every epoch (:355-366), test + checkpoint whenever
the validation accuracy improves (:368-389), checkpoint renamed with time stamp and round at the end (:398-401),
returns (macro-F1, micro-F1, macro recall, macro precision, accuracy) of the selected epoch (:403).

What changes is only where the data lives: paths are int32/uint8 device tensors -- either pre-sampled
[epochs, N, W, L] tensors or a MerwSampler that walks each epoch's paths on the GPU just before the step (no
text file, no per-epoch Python parse, PathNet_run.py:317-334) -- masks are index tensors on the device, and the
metrics (sklearn in the reference, :384-389) are computed on the device as well.
"""
import os
import time

import numpy as np
import torch

from . import modules, optim
from .sampler import DRAW_PHILOX

HOMO_DATASETS = ("cora", "citeseer", "pubmed")          # PathNet_run.py:286


def classification_metrics(y_true, y_pred, num_classes):
    """-> (macro F1, micro F1, macro recall, macro precision, accuracy) with scikit-learn's conventions
    (labels = classes present in y_true or y_pred; 0/0 counts as 0)."""
    y_true, y_pred = y_true.reshape(-1).long(), y_pred.reshape(-1).long()
    conf = torch.bincount(y_true * num_classes + y_pred, minlength=num_classes * num_classes).view(
        num_classes, num_classes).double()
    tp = conf.diag()
    support, predicted = conf.sum(1), conf.sum(0)
    present = (support + predicted) > 0
    prec = torch.where(predicted > 0, tp / predicted.clamp(min=1), torch.zeros_like(tp))
    rec = torch.where(support > 0, tp / support.clamp(min=1), torch.zeros_like(tp))
    f1 = torch.where(prec + rec > 0, 2 * prec * rec / (prec + rec).clamp(min=1e-300), torch.zeros_like(tp))
    k = present.sum().clamp(min=1)
    acc = tp.sum() / conf.sum().clamp(min=1)
    return (float((f1 * present).sum() / k), float(acc), float((rec * present).sum() / k),
            float((prec * present).sum() / k), float(acc))


def _index_tensor(mask_or_index, device):
    if isinstance(mask_or_index, np.ndarray):
        t = torch.from_numpy(np.flatnonzero(mask_or_index) if mask_or_index.dtype == np.bool_ else mask_or_index)
    else:
        t = torch.as_tensor(mask_or_index)
        if t.dtype == torch.bool:
            t = torch.nonzero(t, as_tuple=False).flatten()
    return t.to(device=device, dtype=torch.int64)


def train_fixed_indices(X, Y, num_classes, data_name, train_indices, val_indices, test_indices, num_w, hid_size,
                        walk_len, paths, round_i=0, *, epochs=1000, lr=0.005, weight_decay=0.0005, dropout=0.7,
                        device="cuda:0", save_dir="./saved_models", sampler_seed=0, model=None, verbose=False,
                        fused_step=None):
    """paths: (ids, codes) tensors [epochs, N, W, L] (int32 / uint8, any device) or a MerwSampler.
    fused_step: True = forward, loss and backward of the training step in one library call (module.forward_loss ->
    pn_pagg_train_step), False = three calls, None = fused exactly when the batch needs micro-batches (then it saves every
    micro-batch's second forward).  Same values.
    Returns (test macro-F1, micro-F1, macro recall, macro precision, accuracy) at the best-validation epoch."""
    dev = torch.device(device)
    X = torch.as_tensor(X).to(dev).float()
    Y = torch.as_tensor(Y).to(dev).long()
    tr, va, te = (_index_tensor(m, dev) for m in (train_indices, val_indices, test_indices))
    if model is None:
        cls = modules.PathNet_homo if data_name in HOMO_DATASETS else modules.PathNet
        model = cls(X.shape[-1], hid_size, num_classes, walk_len, dropout=dropout).to(dev)
    opt = optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)     # = torch.optim.Adam, one launch
    from_sampler = hasattr(paths, "sample")
    if not from_sampler:
        ids_all, codes_all = (torch.as_tensor(t).to(dev) for t in paths)
    os.makedirs(save_dir, exist_ok=True)
    ckpt = os.path.join(save_dir, data_name + ".pth")
    best_val, result = 0.0, (0.0, 0.0, 0.0, 0.0, 0.0)
    tr32, va32, te32 = tr.to(torch.int32), va.to(torch.int32), te.to(torch.int32)
    if from_sampler:
        # an epoch needs the paths of its train and validation nodes, and the test nodes' only when the validation
        # accuracy improves: three node-list windows of the sampler instead of all N nodes (a walk's draws depend on
        # (epoch, source node, walk) alone: the rows are those of a whole-epoch sample -- tests/test_gpu_sampler.py)
        def paths_of(nodes32, epoch, check=False):
            i, c = paths.sample(num_w, sampler_seed, epoch_begin=epoch, epoch_count=1, draw_source=DRAW_PHILOX,
                                check=check, nodes=nodes32)
            return i[0], c[0]
    else:
        def paths_of(nodes32, epoch, check=False):
            idx = nodes32.long()
            return ids_all[epoch].index_select(0, idx), codes_all[epoch].index_select(0, idx)
    for epoch in range(epochs):
        model.train()
        ids_tr, codes_tr = paths_of(tr32, epoch, check=(epoch == 0))
        loss, out = model.forward_loss(X, ids_tr, num_w, walk_len, tr32, codes_tr, Y[tr], fused=fused_step)
        opt.zero_grad(set_to_none=True)
        optim.backward(loss)        # = loss.backward(), minus the seed gradient's fill and multiply
        opt.step()
        with torch.no_grad():
            model.eval()
            ids_va, codes_va = paths_of(va32, epoch)
            pred = model(X, ids_va, num_w, walk_len, va32, codes_va, None).argmax(1)
            val_acc = float((pred == Y[va]).double().mean())      # the one host round trip of an epoch
            if best_val < val_acc:
                best_val = val_acc
                torch.save(model.state_dict(), ckpt)
                # same X, same weights as the validation forward just above: its projected features and distance
                # bank are still in the module's workspace (PathNet_run.py:362 and :378 recompute them)
                ids_te, codes_te = paths_of(te32, epoch)
                pred = model(X, ids_te, num_w, walk_len, te32, codes_te, None, reuse_tables=True).argmax(1)
                result = classification_metrics(Y[te], pred, num_classes)
        if verbose and (epoch % 50 == 0 or epoch == epochs - 1):
            print("epoch %d loss %.4f val_acc %.4f test_acc %.4f" % (epoch, float(loss), val_acc, result[4]))
    if os.path.exists(ckpt):
        os.rename(ckpt, os.path.join(save_dir, data_name + time.strftime("%Y-%m-%d_%H:%M:%S", time.localtime())
                                     + str(round_i) + ".pth"))
    return result

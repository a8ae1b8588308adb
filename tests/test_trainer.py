"""pathnet_amd/trainer.py: metric conventions against scikit-learn (CPU), and the whole loop on the GPU."""
import glob
import os

import numpy as np
import pytest
import torch

from pathnet_amd import trainer


def test_metrics_match_sklearn_conventions():
    from sklearn.metrics import accuracy_score, f1_score, precision_score, recall_score
    rng = np.random.default_rng(0)
    for C, n in ((5, 300), (7, 40), (3, 9)):
        y = rng.integers(0, C, n)
        p = rng.integers(0, max(C - 1, 1), n)          # one class never predicted -> the 0/0 case
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = (f1_score(y, p, average="macro"), f1_score(y, p, average="micro"),
                    recall_score(y, p, average="macro"), precision_score(y, p, average="macro"), accuracy_score(y, p))
        got = trainer.classification_metrics(torch.as_tensor(y), torch.as_tensor(p), C)
        assert np.allclose(got, want, atol=1e-12), (got, want)


@pytest.mark.gpu
def test_training_loop_on_gpu_with_on_device_sampler(tmp_path):
    import pathnet_amd
    from test_gpu_sampler import synthetic_graph
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    n, u, v, p = synthetic_graph(600, 4, 1)
    W, L, F, C = 20, 4, 24, 3
    Y = rng.integers(0, C, n)
    X = rng.random((n, F)).astype(np.float32) * 0.5
    X[np.arange(n), Y] += 1.5                              # learnable from the ego features
    perm = rng.permutation(n)
    tr, va, te = np.zeros(n, bool), np.zeros(n, bool), np.zeros(n, bool)
    tr[perm[:288]], va[perm[288:480]], te[perm[480:]] = True, True, True
    smp = pathnet_amd.MerwSampler(n, u, v, p, L)
    res = trainer.train_fixed_indices(X, Y, C, "synthetic", tr, va, te, W, 64, L, smp, round_i=3, epochs=40,
                                      dropout=0.3, save_dir=str(tmp_path))
    assert res[4] > 0.8 and all(0.0 <= r <= 1.0 for r in res)
    saved = glob.glob(os.path.join(tmp_path, "synthetic*3.pth"))
    assert len(saved) == 1
    sd = torch.load(saved[0], map_location="cpu")
    assert set(sd) == {"fc0.weight", "fc0.bias", "LSTM.weight_ih_l0", "LSTM.weight_hh_l0", "LSTM.bias_ih_l0",
                       "LSTM.bias_hh_l0", "fc2.weight", "fc2.bias", "attw.weight", "attw.bias"} | {
        "nets.%d.%s" % (d, k) for d in range(L) for k in ("weight", "bias")}
    # pre-sampled tensors instead of the sampler, homophilous class by name
    ids, codes = smp.sample(W, 1, epoch_count=5)
    res2 = trainer.train_fixed_indices(X, Y, C, "cora", tr, va, te, W, 64, L, (ids, codes), epochs=5, dropout=0.3,
                                       save_dir=str(tmp_path))
    assert len(res2) == 5

"""bench.StepRunner's training step (sample -> forward -> loss -> backward -> Adam, /root/reference/PathNet_run.py:336-352) with the next
epoch's paths prefetched on a sampling stream (round 5) against the same step run strictly in sequence: the same paths, the same
losses."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _small_workload():
    import bench
    n, F, C, H, W, L = 300, 40, 4, 64, 12, 4
    g = bench.synthetic_graph(n, 11)
    rng = np.random.default_rng(12)
    mask = np.zeros(n, bool)
    mask[rng.permutation(n)[:140]] = True
    return dict(n=n, n_loc=n, F=F, C=C, H=H, W=W, L=L, graph=g, X=rng.random((n, F)).astype(np.float32),
                Y=rng.integers(0, C, n), mask=mask)


def _run(prefetch, epochs, monkeypatch):
    import bench
    monkeypatch.setenv("PN_BENCH_PREFETCH", "1" if prefetch else "0")
    torch.manual_seed(0)
    sr = bench.StepRunner(_small_workload(), torch.device("cuda", 0), 0, 1, sharded=False)
    assert sr.prefetch == prefetch
    sr.model.set_dropout(0.0)           # (the dropout seed comes from torch's generator: keep the two runs comparable)
    losses, used = [], []
    for e in epochs:
        losses.append(float(sr.step(e)))
        if prefetch:
            slot = 1 - sr.pref[1]       # the buffer this step read (the other one is being filled for e + 1)
            used.append(sr.bufs[slot][0].clone())
        else:
            used.append(sr.ids_buf.clone())
    torch.cuda.synchronize()
    return losses, used, sr


def test_prefetched_paths_are_the_epochs_own(monkeypatch):
    epochs = [3, 4, 5, 6, 20, 21, 7, 7, 8]         # consecutive runs, a jump, a repeat: the prefetch must never hand out a stale epoch
    l_seq, ids_seq, sr0 = _run(False, epochs, monkeypatch)
    l_pre, ids_pre, sr1 = _run(True, epochs, monkeypatch)
    for e, a, b in zip(epochs, ids_seq, ids_pre):
        assert torch.equal(a, b), e                                      # the walker is bit-exact: identical paths
        want, _ = sr0.smp.sample(sr0.wl["W"], 1234, epoch_begin=e, epoch_count=1, nodes=sr0.sel32)
        assert torch.equal(a, want), e
    assert np.allclose(l_seq, l_pre, rtol=2e-5, atol=1e-6), (l_seq, l_pre)
    assert l_seq[-1] < l_seq[0]                                          # and it trains

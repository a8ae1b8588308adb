"""The fixed-order backward (pn_pagg_shape.deterministic, pathnet_amd module attribute `deterministic`,
torch.use_deterministic_algorithms / PN_DETERMINISTIC): every gradient of two runs on the same inputs and dropout seed
is BITWISE equal, and equals the default (atomic) backward within the parity tolerance.

The reference has no such switch -- its gather backward is torch's embedding / index_select backward, whose CUDA kernels
add with atomics too (PathNet_run.py:164,185-197 via autograd) -- so the contract here is torch's own for
torch.use_deterministic_algorithms(True): same inputs, same bits."""
import os

import numpy as np
import pytest
import torch
from gradcheck import ZERO_OK_HETERO, assert_grads_close

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_env():
    from pathnet_amd import _lib
    old = {k: os.environ.get(k) for k in ("PN_COMPACT", "PN_DETERMINISTIC")}
    seq4 = _lib.get_knob("PN_SEQ4")
    yield
    _lib.set_knob("PN_SEQ4", seq4)
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _case(variant, S, W, L, H=128, cell=None, drop=0.5, N=2708, F=96, C=7, seed=0, hubs=True):
    import pathnet_amd
    g = torch.Generator().manual_seed(seed)
    cls = {"homo": pathnet_amd.PathNet_homo, "hetero": pathnet_amd.PathNet, "pagg": pathnet_amd.PAGG}[variant]
    torch.manual_seed(seed)
    kw = {} if variant == "pagg" else {"cell": cell}
    m = cls(F, H, C, L, dropout=drop, **kw).cuda().train()
    X = torch.rand(N, F, generator=g).cuda()
    sel = torch.randperm(N, generator=g)[:S].sort().values.to(torch.int32)
    ids = torch.randint(0, N, (S, W, L), generator=g).to(torch.int32)
    if hubs:        # a few nodes on thousands of path steps: long runs of equal destination rows, crossing many chunks
        hub = torch.rand(S, W, L, generator=g) < 0.3
        ids[hub] = torch.randint(0, 5, (int(hub.sum()),), generator=g).to(torch.int32)
    ids[:, :, 0] = sel[:, None]
    codes = torch.randint(0, L, (S, W, L), generator=g).to(torch.uint8)
    G = torch.randn(S, C, generator=g).cuda()
    return m, X, ids.cuda(), codes.cuda(), sel.cuda(), G


def _run(case, det, seed=7):
    m, X, ids, codes, sel, G = case
    m.deterministic = det
    torch.manual_seed(seed)          # the module draws its dropout seed from torch's generator
    m.zero_grad(set_to_none=True)
    out = m(X, ids, ids.shape[1], ids.shape[2], sel, codes, None)
    out.backward(G)
    torch.cuda.synchronize()
    return out.detach().clone(), {k: v.grad.detach().clone() for k, v in m.named_parameters()}


def _check(case, runs=3):
    ref_out, ref_g = _run(case, False)
    out0, g0 = _run(case, True)
    assert torch.equal(out0, ref_out)                   # the forward is the same code either way
    hetero = type(case[0]).__name__ == "PathNet"
    assert_grads_close(g0, ref_g, zero_ok=ZERO_OK_HETERO if hetero else ())
    for _ in range(runs - 1):
        out, g = _run(case, True)
        assert torch.equal(out, out0)
        for k in g0:
            assert torch.equal(g[k], g0[k]), k
    return g0


@pytest.mark.parametrize("variant,S,W,L,H,cell", [
    ("homo", 1299, 40, 4, 128, None),       # the headline shape's index plan: 208 k path steps onto <= 10.8 k rows
    ("hetero", 700, 20, 4, 128, None),      # ego rows of other masked nodes' paths
    ("pagg", 500, 16, 4, 128, None),        # tanh RNN, no attention
    ("homo", 300, 12, 6, 128, "gru"),
    ("homo", 300, 12, 4, 64, "mean"),       # order-agnostic encoder: the scatter alone
    ("hetero", 120, 9, 3, 288, None),       # generic recurrence (H > 256), step-by-step scatter
    ("homo", 1, 1, 4, 128, None),           # a single path: one chunk, no crossing run
])
def test_backward_is_bitwise_reproducible(variant, S, W, L, H, cell):
    _check(_case(variant, S, W, L, H=H, cell=cell))


def test_bitwise_reproducible_with_micro_batches_and_compact_rows():
    """several micro-batches (gradients accumulate across them) over the touched-row compaction of the bank"""
    from pathnet_amd import modules as M
    os.environ["PN_COMPACT"] = "1"
    case = _case("homo", 600, 16, 4)
    m = case[0]
    m.workspace_budget = M.workspace_bytes("homo", 2708, 96, 128, 7, 1, 16, 4, deterministic=True) + 200 * (
        M.workspace_bytes("homo", 2708, 96, 128, 7, 1025, 16, 4, deterministic=True) -
        M.workspace_bytes("homo", 2708, 96, 128, 7, 1, 16, 4, deterministic=True)) // 1024
    g = _check(case, runs=2)
    os.environ["PN_COMPACT"] = "0"
    m.workspace_budget = None
    ref_out, ref_g = _run(case, True)
    assert_grads_close(g, ref_g)     # one batch over the dense bank: same sums in another order


def test_bitwise_reproducible_with_the_64_path_kernels():
    from pathnet_amd import _lib
    _lib.set_knob("PN_SEQ4", 7)
    _check(_case("homo", 400, 20, 4), runs=2)


def test_duplicate_masked_nodes_and_out_of_range_are_summed_in_order():
    """sel with repeated nodes (two pooling groups add to one row of d Xh): the default path adds them with atomics, the
    deterministic one through the sorted scatter"""
    case = list(_case("homo", 64, 8, 4, hubs=False))
    sel = case[4].clone()
    sel[1::2] = sel[0::2]          # every node twice
    ids = case[2].clone()
    ids[:, :, 0] = sel[:, None]
    case[2], case[4] = ids, sel
    _check(tuple(case), runs=2)


def test_default_mode_follows_torch_and_the_environment():
    from pathnet_amd import modules as M
    assert M.deterministic_default() is False
    os.environ["PN_DETERMINISTIC"] = "1"
    assert M.deterministic_default() is True
    os.environ["PN_DETERMINISTIC"] = "0"
    torch.use_deterministic_algorithms(True)
    try:
        assert M.deterministic_default() is True
    finally:
        torch.use_deterministic_algorithms(False)


def test_linear_backward_with_workspace_is_bitwise_reproducible():
    """pn_linear_backward (fc0 of the node-sharded path): with the chunk-sum workspace two runs agree to the bit and match
    the atomic path within tolerance"""
    from pathnet_amd import _lib
    lib = _lib.load()
    torch.manual_seed(3)
    rows, in_f, out_f = 20000, 300, 128
    X, W = torch.randn(rows, in_f, device="cuda"), torch.randn(out_f, in_f, device="cuda")
    dY, Y = torch.randn(rows, out_f, device="cuda"), torch.randn(rows, out_f, device="cuda")
    ws = torch.empty(_lib.LINEAR_BWD_SPLIT_MAX * (out_f * in_f + out_f), dtype=torch.float32, device="cuda")

    def run(use_ws):
        gW, gb = torch.full_like(W, 5.0), torch.full((out_f,), 5.0, device="cuda")
        _lib.check(lib.pn_linear_backward(_lib.context("cuda"), dY.data_ptr(), Y.data_ptr(), X.data_ptr(), W.data_ptr(), rows, in_f,
                                          out_f, gW.data_ptr(), gb.data_ptr(), None, ws.data_ptr() if use_ws else None,
                                          ws.numel() * 4 if use_ws else 0, None))
        torch.cuda.synchronize()
        return gW, gb
    a, b = run(True), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    c = run(False)
    d = (dY * (Y > 0)).double()
    for got in (a, c):
        assert (got[0] - (d.t() @ X.double()).float()).abs().max().item() < 2e-3
        assert (got[1] - d.sum(0).float()).abs().max().item() < 2e-3


def test_sharded_step_single_rank_is_bitwise_reproducible():
    """the node-sharded runner (dist.py) on one rank: Xh_in / g_Xh entry + pn_linear_backward with its workspace"""
    from pathnet_amd import dist
    m, X, ids, codes, sel, G = _case("homo", 500, 16, 4)
    m.deterministic = True
    outs = []
    for _ in range(2):
        runner = dist.ShardedAggregator(m, X.shape[0], 0, X.shape[0], dropout_seed=5)
        m.zero_grad(set_to_none=True)
        out = runner(X, ids.reshape(ids.shape[0], -1), ids.shape[1], ids.shape[2], sel, codes)
        out.backward(G)
        torch.cuda.synchronize()
        outs.append({k: v.grad.detach().clone() for k, v in m.named_parameters()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_deterministic_backward_captured_in_a_hip_graph_replays_the_same_bits():
    """the sorted scatter (rocPRIM radix sort included) records into a hipGraph: a replay gives the eager run's bits"""
    m, X, ids, codes, sel, G = _case("homo", 200, 12, 4, N=500, F=40, C=5, drop=0.0)
    m.deterministic = True
    params = list(m.parameters())

    def step():
        out = m(X, ids, ids.shape[1], ids.shape[2], sel, codes, None)
        return torch.autograd.grad(out, params, G)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            ref = step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = [r.clone() for r in ref]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        got = step()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)

"""Aggregator HIP path (through the C ABI) against the reference goldens and the CPU oracle."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import golden, golden_files
from gradcheck import ZERO_OK_HETERO, assert_grads_close, named_grads
from oracle import pagg_oracle as po

pytestmark = pytest.mark.gpu
TOL_OUT = 1e-5          # north_star: within 1e-5 on PAGG fp32 outputs


def build_module(variant, F, H, C, L, N, params):
    import pathnet_amd
    cls = {"hetero": pathnet_amd.PathNet, "homo": pathnet_amd.PathNet_homo, "pagg": pathnet_amd.PAGG}[variant]
    m = cls(F, H, C, L if variant != "pagg" else N)
    if params is not None:
        missing = m.load_state_dict({k: torch.as_tensor(v) for k, v in params.items()}, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    return m.cuda()


def run_module(m, X, ids, codes, mask, W, L):
    S = int(mask.sum())
    neis = torch.as_tensor(ids.reshape(S, W * L).astype(np.int64))           # CPU int64 like the reference
    lt = torch.as_tensor(codes.reshape(S, W, L).astype(np.int64))
    return m(X, neis, W, L, mask, lt, None)


def test_gemm_f32_matches_torch():
    from pathnet_amd import _lib
    lib = _lib.load()
    torch.manual_seed(0)
    for (M, N, K) in [(64, 64, 32), (70, 33, 45), (2708, 128, 1433), (5, 512, 128), (257, 130, 7)]:
        A = torch.randn(M, K, device="cuda")
        B = torch.randn(N, K, device="cuda")
        bias = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        _lib.check(lib.pn_gemm_f32(A.data_ptr(), K, 1, B.data_ptr(), K, 1, C.data_ptr(), N, bias.data_ptr(), M, N, K, 0,
                                   None))
        ref = (A.double() @ B.double().t() + bias.double()).float()
        err = (C - ref).abs().max().item()
        assert err < 1e-4 * max(1.0, K ** 0.5 / 8), (M, N, K, err)
        # transposed operands (m-contiguous A, n-contiguous B) + relu
        At, Bt = A.t().contiguous(), B.t().contiguous()
        _lib.check(lib.pn_gemm_f32(At.data_ptr(), 1, M, Bt.data_ptr(), 1, N, C.data_ptr(), N, None, M, N, K, 1, None))
        ref = torch.relu(A.double() @ B.double().t()).float()
        assert (C - ref).abs().max().item() < 1e-4 * max(1.0, K ** 0.5 / 8), (M, N, K, "T")


def test_linear_forward_and_backward_match_torch():
    """pn_linear_forward (with and without the split-K workspace: same values) and pn_linear_backward (one zero-fill
    launch, bias gradient as row sums of the weight-gradient GEMM) against torch's nn.Linear (+ReLU)."""
    from pathnet_amd import _lib
    lib = _lib.load()
    torch.manual_seed(3)
    for (rows, in_f, out_f, relu) in [(2708, 1433, 128, 1), (300, 77, 64, 0), (5, 16, 32, 1)]:
        X = torch.randn(rows, in_f, device="cuda")
        W = torch.randn(out_f, in_f, device="cuda") / in_f ** 0.5
        b = torch.randn(out_f, device="cuda")
        Y0, Y1 = torch.empty(rows, out_f, device="cuda"), torch.empty(rows, out_f, device="cuda")
        ws = torch.empty(_lib.LINEAR_SPLIT_MAX * rows * out_f, device="cuda")
        _lib.check(lib.pn_linear_forward(_lib.context("cuda"), X.data_ptr(), W.data_ptr(), b.data_ptr(), rows, in_f, out_f, relu,
                                         Y0.data_ptr(), None, 0, None))
        _lib.check(lib.pn_linear_forward(_lib.context("cuda"), X.data_ptr(), W.data_ptr(), b.data_ptr(), rows, in_f, out_f, relu,
                                         Y1.data_ptr(), ws.data_ptr(), ws.numel() * 4, None))
        ref = X.double() @ W.double().t() + b.double()
        ref = (torch.relu(ref) if relu else ref).float()
        assert (Y0 - ref).abs().max().item() < 2e-5 and (Y1 - ref).abs().max().item() < 2e-5
        dY = torch.randn(rows, out_f, device="cuda")
        gW, gb, gX = torch.full_like(W, 7.0), torch.full_like(b, 7.0), torch.empty_like(X)
        _lib.check(lib.pn_linear_backward(_lib.context("cuda"), dY.data_ptr(), Y1.data_ptr() if relu else None, X.data_ptr(), W.data_ptr(),
                                          rows, in_f, out_f, gW.data_ptr(), gb.data_ptr(), gX.data_ptr(), None, 0, None))
        d = (dY * (ref > 0)).double() if relu else dY.double()
        scale = max(1.0, rows ** 0.5 / 8)
        assert (gW - (d.t() @ X.double()).float()).abs().max().item() < 1e-4 * scale
        assert (gb - d.sum(0).float()).abs().max().item() < 1e-4 * scale
        assert (gX - (d @ W.double()).float()).abs().max().item() < 1e-4
        gb2 = torch.full_like(b, 7.0)       # bias gradient alone
        _lib.check(lib.pn_linear_backward(_lib.context("cuda"), dY.data_ptr(), Y1.data_ptr() if relu else None, None, None, rows, in_f, out_f,
                                          None, gb2.data_ptr(), None, None, 0, None))
        assert (gb2 - d.sum(0).float()).abs().max().item() < 1e-4 * scale


def test_gemm_detects_transposes():
    from pathnet_amd import _lib
    lib = _lib.load()
    M, N, K = 64, 64, 64
    A = torch.eye(M, K, device="cuda")
    B = torch.arange(N * K, device="cuda", dtype=torch.float32).reshape(N, K)      # asymmetric
    C = torch.empty(M, N, device="cuda")
    _lib.check(lib.pn_gemm_f32(A.data_ptr(), K, 1, B.data_ptr(), K, 1, C.data_ptr(), N, None, M, N, K, 0, None))
    assert torch.equal(C, B.t().contiguous())


@pytest.mark.parametrize("variant", ["hetero", "homo", "pagg"])
def test_gather_stage_matches_plan(variant):
    from pathnet_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    N, H, S, W, L = 300, 128, 37, 40, 4
    table = torch.randn(N, L, H, device="cuda")
    ids = rng.integers(0, N, (S, W, L)).astype(np.int32)
    codes = rng.integers(0, L, (S, W, L)).astype(np.uint8)
    rows = torch.empty(S * W, L, H, device="cuda")
    sh = _lib.PaggShape({"hetero": 0, "homo": 1, "pagg": 2}[variant], N, 1, H, 1, S, W, L)
    d_ids, d_codes = torch.as_tensor(ids).cuda(), torch.as_tensor(codes).cuda()      # keep alive across the call
    _lib.check(lib.pn_pagg_gather(_lib.context("cuda"), ctypes.byref(sh), table.data_ptr(), d_ids.data_ptr(), d_codes.data_ptr(),
                                  rows.data_ptr(), None))
    torch.cuda.synchronize()
    node, code, group, member, ego = po.plan(variant, ids, codes, S, W, L)
    order = np.argsort(group * W + member, kind="stable")               # rows come out in pooling-group order
    want = table.cpu()[torch.as_tensor(node[order]), torch.as_tensor(code[order])]
    assert torch.equal(rows.cpu(), want)


@pytest.mark.parametrize("name", golden_files("pagg_*.npz"))
def test_forward_matches_reference_golden(name):
    g = golden(name)
    variant = str(g["variant"])
    N, F, H, C, W, L = (int(g[k]) for k in "NFHCWL")
    params = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    m = build_module(variant, F, H, C, L, N, params).eval()
    with torch.no_grad():
        out = run_module(m, torch.as_tensor(g["X"]).cuda(), g["ids"], g["codes"], g["mask"], W, L)
    err = np.abs(out.cpu().numpy() - g["out"]).max()
    assert err < TOL_OUT, err


@pytest.mark.parametrize("variant", ["hetero", "homo", "pagg"])
@pytest.mark.parametrize("H,W,L,S,N,F,C", [(128, 40, 4, 87, 183, 1703, 5), (64, 7, 4, 33, 90, 50, 3),
                                           (32, 40, 4, 130, 200, 30, 7), (256, 10, 4, 21, 64, 40, 4)])
def test_forward_matches_oracle_random(variant, H, W, L, S, N, F, C):
    torch.manual_seed(1)
    rng = np.random.default_rng(2)
    m = build_module(variant, F, H, C, L, N, None)
    with torch.no_grad():
        for k, v in m.named_parameters():
            if k.endswith("bias"):
                v.uniform_(-0.3, 0.3)
    X = torch.rand(N, F)
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    sel = np.flatnonzero(mask)
    ids = rng.integers(0, N, (S, W, L))
    ids[:, :, 0] = sel[:, None]
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    m.eval()
    with torch.no_grad():
        out = run_module(m, X.cuda(), ids, codes, mask, W, L).cpu()
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    want, inter = po.forward(variant, params, X, ids, codes, sel, W, L, return_intermediates=True)
    err = (out - want).abs().max().item()
    assert err < TOL_OUT, (err, want.abs().max().item())


@pytest.mark.parametrize("variant", ["hetero", "homo"])
def test_l6_paths(variant):
    torch.manual_seed(4)
    rng = np.random.default_rng(4)
    N, F, H, C, W, L, S = 120, 24, 128, 3, 40, 6, 50
    m = build_module(variant, F, H, C, L, N, None).eval()
    X = torch.rand(N, F)
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    sel = np.flatnonzero(mask)
    ids = rng.integers(0, N, (S, W, L))
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    with torch.no_grad():
        out = run_module(m, X.cuda(), ids, codes, mask, W, L).cpu()
    want = po.forward(variant, {k: v.cpu() for k, v in m.state_dict().items()}, X, ids, codes, sel, W, L)
    assert (out - want).abs().max().item() < TOL_OUT


@pytest.mark.parametrize("variant", ["hetero", "homo", "pagg"])
def test_training_mode_with_injected_dropout_masks(variant):
    """F.dropout cannot be matched bit for bit across RNGs; with the reference's mask injected the
    training-mode forward must still agree (SURVEY.md §7 'hard parts')."""
    torch.manual_seed(7)
    rng = np.random.default_rng(7)
    N, F, H, C, W, L, S = 80, 20, 64, 4, 12, 4, 31
    m = build_module(variant, F, H, C, L, N, None).train()
    pdrop = 0.7
    P = S * W
    mask_seq = (torch.rand(L, P, H) >= pdrop).float() / (1 - pdrop)
    mask_cls = (torch.rand(S, 2 * H) >= pdrop).float() / (1 - pdrop)
    m._mask_seq, m._mask_cls = mask_seq.cuda(), mask_cls.cuda()
    X = torch.rand(N, F)
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    sel = np.flatnonzero(mask)
    ids = rng.integers(0, N, (S, W, L))
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    with torch.no_grad():
        out = run_module(m, X.cuda(), ids, codes, mask, W, L).cpu()
    want = po.forward(variant, {k: v.cpu() for k, v in m.state_dict().items()}, X, ids, codes, sel, W, L,
                      drop_seq=mask_seq, drop_cls=mask_cls)
    assert (out - want).abs().max().item() < TOL_OUT


def test_builtin_dropout_statistics_and_determinism():
    torch.manual_seed(9)
    rng = np.random.default_rng(9)
    N, F, H, C, W, L, S = 60, 16, 64, 3, 40, 4, 40
    import pathnet_amd
    m = pathnet_amd.PathNet_homo(F, H, C, L, dropout=0.5).cuda().train()
    X = torch.rand(N, F).cuda()
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    ids = rng.integers(0, N, (S, W, L))
    codes = np.zeros((S, W, L), np.int64)
    with torch.no_grad():
        torch.manual_seed(1)
        a = run_module(m, X, ids, codes, mask, W, L)
        torch.manual_seed(1)
        b = run_module(m, X, ids, codes, mask, W, L)
        c = run_module(m, X, ids, codes, mask, W, L)
        m.eval()
        e = run_module(m, X, ids, codes, mask, W, L)
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, e)
    assert torch.isfinite(a).all()


# ------------------------------------------------------------------------------------------------
# backward
# ------------------------------------------------------------------------------------------------
# every comparison of gradients goes through tests/gradcheck.py: relative to the tensor's own largest element
def zero_ok(variant):
    return ZERO_OK_HETERO if variant == "hetero" else ()


@pytest.mark.parametrize("name", golden_files("pagg_*.npz"))
def test_backward_matches_reference_golden(name):
    g = golden(name)
    variant = str(g["variant"])
    N, F, H, C, W, L = (int(g[k]) for k in "NFHCWL")
    params = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    m = build_module(variant, F, H, C, L, N, params).eval()
    X = torch.as_tensor(g["X"]).cuda().requires_grad_(True)
    out = run_module(m, X, g["ids"], g["codes"], g["mask"], W, L)
    (out * torch.as_tensor(g["G"]).cuda()).sum().backward()
    ref = {k: g["grad/" + k] for k, _ in m.named_parameters()}
    ref["X"] = g["grad_X"]
    assert_grads_close(named_grads(m, {"X": X.grad}), ref, zero_ok=zero_ok(variant))


@pytest.mark.parametrize("variant", ["hetero", "homo", "pagg"])
@pytest.mark.parametrize("H,W,S,N,F,C,train", [(128, 40, 87, 183, 300, 5, False), (128, 40, 50, 100, 64, 3, True),
                                                (64, 9, 33, 70, 20, 4, True), (256, 6, 20, 40, 16, 2, False)])
def test_backward_matches_oracle_random(variant, H, W, S, N, F, C, train):
    torch.manual_seed(21)
    rng = np.random.default_rng(22)
    L = 4
    m = build_module(variant, F, H, C, L, N, None)
    with torch.no_grad():
        for k, v in m.named_parameters():
            if k.endswith("bias"):
                v.uniform_(-0.3, 0.3)
    X = torch.rand(N, F)
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    sel = np.flatnonzero(mask)
    ids = rng.integers(0, N, (S, W, L))
    ids[:, :, 0] = sel[:, None]
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    G = torch.randn(S, C)
    drop_seq = drop_cls = None
    if train:
        pdrop = 0.5
        drop_seq = (torch.rand(L, S * W, H) >= pdrop).float() / (1 - pdrop)
        drop_cls = (torch.rand(S, 2 * H) >= pdrop).float() / (1 - pdrop)
        m.train()
        m._mask_seq, m._mask_cls = drop_seq.cuda(), drop_cls.cuda()
    else:
        m.eval()
    Xd = X.cuda().requires_grad_(True)
    out = run_module(m, Xd, ids, codes, mask, W, L)
    (out * G.cuda()).sum().backward()
    pr = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    Xo = X.clone().requires_grad_(True)
    want = po.forward(variant, pr, Xo, ids, codes, sel, W, L, drop_seq=drop_seq, drop_cls=drop_cls)
    (want * G).sum().backward()
    assert (out.detach().cpu() - want.detach()).abs().max().item() < TOL_OUT
    ref = {k: pr[k].grad for k, _ in m.named_parameters()}
    ref["X"] = Xo.grad
    assert_grads_close(named_grads(m, {"X": Xd.grad}), ref, zero_ok=zero_ok(variant))


@pytest.mark.parametrize("variant", ["hetero", "homo", "pagg"])
def test_accuracy_against_float64_truth(variant):
    """The recurrent GEMMs run fp32 products as six bf16 MFMAs (pn_kernels.h).  Measured against a float64
    evaluation of the oracle, the HIP path may not be less accurate than a few times the fp32 CPU restatement."""
    import json, os
    torch.manual_seed(5)
    rng = np.random.default_rng(6)
    H, W, S, N, F, C, L = 128, 40, 120, 300, 200, 6, 4
    m = build_module(variant, F, H, C, L, N, None).eval()
    X = torch.rand(N, F)
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    sel = np.flatnonzero(mask)
    ids = rng.integers(0, N, (S, W, L))
    ids[:, :, 0] = sel[:, None]
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    G = torch.randn(S, C)
    Xd = X.cuda().requires_grad_(True)
    out = run_module(m, Xd, ids, codes, mask, W, L)
    (out * G.cuda()).sum().backward()
    got = {k: v.grad.cpu().double() for k, v in m.named_parameters()}
    got["X"], got["out"] = Xd.grad.cpu().double(), out.detach().cpu().double()

    def oracle(dtype):
        pr = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        Xo = X.clone().requires_grad_(True)
        y = po.forward(variant, pr, Xo, ids, codes, sel, W, L, dtype=dtype)
        (y * G.to(dtype)).sum().backward()
        r = {k: v.grad.double() for k, v in pr.items()}
        r["X"], r["out"] = Xo.grad.double(), y.detach().double()
        return r

    truth, cpu32 = oracle(torch.float64), oracle(torch.float32)
    report, bad = {}, {}
    for k in truth:
        scale = max(1e-30, truth[k].abs().max().item())
        e_hip = (got[k] - truth[k]).abs().max().item() / scale
        e_cpu = (cpu32[k] - truth[k]).abs().max().item() / scale
        report[k] = {"hip": e_hip, "cpu_fp32": e_cpu}
        if not e_hip <= max(8 * e_cpu, 2e-6):
            bad[k] = report[k]
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/accuracy_%s.json" % variant, "w") as f:
        json.dump(report, f, indent=1)
    assert not bad, bad


def test_builtin_dropout_backward_is_consistent_with_forward():
    """With the built-in Philox masks the backward must regenerate exactly the forward's mask:
    check d(out)/d(params) by finite differences along one random direction, same seed."""
    torch.manual_seed(5)
    rng = np.random.default_rng(5)
    import pathnet_amd
    N, F, H, C, W, L, S = 50, 12, 64, 3, 10, 4, 20
    m = pathnet_amd.PathNet_homo(F, H, C, L, dropout=0.4).cuda().train()
    X = torch.rand(N, F).cuda()
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    ids = rng.integers(0, N, (S, W, L))
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    G = torch.randn(S, C).cuda()

    def loss():
        torch.manual_seed(77)           # same dropout seed every call
        return (run_module(m, X, ids, codes, mask, W, L) * G).sum()

    l0 = loss()
    l0.backward()
    direction = {k: torch.randn_like(v) for k, v in m.named_parameters()}
    analytic = sum((v.grad * direction[k]).sum().item() for k, v in m.named_parameters())
    eps = 1e-3
    with torch.no_grad():
        for k, v in m.named_parameters():
            v.add_(eps * direction[k])
        lp = loss().item()
        for k, v in m.named_parameters():
            v.sub_(2 * eps * direction[k])
        lm = loss().item()
    numeric = (lp - lm) / (2 * eps)
    assert abs(numeric - analytic) < 2e-2 * max(1.0, abs(analytic)), (numeric, analytic)


def test_training_loop_reduces_loss():
    """The unchanged reference recipe: Adam(lr=0.005, wd=5e-4) + CrossEntropyLoss (PathNet_run.py:295-297)."""
    torch.manual_seed(3)
    rng = np.random.default_rng(3)
    import pathnet_amd
    N, F, H, C, W, L, S = 120, 32, 64, 4, 20, 4, 60
    m = pathnet_amd.PathNet(F, H, C, L, dropout=0.3).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=0.005, weight_decay=0.0005)
    lossf = torch.nn.CrossEntropyLoss()
    Y = torch.as_tensor(rng.integers(0, C, N))
    X = (torch.rand(N, F) + torch.nn.functional.one_hot(Y, F).float() * 2).cuda()
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    ids = rng.integers(0, N, (S, W, L))
    ids[:, :, 0] = np.flatnonzero(mask)[:, None]
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    first = last = None
    for step in range(60):
        m.train()
        out = run_module(m, X, ids, codes, mask, W, L)
        loss = lossf(out, Y[mask].cuda())
        opt.zero_grad()
        loss.backward()
        opt.step()
        first = loss.item() if first is None else first
        last = loss.item()
    assert last < 0.5 * first, (first, last)


# ------------------------------------------------------------------------------------------------
# node-sharded entry points (Xh_in / g_Xh / pn_linear_backward) on one GPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["homo", "hetero", "pagg"])
def test_sharded_runner_hip_ops_match_plain_module(variant):
    """ShardedAggregator with the HIP backend and no process group = the plain module: exercises the
    projected-features entry (Xh_in), the g_Xh output and pn_linear_backward that the multi-GPU path uses."""
    from pathnet_amd import dist as pdist
    torch.manual_seed(31)
    rng = np.random.default_rng(31)
    N, F, H, C, W, L, S = 150, 40, 128, 5, 40, 4, 70
    m = build_module(variant, F, H, C, L, N, None).eval()
    X = torch.rand(N, F).cuda()
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    sel = np.flatnonzero(mask)
    ids = rng.integers(0, N, (S, W, L))
    ids[:, :, 0] = sel[:, None]
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    G = torch.randn(S, C).cuda()
    out_a = run_module(m, X, ids, codes, mask, W, L)
    (out_a * G).sum().backward()
    grads_a = {k: v.grad.clone() for k, v in m.named_parameters()}
    m.zero_grad()
    runner = pdist.ShardedAggregator(m, N, 0, N)
    out_b = runner(X, torch.as_tensor(ids.reshape(S, -1)), W, L, torch.as_tensor(sel.astype(np.int32)),
                   torch.as_tensor(codes))
    (out_b * G).sum().backward()
    assert (out_a - out_b).abs().max().item() < 1e-6
    assert_grads_close(named_grads(m), grads_a, zero_ok=zero_ok(variant))


# ------------------------------------------------------------------------------------------------
# full-size configurations (BASELINE.json configs): size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["homo", "pagg"])
def test_pubmed_scale_batch_invariance_and_oracle_subset(variant):
    """Pubmed-size forward+backward (N=19717, F=500, 9464 masked nodes = 378 560 paths).  For the homo / PAGG
    classes a node's logits depend only on its own paths, so (a) any sub-batch must reproduce the rows of the
    full batch, and (b) a small sub-batch is checked against the CPU oracle."""
    torch.manual_seed(41)
    rng = np.random.default_rng(41)
    N, F, H, C, W, L = 19717, 500, 128, 3, 40, 4
    S = 9464
    m = build_module(variant, F, H, C, L, N, None).eval()
    X = torch.rand(N, F)
    Xd = X.cuda()
    sel = np.sort(rng.permutation(N)[:S])
    mask = np.zeros(N, bool)
    mask[sel] = True
    ids = rng.integers(0, N, (S, W, L)).astype(np.int32)
    ids[:, :, 0] = sel[:, None]
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :]).astype(np.uint8)
    d_ids, d_codes = torch.as_tensor(ids).cuda(), torch.as_tensor(codes).cuda()
    out = m(Xd, d_ids, W, L, torch.as_tensor(sel.astype(np.int32)).cuda(), d_codes, None)
    assert out.shape == (S, C) and torch.isfinite(out).all()
    out.sum().backward()
    g_full = {k: v.grad.clone() for k, v in m.named_parameters()}
    assert all(torch.isfinite(g).all() for g in g_full.values())
    pick = np.sort(rng.permutation(S)[:64])
    with torch.no_grad():
        sub = m(Xd, d_ids[pick], W, L, torch.as_tensor(sel[pick].astype(np.int32)).cuda(), d_codes[pick], None)
    assert (sub - out[pick].detach()).abs().max().item() < 2e-6          # batch-composition invariance
    want = po.forward(variant, {k: v.detach().cpu() for k, v in m.state_dict().items()}, X, ids[pick], codes[pick],
                      sel[pick], W, L)
    assert (sub.cpu() - want).abs().max().item() < TOL_OUT
    # linearity of the backward in the upstream gradient: grad(2*G) == 2*grad(G)
    m.zero_grad()
    out2 = m(Xd, d_ids, W, L, torch.as_tensor(sel.astype(np.int32)).cuda(), d_codes, None)
    (2.0 * out2).sum().backward()
    assert_grads_close(named_grads(m), {k: 2.0 * v for k, v in g_full.items()})


def test_bgp_scale_hetero_full_batch_matches_the_oracle():
    """configs[3]: a BGP-sized batch of the hetero class PathNet (PathNet_run.py:150-211; :286-291 selects it for bgp) --
    N = 63 977 nodes, F = 287, C = 8, 30 708 masked nodes x 40 paths x 4 steps = 1 228 320 paths.  Its rows read paths
    of OTHER masked nodes (the [W, S] re-view of an S-major index, :196-197), so no sub-batch can stand in for the
    batch: the fp32 CPU oracle runs once on the whole batch (forward + autograd backward, about a minute on the GPU
    box's host cores) and every logit and every parameter gradient is compared.  Then: a slice of the batch
    (group_slice) is bit for bit those rows of the whole-batch result."""
    torch.manual_seed(61)
    rng = np.random.default_rng(61)
    N, F, H, C, W, L, S = 63977, 287, 128, 8, 40, 4, 30708
    m = build_module("hetero", F, H, C, L, N, None).eval()
    X = torch.rand(N, F)
    sel = np.sort(rng.permutation(N)[:S])
    mask = np.zeros(N, bool)
    mask[sel] = True
    ids = rng.integers(0, N, (S, W, L)).astype(np.int32)
    ids[:, :, 0] = sel[:, None]
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :]).astype(np.uint8)
    G = torch.randn(S, C) / S
    Xd = X.cuda()
    d_ids, d_codes, d_sel = torch.as_tensor(ids).cuda(), torch.as_tensor(codes).cuda(), torch.as_tensor(sel.astype(np.int32)).cuda()
    out = m(Xd, d_ids, W, L, d_sel, d_codes, None)
    (out * G.cuda()).sum().backward()
    got = out.detach().cpu()
    grads = {k: v.grad.detach().cpu().clone() for k, v in m.named_parameters()}
    # slices of the batch: the rows of the whole-batch result, bit for bit
    with torch.no_grad():
        for begin, count in ((0, 257), (12345, 1000), (S - 300, 300)):
            part = m(Xd, d_ids, W, L, d_sel, d_codes, None, group_slice=(begin, count))
            assert torch.equal(part.cpu(), got[begin:begin + count]), (begin, count)
    del out, part
    torch.cuda.empty_cache()
    pr = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    want = po.forward("hetero", pr, X, ids, codes, sel, W, L)
    (want * G).sum().backward()
    err = (got - want.detach()).abs()
    assert err.max().item() < TOL_OUT, (err.max().item(), int(err.argmax()) // C)
    # (G is scaled by 1 / S: the gradients are 1e-3 ... 1e-9 -- the bound is relative to each tensor's own largest element)
    assert_grads_close(grads, {k: pr[k].grad for k in grads}, zero_ok=zero_ok("hetero"))


def test_large_batches_are_sized_not_rejected():
    """Round 1 refused S*W*L*5*H >= 2^32 elements; the kernels now address every per-path tensor as a 64-bit tile
    base + a 32-bit in-tile offset, and a batch beyond the workspace budget is walked in micro-batches."""
    from pathnet_amd import _lib, modules
    full = modules.workspace_bytes("homo", 1000, 16, 128, 3, 50000, 40, 4)      # S*W*L*5*H = 5.1e9 elements
    assert full > 50000 * 40 * 4 * 5 * 128 * 4
    bg = modules.pick_batch_groups("homo", 1000, 16, 128, 3, 50000, 40, 4, budget=8 << 30)
    assert 0 < bg < 50000
    assert modules.workspace_bytes("homo", 1000, 16, 128, 3, 50000, 40, 4, batch_groups=bg) <= 8 << 30
    with pytest.raises(_lib.PnError):
        modules.workspace_bytes("homo", 1000, 16, 128, 3, 100, 40, 4, S_total=50, group_begin=0)   # slice > batch


def test_cora_config_forward_backward_full_parity():
    """BASELINE.json configs[1] at its real size (N=2708, F=1433, C=7, hid=128, W=40, L=4, 1300 masked nodes =
    52 000 paths): every logit and every gradient against the CPU oracle (training mode, injected masks)."""
    torch.manual_seed(51)
    rng = np.random.default_rng(51)
    N, F, H, C, W, L, S = 2708, 1433, 128, 7, 40, 4, 1300
    m = build_module("homo", F, H, C, L, N, None).train()
    X = (torch.rand(N, F) < 0.0127).float()
    X = X / X.sum(1, keepdim=True).clamp(min=1.0)
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    sel = np.flatnonzero(mask)
    ids = rng.integers(0, N, (S, W, L))
    ids[:, :, 0] = sel[:, None]
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    pdrop = 0.7
    drop_seq = (torch.rand(L, S * W, H) >= pdrop).float() / (1 - pdrop)
    drop_cls = (torch.rand(S, 2 * H) >= pdrop).float() / (1 - pdrop)
    m._mask_seq, m._mask_cls = drop_seq.cuda(), drop_cls.cuda()
    G = torch.randn(S, C) / S
    out = run_module(m, X.cuda(), ids, codes, mask, W, L)
    (out * G.cuda()).sum().backward()
    pr = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    want = po.forward("homo", pr, X, ids, codes, sel, W, L, drop_seq=drop_seq, drop_cls=drop_cls)
    (want * G).sum().backward()
    assert (out.detach().cpu() - want.detach()).abs().max().item() < TOL_OUT
    assert_grads_close(named_grads(m), {k: pr[k].grad for k, _ in m.named_parameters()})


# ------------------------------------------------------------------------------------------------
# hidden sizes that are not multiples of 32 (`-hid` is any integer, /root/reference/PathNet_run.py:52; default 64)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant,H,cell", [("homo", 50, None), ("hetero", 100, None), ("pagg", 50, None),
                                            ("homo", 20, "gru"), ("hetero", 72, "mean")])
def test_hidden_size_that_is_not_a_multiple_of_32(variant, H, cell):
    """the module pads the hidden size with zero units up to the kernels' multiple of 32: forward (training mode, injected
    dropout masks) and every gradient against the CPU oracle at the module's OWN hidden size"""
    import pathnet_amd
    torch.manual_seed(11)
    rng = np.random.default_rng(11)
    N, F, C, W, L, S = 90, 24, 4, 12, 4, 33
    cls = {"hetero": pathnet_amd.PathNet, "homo": pathnet_amd.PathNet_homo, "pagg": pathnet_amd.PAGG}[variant]
    m = cls(F, H, C, L if variant != "pagg" else N, cell=cell).cuda().train()
    pdrop = 0.5
    mask_seq = (torch.rand(L, S * W, H) >= pdrop).float() / (1 - pdrop)
    mask_cls = (torch.rand(S, 2 * H) >= pdrop).float() / (1 - pdrop)
    m._mask_seq, m._mask_cls = mask_seq.cuda(), mask_cls.cuda()
    X = torch.rand(N, F)
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    sel = np.flatnonzero(mask)
    ids = rng.integers(0, N, (S, W, L))
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    Xd = X.cuda().requires_grad_(True)
    out = run_module(m, Xd, ids, codes, mask, W, L)
    G = torch.randn(S, C)
    out.backward(G.cuda())
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    Xr = X.clone().requires_grad_(True)
    want = po.forward(variant, params, Xr, ids, codes, sel, W, L, drop_seq=mask_seq, drop_cls=mask_cls, cell=cell)
    assert (out.detach().cpu() - want.detach()).abs().max().item() < TOL_OUT
    want.backward(G)
    ref = {k: params[k].grad for k, _ in m.named_parameters()}
    ref["X"] = Xr.grad
    assert_grads_close(named_grads(m, {"X": Xd.grad}), ref, zero_ok=zero_ok(variant))
    # inference, built-in dropout off: same values with the padded tables reused by a second forward
    m.eval()
    with torch.no_grad():
        e1 = run_module(m, X.cuda(), ids, codes, mask, W, L)
        neis = torch.as_tensor(ids.reshape(S, W * L).astype(np.int64))
        e2 = m(X.cuda(), neis, W, L, mask, torch.as_tensor(codes.astype(np.int64)), None, reuse_tables=True)
    we = po.forward(variant, {k: v.detach() for k, v in params.items()}, X, ids, codes, sel, W, L, cell=cell)
    assert (e1.cpu() - we).abs().max().item() < TOL_OUT and torch.equal(e1, e2)


# ---- training mode against the reference classes themselves: their forward / backward in .train() with the dropout masks
# they drew recorded (tests/golden/make_golden_pagg_train.py); the module is handed the same masks -------------------------
@pytest.mark.parametrize("name", golden_files("paggtrain_*.npz"))
def test_training_mode_matches_reference_golden(name):
    g = golden(name)
    variant = str(g["variant"])
    N, F, H, C, W, L = (int(g[k]) for k in "NFHCWL")
    params = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    m = build_module(variant, F, H, C, L, N, params).train()
    m.set_dropout(float(g["p"]))
    m._mask_seq, m._mask_cls = torch.as_tensor(g["mask_seq"]).cuda(), torch.as_tensor(g["mask_cls"]).cuda()
    X = torch.as_tensor(g["X"]).cuda().requires_grad_(True)
    out = run_module(m, X, g["ids"], g["codes"], g["mask"], W, L)
    err = np.abs(out.detach().cpu().numpy() - g["out"]).max()
    # north_star: 1e-5 on the fp32 outputs, ABSOLUTE (the bound used to scale with |out|_inf, which the x10 masks of
    # PAGG's p = 0.9 push to 4): measured worst 8e-7 (profiles/r06_pytest_gpu_final.txt prints it)
    print("training-mode golden %s: max |out - reference| = %.2e (|out|_inf %.2f)" % (name, err, np.abs(g["out"]).max()))
    assert err < TOL_OUT, err
    (out * torch.as_tensor(g["G"]).cuda()).sum().backward()
    ref = {k: g["grad/" + k] for k, _ in m.named_parameters()}
    ref["X"] = g["grad_X"]
    assert_grads_close(named_grads(m, {"X": X.grad}), ref, zero_ok=zero_ok(variant))

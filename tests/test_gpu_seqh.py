"""The recurrent kernels on the fp16 matrix pipe (pathnet_amd/csrc/pn_seqh.hip: three fp16 MFMAs over scaled two-plane
splits per fp32 product; pn_pagg_shape.seq_math, the default) against
  * the bf16 x 3 kernels of rounds 1-3 (pn_pagg.hip, seq_math = bf16x3) on the same module, inputs and dropout seed,
  * the CPU oracle (oracle/pagg_oracle.py, pinned against the reference classes) in fp32 and fp64,
over every hidden size and cell the fused path has, and at the edges of fp16's range: recurrent weights of 1e-14 (the
magnitudes weight decay leaves in saved_models/cornell.pth), gathered rows up to 1e4, upstream gradients scaled by 1e-8
and 1e+4.  They replace nn.LSTM / nn.RNN forward and autograd backward of /root/reference/PathNet_run.py:164,195,265,351
and baseline/GPRGNN/src/copy.py:308,349; the contract is the bf16 kernels': logits within 1e-5, gradients within
3e-5 * max(1, |g|_inf)."""
import numpy as np
import pytest
import torch
from gradcheck import ZERO_OK_HETERO, assert_grads_close

from oracle import pagg_oracle as po

pytestmark = pytest.mark.gpu


def _case(variant, H, S, W, L, cell=None, drop=0.5, N=400, F=48, C=5, seed=0):
    import pathnet_amd
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    if variant == "pagg":
        m = pathnet_amd.PAGG(F, H, C, N, dropout=drop, cell=cell)
    else:
        cls = {"homo": pathnet_amd.PathNet_homo, "hetero": pathnet_amd.PathNet}[variant]
        m = cls(F, H, C, L, dropout=drop, cell=cell)
    m = m.cuda().train()
    X = torch.rand(N, F, generator=g).cuda()
    sel = torch.randperm(N, generator=g)[:S].sort().values.to(torch.int32)
    ids = torch.randint(0, N, (S, W, L), generator=g).to(torch.int32)
    ids[:, :, 0] = sel[:, None]
    codes = torch.randint(0, L, (S, W, L), generator=g).to(torch.uint8)
    G = torch.randn(S, C, generator=g).cuda()
    return m, X, ids.cuda(), codes.cuda(), sel.cuda(), G


def _run(case, math, seed=7, gscale=1.0):
    m, X, ids, codes, sel, G = case
    m.seq_math = math
    torch.manual_seed(seed)          # the module draws its dropout seed from torch's generator
    m.zero_grad(set_to_none=True)
    out = m(X, ids, ids.shape[1], ids.shape[2], sel, codes, None)
    out.backward(G * gscale)
    torch.cuda.synchronize()
    return out.detach().clone(), {k: v.grad.detach().clone() for k, v in m.named_parameters()}


@pytest.mark.parametrize("variant,H,S,W,L,cell,drop", [
    ("homo", 128, 97, 7, 4, None, 0.5),         # the headline cell: full tiles and a ragged one
    ("homo", 128, 1, 1, 4, None, 0.5),          # a single path
    ("homo", 128, 64, 8, 4, None, 0.0),         # whole tiles, no dropout
    ("homo", 128, 40, 9, 6, None, 0.5),         # path length 6 (configs[4])
    ("homo", 128, 40, 9, 1, None, 0.5),         # one step: no recurrent products at all
    ("homo", 128, 97, 7, 4, "gru", 0.5),        # GRU on the four gate slots
    ("hetero", 128, 97, 7, 4, None, 0.7),       # the hetero index plan (no step-0 run merging)
    ("pagg", 128, 97, 7, 4, None, 0.9),         # tanh RNN, dropout 0.9 (x scaled by 10)
    ("homo", 32, 50, 5, 4, None, 0.5),          # one wave per workgroup
    ("pagg", 32, 50, 5, 4, None, 0.5),          # RNN at H = 32: an odd number of weight-stream units
    ("homo", 64, 50, 5, 4, None, 0.5),
    ("pagg", 96, 50, 5, 4, None, 0.5),          # not a power of two: no step-0 merge; odd unit count for the RNN
    ("homo", 96, 50, 5, 4, "gru", 0.5),
    ("homo", 160, 30, 5, 4, None, 0.5),
    ("hetero", 192, 30, 5, 4, None, 0.5),
    ("homo", 224, 30, 5, 4, "rnn", 0.5),
    ("homo", 256, 30, 5, 4, None, 0.5),         # the largest fused size: 133 KB tile in the BPTT
    ("pagg", 256, 30, 5, 4, "lstm", 0.5),
])
def test_f16_kernels_match_the_bf16_kernels(variant, H, S, W, L, cell, drop):
    case = _case(variant, H, S, W, L, cell, drop)
    ref_out, ref_g = _run(case, "bf16x3")
    out, g = _run(case, "f16x2")
    assert not torch.isnan(out).any()
    assert (out - ref_out).abs().max().item() <= 2e-6
    assert_grads_close(g, ref_g, rel=1e-5, zero_ok=ZERO_OK_HETERO if variant == "hetero" else ())


def _oracle_case(variant, H, S, W, L, N=300, F=40, C=4, keep=0.5, seed=1, cell=None):
    import pathnet_amd
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 4)
    if variant == "pagg":
        m = pathnet_amd.PAGG(F, H, C, N, dropout=1.0 - keep, cell=cell)
    else:
        cls = {"homo": pathnet_amd.PathNet_homo, "hetero": pathnet_amd.PathNet}[variant]
        m = cls(F, H, C, L, dropout=1.0 - keep, cell=cell)
    m = m.cuda().train()
    X = torch.rand(N, F, generator=g)
    rng = np.random.default_rng(seed + 1)
    sel = np.sort(rng.choice(N, S, replace=False))
    ids = rng.integers(0, N, (S, W, L)).astype(np.int32)
    ids[:, :, 0] = sel[:, None]
    codes = rng.integers(0, L, (S, W, L)).astype(np.uint8)
    mask_seq = (torch.rand(L, S * W, H, generator=g) < keep).float() / keep
    mask_cls = (torch.rand(S, 2 * H, generator=g) < keep).float() / keep
    Gout = torch.randn(S, C, generator=g)
    return m, X, sel, ids, codes, mask_seq, mask_cls, Gout


def _hip(m, X, sel, ids, codes, mask_seq, mask_cls, Gout, W, L):
    S, N = len(sel), X.shape[0]
    m._mask_seq, m._mask_cls = mask_seq.cuda(), mask_cls.cuda()
    mask = np.zeros(N, bool)
    mask[sel] = True
    m.zero_grad(set_to_none=True)
    out = m(X.cuda(), torch.as_tensor(ids.reshape(S, W * L).astype(np.int64)), W, L, mask,
            torch.as_tensor(codes.astype(np.int64)), None)
    out.backward(Gout.cuda())
    torch.cuda.synchronize()
    return out.detach().cpu(), {k: v.grad.detach().cpu() for k, v in m.named_parameters()}


def _oracle(variant, m, X, sel, ids, codes, mask_seq, mask_cls, Gout, W, L, dtype):
    params = {k: v.detach().cpu().to(dtype).clone().requires_grad_(True) for k, v in m.state_dict().items()}
    want = po.forward(variant, params, X, ids, codes, sel, W, L, drop_seq=mask_seq, drop_cls=mask_cls, dtype=dtype)
    want.backward(Gout.to(dtype))
    return want.detach(), {k: params[k].grad for k, _ in m.named_parameters()}


@pytest.mark.parametrize("variant,H", [("homo", 128), ("hetero", 128), ("pagg", 128), ("homo", 64), ("homo", 256)])
def test_f16_kernels_match_the_oracle(variant, H):
    """forward and every gradient against the CPU oracle with the same explicit dropout masks"""
    W, L = 11, 4
    m, X, sel, ids, codes, ms, mc, G = _oracle_case(variant, H, 70, W, L)
    m.seq_math = "f16x2"
    out, g = _hip(m, X, sel, ids, codes, ms, mc, G, W, L)
    want, wg = _oracle(variant, m, X, sel, ids, codes, ms, mc, G, W, L, torch.float32)
    assert (out - want).abs().max().item() < 1e-5
    assert_grads_close(g, wg, zero_ok=ZERO_OK_HETERO if variant == "hetero" else ())


def _against_fp64(variant, m, X, sel, ids, codes, ms, mc, G, W, L, floor):
    """max |hip - fp64| per tensor must be within 8 x stock fp32 torch's own distance from fp64, or `floor` of |value|_inf"""
    out, g = _hip(m, X, sel, ids, codes, ms, mc, G, W, L)
    w64, g64 = _oracle(variant, m, X, sel, ids, codes, ms, mc, G, W, L, torch.float64)
    w32, g32 = _oracle(variant, m, X, sel, ids, codes, ms, mc, G, W, L, torch.float32)
    assert not torch.isnan(out).any()
    e_hip, e_t = (out.double() - w64).abs().max().item(), (w32.double() - w64).abs().max().item()
    assert e_hip <= max(8.0 * e_t, 2e-6 * max(1.0, w64.abs().max().item())), ("logits", e_hip, e_t)
    for k in g:
        scale = max(g64[k].abs().max().item(), 1e-300)
        e_hip = (g[k].double() - g64[k]).abs().max().item()
        e_t = (g32[k].double() - g64[k]).abs().max().item()
        assert not torch.isnan(g[k]).any(), k
        assert e_hip <= max(8.0 * e_t, floor * scale), (k, e_hip, e_t, scale)


@pytest.mark.parametrize("math", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("variant", ["homo", "pagg"])
def test_recurrent_weights_of_1e_minus_14(variant, math):
    """weight decay drives unused recurrent weights towards zero (saved_models/cornell.pth holds entries of 1e-14): far
    below fp16's smallest normal number 6e-5 -- the planes are scaled by the weights' own maximum"""
    W, L = 9, 4
    m, X, sel, ids, codes, ms, mc, G = _oracle_case(variant, 128, 60, W, L, seed=3)
    cell = m._cell()
    with torch.no_grad():
        cell.weight_hh_l0.mul_(1e-14)               # a whole matrix at 1e-14 ...
        cell.weight_ih_l0[::3].mul_(1e-14)          # ... and rows of it inside an ordinary one
    m.seq_math = math
    _against_fp64(variant, m, X, sel, ids, codes, ms, mc, G, W, L, floor=2e-6)
    out, g = _hip(m, X, sel, ids, codes, ms, mc, G, W, L)
    w64, g64 = _oracle(variant, m, X, sel, ids, codes, ms, mc, G, W, L, torch.float64)
    k = "LSTM.weight_hh_l0" if variant == "homo" else "RNN.weight_hh_l0"
    # the gradient of the tiny matrix is an ordinary number and must be right in relative terms
    assert (g[k].double() - g64[k]).abs().max().item() <= 3e-5 * g64[k].abs().max().item()


@pytest.mark.parametrize("math", ["f16x2", "bf16x3"])
def test_gathered_rows_up_to_1e4(math):
    """|x| up to ~1e4 times the dropout factor 2 would overflow unscaled fp16 (65504)"""
    W, L = 9, 4
    m, X, sel, ids, codes, ms, mc, G = _oracle_case("hetero", 128, 60, W, L, seed=4)
    with torch.no_grad():
        m.fc0.weight.mul_(1e4 / 4.0)                # Xh, and with it the bank's output, reaches ~1e4
        m.fc0.bias.mul_(1e4 / 4.0)
    m.seq_math = math
    _against_fp64("hetero", m, X, sel, ids, codes, ms, mc, G, W, L, floor=2e-6)


@pytest.mark.parametrize("gscale", [1e-8, 1e4])
def test_upstream_gradient_scales(gscale):
    """the BPTT scales every tile of gate gradients by its own maximum and the weight-gradient GEMM by the launch's: the
    gradients of g_out * s are s times the gradients of g_out, to fp32 accuracy, for s far outside fp16's range"""
    case = _case("homo", 128, 97, 7, 4, None, 0.5, seed=5)
    _, g1 = _run(case, "f16x2")
    _, gs = _run(case, "f16x2", gscale=gscale)
    _, gb = _run(case, "bf16x3", gscale=gscale)
    for k in g1:
        n = g1[k].abs().max().item()
        assert not torch.isnan(gs[k]).any(), k
        # (atomics: two runs of the same arithmetic differ in the last bits)
        assert (gs[k] / gscale - g1[k]).abs().max().item() <= 2e-5 * max(n, 1e-30), k
        assert (gs[k] - gb[k]).abs().max().item() <= 2e-5 * gscale * max(n, 1e-30), k


def test_rows_of_very_different_magnitude_in_one_launch():
    """one masked node's upstream gradient is 1e6 times the others': its tiles get their own scale in the BPTT; in the
    weight-gradient GEMM the small rows sit 2^20 below the launch maximum and lose bits the result does not need"""
    W, L = 11, 4
    m, X, sel, ids, codes, ms, mc, G = _oracle_case("homo", 128, 70, W, L, seed=6)
    G = G.clone()
    G[3] *= 1e6
    m.seq_math = "f16x2"
    _against_fp64("homo", m, X, sel, ids, codes, ms, mc, G, W, L, floor=2e-6)


def test_all_zero_inputs_and_gradients():
    """maxima of zero: the scales fall back to 1, nothing divides by zero"""
    case = _case("homo", 128, 20, 5, 4, None, 0.0, seed=8)
    m, X, ids, codes, sel, G = case
    out, g = _run((m, torch.zeros_like(X), ids, codes, sel, torch.zeros_like(G)), "f16x2")
    assert torch.isfinite(out).all()
    for k in g:
        assert torch.isfinite(g[k]).all() and g[k].abs().max().item() == 0.0, k


def test_shape_info_reports_the_arithmetic(monkeypatch):
    from pathnet_amd import _lib, modules
    monkeypatch.delenv("PN_SEQ_MATH", raising=False)     # (the suite also runs with PN_SEQ_MATH=bf16x3: profiles/r04_pytest_gpu_bf16x3.txt)
    sh = modules._shape("homo", 300, 16, 128, 3, 10, 5, 4)
    assert modules.shape_info(sh)[3] == _lib.SEQ_MATH_F16X2
    sh = modules._shape("homo", 300, 16, 128, 3, 10, 5, 4, seq_math=_lib.SEQ_MATH_BF16X3)
    assert modules.shape_info(sh)[3] == _lib.SEQ_MATH_BF16X3
    sh = modules._shape("homo", 300, 16, 512, 3, 10, 5, 4)          # the step-by-step recurrence: bf16 x 3 GEMMs, no mode
    assert modules.shape_info(sh)[3] == 0
    sh = modules._shape("homo", 300, 16, 128, 3, 10, 5, 4, cell="mean")
    assert modules.shape_info(sh)[3] == 0


@pytest.mark.parametrize("variant,H,cell,L", [("homo", 128, None, 4), ("hetero", 128, None, 4), ("pagg", 128, None, 4),
                                              ("homo", 128, "gru", 4), ("homo", 64, None, 6), ("pagg", 96, "lstm", 4),
                                              ("hetero", 256, None, 4), ("homo", 32, "rnn", 1)])
def test_inference_forward_with_the_input_gates_applied_before_the_gather(variant, H, cell, L, monkeypatch):
    """eval-mode no-grad forwards run seq_fwdzw_kernel (ZW = Z . W_ih^T + b once per bank row, only the W_hh products per
    step; PathNet_run.py:359-362, :378): same logits as the plain inference path (PN_EVAL_ZW=0) and as the CPU oracle"""
    import pathnet_amd
    W = 9
    m, X, sel, ids, codes, ms, mc, G = _oracle_case(variant, H, 70, W, L, cell=cell)
    m.eval()
    S, N = len(sel), X.shape[0]
    mask = np.zeros(N, bool)
    mask[sel] = True
    args = (X.cuda(), torch.as_tensor(ids.reshape(S, W * L).astype(np.int64)), W, L, mask, torch.as_tensor(codes.astype(np.int64)), None)
    with torch.no_grad():
        from pathnet_amd import _lib
        _lib.set_knob("PN_EVAL_ZW", 0)          # (a knob of the context: the library never reads the environment in a call)
        try:
            plain = m(*args).clone()
        finally:
            _lib.set_knob("PN_EVAL_ZW", 1)
        got = m(*args).clone()
        again = m(*args, reuse_tables=True).clone()         # the test forward of an epoch: tables reused, ZW rebuilt
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    want = po.forward(variant, params, X, ids, codes, sel, W, L, cell=cell)
    assert (got - plain).abs().max().item() <= 2e-6
    assert torch.equal(got, again)
    assert (got.cpu() - want).abs().max().item() < 1e-5


@pytest.mark.parametrize("variant,H,S,W,L,cell,rows", [
    ("homo", 128, 97, 7, 4, None, 16), ("homo", 128, 97, 7, 4, None, 8), ("homo", 128, 97, 7, 4, None, 24),
    ("hetero", 128, 61, 9, 4, None, 16), ("pagg", 96, 50, 5, 4, None, 8), ("homo", 64, 40, 9, 6, "gru", 24),
    ("homo", 128, 700, 40, 4, None, 1),         # 28 000 paths = 875 tiles on 768 / 512 slots: a real remainder round, sized by the library
])
def test_remainder_round_tiles_match_the_default_tiling(variant, H, S, W, L, cell, rows):
    """pn_seqh.hip "tile geometry": the last round of the forward / BPTT launches cut into tiles of 8 / 16 / 24 paths (context
    knob PN_SEQH_TAIL; off by default -- measured neutral, profiles/r05_tune_scatter_tiling.txt).  A path's arithmetic does not
    depend on the tile it sits in: bit-equal logits, gradients equal up to the order of the scatter's atomics."""
    from pathnet_amd import _lib
    case = _case(variant, H, S, W, L, cell, 0.5, N=max(400, S + 50))
    old = _lib.set_knob("PN_SEQH_TAIL", 0)
    try:
        ref_out, ref_g = _run(case, "f16x2")
        _lib.set_knob("PN_SEQH_TAIL", rows)
        out, g = _run(case, "f16x2")
    finally:
        _lib.set_knob("PN_SEQH_TAIL", old)
    assert torch.equal(out, ref_out)
    assert_grads_close(g, ref_g, rel=1e-5, zero_ok=ZERO_OK_HETERO if variant == "hetero" else ())


def _cora_like(spike, seed=61, S=400):
    """configs[1]'s shapes (N = 2708, F = 1433, hid 128, W = 40, L = 4; S masked nodes): bag-of-words-like features, and
    with `spike` one node whose feature row is 2^spike times a typical one -- its projected / bank rows then sit that far
    above every other row of Z"""
    import pathnet_amd
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    N, F, H, C, W, L = 2708, 1433, 128, 7, 40, 4
    m = pathnet_amd.PathNet_homo(F, H, C, L, dropout=0.0).cuda().eval()
    X = (torch.rand(N, F) < 0.0127).float()
    X = X / X.sum(1, keepdim=True).clamp(min=1.0)
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    sel = np.flatnonzero(mask)
    hub = int(np.flatnonzero(~mask)[0])
    if spike:
        X[hub] *= float(2 ** spike)
    ids = rng.integers(0, N, (S, W, L))
    ids[:, :, 0] = sel[:, None]
    ids[::7, 0, 2] = hub                     # a few paths do pass through the spike node
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    return m, X, ids, codes, mask, sel, W, L, hub


def _fwd(m, X, ids, codes, mask, W, L):
    S = int(mask.sum())
    with torch.no_grad():
        return m(X.cuda(), torch.as_tensor(ids.reshape(S, W * L).astype(np.int64)), W, L, mask,
                 torch.as_tensor(codes.astype(np.int64)), None).cpu()


@pytest.mark.parametrize("spike", [22, 40])
def test_spike_row_beyond_the_fp16_window_falls_back_to_bf16x3(spike):
    """the fp16 x 2 split carries a 2^18 window below the launch-wide maximum of the gathered rows.  One
    node whose bank rows are 2^spike times the typical row (configuration size) takes the typical rows out of it.  The
    forward's kernels leave max |Z| and the spread of the rows' magnitudes in device memory (pn_seq_range); the module reads
    them after its first call, re-runs that call in bf16x3 and stays there.  Checked: the guard fires, the result it returns
    is within 1e-5 of the float64 oracle.  Measured with the same input forced through the fp16 kernels: at 2^22 the typical
    rows still carry ~17 bits and the logits stay within 1e-7 (the guard is conservative there); at 2^40 their planes are
    zero and the forced result is wrong -- the guard is what keeps the contract."""
    import warnings
    m, X, ids, codes, mask, sel, W, L, hub = _cora_like(spike=spike)
    params64 = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    want = po.forward("homo", params64, X.double(), ids, codes, sel, W, L, dtype=torch.float64).float()
    through_hub = np.zeros(len(sel), bool)
    through_hub[::7] = True                  # rows whose paths touch the spike node: huge pre-activations, saturated gates
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = _fwd(m, X, ids, codes, mask, W, L)
    assert m._range_wide and m.range_spread_bits > 18 and any("bf16x3" in str(x.message) for x in w)
    err_guarded = (out - want).abs().max().item()
    assert err_guarded < 1e-5, err_guarded
    # the same call forced through the fp16 kernels: the rows that never see the spike lose bits to its scale
    m2 = _cora_like(spike=spike)[0]
    m2.load_state_dict(m.state_dict())
    m2.seq_math = "f16x2"
    out16 = _fwd(m2, X, ids, codes, mask, W, L)
    err16 = (out16 - want).abs()[~through_hub].max().item()
    print("spike 2^%d: guarded (bf16x3) error %.2e, forced f16x2 error on rows away from the spike %.2e, spread %.1f bits"
          % (spike, err_guarded, err16, m.range_spread_bits))
    if spike >= 40:
        assert err16 > 1e-5 > err_guarded
    # and a second call of the guarded module goes straight to bf16x3 (no read-back, same values)
    assert torch.equal(_fwd(m, X, ids, codes, mask, W, L), out)


def test_ordinary_features_stay_on_the_fp16_kernels():
    import warnings
    m, X, ids, codes, mask, sel, W, L, _ = _cora_like(spike=0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = _fwd(m, X, ids, codes, mask, W, L)
    assert not m._range_wide and m.range_spread_bits is not None and m.range_spread_bits < 12 and not w
    params64 = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    want = po.forward("homo", params64, X.double(), ids, codes, sel, W, L, dtype=torch.float64).float()
    assert (out - want).abs().max().item() < 1e-5
    for _ in range(20):         # past the next ASYNCHRONOUS read-back of the record (every 16th call)
        _fwd(m, X, ids, codes, mask, W, L)
    torch.cuda.synchronize()
    m._range_poll()
    assert not m._range_wide and m.range_spread_bits < 12

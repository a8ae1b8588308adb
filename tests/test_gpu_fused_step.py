"""pn_pagg_train_step (module.forward_loss): forward + torch.nn.CrossEntropyLoss() + backward of a training step
(PathNet_run.py:343-351) in one library call.  Same kernels in the same order as forward / pn_cross_entropy / backward,
so in deterministic mode loss, logits and every gradient are BITWISE those of the three separate calls -- in one batch and
in micro-batches, where the fused call saves each micro-batch's second forward; and it matches the CPU oracle."""
import numpy as np
import pytest
import torch
from gradcheck import ZERO_OK_HETERO, assert_grads_close

from oracle import pagg_oracle as po

pytestmark = pytest.mark.gpu


def _case(variant, S, W, L, H=128, cell=None, N=900, F=40, C=5, seed=0):
    import pathnet_amd
    g = torch.Generator().manual_seed(seed)
    cls = {"homo": pathnet_amd.PathNet_homo, "hetero": pathnet_amd.PathNet, "pagg": pathnet_amd.PAGG}[variant]
    torch.manual_seed(seed)
    kw = {} if variant == "pagg" else {"cell": cell}
    m = cls(F, H, C, L, dropout=0.5, **kw).cuda().train()
    m.deterministic = True
    X = torch.rand(N, F, generator=g).cuda()
    sel = torch.randperm(N, generator=g)[:S].sort().values.to(torch.int32)
    ids = torch.randint(0, N, (S, W, L), generator=g).to(torch.int32)
    ids[:, :, 0] = sel[:, None]
    codes = torch.randint(0, L, (S, W, L), generator=g).to(torch.uint8)
    y = torch.randint(0, C, (S,), generator=g).cuda()
    return m, X, ids.cuda(), codes.cuda(), sel.cuda(), y


def _separate(case, seed=3):
    from pathnet_amd import optim
    m, X, ids, codes, sel, y = case
    torch.manual_seed(seed)
    m.zero_grad(set_to_none=True)
    out = m(X, ids, ids.shape[1], ids.shape[2], sel, codes, None)
    loss = optim.CrossEntropyLoss()(out, y)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), out.detach().clone(), {k: v.grad.clone() for k, v in m.named_parameters()}


def _fused(case, seed=3):
    m, X, ids, codes, sel, y = case
    torch.manual_seed(seed)
    m.zero_grad(set_to_none=True)
    loss, out = m.forward_loss(X, ids, ids.shape[1], ids.shape[2], sel, codes, y, fused=True)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), out.detach().clone(), {k: v.grad.clone() for k, v in m.named_parameters()}


@pytest.mark.parametrize("variant,S,W,L,H,cell,micro", [
    ("homo", 300, 12, 4, 128, None, 0),
    ("homo", 300, 12, 4, 128, None, 70),        # five micro-batches, the last one ragged
    ("hetero", 200, 9, 4, 128, None, 0),
    ("hetero", 200, 9, 4, 128, None, 64),
    ("pagg", 150, 8, 4, 64, None, 40),
    ("homo", 100, 6, 3, 288, "gru", 30),        # generic recurrence
    ("homo", 100, 6, 5, 100, "mean", 0),        # zero-padded hidden size, order-agnostic encoder
])
def test_fused_step_is_the_three_calls(variant, S, W, L, H, cell, micro):
    from pathnet_amd import modules as M
    case = _case(variant, S, W, L, H=H, cell=cell)
    m = case[0]
    if micro:
        Hk = -(-H // 32) * 32
        kw = dict(cell=m._cell_kind, deterministic=True)
        # the workspace of this call when it runs b masked nodes at a time is fixed(S) + b * per (pick_batch_groups)
        m.workspace_budget = M.workspace_bytes(variant, 900, 40, Hk, 5, S, W, L, batch_groups=micro, **kw)
        assert M.pick_batch_groups(variant, 900, 40, Hk, 5, S, W, L, m.workspace_budget, **kw) == micro
    l0, o0, g0 = _separate(case)
    l1, o1, g1 = _fused(case)
    assert torch.equal(o0, o1)
    assert abs(l0.item() - l1.item()) <= 1e-6 * max(1.0, abs(l0.item()))     # (micro-batches add their parts one after the other)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


def test_fused_step_matches_the_oracle_and_scales_with_the_upstream_gradient():
    import pathnet_amd
    torch.manual_seed(1)
    N, F, H, C, S, W, L = 300, 40, 128, 4, 70, 11, 4
    g = torch.Generator().manual_seed(5)
    m = pathnet_amd.PathNet_homo(F, H, C, L, dropout=0.5).cuda().train()
    X = torch.rand(N, F, generator=g)
    sel = np.sort(np.random.default_rng(2).choice(N, S, replace=False))
    ids = np.random.default_rng(3).integers(0, N, (S, W, L)).astype(np.int32)
    ids[:, :, 0] = sel[:, None]
    codes = np.random.default_rng(4).integers(0, L, (S, W, L)).astype(np.uint8)
    y = torch.as_tensor(np.random.default_rng(6).integers(0, C, S))
    keep = 0.5
    mask_seq = (torch.rand(L, S * W, H, generator=g) < keep).float() / keep
    mask_cls = (torch.rand(S, 2 * H, generator=g) < keep).float() / keep
    m._mask_seq, m._mask_cls = mask_seq.cuda(), mask_cls.cuda()
    mask = np.zeros(N, bool)
    mask[sel] = True
    loss, out = m.forward_loss(X.cuda(), torch.as_tensor(ids.reshape(S, W * L).astype(np.int64)), W, L, mask,
                               torch.as_tensor(codes.astype(np.int64)), y, fused=True)
    (2.5 * loss).backward()
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    want = po.forward("homo", params, X, ids, codes, sel, W, L, drop_seq=mask_seq, drop_cls=mask_cls)
    wl = torch.nn.functional.cross_entropy(want, y)
    assert (out.detach().cpu() - want.detach()).abs().max().item() < 1e-5
    assert abs(loss.item() - wl.item()) < 1e-5
    (2.5 * wl).backward()
    assert_grads_close({k: v.grad for k, v in m.named_parameters()}, {k: params[k].grad for k, _ in m.named_parameters()})


def test_forward_loss_picks_the_fused_call_exactly_when_the_batch_needs_micro_batches():
    from pathnet_amd import modules as M
    case = _case("homo", 120, 6, 4)
    m, X, ids, codes, sel, y = case
    torch.manual_seed(3)
    loss, out = m.forward_loss(X, ids, 6, 4, sel, codes, y)
    assert loss.grad_fn is not None and type(loss.grad_fn).__name__ != "_PaggLossFunctionBackward"
    kw = dict(cell=m._cell_kind, deterministic=True)
    fixed = M.workspace_bytes("homo", 900, 40, 128, 5, 1, 6, 4, **kw)
    per = (M.workspace_bytes("homo", 900, 40, 128, 5, 1025, 6, 4, **kw) - fixed) // 1024
    m.workspace_budget = fixed + 50 * per
    torch.manual_seed(3)
    loss2, out2 = m.forward_loss(X, ids, 6, 4, sel, codes, y)
    assert type(loss2.grad_fn).__name__ == "_PaggLossFunctionBackward"
    assert torch.equal(out.detach(), out2) and abs(loss.item() - loss2.item()) < 1e-6


def test_fused_step_needs_grad_mode_and_matching_targets():
    case = _case("homo", 20, 4, 4)
    m, X, ids, codes, sel, y = case
    with torch.no_grad(), pytest.raises(RuntimeError):
        m.forward_loss(X, ids, 4, 4, sel, codes, y, fused=True)
    with pytest.raises(ValueError):
        m.forward_loss(X, ids, 4, 4, sel, codes, y[:-1])


@pytest.mark.parametrize("variant,S,W,L,H,cell,micro", [
    ("homo", 300, 40, 4, 128, None, 0),         # the headline cell and path count per node
    ("homo", 300, 12, 4, 128, None, 70),        # micro-batches: the loss accumulates over five launches
    ("hetero", 200, 9, 4, 128, None, 0),        # softmax attention, a row per member in the ego scatter
    ("pagg", 150, 8, 4, 64, None, 0),           # no attention: the backward body returns early
    ("homo", 100, 6, 3, 288, "gru", 0),         # H > 256: the 16-chunk instantiation
    ("homo", 100, 6, 5, 100, "mean", 0),        # zero-padded hidden size
])
def test_pooling_step_in_one_launch_is_the_three_launches(variant, S, W, L, H, cell, micro):
    """Round 6: in the default (atomic) mode pn_pagg_train_step runs pooling forward, cross entropy and pooling backward of
    a node as ONE launch (pool_step_kernel: the two kernels' bodies back to back in one workgroup, the node's cross entropy
    in between; context knob PN_POOL_STEP = 2).  Same code, same order inside a node: logits and loss bit-equal to the
    three launches and to the three library calls; gradients equal up to the order of the float atomics (which differs
    from run to run anyway).  PN_POOL_STEP = 1 (default) takes pool_step2_kernel where the shape allows -- the node's rows
    held in registers, other summation orders: held to 2e-6 on logits / loss and to the gradient bound."""
    from pathnet_amd import _lib
    from pathnet_amd import modules as M
    case = _case(variant, S, W, L, H=H, cell=cell)
    m = case[0]
    m.deterministic = False
    if micro:
        Hk = -(-H // 32) * 32
        kw = dict(cell=m._cell_kind, deterministic=False)
        m.workspace_budget = M.workspace_bytes(variant, 900, 40, Hk, 5, S, W, L, batch_groups=micro, **kw)
        assert M.pick_batch_groups(variant, 900, 40, Hk, 5, S, W, L, m.workspace_budget, **kw) == micro
    old_step = _lib.set_knob("PN_POOL_STEP", 0)
    try:
        l0, o0, g0 = _separate(case)                # three library calls
        l1, o1, g1 = _fused(case)                   # the same three launches inside the fused call
        _lib.set_knob("PN_POOL_STEP", 2)
        l2, o2, g2 = _fused(case)                   # one launch: the two kernels' bodies back to back, the loss in between
        _lib.set_knob("PN_POOL_STEP", 1)
        l3, o3, g3 = _fused(case)                   # default: the node's rows in registers where the shape allows (W <= 40, H <= 128)
    finally:
        _lib.set_knob("PN_POOL_STEP", old_step)
    assert torch.equal(o0, o1) and torch.equal(o1, o2)
    if micro:
        assert abs(l1.item() - l2.item()) <= 1e-6 * max(1.0, abs(l1.item()))
    else:
        assert l0.item() == l1.item() == l2.item()
    # pool_step2_kernel sums in another order: logits within 2e-6 of the three launches' (fp32 rounding of sums over W and H)
    assert (o3 - o0).abs().max().item() <= 2e-6 * max(1.0, o0.abs().max().item())
    assert abs(l3.item() - l0.item()) <= 2e-6 * max(1.0, abs(l0.item()))
    zero_ok = ZERO_OK_HETERO if variant == "hetero" else ()
    # (noise: the attention bias' gradient is a cancelling sum of S * W terms a thousand times its size, added by atomics in
    #  a different order every run -- two runs of the SAME configuration differ by as much)
    for g in (g1, g2, g3):
        assert_grads_close(g, g0, rel=1e-5, noise=3e-8, zero_ok=zero_ok)


def test_sampler_on_the_stream_that_reads_the_paths_gives_the_same_steps():
    """module.paths_stream() (pn_pagg_paths_stream): the training loop enqueues each step's walk on the stream where the
    aggregator reads the path arrays -- the library's second stream -- instead of the main one: no event on the main stream, the
    walk runs under the tail of the previous step.  Twelve steps of sample -> fused step -> Adam, host racing ahead of the GPU:
    the same losses and parameters, bit for bit (deterministic backward), as with the sampler on the main stream."""
    import numpy as np
    import pathnet_amd
    rng = np.random.default_rng(3)
    n, F, H, C, W, L, S = 600, 40, 128, 5, 8, 4, 200
    u = rng.integers(0, n, 3000)
    v = rng.integers(0, n, 3000)
    keep = u != v
    u, v = np.concatenate([u[keep], np.arange(n)]), np.concatenate([v[keep], (np.arange(n) + 1) % n])
    p = rng.random(len(u))
    smp = pathnet_amd.MerwSampler(n, u.astype(np.int32), v.astype(np.int32), p, L, device="cuda")
    X = torch.rand(n, F, device="cuda")
    sel = torch.arange(0, n, 3, dtype=torch.int32, device="cuda")[:S]
    y = torch.randint(0, C, (S,), device="cuda")
    runs = []
    for beside in (False, True):
        torch.manual_seed(1)
        m = pathnet_amd.PathNet_homo(F, H, C, L, dropout=0.5).cuda().train()
        m.deterministic = True
        opt = pathnet_amd.Adam(m.parameters(), lr=0.01)
        ids = torch.empty((1, S, W, L), dtype=torch.int32, device="cuda")
        codes = torch.empty((1, S, W, L), dtype=torch.uint8, device="cuda")
        losses, streams = [], set()
        for e in range(12):
            st = m.paths_stream() if beside else torch.cuda.current_stream()
            streams.add(st.cuda_stream)
            with torch.cuda.stream(st):
                smp.sample(W, 77, epoch_begin=e, epoch_count=1, nodes=sel, draw_source=pathnet_amd.DRAW_PHILOX, check=False,
                           out=(ids, codes))
            opt.zero_grad(set_to_none=True)
            torch.manual_seed(100 + e)          # (the dropout seed of the step)
            loss, _ = m.forward_loss(X, ids[0], W, L, sel, codes[0], y, fused=True)
            pathnet_amd.backward(loss)
            opt.step()
            losses.append(loss.detach())
        torch.cuda.synchronize()
        runs.append((torch.stack(losses).cpu(), [q.detach().clone() for q in m.parameters()], streams))
    assert len(runs[1][2]) == 2         # the first step's walk on the main stream, every later one on the library's second stream
    assert torch.equal(runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)


def test_steps_are_the_same_with_either_kind_of_fork_and_join_event():
    """The context's fork / join events order kernels of one device and are created without the system-scope release of a
    recorded event (hipEventDisableSystemFence; knob PN_EVENT_DEVICE_SCOPE, default 1, profiles/r06_glue.txt section 22).
    Eight training steps with the host racing ahead -- fused step, both streams at work, Adam -- give the same losses and
    parameters bit for bit (deterministic backward) with either kind of event; changing the knob makes the events again."""
    import pathnet_amd
    from pathnet_amd import _lib
    case = _case("homo", 300, 12, 4)
    _, X, ids, codes, sel, y = case
    old = _lib.set_knob("PN_EVENT_DEVICE_SCOPE", 1)
    runs = []
    try:
        for dev_scope in (1, 0, 1):
            _lib.set_knob("PN_EVENT_DEVICE_SCOPE", dev_scope)
            torch.manual_seed(5)
            m = pathnet_amd.PathNet_homo(X.shape[1], 128, 5, ids.shape[2], dropout=0.5).cuda().train()
            m.deterministic = True
            opt = pathnet_amd.Adam(m.parameters(), lr=0.01)
            losses = []
            for e in range(8):
                opt.zero_grad(set_to_none=True)
                torch.manual_seed(200 + e)
                loss, _ = m.forward_loss(X, ids, ids.shape[1], ids.shape[2], sel, codes, y, fused=True)
                pathnet_amd.backward(loss)
                opt.step()
                losses.append(loss.detach())
            torch.cuda.synchronize()
            runs.append((torch.stack(losses).cpu(), [q.detach().clone() for q in m.parameters()]))
    finally:
        _lib.set_knob("PN_EVENT_DEVICE_SCOPE", old)
    for other in runs[1:]:
        assert torch.equal(runs[0][0], other[0])
        for a, b in zip(runs[0][1], other[1]):
            assert torch.equal(a, b)

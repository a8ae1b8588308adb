"""world_size-2 gloo test of the node-sharded path (pathnet_amd/dist.py) on CPU.

The collectives and the sharding logic are the product's; the arithmetic is supplied by a checker
backend built on the oracle (there is no CPU implementation in the product, by design).  Two ranks,
each owning half of the node rows, must reproduce the single-process result: logits of every masked
node, and -- after the flat all-reduce -- the gradient of every parameter."""
import os
import socket

import numpy as np
import pytest
import torch
from gradcheck import ZERO_OK_HETERO, assert_grads_close
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pagg_oracle as po


class OracleOps:
    """Checker backend with the HipOps interface (project / forward / backward / linear_backward)."""
    KEYMAP = None

    def __init__(self, variant, L):
        self.variant, self.L = variant, L

    def _names(self):
        if self.variant == "pagg":
            bank = ["nei%d" % d for d in range(4)]
            cell = "RNN"
        else:
            bank = ["nets.%d" % d for d in range(self.L)]
            cell = "LSTM"
        return bank, cell

    def project(self, variant, X_loc, w, b):
        return po.project(variant, w.detach(), b.detach(), X_loc)

    def forward(self, cfg, Xh, ids, codes, sel, p):
        with torch.enable_grad():       # autograd.Function.forward runs with grad disabled
            return self._forward(cfg, Xh, ids, codes, sel, p)

    def _forward(self, cfg, Xh, ids, codes, sel, p):
        bank, cell = self._names()
        leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()
                if k not in ("bank_ws", "bank_bs") and v is not None}
        leaf["bank_w"] = cfg["bank_w"].detach().clone().requires_grad_(True)
        leaf["bank_b"] = cfg["bank_b"].detach().clone().requires_grad_(True)
        params = {"fc2.weight": leaf["fc2_w"], "fc2.bias": leaf["fc2_b"],
                  cell + ".weight_ih_l0": leaf["w_ih"], cell + ".weight_hh_l0": leaf["w_hh"],
                  cell + ".bias_ih_l0": leaf["b_ih"], cell + ".bias_hh_l0": leaf["b_hh"]}
        for d, name in enumerate(bank):
            params[name + ".weight"] = leaf["bank_w"][d]
            params[name + ".bias"] = leaf["bank_b"][d]
        if "att_w" in leaf:
            params["attw.weight"] = leaf["att_w"].reshape(1, -1)
            params["attw.bias"] = leaf["att_b"]
        Xh = Xh.detach().clone().requires_grad_(True)
        # a rank computes the pooling groups [group_begin, +S) of the step's batch (pn_pagg_shape.S_total / group_begin).
        # homo / PAGG ranks hold their own rows only (index_rows_local); a hetero rank holds the whole batch's index
        # arrays: by definition its rows are those rows of the whole-batch forward
        drop_seq, drop_cls = cfg.get("mask_seq"), cfg.get("mask_cls")
        W = cfg["W"]
        S, S_total, begin = cfg["S"], cfg.get("S_total") or cfg["S"], cfg.get("group_begin", 0)
        if cfg.get("index_rows_local") and S_total != S:
            assert cfg["variant"] != "hetero" and len(sel) == S
            if drop_seq is not None:
                drop_seq = drop_seq[:, begin * W:(begin + S) * W]
            if drop_cls is not None:
                drop_cls = drop_cls[begin:begin + S]
            out = po.forward(cfg["variant"], params, None, ids.numpy(), codes.numpy(), sel.numpy(), W, cfg["L"],
                             Xh=Xh, drop_seq=drop_seq, drop_cls=drop_cls)
        else:
            assert len(sel) == S_total
            out = po.forward(cfg["variant"], params, None, ids.numpy(), codes.numpy(), sel.numpy(), W, cfg["L"],
                             Xh=Xh, drop_seq=drop_seq, drop_cls=drop_cls)[begin:begin + S]
        return out.detach(), (out, Xh, leaf)

    def backward(self, state, g_out):
        out, Xh, leaf = state
        keys = [k for k in leaf if k not in ("fc0_w", "fc0_b")]
        gs = torch.autograd.grad(out, [Xh] + [leaf[k] for k in keys], g_out, allow_unused=True)
        return gs[0], {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(keys, gs[1:])}

    def linear_backward(self, variant, dXh_loc, Xh_loc, X_loc, w, deterministic=False):
        g = dXh_loc * (Xh_loc > 0).float() if variant == "homo" else dXh_loc     # ReLU backward, PathNet_run.py:243
        return g.t() @ X_loc, g.sum(0)


def make_case(variant, seed=0, uneven=False, N=24, H=32, empty_block=None):
    rng = np.random.default_rng(seed)
    F, C, W, L = 10, 3, 5, 4
    X = torch.as_tensor(rng.random((N, F), dtype=np.float32))
    mask = np.zeros(N, bool)
    if uneven:                      # 9 masked nodes in the first row block, 3 in the second
        mask[rng.permutation(N // 2)[:9]] = True
        mask[N // 2 + rng.permutation(N // 2)[:3]] = True
    elif empty_block is not None:   # ragged counts over `world` blocks, none at all in block `empty_block[1]`
        from pathnet_amd.dist import node_block
        world, empty = empty_block
        for r in range(world):
            lo, cnt = node_block(N, world, r)
            take = 0 if r == empty else min(cnt, 2 + 3 * r)
            mask[lo + rng.permutation(cnt)[:take]] = True
    else:
        mask[rng.permutation(N)[:14]] = True
    sel = np.flatnonzero(mask)
    ids = rng.integers(0, N, (len(sel), W, L))
    ids[:, :, 0] = sel[:, None]
    codes = np.minimum(rng.integers(0, L, (len(sel), W, L)), np.arange(L)[None, None, :])
    G = torch.as_tensor(rng.standard_normal((len(sel), C)).astype(np.float32))
    Y = torch.as_tensor(rng.integers(0, C, len(sel)))
    return dict(N=N, F=F, H=H, C=C, W=W, L=L, X=X, sel=sel, ids=ids, codes=codes, G=G, Y=Y)


def build_module(variant, case):
    import pathnet_amd
    torch.manual_seed(123)
    cls = {"homo": pathnet_amd.PathNet_homo, "pagg": pathnet_amd.PAGG, "hetero": pathnet_amd.PathNet}[variant]
    m = cls(case["F"], case["H"], case["C"], case["L"] if variant != "pagg" else case["N"])
    return m.eval()


def worker(rank, world, port, variant, ret, mode="sum", case_kw=None, exchange="auto"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pathnet_amd import dist as pdist
        case = make_case(variant, uneven=(mode == "mean_uneven"), **(case_kw or {}))
        m = build_module(variant, case)
        lo, n_loc = pdist.node_block(case["N"], world, rank)
        mine = (case["sel"] >= lo) & (case["sel"] < lo + n_loc)
        runner = pdist.ShardedAggregator(m, case["N"], lo, n_loc, ops=OracleOps(variant, case["L"]), exchange=exchange)
        if mode == "masks":             # training mode, explicit masks of the WHOLE batch
            S, W, H = len(case["sel"]), case["W"], case["H"]
            g = torch.Generator().manual_seed(9)
            runner.mask_seq = (torch.rand(case["L"], S * W, H, generator=g) >= 0.5).float() / 0.5
            runner.mask_cls = (torch.rand(S, 2 * H, generator=g) >= 0.5).float() / 0.5
            m.train()
        # the sel argument as a NUMPY integer array of node ids (round 1 mistook that for a bool mask)
        out = runner(case["X"][lo:lo + n_loc], torch.as_tensor(case["ids"][mine].reshape(int(mine.sum()), case["W"] * case["L"])), case["W"],
                     case["L"], case["sel"][mine].astype(np.int64), torch.as_tensor(case["codes"][mine]))
        blocks = [pdist.node_block(case["N"], world, r) for r in range(world)]
        assert runner.batch_counts == [int(((case["sel"] >= b) & (case["sel"] < b + c)).sum()) for b, c in blocks]
        assert runner.last_exchange == (exchange if exchange != "auto" else runner.last_exchange)
        if mode == "mean_uneven":
            # every rank's loss is the MEAN over its own masked nodes (PathNet_run.py:346); scaled by S_r / S_total the
            # summed gradients are those of the mean over the whole batch
            loss = torch.nn.functional.cross_entropy(out, case["Y"][mine]) * runner.loss_scale()
            loss.backward()
        else:
            (out * case["G"][mine]).sum().backward()
        runner.allreduce_grads(average=False)
        flat = runner._flat
        assert all(v.grad.data_ptr() >= flat.data_ptr() and
                   v.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4 for v in m.parameters())   # views of one buffer
        ret[rank] = (out.detach().numpy(), {k: v.grad.numpy().copy() for k, v in m.named_parameters()},
                     np.flatnonzero(mine), runner.last_exchange)
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("variant,mode", [("homo", "sum"), ("pagg", "sum"), ("hetero", "sum"), ("hetero", "masks"),
                                          ("homo", "masks"), ("homo", "mean_uneven"), ("hetero", "mean_uneven")])
def test_two_rank_sharding_matches_single_process(variant, mode):
    """Two ranks = one process on the concatenated batch -- for the hetero class too: its [W, S] re-view
    (PathNet_run.py:196-197) makes a rank's rows read the other rank's paths, which is why the ranks exchange the
    batch's index arrays and compute slices of the WHOLE batch (pn_pagg_shape.S_total / group_begin)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), variant, ret, mode), nprocs=world, join=True)
    case = make_case(variant, uneven=(mode == "mean_uneven"))
    m = build_module(variant, case)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    drop_seq = drop_cls = None
    if mode == "masks":
        S, W, H = len(case["sel"]), case["W"], case["H"]
        g = torch.Generator().manual_seed(9)
        drop_seq = (torch.rand(case["L"], S * W, H, generator=g) >= 0.5).float() / 0.5
        drop_cls = (torch.rand(S, 2 * H, generator=g) >= 0.5).float() / 0.5
    want = po.forward(variant, params, case["X"], case["ids"], case["codes"], case["sel"], case["W"], case["L"],
                      drop_seq=drop_seq, drop_cls=drop_cls)
    if mode == "mean_uneven":
        torch.nn.functional.cross_entropy(want, case["Y"]).backward()
    else:
        (want * case["G"]).sum().backward()
    for rank in range(world):
        out, grads, rows, _ = ret[rank]
        assert np.abs(out - want.detach().numpy()[rows]).max() < 1e-5
        assert_grads_close(grads, {k: params[k].grad for k in grads}, rel=2e-5, zero_ok=ZERO_OK_HETERO, tag="rank %d" % rank)


@pytest.mark.parametrize("variant,world,exchange,N,H", [
    ("homo", 4, "sparse", 26, 32),      # 26 nodes on 4 ranks: blocks of 7, 7, 7, 5 -- no multiple of the world size
    ("homo", 4, "dense", 26, 32),
    ("hetero", 4, "sparse", 26, 32),    # the whole batch's index arrays on every rank, rows fetched for all of them
    ("pagg", 4, "auto", 26, 32),
    ("homo", 3, "sparse", 25, 20),      # hidden size 20: zero-padded to the kernels' 32 through differentiable pads
    ("hetero", 2, "dense", 25, 20),
    ("homo", 5, "sparse", 13, 32),      # 13 nodes on 5 ranks: blocks 3, 3, 3, 3, 1
    ("homo", 8, "dense", 13, 32),       # ... on 8: blocks of 2, the last one EMPTY (13 = 6 x 2 + 1 + 0)
    ("hetero", 8, "sparse", 13, 32),
])
def test_ragged_world_sizes_empty_ranks_and_both_exchange_modes(variant, world, exchange, N, H):
    """World sizes up to 8 with ragged masked-node counts, one rank without any masked node, node counts the world size
    does not divide (node_block: the last block shorter, or empty), a hidden size that is not a multiple of 32, and both
    ways of moving Xh / d Xh: all of them the single-process result."""
    mgr = mp.Manager()
    ret = mgr.dict()
    kw = dict(N=N, H=H, empty_block=(world, 1), seed=5)
    mp.spawn(worker, args=(world, free_port(), variant, ret, "sum", kw, exchange), nprocs=world, join=True)
    case = make_case(variant, **kw)
    m = build_module(variant, case)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    want = po.forward(variant, params, case["X"], case["ids"], case["codes"], case["sel"], case["W"], case["L"])
    (want * case["G"]).sum().backward()
    seen = 0
    for rank in range(world):
        out, grads, rows, used = ret[rank]
        seen += len(rows)
        assert used == (exchange if exchange != "auto" else used) and used in ("dense", "sparse")
        if len(rows):
            assert np.abs(out - want.detach().numpy()[rows]).max() < 1e-5
        assert_grads_close(grads, {k: params[k].grad for k in grads}, rel=2e-5, zero_ok=ZERO_OK_HETERO, tag="rank %d" % rank)
    assert seen == len(case["sel"]) and len(ret[1][2]) == 0         # rank 1 had no masked node
    assert all(ret[r][3] == ret[0][3] for r in range(world))        # one mode per step, on every rank


def test_node_blocks_cover_the_graph():
    from pathnet_amd.dist import node_block
    for n in (1, 7, 13, 24, 2708, 10 ** 7 + 3):
        for world in (1, 2, 3, 5, 8):
            at = 0
            sizes = []
            for r in range(world):
                lo, cnt = node_block(n, world, r)
                assert lo == at and cnt >= 0
                at += cnt
                sizes.append(cnt)
            assert at == n and max(sizes) == -(-n // world) and sorted(sizes, reverse=True) == sizes


def test_single_process_runner_without_process_group():
    """ShardedAggregator degenerates to the plain module when no process group exists."""
    from pathnet_amd import dist as pdist
    case = make_case("homo", seed=3)
    m = build_module("homo", case)
    runner = pdist.ShardedAggregator(m, case["N"], 0, case["N"], ops=OracleOps("homo", case["L"]))
    out = runner(case["X"], torch.as_tensor(case["ids"].reshape(len(case["sel"]), -1)), case["W"], case["L"],
                 torch.as_tensor(case["sel"].astype(np.int32)), torch.as_tensor(case["codes"]))
    want = po.forward("homo", dict(m.state_dict()), case["X"], case["ids"], case["codes"], case["sel"], case["W"],
                      case["L"])
    assert (out - want).abs().max().item() < 1e-5
    with pytest.raises(ValueError):
        pdist.ShardedAggregator(m, case["N"] + 1, 0, case["N"], ops=OracleOps("homo", case["L"]))     # not this rank's block


def test_index_argument_conventions():
    """`indices` is a bool mask (numpy or torch: the reference's two conventions) or the node ids
    themselves in any integer dtype -- a numpy integer array is NOT a mask."""
    from pathnet_amd import modules as M
    N, W, L = 10, 2, 3
    ids_all = np.arange(N)
    mask = np.zeros(N, bool)
    mask[[0, 3, 7]] = True
    neis = np.zeros((3, W * L), np.int64)
    lt = np.zeros((3, W, L), np.int64)
    for idx in (mask, torch.as_tensor(mask), np.array([0, 3, 7]), np.array([0, 3, 7], np.int32),
                torch.tensor([0, 3, 7]), torch.tensor([0, 3, 7], dtype=torch.int32)):
        ids, codes, sel, S = M._as_index_tensors(neis, lt, idx, W, L, "cpu", n_nodes=N)
        assert S == 3 and sel.dtype == torch.int32 and sel.tolist() == [0, 3, 7], type(idx)
        assert ids.shape == (3, W, L) and ids.dtype == torch.int32 and codes.dtype == torch.uint8
    with pytest.raises(IndexError):
        M._as_index_tensors(neis, lt, np.array([0, 3, 10]), W, L, "cpu", n_nodes=N)
    with pytest.raises(ValueError):
        M._as_index_tensors(neis[:2], lt, np.array([0, 3, 7]), W, L, "cpu", n_nodes=N)
    with pytest.raises(TypeError):
        M._as_index_tensors(neis, lt, np.array([0.0, 3.0, 7.0]), W, L, "cpu", n_nodes=N)
    assert ids_all.sum() == 45

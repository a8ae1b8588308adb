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
        out = po.forward(cfg["variant"], params, None, ids.numpy(), codes.numpy(), sel.numpy(), cfg["W"], cfg["L"],
                         Xh=Xh)
        return out.detach(), (out, Xh, leaf)

    def backward(self, state, g_out):
        out, Xh, leaf = state
        keys = [k for k in leaf if k not in ("fc0_w", "fc0_b")]
        gs = torch.autograd.grad(out, [Xh] + [leaf[k] for k in keys], g_out, allow_unused=True)
        return gs[0], {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(keys, gs[1:])}

    def linear_backward(self, variant, dXh_loc, Xh_loc, X_loc, w):
        g = dXh_loc * (Xh_loc > 0).float() if variant == "homo" else dXh_loc     # ReLU backward, PathNet_run.py:243
        return g.t() @ X_loc, g.sum(0)


def make_case(variant, seed=0):
    rng = np.random.default_rng(seed)
    N, F, H, C, W, L = 24, 10, 32, 3, 5, 4
    X = torch.as_tensor(rng.random((N, F), dtype=np.float32))
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:14]] = True
    sel = np.flatnonzero(mask)
    ids = rng.integers(0, N, (len(sel), W, L))
    ids[:, :, 0] = sel[:, None]
    codes = np.minimum(rng.integers(0, L, (len(sel), W, L)), np.arange(L)[None, None, :])
    G = torch.as_tensor(rng.standard_normal((len(sel), C)).astype(np.float32))
    return dict(N=N, F=F, H=H, C=C, W=W, L=L, X=X, sel=sel, ids=ids, codes=codes, G=G)


def build_module(variant, case):
    import pathnet_amd
    torch.manual_seed(123)
    cls = {"homo": pathnet_amd.PathNet_homo, "pagg": pathnet_amd.PAGG, "hetero": pathnet_amd.PathNet}[variant]
    m = cls(case["F"], case["H"], case["C"], case["L"] if variant != "pagg" else case["N"])
    return m.eval()


def worker(rank, world, port, variant, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pathnet_amd import dist as pdist
        case = make_case(variant)
        m = build_module(variant, case)
        n_loc = case["N"] // world
        lo = rank * n_loc
        mine = (case["sel"] >= lo) & (case["sel"] < lo + n_loc)
        runner = pdist.ShardedAggregator(m, case["N"], lo, n_loc, ops=OracleOps(variant, case["L"]))
        out = runner(case["X"][lo:lo + n_loc], torch.as_tensor(case["ids"][mine].reshape(mine.sum(), -1)), case["W"],
                     case["L"], torch.as_tensor(case["sel"][mine].astype(np.int32)),
                     torch.as_tensor(case["codes"][mine]))
        (out * case["G"][mine]).sum().backward()
        runner.allreduce_grads(average=False)
        ret[rank] = (out.detach().numpy(), {k: v.grad.numpy().copy() for k, v in m.named_parameters()},
                     np.flatnonzero(mine))
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("variant", ["homo", "pagg"])
def test_two_rank_sharding_matches_single_process(variant):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), variant, ret), nprocs=world, join=True)
    case = make_case(variant)
    m = build_module(variant, case)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    want = po.forward(variant, params, case["X"], case["ids"], case["codes"], case["sel"], case["W"], case["L"])
    (want * case["G"]).sum().backward()
    for rank in range(world):
        out, grads, rows = ret[rank]
        assert np.abs(out - want.detach().numpy()[rows]).max() < 1e-5
        for k, g in grads.items():
            ref = params[k].grad.numpy()
            assert np.abs(g - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), (rank, k)


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
        pdist.ShardedAggregator(m, case["N"] + 1, 0, case["N"], ops=OracleOps("homo", case["L"]))

"""The node-sharded path with the PRODUCT arithmetic at world size 2 on one GPU: two processes share cuda:0, the
collectives of pathnet_amd/dist.py run over gloo with the device tensors staged through the host (dist.Comm does that for
a host-memory backend; RCCL refuses two ranks on one device), the kernels are libpathnet_hip.so's (HipOps).  Expected: the single-process HIP module on the
concatenated batch -- logits of every masked node and, after the flat all-reduce, every parameter gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gradcheck import ZERO_OK_HETERO, assert_grads_close

pytestmark = pytest.mark.gpu


def make_case(seed=0, empty_rank1=False, N=160, H=128):
    rng = np.random.default_rng(seed)
    F, C, W, L = 48, 5, 40, 4
    X = torch.as_tensor(rng.random((N, F), dtype=np.float32))
    mask = np.zeros(N, bool)
    mask[rng.permutation(N // 2)[:37]] = True                   # uneven: 37 masked nodes in block 0, 21 in block 1
    if not empty_rank1:                                         # (empty_rank1: every masked node lies in block 0 -- Planetoid's
        mask[N // 2 + rng.permutation(N // 2)[:21]] = True      #  train nodes 0..139 with contiguous node blocks)
    sel = np.flatnonzero(mask)
    ids = rng.integers(0, N, (len(sel), W, L))
    ids[:, :, 0] = sel[:, None]
    codes = np.minimum(rng.integers(0, L, (len(sel), W, L)), np.arange(L)[None, None, :])
    G = torch.as_tensor(rng.standard_normal((len(sel), C)).astype(np.float32))
    return dict(N=N, F=F, H=H, C=C, W=W, L=L, X=X, sel=sel, ids=ids, codes=codes, G=G)


def build(variant, case):
    import pathnet_amd
    torch.manual_seed(321)
    cls = {"homo": pathnet_amd.PathNet_homo, "pagg": pathnet_amd.PAGG, "hetero": pathnet_amd.PathNet}[variant]
    return cls(case["F"], case["H"], case["C"], case["L"] if variant != "pagg" else case["N"]).cuda()


def masks(case, p=0.5):
    S, W, H, L = len(case["sel"]), case["W"], case["H"], case["L"]
    g = torch.Generator().manual_seed(17)
    return ((torch.rand(L, S * W, H, generator=g) >= p).float() / (1 - p),
            (torch.rand(S, 2 * H, generator=g) >= p).float() / (1 - p))


def worker(rank, world, port, variant, ret, empty_rank1=False, mode="sharded", case_kw=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pathnet_amd import dist as pdist

        case = make_case(empty_rank1=empty_rank1, **(case_kw or {}))
        m = build(variant, case).train()
        lo, n_loc = pdist.node_block(case["N"], world, rank)
        mine = (case["sel"] >= lo) & (case["sel"] < lo + n_loc)
        if mode.startswith("replicated"):   # every rank holds all of X: plain data parallelism over the masked nodes
            runner = pdist.ReplicatedAggregator(m, case["N"], comm=pdist.Comm())
            runner.compact_nodes = mode == "replicated_touched"     # ... restricted to the rows of X its paths touch
            X_in = case["X"].cuda()
        else:
            runner = pdist.ShardedAggregator(m, case["N"], lo, n_loc, comm=pdist.Comm(),      # ops = HipOps (default); gloo: staged
                                             exchange="sparse" if "sparse" in mode else "dense")
            assert isinstance(runner.ops, pdist.HipOps)
            X_in = case["X"][lo:lo + n_loc].cuda()
        assert runner.distributed
        if mode.endswith("_serial"):        # collectives on the compute stream, as before round 4
            runner.comm.overlap = False
        if mode == "sharded_sparse":
            runner.comm.measure_exposed = True
        if mode == "sharded_begin":         # projection + all-gather started ahead of the call (under the sampler, in a step)
            runner.comm.measure_exposed = True
            runner.begin_step(X_in)
        ms, mc = masks(case)
        runner.mask_seq, runner.mask_cls = ms.cuda(), mc.cuda()         # the whole batch's masks
        out = runner(X_in,
                     torch.as_tensor(case["ids"][mine].reshape(int(mine.sum()), case["W"] * case["L"])),
                     case["W"], case["L"], torch.as_tensor(case["sel"][mine].astype(np.int32)),
                     torch.as_tensor(case["codes"][mine]))
        (out * case["G"][mine].cuda()).sum().backward()
        runner.allreduce_grads(average=False)
        torch.cuda.synchronize()
        if mode == "sharded_begin":
            ex = runner.comm.exposed_ms()
            assert set(ex) == {"all_gather_Xh", "reduce_scatter_dXh+fc0_bwd"} and all(v >= 0.0 for v in ex.values()), ex
        if mode == "sharded_sparse":
            ex = runner.comm.exposed_ms()
            assert set(ex) == {"sparse_Xh", "sparse_dXh+fc0_bwd"} and all(v >= 0.0 for v in ex.values()), ex
        if mode.startswith("sharded"):
            assert runner.last_exchange == ("sparse" if "sparse" in mode else "dense")
        ret[rank] = (out.detach().cpu().numpy(), {k: v.grad.cpu().numpy().copy() for k, v in m.named_parameters()},
                     np.flatnonzero(mine))
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("variant,empty_rank1,mode", [("homo", False, "sharded"), ("hetero", False, "sharded"), ("pagg", False, "sharded"),
                                                      ("homo", True, "sharded"), ("hetero", True, "sharded"),
                                                      ("homo", False, "sharded_begin"), ("hetero", True, "sharded_begin"),
                                                      ("homo", False, "sharded_serial"),
                                                      # the sparse exchange: rows asked for by id, all-to-all both ways (dist.py)
                                                      ("homo", False, "sharded_sparse"), ("hetero", False, "sharded_sparse"),
                                                      ("pagg", False, "sharded_sparse"), ("homo", True, "sharded_sparse"),
                                                      ("hetero", True, "sharded_sparse_serial"),
                                                      # 163 nodes on two ranks (blocks of 82 and 81), hidden size 100 (padded to 128)
                                                      ("homo", False, "sharded_odd"), ("hetero", False, "sharded_sparse_odd"),
                                                      ("pagg", True, "sharded_sparse_odd"),
                                                      # three ranks on the one GPU: blocks of 55 / 55 / 53 nodes, ragged counts
                                                      ("homo", False, "sharded_sparse_odd_w3"), ("hetero", False, "sharded_odd_w3")])
def test_two_ranks_hip_ops_match_the_single_process_module(variant, empty_rank1, mode):
    """empty_rank1: rank 1 has no masked node -- its aggregator calls run with S = 0 (empty index arrays, NULL pointers) and must
    still take part in the collectives with zero gradients"""
    world = 3 if mode.endswith("_w3") else 2
    mode = mode[:-3] if mode.endswith("_w3") else mode
    mgr = mp.Manager()
    ret = mgr.dict()
    kw = dict(N=163, H=100) if mode.endswith("_odd") else {}
    mp.spawn(worker, args=(world, free_port(), variant, ret, empty_rank1, mode, kw), nprocs=world, join=True)
    case = make_case(empty_rank1=empty_rank1, **kw)
    m = build(variant, case).train()
    ms, mc = masks(case)
    m._mask_seq, m._mask_cls = ms.cuda(), mc.cuda()
    S = len(case["sel"])
    mask = np.zeros(case["N"], bool)
    mask[case["sel"]] = True
    out = m(case["X"].cuda(), torch.as_tensor(case["ids"].reshape(S, -1)), case["W"], case["L"], mask,
            torch.as_tensor(case["codes"]), None)
    (out * case["G"].cuda()).sum().backward()
    want = out.detach().cpu().numpy()
    for rank in range(world):
        got, grads, rows = ret[rank]
        assert got.shape[0] == len(rows)
        if len(rows):
            assert np.abs(got - want[rows]).max() < 2e-6, rank
        assert_grads_close(grads, {k: v.grad for k, v in m.named_parameters()}, zero_ok=ZERO_OK_HETERO, tag="rank %d" % rank)


@pytest.mark.parametrize("mode", ["replicated", "replicated_touched"])
@pytest.mark.parametrize("variant,empty_rank1", [("homo", False), ("hetero", False), ("pagg", False), ("homo", True)])
def test_two_ranks_replicated_features_match_the_single_process_module(variant, empty_rank1, mode):
    """dist.ReplicatedAggregator: all of X on every rank, each rank aggregates its own masked nodes, the flat gradient
    all-reduce is the only collective of the homo / PAGG classes (the hetero class also gathers the batch's index arrays)"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), variant, ret, empty_rank1, mode), nprocs=world, join=True)
    case = make_case(empty_rank1=empty_rank1)
    m = build(variant, case).train()
    ms, mc = masks(case)
    m._mask_seq, m._mask_cls = ms.cuda(), mc.cuda()
    S = len(case["sel"])
    mask = np.zeros(case["N"], bool)
    mask[case["sel"]] = True
    out = m(case["X"].cuda(), torch.as_tensor(case["ids"].reshape(S, -1)), case["W"], case["L"], mask,
            torch.as_tensor(case["codes"]), None)
    (out * case["G"].cuda()).sum().backward()
    want = out.detach().cpu().numpy()
    for rank in range(world):
        got, grads, rows = ret[rank]
        assert got.shape[0] == len(rows)
        if len(rows):
            assert np.abs(got - want[rows]).max() < 2e-6, rank
        assert_grads_close(grads, {k: v.grad for k, v in m.named_parameters()}, zero_ok=ZERO_OK_HETERO, tag="rank %d" % rank)


RCCL_SCRIPT = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))       # backend "nccl" IS RCCL on ROCm
from pathnet_amd import dist as pdist
sys.path.insert(0, os.path.join(%r, "tests"))
from test_gpu_dist import make_case, build
for variant in ("homo", "hetero"):
    case = make_case()
    m = build(variant, case).eval()
    S = len(case["sel"])
    mask = np.zeros(case["N"], bool); mask[case["sel"]] = True
    X = case["X"].cuda()
    want = m(X, torch.as_tensor(case["ids"].reshape(S, -1)), case["W"], case["L"], mask, torch.as_tensor(case["codes"]), None)
    (want * case["G"].cuda()).sum().backward()
    ref = {k: v.grad.clone() for k, v in m.named_parameters()}
    m.zero_grad()
    from gradcheck import ZERO_OK_HETERO, assert_grads_close
    # dense: all-gather / reduce-scatter; sparse: the three all-to-alls of the touched-row exchange (row ids, Xh rows, dXh
    # rows -- all_to_all_single with split sizes through RCCL); replicated: the index all-gather + the gradient all-reduce
    for mode in ("dense", "sparse", "replicated"):
        m.zero_grad(set_to_none=True)
        comm = pdist.Comm(always=True, timing=True)
        if mode == "replicated":
            runner = pdist.ReplicatedAggregator(m, case["N"], comm=comm)
        else:
            runner = pdist.ShardedAggregator(m, case["N"], 0, case["N"], comm=comm, exchange=mode)
        assert runner.distributed
        out = runner(X, torch.as_tensor(case["ids"].reshape(S, -1)), case["W"], case["L"],
                     torch.as_tensor(case["sel"].astype(np.int32)), torch.as_tensor(case["codes"]))
        (out * case["G"].cuda()).sum().backward()
        runner.allreduce_grads(average=True)
        torch.cuda.synchronize()
        assert (out - want).abs().max().item() < 2e-6, (variant, mode)
        assert_grads_close({k: v.grad for k, v in m.named_parameters()}, ref, zero_ok=ZERO_OK_HETERO, tag=variant + "/" + mode)
        sec = runner.comm.seconds
        assert all(t >= 0 for t in sec.values()) and sec["all_reduce_grads"] > 0, sec
        if mode == "dense":
            assert sec["all_gather_Xh"] > 0 and sec["reduce_scatter_dXh"] > 0, sec
        if mode == "sparse":
            assert sec["sparse_index"] > 0 and sec["sparse_Xh"] > 0 and sec["sparse_dXh"] > 0 and sec["all_gather_Xh"] == 0, sec
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
'''


def test_rccl_collectives_in_a_one_rank_group():
    """The collectives of the sharded step through RCCL itself (backend "nccl"), in a one-rank group forced through
    the multi-rank path (Comm(always=True)): all-gather of Xh, of the counts and of the hetero class's index arrays,
    reduce-scatter of dXh, the sparse exchange's three all-to-alls, the replicated mode, all-reduce of the flat gradient
    buffer -- identities at world size 1, but the very calls
    `bench.py --gpus N` makes on an 8-GPU node, here on the one GPU a test box has."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", RCCL_SCRIPT % (root, root)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("workload,launcher", [("cora", "self"), ("bgp", "torchrun")])
def test_bench_with_two_ranks_on_one_gpu(workload, launcher):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank) AND as a bare command
    (`python bench.py --gpus 2`: it then starts its own ranks), with both ranks on the one GPU of a test
    box: gloo replaces RCCL (which needs a GPU per rank; chosen automatically when there are fewer devices than ranks),
    everything else -- sharded workload, ShardedAggregator with the HIP kernels, max-over-ranks timing, per-rank collective
    times, the JSON line -- is the N > 1 path of the bench.  The cora line also carries the configurations north_star names
    for the 8-GPU box (bgp_strong, configs4_replicated, configs4_sharded: shrunk here, PN_BENCH_MULTI_SMALL) with the
    exposed time of every collective per rank and the scale model's prediction for the same world size."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PN_BENCH_MULTI_SMALL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PN_DIST_BACKEND"):
        env.pop(k, None)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        env["PN_DIST_BACKEND"] = "gloo"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + tail
    import tempfile
    extras = os.path.join(tempfile.mkdtemp(), "bench_extras.json")
    env["PN_BENCH_EXTRAS"] = extras
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    # stdout: the slim line the driver parses (< 6 KB, the contract's keys); the full record: bench_extras.json
    assert len(lines[0]) < 6144
    slim = json.loads(lines[0])
    d = json.load(open(extras))
    assert slim["n_gpus"] == 2 and slim["value"] == float("%.6g" % d["value"]) and slim["roofline"]["kernel"] == d["roofline"]["kernel"]
    assert slim["collectives"]["rccl_ranks_seen"] == 2 and slim["config"] == {k: d["config"][k] for k in slim["config"]}
    if workload == "cora":
        assert all(slim[k]["value"] == float("%.6g" % d[k]["value"]) for k in ("bgp_strong", "configs4_replicated", "configs4_sharded"))
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["parallelism"] == "node-shard x2"
    assert d["scaling"] == ("weak" if workload == "cora" else "strong")
    per_rank = d["collectives"]["ms_per_step_by_rank"]
    # two ranks answered through the backend; here they share one device (on the driver's 8-GPU run: N distinct ones)
    assert d["collectives"]["rccl_ranks_seen"] == 2 and d["collectives"]["backend"] == "gloo"
    assert d["collectives"]["distinct_devices"] == 1
    assert len(per_rank) == 2 and all(c["all_gather_Xh"] > 0 and c["all_reduce_grads"] > 0 for c in per_rank)
    assert d["dispersion"]["step_ms"]["min"] <= d["dispersion"]["step_ms"]["median"] <= d["dispersion"]["step_ms"]["max"]
    if workload == "cora":
        assert d["config"]["nodes"] == 2 * 2708 and abs(d["config"]["paths_per_step"] - 2 * 1299 * 40) <= 2 * 40
        pred = d["collectives"]["model_prediction"]
        assert pred and "error" not in pred and 0.5 < pred["speed_up_or_efficiency"] <= 1.0
        for name, exch in (("bgp_strong", "dense"), ("configs4_replicated", None), ("configs4_sharded", "sparse")):
            blk = d[name]
            assert blk["value"] > 0 and blk["scaling"] == "strong" and len(blk["by_rank"]) == 2, name
            assert sum(rk["masked_nodes"] for rk in blk["by_rank"]) * 40 == round(blk["value"] * blk["ms_per_step"] * 1e-3), name
            assert all(rk["exchange"] == exch for rk in blk["by_rank"]), (name, blk["by_rank"])
            assert blk["model_prediction"] and "error" not in blk["model_prediction"], name
        assert set(d["configs4_sharded"]["by_rank"][0]["exposed_ms"]) == {"sparse_Xh", "sparse_dXh+fc0_bwd"}
        assert set(d["bgp_strong"]["by_rank"][0]["exposed_ms"]) == {"all_gather_Xh", "reduce_scatter_dXh+fc0_bwd"}
    # (which stage comes out as the dominant one is not asserted: two processes time-share the GPU here)
    assert "cpu_baseline" not in d and d["roofline"]["kernel"] in d["stages_ms"]

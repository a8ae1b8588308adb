"""Node sharding of the aggregator across the GPUs of one box (one process per GPU, RCCL over xGMI).

The reference is single-process; this is the multi-GPU design of §8(e):

  * the graph's nodes are split into equal contiguous row blocks, one per rank; a rank holds the
    feature rows X[row_begin : row_begin+row_count] of its own nodes and trains on the masked nodes
    among them (their paths can visit any node);
  * forward : every rank projects its own rows (fc0, the only place X is read), then ONE all-gather
    of the projected feature matrix Xh [N, H] -- N*H*4 bytes, e.g. 10 MB at Pubmed size -- gives
    every rank the table its path gather reads;
  * backward: d loss / d Xh comes out of the aggregator backward for all N rows; ONE reduce-scatter
    returns each row block to its owner, which finishes fc0's backward on its rows;
  * parameter gradients: ONE flat all-reduce (a single bucket: the model has ~0.4 M parameters).

No other data moves between GPUs: path sampling and aggregation of a node are independent of every
other node's.  Exact for the homo / PAGG classes.  For the hetero class (PathNet) the reference's
[W, S] re-view of the hidden states mixes paths of different masked nodes *within a batch*
(PathNet_run.py:196-197), so its output depends on how the masked nodes are batched; sharding
changes the batch and therefore (by the reference's own definition) the result -- each shard is
exactly what the reference computes when given that shard as its batch.

The compute backend is an object with four methods (project / forward / backward /
linear_backward).  The product backend is HipOps (libpathnet_hip.so); CPU tests plug in a checker
backend to exercise the sharding and the collectives with gloo.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from . import modules as M


class HipOps:
    """The HIP kernels behind the sharded aggregator."""

    def project(self, variant, X_loc, w, b):
        lib = _lib.load()
        X_loc = X_loc.contiguous()
        rows, out_f = X_loc.shape[0], w.shape[0]
        out = torch.empty((rows, out_f), dtype=torch.float32, device=X_loc.device)
        ws = torch.empty(max(_lib.LINEAR_SPLIT_MAX * rows * out_f, 1), dtype=torch.float32, device=X_loc.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(X_loc.device).cuda_stream)
        _lib.check(lib.pn_linear_forward(X_loc.data_ptr(), w.data_ptr(), b.data_ptr(), rows, X_loc.shape[1], out_f,
                                         1 if variant == "homo" else 0, out.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                         stream))
        return out

    def _args(self, cfg, Xh, ids, codes, sel, p):
        a = _lib.PaggArgs()
        a.shape = M._shape(cfg["variant"], cfg["N"], cfg["F"], cfg["H"], cfg["C"], cfg["S"], cfg["W"], cfg["L"])
        a.Xh_in = Xh.data_ptr()
        a.ids, a.codes, a.sel = ids.data_ptr(), codes.data_ptr(), sel.data_ptr()
        for k in M._HEAD_PARAMS[2:]:
            setattr(a, k, p[k].data_ptr() if p.get(k) is not None else None)
        a.bank_w, a.bank_b = cfg["bank_w"].data_ptr(), cfg["bank_b"].data_ptr()
        a.p_seq, a.p_cls, a.seed = cfg["p_seq"], cfg["p_cls"], cfg["seed"]
        return a

    def forward(self, cfg, Xh, ids, codes, sel, p):
        lib = _lib.load()
        dev = Xh.device
        out = torch.empty((cfg["S"], cfg["C"]), dtype=torch.float32, device=dev)
        ws = torch.empty(max(M.workspace_bytes(cfg["variant"], cfg["N"], cfg["F"], cfg["H"], cfg["C"], cfg["S"],
                                               cfg["W"], cfg["L"]), 1), dtype=torch.uint8, device=dev)
        a = self._args(cfg, Xh, ids, codes, sel, p)
        a.out = out.data_ptr()
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        if cfg["S"] > 0:
            _lib.check(lib.pn_pagg_forward(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out, (cfg, Xh, ids, codes, sel, p, ws)

    def backward(self, state, g_out):
        """-> (g_Xh [N,H], grads): grads maps the head parameter names (without fc0) to tensors and
        "bank_w"/"bank_b" to the stacked [L,H,H] / [L,H] gradients."""
        lib = _lib.load()
        cfg, Xh, ids, codes, sel, p, ws = state
        dev = Xh.device
        a = self._args(cfg, Xh, ids, codes, sel, p)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        g_out = g_out.contiguous().float()
        a.g_out = g_out.data_ptr()
        g_Xh = torch.empty_like(Xh)
        a.g_Xh = g_Xh.data_ptr()
        grads = {"bank_w": torch.empty_like(cfg["bank_w"]), "bank_b": torch.empty_like(cfg["bank_b"])}
        a.g_bank_w, a.g_bank_b = grads["bank_w"].data_ptr(), grads["bank_b"].data_ptr()
        for k in M._HEAD_PARAMS[2:]:
            if p.get(k) is None:
                continue
            grads[k] = torch.empty_like(p[k])
            setattr(a, "g_" + k, grads[k].data_ptr())
        if cfg["S"] > 0:
            _lib.check(lib.pn_pagg_backward(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        else:
            g_Xh.zero_()
            for g in grads.values():
                g.zero_()
        return g_Xh, grads

    def linear_backward(self, variant, dXh_loc, Xh_loc, X_loc, w):
        lib = _lib.load()
        dXh_loc = dXh_loc.contiguous()
        g_w, g_b = torch.empty_like(w), torch.empty(w.shape[0], dtype=torch.float32, device=w.device)
        gate = Xh_loc.data_ptr() if variant == "homo" else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(w.device).cuda_stream)
        _lib.check(lib.pn_linear_backward(dXh_loc.data_ptr(), gate, X_loc.data_ptr(), w.data_ptr(), X_loc.shape[0],
                                          w.shape[1], w.shape[0], g_w.data_ptr(), g_b.data_ptr(), None, stream))
        return g_w, g_b


class _ShardedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, cfg, X_loc, ids, codes, sel, *params):
        p = M._split_params(params, cfg["L"])
        ops, group = runner.ops, runner.group
        Xh_loc = ops.project(cfg["variant"], X_loc, p["fc0_w"], p["fc0_b"])
        world = dist.get_world_size(group) if runner.distributed else 1
        if world > 1:
            Xh = torch.empty((cfg["N"], cfg["H"]), dtype=Xh_loc.dtype, device=Xh_loc.device)
            dist.all_gather_into_tensor(Xh, Xh_loc.contiguous(), group=group)     # the one forward collective
        else:
            Xh = Xh_loc
        out, state = ops.forward(cfg, Xh, ids, codes, sel, p)
        ctx.runner, ctx.state, ctx.cfg = runner, state, cfg
        ctx.save_for_backward(X_loc, Xh_loc, p["fc0_w"])
        ctx.present = [t is not None for t in params]
        return out

    @staticmethod
    def backward(ctx, g_out):
        runner, cfg = ctx.runner, ctx.cfg
        ops, group = runner.ops, runner.group
        X_loc, Xh_loc, fc0_w = ctx.saved_tensors
        g_Xh, grads = ops.backward(ctx.state, g_out)
        world = dist.get_world_size(group) if runner.distributed else 1
        if world > 1:
            g_loc = torch.empty_like(Xh_loc)
            dist.reduce_scatter_tensor(g_loc, g_Xh.contiguous(), group=group)     # the one backward collective
        else:
            g_loc = g_Xh
        grads["fc0_w"], grads["fc0_b"] = ops.linear_backward(cfg["variant"], g_loc, Xh_loc, X_loc, fc0_w)
        L = cfg["L"]
        head = tuple(grads.get(k) if pres else None for k, pres in zip(M._HEAD_PARAMS, ctx.present[:10]))
        return (None, None, None, None, None, None) + head + tuple(grads["bank_w"][d] for d in range(L)) + tuple(
            grads["bank_b"][d] for d in range(L))


class ShardedAggregator:
    """Runs a PathNet / PathNet_homo / PAGG module on this rank's node block.

    module      : the (replicated) aggregator module; its parameters are the trainable state
    n_total     : nodes in the whole graph;  row_begin/row_count: this rank's block (equal on every rank)
    """

    def __init__(self, module, n_total, row_begin, row_count, group=None, ops=None):
        self.module, self.n_total, self.row_begin, self.row_count = module, int(n_total), int(row_begin), int(row_count)
        self.group = group
        self.ops = ops if ops is not None else HipOps()
        self.distributed = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size(group) if self.distributed else 1
        if self.row_count * world != self.n_total:
            raise ValueError("node blocks must be equal: %d rows x %d ranks != %d nodes (pad the graph)"
                             % (self.row_count, world, self.n_total))
        self._flat = None

    def __call__(self, X_loc, neis, num_w, walk_len, sel_global, layer_type):
        """X_loc [row_count, F]; sel_global: global node ids of this rank's masked nodes (inside its block);
        neis / layer_type: their paths [S, W*L] / [S, W, L] with GLOBAL node ids.  -> logits [S, C]."""
        m = self.module
        dev = X_loc.device
        ids, codes, sel, S = M._as_index_tensors(neis, layer_type, sel_global, num_w, walk_len, dev)
        fw, fb, params = m._param_inputs()
        p = m.dropout_p() if m.training else 0.0
        cfg = dict(variant=m.variant, N=self.n_total, F=X_loc.shape[1], H=m.hidden_size, C=m.out_size, S=S,
                   W=int(num_w), L=int(walk_len), p_seq=p, p_cls=p, bank_w=fw, bank_b=fb,
                   seed=int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0)
        return _ShardedFn.apply(self, cfg, X_loc.contiguous().float(), ids, codes, sel, *params)

    def allreduce_grads(self, average=True):
        """One flat all-reduce of every parameter gradient (single bucket)."""
        if not self.distributed or dist.get_world_size(self.group) == 1:
            return
        ps = [q for q in self.module.parameters() if q.requires_grad]
        for q in ps:
            if q.grad is None:
                q.grad = torch.zeros_like(q)
        flat = torch.cat([q.grad.reshape(-1) for q in ps])
        dist.all_reduce(flat, group=self.group)
        if average:
            flat /= dist.get_world_size(self.group)
        at = 0
        for q in ps:
            n = q.numel()
            q.grad.copy_(flat[at:at + n].view_as(q))
            at += n

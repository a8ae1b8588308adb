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
  * parameter gradients: ONE all-reduce of a persistent flat buffer the .grad tensors are views of.

The step's batch is the concatenation of the ranks' masked-node lists in rank order; rank r computes
the pooling groups [offset_r, offset_r + S_r) of it (pn_pagg_shape.S_total / group_begin).  One tiny
all-gather of the S_r tells every rank its offset; dropout counters are positions in that batch, so the
masks of different ranks are independent draws of one stream, as in a single process.

homo / PAGG: a group reads only its own paths -- no other exchange, result identical to one process.
hetero (PathNet): the reference's [W, S] re-view of the hidden states (PathNet_run.py:196-197) makes
group g read paths of OTHER masked nodes of the batch.  The ranks therefore all-gather the batch's index
arrays (ids / codes / sel: 5 bytes per path step, 1 MB at Cora size per rank) and each computes its
groups from the whole batch's paths: every output row is exactly what a single process computes for
the whole batch.  (The alternative, an all-to-all of the [P, H] hidden states, moves 100x the bytes.)

What is replicated: the distance bank Z = bank(Xh) over all N rows and its backward run on every rank
(2*N*L*H^2 flops each way).  Sharding them would replace the all-gather of Xh by one of Z -- L times
the bytes over xGMI -- for GEMMs that take 0.06 ms at 8 x 2708 nodes and ~25 ms at 10^7 nodes against
~90 ms for moving 30 GB of Z at 7 x 50 GB/s: replication is cheaper at every size.

Two ways to move Xh / d Xh (ShardedAggregator.exchange): "dense" -- the all-gather / reduce-scatter above, every row to
every rank; "sparse" -- a rank's paths touch a fraction of a large graph (12 500 masked nodes x 40 paths x 6 steps reach
~26 % of the 10 M nodes of BASELINE.json configs[4]), so each rank asks the owners for exactly the rows its paths name:
one all-to-all of row ids, one of Xh rows forward, one of d Xh rows back (the owner adds what it receives, in rank
order).  The aggregator then runs on the compact table of touched rows.  "auto" picks sparse when every rank touches
less than half of the graph.  Node blocks are ceil(N / world) rows, the last one shorter: any N runs on any world size.

The compute backend is an object with four methods (project / forward / backward /
linear_backward).  The product backend is HipOps (libpathnet_hip.so); CPU tests plug in a checker
backend to exercise the sharding and the collectives with gloo.  Collectives go through a Comm object
(torch.distributed by default) so tests can stage device tensors through the host for gloo.
"""
import ctypes
import time

import torch
import torch.distributed as dist

from . import _lib
from . import modules as M


class Comm:
    """The four collectives of a step over a torch.distributed group; seconds spent are accumulated per kind
    (device-synchronised only when timing is switched on)."""

    def __init__(self, group=None, timing=False, always=False):
        self.group = group
        self.timing = timing
        self.always = always        # run the collectives even in a one-rank group (exercises RCCL on a single GPU)
        self.seconds = {"all_gather_Xh": 0.0, "reduce_scatter_dXh": 0.0, "all_reduce_grads": 0.0, "all_gather_index": 0.0,
                        "sparse_index": 0.0, "sparse_Xh": 0.0, "sparse_dXh": 0.0}
        # overlap: the two big collectives of a step run on a stream of their own, ordered against the compute stream by
        # events (ShardedAggregator.begin_step / _ShardedFn).  measure_exposed: keep, per kind, the time the compute stream
        # would have to wait for the collective -- from the moment it needs the result to the collective's end (timed events,
        # read back by exposed_ms(); only what the overlap does not hide shows up here)
        self.overlap = True
        self.measure_exposed = False
        self._streams = {}
        self._exposed = []          # (kind, event the consumer was ready at, event the collective ended at)

    def stream(self, device):
        """this Comm's communication stream on `device`"""
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        st = self._streams.get(key)
        if st is None:
            st = self._streams[key] = torch.cuda.Stream(device=key)
        return st

    def note_exposed(self, kind, need_ev, done_ev):
        if self.measure_exposed:
            self._exposed.append((kind, need_ev, done_ev))

    def exposed_ms(self):
        """-> {kind: summed milliseconds the consumers were (or would have been) stalled}; clears the record.  Synchronises."""
        out = {}
        for kind, need, done in self._exposed:
            done.synchronize()
            out[kind] = out.get(kind, 0.0) + max(0.0, need.elapsed_time(done))
        self._exposed = []
        return out

    def world(self):
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def rank(self):
        return dist.get_rank(self.group) if (dist.is_available() and dist.is_initialized()) else 0

    def active(self):
        """collectives are issued: more than one rank, or forced"""
        return self.world() > 1 or (self.always and dist.is_available() and dist.is_initialized())

    def _timed(self, kind, fn, tensor):
        if not self.timing:
            return fn()
        if tensor.is_cuda:
            torch.cuda.synchronize(tensor.device)
        t0 = time.perf_counter()
        out = fn()
        if tensor.is_cuda:
            torch.cuda.synchronize(tensor.device)
        self.seconds[kind] += time.perf_counter() - t0
        return out

    # -- primitives: RCCL moves device memory; a backend that moves host memory (gloo: two ranks on ONE GPU in the tests,
    #    where RCCL refuses to run) gets the device tensors staged through the host ------------------------------------
    def _staged(self, t):
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _all_gather(self, out, inp):
        if self._staged(inp):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def _reduce_scatter(self, out, inp):
        if self._staged(inp):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.reduce_scatter_tensor(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.reduce_scatter_tensor(out, inp, group=self.group)

    def _all_reduce(self, t):
        if self._staged(t):
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)

    def _all_to_all(self, out, inp, out_splits, in_splits):
        if self._staged(inp):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    # -- what a step uses ----------------------------------------------------------------------------------------------
    def all_gather_rows(self, t, kind="all_gather_Xh"):
        """[rows, ...] per rank (equal rows) -> [world*rows, ...]"""
        t = t.contiguous()
        out = torch.empty((t.shape[0] * self.world(),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self._timed(kind, lambda: self._all_gather(out, t), t)
        return out

    def reduce_scatter_rows(self, t, rows):
        t = t.contiguous()
        out = torch.empty((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self._timed("reduce_scatter_dXh", lambda: self._reduce_scatter(out, t), t)
        return out

    def all_reduce(self, t):
        self._timed("all_reduce_grads", lambda: self._all_reduce(t), t)

    def all_to_all_rows(self, t, in_counts, out_counts, kind):
        """rows of t, in_counts[q] of them for rank q (in rank order) -> the rows the ranks sent here, out_counts[q] from
        rank q, in rank order"""
        t = t.contiguous()
        out = torch.empty((int(sum(out_counts)),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self._timed(kind, lambda: self._all_to_all(out, t, [int(c) for c in out_counts], [int(c) for c in in_counts]), t)
        return out

    def count_matrix(self, row, device):
        """every rank's world counts -> [world][world] (row r = rank r's list).  `row` may be a device tensor: the counts then
        travel without visiting the host first, and the ONE read-back is that of the gathered matrix"""
        if torch.is_tensor(row):
            mine = row.to(device=device, dtype=torch.int64).reshape(1, -1)
        else:
            mine = torch.tensor([[int(v) for v in row]], dtype=torch.int64, device=device)
        return self.all_gather_rows(mine, kind="all_gather_index").tolist()

    def counts(self, n, device):
        """every rank's n -> list of ints"""
        mine = torch.tensor([int(n)], dtype=torch.int64, device=device)
        return [int(v) for v in self.all_gather_rows(mine, kind="all_gather_index").tolist()]

    def all_gather_ragged(self, t, counts):
        """[n_r, ...] per rank -> [sum n_r, ...] in rank order (padded to the largest n_r on the wire)"""
        cap = max(counts)
        pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        full = self.all_gather_rows(pad, kind="all_gather_index")
        return torch.cat([full[r * cap: r * cap + counts[r]] for r in range(len(counts))])


class HipOps:
    """The HIP kernels behind the sharded aggregator."""

    def project(self, variant, X_loc, w, b):
        lib = _lib.load()
        X_loc = X_loc.contiguous()
        dev = X_loc.device
        rows, out_f = X_loc.shape[0], w.shape[0]
        with torch.cuda.device(dev):
            out = torch.empty((rows, out_f), dtype=torch.float32, device=dev)
            ws = torch.empty(max(_lib.LINEAR_SPLIT_MAX * rows * out_f, 1), dtype=torch.float32, device=dev)
            _lib.check(lib.pn_linear_forward(_lib.context(dev), X_loc.data_ptr(), w.data_ptr(), b.data_ptr(), rows,
                                             X_loc.shape[1], out_f, 1 if variant == "homo" else 0, out.data_ptr(),
                                             ws.data_ptr(), ws.numel() * 4, _lib.stream_ptr(dev)))
        return out

    def _args(self, cfg, Xh, ids, codes, sel, p):
        a = _lib.PaggArgs()
        a.shape = M._cfg_shape(cfg)
        a.Xh_in = Xh.data_ptr()
        a.ids, a.codes, a.sel = ids.data_ptr(), codes.data_ptr(), sel.data_ptr()
        a.index_rows_local = 1 if cfg.get("index_rows_local") else 0
        for k in M._HEAD_PARAMS[2:]:
            setattr(a, k, p[k].data_ptr() if p.get(k) is not None else None)
        a.bank_w, a.bank_b = cfg["bank_w"].data_ptr(), cfg["bank_b"].data_ptr()
        a.p_seq, a.p_cls, a.seed = cfg["p_seq"], cfg["p_cls"], cfg["seed"]
        ms, mc = cfg.get("mask_seq"), cfg.get("mask_cls")
        a.mask_seq = ms.data_ptr() if ms is not None else None
        a.mask_cls = mc.data_ptr() if mc is not None else None
        return a

    def forward(self, cfg, Xh, ids, codes, sel, p, ready=None):
        """ready: torch.cuda.Event recorded after the all-gather of Xh on the communication stream, or None (Xh is complete
        in stream order)"""
        lib = _lib.load()
        dev = Xh.device
        with torch.cuda.device(dev):
            out = torch.empty((cfg["S"], cfg["C"]), dtype=torch.float32, device=dev)
            ws = torch.empty(max(M._cfg_workspace_bytes(cfg), 1), dtype=torch.uint8, device=dev)
            a = self._args(cfg, Xh, ids, codes, sel, p)
            a.Xh_ready = ready.cuda_event if ready is not None else None
            a.out = out.data_ptr()
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            if cfg["S"] > 0:
                _lib.check(lib.pn_pagg_forward(_lib.context(dev), ctypes.byref(a), _lib.stream_ptr(dev)))
        return out, (cfg, Xh, ids, codes, sel, p, ws)

    def backward(self, state, g_out, ready=None):
        """-> (g_Xh [N,H], grads): grads maps the head parameter names (without fc0) to tensors and
        "bank_w"/"bank_b" to the stacked [L,H,H] / [L,H] gradients.  ready: a torch.cuda.Event the library records as soon
        as g_Xh is complete (the weight gradients follow it), or None."""
        lib = _lib.load()
        cfg, Xh, ids, codes, sel, p, ws = state
        dev = Xh.device
        with torch.cuda.device(dev):
            a = self._args(cfg, Xh, ids, codes, sel, p)
            a.g_Xh_ready = ready.cuda_event if ready is not None else None
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
            # (S == 0: the library zero-fills every gradient it was given)
            _lib.check(lib.pn_pagg_backward(_lib.context(dev), ctypes.byref(a), _lib.stream_ptr(dev)))
        return g_Xh, grads

    def linear_backward(self, variant, dXh_loc, Xh_loc, X_loc, w, deterministic=False):
        lib = _lib.load()
        dev = w.device
        with torch.cuda.device(dev):
            dXh_loc = dXh_loc.contiguous()
            g_w, g_b = torch.empty_like(w), torch.empty(w.shape[0], dtype=torch.float32, device=dev)
            gate = Xh_loc.data_ptr() if variant == "homo" else None
            ws = None       # deterministic: chunk sums of the weight gradient, added in a fixed order
            if deterministic:
                ws = torch.empty(_lib.LINEAR_BWD_SPLIT_MAX * (w.numel() + w.shape[0]), dtype=torch.float32, device=dev)
            _lib.check(lib.pn_linear_backward(_lib.context(dev), dXh_loc.data_ptr(), gate, X_loc.data_ptr(), w.data_ptr(),
                                              X_loc.shape[0], w.shape[1], w.shape[0], g_w.data_ptr(), g_b.data_ptr(),
                                              None, ws.data_ptr() if ws is not None else None,
                                              ws.numel() * 4 if ws is not None else 0, _lib.stream_ptr(dev)))
        return g_w, g_b


def _new_event(device, timing=False):
    """an event whose handle exists (torch creates it at the first record) -- the library records / waits on the raw handle"""
    ev = torch.cuda.Event(enable_timing=timing)
    ev.record(torch.cuda.current_stream(device))
    return ev


def node_block(n_total, world, rank):
    """(row_begin, row_count) of `rank`'s node block: ceil(n_total / world) rows each, the last blocks shorter (a block may
    be empty when world does not divide the graph)"""
    B = -(-int(n_total) // max(int(world), 1))
    lo = min(int(rank) * B, int(n_total))
    return lo, min(B, int(n_total) - lo)


class _SparsePlan:
    """What one step's sparse exchange needs (ShardedAggregator._plan_sparse): the sorted global ids of the rows this rank's
    paths touch, how many of them each owner holds (need), which of this rank's rows every other rank asked for (serve_local,
    serve counts), and the node-id -> compact-row map of the step."""
    __slots__ = ("touched", "need", "serve", "serve_local", "T")


class _ShardedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, cfg, X_loc, ids, codes, sel, *params):
        p = M._split_params(params, cfg["L"])
        ops, comm = runner.ops, runner.comm
        plan = cfg.get("sparse_plan")
        pre = runner._take_prefetched(X_loc, p["fc0_w"], p["fc0_b"]) if plan is None else None
        if pre is not None:             # begin_step projected and gathered already, on the communication stream
            Xh_loc, Xh, ready = pre
        elif plan is not None:
            Xh_loc, Xh, ready = runner._project_and_fetch(cfg["variant"], X_loc, p["fc0_w"], p["fc0_b"], plan)
        else:
            Xh_loc, Xh, ready = runner._project_and_gather(cfg["variant"], X_loc, p["fc0_w"], p["fc0_b"])
        if ready is not None and comm.measure_exposed:
            need = torch.cuda.Event(enable_timing=True)
            need.record()
            comm.note_exposed("sparse_Xh" if plan is not None else "all_gather_Xh", need, runner._ag_done)
        if ready is not None and not isinstance(ops, HipOps):      # a checker backend: plain stream order
            torch.cuda.current_stream(X_loc.device).wait_event(ready)
            ready = None
        out, state = ops.forward(cfg, Xh, ids, codes, sel, p, ready) if isinstance(ops, HipOps) else ops.forward(cfg, Xh, ids, codes, sel, p)
        ctx.runner, ctx.state, ctx.cfg = runner, state, cfg
        ctx.save_for_backward(X_loc, Xh_loc, p["fc0_w"])
        ctx.present = [t is not None for t in params]
        return out

    @staticmethod
    def backward(ctx, g_out):
        runner, cfg = ctx.runner, ctx.cfg
        ops, comm = runner.ops, runner.comm
        X_loc, Xh_loc, fc0_w = ctx.saved_tensors
        dev = X_loc.device
        plan = cfg.get("sparse_plan")
        rows = Xh_loc.shape[0]

        def to_owner(g_Xh):
            """d Xh of the table this rank computed on -> the rows of its own block, summed over the ranks"""
            if not comm.active():
                return g_Xh
            if plan is not None:
                return runner._return_sparse(g_Xh, plan, rows)
            return comm.reduce_scatter_rows(g_Xh, runner.block_rows)[:rows]    # the one backward collective

        overlap = comm.active() and comm.overlap and X_loc.is_cuda and isinstance(ops, HipOps)
        if not overlap:
            g_Xh, grads = ops.backward(ctx.state, g_out)
            g_loc = to_owner(g_Xh)
            grads["fc0_w"], grads["fc0_b"] = ops.linear_backward(cfg["variant"], g_loc, Xh_loc, X_loc, fc0_w,
                                                                 deterministic=cfg.get("deterministic", False))
        else:
            # the library records `ready` as soon as d Xh is complete; its weight-gradient GEMMs then run under the
            # reduce-scatter (or the sparse return) and fc0's backward, which follow on the communication stream
            cur, cst = torch.cuda.current_stream(dev), comm.stream(dev)
            ready = _new_event(dev, timing=comm.measure_exposed)
            g_Xh, grads = ops.backward(ctx.state, g_out, ready)
            g_Xh.record_stream(cst)
            with torch.cuda.stream(cst):
                cst.wait_event(ready)
                g_loc = to_owner(g_Xh)
                grads["fc0_w"], grads["fc0_b"] = ops.linear_backward(cfg["variant"], g_loc, Xh_loc, X_loc, fc0_w,
                                                                     deterministic=cfg.get("deterministic", False))
                done = torch.cuda.Event(enable_timing=comm.measure_exposed)
                done.record(cst)
            if comm.measure_exposed:        # the compute stream needs fc0's gradients when the library's own work is over
                need = torch.cuda.Event(enable_timing=True)
                need.record(cur)
                comm.note_exposed(("sparse_dXh" if plan is not None else "reduce_scatter_dXh") + "+fc0_bwd", need, done)
            cur.wait_event(done)
            for t in (g_loc, grads["fc0_w"], grads["fc0_b"]):
                t.record_stream(cur)
        L = cfg["L"]
        head = tuple(grads.get(k) if pres else None for k, pres in zip(M._HEAD_PARAMS, ctx.present[:10]))
        return (None, None, None, None, None, None) + head + tuple(grads["bank_w"][d] for d in range(L)) + tuple(
            grads["bank_b"][d] for d in range(L))


class ShardedAggregator:
    """Runs a PathNet / PathNet_homo / PAGG module on this rank's node block.

    module      : the (replicated) aggregator module; its parameters are the trainable state
    n_total     : nodes in the whole graph;  row_begin/row_count: this rank's block = node_block(n_total, world, rank)
    exchange    : "auto" (default) / "dense" / "sparse" -- how Xh and d Xh travel (module docstring)
    """

    def __init__(self, module, n_total, row_begin, row_count, group=None, ops=None, comm=None, dropout_seed=0,
                 exchange="auto"):
        self.module, self.n_total, self.row_begin, self.row_count = module, int(n_total), int(row_begin), int(row_count)
        self.comm = comm if comm is not None else Comm(group)
        self.group = self.comm.group
        self.ops = ops if ops is not None else HipOps()
        self.distributed = self.comm.active()
        world = self.comm.world()
        want = node_block(self.n_total, world, self.comm.rank())
        if (self.row_begin, self.row_count) != want:
            raise ValueError("rank %d of %d owns the node block (begin %d, %d rows) of a %d-node graph -- node_block(); got "
                             "(%d, %d)" % (self.comm.rank(), world, want[0], want[1], self.n_total, self.row_begin, self.row_count))
        self.block_rows = -(-self.n_total // world)     # rows of a full block: what the dense collectives move per rank
        self.n_pad = self.block_rows * world            # rows of the gathered table (rows >= n_total are zero, never indexed)
        if exchange not in ("auto", "dense", "sparse"):
            raise ValueError("exchange: auto, dense or sparse")
        self.exchange = exchange
        self.last_exchange = None                       # what the last call used ("dense" / "sparse"; None: one rank)
        self._flat = None           # persistent flat gradient buffer (see flat_grads)
        self._flat_ids = None
        self.batch_counts = None    # masked nodes per rank of the last call
        self._fixed_counts = None
        # every rank draws the step's dropout seed from its own copy of the same generator: equal seeds, no traffic
        self._seed_gen = torch.Generator()
        self._seed_gen.manual_seed(int(dropout_seed))
        self.mask_seq = self.mask_cls = None    # test hook: explicit dropout masks of the WHOLE batch
        self._prefetched = None
        self._ag_done = None

    # ---- forward collective, overlapped ------------------------------------------------------------------------------
    def _on_comm_stream(self, X_loc, Xh_loc, fn):
        """fn() on the communication stream, after Xh_loc (current stream) is complete.  -> (result, ready event) -- or fn()
        in plain stream order with ready = None when nothing can overlap."""
        comm = self.comm
        if not (comm.overlap and X_loc.is_cuda):
            return fn(), None
        dev = X_loc.device
        cur, cst = torch.cuda.current_stream(dev), comm.stream(dev)
        projected = torch.cuda.Event()
        projected.record(cur)
        Xh_loc.record_stream(cst)
        with torch.cuda.stream(cst):
            cst.wait_event(projected)
            Xh = fn()
            ready = torch.cuda.Event(enable_timing=comm.measure_exposed)
            ready.record(cst)
        Xh.record_stream(cur)
        self._ag_done = ready
        return Xh, ready

    def _project_and_gather(self, variant, X_loc, fc0_w, fc0_b):
        """Xh_loc = fc0(X_loc) on the current stream, then the all-gather of Xh on the communication stream.
        -> (Xh_loc, Xh, ready): `ready` is the event the consumer of Xh has to wait for (None: plain stream order)."""
        ops, comm = self.ops, self.comm
        Xh_loc = ops.project(variant, X_loc, fc0_w, fc0_b)
        if not comm.active():
            return Xh_loc, Xh_loc, None

        def gather():
            send = Xh_loc
            if Xh_loc.shape[0] < self.block_rows:       # a short last block: zero rows up to the common block size
                send = torch.nn.functional.pad(Xh_loc, (0, 0, 0, self.block_rows - Xh_loc.shape[0]))
            return comm.all_gather_rows(send)
        Xh, ready = self._on_comm_stream(X_loc, Xh_loc, gather)
        return Xh_loc, Xh, ready

    # ---- sparse exchange: the rows the step's paths touch, and nothing else ---------------------------------------------
    def _plan_sparse(self, ids, sel):
        """Collective.  Decides the step's exchange mode and, for the sparse one, builds its plan and the compact indices.
        -> (plan or None, ids', sel').  ONE host round trip since round 6 -- the ranks' count matrix, gathered from device
        tensors (the per-owner counts no longer visit the host on their own, the touched count is the matrix row's sum) --
        and the exchange of the row ids rides on the communication stream, under fc0's projection of the rank's rows."""
        comm = self.comm
        if self.exchange == "dense" or not comm.active():
            return None, ids, sel
        dev, N, R, B = ids.device, self.n_total, comm.world(), self.block_rows
        flags = torch.zeros(self.n_pad, dtype=torch.bool, device=dev)
        flags[ids.reshape(-1).long().clamp_(0, N - 1)] = True        # (the kernels clamp ids and sel the same way)
        flags[sel.long().clamp_(0, N - 1)] = True
        # owners' counts straight from the flags (a block's touched rows = the sum of its flags): no nonzero() -- whose size is a
        # host read-back of its own -- in front of the collective
        need_dev = flags.view(R, B).sum(1)
        matrix = comm.count_matrix(need_dev, dev)           # matrix[r][o]: rows rank r needs from owner o  (the host round trip)
        if self.exchange == "auto" and max(sum(row) for row in matrix) * 2 >= N:
            return None, ids, sel                           # some rank touches half of the graph: the dense collectives
        me = comm.rank()
        touched = torch.nonzero(flags).flatten()            # (its size is known by now: no further wait)
        plan = _SparsePlan()
        plan.touched, plan.need, plan.T = touched, [int(v) for v in matrix[me]], int(sum(matrix[me]))
        plan.serve = [int(matrix[q][me]) for q in range(R)]
        ask = touched.to(torch.int32)

        def exchange_ids():
            asked = comm.all_to_all_rows(ask, plan.need, plan.serve, "sparse_index")
            return asked.long() - self.row_begin
        # on the communication stream: fc0's projection (current stream) runs beside it, the fetch of the rows follows it there
        plan.serve_local, _ = self._on_comm_stream(ids, ask, exchange_ids)
        rank_map = torch.cumsum(flags, 0, dtype=torch.int32) - 1
        ids_c = rank_map[ids.long().clamp_(0, N - 1)]
        sel_c = rank_map[sel.long().clamp_(0, N - 1)]
        return plan, ids_c, sel_c

    def _project_and_fetch(self, variant, X_loc, fc0_w, fc0_b, plan):
        """sparse counterpart of _project_and_gather: -> (Xh_loc, the compact table [T, H] of touched rows in ascending
        node order, ready)"""
        ops, comm = self.ops, self.comm
        Xh_loc = ops.project(variant, X_loc, fc0_w, fc0_b)

        def fetch():
            send = Xh_loc.index_select(0, plan.serve_local)
            got = comm.all_to_all_rows(send, plan.serve, plan.need, "sparse_Xh")
            if got.shape[0] == 0:       # a rank without masked nodes asks for nothing: the kernels still want a table (one zero row)
                got = torch.zeros((1, got.shape[1]), dtype=got.dtype, device=got.device)
            return got
        Xh, ready = self._on_comm_stream(X_loc, Xh_loc, fetch)
        return Xh_loc, Xh, ready

    def _return_sparse(self, g_Xh, plan, rows):
        """d Xh of the compact table -> the owners; this rank adds what it receives to its block, rank by rank (the rows one
        rank sends are distinct, so every index_add_ is a plain store-add and the order of the sums is fixed)"""
        got = self.comm.all_to_all_rows(g_Xh[:plan.T], plan.need, plan.serve, "sparse_dXh")
        g_loc = torch.zeros((rows, g_Xh.shape[1]), dtype=g_Xh.dtype, device=g_Xh.device)
        at = 0
        for q, n in enumerate(plan.serve):
            if n:
                g_loc.index_add_(0, plan.serve_local[at:at + n], got[at:at + n])
            at += n
        return g_loc

    @staticmethod
    def _norm_key(X_loc):
        return (X_loc.data_ptr(), X_loc._version, tuple(X_loc.shape), tuple(X_loc.stride()), X_loc.dtype)

    def begin_step(self, X_loc):
        """Start the step's projection and all-gather NOW, before the paths of the step are sampled: the walker (and, inside
        the aggregator call, the touched-row marking, the index plan and the weight packing) then run under the all-gather.
        Optional -- without it the call itself projects and gathers, and only the library's own preamble overlaps.
        Valid for the next call with the same X_loc and unchanged fc0 parameters (the optimizers -- torch's and
        pathnet_amd.Adam -- bump the parameters' version counters, which is what the check reads); no-grad bookkeeping only:
        the autograd graph is that of the call (fc0's backward runs on X_loc as always).  The step then uses the DENSE
        exchange (the sparse one needs the step's paths before it can ask for rows): with exchange="auto" a begin_step therefore
        MEANS dense for that step, with exchange="sparse" begin_step does nothing."""
        m = self.module
        if m.hidden_size % 32:
            return          # (the padded parameters are functions of the real ones, rebuilt per call: nothing to key on)
        if self.exchange == "sparse":
            return          # (the sparse exchange asks for rows by the step's paths: nothing can start before they exist)
        Xn = X_loc.contiguous().float()         # normalised ONCE: the call compares against this very tensor
        with torch.no_grad():
            pre = self._project_and_gather(m.variant, Xn, m.fc0.weight, m.fc0.bias)
        self._prefetched = ((self._norm_key(X_loc), m.fc0.weight.data_ptr(), m.fc0.weight._version, m.fc0.bias._version),
                            Xn, pre)

    def _prefetched_input(self, X_loc):
        """the normalised X of a matching begin_step (so that _take_prefetched sees the tensor begin_step projected), or None"""
        pre, m = self._prefetched, self.module
        if pre is None:
            return None
        key = (self._norm_key(X_loc), m.fc0.weight.data_ptr(), m.fc0.weight._version, m.fc0.bias._version)
        if pre[0] != key:
            self._prefetched = None
            if self.exchange == "auto" and self.comm.active():
                # With "auto" a matching prefetch means the dense collectives, none means the (collective) sparse plan: a rank
                # that silently dropped a stale prefetch would enter other collectives than its peers and hang the group
                # until the process-group timeout.  Stale = begin_step saw another X, or fc0 changed since.
                raise RuntimeError("ShardedAggregator(exchange='auto'): begin_step() was called for another X_loc or the fc0 "
                                   "parameters changed since; call begin_step right before the step it belongs to")
            return None
        return pre[1]

    def _take_prefetched(self, Xn, fc0_w, fc0_b):
        pre, self._prefetched = self._prefetched, None
        if pre is None or pre[1] is not Xn:
            return None
        return pre[2]

    def __call__(self, X_loc, neis, num_w, walk_len, sel_global, layer_type):
        """X_loc [row_count, F]; sel_global: global node ids of this rank's masked nodes (inside its block);
        neis / layer_type: their paths [S, W*L] / [S, W, L] with GLOBAL node ids.  -> logits [S, C]."""
        m = self.module
        dev = X_loc.device
        ids, codes, sel, S = M._as_index_tensors(neis, layer_type, sel_global, num_w, walk_len, dev, n_nodes=self.n_total)
        H = m.hidden_size
        Hk = -(-H // 32) * 32       # the kernels' hidden size; other sizes are zero-padded exactly as the single-GPU module does
        fw, fb, params = m._param_inputs() if Hk == H else m._padded_param_inputs(Hk)
        p = m.dropout_p() if m.training else 0.0
        comm = self.comm
        multi = comm.active()           # several ranks (or a one-rank group forced through the multi-rank path)
        if not multi:
            counts = [S]
        elif self._fixed_counts is not None:
            counts = self._fixed_counts
            if counts[comm.rank()] != S:
                raise ValueError("set_batch_counts said %d masked nodes for this rank, the call has %d"
                                 % (counts[comm.rank()], S))
        else:
            counts = comm.counts(S, dev)        # (a host round trip: see set_batch_counts)
        self.batch_counts = counts
        S_total, begin = sum(counts), sum(counts[:comm.rank()])
        local_rows = True
        if multi and m.variant == "hetero":
            # the [W, S] re-view reads other ranks' paths: every rank gets the whole batch's index arrays
            ids, codes, sel = (comm.all_gather_ragged(t, counts) for t in (ids, codes, sel))
            local_rows = False
        Xn = self._prefetched_input(X_loc)
        plan = None
        if Xn is None:
            Xn = X_loc.contiguous().float()
            plan, ids, sel = self._plan_sparse(ids, sel)
        self.last_exchange = None if not multi else "sparse" if plan is not None else "dense"
        # one seed per step for all ranks (positions in the batch separate their masks)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, generator=self._seed_gen).item()) if p > 0 else 0
        n_table = plan.T if plan is not None else (self.n_pad if multi else self.n_total)
        cfg = dict(variant=m.variant, N=max(n_table, 1), F=X_loc.shape[1], H=Hk, C=m.out_size, S=S,
                   W=int(num_w), L=int(walk_len), p_seq=p, p_cls=p, bank_w=fw, bank_b=fb, seed=seed,
                   S_total=S_total, group_begin=begin, index_rows_local=local_rows and multi, cell=m._cell_kind,
                   mask_seq=None, mask_cls=None, sparse_plan=plan,
                   deterministic=M.deterministic_default() if m.deterministic is None else bool(m.deterministic),
                   compact=M.compact_default(),
                   seq_math=M.seq_math_default() if m.seq_math is None else M._SEQ_MATH[m.seq_math])
        if not multi:
            cfg["S_total"], cfg["group_begin"] = 0, 0
        if m.training and (self.mask_seq is not None or self.mask_cls is not None):
            ms, mc = self.mask_seq, self.mask_cls
            if Hk != H:         # explicit masks are given for the module's own hidden size
                P = torch.nn.functional.pad
                ms = P(ms, (0, Hk - H)).contiguous() if ms is not None else None
                mc = P(mc.view(mc.shape[0], 2, H), (0, Hk - H)).reshape(mc.shape[0], 2 * Hk).contiguous() if mc is not None else None
            if (ms is not None and isinstance(self.ops, HipOps) and cfg["seq_math"] != _lib.SEQ_MATH_BF16X3
                    and float(ms.abs().max()) > 16.0):
                raise ValueError("explicit sequence masks must satisfy |m| <= 16 with seq_math f16x2 (include/pathnet_hip.h); "
                                 "use seq_math='bf16x3' for others")
            cfg["mask_seq"], cfg["mask_cls"] = ms, mc
            cfg["p_seq"] = cfg["p_cls"] = 0.0
        cfg["batch_groups"] = M.pick_batch_groups(m.variant, cfg["N"], cfg["F"], cfg["H"], cfg["C"], S, cfg["W"],
                                                  cfg["L"], m.workspace_budget, cell=m._cell_kind,
                                                  deterministic=cfg["deterministic"], S_total=cfg["S_total"],
                                                  group_begin=cfg["group_begin"], compact=cfg["compact"],
                                                  seq_math=cfg["seq_math"]) if isinstance(self.ops, HipOps) else 0
        return _ShardedFn.apply(self, cfg, Xn, ids, codes, sel, *params)

    def set_batch_counts(self, counts):
        """Masked nodes per rank, when they are the same every step (a fixed train mask): spares the per-step
        all-gather of the counts and its host round trip.  None returns to asking every step."""
        self._fixed_counts = None if counts is None else [int(c) for c in counts]

    # ---- loss over the whole batch ------------------------------------------------------------------------------------
    def loss_scale(self):
        """Factor for this rank's MEAN loss over its own S_r masked nodes so that the summed gradients are those of
        the mean over the whole batch (PathNet_run.py:297, :346): S_r / S_total.  Use with allreduce_grads(average=False);
        with equal S_r on every rank it equals 1/world, i.e. average=True."""
        counts = self.batch_counts or [1]
        tot = sum(counts)
        return counts[self.comm.rank()] / tot if tot else 0.0

    # ---- parameter gradients: one persistent flat buffer, .grad tensors are views of it ------------------------------
    def flat_grads(self):
        ps = [q for q in self.module.parameters() if q.requires_grad]
        key = tuple((id(q), q.data_ptr()) for q in ps)
        if self._flat is None or self._flat_ids != key:
            n = sum(q.numel() for q in ps)
            self._flat = torch.zeros(n, dtype=ps[0].dtype, device=ps[0].device)
            self._flat_ids = key
        at = 0
        views = []
        for q in ps:
            v = self._flat[at:at + q.numel()].view_as(q)
            at += q.numel()
            views.append((q, v))
        return views

    def allreduce_grads(self, average=True):
        """One all-reduce of every parameter gradient (single bucket): the gradients autograd produced this step are
        copied into the persistent flat buffer by one multi-tensor launch, the buffer is all-reduced in place and
        every .grad becomes a view of it (use optimizer.zero_grad(set_to_none=True)).  average=True divides by the world size:
        right when every rank's loss is a mean over equally many masked nodes; otherwise scale the local loss
        by loss_scale() and pass average=False."""
        if not self.distributed:
            return
        views = self.flat_grads()
        src, dst = [], []
        for q, v in views:
            if q.grad is None:
                v.zero_()
            elif q.grad.data_ptr() != v.data_ptr():      # autograd handed out a fresh tensor this step
                src.append(q.grad)
                dst.append(v)
            q.grad = v
        if src:
            torch._foreach_copy_(dst, src)                # one multi-tensor launch into the flat buffer
        self.comm.all_reduce(self._flat)
        if average:
            self._flat /= self.comm.world()


class ReplicatedAggregator(ShardedAggregator):
    """Plain data parallelism over the masked nodes: every rank holds ALL rows of X and runs the single-GPU aggregator on its
    own masked nodes; the only collective of a step is the all-reduce of the flat gradient buffer (plus, for the hetero
    class, the all-gather of the batch's index arrays -- its rows read paths of other masked nodes).

    When it beats node sharding: ShardedAggregator exchanges Xh and d Xh (2 x N x H x 4 bytes per step, 1/R of it per link),
    this class instead runs fc0 over all N rows on every rank (2 N F H flops, forward and backward).  At F = 128 (the
    10 M-node configuration) the replicated projection costs ~11 ms per step against ~26 ms of exchange on 8 ranks
    (tools/scale_model.py); at F = 1433 (Cora) the exchange is the cheaper one.  `cheaper_than_sharding` does the
    arithmetic.  Results are those of one process on the whole batch (dropout positions are batch-wide; one seed per
    step from identically seeded generators)."""

    def __init__(self, module, n_total, group=None, comm=None, dropout_seed=0):
        self.module, self.n_total, self.row_begin, self.row_count = module, int(n_total), 0, int(n_total)
        self.comm = comm if comm is not None else Comm(group)
        self.group = self.comm.group
        self.ops = None
        self.distributed = self.comm.active()
        self._flat = None
        self._flat_ids = None
        self.batch_counts = None
        self._fixed_counts = None
        self._seed_gen = torch.Generator()
        self._seed_gen.manual_seed(int(dropout_seed))
        self.mask_seq = self.mask_cls = None
        self.compact_nodes = None       # True / False: run on the rows of X the call's paths can touch; None: when they are few

    def _touched_rows(self, X, ids, sel):
        """(X', ids', sel'): the call restricted to the nodes its paths and masked nodes name.  A rank's share of the batch
        touches a fraction of a large graph (12 500 masked nodes x 40 paths x 6 steps reach ~26 % of the 10 M nodes of
        configs[4]); without this every rank projects, zero-fills and flags ALL N rows each step -- the terms that do not
        shrink with the number of ranks (tools/scale_model.py).  Rows keep their relative order; every row's arithmetic
        is unchanged (the projection of a row does not depend on the other rows).  Device-resident ids / sel are clamped to
        the graph first -- the contract of the kernels, which never read them back (modules._as_index_tensors).  One host
        round trip per step (the number of touched rows sizes X'): a step that is to be captured in a hipGraph
        (module.step_state set) keeps all rows instead."""
        N = X.shape[0]
        steps = ids.numel() + sel.numel()
        on = self.compact_nodes if self.compact_nodes is not None else (2 * steps < N and self.module.step_state is None)
        if not on or steps == 0:
            return X, ids, sel
        ids_l, sel_l = ids.long().clamp_(0, N - 1), sel.long().clamp_(0, N - 1)
        flags = torch.zeros(N, dtype=torch.bool, device=X.device)
        flags[ids_l.reshape(-1)] = True
        flags[sel_l] = True
        nodes = torch.nonzero(flags).flatten()                      # (one host round trip for the count)
        rank = torch.cumsum(flags, 0, dtype=torch.int32) - 1
        return X.index_select(0, nodes), rank[ids_l], rank[sel_l]

    def begin_step(self, X_loc):
        """nothing to start early: every rank projects its own copy of X inside the call, there is no forward collective"""
        return None

    @staticmethod
    def cheaper_than_sharding(n_total, in_features, hidden, world, link_GBs=50.0, gemm_TFLOPs=50.0):
        """fc0 forward + backward over all rows on every rank  vs  all-gather of Xh + reduce-scatter of d Xh."""
        replicated = 2 * 2.0 * n_total * in_features * hidden * (1.0 - 1.0 / world) / (gemm_TFLOPs * 1e12)
        exchange = 2 * (n_total / world) * hidden * 4 / (link_GBs * 1e9)
        return replicated < exchange

    def __call__(self, X, neis, num_w, walk_len, sel_global, layer_type):
        """X [n_total, F] (the same on every rank); sel_global: this rank's masked nodes; neis / layer_type: their paths.
        -> logits [S, C] of this rank's masked nodes."""
        m = self.module
        dev = X.device
        comm = self.comm
        ids, codes, sel, S = M._as_index_tensors(neis, layer_type, sel_global, num_w, walk_len, dev, n_nodes=self.n_total)
        if not comm.active():
            self.batch_counts = [S]
            X, ids, sel = self._touched_rows(X, ids, sel)
            return m._run(X, ids, num_w, walk_len, sel, codes)
        if self._fixed_counts is not None:
            counts = self._fixed_counts
            if counts[comm.rank()] != S:
                raise ValueError("set_batch_counts said %d masked nodes for this rank, the call has %d" % (counts[comm.rank()], S))
        else:
            counts = comm.counts(S, dev)
        self.batch_counts = counts
        S_total, begin = sum(counts), sum(counts[:comm.rank()])
        p = m.dropout_p() if m.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, generator=self._seed_gen).item()) if p > 0 else None
        old = m._mask_seq, m._mask_cls
        if self.mask_seq is not None or self.mask_cls is not None:
            m._mask_seq, m._mask_cls = self.mask_seq, self.mask_cls
        try:
            if m.variant == "hetero":
                ids, codes, sel = (comm.all_gather_ragged(t, counts) for t in (ids, codes, sel))
                X, ids, sel = self._touched_rows(X, ids, sel)
                return m._run(X, ids, num_w, walk_len, sel, codes, group_slice=(begin, S), seed=seed)
            X, ids, sel = self._touched_rows(X, ids, sel)
            return m._run(X, ids, num_w, walk_len, sel, codes, batch_position=(S_total, begin), seed=seed)
        finally:
            m._mask_seq, m._mask_cls = old

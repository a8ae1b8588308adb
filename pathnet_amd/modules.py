"""Drop-in aggregator modules with the reference's constructor / forward surface and state_dict keys.

  PathNet       <- /root/reference/PathNet_run.py:150-211   (heterophilous datasets)
  PathNet_homo  <- /root/reference/PathNet_run.py:214-278   (cora / citeseer / pubmed)
  PAGG          <- /root/reference/baseline/GPRGNN/src/copy.py:299-359

``forward(X, neis, num_w, walk_len, indices, layer_type, indxx)`` keeps the reference's argument
meaning (PathNet_run.py:172): X fp32 [N, F] on the GPU, neis [S, W*L] node ids, indices a bool mask
[N] (numpy or torch), layer_type [S, W, L] distance codes, indxx ignored (it is arange(S*W*L)).
All arithmetic runs in hand-written HIP kernels through libpathnet_hip.so; there is no PyTorch
fallback -- if the library is missing, construction fails.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib

# The reference reads the module-level global `dropout` inside forward (PathNet_run.py:71, :194).
# A training script can set pathnet_amd.modules.dropout the same way, or pass dropout= to the ctor.
dropout = 0.7

_VARIANT = {"hetero": _lib.VARIANT_HETERO, "homo": _lib.VARIANT_HOMO, "pagg": _lib.VARIANT_PAGG}
# path encoders: the classes' own (lstm / rnn) and the ablation rows of the paper's table (gru / mean / sum, README.md:118)
_CELL = {None: _lib.CELL_DEFAULT, "lstm": _lib.CELL_LSTM, "rnn": _lib.CELL_RNN, "gru": _lib.CELL_GRU,
         "mean": _lib.CELL_MEAN, "sum": _lib.CELL_SUM}
_HEAD_PARAMS = ("fc0_w", "fc0_b", "w_ih", "w_hh", "b_ih", "b_hh", "att_w", "att_b", "fc2_w", "fc2_b")


# Per-path tensors of one aggregator call (saved gates, [x|h] rows, gate gradients: ~6 KB per path step) are kept
# below this many bytes: a batch with more paths is walked in micro-batches inside the library (pn_pagg_shape
# .batch_groups), the backward re-running each micro-batch's recurrence.  Cora / Pubmed-size batches fit in one.
WORKSPACE_BUDGET_BYTES = 48 << 30
MIN_BATCH_BYTES = 8 << 30           # floor of the per-micro-batch part (pick_batch_groups)


def default_budget(device=None):
    """Workspace budget when the module has none of its own: WORKSPACE_BUDGET_BYTES, but never more than 60 % of the
    HBM that is free on `device` right now (a smaller GPU, or one shared with other tensors, then gets micro-batches
    instead of an out-of-memory error)."""
    try:
        free, _ = torch.cuda.mem_get_info(device)
        return int(min(WORKSPACE_BUDGET_BYTES, max(0.6 * free, MIN_BATCH_BYTES)))
    except Exception:       # noqa: BLE001  (no CUDA context yet / CPU-only import)
        return WORKSPACE_BUDGET_BYTES


_SEQ_MATH = {None: _lib.SEQ_MATH_DEFAULT, "": _lib.SEQ_MATH_DEFAULT, "f16x2": _lib.SEQ_MATH_F16X2, "bf16x3": _lib.SEQ_MATH_BF16X3}


def compact_default():
    """pn_pagg_shape.compact when a call does not say: PN_COMPACT=1 / 0 in the environment forces the touched-row
    compaction of the distance bank on / off (tests, A/B runs); otherwise the library decides from the shape.  Read when a
    call is set up -- the value then travels with the call's shape through workspace size, forward and backward."""
    e = os.environ.get("PN_COMPACT")
    if e is None or e == "":
        return _lib.COMPACT_AUTO
    return _lib.COMPACT_ON if int(e) != 0 else _lib.COMPACT_OFF


RANGE_WINDOW_BITS = 18      # spread (bits) of the gathered rows' magnitudes the fp16 x 2 kernels carry at full precision
RANGE_POLL_EVERY = 16       # calls between two asynchronous read-backs of the range record


def seq_math_default():
    """pn_pagg_shape.seq_math when a module does not say: PN_SEQ_MATH=bf16x3 selects rounds 1-3's six-MFMA bf16 products
    for the recurrent GEMMs, the default (f16x2) is three fp16 MFMAs over scaled two-plane splits (include/pathnet_hip.h)."""
    e = os.environ.get("PN_SEQ_MATH", "").strip().lower()
    if e not in _SEQ_MATH:
        raise ValueError("PN_SEQ_MATH=%r: f16x2 or bf16x3" % e)
    return _SEQ_MATH[e]


def _shape(variant, N, F, H, C, S, W, L, S_total=0, group_begin=0, batch_groups=0, cell=None, deterministic=False,
           compact=None, seq_math=None):
    return _lib.PaggShape(_VARIANT[variant], N, F, H, C, S, W, L, S_total, group_begin, batch_groups, _CELL[cell],
                          1 if deterministic else 0, compact_default() if compact is None else int(compact),
                          seq_math_default() if seq_math is None else int(seq_math))


def shape_info(sh):
    """(compact, rows of Z, micro-batches, seq_math) the library derives from a pn_pagg_shape"""
    out = (ctypes.c_int64 * 4)()
    _lib.check(_lib.load().pn_pagg_shape_info(ctypes.byref(sh), out))
    return bool(out[0]), int(out[1]), int(out[2]), int(out[3])


def deterministic_default():
    """The backward's mode when a module does not say: torch.use_deterministic_algorithms(True) or PN_DETERMINISTIC=1
    select the fixed-order backward (pn_pagg_shape.deterministic: bitwise reproducible gradients, no fp32 atomics)."""
    return torch.are_deterministic_algorithms_enabled() or os.environ.get("PN_DETERMINISTIC", "0") not in ("", "0")


def _cfg_shape(cfg):
    return _shape(cfg["variant"], cfg["N"], cfg["F"], cfg["H"], cfg["C"], cfg["S"], cfg["W"], cfg["L"],
                  cfg.get("S_total", 0), cfg.get("group_begin", 0), cfg.get("batch_groups", 0), cfg.get("cell"),
                  cfg.get("deterministic", False), cfg.get("compact"), cfg.get("seq_math"))


def workspace_bytes(variant, N, F, H, C, S, W, L, S_total=0, group_begin=0, batch_groups=0, cell=None, deterministic=False,
                    compact=None, seq_math=None):
    n = ctypes.c_int64(0)
    sh = _shape(variant, N, F, H, C, S, W, L, S_total, group_begin, batch_groups, cell, deterministic, compact, seq_math)
    _lib.check(_lib.load().pn_pagg_workspace_bytes(ctypes.byref(sh), ctypes.byref(n)))
    return n.value


def pick_batch_groups(variant, N, F, H, C, S, W, L, budget=None, cell=None, device=None, deterministic=False,
                      S_total=0, group_begin=0, compact=None, seq_math=None):
    """0 when the whole batch fits the workspace budget, else the largest micro-batch (in masked nodes) that does.
    budget=None: default_budget(device) -- queried only when the batch needs more than MIN_BATCH_BYTES, so that ordinary
    steps make no runtime call.
    The workspace of a call that runs b masked nodes at a time is fixed(S) + b * per_group: the node tables -- whose rows
    depend on the WHOLE call's S (and S_total for a slice of a hetero batch) when the bank runs over compact rows -- plus
    the per-path tensors of one micro-batch.  Both terms are measured on this call's own shape (batch_groups = 1 and
    1 + step), and the pick is checked against the budget afterwards."""
    kw = dict(cell=cell, deterministic=deterministic, S_total=S_total, group_begin=group_begin, compact=compact, seq_math=seq_math)

    def ws(bg):
        return workspace_bytes(variant, N, F, H, C, S, W, L, batch_groups=bg, **kw)
    need = ws(0) if S > 1 else 0
    floor = MIN_BATCH_BYTES if budget is None else 0       # an explicit budget is kept to the byte
    if budget is None:
        if need <= MIN_BATCH_BYTES:
            return 0
        budget = default_budget(device)
    budget = int(budget)
    if S <= 1 or need <= budget:
        return 0
    step = min(1024, S - 1)
    w1 = ws(1)
    per_group = max((ws(1 + step) - w1) // step, 1) if step > 0 else max(need, 1)
    fixed = w1 - per_group                  # the node tables (and the compact-row arrays): needed whatever the micro-batch
    avail = max(budget - fixed, floor)      # default budget: graphs whose tables alone exceed it still get real batches
    bg = int(max(1, min(S, avail // per_group if avail > 0 else 1)))
    if floor == 0:                          # an explicit budget is kept to the byte: settle on the real layout (alignment,
        while bg > 1 and ws(bg) > budget:   # terms that are not linear in the micro-batch such as the sort's scratch)
            bg = max(1, min(bg - 1, int(bg * 0.98)))
        for _ in range(16):
            if bg >= S or ws(bg + 1) > budget:
                break
            bg += 1
    return bg


def _cfg_workspace_bytes(cfg):
    return workspace_bytes(cfg["variant"], cfg["N"], cfg["F"], cfg["H"], cfg["C"], cfg["S"], cfg["W"], cfg["L"],
                           cfg.get("S_total", 0), cfg.get("group_begin", 0), cfg.get("batch_groups", 0), cfg.get("cell"),
                           cfg.get("deterministic", False), cfg.get("compact"), cfg.get("seq_math"))


def _split_params(params, L):
    """Function inputs -> dict of the tensors the C ABI wants.  `params` is
    (fc0_w, fc0_b, w_ih, w_hh, b_ih, b_hh, att_w, att_b, fc2_w, fc2_b, bank_w_0..L-1, bank_b_0..L-1);
    the bank entries are views of one contiguous [L,H,H] / [L,H] buffer (see _Aggregator._bank)."""
    names = ("fc0_w", "fc0_b", "w_ih", "w_hh", "b_ih", "b_hh", "att_w", "att_b", "fc2_w", "fc2_b")
    p = dict(zip(names, params[:10]))
    p["bank_ws"], p["bank_bs"] = params[10:10 + L], params[10 + L:10 + 2 * L]
    return p


class _PaggFunction(torch.autograd.Function):
    """forward/backward through pn_pagg_forward / pn_pagg_backward."""

    @staticmethod
    def _args(cfg, X, ids, codes, sel, p):
        a = _lib.PaggArgs()
        a.shape = _cfg_shape(cfg)
        a.X, a.ids, a.codes, a.sel = X.data_ptr(), ids.data_ptr(), codes.data_ptr(), sel.data_ptr()
        for k in ("fc0_w", "fc0_b", "w_ih", "w_hh", "b_ih", "b_hh", "att_w", "att_b", "fc2_w", "fc2_b"):
            setattr(a, k, p[k].data_ptr() if p[k] is not None else None)
        a.bank_w, a.bank_b = cfg["bank_w"].data_ptr(), cfg["bank_b"].data_ptr()
        a.p_seq, a.p_cls, a.seed = cfg["p_seq"], cfg["p_cls"], cfg["seed"]
        ms, mc = cfg.get("mask_seq"), cfg.get("mask_cls")
        a.mask_seq = ms.data_ptr() if ms is not None else None
        a.mask_cls = mc.data_ptr() if mc is not None else None
        st = cfg.get("step_state")
        a.step_state = st.ptr() if st is not None else None
        a.index_rows_local = 1 if cfg.get("index_rows_local") else 0
        return a

    @staticmethod
    def forward(ctx, cfg, X, ids, codes, sel, *params):
        lib = _lib.load()
        p = _split_params(params, cfg["L"])
        dev = X.device
        nbytes = _cfg_workspace_bytes(cfg)
        ws = cfg.get("workspace")
        with torch.cuda.device(dev):        # the library launches on the current device
            out = torch.empty((cfg["S"], cfg["C"]), dtype=torch.float32, device=dev)
            if ws is None or ws.numel() < nbytes or ws.device != dev:
                ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
            a = _PaggFunction._args(cfg, X, ids, codes, sel, p)
            a.out = out.data_ptr()
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            a.no_save = 0 if cfg.get("grad", True) else 1       # torch.no_grad() forwards skip the saved tensors
            a.reuse_tables = int(cfg.get("reuse_tables") or 0)     # 1: Xh and the dense bank are valid, 2: Xh only
            if cfg["S"] > 0:
                _lib.check(lib.pn_pagg_forward(_lib.context(dev), ctypes.byref(a), _lib.stream_ptr(dev)))
        ctx.cfg, ctx.ws = cfg, ws
        cfg["ws_used"] = ws         # (the module's range guard reads the call's pn_seq_range back from it)
        ctx.present = [t is not None for t in params]
        ctx.save_for_backward(X, ids, codes, sel, *[t for t in params if t is not None])
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        cfg = ctx.cfg
        saved = list(ctx.saved_tensors)
        X, ids, codes, sel = saved[:4]
        it = iter(saved[4:])
        params = [next(it) if pres else None for pres in ctx.present]
        L = cfg["L"]
        p = _split_params(params, L)
        dev = X.device
        with torch.cuda.device(dev):
            g_out = g_out.contiguous().float()
            a = _PaggFunction._args(cfg, X, ids, codes, sel, p)
            grads = {}
            for k in ("fc0_w", "fc0_b", "w_ih", "w_hh", "b_ih", "b_hh", "att_w", "att_b", "fc2_w", "fc2_b"):
                if p[k] is not None:
                    grads[k] = torch.empty_like(p[k])
                    setattr(a, "g_" + k, grads[k].data_ptr())
            g_bank_w, g_bank_b = torch.empty_like(cfg["bank_w"]), torch.empty_like(cfg["bank_b"])
            a.g_bank_w, a.g_bank_b = g_bank_w.data_ptr(), g_bank_b.data_ptr()
            gX = torch.empty_like(X) if ctx.needs_input_grad[1] else None
            a.g_X = gX.data_ptr() if gX is not None else None
            a.workspace, a.workspace_bytes = ctx.ws.data_ptr(), ctx.ws.numel()
            a.g_out = g_out.data_ptr()
            if cfg["S"] > 0:
                _lib.check(lib.pn_pagg_backward(_lib.context(dev), ctypes.byref(a), _lib.stream_ptr(dev)))
            else:
                for g in list(grads.values()) + [g_bank_w, g_bank_b]:
                    g.zero_()
                if gX is not None:
                    gX.zero_()
        head = tuple(grads.get(k) for k in ("fc0_w", "fc0_b", "w_ih", "w_hh", "b_ih", "b_hh", "att_w", "att_b",
                                            "fc2_w", "fc2_b"))
        return (None, gX, None, None, None) + head + tuple(g_bank_w[d] for d in range(L)) + tuple(
            g_bank_b[d] for d in range(L))


class _PaggLossFunction(torch.autograd.Function):
    """loss, logits = one pn_pagg_train_step: forward, softmax cross entropy and backward in one library call (what it
    saves: the second forward of every micro-batch, include/pathnet_hip.h).  The gradients are computed here and handed
    out by backward(), scaled by the upstream gradient of the loss."""

    @staticmethod
    def forward(ctx, cfg, X, ids, codes, sel, target, *params):
        lib = _lib.load()
        L = cfg["L"]
        p = _split_params(params, L)
        dev = X.device
        nbytes = _cfg_workspace_bytes(cfg)
        ws = cfg.get("workspace")
        with torch.cuda.device(dev):
            out = torch.empty((cfg["S"], cfg["C"]), dtype=torch.float32, device=dev)
            # (pn_pagg_train_step stores the loss -- or clears it itself before accumulating over micro-batches: no fill here)
            loss = (torch.empty if cfg["S"] > 0 else torch.zeros)((), dtype=torch.float32, device=dev)
            if ws is None or ws.numel() < nbytes or ws.device != dev:
                ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
            a = _PaggFunction._args(cfg, X, ids, codes, sel, p)
            a.out = out.data_ptr()
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            # every parameter gradient is a view of one flat buffer: backward() scales them by the upstream gradient of
            # the loss with one launch instead of one per tensor
            keys = [k for k in ("fc0_w", "fc0_b", "w_ih", "w_hh", "b_ih", "b_hh", "att_w", "att_b", "fc2_w", "fc2_b")
                    if p[k] is not None]
            shapes = [p[k].shape for k in keys] + [cfg["bank_w"].shape, cfg["bank_b"].shape]
            sizes = [-(-int(np.prod(sh)) // 4) * 4 for sh in shapes]         # 16-byte aligned pieces
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            pieces = [v[:int(np.prod(sh))].view(sh) for v, sh in zip(flat.split(sizes), shapes)]
            grads = dict(zip(keys, pieces))
            for k in keys:
                setattr(a, "g_" + k, grads[k].data_ptr())
            g_bank_w, g_bank_b = pieces[-2], pieces[-1]
            a.g_bank_w, a.g_bank_b = g_bank_w.data_ptr(), g_bank_b.data_ptr()
            gX = torch.empty_like(X) if ctx.needs_input_grad[1] else None
            a.g_X = gX.data_ptr() if gX is not None else None
            target = target.to(device=dev, dtype=torch.int64).contiguous()
            if cfg["S"] > 0:
                _lib.check(lib.pn_pagg_train_step(_lib.context(dev), ctypes.byref(a), target.data_ptr(),
                                                  float(cfg["grad_scale"]), loss.data_ptr(), _lib.stream_ptr(dev)))
            else:
                flat.zero_()
                if gX is not None:
                    gX.zero_()
        head = tuple(grads.get(k) for k in ("fc0_w", "fc0_b", "w_ih", "w_hh", "b_ih", "b_hh", "att_w", "att_b", "fc2_w", "fc2_b"))
        ctx.grads = (gX,) + head + tuple(g_bank_w[d] for d in range(L)) + tuple(g_bank_b[d] for d in range(L))
        ctx.flat = flat
        cfg["ws_used"] = ws
        ctx.mark_non_differentiable(out)
        # (without this autograd materialises a zero gradient for `out` -- an [S, C] fill launch in front of the optimizer,
        #  on the step's critical path: profiles/r06_glue.txt)
        ctx.set_materialize_grads(False)
        return loss, out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss, _g_out):
        # (the context lets go of the tensors: with a second owner alive autograd's AccumulateGrad would copy every
        #  gradient into a fresh tensor instead of adopting it -- 18 copy launches per step)
        grads, flat = ctx.grads, ctx.flat
        ctx.grads = ctx.flat = None
        if grads is None:
            raise RuntimeError("forward_loss: the gradients were handed out already (backward twice)")
        # (d loss' / d loss: 1 for loss.backward().  The gradients were computed once, in forward(): a single backward, no
        #  double backward -- once_differentiable says so to autograd; a second backward raises above)
        from . import optim
        if g_loss is None:      # (set_materialize_grads(False): the loss took no part in what is being differentiated)
            return (None,) * (6 + len(grads) - 1)
        one = optim._UNIT.get(g_loss.device)
        if one is None or g_loss.data_ptr() != one.data_ptr():       # (optim.backward(loss) seeds with the cached 1.0: nothing to scale)
            flat.mul_(g_loss)
            if grads[0] is not None:
                grads[0].mul_(g_loss)
        return (None, grads[0], None, None, None, None) + tuple(grads[1:])


def _as_index_tensors(neis, layer_type, indices, num_w, walk_len, device, n_nodes=None):
    """Reference argument conventions -> device int32 ids [S,W,L], uint8 codes, int32 sel [S].
    `indices`: a bool mask over the nodes (numpy or torch: the reference's two conventions) or the node ids
    themselves (any integer dtype, numpy or torch)."""
    if isinstance(indices, np.ndarray):
        if indices.dtype == np.bool_:
            sel = torch.from_numpy(np.flatnonzero(indices).astype(np.int32))
        elif np.issubdtype(indices.dtype, np.integer):
            sel = torch.from_numpy(np.ascontiguousarray(indices).reshape(-1).astype(np.int32))
        else:
            raise TypeError("indices: bool mask or integer node ids expected, got numpy %s" % indices.dtype)
    else:
        idx = torch.as_tensor(indices)
        if idx.dtype == torch.bool:
            sel = torch.nonzero(idx, as_tuple=False).flatten().to(torch.int32)
        elif not idx.dtype.is_floating_point and not idx.dtype.is_complex:
            sel = idx.flatten().to(torch.int32)
        else:
            raise TypeError("indices: bool mask or integer node ids expected, got %s" % idx.dtype)
    S = int(sel.numel())
    ids = torch.as_tensor(neis)
    codes = torch.as_tensor(layer_type)
    if ids.numel() != S * num_w * walk_len or codes.numel() != S * num_w * walk_len:
        raise ValueError("neis / layer_type hold %d / %d entries, %d masked nodes x %d paths x %d steps = %d expected"
                         % (ids.numel(), codes.numel(), S, num_w, walk_len, S * num_w * walk_len))
    if n_nodes is not None and S and sel.device.type == "cpu":      # (device-resident indices are not read back: the
                                                                    #  kernels clamp ids, codes and sel to the tables)
        lo, hi = int(sel.min()), int(sel.max())
        if lo < 0 or hi >= n_nodes:
            raise IndexError("indices name node %d, the feature matrix has %d rows" % (lo if lo < 0 else hi, n_nodes))
    if ids.dtype != torch.int32:
        ids = ids.to(torch.int32)
    if codes.dtype != torch.uint8:
        codes = codes.to(torch.uint8)
    ids = ids.reshape(S, num_w, walk_len).to(device, non_blocking=True).contiguous()
    codes = codes.reshape(S, num_w, walk_len).to(device, non_blocking=True).contiguous()
    return ids, codes, sel.to(device, non_blocking=True).contiguous(), S


class _Aggregator(nn.Module):
    variant = None

    def _common_init(self, feature_length, hidden_size, out_size, dropout_p, cell=None):
        _lib.load()     # fail at construction if the HIP library is missing
        if cell not in _CELL:
            raise ValueError("cell must be one of %s" % sorted(k for k in _CELL if k))
        # None: the class's own path encoder (its state_dict is the reference's); "gru" / "mean" / "sum" / "lstm" / "rnn":
        # the ablation rows of the paper's table ("Changing the PAGG class can deliver other variants", README.md:118)
        self.cell = cell
        self.feature_length, self.hidden_size, self.out_size = feature_length, hidden_size, out_size
        self._dropout = dropout_p
        self._ws_eval = None
        self._ws_tables = None            # (X address, shape, L) whose tables sit in _ws_eval
        self.step_state = None            # pathnet_amd.StepState: dropout seed read from device memory (hipGraph replay)
        self.workspace_budget = None      # bytes; None = modules.WORKSPACE_BUDGET_BYTES (see pick_batch_groups)
        self.deterministic = None         # True / False: fixed-order backward or not; None: deterministic_default()
        self.seq_math = None              # "f16x2" / "bf16x3": arithmetic of the recurrent GEMMs; None: seq_math_default()
        # Range guard of the default arithmetic (seq_math None -> f16x2).  Its two fp16 planes keep 22 bits of a gathered row
        # within ~2^18 of the largest one; a single spike row 2^20 times the typical row -- bag-of-words features after fc0
        # can do that -- costs the typical rows bits (include/pathnet_hip.h: pn_seq_range).  The forward's kernels leave the
        # maximum and the spread of the rows' magnitudes in device memory; the module reads them back -- synchronously after
        # its FIRST call (which is run again in bf16x3 when the spread is beyond the window), asynchronously every
        # RANGE_POLL_EVERY-th call afterwards -- and switches itself to bf16x3 for good once it has seen such an input.
        self.range_guard = True
        self._range_wide = False          # sticky: an input beyond the window was seen
        self._range_calls = 0
        self._range_pending = None        # (pinned int32[6], event) of a read-back in flight
        self.range_spread_bits = None     # last evaluated spread: exponent(max |Z|) - mean exponent of the sampled tiles
        self._mask_seq = None     # test hook: explicit dropout masks (reference order)
        self._mask_cls = None
        self._bank_flat = (None, None)

    # ---- dropout probability: ctor argument, else the module-level global like the reference -------
    def dropout_p(self):
        return float(dropout if self._dropout is None else self._dropout)

    def set_dropout(self, p):
        self._dropout = p

    def _bank_layers(self):
        raise NotImplementedError

    def _tables_key(self, X, L):
        """what the projected feature matrix and the distance bank in the eval workspace were computed from: X (address,
        shape, in-place version) and the versions of the fc0 / bank parameters (an optimizer step bumps them)"""
        lins = [self.fc0] + list(self._bank_layers())
        return (X.data_ptr(), tuple(X.shape), int(X._version), int(L),
                tuple(int(t._version) for l in lins for t in (l.weight, l.bias)))

    def _make_cell(self, default):
        """the recurrent sub-module, under the attribute name torch users expect (LSTM / RNN / GRU); none for mean / sum"""
        kind = self.cell or default
        H = self.hidden_size
        if kind == "lstm":
            self.LSTM = nn.LSTM(H, H)
        elif kind == "rnn":
            self.RNN = nn.RNN(H, H)
        elif kind == "gru":
            self.GRU = nn.GRU(H, H)
        self._cell_kind = kind

    def _cell(self):
        return {"lstm": getattr(self, "LSTM", None), "rnn": getattr(self, "RNN", None),
                "gru": getattr(self, "GRU", None)}.get(self._cell_kind)

    def _bank(self):
        """The L distance layers as ONE contiguous [L,H,H] / [L,H] pair (what the kernels read), without a
        per-step torch.stack: the layers' weight/bias Parameters are re-pointed to be views of two flat
        buffers.  Anything that re-creates the parameters (.to(), .cuda()) is detected by address and the
        buffers are rebuilt.  state_dict keys and Parameter identities are unchanged."""
        lins = self._bank_layers()
        H = self.hidden_size
        fw, fb = self._bank_flat
        ok = (fw is not None and fw.device == lins[0].weight.device and
              all(l.weight.data_ptr() == fw.data_ptr() + d * H * H * 4 and l.bias.data_ptr() == fb.data_ptr() + d * H * 4
                  for d, l in enumerate(lins)))
        if not ok:
            with torch.no_grad():
                fw = torch.stack([l.weight.detach() for l in lins]).contiguous()
                fb = torch.stack([l.bias.detach() for l in lins]).contiguous()
                for d, l in enumerate(lins):
                    l.weight.data = fw[d]
                    l.bias.data = fb[d]
            self._bank_flat = (fw, fb)
        return fw, fb, [l.weight for l in lins], [l.bias for l in lins]

    def _padded_param_inputs(self, Hk):
        """The same inputs for a hidden size that is not a multiple of 32 (`-hid` is any integer, PathNet_run.py:52), zero-padded
        to Hk units as differentiable functions of the parameters.  Exact: a padded unit's projected feature, bank row,
        gates' inputs and attention / classifier weights are 0, so its cell and hidden state stay 0 (sigmoid(0) * tanh(0))
        and it feeds nothing into the real units; autograd drops the padded gradient entries on the way back."""
        H, G = self.hidden_size, {"lstm": 4, "rnn": 1, "gru": 3}.get(self._cell_kind, 0)
        pad = Hk - H
        P = torch.nn.functional.pad

        def rows(w):            # [H, *] -> [Hk, *]
            return P(w, (0, 0) * (w.dim() - 1) + (0, pad))

        def gates(w):           # [G*H, H] -> [G*Hk, Hk]  /  [G*H] -> [G*Hk]
            if w.dim() == 2:
                return P(w.view(G, H, H), (0, pad, 0, pad)).reshape(G * Hk, Hk)
            return P(w.view(G, H), (0, pad)).reshape(G * Hk)

        def halves(w):          # [C, 2H] -> [C, 2Hk]  ([ego | pooled paths])
            return P(w.view(w.shape[0], 2, H), (0, pad)).reshape(w.shape[0], 2 * Hk)
        lins = self._bank_layers()
        ws = [P(l.weight, (0, pad, 0, pad)) for l in lins]
        bs = [P(l.bias, (0, pad)) for l in lins]
        cell = self._cell()
        att = getattr(self, "attw", None)
        rec = tuple(gates(t) for t in (cell.weight_ih_l0, cell.weight_hh_l0, cell.bias_ih_l0, cell.bias_hh_l0)) \
            if cell is not None else (None,) * 4
        head = (rows(self.fc0.weight), P(self.fc0.bias, (0, pad))) + rec + (
            halves(att.weight) if att is not None else None, att.bias if att is not None else None,
            halves(self.fc2.weight), self.fc2.bias)
        with torch.no_grad():
            fw, fb = torch.stack([w.detach() for w in ws]).contiguous(), torch.stack([b.detach() for b in bs]).contiguous()
        return fw, fb, head + tuple(ws) + tuple(bs)

    def _param_inputs(self):
        fw, fb, ws, bs = self._bank()
        cell = self._cell()
        att = getattr(self, "attw", None)
        rec = (cell.weight_ih_l0, cell.weight_hh_l0, cell.bias_ih_l0, cell.bias_hh_l0) if cell is not None else (None,) * 4
        head = (self.fc0.weight, self.fc0.bias) + rec + (
                att.weight if att is not None else None, att.bias if att is not None else None,
                self.fc2.weight, self.fc2.bias)
        return fw, fb, head + tuple(ws) + tuple(bs)

    def forward_loss(self, X, neis, num_w, walk_len, indices, layer_type, target, indxx=None, grad_scale=None, fused=None):
        """(loss, logits) of a training step's forward + torch.nn.CrossEntropyLoss() (PathNet_run.py:343-346); loss.backward()
        yields the parameter gradients.  target: class index of every masked node, in the order of the logits' rows.
        grad_scale: weight of a row's loss (default 1 / rows = the mean).
        fused=True: forward, loss and backward in ONE library call (pn_pagg_train_step; backward() hands out the gradients that
        call already computed) -- for a batch that runs in micro-batches this saves every micro-batch's second forward (the
        10 M-node configuration: 0.250 -> 0.222 s per step).  fused=False: the three calls, which is what a batch that fits
        the workspace is quicker with (measured 1.26 vs 1.29 ms per step at the headline shape).  None: fused exactly when
        the batch needs micro-batches.  Same values either way."""
        return self._run(X, neis, num_w, walk_len, indices, layer_type, target=target, grad_scale=grad_scale, fused=fused)

    def forward(self, X, neis, num_w, walk_len, indices, layer_type, indxx=None, reuse_tables=False, group_slice=None):
        return self._run(X, neis, num_w, walk_len, indices, layer_type, reuse_tables=reuse_tables, group_slice=group_slice)

    def paths_stream(self):
        """The torch stream on which this module's NEXT call will read its path arrays (neis / layer_type), judged by the
        shape of its last call on the current stream (pn_pagg_paths_stream): the library's second stream when the call forks
        the index plan off, else the current stream.  A training loop that enqueues its sampler there
        (``with torch.cuda.stream(model.paths_stream()): sampler.sample(..., out=bufs)``) takes the walk off the step's
        critical path -- it runs under the tail of the previous step -- without an event on the main stream.  The path
        buffers must then be written by that stream only."""
        cur = torch.cuda.current_stream()
        last = getattr(self, "_last_shape", None)
        if last is None:
            self._paths_stream_prev = cur
            return cur
        shape, dev = last
        out = ctypes.c_void_p()
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            _lib.check(_lib.load().pn_pagg_paths_stream(_lib.context(dev), ctypes.byref(shape), _lib.stream_ptr(dev), ctypes.byref(out)))
        new = cur if (not out.value or out.value == cur.cuda_stream) else torch.cuda.ExternalStream(out.value, device=dev)
        # the readers of the previous batch's paths ran on the stream handed out last time: when the answer changes (first
        # steps, pn_profile_configure, another batch shape) the new producer is ordered behind them once, with an event
        prev = getattr(self, "_paths_stream_prev", None)
        if prev is not None and prev.cuda_stream != new.cuda_stream:
            new.wait_stream(prev)
            new.wait_stream(cur)
        self._paths_stream_prev = new
        return new

    # ---- range guard (see _common_init) ---------------------------------------------------------------------------------
    def _range_eval(self, rec):
        x_bits, esum, cnt = int(rec[0]) & 0xFFFFFFFF, int(rec[1]), int(rec[2]) & 0xFFFFFFFF
        if cnt > 0 and x_bits:
            self.range_spread_bits = ((x_bits >> 23) & 255) - esum / cnt
            if self.range_spread_bits > RANGE_WINDOW_BITS and not self._range_wide:
                self._range_wide = True
                import warnings
                warnings.warn("pathnet_amd: the distance-bank rows span %.1f bits between their largest value and the typical row "
                              "(window of the fp16 x 2 recurrent kernels: %d); this module now runs seq_math='bf16x3'"
                              % (self.range_spread_bits, RANGE_WINDOW_BITS))

    def _range_poll(self):
        if self._range_pending is not None and self._range_pending[1].query():
            rec, _ = self._range_pending
            self._range_pending = None
            self._range_eval(rec)

    def _range_read(self, cfg, sync):
        ws = cfg.get("ws_used")
        off = ctypes.c_int64(-1)
        _lib.check(_lib.load().pn_pagg_range_offset(ctypes.byref(_cfg_shape(cfg)), ctypes.byref(off)))
        if ws is None or off.value < 0:
            return
        rec = torch.empty(6, dtype=torch.int32, pin_memory=True)
        rec.copy_(ws[off.value:off.value + 24].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        if sync:
            ev.synchronize()
            self._range_eval(rec)
        else:
            self._range_pending = (rec, ev)

    def _run(self, X, neis, num_w, walk_len, indices, layer_type, reuse_tables=False, group_slice=None, target=None,
             grad_scale=None, fused=None, batch_position=None, seed=None, _math=None):
        """reuse_tables (extension, inference only): X and the weights are those of the previous no-grad forward of this
        module -- the validation and the test forward of an epoch (PathNet_run.py:362, :378) -- so the projected
        feature matrix and the distance bank still sitting in the module's workspace are used again.
        group_slice=(begin, count) (extension): neis / indices / layer_type describe the whole batch as always, but only
        the logits of its masked nodes [begin, begin + count) are computed and returned -- exactly those rows of the
        whole-batch result, also for the hetero class whose rows read paths of other masked nodes of the batch."""
        dev = X.device
        if dev.type != "cuda":
            raise RuntimeError("pathnet_amd aggregators run on the GPU only (X is on %s); no CPU fallback" % dev)
        X = X.contiguous().float()
        ids, codes, sel, S = _as_index_tensors(neis, layer_type, indices, num_w, walk_len, dev, n_nodes=X.shape[0])
        H = self.hidden_size
        Hk = -(-H // 32) * 32           # the kernels' hidden size: H, or H zero-padded to the next multiple of 32
        fw, fb, params = self._param_inputs() if Hk == H else self._padded_param_inputs(Hk)
        training = self.training
        p = self.dropout_p() if training else 0.0
        math = _math if _math is not None else (seq_math_default() if self.seq_math is None else _SEQ_MATH[self.seq_math])
        guard = (_math is None and self.seq_math is None and self.range_guard and math != _lib.SEQ_MATH_BF16X3 and
                 self.step_state is None)
        if guard:
            self._range_poll()
            if self._range_wide:
                math, guard = _lib.SEQ_MATH_BF16X3, False
        cfg = dict(variant=self.variant, N=X.shape[0], F=X.shape[1], H=Hk, C=self.out_size, S=S,
                   W=int(num_w), L=int(walk_len), p_seq=p, p_cls=p, mask_seq=None, mask_cls=None, bank_w=fw, bank_b=fb,
                   step_state=self.step_state, cell=self._cell_kind,
                   deterministic=deterministic_default() if self.deterministic is None else bool(self.deterministic),
                   # decisions that shape the workspace are taken once per call and travel with its shape
                   compact=compact_default(), seq_math=math,
                   seed=(int(seed) if seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item()))
                   if (p > 0 and self.step_state is None) else 0)
        if batch_position is not None:
            # (data-parallel ranks, dist.ReplicatedAggregator) the S masked nodes of this call are the rows [begin, begin + S)
            # of a batch of S_total whose other rows are computed elsewhere; neis / indices / layer_type hold this call's
            # rows only.  Dropout counters and explicit masks are positions in the whole batch.  Not for the hetero class,
            # whose rows read paths of other masked nodes: give it the whole batch and a group_slice.
            if self.variant == "hetero" or group_slice is not None:
                raise ValueError("batch_position: homo / PAGG classes, without group_slice")
            S_total, begin = int(batch_position[0]), int(batch_position[1])
            if begin < 0 or begin + S > S_total:
                raise ValueError("batch_position (%d, %d): %d rows do not fit" % (S_total, begin, S))
            cfg["S_total"], cfg["group_begin"], cfg["index_rows_local"] = S_total, begin, True
        if group_slice is not None:
            begin, count = int(group_slice[0]), int(group_slice[1])
            if begin < 0 or count < 0 or begin + count > S:
                raise ValueError("group_slice (%d, %d) outside the batch of %d masked nodes" % (begin, count, S))
            cfg["S_total"], cfg["group_begin"], cfg["S"] = S, begin, count
            S = count
        if len(params) != 10 + 2 * cfg["L"]:
            raise ValueError("walk_len=%d but the module has %d distance layers" % (cfg["L"], (len(params) - 10) // 2))
        if training and (self._mask_seq is not None or self._mask_cls is not None):
            ms, mc = self._mask_seq, self._mask_cls
            if Hk != H:         # explicit masks are given for the module's own hidden size
                P = torch.nn.functional.pad
                ms = P(ms, (0, Hk - H)).contiguous() if ms is not None else None
                mc = P(mc.view(mc.shape[0], 2, H), (0, Hk - H)).reshape(mc.shape[0], 2 * Hk).contiguous() if mc is not None else None
            if ms is not None and cfg["seq_math"] != _lib.SEQ_MATH_BF16X3 and float(ms.abs().max()) > 16.0:
                # (test hook only: one device round trip) the fp16 kernels scale the gathered rows for masks up to 16 = 1 / (1 - 0.9375)
                raise ValueError("explicit sequence masks must satisfy |m| <= 16 with seq_math f16x2 (include/pathnet_hip.h); "
                                 "use seq_math='bf16x3' for others")
            cfg["mask_seq"], cfg["mask_cls"] = ms, mc
            cfg["p_seq"] = cfg["p_cls"] = 0.0
        cfg["grad"] = torch.is_grad_enabled()
        cfg["batch_groups"] = pick_batch_groups(self.variant, cfg["N"], cfg["F"], cfg["H"], cfg["C"], S, cfg["W"],
                                                cfg["L"], self.workspace_budget, cell=self._cell_kind, device=dev,
                                                deterministic=cfg["deterministic"], S_total=cfg.get("S_total", 0),
                                                group_begin=cfg.get("group_begin", 0), compact=cfg["compact"],
                                                seq_math=cfg["seq_math"])
        self._last_shape = (_cfg_shape(cfg), dev)       # (paths_stream: where the NEXT call of this shape reads its paths)
        if not cfg["grad"]:
            need = _cfg_workspace_bytes(cfg)
            fits = self._ws_eval is not None and self._ws_eval.numel() >= need and self._ws_eval.device == dev
            key = self._tables_key(X, cfg["L"])
            compact_now = shape_info(_cfg_shape(cfg))[0]
            # what the workspace holds: (key, dense) -- Xh for `key`, and with dense also the distance bank over all N * L
            # rows.  A compact call leaves only ITS batch's rows of the bank behind (a later dense call must not
            # read them as the full table), so it hands on Xh alone.
            have_key, have_dense = self._ws_tables if self._ws_tables is not None else (None, False)
            mode = 0
            if reuse_tables and fits and have_key == key:
                mode = 1 if (have_dense and not compact_now) else 2
            if not fits:
                with torch.cuda.device(dev):
                    self._ws_eval = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
            cfg["workspace"] = self._ws_eval
            cfg["reuse_tables"] = mode
            # the tables are in the workspace after this call only if it computes (or keeps) them: S = 0 returns early
            self._ws_tables = (key, not compact_now) if S > 0 else None
        elif reuse_tables:
            raise RuntimeError("reuse_tables is for no-grad (inference) forwards")
        else:
            self._ws_tables = None          # a training forward: the weights are about to change
        if target is not None:
            if not cfg["grad"]:
                raise RuntimeError("forward_loss computes gradients: call it with grad enabled")
            target = torch.as_tensor(target)
            if target.numel() != S:
                raise ValueError("target holds %d classes, the batch has %d masked nodes" % (target.numel(), S))
            cfg["grad_scale"] = (1.0 / max(S, 1)) if grad_scale is None else float(grad_scale)
            target = target.reshape(-1)
            if fused is None:
                fused = cfg["batch_groups"] > 0
            if fused:
                res = _PaggLossFunction.apply(cfg, X, ids, codes, sel, target, *params)
            else:
                from . import optim
                out = _PaggFunction.apply(cfg, X, ids, codes, sel, *params)
                loss = optim.CrossEntropyLoss()(out, target.to(device=dev, dtype=torch.int64))        # the mean over the S rows
                if grad_scale is not None:
                    loss = loss * (float(grad_scale) * max(S, 1))
                res = (loss, out)
        else:
            res = _PaggFunction.apply(cfg, X, ids, codes, sel, *params)
        if guard and S > 0:
            first = self._range_calls == 0
            self._range_calls += 1
            if first or (self._range_calls % RANGE_POLL_EVERY == 0 and self._range_pending is None):
                self._range_read(cfg, sync=first)
                if first and self._range_wide:      # the very first result is not handed out on a range it cannot carry
                    return self._run(X, neis, num_w, walk_len, indices, layer_type, reuse_tables=False, group_slice=group_slice,
                                     target=target, grad_scale=grad_scale, fused=fused, batch_position=batch_position,
                                     seed=cfg["seed"] if (p > 0 and self.step_state is None) else seed,
                                     _math=_lib.SEQ_MATH_BF16X3)
        return res


class PathNet(_Aggregator):
    """PathNet_run.py:150-211.  state_dict: fc0, LSTM, fc2, nets.<d>, attw."""
    variant = "hetero"

    def __init__(self, feature_length, hidden_size, out_size, wl, dropout=None, cell=None, **kwargs):
        super().__init__()
        self._common_init(feature_length, hidden_size, out_size, dropout, cell)
        self.fc0 = nn.Linear(feature_length, hidden_size)
        self._make_cell("lstm")
        self.fc2 = nn.Linear(2 * hidden_size, out_size)
        self.nets = nn.ModuleList([nn.Linear(hidden_size, hidden_size) for _ in range(wl)])
        self.attw = nn.Linear(2 * hidden_size, 1)
        self.Lrelu = nn.LeakyReLU()

    def _bank_layers(self):
        return list(self.nets)


class PathNet_homo(PathNet):
    """PathNet_run.py:214-278 (xavier_uniform on fc0/fc2, :236-237)."""
    variant = "homo"

    def __init__(self, feature_length, hidden_size, out_size, wl, dropout=None, cell=None, **kwargs):
        super().__init__(feature_length, hidden_size, out_size, wl, dropout=dropout, cell=cell, **kwargs)
        nn.init.xavier_uniform_(self.fc0.weight)
        nn.init.xavier_uniform_(self.fc2.weight)


class PAGG(_Aggregator):
    """baseline/GPRGNN/src/copy.py:299-359.  state_dict: fc0, RNN, fc2, nei0..nei3.  dropout is 0.9 there."""
    variant = "pagg"

    def __init__(self, feature_length, hidden_size, out_size, node_num, dropout=0.9, cell=None, **kwargs):
        super().__init__()
        self._common_init(feature_length, hidden_size, out_size, dropout, cell)
        self.node_num = node_num
        self.fc0 = nn.Linear(feature_length, hidden_size)
        self._make_cell("rnn")
        self.fc2 = nn.Linear(2 * hidden_size, out_size)
        self.nei0 = nn.Linear(hidden_size, hidden_size)
        self.nei1 = nn.Linear(hidden_size, hidden_size)
        self.nei2 = nn.Linear(hidden_size, hidden_size)
        self.nei3 = nn.Linear(hidden_size, hidden_size)
        for lin in (self.fc0, self.fc2, self.nei0, self.nei1, self.nei2, self.nei3):
            nn.init.xavier_uniform_(lin.weight)

    def _bank_layers(self):
        return [self.nei0, self.nei1, self.nei2, self.nei3]

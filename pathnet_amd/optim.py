"""Loss and optimizer of the reference's training step (/root/reference/PathNet_run.py:295-297, :346-352) as single
launches of the HIP library (``pn_cross_entropy``, ``pn_adam_step``): same arithmetic as ``torch.nn.CrossEntropyLoss()``
and ``torch.optim.Adam(lr, betas, eps, weight_decay)``, ~10 us instead of ~100 us per step on the bench workload."""
import ctypes

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        if not logits.is_cuda:
            raise RuntimeError("pathnet_amd.cross_entropy: logits must be on the GPU (there is no CPU fallback)")
        lib = _lib.load()
        x = logits.contiguous().float()
        t = target.to(device=x.device, dtype=torch.int64).contiguous()
        if x.dim() != 2 or t.shape != (x.shape[0],):
            raise ValueError("cross_entropy: logits [rows, classes] and int targets [rows] expected")
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        g = torch.empty_like(x) if logits.requires_grad else None
        with torch.cuda.device(x.device):
            _lib.check(lib.pn_cross_entropy(_lib.ptr(x), _lib.ptr(t), x.shape[0], x.shape[1], _lib.ptr(loss),
                                            _lib.ptr(g), _stream()))
        ctx.g = g
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.g is None:
            return None, None
        # loss.backward() hands a freshly filled ones tensor down and costs a multiply by it: two launches for nothing.
        # optim.backward(loss) passes the cached tensor of `unit()` instead, recognised here by identity
        one = _UNIT.get(grad_out.device)
        if one is not None and grad_out.data_ptr() == one.data_ptr():
            return ctx.g, None
        return ctx.g * grad_out, None


_UNIT = {}


def unit(device):
    """the cached scalar 1.0 on `device` that `backward(loss)` seeds autograd with"""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    t = _UNIT.get(dev)
    if t is None:
        t = _UNIT[dev] = torch.ones((), dtype=torch.float32, device=dev)
    return t


def backward(loss):
    """``loss.backward()`` for a scalar loss of this module's cross entropy without the two launches autograd spends on the
    seed gradient (a ones_like fill, and the multiply by it in the loss's backward): the seed is a cached tensor that the
    loss recognises.  Same gradients, bit for bit."""
    loss.backward(unit(loss.device))


def cross_entropy(logits, target):
    """Mean softmax cross entropy, = ``torch.nn.functional.cross_entropy(logits, target)``."""
    return _CrossEntropy.apply(logits, target)


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for ``torch.nn.CrossEntropyLoss()`` as the reference constructs it (PathNet_run.py:297)."""

    def forward(self, logits, target):
        return cross_entropy(logits, target)


class Adam(torch.optim.Optimizer):
    """Drop-in for ``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` (no amsgrad / maximize): one launch
    per step over all parameters.  State keys match torch's (``step``, ``exp_avg``, ``exp_avg_sq``)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, step_state=None):
        """step_state (StepState): the step count of the bias corrections is read from device memory when the kernel runs
        (a step captured in a hipGraph can be replayed); state["step"] then only counts the Python-side calls."""
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("Adam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.step_state = step_state

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            todo = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("pathnet_amd.Adam: contiguous fp32 GPU parameters only")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                todo.append((p, p.grad.contiguous(), st))
            # tensors that have taken the same number of steps share a launch (normally: all of them)
            by_step = {}
            for item in todo:
                by_step.setdefault(item[2]["step"], []).append(item)
            groups_left = len(by_step)
            for step, items in by_step.items():
                groups_left -= 1
                arr = (_lib.AdamTensor * len(items))()
                for i, (p, g, st) in enumerate(items):
                    arr[i] = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                             st["exp_avg_sq"].data_ptr(), p.numel())
                ss = self.step_state
                with torch.cuda.device(items[0][0].device):
                    if ss is not None and ss.advance_in_adam and groups_left == 0 and group is self.param_groups[-1]:
                        # the optimizer's last launch of the step moves the step state on: the next StepState.advance() is free
                        _lib.check(lib.pn_adam_step_advance(arr, len(items), group["lr"], group["betas"][0], group["betas"][1],
                                                            group["eps"], group["weight_decay"], ss.ptr(), _stream()))
                        ss._advanced_by_adam = True
                    else:
                        _lib.check(lib.pn_adam_step(arr, len(items), group["lr"], group["betas"][0], group["betas"][1],
                                                    group["eps"], group["weight_decay"], step,
                                                    ss.ptr() if ss is not None else None, _stream()))
                # the kernel wrote through raw pointers: tell autograd's version counters, which is what everything that
                # caches a function of the parameters keys on (the modules' reuse_tables, dist.ShardedAggregator.begin_step) --
                # torch.optim.Adam's in-place ops bump them as a matter of course.  No launch.
                for p, _, _ in items:
                    torch.autograd.graph.increment_version(p)
        return loss


class StepState:
    """struct pn_step_state in device memory: {epoch, seed, adam_step} of the current training step, read by the sampler,
    the aggregator's dropout and Adam WHEN THEIR KERNELS RUN -- so a whole step (sample -> forward -> loss -> backward ->
    Adam) captured into a hipGraph (torch.cuda.CUDAGraph) replays with a new epoch / seed / step count each time.
    ``advance()`` is the first launch of a step: epoch += 1, adam_step += 1, seed = splitmix64(seed).
    advance_in_adam (round 6): pathnet_amd.Adam's last launch of a step performs the advance for the NEXT step
    (pn_adam_step_advance), and the ``advance()`` that follows it launches nothing -- one launch less on the step's critical
    path; every ``advance()`` still stands for exactly one advance of the state."""

    def __init__(self, device="cuda", seed=0, first_epoch=0, advance_in_adam=False):
        dev = torch.device(device)
        self.t = torch.tensor([int(first_epoch) - 1, int(seed) & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64, device=dev)
        self.advance_in_adam = bool(advance_in_adam)
        self._advanced_by_adam = False

    def ptr(self):
        return ctypes.c_void_p(self.t.data_ptr())

    def advance(self):
        if self._advanced_by_adam:      # the optimizer step before this one has already moved the state
            self._advanced_by_adam = False
            return
        with torch.cuda.device(self.t.device):
            _lib.check(_lib.load().pn_step_state_advance(self.ptr(), _lib.stream_ptr(self.t.device)))

    def values(self):
        e, s, a, _ = self.t.tolist()
        return {"epoch": e, "seed": s & 0xFFFFFFFFFFFFFFFF, "adam_step": a}

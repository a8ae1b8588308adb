"""How the parity tests compare gradients (PathNet_run.py:348-352: autograd through the aggregator).

Every tensor is held to a bound RELATIVE TO ITS OWN largest gradient element,

    max |got - ref|  <=  rel * max |ref|  +  NOISE * scale,        scale = the largest |ref| over all tensors of the case,

so that a test whose upstream gradient is scaled by 1 / S (gradients of 1e-3 ... 1e-7) is exactly as strict as one whose
gradients are of order one.  The additive term is there for gradients that are cancelling sums of much larger terms -- the
hetero class's attention bias (its softmax makes it vanish whenever the scores share a sign: reference and product both
return rounding noise there), attention weights of a freshly initialised module -- and is nine orders below the case's
largest gradient, sixty times below one fp32 ulp of it: no tensor that carries signal is excused by it.  Calibrated on
the 238 comparisons of the GPU suite (PN_GRADCHECK_LOG, round 5): with 1e-9 one of them fails -- a 5-node x 100-path
degenerate case whose attention-bias gradient is a sum of 500 terms each a thousand times larger than the result; that
test passes its own `noise`.

`assert_grads_close` also refuses to be vacuous: a tensor whose reference gradient lies inside its own tolerance would pass
with a zeroed gradient, so every such tensor must be named in `zero_ok` (tests/test_gradcheck.py exercises the checker
itself: a zeroed, a scaled and a slightly perturbed gradient all fail).

Until round 4 the bound was 3e-5 * max(1, |ref|): an ABSOLUTE 3e-5 for every gradient below one, which the
configuration-size tests' 1 / S scaling made vacuous for half of the parameters."""
import json
import os

import numpy as np

GRAD_REL = 3e-5
NOISE = 1e-9
ZERO_OK_HETERO = ("attw.bias",)        # softmax over the W paths of a node: a constant added to same-signed scores changes nothing


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def grad_report(got, ref, rel=GRAD_REL, noise=NOISE):
    """{name: (max |got - ref|, tolerance, max |ref|)} and the case's scale"""
    ref = {k: _np(v) for k, v in ref.items()}
    scale = max([float(np.abs(v).max()) if v.size else 0.0 for v in ref.values()] + [0.0])
    rows = {}
    for k, r in ref.items():
        g = _np(got[k])
        assert g.shape == r.shape, (k, g.shape, r.shape)
        m = float(np.abs(r).max()) if r.size else 0.0
        err = float(np.abs(g.astype(np.float64) - r.astype(np.float64)).max()) if r.size else 0.0
        if not np.isfinite(g).all():
            err = float("inf")
        rows[k] = (err, rel * m + noise * scale, m)
    return rows, scale


def assert_grads_close(got, ref, rel=GRAD_REL, noise=NOISE, zero_ok=(), tag=None):
    """got / ref: {name: tensor or array}.  Fails with the full table of offenders; see the module docstring."""
    rows, scale = grad_report(got, ref, rel, noise)
    _log(tag, rows, scale, rel)
    bad = {k: (e, t, m) for k, (e, t, m) in rows.items() if not e <= t}
    assert not bad, "gradients off (name: error, tolerance, |ref|_inf): %r" % (bad,)
    # (a reference gradient that is EXACTLY zero -- a distance layer no path of the case uses, W_hh at path length 1 -- is
    #  checked by the bound itself: the product must return zero up to the noise term)
    vacuous = sorted(k for k, (e, t, m) in rows.items() if 0.0 < m <= t)
    assert set(vacuous) <= set(zero_ok), "a zeroed gradient would pass for %r (scale %.3g)" % (vacuous, scale)
    return rows


def _log(tag, rows, scale, rel):
    """PN_GRADCHECK_LOG=<file>: one JSON line per comparison (calibration runs on the GPU box)"""
    path = os.environ.get("PN_GRADCHECK_LOG")
    if not path:
        return
    worst = max(((e / m if m > 0 else 0.0), k) for k, (e, t, m) in rows.items()) if rows else (0.0, "")
    rec = {"test": tag or os.environ.get("PYTEST_CURRENT_TEST", ""), "rel": rel, "scale": scale, "worst_rel": worst[0],
           "worst": worst[1], "rows": {k: [e, t, m] for k, (e, t, m) in rows.items()}}
    with open(path, "a") as f:
        f.write(json.dumps(rec) + "\n")


def named_grads(module, extra=None):
    """{parameter name: .grad} of a module (+ extra entries, e.g. {"X": X.grad})"""
    out = {k: v.grad for k, v in module.named_parameters()}
    for k, v in out.items():
        assert v is not None, k
    out.update(extra or {})
    return out

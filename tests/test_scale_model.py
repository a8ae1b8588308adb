"""tools/scale_model.py: the arithmetic behind the modelled 2 / 4 / 8-rank step times runs on the
committed single-GPU bench line and stays inside what arithmetic allows -- no rank count is faster than perfect scaling, one
rank is the measured step, collectives cost something as soon as there are two ranks."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("scale_model", os.path.join(ROOT, "tools", "scale_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_model_rows_are_consistent_with_the_measured_step():
    sm = _load()
    path = next(p for p in (os.path.join(ROOT, "profiles", n) for n in ("r04_bench_final.json", "r04_bench_f16_v1.json",
                                                                        "r03_bench_final.json")) if os.path.exists(p))
    b = json.load(open(path))
    H, F, C, L = 128, 1433, 7, 4
    grad = (F * H + H + L * (H * H + H) + 2 * (4 * H * H + 4 * H) + 2 * H + 1 + 2 * H * C + C) * 4
    weak = sm.model(b["stages_ms"], b["ms_per_step"], 2708, H, grad, True, 50.0, 25.0)
    assert [r[0] for r in weak] == [1, 2, 4, 8]
    assert abs(weak[0][1] - b["ms_per_step"]) < 1e-6 + 0.05 * b["ms_per_step"]        # one rank = the measured step
    for R, t, eff, parts in weak[1:]:
        assert 0.0 < eff <= 1.0 and t >= weak[0][1] and parts["collectives"] > 0.0      # weak scaling: never faster than one rank
    g = b["bgp_scale_step"]
    strong = sm.model(g["stages_ms"], g["ms_per_step"], 63977, H, grad, False, 50.0, 25.0, idx_bytes=1000)
    for R, t, sp, parts in strong:
        assert sp <= R + 1e-9 and (R == 1 or sp > 1.0)                                  # strong scaling: between 1 and R
    rep = sm.model(g["stages_ms"], g["ms_per_step"], 63977, H, grad, False, 50.0, 25.0, replicated=True)
    assert all(p["all_gather_Xh"] == 0.0 for R, t, sp, p in rep if R > 1)              # no exchange of Xh in that mode
    # overlap (round 4): a collective hidden under compute costs between nothing and its full time -- never less than zero,
    # and the overlapped step is never slower than the serial one nor faster than the step without collectives
    for weak_, st, tot, n in ((True, b["stages_ms"], b["ms_per_step"], 2708), (False, g["stages_ms"], g["ms_per_step"], 63977)):
        ser = sm.model(st, tot, n, H, grad, weak_, 50.0, 25.0)
        ovl = sm.model(st, tot, n, H, grad, weak_, 50.0, 25.0, overlap=True)
        for (R, t0, _, p0), (_, t1, _, p1) in zip(ser, ovl):
            assert t1 <= t0 + 1e-9 and t1 >= t0 - p0["collectives"] - 1e-9
            if R == 1:
                continue
            assert 0.0 <= p1["all_gather_Xh"] <= p0["all_gather_Xh"] + 1e-12
            assert 0.0 <= p1["reduce_scatter_dXh"]
    # replicated mode restricted to the touched rows: fc0 shrinks with the rank's share of the nodes, nothing else changes
    full = sm.model(g["stages_ms"], g["ms_per_step"], 63977, H, grad, False, 50.0, 25.0, replicated=True)
    part = sm.model(g["stages_ms"], g["ms_per_step"], 63977, H, grad, False, 50.0, 25.0, replicated=True,
                    touched_nodes=lambda R: 1.0 / R)
    zx = g["stages_ms"].get("zero_fill", 0.0) * (1.0 - 0.8)        # the d Xh part of the zero fill follows the touched nodes too
    for (R, t0, _, p0), (_, t1, _, p1) in zip(full, part):
        assert abs(p1["own_rows"] - p0["own_rows"] / R) < 1e-9
        assert abs((t0 - t1) - (p0["own_rows"] - p1["own_rows"]) - zx * (1.0 - 1.0 / R)) < 1e-9

"""The aggregator oracle (oracle/pagg_oracle.py) against the committed golden vectors produced by the
reference classes, and against the reference classes themselves when /root/reference is mounted."""
import numpy as np
import pytest
import torch

from conftest import golden, golden_files
from oracle import pagg_oracle as po

PAGG_GOLDENS = golden_files("pagg_*.npz")


def load_case(name):
    g = golden(name)
    params = {k[len("param/"):]: torch.tensor(v) for k, v in g.items() if k.startswith("param/")}
    grads = {k[len("grad/"):]: torch.tensor(v) for k, v in g.items() if k.startswith("grad/")}
    return g, params, grads


@pytest.mark.parametrize("name", PAGG_GOLDENS)
def test_oracle_forward_matches_reference_golden(name):
    g, params, _ = load_case(name)
    sel = np.nonzero(g["mask"])[0]
    out = po.forward(str(g["variant"]), params, torch.tensor(g["X"]), g["ids"], g["codes"], sel,
                     int(g["W"]), int(g["L"]))
    assert np.abs(out.numpy() - g["out"]).max() < 2e-6


@pytest.mark.parametrize("name", PAGG_GOLDENS)
def test_oracle_backward_matches_reference_golden(name):
    g, params, grads = load_case(name)
    sel = np.nonzero(g["mask"])[0]
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    X = torch.tensor(g["X"]).requires_grad_(True)
    out = po.forward(str(g["variant"]), params, X, g["ids"], g["codes"], sel, int(g["W"]), int(g["L"]))
    (out * torch.tensor(g["G"])).sum().backward()
    for k, ref in grads.items():
        got = params[k].grad
        assert got is not None, k
        # two fp32 CPU evaluations that sum the same terms in different orders (3480 paths x 4 steps in the hid-128
        # fixture): a few ulp of the largest entry
        tol = 5e-6 * max(1.0, ref.abs().max().item())
        assert (got - ref).abs().max().item() < tol, k
    assert (X.grad - torch.tensor(g["grad_X"])).abs().max().item() < 2e-6


def test_plan_hetero_quirks():
    S, W, L = 3, 2, 4
    P = S * W
    ids = np.arange(P * L).reshape(P, L)
    codes = (np.arange(P * L) % L).reshape(P, L)
    node, code, group, member, ego = po.plan("hetero", ids, codes, S, W, L)
    # slot 0, step 0 is row r=0: t'=0, p=0 -> last node of path 0
    assert node[0, 0] == ids[0, L - 1]
    # slot 1, step 2: r = 6 -> t' = 1, p = 0 -> ids[0, L-2]
    assert node[1, 2] == ids[0, L - 2]
    assert (code == codes).all()
    assert (group == np.arange(P) % S).all() and (member == np.arange(P) // S).all()
    assert (ego == ids[:, 0]).all()
    node, code, group, member, ego = po.plan("homo", ids, codes, S, W, L)
    assert (node == ids).all() and (group == np.arange(P) // W).all()


@pytest.mark.reference
@pytest.mark.parametrize("variant,cname", [("hetero", "PathNet"), ("homo", "PathNet_homo"), ("pagg", "PAGG")])
def test_oracle_matches_live_reference_classes(variant, cname):
    import warnings
    warnings.filterwarnings("ignore")
    from ref_extract import reference_classes
    cls = reference_classes(0.0)
    torch.manual_seed(3)
    rng = np.random.default_rng(3)
    N, F, H, C, W, L, S = 64, 21, 128, 6, 40, 4, 29
    model = cls[cname](F, H, C, L if variant != "pagg" else N)
    X = torch.rand(N, F)
    mask = np.zeros(N, bool)
    mask[rng.permutation(N)[:S]] = True
    sel = np.nonzero(mask)[0]
    ids = rng.integers(0, N, size=(S, W, L))
    ids[:, :, 0] = sel[:, None]
    codes = rng.integers(0, L, size=(S, W, L))
    model.eval()
    ref = model(X, torch.tensor(ids.reshape(S, W * L)), W, L, torch.tensor(mask), torch.tensor(codes),
                torch.arange(S * W * L))
    got = po.forward(variant, dict(model.state_dict()), X, ids, codes, sel, W, L)
    assert (ref - got).abs().max().item() < 2e-6


@pytest.mark.reference
def test_state_dict_contract_cornell_pth():
    """saved_models/cornell.pth pins the state_dict keys/shapes of PathNet(1703,128,5,4)."""
    sd = torch.load("/root/reference/saved_models/cornell.pth", map_location="cpu")
    want = {"fc0.weight": (128, 1703), "fc0.bias": (128,), "LSTM.weight_ih_l0": (512, 128),
            "LSTM.weight_hh_l0": (512, 128), "LSTM.bias_ih_l0": (512,), "LSTM.bias_hh_l0": (512,),
            "fc2.weight": (5, 256), "fc2.bias": (5,), "attw.weight": (1, 256), "attw.bias": (1,)}
    for d in range(4):
        want["nets.%d.weight" % d] = (128, 128)
        want["nets.%d.bias" % d] = (128,)
    assert {k: tuple(v.shape) for k, v in sd.items()} == want


def test_oracle_gru_cell_is_torch_nn_gru():
    """The ablation cells have no code in the reference (README.md:118): the oracle's "gru" is pinned to torch.nn.GRU
    itself, "mean" / "sum" to the definitions, on the bank outputs the oracle reports."""
    torch.manual_seed(3)
    rng = np.random.default_rng(3)
    N, F, H, C, W, L, S = 40, 12, 32, 3, 6, 4, 9
    lin = torch.nn.Linear
    gru = torch.nn.GRU(H, H)
    params = {"fc0.weight": lin(F, H).weight, "fc0.bias": torch.randn(H), "fc2.weight": lin(2 * H, C).weight,
              "fc2.bias": torch.randn(C), "attw.weight": lin(2 * H, 1).weight, "attw.bias": torch.randn(1)}
    for d in range(L):
        params["nets.%d.weight" % d] = lin(H, H).weight
        params["nets.%d.bias" % d] = torch.randn(H) * 0.1
    for k, v in gru.named_parameters():
        params["GRU." + k] = v
    params = {k: v.detach() for k, v in params.items()}
    X = torch.rand(N, F)
    sel = np.sort(rng.permutation(N)[:S])
    ids = rng.integers(0, N, (S, W, L))
    codes = np.minimum(rng.integers(0, L, (S, W, L)), np.arange(L)[None, None, :])
    for variant in ("hetero", "homo"):
        out, inter = po.forward(variant, params, X, ids, codes, sel, W, L, cell="gru", return_intermediates=True)
        with torch.no_grad():
            _, hn = gru(inter["y"].transpose(0, 1).contiguous())
        assert (inter["h"] - hn[0]).abs().max().item() < 1e-6
        _, im = po.forward(variant, params, X, ids, codes, sel, W, L, cell="mean", return_intermediates=True)
        _, isum = po.forward(variant, params, X, ids, codes, sel, W, L, cell="sum", return_intermediates=True)
        assert torch.allclose(im["h"], inter["y"].mean(dim=1), atol=1e-7)
        assert torch.allclose(isum["h"], inter["y"].sum(dim=1), atol=1e-6)
    with pytest.raises(ValueError):
        po.forward("homo", params, X, ids, codes, sel, W, L, cell="lstm2")


# ---- training mode: the reference classes' own forward/backward with the dropout masks they drew recorded
# (tests/golden/make_golden_pagg_train.py) ---------------------------------------------------------------------------------
TRAIN_GOLDENS = golden_files("paggtrain_*.npz")


def test_training_mode_goldens_are_committed():
    assert len(TRAIN_GOLDENS) >= 8
    for name in TRAIN_GOLDENS:
        g = golden(name)
        L, S, W, H = (int(g[k]) for k in "LSWH")
        assert g["mask_seq"].shape == (L, S * W, H) and g["mask_cls"].shape == (S, 2 * H)
        p = float(g["p"])
        scale = np.float32(1.0 / (1.0 - p))
        for m in (g["mask_seq"], g["mask_cls"]):        # keep bits scaled by 1 / (1 - p), nothing else
            assert np.isin(m, (np.float32(0.0), m.max())).all() and abs(m.max() - scale) < 1e-5 * scale
            assert abs(float((m > 0).mean()) - (1.0 - p)) < 0.05


@pytest.mark.parametrize("name", TRAIN_GOLDENS)
def test_oracle_training_mode_matches_reference_golden(name):
    from gradcheck import ZERO_OK_HETERO, assert_grads_close
    g, params, grads = load_case(name)
    variant = str(g["variant"])
    sel = np.nonzero(g["mask"])[0]
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    X = torch.tensor(g["X"]).requires_grad_(True)
    out = po.forward(variant, params, X, g["ids"], g["codes"], sel, int(g["W"]), int(g["L"]),
                     drop_seq=torch.tensor(g["mask_seq"]), drop_cls=torch.tensor(g["mask_cls"]))
    assert np.abs(out.detach().numpy() - g["out"]).max() < 2e-6 * max(1.0, np.abs(g["out"]).max())
    (out * torch.tensor(g["G"])).sum().backward()
    got = {k: v.grad for k, v in params.items() if k in grads}
    got["X"] = X.grad
    ref = dict(grads, X=torch.tensor(g["grad_X"]))
    assert_grads_close(got, ref, zero_ok=ZERO_OK_HETERO if variant == "hetero" else ())


@pytest.mark.parametrize("name", TRAIN_GOLDENS[:3])
def test_training_mode_golden_needs_its_masks(name):
    """the fixtures are not eval-mode outputs in disguise: without the masks the oracle lands far away"""
    g, params, _ = load_case(name)
    sel = np.nonzero(g["mask"])[0]
    out = po.forward(str(g["variant"]), params, torch.tensor(g["X"]), g["ids"], g["codes"], sel, int(g["W"]), int(g["L"]))
    assert np.abs(out.numpy() - g["out"]).max() > 1e-2

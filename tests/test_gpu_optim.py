"""Loss and optimizer launches (pn_cross_entropy, pn_adam_step) against torch's own implementations."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,classes", [(1299, 7), (1, 2), (70000, 5), (33, 300)])
def test_cross_entropy_matches_torch(rows, classes):
    import pathnet_amd
    torch.manual_seed(rows)
    x = (torch.randn(rows, classes) * 3).cuda().requires_grad_(True)
    t = torch.randint(0, classes, (rows,)).cuda()
    loss = pathnet_amd.cross_entropy(x, t)
    loss.backward()
    xr = x.detach().double().cpu().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(xr, t.cpu())
    want.backward()
    assert abs(loss.item() - want.item()) < 2e-6 * max(1.0, abs(want.item()))
    gmax = xr.grad.abs().max().item()
    assert (x.grad.cpu().double() - xr.grad).abs().max().item() < 2e-6 * gmax
    # the module form, scaled upstream gradient
    x.grad = None
    (pathnet_amd.CrossEntropyLoss()(x, t) * 3.0).backward()
    assert (x.grad.cpu().double() - 3.0 * xr.grad).abs().max().item() < 6e-6 * gmax


def test_adam_matches_torch_optim_adam():
    import pathnet_amd
    torch.manual_seed(3)
    shapes = [(128, 1433), (128,), (512, 128), (7, 256), (1,), (3, 5, 7)] + [(17,)] * 40    # > PN_ADAM_MAX_TENSORS
    ours = [torch.nn.Parameter(torch.randn(*s).cuda()) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().cpu().double().clone()) for p in ours]
    o1 = pathnet_amd.Adam(ours, lr=0.005, weight_decay=0.0005)
    o2 = torch.optim.Adam(ref, lr=0.005, weight_decay=0.0005)
    for it in range(25):
        for p, q in zip(ours, ref):
            g = torch.randn(*p.shape)
            if it % 7 == 3 and p.numel() == 17:
                p.grad, q.grad = None, None          # parameters without a gradient are skipped, as in torch
            else:
                p.grad, q.grad = g.cuda(), g.double()
        o1.step()
        o2.step()
    for p, q in zip(ours, ref):
        assert (p.detach().cpu().double() - q.detach()).abs().max().item() < 2e-6
    st = o1.state[ours[0]]
    assert st["step"] == 25 and set(st) == {"step", "exp_avg", "exp_avg_sq"}


def test_adam_rejects_cpu_parameters():
    import pathnet_amd
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        pathnet_amd.Adam([p]).step()


def test_seeded_backward_equals_loss_backward():
    """pathnet_amd.backward(loss) = loss.backward() without the ones_like fill and the multiply by it: same gradients, bit for
    bit, through the three-call step and through the fused one (deterministic backward: run-to-run bits are comparable)"""
    import pathnet_amd
    torch.manual_seed(5)
    N, F, H, C, S, W, L = 150, 24, 64, 4, 40, 6, 4
    m = pathnet_amd.PathNet_homo(F, H, C, L, dropout=0.5).cuda().train()
    m.deterministic = True
    X = torch.rand(N, F, device="cuda")
    sel = torch.randperm(N)[:S].sort().values.to(torch.int32).cuda()
    ids = torch.randint(0, N, (S, W, L), dtype=torch.int32, device="cuda")
    ids[:, :, 0] = sel[:, None]
    codes = torch.randint(0, L, (S, W, L), dtype=torch.uint8, device="cuda")
    y = torch.randint(0, C, (S,), device="cuda")
    for fused in (False, True):
        got = []
        for seeded in (False, True):
            torch.manual_seed(9)
            m.zero_grad(set_to_none=True)
            loss, _ = m.forward_loss(X, ids, W, L, sel, codes, y, fused=fused)
            if seeded:
                pathnet_amd.backward(loss)
            else:
                loss.backward()
            got.append({k: v.grad.clone() for k, v in m.named_parameters()})
        for k in got[0]:
            assert torch.equal(got[0][k], got[1][k]), (fused, k)


def test_adam_that_advances_the_step_state_equals_the_explicit_advance():
    """StepState(advance_in_adam=True): the optimizer's last launch of a step moves {epoch, seed, adam_step} on
    (pn_adam_step_advance: the workgroup that finishes last, after every workgroup has read the step count) and the next
    advance() launches nothing.  Same states after every step and bit-equal parameters as with the explicit launch, with
    more tensors than one launch takes and with a step in which no parameter has a gradient."""
    import pathnet_amd
    torch.manual_seed(11)
    shapes = [(300, 129), (64,), (5000,)] + [(33,)] * 45        # > PN_ADAM_MAX_TENSORS: two launches per step
    runs = []
    for auto in (False, True):
        torch.manual_seed(12)
        ps = [torch.nn.Parameter(torch.randn(*s).cuda()) for s in shapes]
        st = pathnet_amd.StepState("cuda", seed=77, first_epoch=3, advance_in_adam=auto)
        opt = pathnet_amd.Adam(ps, lr=0.01, weight_decay=0.001, step_state=st)
        states = []
        for it in range(6):
            st.advance()
            states.append(st.values())
            for p in ps:
                p.grad = None if it == 4 else torch.randn(*p.shape, device="cuda")
            opt.step()
        torch.cuda.synchronize()
        assert int(st.t[3].item()) & 0xFFFFFFFF == 0            # the ticket counter is back at zero
        runs.append((states, [p.detach().clone() for p in ps]))
    assert runs[0][0] == runs[1][0]
    assert runs[0][0][0] == {"epoch": 3, "seed": runs[0][0][0]["seed"], "adam_step": 1} and runs[0][0][5]["adam_step"] == 6
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)

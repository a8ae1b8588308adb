"""Host-side behaviour of the loss / optimizer wrappers (no GPU): they refuse CPU tensors instead of falling back."""
import pytest
import torch


def test_adam_refuses_cpu_parameters():
    import pathnet_amd
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    opt = pathnet_amd.Adam([p], lr=0.01, weight_decay=0.1)
    with pytest.raises(RuntimeError, match="GPU"):
        opt.step()
    assert torch.equal(p.detach(), torch.zeros(4))          # untouched


def test_adam_validates_hyper_parameters():
    import pathnet_amd
    p = torch.nn.Parameter(torch.zeros(4))
    for kw in (dict(lr=-1.0), dict(eps=-1e-8), dict(weight_decay=-0.1), dict(betas=(1.0, 0.999))):
        with pytest.raises(ValueError):
            pathnet_amd.Adam([p], **kw)


def test_cross_entropy_refuses_cpu_logits():
    import pathnet_amd
    with pytest.raises(RuntimeError, match="GPU"):
        pathnet_amd.cross_entropy(torch.zeros(3, 2), torch.zeros(3, dtype=torch.long))

"""CPU: the host logic of xnrs_amd.explain.integrated_gradients (interpolation path, batching of the steps, chunking, the
attribution sums of explain.py:167-173) on a stand-in model in plain torch -- the HIP modules cannot run without a GPU; their
explanation is checked against the CPU oracle in tests/test_hip_explain.py."""
import pytest
import torch

from xnrs_amd.explain import integrated_gradients


class _News(torch.nn.Module):
    def __init__(self, D, E):
        super().__init__()
        self.fc = torch.nn.Linear(D, E)

    def forward(self, inpt):
        x, m = inpt                                   # (B, N, S, D), (B, N, S, 1)
        h = torch.tanh(self.fc(x)) * m
        return h.sum(dim=2) / (m.sum(dim=2) + 1e-8), (m.sum(dim=2) > 0).float()


class _User(torch.nn.Module):
    def forward(self, inpt):
        h, hm = inpt                                  # (B, N, E), (B, N, 1)
        return ((h * hm).sum(dim=1, keepdim=True) / (hm.sum(dim=1, keepdim=True) + 1e-8)) ** 2


class _Stub(torch.nn.Module):
    def __init__(self, D=6, E=5):
        super().__init__()
        self.news_encoder, self.user_encoder = _News(D, E), _User()

    def rec_model(self, u, c):
        return torch.bmm(c, u.transpose(-1, -2))


def _case(seed=0, H=4, C=3, S=5, D=6):
    g = torch.Generator().manual_seed(seed)
    hx = torch.randn(1, H, S, D, generator=g)
    hm = (torch.rand(1, H, S, 1, generator=g) < 0.8).float()
    hm[:, :, 0] = 1
    cx = torch.randn(1, C, S, D, generator=g)
    cm = torch.ones(1, C, S, 1)
    return hx, hm, cx, cm


def _reference_loop(model, hx, hm, cx, cm, cidx, n_steps, act):
    """explain.py:152-173 as written (grads collected, summed, multiplied by the input)."""
    c, _ = model.news_encoder((cx[:, cidx:cidx + 1], cm[:, cidx:cidx + 1]))
    c = c.detach()
    da = 1 / n_steps
    grads = []
    for a in torch.arange(da, 1 + da, da)[:n_steps]:
        ga = (a * hx).requires_grad_()
        ha, ham = model.news_encoder((ga, hm))
        sa = act(model.rec_model(model.user_encoder.forward(inpt=(ha, ham)), c))
        grads.append(torch.autograd.grad(sa, ga)[0])
    int_grads = torch.sum(torch.cat(grads) * da, dim=0)
    attr = torch.sum(int_grads * hx.detach(), dim=(0, 3))
    return attr, float(sa.item())


@pytest.mark.parametrize("n_steps,per", [(1, 0), (7, 0), (20, 6), (20, 20), (20, 64)])
def test_batched_steps_equal_the_reference_loop(n_steps, per):
    torch.manual_seed(1)
    model = _Stub().double()
    hx, hm, cx, cm = (t.double() for t in _case())
    ident = lambda t: t
    attr, s_true = _reference_loop(model, hx, hm, cx, cm, 2, n_steps, ident)
    out = integrated_gradients(model, hx, hm, cx, cm, candidate_idx=2, n_steps=n_steps, activation=None, steps_per_batch=per)
    assert torch.allclose(out["attr"], attr, rtol=1e-10, atol=1e-12)
    assert abs(out["s_true"] - s_true) < 1e-10
    assert abs(out["s_attr"] - float(attr.sum())) < 1e-10
    assert torch.allclose(out["news_attribution"], attr.sum(dim=1), rtol=1e-10, atol=1e-12)
    loop = integrated_gradients(model, hx, hm, cx, cm, candidate_idx=2, n_steps=n_steps, activation=None, batched=False)
    assert torch.allclose(loop["attr"], attr, rtol=1e-10, atol=1e-12)
    assert all(p.grad is None for p in model.parameters())


def test_completeness_and_argument_checks():
    """Sum of the attributions -> f(x) - f(0) as the path is refined (the defining property of integrated gradients; f(0) = 0
    for the stand-in), and the argument errors."""
    torch.manual_seed(2)
    model = _Stub().double()
    hx, hm, cx, cm = (t.double() for t in _case(seed=3))
    out = integrated_gradients(model, hx, hm, cx, cm, candidate_idx=0, n_steps=400, activation=None)
    with torch.no_grad():
        c0, _ = model.news_encoder((cx[:, :1], cm[:, :1]))
        f0 = float(model.rec_model(model.user_encoder((model.news_encoder((hx * 0, hm)))), c0))
    assert abs(out["s_attr"] - (out["s_true"] - f0)) < 2e-2 * max(abs(out["s_true"] - f0), 1e-3)
    with pytest.raises(ValueError):
        integrated_gradients(model, hx[0], hm, cx, cm)
    with pytest.raises(ValueError):
        integrated_gradients(model, hx, hm, cx, cm, n_steps=0)

"""pase_mlp_head1_step (pase_amd/csrc/mlp_head1.hip): the decoder worker's pointwise tail -- PReLU, MLPBlock(128 -> 64, context 1),
PReLU, Conv1d(64, 1, 1), loss -- forward and backward in one pass, against an fp64 autograd evaluation of the same modules
(pase/models/modules.py:527-556, Minions/minions.py:416-417,446, pase/losses.py:33-37) and against the six launches it replaces.
"""
import pytest
import torch
import torch.nn.functional as F

from pase_amd import kernels as K


def _rel(a, ref):
    return float((a.detach().cpu().double() - ref).norm() / ref.norm().clamp_min(1e-300))


LOSSES = {"l1": K.LOSS_L1, "mse": K.LOSS_MSE, "bce": K.LOSS_BCE}


@pytest.mark.parametrize("S,T,loss,nwg", [
    (2, 300, "l1", 0),        # three tiles per sequence, the last one ragged (44 live time steps)
    (3, 128, "mse", 2),       # whole tiles, two workgroups walk three tiles
    (1, 50, "bce", 0),        # a single partial tile
    (2, 256, "l1", 1),        # one workgroup accumulates every tile
])
def test_mlp_head1_step_vs_fp64_autograd(dev, monkeypatch, S, T, loss, nwg):
    if nwg:
        monkeypatch.setenv("PASE_X6C_MAXWG", str(nwg))
    C, H = 128, 64
    torch.manual_seed(7)
    y = torch.randn(S, C, T)
    a0, a1 = torch.rand(C) * 0.5, torch.rand(H) * 0.5
    w1, b1 = torch.randn(H, C) * 0.1, torch.randn(H) * 0.1
    w2, b2 = torch.randn(H) * 0.2, torch.randn(1) * 0.1
    tgt = torch.rand(S, T) if loss == "bce" else torch.randn(S, T)
    gscale = 0.37 / (S * T)
    # ---- fp64 reference ---------------------------------------------------------------------------------------------------
    P = [v.double().requires_grad_(True) for v in (y, a0, w1, b1, a1, w2, b2)]
    yd, a0d, w1d, b1d, a1d, w2d, b2d = P
    h0 = F.prelu(yd, a0d)
    h1 = F.prelu(F.conv1d(h0, w1d[:, :, None], b1d), a1d)
    pred = F.conv1d(h1, w2d[None, :, None], b2d)[:, 0]
    td = tgt.double()
    if loss == "l1":
        el = (pred - td).abs()
    elif loss == "mse":
        el = (pred - td) ** 2
    else:
        el = F.binary_cross_entropy_with_logits(pred, td, reduction="none")
    (el.sum() * gscale).backward()
    # ---- device -----------------------------------------------------------------------------------------------------------
    t = lambda v: v.to(dev)
    dy = torch.full((S, C, T), float("nan"), device=dev)
    predv = torch.full((S, T), float("nan"), device=dev)
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    sums0 = torch.zeros(C, 3, dtype=torch.float64, device=dev)
    sums1 = torch.zeros(3 * H + 1, dtype=torch.float64, device=dev)
    base = (torch.randn(H, C) * 1e-3).to(dev)                        # dw1 is a += target
    dw1 = base.clone()
    assert K.mlp_head1_supported(S=S, C_=C, T=T, H=H)
    K.mlp_head1_step(t(y), t(a0), t(w1), t(b1), t(a1), t(w2), t(b2), t(tgt), predv, dy, acc, sums0, sums1, dw1,
                     S=S, C_=C, T=T, H=H, loss_type=LOSSES[loss], grad_scale=gscale)
    assert _rel(predv, pred.detach()) < 1e-6
    assert abs(float(acc) - float(el.detach().sum())) <= 1e-6 * float(el.detach().sum())
    assert _rel(dy, yd.grad) < 2e-6
    assert _rel(dw1 - base, w1d.grad) < 2e-6
    s1 = sums1.cpu()[:3 * H].view(H, 3)
    assert _rel(s1[:, 0], w2d.grad) < 2e-6 and _rel(s1[:, 1], a1d.grad) < 2e-6 and _rel(s1[:, 2], b1d.grad) < 2e-6
    assert abs(float(sums1[3 * H]) - float(b2d.grad)) <= 2e-6 * max(1e-12, abs(float(b2d.grad))) + 1e-12
    s0 = sums0.cpu()
    assert _rel(s0[:, 0], yd.grad.sum((0, 2))) < 2e-6 and _rel(s0[:, 2], a0d.grad) < 2e-6
    assert float(s0[:, 1].abs().max()) == 0.0


def test_mlp_head1_form_exists_for_the_decoder_shape_only(dev):
    assert K.mlp_head1_supported(S=32, C_=128, T=32000, H=64)
    assert not K.mlp_head1_supported(S=32, C_=256, T=800, H=64)
    assert not K.mlp_head1_supported(S=32, C_=128, T=800, H=256)

"""The SincNet layer (one input channel) on the bf16 matrix pipe: pase_amd/csrc/sinc_x6.hip against fp64 torch references.

  forward          y = F.conv1d(F.pad(x, reflect), filt)          (SincConv_fast.forward, pase/models/modules.py:916-934)
  weight gradient  dfilt = d/dfilt sum(y * g)                     (what autograd hands to sinc_filters' backward)

Both must have the error of a GOOD fp32 evaluation (relative L2 against fp64 <= 1e-6; measured ~2e-7) and must really run on
that kernel (plan kind 3 / 5); PASE_SINC_X6=0 keeps the layer on the exact-fp32 matrix pipe (A/B runs).
"""
import pytest
import torch
import torch.nn.functional as F

from pase_amd import kernels as K


@pytest.fixture(autouse=True)
def _x6_on():
    saved = K.X6
    K.X6 = True
    yield
    K.X6 = saved


def _rel(a, ref):
    return float((a.cpu().double() - ref).norm() / ref.norm().clamp_min(1e-300))


@pytest.mark.parametrize("M,taps,T,S,pad_mode", [
    (64, 251, 700, 2, "reflect"),      # the PASE+ layer: 16 tap groups, three column tiles per sequence (the last one ragged)
    (64, 251, 512, 1, "reflect"),      # whole tiles
    (40, 101, 300, 3, "reflect"),      # fewer filters than a row tile pair, 7 tap groups, two tiles per sequence
    (64, 64, 256, 2, "zero"),          # zero padding, even tap count (asymmetric pads)
])
def test_sinc_forward(dev, M, taps, T, S, pad_mode):
    torch.manual_seed(0)
    x = torch.randn(S, 1, T) * 0.3
    filt = torch.randn(M, taps) * 0.1
    P = (taps // 2 - 1, taps // 2) if taps % 2 == 0 else (taps // 2, taps // 2)
    xp = F.pad(x.double(), P, mode="reflect") if pad_mode == "reflect" else F.pad(x.double(), P)
    ref = F.conv1d(xp, filt.double()[:, None, :])
    y = torch.zeros(S, M, T, device=dev)
    stat = K.conv_gemm(x.to(dev), filt.to(dev), y, want_stats=True, S=S, Cin=1, Tin=T, M=M, K=taps, taps=taps, Ncols=T, Tout=T,
                       padL=P[0], pad_mode=K.PAD_REFLECT if pad_mode == "reflect" else K.PAD_ZERO)
    assert K.LAST_PLAN_KIND == 3
    assert _rel(y, ref) < 1e-6
    st = stat.cpu().double().sum(0)
    torch.testing.assert_close(st[:, 0], ref.sum((0, 2)), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(st[:, 1], (ref ** 2).sum((0, 2)), rtol=1e-5, atol=1e-4)


def test_sinc_forward_into_a_channel_slice_with_bias(dev):
    torch.manual_seed(1)
    S, M, taps, T = 2, 48, 129, 400
    xw = torch.randn(S, 3, T)
    filt = torch.randn(M, taps) * 0.1
    b = torch.randn(M)
    ref = F.conv1d(F.pad(xw[:, 1:2].double(), (64, 64), mode="reflect"), filt.double()[:, None, :], b.double())
    yw = torch.full((S, M + 5, T), 7.0, device=dev)
    K.conv_gemm(xw.to(dev), filt.to(dev), yw, S=S, Cin=1, Tin=T, M=M, K=taps, taps=taps, Ncols=T, Tout=T, padL=64,
                pad_mode=K.PAD_REFLECT, x_ctot=3, x_coff=1, y_ctot=M + 5, y_coff=2, Cout_store=M, bias=b.to(dev))
    assert K.LAST_PLAN_KIND == 3
    assert _rel(yw[:, 2:2 + M], ref) < 1e-6
    assert float(yw[:, :2].min()) == 7.0 and float(yw[:, 2 + M:].min()) == 7.0


@pytest.mark.parametrize("M,taps,T,S,splitk", [
    (64, 251, 700, 2, 0),         # the PASE+ layer; the last stage of a sequence is ragged (700 = 10 * 64 + 60)
    (64, 251, 640, 3, 5),         # five position ranges: stages of different sequences in one workgroup
    (30, 101, 300, 2, 1),         # one workgroup walks everything
])
def test_sinc_weight_gradient(dev, M, taps, T, S, splitk):
    torch.manual_seed(2)
    x = torch.randn(S, 1, T) * 0.3
    P = (taps // 2, taps // 2)
    filt = torch.randn(M, taps, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(F.pad(x.double(), P, mode="reflect"), filt[:, None, :])
    g = torch.randn(y.shape)
    (y * g.double()).sum().backward()
    dw = torch.full((M, taps), 0.5, device=dev)            # a += target
    K.wgrad_gemm(g.to(dev), x.to(dev), dw, S=S, M=M, Tg=T, Ncols=T, Cin=1, Tz=T, taps=taps, padL=P[0], pad_mode=K.PAD_REFLECT,
                 splitk=splitk)
    assert K.LAST_WGRAD_X6 and K.LAST_WGRAD_KIND == 5
    assert _rel(dw - 0.5, filt.grad) < 1e-6


def test_the_fp32_pipe_is_still_selectable(dev, monkeypatch):
    monkeypatch.setenv("PASE_SINC_X6", "0")
    torch.manual_seed(3)
    S, M, taps, T = 1, 64, 251, 300
    x = torch.randn(S, 1, T)
    filt = torch.randn(M, taps) * 0.1
    ref = F.conv1d(F.pad(x.double(), (125, 125), mode="reflect"), filt.double()[:, None, :])
    y = torch.zeros(S, M, T, device=dev)
    K.conv_gemm(x.to(dev), filt.to(dev), y, S=S, Cin=1, Tin=T, M=M, K=taps, taps=taps, Ncols=T, Tout=T, padL=125,
                pad_mode=K.PAD_REFLECT)
    assert K.LAST_PLAN_KIND == 0
    assert _rel(y, ref) < 2e-6

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


@pytest.mark.parametrize("M,taps,T,S,has_bn,pool_d,splitk", [
    (64, 251, 640, 2, 1, 160, 0),      # the PASE+ layer: batch statistics, reflect-folded data gradient, pooled dense-skip gradient
    (64, 251, 700, 2, 1, 0, 3),        # ragged last stage, no pooled branch, stages of two sequences in one workgroup
    (30, 101, 300, 2, 2, 0, 1),        # frozen statistics (dy = scale * dz), fewer filters than the row tile
    (48, 65, 200, 1, 0, 50, 0),        # no norm (dy = dz)
])
def test_sinc_weight_gradient_with_the_norm_backward_on_load(dev, M, taps, T, S, has_bn, pool_d, splitk):
    """pase_wgrad_gemm_act_bwd: the gradient operand is the apply pass of the BatchNorm + PReLU backward evaluated while it is
    staged (dy never written).  Judged against an fp64 autograd evaluation of the same chain
    (F.pad(reflect) / view(..).mean(3) / batch_norm / prelu, pase/models/modules.py:1061-1077, frontend.py:213-232), and
    against the two-launch form (apply pass, then the plain weight gradient) on the same device."""
    torch.manual_seed(5)
    x = torch.randn(S, 1, T) * 0.3
    y = torch.randn(S, M, T) * 1.5 + 0.2
    gamma, beta, alpha = torch.rand(M) + 0.5, torch.randn(M) * 0.2, torch.rand(M) * 0.5
    pL, pR = 9, 10
    dsrc = torch.randn(S, M + 3, T + pL + pR) * 0.1                  # a channel slice of a wider padded data gradient
    F_ = T // pool_d if pool_d else 0
    dpool = torch.randn(S, M + 2, F_) * 0.1 if pool_d else None
    # ---- fp64 reference ---------------------------------------------------------------------------------------------------
    yd = y.double().requires_grad_(True)
    mean, var = yd.detach().mean((0, 2)), yd.detach().var((0, 2), unbiased=False)
    if has_bn == 1:
        z = F.batch_norm(yd, None, None, gamma.double(), beta.double(), True, 0.0, 1e-5)
    elif has_bn == 2:
        z = F.batch_norm(yd, mean.clone(), var.clone(), gamma.double(), beta.double(), False, 0.0, 1e-5)
    else:
        z = yd
    a = F.prelu(z, alpha.double())
    loss = (F.pad(a, (pL, pR), mode="reflect") * dsrc[:, 1:1 + M].double()).sum()
    if pool_d:
        loss = loss + (a[:, :, :F_ * pool_d].reshape(S, M, F_, pool_d).mean(3) * dpool[:, 2:2 + M].double()).sum()
    dy_ref, = torch.autograd.grad(loss, yd)
    P = (taps // 2, taps // 2)
    filt = torch.zeros(M, taps, dtype=torch.float64, requires_grad=True)
    (F.conv1d(F.pad(x.double(), P, mode="reflect"), filt[:, None, :]) * dy_ref).sum().backward()
    # ---- device -----------------------------------------------------------------------------------------------------------
    rstd = (var + 1e-5).rsqrt()
    scale = (gamma.double() * rstd).float() if has_bn else None
    shift = (beta.double() - mean * gamma.double() * rstd).float() if has_bn else None
    t = lambda v: None if v is None else v.to(dev)
    sums = torch.zeros(M, 3, dtype=torch.float64, device=dev)
    kw = dict(S=S, C_=M, T=T, dsrc=t(dsrc), dsrc_ctot=M + 3, dsrc_coff=1, Tp=T + pL + pR, padL=pL, pad_mode=K.PAD_REFLECT,
              dpool=t(dpool), dpool_ctot=M + 2 if pool_d else 0, dpool_coff=2 if pool_d else 0, pool_F=F_, pool_d=max(1, pool_d),
              scale=t(scale), shift=t(shift), alpha=t(alpha), mean=t(mean.float()) if has_bn else None,
              rstd=t(rstd.float()) if has_bn else None, sums=sums, dy=None, has_bn=has_bn)
    yv = y.to(dev)
    K.act_bwd_reduce(yv, **kw)
    dw = torch.full((M, taps), 0.5, device=dev)
    assert K.wgrad_gemm(None, x.to(dev), dw, S=S, M=M, Tg=T, Ncols=T, Cin=1, Tz=T, taps=taps, padL=P[0], pad_mode=K.PAD_REFLECT,
                        splitk=splitk, g_bwd=dict(kw, y=yv))
    assert K.LAST_WGRAD_X6 and K.LAST_WGRAD_KIND == 5
    assert _rel(dw - 0.5, filt.grad) < 1e-6
    # the two-launch form
    dy = torch.empty(S, M, T, device=dev)
    if has_bn == 1:
        K.act_bwd_apply(yv, **dict(kw, dy=dy))
    else:
        K.act_bwd_reduce(yv, **dict(kw, dy=dy, sums=torch.zeros_like(sums)))
    assert _rel(dy, dy_ref) < 1e-6
    dw2 = torch.zeros(M, taps, device=dev)
    K.wgrad_gemm(dy, x.to(dev), dw2, S=S, M=M, Tg=T, Ncols=T, Cin=1, Tz=T, taps=taps, padL=P[0], pad_mode=K.PAD_REFLECT,
                 splitk=splitk)
    assert _rel(dw - 0.5, dw2.cpu().double()) < 1e-6


def test_on_load_form_on_sequences_shorter_than_the_window_image(dev):
    """with reflect padding and a sequence shorter than the staged window image (Tz below ~330) some window samples
    stay outside [0, Tz) after ONE reflection; the on-load form never copies them into its raw-sample LDS buffer and must
    select 0 there (as the plain form does) instead of reading what the previous stage / launch left behind: those windows
    only meet dead positions (g = 0), but 0 * NaN is NaN.  A first launch whose samples are all NaN leaves NaN patterns in
    that buffer (the emulator's LDS persists per worker thread; on the GPU LDS contents are simply undefined), the second,
    short launch must come out finite and equal to fp64."""
    M, taps, S = 48, 65, 3
    P = (taps // 2, taps // 2)

    def run(x, y, T):
        sums = torch.zeros(M, 3, dtype=torch.float64, device=dev)
        kw = dict(S=S, C_=M, T=T, alpha=(torch.rand(M) * 0.5 + 0.75).to(dev), sums=sums, dy=None, has_bn=0)
        yv = y.to(dev)
        K.act_bwd_reduce(yv, **kw)
        dw = torch.zeros(M, taps, device=dev)
        assert K.wgrad_gemm(None, x.to(dev), dw, S=S, M=M, Tg=T, Ncols=T, Cin=1, Tz=T, taps=taps, padL=P[0],
                            pad_mode=K.PAD_REFLECT, g_bwd=dict(kw, y=yv))
        assert K.LAST_WGRAD_KIND == 5
        return dw, kw["alpha"]
    torch.manual_seed(11)
    for _ in range(3):            # poison every worker's buffer
        run(torch.full((S, 1, 640), float("nan")), torch.randn(S, M, 640), 640)
    T = 40
    x, y = torch.randn(S, 1, T) * 0.3, torch.randn(S, M, T)
    dw, alpha = run(x, y, T)
    assert bool(torch.isfinite(dw).all())
    # without a norm and without a data gradient / pooled branch dA = 0: dy = 0, so dfilt must be exactly zero ...
    assert float(dw.abs().max()) == 0.0
    # ... and with a data gradient: against fp64 autograd
    dsrc = torch.randn(S, M, T) * 0.1
    sums = torch.zeros(M, 3, dtype=torch.float64, device=dev)
    kw = dict(S=S, C_=M, T=T, alpha=alpha, dsrc=dsrc.to(dev), dsrc_ctot=M, Tp=T, padL=0, pad_mode=K.PAD_ZERO, sums=sums, dy=None,
              has_bn=0)
    yv = y.to(dev)
    K.act_bwd_reduce(yv, **kw)
    dw = torch.zeros(M, taps, device=dev)
    assert K.wgrad_gemm(None, x.to(dev), dw, S=S, M=M, Tg=T, Ncols=T, Cin=1, Tz=T, taps=taps, padL=P[0], pad_mode=K.PAD_REFLECT,
                        g_bwd=dict(kw, y=yv))
    yd = y.double().requires_grad_(True)
    dy_ref, = torch.autograd.grad((F.prelu(yd, alpha.cpu().double()) * dsrc.double()).sum(), yd)
    filt = torch.zeros(M, taps, dtype=torch.float64, requires_grad=True)
    (F.conv1d(F.pad(x.double(), P, mode="reflect"), filt[:, None, :]) * dy_ref).sum().backward()
    assert bool(torch.isfinite(dw).all())
    assert _rel(dw, filt.grad) < 1e-6


def test_the_on_load_form_refuses_what_it_cannot_stage_and_enqueues_nothing(dev):
    """pase_wgrad_gemm_act_bwd_ok: a pooled dense-skip branch with pool_d < 16 (a staged run of 16 positions would touch more
    than two pooled frames) has no on-load form; wgrad_gemm(g_bwd=...) returns False and leaves dw untouched."""
    S, M, T, taps = 2, 16, 128, 65
    y = torch.randn(S, M, T, device=dev)
    dpool = torch.randn(S, M, T // 8, device=dev)
    dw = torch.full((M, taps), 0.25, device=dev)
    g_bwd = dict(y=y, S=S, C_=M, T=T, has_bn=0, dpool=dpool, dpool_ctot=M, pool_F=T // 8, pool_d=8)
    args = dict(S=S, M=M, Tg=T, Ncols=T, Cin=1, Tz=T, taps=taps, padL=32, pad_mode=K.PAD_REFLECT)
    assert K.wgrad_gemm(None, torch.randn(S, 1, T, device=dev), dw, g_bwd=g_bwd, **args) is False
    assert K.LAST_WGRAD_KIND == 5                      # the plan was right, the operand description was not
    assert float((dw - 0.25).abs().max()) == 0.0
    assert K.wgrad_gemm(None, torch.randn(S, 1, T, device=dev), dw, g_bwd=dict(g_bwd, pool_d=16, pool_F=T // 16,
                        dpool=dpool[:, :, :T // 16].contiguous()), **args) is True


def test_the_on_load_form_exists_only_on_the_one_channel_plan(dev):
    S, M, Cin, T = 1, 64, 8, 128
    g_bwd = dict(y=torch.zeros(S, M, T, device=dev), S=S, C_=M, T=T, has_bn=0)
    dw = torch.zeros(M, Cin * 3, device=dev)
    assert K.wgrad_gemm(None, torch.zeros(S, Cin, T, device=dev), dw, S=S, M=M, Tg=T, Ncols=T, Cin=Cin, Tz=T, taps=3, padL=1,
                        g_bwd=g_bwd) is False

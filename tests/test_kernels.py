"""Unit parity tests of the non-GEMM kernels and the wgrad contraction against torch fp32 / autograd."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pase_amd import kernels as K
from util import GOLD, assert_close


def test_wgrad_conv_reflect_stride(dev):
    torch.manual_seed(0)
    S, Cin, Cout, k, st, T = 3, 5, 70, 11, 2, 80
    x = torch.randn(S, Cin, T, requires_grad=False)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
    w = torch.randn(Cout, Cin, k, requires_grad=True)
    b = torch.zeros(Cout, requires_grad=True)
    xin = x * sc[None, :, None] + sh[None, :, None]
    xin = torch.where(xin > 0, xin, xin * al[None, :, None])
    y = F.conv1d(F.pad(xin, (4, 5), mode="reflect"), w, b, stride=st)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    dw = torch.zeros(Cout, Cin * k, device=dev)
    db = torch.zeros(Cout, device=dev)
    K.wgrad_gemm(g.to(dev), x.to(dev), dw, S=S, M=Cout, Tg=y.shape[2], Ncols=y.shape[2], Cin=Cin, Tz=T, taps=k,
                 dbias=db, in_scale=sc.to(dev), in_shift=sh.to(dev), in_alpha=al.to(dev), stride=st, padL=4,
                 pad_mode=K.PAD_REFLECT)
    assert_close(dw.view(Cout, Cin, k), w.grad, rtol=1e-4, atol=1e-3, what="dW")
    assert_close(db, b.grad, rtol=1e-4, atol=1e-3, what="db")


def test_wgrad_flat_1x1_and_narrow(dev):
    torch.manual_seed(1)
    # the last two have Cin a multiple of the column tile: bias gradient via G row sums, no ones-column tile
    for (S, Cin, Cout, T) in [(5, 40, 9, 37), (4, 20, 150, 50), (3, 128, 70, 40), (2, 256, 9, 33)]:
        x = torch.randn(S, Cin, T)
        g = torch.randn(S, Cout, T)
        ref = torch.einsum("sot,sct->oc", g, x)
        dw = torch.zeros(Cout, Cin, device=dev)
        db = torch.zeros(Cout, device=dev)
        K.wgrad_gemm(g.to(dev), x.to(dev), dw, S=S, M=Cout, Tg=T, Ncols=T, Cin=Cin, Tz=T, taps=1, dbias=db)
        assert_close(dw, ref, rtol=1e-4, atol=1e-3)
        assert_close(db, g.sum((0, 2)), rtol=1e-4, atol=1e-3)


def test_wgrad_flat_vector_kernel(dev):
    """1x1 layers with T % 4 == 0 take the dedicated NT kernel: ragged M / Cin, chunk tail (S*T % 32 != 0),
    channel slices of wider tensors, on-load PReLU on G and affine + PReLU on Z, bias from G row sums."""
    torch.manual_seed(7)
    for (S, Cin, Cout, T, gx, zx) in [(3, 84, 273, 200, 0, 0), (2, 130, 150, 36, 5, 7), (5, 256, 64, 20, 0, 3),
                                     (1, 12, 300, 4, 2, 0)]:
        xw = torch.randn(S, Cin + zx + 2, T)
        gw = torch.randn(S, Cout + gx + 3, T)
        sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin), torch.rand(Cin) * 0.5
        ga = torch.rand(Cout) * 0.5
        x = xw[:, zx:zx + Cin] * sc[None, :, None] + sh[None, :, None]
        x = torch.where(x > 0, x, x * al[None, :, None])
        g = gw[:, gx:gx + Cout]
        g = torch.where(g > 0, g, g * ga[None, :, None])
        ref = torch.einsum("sot,sct->oc", g.double(), x.double()).float()
        dw = torch.zeros(Cout, Cin, device=dev)
        db = torch.zeros(Cout, device=dev)
        K.wgrad_gemm(gw.to(dev), xw.to(dev), dw, S=S, M=Cout, Tg=T, Ncols=T, Cin=Cin, Tz=T, taps=1, dbias=db,
                     g_ctot=gw.shape[1], g_coff=gx, z_ctot=xw.shape[1], z_coff=zx, in_scale=sc.to(dev),
                     in_shift=sh.to(dev), in_alpha=al.to(dev), g_alpha=ga.to(dev))
        assert_close(dw, ref, rtol=1e-4, atol=2e-3, what="dW %s" % ((S, Cin, Cout, T),))
        assert_close(db, g.sum((0, 2)), rtol=1e-4, atol=2e-3, what="db")


def test_wgrad_conv_transpose_roles_swapped(dev):
    """nn.ConvTranspose1d weight gradient: G = PReLU(layer input) at the low rate, Z = dY."""
    torch.manual_seed(2)
    S, Cin, Cout, k, st, T = 2, 70, 6, 30, 4, 12
    z_in = torch.randn(S, Cin, T)
    al = torch.rand(Cin) * 0.5
    w = torch.randn(Cin, Cout, k, requires_grad=True)
    a = torch.where(z_in > 0, z_in, z_in * al[None, :, None])
    pad = (k - st) // 2
    y = F.conv_transpose1d(a, w, None, stride=st, padding=pad)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    dw = torch.zeros(Cin, Cout * k, device=dev)
    K.wgrad_gemm(z_in.to(dev), g.to(dev), dw, S=S, M=Cin, Tg=T, Ncols=T, Cin=Cout, Tz=y.shape[2], taps=k,
                 stride=st, padL=pad, pad_mode=K.PAD_ZERO, g_alpha=al.to(dev))
    assert_close(dw.view(Cin, Cout, k), w.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("F_", [7, 200, 625])
def test_qrnn_scan_fwd_bwd(dev, F_):
    """ForgetMult recurrence incl. the multi-chunk carry (F > 256) vs a sequential torch loop."""
    torch.manual_seed(3)
    S, H = 2, 5
    gates = torch.randn(S, 3 * H, F_, requires_grad=True)
    Z, Fg, O = gates.chunk(3, 1)
    Z, Fg, O = torch.tanh(Z), torch.sigmoid(Fg), torch.sigmoid(O)
    cs, c = [], None
    for t in range(F_):
        ct = Fg[:, :, t] * Z[:, :, t]
        if c is not None:
            ct = ct + (1 - Fg[:, :, t]) * c
        cs.append(ct)
        c = ct
    C = torch.stack(cs, 2)
    Hh = O * C
    dh = torch.randn_like(Hh)
    (Hh * dh).sum().backward()
    hbuf = torch.zeros(S, H + 3, F_, device=dev)       # write into a channel slice of a wider buffer
    cbuf = torch.zeros(S, H, F_, device=dev)
    gd = gates.detach().to(dev)
    K.qrnn_scan_fwd(gd, hbuf, cbuf, S=S, H=H, F=F_, h_ctot=H + 3, h_coff=2)
    assert_close(hbuf[:, 2:2 + H], Hh, rtol=1e-5, atol=1e-5)
    assert_close(cbuf, C, rtol=1e-5, atol=1e-5)
    dg = torch.zeros(S, 3 * H, F_, device=dev)
    dhw = torch.zeros(S, H + 1, F_)
    dhw[:, 1:] = dh
    K.qrnn_scan_bwd(gd, cbuf, dhw.to(dev), dg, S=S, H=H, F=F_, dh_ctot=H + 1, dh_coff=1)
    assert_close(dg, gates.grad, rtol=1e-4, atol=1e-5)


def test_sinc_filters_and_gradient(dev):
    from oracle import pase_oracle as O
    p = np.load(os.path.join(GOLD, "sinc_perturbed.npz"))
    low, band = torch.tensor(p["low_hz_"]), torch.tensor(p["band_hz_"])
    win, n_ = O.sinc_constants()
    filt = torch.zeros(64, 251, device=dev)
    args = dict(C_=64, Kw=251, min_low=50.0, min_band=50.0, sr=16000.0)
    K.sinc_filters(low.to(dev), band.to(dev), n_.contiguous().to(dev), win.contiguous().to(dev), filt, **args)
    assert_close(filt, p["filters"][:, 0], rtol=1e-5, atol=2e-6, what="filters vs live reference")
    # gradient: dF from the reference conv, chained by the HIP kernel
    lo = torch.tensor(p["low_hz_"], requires_grad=True)
    ba = torch.tensor(p["band_hz_"], requires_grad=True)
    f = O.sinc_filters(lo, ba)
    f.retain_grad()
    y = F.conv1d(F.pad(torch.tensor(p["x"]), (125, 125), mode="reflect"), f)
    (y * torch.tensor(p["g"])).sum().backward()
    dlow, dband = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    K.sinc_filters_bwd(low.to(dev), band.to(dev), n_.contiguous().to(dev), win.contiguous().to(dev),
                       f.grad[:, 0].contiguous().to(dev), dlow, dband, **args)
    assert_close(dlow, p["dlow"][:, 0], rtol=1e-3, atol=1e-6 * float(np.abs(p["dlow"]).max()) + 1e-7)
    assert_close(dband, p["dband"][:, 0], rtol=1e-3, atol=1e-6 * float(np.abs(p["dband"]).max()) + 1e-7)


def test_adam_matches_torch(dev):
    torch.manual_seed(4)
    p0 = torch.randn(1000)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=3e-3)
    p = p0.clone().to(dev)
    m, v = torch.zeros(1000, device=dev), torch.zeros(1000, device=dev)
    lr = torch.tensor([3e-3], device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    for i in range(5):
        g = torch.randn(1000)
        pt.grad = g.clone()
        opt.step()
        K.step_tick(step)
        K.adam_step(p, (2.0 * g).to(dev), m, v, lr, step, grad_mul=0.5)
    assert int(step.item()) == 5
    assert_close(p, pt.detach(), rtol=1e-5, atol=1e-6)


def test_bn_finalize_running_stats(dev):
    torch.manual_seed(5)
    C, S, T = 6, 4, 50
    y = torch.randn(S, C, T) * 2 + 1
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    ref = bn(y)
    # emulate the conv epilogue partials: 3 column tiles
    parts = torch.stack([torch.stack([y[:, :, a:b].sum((0, 2)), (y[:, :, a:b] ** 2).sum((0, 2))], 1)
                         for a, b in ((0, 20), (20, 35), (35, 50))]).contiguous()
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    sc, sh, mean, rstd = (torch.zeros(C, device=dev) for _ in range(4))
    K.bn_finalize(parts.to(dev), C, S * T, bn.weight.detach().to(dev), bn.bias.detach().to(dev), 1e-5, 0.1, rm, rv,
                  sc, sh, mean, rstd)
    out = torch.zeros(S, C, T, device=dev)
    K.bn_act_apply(y.to(dev), out, sc, sh, None, S=S, C_=C, T=T)
    assert_close(out, ref, rtol=1e-5, atol=1e-5)
    assert_close(rm, bn.running_mean, rtol=1e-5, atol=1e-6)
    assert_close(rv, bn.running_var, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("d", [1, 4, 16, 160])
def test_bn_act_pool(dev, d):
    torch.manual_seed(6)
    S, C, F_ = 2, 3, 5
    y = torch.randn(S, C, F_ * d + (1 if d > 1 else 0))
    sc, sh, al = torch.rand(C) + 0.5, torch.randn(C), torch.rand(C)
    a = y * sc[None, :, None] + sh[None, :, None]
    a = torch.where(a > 0, a, a * al[None, :, None])
    ref = a[:, :, :F_ * d].reshape(S, C, F_, d).mean(3)
    out = torch.zeros(S, C + 2, F_, device=dev)
    K.bn_act_pool(y.to(dev), out, sc.to(dev), sh.to(dev), al.to(dev), S=S, C_=C, T=y.shape[2], F=F_, d=d,
                  o_ctot=C + 2, o_coff=1)
    assert_close(out[:, 1:1 + C], ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("T", [24, 2052, 8200], ids=["wave-per-row", "block-per-row", "two-segments"])
def test_act_backward_reflect_fold_and_pool_branch(dev, T):
    """d/dy of sum(g1 * conv-input-padded(a)) + sum(g2 * meanpool(a)), a = PReLU(BN_train(y)); the three work
    decompositions of the kernel (one wave per short row, one block per row, several segments per long row)."""
    torch.manual_seed(7)
    S, C, pL, pR, d = (3, 4, 4, 5, 4) if T < 100 else (2, 3, 4, 5, 4)
    y = torch.randn(S, C, T, requires_grad=True)
    gamma = (torch.rand(C) + 0.5).requires_grad_(True)
    beta = torch.randn(C, requires_grad=True)
    al = (torch.rand(C) * 0.5).requires_grad_(True)
    z = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)
    a = F.prelu(z, al)
    g1 = torch.randn(S, C, T + pL + pR)
    g2 = torch.randn(S, C, T // d)
    loss = (F.pad(a, (pL, pR), mode="reflect") * g1).sum() + (a.view(S, C, T // d, d).mean(3) * g2).sum()
    loss.backward()
    mean = y.detach().mean((0, 2))
    var = y.detach().var((0, 2), unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    scale = gamma.detach() * rstd
    shift = beta.detach() - mean * scale
    sums = torch.zeros(C, 3, dtype=torch.float64, device=dev)
    dy = torch.zeros(S, C, T, device=dev)
    kw = dict(S=S, C_=C, T=T, dsrc=g1.to(dev), Tp=T + pL + pR, padL=pL, pad_mode=K.PAD_REFLECT, dpool=g2.to(dev),
              dpool_ctot=C, dpool_coff=0, pool_F=T // d, pool_d=d, scale=scale.to(dev), shift=shift.to(dev),
              alpha=al.detach().to(dev), mean=mean.to(dev), rstd=rstd.to(dev), sums=sums, dy=dy, has_bn=1)
    K.act_bwd_reduce(y.detach().to(dev), **kw)
    K.act_bwd_apply(y.detach().to(dev), **kw)
    assert_close(dy, y.grad, rtol=1e-4, atol=1e-5, what="dy")
    big = 1e-5 * max(1.0, T / 100.0)       # sums over S*T terms of O(1)
    assert_close(sums[:, 0], beta.grad, rtol=1e-4, atol=big, what="dbeta")
    assert_close(sums[:, 1], gamma.grad, rtol=1e-4, atol=big, what="dgamma")
    assert_close(sums[:, 2], al.grad, rtol=1e-4, atol=big, what="dalpha")


def test_act_backward_without_batchnorm_is_single_pass(dev):
    """PReLU only (decoder / worker hidden layers): the reduce pass writes dy = dz itself (zero-padded dA in padded
    coordinates, channel slice of a wider buffer)."""
    torch.manual_seed(9)
    S, C, T, pL = 2, 5, 37, 3
    yw = torch.randn(S, C + 3, T)
    y = yw[:, 2:2 + C].clone().requires_grad_(True)
    al = (torch.rand(C) * 0.5).requires_grad_(True)
    g = torch.randn(S, C, T + 2 * pL)
    (F.pad(F.prelu(y, al), (pL, pL)) * g).sum().backward()
    sums = torch.zeros(C, 3, dtype=torch.float64, device=dev)
    dyw = torch.full((S, C + 3, T), 7.0, device=dev)
    K.act_bwd_reduce(yw.to(dev), S=S, C_=C, T=T, y_ctot=C + 3, y_coff=2, dsrc=g.to(dev), Tp=T + 2 * pL, padL=pL,
                     pad_mode=K.PAD_ZERO, alpha=al.detach().to(dev), sums=sums, dy=dyw, has_bn=0)
    assert_close(dyw[:, 2:2 + C], y.grad, rtol=1e-5, atol=1e-6, what="dy")
    assert float((dyw[:, :2] - 7.0).abs().max()) == 0.0
    assert_close(sums[:, 2], al.grad, rtol=1e-4, atol=1e-5, what="dalpha")
    assert_close(sums[:, 0], y.grad.sum((0, 2)), rtol=1e-4, atol=1e-5, what="sum dz")


@pytest.mark.parametrize("loss_name,lt", [("L1Loss", K.LOSS_L1), ("BCEWithLogitsLoss", K.LOSS_BCE)])
def test_head1_forward_backward(dev, loss_name, lt):
    torch.manual_seed(8)
    S, C, T = 3, 6, 40
    z = torch.randn(S, C, T, requires_grad=True)
    al = (torch.rand(C) * 0.5).requires_grad_(True)
    w = torch.randn(1, C, 1, requires_grad=True)
    b = torch.randn(1, requires_grad=True)
    y = F.conv1d(F.prelu(z, al), w, b)
    tgt = torch.rand(S, 1, T) if lt == K.LOSS_BCE else torch.randn(S, 1, T)
    loss = getattr(torch.nn, loss_name)()(y, tgt)
    loss.backward()
    yk = torch.zeros(S, 1, T, device=dev)
    dyk = torch.zeros(S, 1, T, device=dev)
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    n = S * T
    K.head1_fwd(z.detach().to(dev), w.detach().view(-1).to(dev), b.detach().to(dev), S=S, C_=C, T=T,
                in_alpha=al.detach().to(dev), target=tgt.to(dev), y=yk, dy=dyk, loss_acc=acc, loss_type=lt,
                grad_scale=1.0 / n)
    assert_close(yk, y, rtol=1e-5, atol=1e-5)
    assert abs(float(acc) / n - float(loss)) < 1e-5
    dz = torch.zeros(S, C, T, device=dev)
    sums = torch.zeros(C * 3 + 1, dtype=torch.float64, device=dev)
    K.head1_bwd(z.detach().to(dev), al.detach().to(dev), w.detach().view(-1).to(dev), dyk, dz, sums, S=S, C_=C, T=T)
    assert_close(dz, z.grad, rtol=1e-4, atol=1e-6)
    s3 = sums[:C * 3].view(C, 3)
    assert_close(s3[:, 0], w.grad.view(-1), rtol=1e-4, atol=1e-6)
    assert_close(s3[:, 1], al.grad, rtol=1e-4, atol=1e-6)
    assert_close(s3[:, 2], z.grad.sum((0, 2)), rtol=1e-4, atol=1e-6)
    assert_close(sums[C * 3:], b.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("Cout", [40, 70], ids=["narrow-64x256", "wide-128x128"])
def test_wgrad_large_span_float4_staging(dev, Cout):
    """The stride-10 / 20-tap block-1 geometry (pase/models/modules.py:1058-1071 pads (9, 10)): 14 channels x 330
    samples per stage do not fit the scalar small-span slab -> the 16-byte staged instantiation (aligned loads 3
    samples in front of the span, rows padded to whole float4s), incl. the reflect-padded first / last chunks."""
    torch.manual_seed(3)
    S, Cin, k, st, T = 2, 16, 20, 10, 1280
    x = torch.randn(S, Cin, T)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
    w = torch.randn(Cout, Cin, k, requires_grad=True)
    b = torch.zeros(Cout, requires_grad=True)
    xin = x * sc[None, :, None] + sh[None, :, None]
    xin = torch.where(xin > 0, xin, xin * al[None, :, None])
    y = F.conv1d(F.pad(xin, (9, 10), mode="reflect"), w, b, stride=st)
    assert y.shape[2] == T // st
    g = torch.randn_like(y)
    (y * g).sum().backward()
    dw = torch.zeros(Cout, Cin * k, device=dev)
    db = torch.zeros(Cout, device=dev)
    K.wgrad_gemm(g.to(dev), x.to(dev), dw, S=S, M=Cout, Tg=y.shape[2], Ncols=y.shape[2], Cin=Cin, Tz=T, taps=k,
                 dbias=db, in_scale=sc.to(dev), in_shift=sh.to(dev), in_alpha=al.to(dev), stride=st, padL=9,
                 pad_mode=K.PAD_REFLECT)
    assert_close(dw.view(Cout, Cin, k), w.grad, rtol=1e-4, atol=2e-3, what="dW")
    assert_close(db, b.grad, rtol=1e-4, atol=2e-3, what="db")


@pytest.mark.parametrize("Cin,Cout,k,st,T,S", [(32, 130, 11, 1, 400, 2), (32, 70, 11, 2, 400, 3), (128, 130, 1, 1, 200, 3),
                                               (8, 40, 20, 10, 3000, 2)])
def test_wgrad_split_bf16_is_fp32_grade(dev, Cin, Cout, k, st, T, S):
    """PaseWgrad::x6 (both operands split into three bf16 pieces on the fly) vs an fp64 reference: fp32-grade error,
    compared with the fp32-pipe launch of the same call."""
    torch.manual_seed(5)
    x = torch.randn(S, Cin, T)
    P = (0, 0) if k == 1 else ((k // 2 - 1, k // 2) if (st > 1 or k % 2 == 0) else (k // 2, k // 2))
    w = torch.randn(Cout, Cin, k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    xp = F.pad(x.double(), P, mode="reflect") if k > 1 else x.double()
    y = F.conv1d(xp, w, b, stride=st)
    g = torch.randn(y.shape)
    (y * g.double()).sum().backward()
    err = {}
    saved = K.X6
    try:
        for mode in (True, False):
            K.X6 = mode
            dw = torch.zeros(Cout, Cin * k, device=dev)
            db = torch.zeros(Cout, device=dev)
            K.wgrad_gemm(g.to(dev), x.to(dev), dw, S=S, M=Cout, Tg=y.shape[2], Ncols=y.shape[2], Cin=Cin, Tz=T, taps=k,
                         dbias=db, stride=st, padL=P[0], pad_mode=K.PAD_REFLECT if k > 1 else K.PAD_ZERO)
            e_w = ((dw.cpu().double().view_as(w.grad) - w.grad).norm() / w.grad.norm()).item()
            e_b = ((db.cpu().double() - b.grad).norm() / b.grad.norm()).item()
            err[mode] = (e_w, e_b)
    finally:
        K.X6 = saved
    assert err[True][0] < 1e-6 and err[True][1] < 1e-6, err
    assert err[True][0] < 2.0 * err[False][0] + 1e-8, err


def test_add_blocks_commits_column_and_row_slices(dev):
    """pase_add_blocks: dst_k += src_k for several row-major blocks in one launch -- column slices of a concatenated weight
    gradient (row stride on the source) and row slices of a stacked one, into contiguous parameter-gradient buffers."""
    from pase_amd import kernels as K
    torch.manual_seed(3)
    cat = torch.randn(12, 50, device=dev)
    widths = [7, 20, 1, 22]
    dsts = [torch.randn(12, w_, device=dev) for w_ in widths]
    want = []
    off = 0
    pairs = []
    for d_, w_ in zip(dsts, widths):
        want.append(d_.clone() + cat[:, off:off + w_])
        pairs.append((d_, cat[:, off:off + w_]))
        off += w_
    stacked = torch.randn(30, 9, device=dev)
    rows = [torch.randn(10, 9, device=dev) for _ in range(3)]
    for i, r_ in enumerate(rows):
        want.append(r_.clone() + stacked[10 * i:10 * i + 10])
        pairs.append((r_, stacked[10 * i:10 * i + 10]))
    assert K.add_blocks(pairs)
    for got, w_ in zip(dsts + rows, want):
        torch.testing.assert_close(got, w_, rtol=0, atol=0)
    # more than 16 blocks: several launches; a 3-D operand: refused (the caller falls back to torch)
    many = [(torch.zeros(2, 3, device=dev), torch.ones(2, 3, device=dev)) for _ in range(20)]
    assert K.add_blocks(many) and all(float(d_.sum()) == 6.0 for d_, _ in many)
    assert not K.add_blocks([(torch.zeros(2, 3, 1, device=dev), torch.ones(2, 3, 1, device=dev))])

"""pase_conv_gemm on the split-bf16 channel-minor kernel (pase_amd/csrc/conv_x6c.hip) against fp64 torch references.

Every case asserts that the launch really ran on that kernel (pase_conv_gemm_plan_kind == 2) and that the result has
the error of a GOOD fp32 evaluation: relative L2 against fp64 <= 1e-6 (measured 1e-7 ... 4e-7; a bf16-grade result
would be 1e-3, a dropped split term 1e-5).  Shapes cover what the PASE+ step launches: strided
convs as polyphase channels (stride 2 / 4 / 10), reversed taps (data-gradients, the QRNN Linear), pixel-shuffle stores
(transposed convs) with and without the (channel, phase) row order, 1x1 layers (two k-groups per stage), column tiles
that touch three sequences, 64-row and multi-row-tile launches, ragged channel groups, split-K, BatchNorm statistics,
the fused r-context MSE epilogue and the spectra post-ops.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from pase_amd import kernels as K


@pytest.fixture(autouse=True, params=["split-on-load", "presplit"])
def _x6_on(request, monkeypatch):
    """every case twice: the activation split while it is staged (the default of most shapes), and pre-split by pase_pack_xp
    (forced on every stride-1 launch: PaseConvGemm::x6_ctl bit 1) so that staging is a copy"""
    saved = K.X6
    K.X6 = True
    monkeypatch.setenv("PASE_X6C_XP", "1" if request.param == "presplit" else "0")
    yield request.param
    K.X6 = saved


def _rel(a, ref):
    return float((a.cpu().double() - ref).norm() / ref.norm().clamp_min(1e-300))


def _xf(x, sc, sh, al):
    v = x.double() * sc.double()[None, :, None] + sh.double()[None, :, None]
    return torch.where(v > 0, v, v * al.double()[None, :, None])


@pytest.mark.parametrize("Cin,Cout,k,stride,T,S,xf", [
    (16, 40, 11, 1, 300, 2, True),       # one k-group, 11 steps, M = 40: one ragged row tile
    (32, 130, 11, 1, 250, 3, True),      # 128-row tiles (two of them, second ragged), tiles straddle sequences
    (24, 70, 11, 2, 420, 2, True),       # stride 2: 48 channels', 6 taps' (one zero tap), reflect pad on both phases
    (8, 72, 20, 10, 900, 2, False),      # stride 10: 80 channels', 2 taps' (block 1 with more rows)
    (40, 20, 30, 4, 600, 2, True),       # stride 4: 160 channels', 8 taps' (30 -> 32 taps)
    (56, 96, 3, 1, 90, 5, True),         # 56 channels: ragged second k-group (zero channels'), Ncols < 128: 3 sequences per tile
    (48, 64, 5, 1, 1000, 1, False),      # 64-row launch: half of the 128-row tile is zero weights
    (32, 64, 20, 10, 2700, 3, True),     # 64 rows, 320 channels' x 2 taps': the 64 x 256 tile (block 1), tiles straddle sequences
    (64, 50, 11, 1, 300, 2, True),       # ... 50 rows (ragged), 11 taps, 256-column tiles over two sequences and a ragged end
])
def test_conv_forward(dev, _x6_on, Cin, Cout, k, stride, T, S, xf):
    torch.manual_seed(0)
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin, k) * 0.2
    b = torch.randn(Cout)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
    P = (k // 2 - 1, k // 2) if (stride > 1 or k % 2 == 0) else (k // 2, k // 2)
    xin = _xf(x, sc, sh, al) if xf else x.double()
    ref = F.conv1d(F.pad(xin, P, mode="reflect"), w.double(), b.double(), stride=stride)
    Tout = ref.shape[2]
    y = torch.zeros(S, Cout, Tout, device=dev)
    kw = dict(in_scale=sc.to(dev), in_shift=sh.to(dev), in_alpha=al.to(dev)) if xf else {}
    stat = K.conv_gemm(x.to(dev), w.reshape(Cout, -1).contiguous().to(dev), y, want_stats=True, S=S, Cin=Cin, Tin=T,
                       M=Cout, K=Cin * k, taps=k, Ncols=Tout, Tout=Tout, bias=b.to(dev), stride=stride, padL=P[0],
                       pad_mode=K.PAD_REFLECT, **kw)
    assert K.LAST_PLAN_KIND == 2
    assert K.LAST_XP == (_x6_on == "presplit" and stride == 1)      # pre-split activations: stride-1 launches only
    if Cout <= 64 and Cin * k >= 512 and not K.LAST_XP:             # the 64 x 256 tile: half the column tiles
        assert stat.shape[0] == -(-S * Tout // 256)
    assert _rel(y, ref) < 1e-6
    st = stat.cpu().double().sum(0)
    torch.testing.assert_close(st[:, 0], ref.sum((0, 2)), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(st[:, 1], (ref ** 2).sum((0, 2)), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("Cin,Cout,k,stride,T,S", [
    (64, 16, 30, 10, 140, 2),   # ps = 10: (channel, phase)-ordered rows, quads straddle channels
    (80, 20, 8, 4, 150, 3),     # ps = 4
    (72, 40, 4, 2, 200, 2),     # ps = 2
    (48, 36, 30, 4, 50, 4),     # few columns per sequence (57 per sequence: tiles touch three)
])
def test_conv_transpose_pixel_shuffle(dev, Cin, Cout, k, stride, T, S):
    """nn.ConvTranspose1d (modules.py:558-589) = stride-1 conv with reversed taps, stride * Cout rows, pixel-shuffle store."""
    torch.manual_seed(1)
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cin, Cout, k) * 0.2
    b = torch.randn(Cout)
    al = torch.rand(Cin) * 0.5
    pad = max(0, (stride - k) // -2)
    xin = torch.where(x > 0, x, x * al[None, :, None]).double()
    ref = F.conv_transpose1d(xin, w.double(), b.double(), stride=stride, padding=pad)
    Tout = ref.shape[2]
    taps_p = -(-k // stride)
    wt = K.pack_dgrad_t(w.to(dev), R=Cin, O=Cout, k=k, st=stride, s_red=Cout * k, s_out=k, s_k=1)
    y = torch.zeros(S, Cout, Tout, device=dev)
    K.conv_gemm(x.to(dev), None, y, wt=wt, S=S, Cin=Cin, Tin=T, M=stride * Cout, K=Cin * taps_p, taps=taps_p,
                Ncols=T + taps_p - 1, Tout=Tout, bias=b.to(dev), in_alpha=al.to(dev), stride=1, tapstep=-1, padL=0,
                pad_mode=K.PAD_ZERO, Cout_store=Cout, ps=stride, poff=-pad, splitk=1)
    assert K.LAST_PLAN_KIND == 2
    assert _rel(y, ref) < 1e-6


@pytest.mark.parametrize("splitk", [1, 3])
def test_64_row_tile_with_and_without_splitk(dev, splitk):
    """<= 64 rows, 11 taps, K = 704: the 64 x 256 tile (compute waves 2 x 2); split-K adds the slices' tiles atomically into the
    zeroed output (two column halves per row block), only slice 0 carries the bias (the accumulators' initial value)."""
    torch.manual_seed(21)
    S, Cin, Cout, k, T = 2, 64, 56, 11, 300
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin, k) * 0.1
    b = torch.randn(Cout)
    ref = F.conv1d(F.pad(x.double(), (5, 5), mode="reflect"), w.double(), b.double())
    y = torch.zeros(S, Cout, T, device=dev)
    K.conv_gemm(x.to(dev), w.reshape(Cout, -1).contiguous().to(dev), y, y_zeroed=True, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * k,
                taps=k, Ncols=T, Tout=T, bias=b.to(dev), padL=5, pad_mode=K.PAD_REFLECT, splitk=splitk)
    assert K.LAST_PLAN_KIND == 2
    assert _rel(y, ref) < 1e-6


@pytest.mark.parametrize("splitk", [1, 0, 3])
def test_strided_data_gradient_splitk(dev, splitk):
    """Data-gradient of a stride-2 conv in padded coordinates (engine.conv_dgrad): rows = (phase, input channel),
    reversed taps, pixel-shuffle store, long reduction over the output channels with and without split-K."""
    torch.manual_seed(2)
    S, Cin, Cout, k, stride, T = 2, 12, 160, 11, 2, 200
    w = torch.randn(Cout, Cin, k) * 0.1
    padL, padR = k // 2 - 1, k // 2
    Tg = (T + padL + padR - k) // stride + 1
    dy = torch.randn(S, Cout, Tg)
    xp = torch.zeros(S, Cin, T + padL + padR, dtype=torch.float64, requires_grad=True)
    F.conv1d(xp, w.double(), None, stride=stride).backward(dy.double())
    ref = xp.grad
    from pase_amd import engine as E
    import unittest.mock as mock
    with mock.patch.object(K, "conv_gemm", wraps=K.conv_gemm) as cg:
        orig = K._conv_desc

        def desc(*a, **kw):
            kw["splitk"] = splitk
            return orig(*a, **kw)
        with mock.patch.object(K, "_conv_desc", desc):
            dx = E.conv_dgrad(dy.to(dev), w.to(dev), R=Cout, O=Cin, k=k, stride=stride, Tin=T, padL=padL, padR=padR,
                              s_red=Cin * k, s_out=k, s_k=1)
        assert cg.called
    assert K.LAST_PLAN_KIND == 2
    assert _rel(dx, ref) < 1e-6


def test_qrnn_linear_tap_major(dev):
    """torchqrnn's Linear over [x_t ; x_{t-1}] (tap-major columns, reversed taps, zero x_{-1})."""
    torch.manual_seed(3)
    S, C_, H, T = 3, 80, 24, 50
    x = torch.randn(S, C_, T)
    lin = torch.randn(3 * H, 2 * C_) * 0.2
    b = torch.randn(3 * H)
    xm1 = torch.cat([torch.zeros(S, C_, 1), x[:, :, :-1]], 2)
    src = torch.cat([x, xm1], 1).double()                      # (S, 2C, T)
    ref = torch.einsum("mk,skt->smt", lin.double(), src) + b.double()[None, :, None]
    y = torch.zeros(S, 3 * H, T, device=dev)
    K.conv_gemm(x.to(dev), lin.to(dev), y, S=S, Cin=C_, Tin=T, M=3 * H, K=2 * C_, taps=2, Ncols=T, Tout=T,
                bias=b.to(dev), tap_major=1, tapstep=-1, padL=0, pad_mode=K.PAD_ZERO)
    assert K.LAST_PLAN_KIND == 2
    assert _rel(y, ref) < 1e-6


@pytest.mark.parametrize("Cin,Cout,T,S,splitk", [
    (800, 200, 100, 3, 1),       # three k-groups per stage, tiles across sequences
    (848, 130, 77, 2, 1),        # 53 k-groups: the last stage holds two real groups and a zero one
    (1500, 100, 64, 2, 0),       # long reduction: auto split-K (data-gradient of a wide head)
    (960, 72, 200, 1, 1),        # 60 k-groups, one ragged row tile
])
def test_flat_1x1(dev, _x6_on, Cin, Cout, T, S, splitk):
    torch.manual_seed(4)
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin) * 0.2
    b = torch.randn(Cout)
    al = torch.rand(Cin) * 0.5
    xin = torch.where(x > 0, x, x * al[None, :, None]).double()
    ref = torch.einsum("mk,skt->smt", w.double(), xin) + b.double()[None, :, None]
    y = torch.full((S, Cout, T), 3.0, device=dev)
    K.conv_gemm(x.to(dev), w.to(dev), y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin, taps=1, Ncols=T, Tout=T, bias=b.to(dev),
                in_alpha=al.to(dev), splitk=splitk)
    assert K.LAST_PLAN_KIND == 2
    assert K.LAST_XP == (_x6_on == "presplit")
    assert _rel(y, ref) < 1e-6


def test_presplit_activation_is_the_default_of_wide_1x1_and_2tap_launches(dev, monkeypatch):
    """>= 1024 rows and one or two taps: the library asks for the pre-split activation by itself (pase_conv_gemm_xp_bytes > 0),
    also for K < 768 (the stacked first layers of the MLP heads: M = 2304, K = 256), with an on-load affine + PReLU, a channel
    slice of a wider input and reversed taps (the QRNN Linear's shape)."""
    monkeypatch.delenv("PASE_X6C_XP")
    torch.manual_seed(14)
    S, Cin, Cout, T = 2, 136, 1030, 150
    xw = torch.randn(S, Cin + 9, T)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
    xin = _xf(xw[:, 3:3 + Cin], sc, sh, al)
    w = torch.randn(Cout, Cin) * 0.2
    ref = torch.einsum("mk,skt->smt", w.double(), xin)
    y = torch.zeros(S, Cout, T, device=dev)
    K.conv_gemm(xw.to(dev), w.to(dev), y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin, taps=1, Ncols=T, Tout=T, x_ctot=Cin + 9, x_coff=3,
                in_scale=sc.to(dev), in_shift=sh.to(dev), in_alpha=al.to(dev))
    assert K.LAST_PLAN_KIND == 2 and K.LAST_XP
    assert _rel(y, ref) < 1e-6
    # two reversed taps, zero padding on the left (causal): y[t] = w0 x[t] + w1 x[t - 1]
    w2 = torch.randn(Cout, 2 * Cin) * 0.2           # tap-major columns [x_t ; x_{t-1}]
    xz = F.pad(xin, (1, 0))
    ref2 = torch.einsum("mk,skt->smt", w2[:, :Cin].double(), xz[:, :, 1:]) + torch.einsum("mk,skt->smt", w2[:, Cin:].double(), xz[:, :, :-1])
    y2 = torch.zeros(S, Cout, T, device=dev)
    K.conv_gemm(xw.to(dev), w2.to(dev), y2, S=S, Cin=Cin, Tin=T, M=Cout, K=2 * Cin, taps=2, tap_major=1, tapstep=-1, padL=0,
                pad_mode=K.PAD_ZERO, Ncols=T, Tout=T, x_ctot=Cin + 9, x_coff=3, in_scale=sc.to(dev), in_shift=sh.to(dev),
                in_alpha=al.to(dev))
    assert K.LAST_PLAN_KIND == 2 and K.LAST_XP
    assert _rel(y2, ref2) < 1e-6


@pytest.mark.parametrize("maxwg", [0, 2])
def test_symmetric_form_of_presplit_launches(dev, monkeypatch, maxwg):
    """Round 6: launches on a pre-split activation with >= 256 rows and no BatchNorm partial sums run the symmetric form of the
    kernel (conv_x6c_kernel<..., SYM>: 256 x 128 tile, all eight waves multiply and share out the LDS DMA -- PASE_X6C_SYM=8 -- or its
    four-wave variant DUO, two 128 x 128 workgroups per CU -- PASE_X6C_SYM=duo; PaseConvGemm::x6_ctl bit 7 / PASE_X6C_SYM=0 keeps
    the staging-wave form).  A 1x1 layer with bias and a ragged second 256-row tile whose k-groups
    do not fill the last stage, and a ConvTranspose1d (k 30, stride 10: three taps', pixel-shuffle store with (channel, phase)
    rows) -- each against fp64 and against the staging-wave form."""
    from pase_amd import engine as E
    from pase_amd.engine import Act
    monkeypatch.setenv("PASE_X6C_XP", "1")
    monkeypatch.setenv("PASE_X6C_FORCE", "1")       # (the routing takes the form only where its 256-row tiles fill whole rounds)
    if maxwg:
        monkeypatch.setenv("PASE_X6C_MAXWG", str(maxwg))
    torch.manual_seed(21)
    # ---- 1x1, bias, on-load PReLU
    S, Cin, Cout, T = 5, 272, 300, 90
    x = torch.randn(S, Cin, T)
    al = torch.rand(Cin) * 0.5
    w = torch.randn(Cout, Cin) * 0.2
    b = torch.randn(Cout)
    xin = torch.where(x > 0, x, x * al[None, :, None]).double()
    ref = torch.einsum("mk,skt->smt", w.double(), xin) + b.double()[None, :, None]
    got = {}
    for sym in ("8", "duo", "0"):
        monkeypatch.setenv("PASE_X6C_SYM", sym)
        y = torch.zeros(S, Cout, T, device=dev)
        K.conv_gemm(x.to(dev), w.to(dev), y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin, taps=1, Ncols=T, Tout=T, bias=b.to(dev),
                    in_alpha=al.to(dev))
        assert K.LAST_PLAN_KIND == 2 and K.LAST_XP
        assert K.LAST_KERNEL == "conv_x6c_kernel<128, 3, false, true, false, %s>" % (
            {"8": "true, false", "0": "false, false", "duo": "true, true"}[sym])
        got[sym] = y.cpu()
        assert _rel(y, ref) < 1e-6, sym
    assert _rel(got["8"], got["0"].double()) < 5e-7 and _rel(got["duo"], got["0"].double()) < 5e-7
    # ---- ConvTranspose1d(48 -> 40, k 30, stride 10): rows = (channel, phase) = 400, three taps'
    S, Cin, Cout, k, st, T = 3, 48, 40, 30, 10, 70
    x = torch.randn(S, Cin, T)
    al = torch.rand(Cin) * 0.5
    w = torch.randn(Cin, Cout, k) * 0.1
    b = torch.randn(Cout)
    xin = torch.where(x > 0, x, x * al[None, :, None]).double()
    ref = F.conv_transpose1d(xin, w.double(), b.double(), stride=st, padding=(k - st) // 2)
    for sym in ("8", "duo", "0"):
        monkeypatch.setenv("PASE_X6C_SYM", sym)
        y = E.deconv_fwd(Act(x.to(dev), C=Cin, alpha=al.to(dev)), w.to(dev), b.to(dev), Cout=Cout, k=k, stride=st)
        assert K.LAST_PLAN_KIND == 2 and K.LAST_XP
        assert K.LAST_KERNEL == "conv_x6c_kernel<192, 2, false, true, false, %s>" % (
            {"8": "true, false", "0": "false, false", "duo": "true, true"}[sym])
        assert tuple(y.shape) == tuple(ref.shape)
        got[sym] = y.cpu()
        assert _rel(y, ref) < 1e-6, sym
    assert _rel(got["8"], got["0"].double()) < 5e-7 and _rel(got["duo"], got["0"].double()) < 5e-7


def test_infinite_activation_stays_non_finite_where_the_reference_is(dev):
    """+-Inf in the activation: exactly the outputs it reaches are non-finite (+-Inf in the reference and on the fp32 pipe;
    NaN or +-Inf here: hi * Inf has the right sign, but the weight's mid / lo pieces times Inf have arbitrary signs and the sum
    of the six terms is NaN about half of the time -- hip_compat.h), every other output is as accurate as ever."""
    torch.manual_seed(15)
    S, Cin, Cout, k, T = 1, 48, 70, 3, 200
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin, k) * 0.2
    x[0, 5, 100] = float("inf")
    x[0, 9, 30] = float("-inf")
    ref = F.conv1d(F.pad(x.double(), (1, 1)), w.double())
    y = torch.zeros(S, Cout, T, device=dev)
    K.conv_gemm(x.to(dev), w.reshape(Cout, -1).contiguous().to(dev), y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * k, taps=k,
                Ncols=T, Tout=T, padL=1, pad_mode=K.PAD_ZERO)
    assert K.LAST_PLAN_KIND == 2
    yc = y.cpu().double()
    bad_ref = ~torch.isfinite(ref)
    assert int(bad_ref.sum()) == 2 * 3 * Cout
    assert bool((~torch.isfinite(yc) == bad_ref).all())
    assert _rel(torch.where(bad_ref, torch.zeros_like(yc), yc), torch.where(bad_ref, torch.zeros_like(ref), ref)) < 1e-6


def test_channel_slice_in_and_out(dev):
    """x_coff / x_ctot (a slice of a wider input) and y_coff / y_ctot (a slice of a wider output)."""
    torch.manual_seed(5)
    S, Cin, Cout, k, T = 2, 48, 70, 3, 140
    xw = torch.randn(S, Cin + 7, T)
    w = torch.randn(Cout, Cin, k) * 0.2
    ref = F.conv1d(F.pad(xw[:, 5:5 + Cin].double(), (1, 1)), w.double())
    yw = torch.full((S, Cout + 3, T), 9.0, device=dev)
    K.conv_gemm(xw.to(dev), w.reshape(Cout, -1).contiguous().to(dev), yw, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * k, taps=k,
                Ncols=T, Tout=T, x_ctot=Cin + 7, x_coff=5, y_ctot=Cout + 3, y_coff=2, Cout_store=Cout, padL=1,
                pad_mode=K.PAD_ZERO)
    assert K.LAST_PLAN_KIND == 2
    assert _rel(yw[:, 2:2 + Cout], ref) < 1e-6
    assert float(yw[:, :2].min()) == 9.0 and float(yw[:, 2 + Cout:].min()) == 9.0


@pytest.mark.parametrize("outs", ["both", "grad", "pred"])
def test_mse_context_epilogue(dev, outs):
    """ContextualizedLoss(MSELoss, r = 7) fused into the projection (pase/losses.py:6-37): loss sum, prediction and
    d(loss)/d(prediction) against the stacked-target definition.  (Tile (0, 0) has whole rows and 128 valid columns: the
    lean epilogue, with column blocks inside a sequence and blocks that touch sequence edges; the second row tile and the
    second column tile take the general one.)"""
    torch.manual_seed(6)
    B, Cin, D, r, Fr = 3, 768, 21, 7, 60
    M = D * r
    h = torch.randn(B, Cin, Fr)
    w = torch.randn(M, Cin) * 0.2
    b = torch.randn(M)
    lab = torch.randn(B, D, Fr)
    pred = torch.einsum("mk,bkt->bmt", w.double(), h.double()) + b.double()[None, :, None]
    padded = F.pad(lab.double(), (r // 2, r // 2))
    tgt = torch.stack([padded[:, :, t:t + r].reshape(B, -1) for t in range(Fr)], 2)      # (B, D*r, F), channel d*r + j
    ref_loss = ((pred - tgt) ** 2).sum()
    y = torch.zeros(B, M, Fr, device=dev) if outs != "grad" else None
    g = torch.zeros(B, M, Fr, device=dev) if outs != "pred" else None
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    K.conv_gemm(h.to(dev), w.to(dev), y, S=B, Cin=Cin, Tin=Fr, M=M, K=Cin, taps=1, Ncols=Fr, Tout=Fr, bias=b.to(dev),
                epilogue=K.EPI_MSE_CTX, label=lab.to(dev), grad_out=g, loss_acc=acc, grad_scale=0.5, r_ctx=r, label_D=D)
    assert K.LAST_PLAN_KIND == 2
    assert abs(float(acc) - float(ref_loss)) <= 1e-6 * float(ref_loss)
    if y is not None:
        assert _rel(y, pred) < 1e-6
    if g is not None:
        assert _rel(g, 0.5 * (pred - tgt)) < 2e-6


@pytest.mark.parametrize("post", ["pow", "logpow", "mag"])
def test_spectrum_post_ops(dev, post):
    """DFT-basis convolution with a |.|^2 / log |.|^2 / |.| epilogue (on-device LPS / SWIPE' spectra)."""
    torch.manual_seed(7)
    S, Cin, k, T, nb = 2, 48, 4, 90, 20
    x = torch.randn(S, Cin, T)
    w = torch.randn(2 * nb, Cin, k) * 0.3                      # rows (re, im) interleaved
    z = F.conv1d(x.double(), w.double())
    Tout = z.shape[2]
    pw = z[:, 0::2] ** 2 + z[:, 1::2] ** 2
    ref = {"pow": 0.5 * pw, "logpow": 2.0 * torch.log(pw + 1e-9), "mag": 0.5 * pw.sqrt()}[post]
    y = torch.zeros(S, nb, Tout, device=dev)
    K.conv_gemm(x.to(dev), w.reshape(2 * nb, -1).contiguous().to(dev), y, S=S, Cin=Cin, Tin=T, M=2 * nb, K=Cin * k, taps=k,
                Ncols=Tout, Tout=Tout, Cout_store=nb, y_ctot=nb,
                post_op={"pow": K.POST_POW, "logpow": K.POST_LOGPOW, "mag": K.POST_MAG}[post],
                post_scale=2.0 if post == "logpow" else 0.5, post_eps=1e-9)
    assert K.LAST_PLAN_KIND == 2
    torch.testing.assert_close(y.cpu().double(), ref, rtol=2e-5, atol=2e-5)


def test_zero_padding_applies_after_the_transform(dev):
    """padded samples are zeros of the TRANSFORMED activation (the QRNN's x_{-1} = 0, ConvTranspose borders), not
    transform(0) = shift."""
    torch.manual_seed(8)
    S, Cin, Cout, k, T = 1, 32, 33, 5, 40
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin, k) * 0.2
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) + 2.0, torch.rand(Cin) * 0.5
    ref = F.conv1d(F.pad(_xf(x, sc, sh, al), (2, 2)), w.double())
    y = torch.zeros(S, Cout, T, device=dev)
    K.conv_gemm(x.to(dev), w.reshape(Cout, -1).contiguous().to(dev), y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * k, taps=k,
                Ncols=T, Tout=T, in_scale=sc.to(dev), in_shift=sh.to(dev), in_alpha=al.to(dev), padL=2, pad_mode=K.PAD_ZERO)
    assert K.LAST_PLAN_KIND == 2
    assert _rel(y, ref) < 1e-6


def test_unbiased_on_same_signed_sums(dev):
    """The property the two-accumulator form exists for (conv_x6c.hip header): on all-positive operands the SIGNED error
    of the round-2 single-accumulator kernels was -2e-6 * K / 2048 of the result on every output (the matrix core drops
    the small terms' low bits toward -inf).  Here |mean error| must stay below 2e-8 of the result and the launch must
    beat the error class of an fp32 fma chain."""
    torch.manual_seed(9)
    S, Cin, Cout, T = 1, 2048, 96, 256
    x = torch.rand(S, Cin, T) + 0.1
    w = (torch.rand(Cout, Cin) + 0.1) * 0.2
    ref = torch.einsum("mk,skt->smt", w.double(), x.double())
    y = torch.zeros(S, Cout, T, device=dev)
    K.conv_gemm(x.to(dev), w.to(dev), y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin, taps=1, Ncols=T, Tout=T, splitk=1)
    assert K.LAST_PLAN_KIND == 2
    e = (y.cpu().double() - ref) / ref
    assert abs(float(e.mean())) < 2e-8, float(e.mean())
    assert float(e.pow(2).mean().sqrt()) < 4e-7


def test_plan_kind_and_pack_contract(dev):
    """no split-bf16 pack -> fp32 pipe (kind 0); shapes without a 16-channel' k-group report 0 pack bytes."""
    from pase_amd import _lib
    lib = _lib.lib()
    x = torch.zeros(1, 64, 64, device=dev)
    w = torch.zeros(72, 64 * 3, device=dev)
    d = K._conv_desc(x, w, torch.zeros(1, 72, 64, device=dev), wt=K.pack_wt(w, M=72, K=192, Cin=64, taps=3), S=1, Cin=64,
                     Tin=64, M=72, K=192, taps=3, Ncols=64, Tout=64, padL=1)
    assert lib.pase_conv_gemm_plan_kind(C.byref(d)) == 0
    nbytes = lib.pase_conv_gemm_x6_bytes(C.byref(d))
    # [32-row tiles][steps][3 planes][64 lanes] 16-byte chunks: the 4 row tiles of one 128-row tile, 4 k-groups x 3 taps
    # ... followed by the on-load parameters expanded per polyphase channel: 3 arrays x 64 channels
    assert nbytes == 4 * (4 * 3) * 3 * 64 * 16 + 3 * 64 * 4
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    d.wx6 = buf.data_ptr()
    assert lib.pase_conv_gemm_plan_kind(C.byref(d)) == 2
    # one input channel (the Sinc FIR): no k-group of 16 channels' -> not this kernel; its own window-image kernel
    # (sinc_x6.hip, plan kind 3: pack = [2 row tiles][16 tap groups][3 planes][64 lanes] chunks), or the fp32 pipe on request
    d1 = K._conv_desc(torch.zeros(1, 1, 300, device=dev), None, torch.zeros(1, 8, 300, device=dev),
                      wt=torch.zeros(251, 8, device=dev), S=1, Cin=1, Tin=300, M=8, K=251, taps=251, Ncols=300, Tout=300,
                      padL=125)
    assert lib.pase_conv_gemm_x6_bytes(C.byref(d1)) == 2 * 16 * 3 * 64 * 16
    d1.wx6 = buf.data_ptr()
    assert lib.pase_conv_gemm_plan_kind(C.byref(d1)) == 3
    d1.x6_ctl = 8
    assert lib.pase_conv_gemm_x6_bytes(C.byref(d1)) == 0
    # a launch the library wants the pre-split activation for (1x1, >= 1024 rows) has its weight pack laid out for the symmetric
    # kernel forms, which stage by copy only: without xp6 the launch is refused (-12), nothing is written
    x2 = torch.randn(2, 256, 200, device=dev)
    w2 = torch.randn(1024, 256, device=dev)
    y2 = torch.full((2, 1024, 200), 7.0, device=dev)
    d2 = K._conv_desc(x2, w2, y2, S=2, Cin=256, Tin=200, M=1024, K=256, taps=1, Ncols=200, Tout=200)
    d2.x6_ctl = 0          # the library's own routing, whatever this parametrisation forces on the other cases
    buf2 = torch.zeros(lib.pase_conv_gemm_x6_bytes(C.byref(d2)), dtype=torch.uint8, device=dev)
    d2.wx6 = buf2.data_ptr()
    assert lib.pase_conv_gemm_xp_bytes(C.byref(d2)) > 0
    assert lib.pase_pack_x6(C.byref(d2), None) == 0
    assert lib.pase_conv_gemm(C.byref(d2), None) == -12
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert bool((y2 == 7.0).all())


def test_persistent_workgroups_take_several_items(dev, monkeypatch):
    """The grid is persistent (one workgroup per CU walks through its items; the staging waves start on the next item's
    first stage during the epilogue of the current one).  With the workgroup count capped at 3 every workgroup owns many
    items of different kinds: plain stores + BatchNorm partial sums (barrier inside the epilogue), split-K atomics, and
    the fused MSE epilogue."""
    monkeypatch.setenv("PASE_X6C_MAXWG", "3")
    torch.manual_seed(11)
    S, Cin, Cout, k, T = 3, 48, 200, 5, 330
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin, k) * 0.2
    b = torch.randn(Cout)
    ref = F.conv1d(F.pad(x.double(), (2, 2), mode="reflect"), w.double(), b.double())
    y = torch.zeros(S, Cout, T, device=dev)
    stat = K.conv_gemm(x.to(dev), w.reshape(Cout, -1).contiguous().to(dev), y, want_stats=True, S=S, Cin=Cin, Tin=T, M=Cout,
                       K=Cin * k, taps=k, Ncols=T, Tout=T, bias=b.to(dev), padL=2, pad_mode=K.PAD_REFLECT)
    assert K.LAST_PLAN_KIND == 2
    assert _rel(y, ref) < 1e-6
    torch.testing.assert_close(stat.cpu().double().sum(0)[:, 1], (ref ** 2).sum((0, 2)), rtol=1e-5, atol=1e-4)
    # long reduction, forced split-K: 2 row tiles x 2 column tiles x 4 slices = 16 items on 3 workgroups
    Cin2 = 1024
    x2 = torch.randn(2, Cin2, 100)
    w2 = torch.randn(Cout, Cin2) * 0.1
    ref2 = torch.einsum("mk,skt->smt", w2.double(), x2.double())
    y2 = torch.full((2, Cout, 100), 5.0, device=dev)
    K.conv_gemm(x2.to(dev), w2.to(dev), y2, S=2, Cin=Cin2, Tin=100, M=Cout, K=Cin2, taps=1, Ncols=100, Tout=100, splitk=4)
    assert K.LAST_PLAN_KIND == 2
    assert _rel(y2, ref2) < 1e-6
    # fused MSE epilogue, 5 row tiles x 2 column tiles
    B, D, r, Fr = 2, 90, 7, 100
    M = D * r
    h = torch.randn(B, 768, Fr)
    w3 = torch.randn(M, 768) * 0.1
    lab = torch.randn(B, D, Fr)
    pred = torch.einsum("mk,bkt->bmt", w3.double(), h.double())
    padded = F.pad(lab.double(), (r // 2, r // 2))
    tgt = torch.stack([padded[:, :, t:t + r].reshape(B, -1) for t in range(Fr)], 2)
    g = torch.zeros(B, M, Fr, device=dev)
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    K.conv_gemm(h.to(dev), w3.to(dev), None, S=B, Cin=768, Tin=Fr, M=M, K=768, taps=1, Ncols=Fr, Tout=Fr,
                epilogue=K.EPI_MSE_CTX, label=lab.to(dev), grad_out=g, loss_acc=acc, grad_scale=1.0, r_ctx=r, label_D=D)
    assert K.LAST_PLAN_KIND == 2
    assert abs(float(acc) - float(((pred - tgt) ** 2).sum())) <= 1e-6 * float(((pred - tgt) ** 2).sum())
    assert _rel(g, pred - tgt) < 2e-6


@pytest.mark.parametrize("S,Cin,Cout,k,st,T", [(8, 16, 16, 11, 1, 800), (4, 8, 32, 11, 2, 800)])
def test_zero_padded_data_gradient_with_interior_and_edge_lanes(dev, S, Cin, Cout, k, st, T):
    """Sequences long enough that a staging slot mixes in-range lanes with zero-padding lanes (the first / last taps'
    positions of a sequence inside an otherwise interior tile).  This is the shape class on which hidden (inline-asm)
    activation loads returned stale registers on the GPU while every smaller test and the emulator passed."""
    from pase_amd import engine as E
    torch.manual_seed(0)
    w = torch.randn(Cout, Cin, k) * 0.1
    padL, padR = E.reflect_pads(k, st)
    Tg = (T + padL + padR - k) // st + 1
    dy = torch.randn(S, Cout, Tg)
    xp = torch.zeros(S, Cin, T + padL + padR, dtype=torch.float64, requires_grad=True)
    F.conv1d(xp, w.double(), None, stride=st).backward(dy.double())
    dx = E.conv_dgrad(dy.to(dev), w.to(dev), R=Cout, O=Cin, k=k, stride=st, Tin=T, padL=padL, padR=padR, s_red=Cin * k,
                      s_out=k, s_k=1)
    assert K.LAST_PLAN_KIND == 2
    assert _rel(dx, xp.grad) < 1e-6


# ---- the persistent grid under stress: few workgroups walking many items (shapes kept from round 5's streamed-form tests) ----
def _stream_case(dev, case):
    """(reference fp64 output, callable that launches and returns (y, stat or None, extra))"""
    torch.manual_seed(31)
    if case in ("three-stages", "stride2", "one-by-one-ragged-cols", "slice-of-wider-output"):
        Cin, Cout, k, stride, T, S = {"three-stages": (48, 130, 11, 1, 250, 3), "stride2": (24, 200, 11, 2, 420, 3),
                                      "one-by-one-ragged-cols": (144, 200, 1, 1, 90, 5),
                                      "slice-of-wider-output": (80, 70, 3, 1, 131, 4)}[case]
        x = torch.randn(S, Cin, T)
        w = torch.randn(Cout, Cin, k) * 0.2
        b = torch.randn(Cout)
        sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
        P = (0, 0) if k == 1 else ((k // 2 - 1, k // 2) if (stride > 1 or k % 2 == 0) else (k // 2, k // 2))
        xin = _xf(x, sc, sh, al)
        ref = F.conv1d(F.pad(xin, P, mode="reflect") if k > 1 else xin, w.double(), b.double(), stride=stride)
        Tout = ref.shape[2]
        wide = case == "slice-of-wider-output"

        def run():
            y = torch.full((S, Cout + (5 if wide else 0), Tout), 7.0, device=dev)
            stat = K.conv_gemm(x.to(dev), w.reshape(Cout, -1).contiguous().to(dev), y, want_stats=not wide, S=S, Cin=Cin, Tin=T,
                               M=Cout, K=Cin * k, taps=k, Ncols=Tout, Tout=Tout, bias=b.to(dev), stride=stride, padL=P[0],
                               pad_mode=K.PAD_REFLECT if k > 1 else K.PAD_ZERO, in_scale=sc.to(dev), in_shift=sh.to(dev),
                               in_alpha=al.to(dev), y_ctot=Cout + (5 if wide else 0), y_coff=3 if wide else 0, Cout_store=Cout)
            if wide:
                assert float((y[:, :3] - 7.0).abs().max()) == 0.0 and float((y[:, 3 + Cout:] - 7.0).abs().max()) == 0.0
                return y[:, 3:3 + Cout], None, None
            return y, stat, None
        return ref, run
    # fused r-context MSE: ragged last row tile (D * r = 623 rows), 90 columns per sequence (column quads straddle
    # sequences), prediction AND gradient stored
    B, D, r, Fr, Ck = 3, 89, 7, 90, 768
    M = D * r
    h = torch.randn(B, Ck, Fr)
    w3 = torch.randn(M, Ck) * 0.1
    b3 = torch.randn(M)
    lab = torch.randn(B, D, Fr)
    pred = torch.einsum("mk,bkt->bmt", w3.double(), h.double()) + b3.double()[None, :, None]
    padded = F.pad(lab.double(), (r // 2, r // 2))
    tgt = torch.stack([padded[:, :, t:t + r].reshape(B, -1) for t in range(Fr)], 2)

    def run():
        g = torch.zeros(B, M, Fr, device=dev)
        yp = torch.zeros(B, M, Fr, device=dev)
        acc = torch.zeros(1, dtype=torch.float64, device=dev)
        K.conv_gemm(h.to(dev), w3.to(dev), yp, S=B, Cin=Ck, Tin=Fr, M=M, K=Ck, taps=1, Ncols=Fr, Tout=Fr, bias=b3.to(dev),
                    epilogue=K.EPI_MSE_CTX, label=lab.to(dev), grad_out=g, loss_acc=acc, grad_scale=0.5, r_ctx=r, label_D=D)
        return yp, None, (g, acc)
    return (pred, tgt), run


@pytest.mark.parametrize("maxwg", [1, 2, 3, 0])
@pytest.mark.parametrize("case", ["three-stages", "stride2", "one-by-one-ragged-cols", "slice-of-wider-output", "mse-ragged"])
def test_persistent_grid_with_few_workgroups_matches_fp64(dev, monkeypatch, case, maxwg):
    """1, 2, 3 workgroups (every workgroup walks several items: the staging waves' prologue of the next item runs beside the
    compute waves' epilogue of the current one, uneven item counts) and uncapped (one item per workgroup).  Shapes: exactly two
    stages per item, a strided layer, a 1x1 layer whose column quads straddle sequences (90 columns per sequence), a channel
    slice of a wider output, and the fused MSE epilogue with a ragged row tile -- against fp64.  (These were round 5's tests of
    the streamed form of the kernel -- built, correct, measured slower on every launch class of the PASE+ step, DESIGN.md
    section 3.0f -- which round 6 removed from the tree; the shapes stay as tests of the form that ships.)"""
    if dev.type == "cpu" and maxwg in (1, 3):
        pytest.skip("emulator: two of the four workgroup counts (the CPU suite's time budget); all four run on the GPU")
    if maxwg:
        monkeypatch.setenv("PASE_X6C_MAXWG", str(maxwg))
    monkeypatch.setenv("PASE_X6C_FORCE", "1")       # (the 96-channel 1x1 case is routed to the fp32 pipe otherwise)
    ref, run = _stream_case(dev, case)
    y, st, extra = run()
    assert K.LAST_PLAN_KIND == 2 and K.LAST_KERNEL.startswith("conv_x6c_kernel<"), K.LAST_KERNEL
    # (the instantiation the report names is the one the shape implies: 1x1 -> <128, 3, ...>, taps -> <192, 2, ...>; ZP = pre-split)
    # ... and SYM = the symmetric 256 x 128 form: pre-split launches of >= 256 rows without BatchNorm partial sums (the MSE case)
    # (1x1 launches take its four-wave variant, DUO: two workgroups per CU; 128 rows are enough)
    sym = K.LAST_XP and case == "mse-ragged"          # (the store cases write BatchNorm partial sums or have 70 rows)
    assert K.LAST_KERNEL == "conv_x6c_kernel<%s, false, %s, false, %s>" % (
        "128, 3" if case in ("one-by-one-ragged-cols", "mse-ragged") else "192, 2", "true" if K.LAST_XP else "false",
        "true, true" if sym else "false, false"), (case, K.LAST_KERNEL)
    if case == "mse-ragged":
        pred, tgt = ref
        g, acc = extra
        want = float(((pred - tgt) ** 2).sum())
        assert abs(float(acc) - want) <= 1e-6 * want, (float(acc), want)
        assert _rel(y, pred) < 1e-6 and _rel(g, 0.5 * (pred - tgt)) < 2e-6
        return
    assert _rel(y, ref) < 1e-6, _rel(y, ref)
    if st is not None:
        st = st.cpu().double().sum(0)
        torch.testing.assert_close(st[:, 0], ref.sum((0, 2)), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(st[:, 1], (ref ** 2).sum((0, 2)), rtol=1e-5, atol=1e-4)

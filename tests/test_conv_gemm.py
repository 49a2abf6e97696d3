"""pase_conv_gemm vs torch CPU fp32 reference (F.conv1d / F.conv_transpose1d), on the emulator
(CPU, `not gpu`) and on the real gfx950 library (`gpu`)."""
import pytest
import torch
import torch.nn.functional as F

from pase_amd import kernels as K


def _tol(dev):
    return dict(rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("Cin,Cout,k,stride,T,S", [
    (3, 5, 11, 1, 70, 2),       # odd k stride 1: pad (5,5)
    (4, 70, 11, 2, 64, 3),      # strided: pad (4,5); Cout > 64 -> 128x128 tile
    (2, 6, 20, 10, 200, 2),     # block-1 shape: k=20 s=10 pad (9,10)
    (1, 8, 251, 1, 300, 2),     # sinc shape: Cin=1, K=251 (not a multiple of 16)
    (20, 130, 1, 1, 37, 3),     # 1x1, M spans two row tiles, N ragged
    (8, 70, 11, 1, 200, 3),     # column tiles straddle sequences (two spans per tile), float4 weight loads
    (8, 70, 11, 2, 410, 3),     # same, strided, T_out = 205
    (4, 8, 20, 10, 3000, 2),    # narrow 64x256 tile straddling sequences, stride 10
    (14, 70, 6, 1, 300, 2),     # 128-row tile, 6 taps: 6-channel stage (36 rows) of the 3-workgroup 6-slot instantiation
    (10, 130, 8, 1, 200, 3),    # 8 taps: 4-channel stage (32 rows) of the 3-workgroup 3-slot instantiation
    (3, 70, 30, 10, 2900, 2),   # 30 taps stride 10, 128-row tile: single-channel stage, 6 slots, 3 workgroups / CU
    (6, 70, 11, 2, 700, 2),     # 11 taps stride 2: 2-channel stage (22 rows), 3 slots
])
def test_conv_fwd_reflect(dev, Cin, Cout, k, stride, T, S):
    torch.manual_seed(0)
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin, k) * 0.2
    b = torch.randn(Cout)
    sc = torch.rand(Cin) + 0.5
    sh = torch.randn(Cin) * 0.1
    al = torch.rand(Cin) * 0.5
    if k > 1:
        P = (k // 2 - 1, k // 2) if (stride > 1 or k % 2 == 0) else (k // 2, k // 2)
    else:
        P = (0, 0)
    xin = x * sc[None, :, None] + sh[None, :, None]
    xin = torch.where(xin > 0, xin, xin * al[None, :, None])
    xp = F.pad(xin, P, mode="reflect") if k > 1 else xin
    ref = F.conv1d(xp, w, b, stride=stride)
    Tout = ref.shape[2]
    y = torch.zeros(S, Cout, Tout, device=dev)
    stat = K.conv_gemm(x.to(dev), w.reshape(Cout, -1).contiguous().to(dev), y, want_stats=True, S=S, Cin=Cin, Tin=T,
                       M=Cout, K=Cin * k, taps=k, Ncols=Tout, Tout=Tout, bias=b.to(dev), in_scale=sc.to(dev),
                       in_shift=sh.to(dev), in_alpha=al.to(dev), stride=stride, padL=P[0], pad_mode=K.PAD_REFLECT)
    torch.testing.assert_close(y.cpu(), ref, **_tol(dev))
    st = stat.cpu().double().sum(0)
    torch.testing.assert_close(st[:, 0], ref.double().sum((0, 2)), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(st[:, 1], (ref.double() ** 2).sum((0, 2)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("Cin,Cout,k,stride,T,S", [
    (6, 5, 30, 4, 10, 2),
    (5, 70, 30, 10, 7, 2),
    (4, 3, 11, 1, 20, 1),
    (32, 16, 30, 10, 140, 2),   # split-bf16 launch, ps = 10: (channel, phase)-ordered rows, 16-byte runs (quads straddle channels)
    (16, 20, 8, 4, 150, 3),     # ps = 4: every quad is one channel's four phases
    (16, 40, 4, 2, 200, 2),     # ps = 2: every quad is two channels x two phases
])
def test_conv_transpose_as_pixel_shuffle(dev, Cin, Cout, k, stride, T, S):
    """nn.ConvTranspose1d(k, stride, padding=(k-stride)//2) == stride-1 conv with stride*Cout rows
    + pixel-shuffle store (modules.py:558-589 GDeconv1DBlock)."""
    torch.manual_seed(1)
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cin, Cout, k) * 0.2       # ConvTranspose1d weight layout (in, out, k)
    b = torch.randn(Cout)
    pad = max(0, (stride - k) // -2)
    ref = F.conv_transpose1d(x, w, b, stride=stride, padding=pad)
    Tout = ref.shape[2]
    R = -(-k // stride)                         # taps per phase
    # pack: Wp[(p, co), (ci, r)] = w[ci, co, p + stride*r]
    wp = torch.zeros(stride, Cout, Cin, R)
    for p in range(stride):
        for r in range(R):
            kk = p + stride * r
            if kk < k:
                wp[p, :, :, r] = w[:, :, kk].t()
    wp = wp.reshape(stride * Cout, Cin * R).contiguous()
    # output u = stride*q + p - pad ; input t = q - r  ->  q ranges over [0, T + R - 1)
    Ncols = T + R - 1
    y = torch.full((S, Cout, Tout), 7.0, device=dev)
    K.conv_gemm(x.to(dev), wp.to(dev), y, S=S, Cin=Cin, Tin=T, M=stride * Cout, K=Cin * R, taps=R,
                Ncols=Ncols, Tout=Tout, bias=b.to(dev), stride=1, tapstep=-1, padL=0,
                pad_mode=K.PAD_ZERO, Cout_store=Cout, ps=stride, poff=-pad)
    torch.testing.assert_close(y.cpu(), ref, **_tol(dev))


def test_qrnn_linear_tap_major(dev):
    """torchqrnn Linear over cat([x_t, x_{t-1}], channel) == k=2 causal conv, tap-major K order."""
    torch.manual_seed(2)
    S, Cin, H, T = 2, 6, 9, 33
    x = torch.randn(S, Cin, T)
    W = torch.randn(3 * H, 2 * Cin) * 0.3
    b = torch.randn(3 * H)
    xm1 = torch.cat([torch.zeros(S, Cin, 1), x[:, :, :-1]], 2)
    src = torch.cat([x, xm1], 1)                 # (S, 2Cin, T)
    ref = torch.einsum("ok,skt->sot", W, src) + b[None, :, None]
    y = torch.zeros(S, 3 * H, T, device=dev)
    K.conv_gemm(x.to(dev), W.to(dev), y, S=S, Cin=Cin, Tin=T, M=3 * H, K=2 * Cin, taps=2, Ncols=T,
                Tout=T, bias=b.to(dev), tap_major=1, tapstep=-1, padL=0, pad_mode=K.PAD_ZERO)
    torch.testing.assert_close(y.cpu(), ref, **_tol(dev))


def test_mse_context_epilogue(dev):
    """Fused MLP-head projection + ContextualizedLoss(MSELoss, r=7) (losses.py:6-37)."""
    torch.manual_seed(3)
    B, Cin, D, r, Fr = 3, 10, 5, 7, 21
    h = torch.randn(B, Cin, Fr)
    W = torch.randn(D * r, Cin) * 0.3
    b = torch.randn(D * r)
    label = torch.randn(B, D, Fr)
    pred = torch.einsum("ok,bkt->bot", W, h) + b[None, :, None]
    pad_ = F.pad(label, (r // 2, r // 2))
    tg = torch.cat([pad_[:, :, t:t + r].contiguous().view(B, -1).unsqueeze(2) for t in range(Fr)], 2)
    ref_loss = F.mse_loss(pred, tg)
    y = torch.zeros(B, D * r, Fr, device=dev)
    g = torch.zeros(B, D * r, Fr, device=dev)
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    n = pred.numel()
    K.conv_gemm(h.to(dev), W.to(dev), y, S=B, Cin=Cin, Tin=Fr, M=D * r, K=Cin, taps=1, Ncols=Fr, Tout=Fr,
                bias=b.to(dev), epilogue=K.EPI_MSE_CTX, label=label.to(dev), grad_out=g, loss_acc=acc,
                grad_scale=2.0 / n, r_ctx=r, label_D=D)
    torch.testing.assert_close(y.cpu(), pred, **_tol(dev))
    torch.testing.assert_close((acc.cpu() / n).float()[0], ref_loss, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(g.cpu(), 2.0 * (pred - tg) / n, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("splitk", [1, 3])
def test_long_reduction_splitk_and_ragged_flat(dev, splitk):
    """1x1 data-gradient shape of the wide heads: few output tiles, long K (split-K with atomics);
    columns flattened across sequences with a ragged tail (T=37 x S=5)."""
    torch.manual_seed(4)
    S, Cin, Cout, T = 5, 300, 70, 37
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin) * 0.1
    ref = torch.einsum("ok,skt->sot", w, x)
    y = torch.full((S, Cout, T), 3.0, device=dev)
    K.conv_gemm(x.to(dev), w.to(dev), y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin, taps=1, Ncols=T, Tout=T, splitk=splitk)
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=1e-4)


def test_channel_slices_and_large_taps(dev):
    """reads a channel slice of a wider buffer, writes a channel slice of a wider buffer, taps > 48
    (tap sub-ranges), zero padding."""
    torch.manual_seed(5)
    S, Cin, Cout, T, k = 2, 3, 9, 150, 61
    xw = torch.randn(S, Cin + 4, T)
    w = torch.randn(Cout, Cin, k) * 0.2
    ref = F.conv1d(xw[:, 2:2 + Cin], w, padding=k // 2)
    yw = torch.zeros(S, Cout + 5, T, device=dev)
    K.conv_gemm(xw.to(dev), w.reshape(Cout, -1).contiguous().to(dev), yw, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * k,
                taps=k, Ncols=T, Tout=T, x_ctot=Cin + 4, x_coff=2, y_ctot=Cout + 5, y_coff=1, padL=k // 2,
                pad_mode=K.PAD_ZERO)
    torch.testing.assert_close(yw.cpu()[:, 1:1 + Cout], ref, rtol=2e-5, atol=2e-5)
    assert float(yw.cpu()[:, 0].abs().max()) == 0.0


def test_reversed_taps_single_row_odd_count(dev):
    """tapstep = -1 with more taps than a stage holds per row pair (taps 27 > 24 -> one channel row per stage,
    taps paired, odd count -> the zero-weight tap reads the guard sample in front of the span)."""
    torch.manual_seed(6)
    S, Cin, Cout, T, k = 2, 3, 7, 90, 27
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin, k) * 0.2
    # out[q] = sum_kk w[kk] x[q - kk]  (causal, reversed taps, zero history)
    ref = F.conv1d(F.pad(x, (k - 1, 0)), w.flip(2))
    y = torch.zeros(S, Cout, T, device=dev)
    K.conv_gemm(x.to(dev), w.reshape(Cout, -1).contiguous().to(dev), y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * k,
                taps=k, Ncols=T, Tout=T, tapstep=-1, padL=0, pad_mode=K.PAD_ZERO)
    torch.testing.assert_close(y.cpu(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("Cin,T", [(50, 40), (21, 64), (1, 8), (70, 37)])
def test_flat_1x1_paths(dev, Cin, T):
    """1x1 layers: float4 flat path (T % 4 == 0: ragged last channel group shifted back, odd Cin, on-load affine +
    PReLU per row) and the one-tap convolution fallback (T = 37)."""
    torch.manual_seed(7)
    S, Cout = 3, 150
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin) * 0.2
    b = torch.randn(Cout)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.2, torch.rand(Cin) * 0.5
    xin = x * sc[None, :, None] + sh[None, :, None]
    xin = torch.where(xin > 0, xin, xin * al[None, :, None])
    ref = torch.einsum("ok,skt->sot", w, xin) + b[None, :, None]
    y = torch.zeros(S, Cout, T, device=dev)
    K.conv_gemm(x.to(dev), w.to(dev), y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin, taps=1, Ncols=T, Tout=T, bias=b.to(dev),
                in_scale=sc.to(dev), in_shift=sh.to(dev), in_alpha=al.to(dev), splitk=1)
    torch.testing.assert_close(y.cpu(), ref, rtol=3e-5, atol=3e-5)


def test_auto_splitk_with_pixel_shuffle_and_direct_kmajor_weight(dev):
    """library-chosen split-K (splitk=0) on a transposed conv (atomic partial tiles + bias from split 0), and a
    weight that already is K-major passed as wt= without a pack."""
    torch.manual_seed(8)
    S, Cin, Cout, k, st, T = 2, 200, 6, 8, 4, 9
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cin, Cout, k) * 0.1
    b = torch.randn(Cout)
    pad = (k - st) // 2
    ref = F.conv_transpose1d(x, w, b, stride=st, padding=pad)
    from pase_amd import engine as E
    y = E.deconv_fwd(E.Act(x.to(dev), C=Cin), w.to(dev), b.to(dev), Cout=Cout, k=k, stride=st)
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=1e-4)
    # A = W^T of a 1x1 data-gradient: W (R=12, O=8) row-major is its own K-major pack
    W = torch.randn(12, 8)
    g = torch.randn(2, 12, 20)
    wt = K.pack_dgrad_t(W.to(dev), R=12, O=8, k=1, st=1, s_red=8, s_out=1, s_k=1)
    assert wt.data_ptr() == wt.data_ptr() and tuple(wt.shape) == (12, 8)
    dx = torch.zeros(2, 8, 20, device=dev)
    K.conv_gemm(g.to(dev), None, dx, wt=wt, S=2, Cin=12, Tin=20, M=8, K=12, taps=1, Ncols=20, Tout=20, splitk=1)
    torch.testing.assert_close(dx.cpu(), torch.einsum("ro,srt->sot", W, g), rtol=2e-5, atol=2e-5)


# ---- flat 1x1 path at 128-row tiles: M > 64, Ncols % 4 == 0 --------------------------------------------------
def _ws_inputs(S, Cin, Cout, T, seed=0):
    torch.manual_seed(seed)
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin) * 0.2
    b = torch.randn(Cout)
    return x, w, b


@pytest.mark.parametrize("xf", ["none", "alpha", "affine"])
@pytest.mark.parametrize("S,Cin,Cout,T", [(3, 70, 130, 40), (5, 33, 260, 100), (2, 256, 140, 200)])
def test_flat_ws_store_bias_stats(dev, xf, S, Cin, Cout, T):
    """ragged M (two / three row tiles), ragged K (Cin % 32 != 0), column tiles crossing sequences, the three on-load
    transform specialisations, BatchNorm partial sums"""
    x, w, b = _ws_inputs(S, Cin, Cout, T)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
    xin = x
    kw = {}
    if xf == "affine":
        xin = x * sc[None, :, None] + sh[None, :, None]
        kw.update(in_scale=sc.to(dev), in_shift=sh.to(dev))
    if xf in ("alpha", "affine"):
        xin = torch.where(xin > 0, xin, xin * al[None, :, None])
        kw.update(in_alpha=al.to(dev))
    ref = F.conv1d(xin, w[:, :, None], b)
    y = torch.zeros(S, Cout, T, device=dev)
    stat = K.conv_gemm(x.to(dev), w.to(dev), y, want_stats=True, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin, taps=1, Ncols=T,
                       Tout=T, bias=b.to(dev), **kw)
    torch.testing.assert_close(y.cpu(), ref, **_tol(dev))
    st = stat.cpu().double().sum(0)
    torch.testing.assert_close(st[:, 0], ref.double().sum((0, 2)), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(st[:, 1], (ref.double() ** 2).sum((0, 2)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("splitk", [1, 4, 0])
def test_flat_ws_long_reduction_splitk(dev, splitk):
    S, Cin, Cout, T = 2, 1500, 130, 64
    x, w, b = _ws_inputs(S, Cin, Cout, T, seed=1)
    ref = F.conv1d(x, w[:, :, None], b)
    y = torch.full((S, Cout, T), 3.0, device=dev)            # conv_gemm zero-fills it when the plan splits
    K.conv_gemm(x.to(dev), w.to(dev), y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin, taps=1, Ncols=T, Tout=T, bias=b.to(dev),
                splitk=splitk)
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=1e-4)


def test_flat_ws_mse_context_epilogue_wide(dev):
    """the worker-head shape: rows m = d*r + j against the r-context of the label, M over several row tiles"""
    torch.manual_seed(2)
    B, Cin, D, r, Fr = 3, 40, 37, 7, 20
    M = D * r
    h = torch.randn(B, Cin, Fr)
    al = torch.rand(Cin) * 0.5
    w = torch.randn(M, Cin) * 0.2
    b = torch.randn(M)
    lab = torch.randn(B, D, Fr)
    hin = torch.where(h > 0, h, h * al[None, :, None])
    pred = F.conv1d(hin, w[:, :, None], b)
    pl = F.pad(lab, (r // 2, r // 2))
    tgt = torch.stack([pl[:, :, j:j + Fr] for j in range(r)], 2).reshape(B, M, Fr)      # channel d*r + j
    gs = 0.37
    y = torch.zeros(B, M, Fr, device=dev)
    g = torch.zeros(B, M, Fr, device=dev)
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    K.conv_gemm(h.to(dev), w.to(dev), y, S=B, Cin=Cin, Tin=Fr, M=M, K=Cin, taps=1, Ncols=Fr, Tout=Fr, bias=b.to(dev),
                in_alpha=al.to(dev), epilogue=K.EPI_MSE_CTX, label=lab.to(dev), grad_out=g, loss_acc=acc,
                grad_scale=gs, r_ctx=r, label_D=D)
    torch.testing.assert_close(y.cpu(), pred, **_tol(dev))
    torch.testing.assert_close(g.cpu(), (pred - tgt) * gs, rtol=1e-4, atol=1e-5)
    assert abs(float(acc) - float(((pred - tgt).double() ** 2).sum())) < 1e-3 * float(((pred - tgt) ** 2).sum())


@pytest.mark.parametrize("Cin,Cout,k,stride,T,S", [
    (64, 130, 11, 1, 300, 3),     # 11 steps per 16-channel group
    (64, 130, 11, 2, 600, 2),     # stride 2: 128 polyphase channels, 6 taps
    (48, 70, 6, 1, 300, 2),
    (40, 70, 8, 1, 200, 2),       # 40 channels: the third k-group is half zeros
    (64, 70, 3, 1, 300, 2),       # two k-groups per stage
    (20, 70, 30, 10, 2900, 2),    # stride 10: 200 polyphase channels, 3 taps
    (768, 130, 1, 1, 200, 3),     # 1x1
])
def test_split_bf16_is_fp32_grade(dev, Cin, Cout, k, stride, T, S):
    """The split-bf16 contraction (PaseConvGemm::wx6: hi+mid+lo pieces, 6 bf16 MFMAs per product) against an fp64
    reference: its error must be that of the fp32 matrix pipe on the same launch (not bf16-grade)."""
    import ctypes as C
    from pase_amd import _lib
    torch.manual_seed(3)
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin, k) * 0.2
    if k > 1:
        P = (k // 2 - 1, k // 2) if (stride > 1 or k % 2 == 0) else (k // 2, k // 2)
        xp = F.pad(x.double(), P, mode="reflect")
    else:
        P, xp = (0, 0), x.double()
    ref = F.conv1d(xp, w.double(), None, stride=stride)
    Tout = ref.shape[2]
    kw = dict(S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * k, taps=k, Ncols=Tout, Tout=Tout, stride=stride, padL=P[0],
              pad_mode=K.PAD_REFLECT if k > 1 else K.PAD_ZERO, splitk=1)
    w2 = w.reshape(Cout, -1).contiguous().to(dev)
    d = K._conv_desc(x.to(dev), w2, torch.empty(S, Cout, Tout, device=dev), wt=K.pack_wt(w2, M=Cout, K=Cin * k, Cin=Cin, taps=k), **kw)
    assert _lib.lib().pase_conv_gemm_x6_bytes(C.byref(d)) > 0, "shape expected to have a split-bf16 plan"
    err = {}
    saved = K.X6
    try:
        for mode in (True, False):
            K.X6 = mode
            y = torch.zeros(S, Cout, Tout, device=dev)
            K.conv_gemm(x.to(dev), w2, y, **kw)
            err[mode] = ((y.cpu().double() - ref).norm() / ref.norm()).item()
    finally:
        K.X6 = saved
    # the two-accumulator split contraction (conv_x6c.hip) is a BETTER fp32 evaluation than the k-ordered fma chain of
    # the fp32 matrix pipe: error against fp64 no larger, and under 4e-7
    assert err[True] < 4e-7, err
    assert err[True] < 1.05 * err[False] + 1e-8, err


def test_split_bf16_api_contract(dev):
    """pase_conv_gemm_x6_bytes / pase_pack_x6 / PaseConvGemm::wx6: shapes without a split-bf16 plan report 0 bytes and a
    pack pointer on such a launch is refused (-11) instead of being ignored (pack layout: tests/test_conv_x6c.py)."""
    import ctypes as C
    from pase_amd import _lib
    lib = _lib.lib()
    S, Cin, Cout, T = 2, 1, 8, 300                       # one input channel (the Sinc FIR): no 16-channel k-group
    x = torch.randn(S, Cin, T, device=dev)
    w = torch.randn(Cout, Cin * 20, device=dev)
    y = torch.zeros(S, Cout, 281, device=dev)
    kw = dict(S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * 20, taps=20, Ncols=281, Tout=281, stride=1, padL=0, splitk=1)
    d = K._conv_desc(x, w, y, wt=K.pack_wt(w, M=Cout, K=Cin * 20, Cin=Cin, taps=20), **kw)
    assert lib.pase_conv_gemm_x6_bytes(C.byref(d)) == 0
    junk = torch.zeros(4096, dtype=torch.uint8, device=dev)
    d.wx6 = junk.data_ptr()
    assert lib.pase_pack_x6(C.byref(d), None) == -11
    assert lib.pase_conv_gemm(C.byref(d), None) == -11

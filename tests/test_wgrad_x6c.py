"""pase_wgrad_gemm on the split-bf16 kernel (conv_x6c.hip, T-mode: the contraction runs over positions) against fp64.

  dw[m, (ci,kk)] += sum_{s,q} g~[s, m, q] * z~[s, ci, q * stride + kk * tapstep - padL],   dbias[m] += sum g~[s, m, q]

(autograd's conv1d / conv_transpose1d / linear weight gradients of `tot_loss.backward()`, worker_scheduler.py:67).
Every case asserts that the launch ran on that kernel and that the result has the error of a GOOD fp32 evaluation
(relative L2 against fp64 <= 1e-6).  Normal orientation (rows = g) and the swapped one the 1x1 layers with more output
than input channels use (rows = z's channels, transposed accumulation, bias gradient from the staged column sums).
"""
import pytest
import torch
import torch.nn.functional as F

from pase_amd import kernels as K


@pytest.fixture(params=["4", "1", "3"], ids=["presplit-planes", "staged-toeplitz", "plane-rows"])
def wmode(request, monkeypatch):
    """layers with taps: mode 4 (the default) copies the (channel, tap) columns out of pre-split phase-decomposed bf16 planes
    of z~; mode 1 stages them from fp32 with the conversion in the GEMM; mode 3 reads (channel, tap) ROWS at 2-byte
    granularity out of row-major bf16 planes of z and stages g"""
    monkeypatch.setenv("PASE_X6C_WGRAD_MODE", request.param)
    return request.param


@pytest.fixture(autouse=True)
def _x6_on(monkeypatch):
    # the library routes 1x1 weight gradients to the fp32 matrix pipe (faster there on every PASE+ shape); the T-mode
    # kernel supports them in both orientations and is tested on them here
    monkeypatch.setenv("PASE_X6C_WGRAD_FLAT", "1")
    saved = K.X6
    K.X6 = True
    yield
    K.X6 = saved


def _rel(a, ref):
    return float((a.cpu().double() - ref).norm() / ref.norm().clamp_min(1e-300))


def _xf(x, sc, sh, al):
    v = x.double() * sc.double()[None, :, None] + sh.double()[None, :, None]
    return torch.where(v > 0, v, v * al.double()[None, :, None])


@pytest.mark.parametrize("Cin,Cout,k,st,T,S", [
    (12, 130, 11, 1, 90, 3),       # two row tiles (second ragged), 132 + 1 columns: a second column tile for the bias column
    (20, 100, 11, 2, 168, 2),      # stride 2, reflect padding on both sides, Ncols = 84 (not a multiple of 16)
    (6, 96, 20, 10, 400, 2),       # stride 10 (block 1 shape)
    (40, 200, 3, 1, 50, 4),        # short sequences: most k-groups touch the padding
    (24, 130, 11, 1, 96, 3),       # aligned rows of g, whole chunks: the row-coalesced staging path (mode 3), two column tiles
    (16, 140, 11, 2, 160, 3),      # ... with stride 2 (two phase rows per channel in the planes)
])
def test_conv_weight_gradient(dev, wmode, Cin, Cout, k, st, T, S):
    torch.manual_seed(0)
    x = torch.randn(S, Cin, T)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
    P = (k // 2 - 1, k // 2) if (st > 1 or k % 2 == 0) else (k // 2, k // 2)
    w = torch.randn(Cout, Cin, k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(F.pad(_xf(x, sc, sh, al), P, mode="reflect"), w, b, stride=st)
    g = torch.randn(y.shape)
    (y * g.double()).sum().backward()
    dw = torch.zeros(Cout, Cin * k, device=dev)
    db = torch.zeros(Cout, device=dev)
    K.wgrad_gemm(g.to(dev), x.to(dev), dw, S=S, M=Cout, Tg=y.shape[2], Ncols=y.shape[2], Cin=Cin, Tz=T, taps=k, dbias=db,
                 in_scale=sc.to(dev), in_shift=sh.to(dev), in_alpha=al.to(dev), stride=st, padL=P[0], pad_mode=K.PAD_REFLECT)
    assert K.LAST_WGRAD_X6 and K.LAST_WGRAD_KIND == int(wmode)
    assert _rel(dw.view(Cout, Cin, k), w.grad) < 1e-6
    assert _rel(db, b.grad) < 1e-6


@pytest.mark.parametrize("Cin,Cout,k,st,T,S,maxwg", [
    (24, 256, 11, 1, 200, 3, 0),     # one row tile of 256, 264 + 1 columns = three column tiles, 39 k-groups = 7 stages of six (odd)
    (16, 300, 11, 2, 420, 2, 2),     # ragged second row tile (300 of 512), stride 2 (phase-ordered columns), two workgroups
    (40, 520, 3, 1, 96, 4, 1),       # three row tiles, ONE workgroup walking every (slice, tile) item, 24 k-groups = 4 stages (even)
    (12, 260, 30, 4, 400, 2, 3),     # ConvTranspose-like taps / stride (k 30, stride 4): eight tap-phase columns per plane row
])
def test_conv_weight_gradient_symmetric_form(dev, monkeypatch, Cin, Cout, k, st, T, S, maxwg):
    """Round 6: pre-split weight gradients with at least 256 rows of g run x6c_wgrad_sym_kernel (plan kind 7): 256 x 128
    workgroup tile, all eight waves multiply, six k-groups per stage in two separate LDS buffers; against fp64 and against the
    four-compute-wave kernel (x6 bit 11, PASE_X6C_WGRAD_SYM=0)."""
    if maxwg:
        monkeypatch.setenv("PASE_X6C_MAXWG", str(maxwg))
    torch.manual_seed(4)
    x = torch.randn(S, Cin, T)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
    P = (k // 2 - 1, k // 2) if (st > 1 or k % 2 == 0) else (k // 2, k // 2)
    w = torch.randn(Cout, Cin, k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(F.pad(_xf(x, sc, sh, al), P, mode="reflect"), w, b, stride=st)
    g = torch.randn(y.shape)
    (y * g.double()).sum().backward()
    args = dict(S=S, M=Cout, Tg=y.shape[2], Ncols=y.shape[2], Cin=Cin, Tz=T, taps=k, in_scale=sc.to(dev), in_shift=sh.to(dev),
                in_alpha=al.to(dev), stride=st, padL=P[0], pad_mode=K.PAD_REFLECT)
    dw = torch.full((Cout, Cin * k), 0.25, device=dev)           # (the kernel ADDS into dw)
    db = torch.zeros(Cout, device=dev)
    K.wgrad_gemm(g.to(dev), x.to(dev), dw, dbias=db, **args)
    assert K.LAST_WGRAD_X6 and K.LAST_WGRAD_KIND == 7
    assert _rel((dw - 0.25).view(Cout, Cin, k), w.grad) < 1e-6
    assert _rel(db, b.grad) < 1e-6
    dw2 = torch.zeros(Cout, Cin * k, device=dev)                 # no bias gradient: no all-ones column
    K.wgrad_gemm(g.to(dev), x.to(dev), dw2, **args)
    assert K.LAST_WGRAD_KIND == 7 and _rel(dw2.view(Cout, Cin, k), w.grad) < 1e-6
    monkeypatch.setenv("PASE_X6C_WGRAD_SYM", "0")
    dw3 = torch.zeros(Cout, Cin * k, device=dev)
    K.wgrad_gemm(g.to(dev), x.to(dev), dw3, **args)
    assert K.LAST_WGRAD_KIND == 4 and _rel(dw3.view(Cout, Cin, k), w.grad) < 1e-6


def test_conv_transpose_weight_gradient(dev, wmode):
    """nn.ConvTranspose1d weight gradient: G = PReLU(layer input) at the low rate (g_alpha), Z = dY, zero padding."""
    torch.manual_seed(2)
    S, Cin, Cout, k, st, T = 2, 100, 6, 30, 4, 24
    z_in = torch.randn(S, Cin, T)
    al = torch.rand(Cin) * 0.5
    w = torch.randn(Cin, Cout, k, dtype=torch.float64, requires_grad=True)
    a = torch.where(z_in > 0, z_in, z_in * al[None, :, None]).double()
    pad = (k - st) // 2
    y = F.conv_transpose1d(a, w, None, stride=st, padding=pad)
    g = torch.randn(y.shape)
    (y * g.double()).sum().backward()
    dw = torch.zeros(Cin, Cout * k, device=dev)
    K.wgrad_gemm(z_in.to(dev), g.to(dev), dw, S=S, M=Cin, Tg=T, Ncols=T, Cin=Cout, Tz=y.shape[2], taps=k, stride=st,
                 padL=pad, pad_mode=K.PAD_ZERO, g_alpha=al.to(dev))
    assert K.LAST_WGRAD_X6 and K.LAST_WGRAD_KIND == int(wmode)
    assert _rel(dw.view(Cin, Cout, k), w.grad) < 1e-6


def test_reversed_taps(dev, wmode):
    """tapstep = -1 (the QRNN's x_{t-1} tap as its own launch: taps = 1 shifted; here a 3-tap reversed window)."""
    torch.manual_seed(3)
    S, Cin, M, k, T = 2, 24, 100, 3, 60
    z = torch.randn(S, Cin, T)
    g = torch.randn(S, M, T)
    zp = F.pad(z.double(), (k - 1, 0))
    # dw[m, ci*k + kk] = sum g[m, q] * z[ci, q - kk]
    ref = torch.stack([torch.einsum("smq,scq->mc", g.double(), zp[:, :, k - 1 - kk:k - 1 - kk + T]) for kk in range(k)], 2)
    dw = torch.zeros(M, Cin * k, device=dev)
    K.wgrad_gemm(g.to(dev), z.to(dev), dw, S=S, M=M, Tg=T, Ncols=T, Cin=Cin, Tz=T, taps=k, tapstep=-1, padL=0,
                 pad_mode=K.PAD_ZERO)
    assert K.LAST_WGRAD_X6 and K.LAST_WGRAD_KIND == int(wmode)
    assert _rel(dw.view(M, Cin, k), ref) < 1e-6


@pytest.mark.parametrize("S,Cin,Cout,T,gx,zx", [
    (3, 84, 273, 200, 0, 0),        # swapped: 273 output channels > 84 inputs; three column tiles, one ragged
    (3, 70, 200, 64, 4, 1),         # swapped, row-coalesced staging of a channel slice of g
    (2, 130, 150, 36, 5, 7),        # swapped, channel slices of wider tensors on both operands
    (5, 256, 96, 20, 0, 3),         # normal orientation (rows = g): bias through the ones column
    (1, 200, 300, 52, 2, 0),        # one sequence, Ncols not a multiple of 16
])
def test_flat_1x1_weight_gradient(dev, S, Cin, Cout, T, gx, zx):
    torch.manual_seed(7)
    xw = torch.randn(S, Cin + zx + 2, T)
    gw = torch.randn(S, Cout + gx + 3, T)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin), torch.rand(Cin) * 0.5
    ga = torch.rand(Cout) * 0.5
    x = _xf(xw[:, zx:zx + Cin], sc, sh, al)
    g = gw[:, gx:gx + Cout].double()
    g = torch.where(g > 0, g, g * ga.double()[None, :, None])
    ref = torch.einsum("sot,sct->oc", g, x)
    dw = torch.zeros(Cout, Cin, device=dev)
    db = torch.zeros(Cout, device=dev)
    K.wgrad_gemm(gw.to(dev), xw.to(dev), dw, S=S, M=Cout, Tg=T, Ncols=T, Cin=Cin, Tz=T, taps=1, dbias=db,
                 g_ctot=gw.shape[1], g_coff=gx, z_ctot=xw.shape[1], z_coff=zx, in_scale=sc.to(dev), in_shift=sh.to(dev),
                 in_alpha=al.to(dev), g_alpha=ga.to(dev))
    assert K.LAST_WGRAD_X6
    assert _rel(dw, ref) < 1e-6
    assert _rel(db, g.sum((0, 2))) < 1e-6


def test_accumulates_into_dw_and_persistent_items(dev, monkeypatch):
    """dw / dbias are += targets (split-K slices and other launches add into the same gradient buffer); with the
    workgroup count capped every workgroup walks through several (slice, tile) items."""
    monkeypatch.setenv("PASE_X6C_MAXWG", "3")
    torch.manual_seed(9)
    S, Cin, M, T = 4, 140, 150, 100
    z = torch.randn(S, Cin, T)
    g = torch.randn(S, M, T)
    ref = torch.einsum("sot,sct->oc", g.double(), z.double())
    dw = torch.full((M, Cin), 2.0, device=dev)
    db = torch.full((M,), -1.0, device=dev)
    K.wgrad_gemm(g.to(dev), z.to(dev), dw, S=S, M=M, Tg=T, Ncols=T, Cin=Cin, Tz=T, taps=1, dbias=db, splitk=5)
    assert K.LAST_WGRAD_X6
    assert _rel(dw, ref + 2.0) < 1e-6
    assert _rel(db, g.double().sum((0, 2)) - 1.0) < 1e-6


@pytest.mark.parametrize("mode", ["4", "1"])
def test_persistent_items_on_planes_and_on_registers(dev, monkeypatch, mode):
    """Three workgroups walk all (split-K slice, tile) items of an 11-tap stride-2 weight gradient: every workgroup stages the
    NEXT item's first stage (LDS DMA from the pre-split planes in mode 4, hand-waited register loads in mode 1) while its compute
    waves flush the current item's tile."""
    monkeypatch.setenv("PASE_X6C_WGRAD_MODE", mode)
    monkeypatch.setenv("PASE_X6C_MAXWG", "3")
    torch.manual_seed(11)
    S, Cin, Cout, k, st, T = 3, 20, 150, 11, 2, 336
    x = torch.randn(S, Cin, T)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
    P = (k // 2 - 1, k // 2)
    w = torch.randn(Cout, Cin, k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(F.pad(_xf(x, sc, sh, al), P, mode="reflect"), w, b, stride=st)
    g = torch.randn(y.shape)
    (y * g.double()).sum().backward()
    dw = torch.zeros(Cout, Cin * k, device=dev)
    db = torch.zeros(Cout, device=dev)
    K.wgrad_gemm(g.to(dev), x.to(dev), dw, S=S, M=Cout, Tg=y.shape[2], Ncols=y.shape[2], Cin=Cin, Tz=T, taps=k, dbias=db,
                 in_scale=sc.to(dev), in_shift=sh.to(dev), in_alpha=al.to(dev), stride=st, padL=P[0], pad_mode=K.PAD_REFLECT,
                 splitk=4)
    assert K.LAST_WGRAD_X6 and K.LAST_WGRAD_KIND == int(mode)
    assert _rel(dw.view(Cout, Cin, k), w.grad) < 1e-6
    assert _rel(db, b.grad) < 1e-6


def test_small_launches_stay_on_the_fp32_pipe(dev):
    """at most 64 rows on the packed side: no split-bf16 plan (pase_wgrad_x6_bytes == 0), exact-fp32 MFMA kernel."""
    torch.manual_seed(4)
    S, Cin, M, T = 2, 40, 9, 37
    z, g = torch.randn(S, Cin, T), torch.randn(S, M, T)
    dw = torch.zeros(M, Cin, device=dev)
    K.wgrad_gemm(g.to(dev), z.to(dev), dw, S=S, M=M, Tg=T, Ncols=T, Cin=Cin, Tz=T, taps=1)
    assert not K.LAST_WGRAD_X6
    assert _rel(dw, torch.einsum("sot,sct->oc", g.double(), z.double())) < 1e-5

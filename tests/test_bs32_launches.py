"""The encoder launches of the PASE+ bs32 step (BASELINE.json configs[1]: 96 chunks of 32000 samples) AT FULL SIZE, split-bf16
kernel against the exact-fp32 matrix pipe of the same library: forward conv + BatchNorm partial sums (reflect padding),
data gradient (zero padding, several sequences per tile, thousands of persistent items) and weight gradient.

GPU only: the emulator tests (tests/test_conv_x6c.py, tests/test_wgrad_x6c.py) cover the same code at sizes a CPU finishes,
but a hardware-only failure of the zero-padding lanes at exactly these sizes went through them in round 3 -- this file is
the tripwire.  Both pipes evaluate the same fp32 sums; they may differ by fp32 rounding only (relative L2 <= 3e-6).
"""
import pytest
import torch

from pase_amd import engine as E
from pase_amd import kernels as K
from pase_amd.engine import Act

pytestmark = pytest.mark.gpu

S = 96
SHAPES = [("blk1", 64, 64, 20, 10, 32000), ("blk2", 64, 128, 11, 2, 3200), ("blk3", 128, 128, 11, 1, 1600),
          ("blk4", 128, 256, 11, 2, 1600), ("blk5", 256, 256, 11, 1, 800), ("blk6", 256, 512, 11, 2, 800),
          ("blk7", 512, 512, 11, 2, 400)]


@pytest.fixture()
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pase_amd import _lib
    _lib.use_library(None, "cuda")
    _lib.lib()
    saved = K.X6
    yield torch.device("cuda:0")
    K.X6 = saved


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("name,Cin,Cout,k,st,Tin", SHAPES, ids=[s[0] for s in SHAPES])
def test_encoder_block_launches_agree_between_the_pipes(gpu, name, Cin, Cout, k, st, Tin):
    torch.manual_seed(0)
    pL, pR = E.reflect_pads(k, st)
    x = torch.randn(S, Cin, Tin, device=gpu)
    w = torch.randn(Cout, Cin, k, device=gpu) * 0.05
    b = torch.randn(Cout, device=gpu)
    a = Act(x, C=Cin, scale=torch.rand(Cin, device=gpu) + 0.5, shift=torch.randn(Cin, device=gpu) * 0.1,
            alpha=torch.rand(Cin, device=gpu) * 0.3)
    res = {}
    for mode in (True, False):
        K.X6 = mode
        y, stat = E.conv_fwd(a, w.view(Cout, -1), b, Cout=Cout, taps=k, stride=st, padL=pL, padR=pR,
                             pad_mode=K.PAD_REFLECT, want_stats=True)
        kind_f = K.LAST_PLAN_KIND
        dy = torch.randn(y.shape, device=gpu, generator=torch.Generator(device=gpu).manual_seed(1))
        dx = E.conv_dgrad(dy, w, R=Cout, O=Cin, k=k, stride=st, Tin=Tin, padL=pL, padR=pR, s_red=Cin * k, s_out=k, s_k=1)
        kind_d = K.LAST_PLAN_KIND
        dw = torch.zeros(Cout, Cin * k, device=gpu)
        db = torch.zeros(Cout, device=gpu)
        E.conv_wgrad(dy, a, dw, db, taps=k, stride=st, padL=pL, pad_mode=K.PAD_REFLECT)
        res[mode] = (y.clone(), stat.sum(0).clone(), dx.clone(), dw.clone(), db.clone(), kind_f, kind_d)
    assert res[False][5] == 0 and res[False][6] == 0          # the comparison side really is the fp32 pipe
    assert res[True][6] == 2                                  # every encoder data gradient runs on the split-bf16 kernel
    for i, what in enumerate(("forward", "BatchNorm sums", "data gradient", "weight gradient", "bias gradient")):
        assert _rel(res[True][i], res[False][i]) < 3e-6, (name, what)
    # the edges are where zero padding lives: compare them on their own (a relative-L2 over 800 positions hides 10 of them)
    dx6, dx32 = res[True][2], res[False][2]
    Td = dx6.shape[-1]
    for sl in (slice(0, 16), slice(Td - 16, Td)):
        assert _rel(dx6[:, :, sl], dx32[:, :, sl]) < 3e-6, (name, "data-gradient edge", sl)


def test_prelu_negative_slope_path_at_full_layer_size_vs_fp64(gpu):
    """The PReLU negative-slope path with slopes != 1 at the size of the step's largest layer (block 0's output, 96 x 64 x
    32 000), judged against fp64: the kink-free live-reference goldens (tests/test_pase_step.py `smooth`) set every slope to
    1, so this is where a != 1 gets its tight gate.  Three consumers of a = PReLU(BN_train(y)) (modules.py:1073-1077):
      * backward (pase_act_bwd_reduce / _apply): d/dy of  <g1, reflect-pad(a)> + <g2, mean-pool_160(a)>  -- block 1's data
        gradient in padded coordinates (pads 9 / 10) folded back, merged with the pooled dense-skip gradient
        (frontend.py:213-232) -- plus the three per-channel sums (d beta, d gamma, d alpha);
      * forward on-load (conv_x6c staging: affine + PReLU applied while the operand is split): block 1's convolution;
      * forward pooled (pase_bn_act_pool).
    relative L2 <= 1e-6 against the fp64 evaluation of the same expressions (torch autograd in double on the same GPU)."""
    import torch.nn.functional as F
    torch.manual_seed(11)
    C, T, pL, pR, d = 64, 32000, 9, 10, 160
    gen = torch.Generator(device=gpu).manual_seed(5)
    y = torch.randn(S, C, T, device=gpu, generator=gen)
    gamma = torch.rand(C, device=gpu, generator=gen) * 0.6 + 0.7
    beta = torch.randn(C, device=gpu, generator=gen) * 0.3
    al = torch.rand(C, device=gpu, generator=gen) * 0.45 + 0.02          # slopes in (0.02, 0.47): a kink in every channel
    g1 = torch.randn(S, C, T + pL + pR, device=gpu, generator=gen)
    g2 = torch.randn(S, C, T // d, device=gpu, generator=gen)
    # ---- fp64 truth -------------------------------------------------------------------------------------------------
    y64 = y.double().requires_grad_(True)
    ga64, be64, al64 = (t.double().requires_grad_(True) for t in (gamma, beta, al))
    z = F.batch_norm(y64, None, None, ga64, be64, True, 0.1, 1e-5)
    a64 = F.prelu(z, al64)
    loss = (F.pad(a64, (pL, pR), mode="reflect") * g1.double()).sum() + (a64.view(S, C, T // d, d).mean(3) * g2.double()).sum()
    loss.backward()
    want_dy, want_db, want_dg, want_da = y64.grad, be64.grad, ga64.grad, al64.grad
    pooled64 = a64.detach().view(S, C, T // d, d).mean(3)
    w = torch.randn(64, C, 20, device=gpu, generator=gen) * 0.05
    bias = torch.randn(64, device=gpu, generator=gen)
    conv64 = F.conv1d(F.pad(a64.detach(), (pL, pR), mode="reflect"), w.double(), bias.double(), stride=10)
    del z, a64, loss, y64
    # ---- statistics the forward would have produced (fp64 -> fp32, as pase_bn_finalize hands them on) ----------------
    mean = y.double().mean((0, 2))
    var = y.double().var((0, 2), unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    scale = (gamma.double() * rstd).float()
    shift = (beta.double() - mean * gamma.double() * rstd).float()
    mean, rstd = mean.float(), rstd.float()
    # ---- backward ----------------------------------------------------------------------------------------------------
    sums = torch.zeros(C, 3, dtype=torch.float64, device=gpu)
    dy = torch.empty(S, C, T, device=gpu)
    kw = dict(S=S, C_=C, T=T, dsrc=g1, Tp=T + pL + pR, padL=pL, pad_mode=K.PAD_REFLECT, dpool=g2, dpool_ctot=C, dpool_coff=0,
              pool_F=T // d, pool_d=d, scale=scale, shift=shift, alpha=al, mean=mean, rstd=rstd, sums=sums, dy=dy, has_bn=1)
    K.act_bwd_reduce(y, **kw)
    K.act_bwd_apply(y, **kw)
    torch.cuda.synchronize()
    errs = dict(dy=_rel(dy, want_dy), dbeta=_rel(sums[:, 0], want_db), dgamma=_rel(sums[:, 1], want_dg),
                dalpha=_rel(sums[:, 2], want_da))
    del dy, want_dy
    # ---- forward: on-load in the convolution's staging, and in the pooling pass --------------------------------------
    a = Act(y, C=C, scale=scale, shift=shift, alpha=al)
    for mode in (True, False):
        K.X6 = mode
        out, _ = E.conv_fwd(a, w.view(64, -1), bias, Cout=64, taps=20, stride=10, padL=pL, padR=pR, pad_mode=K.PAD_REFLECT)
        errs["conv_onload_%s" % ("x6" if mode else "fp32pipe")] = _rel(out, conv64)
    pooled = torch.empty(S, C, T // d, device=gpu)
    K.bn_act_pool(y, pooled, scale, shift, al, S=S, C_=C, T=T, F=T // d, d=d, o_ctot=C, o_coff=0)
    errs["pool"] = _rel(pooled, pooled64)
    print("PReLU negative-slope path, relative L2 vs fp64:", {k: "%.2e" % v for k, v in errs.items()})
    for k, v in errs.items():
        assert v <= 1e-6, (k, v)

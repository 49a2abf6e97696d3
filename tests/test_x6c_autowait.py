"""Tripwire for the hand-counted staging waits of conv_x6c.hip (x6c_gload / x6c_vmwait_slots / x6c_claim).

The staging waves' activation loads are inline asm whose destination VGPRs the compiler considers defined as soon as the
asm statement is issued; they are waited for by hand-counted `s_waitcnt vmcnt(N)`.  A register copy, re-coalescing or spill
of those VGPRs before the wait would read stale data silently.  Two guards:
  * build time (pase_amd/build.py check_x6c_resources): the register-staged instantiations must report 0 bytes of scratch
    and 0 spilled VGPRs, or the build fails;
  * here, on the GPU: the SAME sources compiled with -DPASE_X6C_AUTOWAIT (plain loads, the compiler's own s_waitcnt
    bookkeeping: slow but correct by construction; tests/libpase_hip_autowait.so, built by __graft_entry__.build()) must
    give BIT-IDENTICAL convolution outputs on launches that exercise every register-staged path (interior / padded slots,
    stride-1 / strided address walks, 1x1 with three k-groups per stage, the 64 x 256 tile), and weight gradients (atomics:
    order-nondeterministic) that agree to fp32 round-off.
"""
import os

import pytest
import torch

from pase_amd import _lib, build
from pase_amd import kernels as K

pytestmark = pytest.mark.gpu


@pytest.fixture()
def libs():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    so = build.AUTOWAIT_SO
    stamp = so + ".sha256"
    assert os.path.exists(so) and os.path.exists(stamp), "tests/libpase_hip_autowait.so is missing: run __graft_entry__.build()"
    assert open(stamp).read().strip() == build.hip_digest(), "tests/libpase_hip_autowait.so is older than pase_amd/csrc"
    saved = K.X6
    K.X6 = True

    def use(autowait):
        _lib.use_library(so if autowait else None, "cuda")
        _lib.lib()
    yield use
    _lib.use_library(None, "cuda")
    K.X6 = saved


CONVS = [  # Cin, Cout, k, stride, T, S
    (128, 256, 11, 2, 1600, 8),      # strided walk, reflect padding at both ends, many items per workgroup
    (256, 256, 11, 1, 800, 8),       # stride-1 interior slots
    (64, 64, 20, 10, 3200, 6),       # the 64 x 256 tile, stride 10
    (840, 256, 1, 1, 200, 32),       # 1x1, three k-groups per stage, ragged last group
    (56, 96, 3, 1, 90, 5),           # ragged channels, three sequences per tile
]


@pytest.mark.parametrize("Cin,Cout,k,stride,T,S", CONVS)
def test_conv_outputs_are_bit_identical_with_compiler_waits(libs, monkeypatch, Cin, Cout, k, stride, T, S):
    monkeypatch.setenv("PASE_X6C_FORCE", "1")          # the split-bf16 kernel on every shape
    monkeypatch.setenv("PASE_X6C_XP", "0")             # registers, not LDS DMA
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(S, Cin, T, device=dev, generator=g)
    w = torch.randn(Cout, Cin * k, device=dev, generator=g) * 0.1
    b = torch.randn(Cout, device=dev, generator=g)
    sc = torch.rand(Cin, device=dev, generator=g) + 0.5
    sh = torch.randn(Cin, device=dev, generator=g) * 0.1
    al = torch.rand(Cin, device=dev, generator=g) * 0.5
    P = (k // 2 - 1, k // 2) if (stride > 1 or k % 2 == 0) else (k // 2, k // 2)
    Tout = (T + P[0] + P[1] - k) // stride + 1
    outs = []
    for autowait in (False, True, False):
        libs(autowait)
        y = torch.full((S, Cout, Tout), float("nan"), device=dev)
        stat = K.conv_gemm(x, w, y, want_stats=True, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * k, taps=k, Ncols=Tout, Tout=Tout,
                           bias=b, stride=stride, padL=P[0], pad_mode=K.PAD_REFLECT, in_scale=sc, in_shift=sh, in_alpha=al)
        assert K.LAST_PLAN_KIND == 2 and not K.LAST_XP
        torch.cuda.synchronize()
        outs.append((y, stat))
    assert torch.isfinite(outs[0][0]).all()
    assert torch.equal(outs[0][0], outs[2][0])                         # the shipped build is deterministic on these launches
    assert torch.equal(outs[0][0], outs[1][0]), float((outs[0][0] - outs[1][0]).abs().max())
    assert torch.equal(outs[0][1], outs[1][1])


def test_weight_gradient_register_staged_orientation_agrees(libs, monkeypatch):
    """stride 10 (>= 8: the fp32-staged orientation, conv_x6c_kernel<128, 4, true>) and the swapped 1x1 orientation"""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(4)
    for (S, M, Cin, k, st, Tz) in [(4, 128, 96, 30, 10, 3200), (8, 1536, 256, 1, 1, 200)]:
        Tg = Tz // st if k > 1 else Tz
        z = torch.randn(S, Cin, Tz, device=dev, generator=g)
        gg = torch.randn(S, M, Tg, device=dev, generator=g)
        al = torch.rand(Cin, device=dev, generator=g) * 0.5
        res = []
        for autowait in (False, True):
            libs(autowait)
            dw = torch.zeros(M, Cin * k, device=dev)
            db = torch.zeros(M, device=dev)
            K.wgrad_gemm(gg, z, dw, S=S, M=M, Tg=Tg, Ncols=Tg, Cin=Cin, Tz=Tz, taps=k, dbias=db, in_alpha=al, stride=st,
                         padL=(k // 2 - 1) if k > 1 else 0, pad_mode=K.PAD_REFLECT if k > 1 else K.PAD_ZERO)
            assert K.LAST_WGRAD_X6 and K.LAST_WGRAD_KIND in (1, 2)
            torch.cuda.synchronize()
            res.append((dw, db))
        for a, b_ in zip(res[0], res[1]):
            rel = float((a.double() - b_.double()).norm() / b_.double().norm())
            assert rel < 1e-6, rel

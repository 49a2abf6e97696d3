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

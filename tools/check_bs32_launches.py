"""Every encoder data-gradient / forward conv of the PASE+ bs32 step: split-bf16 (x6c) result vs the exact-fp32 pipe."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pase_amd import kernels as K, engine as E
from pase_amd.engine import Act
dev = torch.device("cuda:0"); S = 96
shapes = [("blk1", 64, 64, 20, 10, 32000), ("blk2", 64, 128, 11, 2, 3200), ("blk3", 128, 128, 11, 1, 1600),
          ("blk4", 128, 256, 11, 2, 1600), ("blk5", 256, 256, 11, 1, 800), ("blk6", 256, 512, 11, 2, 800),
          ("blk7", 512, 512, 11, 2, 400)]
torch.manual_seed(0)
for name, Cin, Cout, k, st, Tin in shapes:
    pL, pR = E.reflect_pads(k, st)
    x = torch.randn(S, Cin, Tin, device=dev)
    w = torch.randn(Cout, Cin, k, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    a = Act(x, C=Cin, scale=torch.rand(Cin, device=dev) + 0.5, shift=torch.randn(Cin, device=dev) * 0.1,
            alpha=torch.rand(Cin, device=dev) * 0.3)
    res = {}
    for mode in (True, False):
        K.X6 = mode
        y, stat = E.conv_fwd(a, w.view(Cout, -1), b, Cout=Cout, taps=k, stride=st, padL=pL, padR=pR, pad_mode=K.PAD_REFLECT,
                             want_stats=True)
        kind_f = K.LAST_PLAN_KIND
        dy = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        dx = E.conv_dgrad(dy, w, R=Cout, O=Cin, k=k, stride=st, Tin=Tin, padL=pL, padR=pR, s_red=Cin * k, s_out=k, s_k=1)
        kind_d = K.LAST_PLAN_KIND
        dw = torch.zeros(Cout, Cin * k, device=dev); db = torch.zeros(Cout, device=dev)
        E.conv_wgrad(dy, a, dw, db, taps=k, stride=st, padL=pL, pad_mode=K.PAD_REFLECT)
        res[mode] = (y.clone(), stat.sum(0).clone(), dx.clone(), dw.clone(), db.clone(), kind_f, kind_d, K.LAST_WGRAD_X6)
    def rel(i):
        a_, b_ = res[True][i].double(), res[False][i].double()
        return float((a_ - b_).norm() / b_.norm())
    print("%s: fwd kind %d relL2 %.2e stats %.2e | dgrad kind %d relL2 %.2e | wgrad x6c %s dw %.2e db %.2e" % (
        name, res[True][5], rel(0), rel(1), res[True][6], rel(2), res[True][7], rel(3), rel(4)), flush=True)

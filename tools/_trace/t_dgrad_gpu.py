import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, torch.nn.functional as F
from pase_amd import _lib, build, kernels as K, engine as E
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (S, Cin, Cout, k, st, T) in [(8, 16, 16, 11, 1, 800), (4, 8, 32, 11, 2, 800), (8, 16, 160, 11, 1, 800), (8, 128, 16, 11, 1, 800), (8, 256, 256, 11, 1, 800)]:
    w = torch.randn(Cout, Cin, k) * 0.1
    padL, padR = E.reflect_pads(k, st)
    Tg = (T + padL + padR - k) // st + 1
    dy = torch.randn(S, Cout, Tg)
    xp = torch.zeros(S, Cin, T + padL + padR, dtype=torch.float64, requires_grad=True)
    F.conv1d(xp, w.double(), None, stride=st).backward(dy.double())
    ref = xp.grad
    for cap in ("256", "3"):
        os.environ["PASE_X6C_MAXWG"] = cap
        K.X6 = True
        dx = E.conv_dgrad(dy.to(dev), w.to(dev), R=Cout, O=Cin, k=k, stride=st, Tin=T, padL=padL, padR=padR, s_red=Cin * k, s_out=k, s_k=1)
        print(S, Cin, Cout, k, st, T, "cap", cap, "kind", K.LAST_PLAN_KIND, "rel", float((dx.double().cpu() - ref).norm() / ref.norm()))

"""Run one conv shape a few times (for rocprofv3 --pmc passes).  usage: run_one_shape.py blk5 fwd|wgrad|dgrad"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pase_amd import kernels as K, engine as E
from pase_amd.engine import Act
shapes = {"sinc": (1, 64, 251, 1, 32000), "blk1": (64, 64, 20, 10, 32000), "blk3": (128, 128, 11, 1, 1600),
          "blk5": (256, 256, 11, 1, 800), "blk7": (512, 512, 11, 2, 400)}
name, what = sys.argv[1], sys.argv[2]
Cin, Cout, k, st, Tin = shapes[name]
dev = torch.device("cuda:0"); S = 96
pL, pR = E.reflect_pads(k, st) if name != "sinc" else (125, 125)
x = torch.randn(S, Cin, Tin, device=dev); w = torch.randn(Cout, Cin, k, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev); al = torch.full((Cin,), 0.1, device=dev)
a = Act(x, C=Cin, scale=sc, shift=sh, alpha=al)
y, _ = E.conv_fwd(a, w.view(Cout, -1), b, Cout=Cout, taps=k, stride=st, padL=pL, padR=pR, pad_mode=K.PAD_REFLECT, want_stats=True)
dy = torch.randn_like(y); dw = torch.zeros(Cout, Cin * k, device=dev); db = torch.zeros(Cout, device=dev)
for _ in range(3):
    if what == "fwd":
        E.conv_fwd(a, w.view(Cout, -1), b, Cout=Cout, taps=k, stride=st, padL=pL, padR=pR, pad_mode=K.PAD_REFLECT, want_stats=True)
    elif what == "wgrad":
        E.conv_wgrad(dy, a, dw, db, taps=k, stride=st, padL=pL, pad_mode=K.PAD_REFLECT)
    else:
        E.conv_dgrad(dy, w, R=Cout, O=Cin, k=k, stride=st, Tin=Tin, padL=pL, padR=pR, s_red=Cin * k, s_out=k, s_k=1)
torch.cuda.synchronize()

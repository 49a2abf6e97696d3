import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pase_amd import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
S, Cin, Cout, k, st, T = int(os.environ.get("S", "12")), 64, 64, 20, 10, 32000
x = torch.randn(S, Cin, T)
w = torch.randn(Cout, Cin, k) * 0.05
b = torch.randn(Cout) * 0.1
sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
v = x.double() * sc.double()[None, :, None] + sh.double()[None, :, None]
xin = torch.where(v > 0, v, v * al.double()[None, :, None])
P = (k // 2 - 1, k // 2)
ref = F.conv1d(F.pad(xin, P, mode="reflect"), w.double(), b.double(), stride=st)
Tout = ref.shape[2]
for narrow in ("1", "0"):
    os.environ["PASE_X6C_NARROW"] = narrow
    y = torch.zeros(S, Cout, Tout, device=dev)
    stat = K.conv_gemm(x.to(dev), w.reshape(Cout, -1).contiguous().to(dev), y, want_stats=True, S=S, Cin=Cin, Tin=T, M=Cout,
                       K=Cin * k, taps=k, Ncols=Tout, Tout=Tout, bias=b.to(dev), stride=st, padL=P[0], pad_mode=K.PAD_REFLECT,
                       in_scale=sc.to(dev), in_shift=sh.to(dev), in_alpha=al.to(dev))
    d = (y.cpu().double() - ref)
    rel = float(d.norm() / ref.norm())
    bad = (d.abs() > 1e-3).nonzero()
    print("narrow", narrow, "kind", K.LAST_PLAN_KIND, "rel", rel, "max abs", float(d.abs().max()), "n bad", len(bad), bad[:8].tolist())
    s = stat.cpu().double().sum(0)
    print("   stats rel", float((s[:, 0] - ref.sum((0, 2))).abs().max() / ref.sum((0, 2)).abs().max()),
          float(((s[:, 1] - (ref ** 2).sum((0, 2))) / (ref ** 2).sum((0, 2))).abs().max()))

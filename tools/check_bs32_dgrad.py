"""bisect: block-5 shaped launches at bs32, split-bf16 vs fp32 pipe, with / without on-load parameters, forward / dgrad"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pase_amd import kernels as K, engine as E
from pase_amd.engine import Act
dev = torch.device("cuda:0"); S = 96
Cin, Cout, k, st, Tin = 256, 256, 11, 1, 800
torch.manual_seed(0)
pL, pR = E.reflect_pads(k, st)
x = torch.randn(S, Cin, Tin, device=dev)
w = torch.randn(Cout, Cin, k, device=dev) * 0.05
b = torch.randn(Cout, device=dev)
for tag, a, pm in (("fwd params reflect", Act(x, C=Cin, scale=torch.rand(Cin, device=dev) + 0.5, shift=torch.randn(Cin, device=dev) * 0.1, alpha=torch.rand(Cin, device=dev) * 0.3), K.PAD_REFLECT),
                   ("fwd noparams reflect", Act(x, C=Cin), K.PAD_REFLECT), ("fwd noparams zero", Act(x, C=Cin), K.PAD_ZERO),
                   ("fwd alpha zero", Act(x, C=Cin, alpha=torch.rand(Cin, device=dev) * 0.3), K.PAD_ZERO)):
    out = {}
    for mode in (True, False):
        K.X6 = mode
        y, _ = E.conv_fwd(a, w.view(Cout, -1), b, Cout=Cout, taps=k, stride=st, padL=pL, padR=pR, pad_mode=pm, want_stats=False)
        out[mode] = y.clone()
    d = (out[True].double() - out[False].double())
    print("%-22s relL2 %.2e  max %.2e" % (tag, float(d.norm() / out[False].double().norm()), float(d.abs().max())), flush=True)
dy = torch.randn(S, Cout, Tin, device=dev)
out = {}
for mode in (True, False):
    K.X6 = mode
    out[mode] = E.conv_dgrad(dy, w, R=Cout, O=Cin, k=k, stride=st, Tin=Tin, padL=pL, padR=pR, s_red=Cin * k, s_out=k, s_k=1).clone()
d = (out[True].double() - out[False].double())
print("dgrad                  relL2 %.2e  max %.2e  (|ref| max %.2e)" % (float(d.norm() / out[False].double().norm()), float(d.abs().max()), float(out[False].abs().max())))
bad = (d.abs() > 1e-3 * float(out[False].abs().max()))
print("bad elements: %d of %d" % (int(bad.sum()), bad.numel()))
idx = bad.nonzero()
if len(idx):
    import collections
    print("bad s:", sorted(collections.Counter(idx[:, 0].tolist()).items())[:12])
    print("bad channel (first 12):", sorted(collections.Counter(idx[:, 1].tolist()).items())[:12])
    tt = idx[:, 2]
    print("bad t: min %d max %d; histogram by t//64:" % (int(tt.min()), int(tt.max())), sorted(collections.Counter((tt // 64).tolist()).items()))

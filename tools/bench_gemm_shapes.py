"""Micro-benchmark of pase_conv_gemm / pase_wgrad_gemm on the PASE+ bs32 layer shapes (GPU box)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pase_amd import kernels as K
from pase_amd import engine as E
from pase_amd.engine import Act

dev = torch.device("cuda:0")
S = 96


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [("sinc", 1, 64, 251, 1, 32000), ("blk1", 64, 64, 20, 10, 32000), ("blk2", 64, 128, 11, 2, 3200),
          ("blk3", 128, 128, 11, 1, 1600), ("blk4", 128, 256, 11, 2, 1600), ("blk5", 256, 256, 11, 1, 800),
          ("blk6", 256, 512, 11, 2, 800), ("blk7", 512, 512, 11, 2, 400)]
out = []
for name, Cin, Cout, k, st, Tin in shapes:
    pL, pR = E.reflect_pads(k, st) if name != "sinc" else (125, 125)
    x = torch.randn(S, Cin, Tin, device=dev)
    w = torch.randn(Cout, Cin, k, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev); al = torch.full((Cin,), 0.1, device=dev)
    a = Act(x, C=Cin, scale=sc, shift=sh, alpha=al)
    y, _ = E.conv_fwd(a, w.view(Cout, -1), b, Cout=Cout, taps=k, stride=st, padL=pL, padR=pR, pad_mode=K.PAD_REFLECT,
                      want_stats=True)
    Tout = y.shape[2]
    gmac = S * Tout * Cout * Cin * k / 1e9
    t_f = timeit(lambda: E.conv_fwd(a, w.view(Cout, -1), b, Cout=Cout, taps=k, stride=st, padL=pL, padR=pR,
                                    pad_mode=K.PAD_REFLECT, want_stats=True))
    dy = torch.randn_like(y)
    dw = torch.zeros(Cout, Cin * k, device=dev); db = torch.zeros(Cout, device=dev)
    t_w = timeit(lambda: E.conv_wgrad(dy, a, dw, db, taps=k, stride=st, padL=pL, pad_mode=K.PAD_REFLECT))
    t_d = None
    if name != "sinc":
        t_d = timeit(lambda: E.conv_dgrad(dy, w, R=Cout, O=Cin, k=k, stride=st, Tin=Tin, padL=pL, padR=pR,
                                          s_red=Cin * k, s_out=k, s_k=1))
    rec = dict(name=name, gmac=round(gmac, 2), fwd_ms=round(t_f, 3), fwd_tf=round(2 * gmac / t_f, 1),
               wgrad_ms=round(t_w, 3), wgrad_tf=round(2 * gmac / t_w, 1),
               dgrad_ms=None if t_d is None else round(t_d, 3), dgrad_tf=None if t_d is None else round(2 * gmac / t_d, 1))
    print(json.dumps(rec), flush=True)
    out.append(rec)
# wide head: 256 -> 21525 (lps) fwd with fused MSE, wgrad, dgrad
B, F_, D, r = 32, 200, 3075, 7
h = torch.randn(B, 256, F_, device=dev); W = torch.randn(D * r, 256, device=dev) * 0.05; bb = torch.zeros(D * r, device=dev)
lab = torch.randn(B, D, F_, device=dev); g = torch.empty(B, D * r, F_, device=dev); acc = torch.zeros(1, dtype=torch.float64, device=dev)
gmac = B * F_ * D * r * 256 / 1e9
t_f = timeit(lambda: K.conv_gemm(h, W, None, S=B, Cin=256, Tin=F_, M=D * r, K=256, taps=1, Ncols=F_, Tout=F_, bias=bb,
                                 epilogue=K.EPI_MSE_CTX, label=lab, grad_out=g, loss_acc=acc, grad_scale=1e-6, r_ctx=r, label_D=D))
dw = torch.zeros(D * r, 256, device=dev); db = torch.zeros(D * r, device=dev)
t_w = timeit(lambda: E.conv_wgrad(g, Act(h, C=256), dw, db, taps=1))
t_d = timeit(lambda: E.conv_dgrad(g, W, R=D * r, O=256, k=1, stride=1, Tin=F_, padL=0, padR=0, s_red=256, s_out=1, s_k=1))
rec = dict(name="lps_head", gmac=round(gmac, 2), fwd_ms=round(t_f, 3), fwd_tf=round(2 * gmac / t_f, 1), wgrad_ms=round(t_w, 3),
           wgrad_tf=round(2 * gmac / t_w, 1), dgrad_ms=round(t_d, 3), dgrad_tf=round(2 * gmac / t_d, 1))
print(json.dumps(rec), flush=True)
out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_gemm_shapes.json", "w"), indent=1)

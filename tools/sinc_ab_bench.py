"""SincNet layer backward at bs32 size (96 x 64 x 32000): apply pass + weight gradient against the weight gradient with the
apply pass evaluated on load (pase_wgrad_gemm_act_bwd).  usage (GPU box): python tools/sinc_ab_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pase_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
K.X6 = True
S, M, T, taps, pL, pR, d = 96, 64, 32000, 251, 9, 10, 160
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(S, 1, T, device=dev, generator=g)
y = torch.randn(S, M, T, device=dev, generator=g)
dsrc = torch.randn(S, M, T + pL + pR, device=dev, generator=g) * 0.1
dpool = torch.randn(S, 1920, T // d, device=dev, generator=g) * 0.1
scale, shift, alpha = (torch.rand(M, device=dev, generator=g) + 0.5 for _ in range(3))
mean, rstd = torch.randn(M, device=dev, generator=g) * 0.1, torch.rand(M, device=dev, generator=g) + 0.5
sums = torch.zeros(M, 3, dtype=torch.float64, device=dev)
dy = torch.empty(S, M, T, device=dev)
kw = dict(S=S, C_=M, T=T, dsrc=dsrc, Tp=T + pL + pR, padL=pL, pad_mode=K.PAD_REFLECT, dpool=dpool, dpool_ctot=1920, dpool_coff=64,
          pool_F=T // d, pool_d=d, scale=scale, shift=shift, alpha=alpha, mean=mean, rstd=rstd, sums=sums, dy=dy, has_bn=1)
K.act_bwd_reduce(y, **kw)
wk = dict(S=S, M=M, Tg=T, Ncols=T, Cin=1, Tz=T, taps=taps, padL=125, pad_mode=K.PAD_REFLECT)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


dw1, dw2 = torch.zeros(M, taps, device=dev), torch.zeros(M, taps, device=dev)
t_apply = timed(lambda: K.act_bwd_apply(y, **kw))
t_wgrad = timed(lambda: K.wgrad_gemm(dy, x, dw1, **wk))
t_reduce = timed(lambda: K.act_bwd_reduce(y, **dict(kw, sums=torch.zeros_like(sums))))
t_fused = timed(lambda: K.wgrad_gemm(None, x, dw2, g_bwd=dict(kw, y=y, dy=None), **wk))
dw1.zero_(); dw2.zero_()
K.wgrad_gemm(dy, x, dw1, **wk)
K.wgrad_gemm(None, x, dw2, g_bwd=dict(kw, y=y, dy=None), **wk)
rel = float((dw1.double() - dw2.double()).norm() / dw1.double().norm())
print("reduce %.3f ms | apply %.3f + wgrad %.3f = %.3f ms | on-load wgrad %.3f ms | rel diff of the two dfilt %.2e"
      % (t_reduce, t_apply, t_wgrad, t_apply + t_wgrad, t_fused, rel))

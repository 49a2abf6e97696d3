"""The decoder worker's pointwise tail at bs32 size (32 x 128 x 32000): pase_mlp_head1_step against the six launches it replaces
(64-row 1x1 convolution, head1_fwd, head1_bwd, 1x1 weight gradient, 1x1 data gradient, PReLU backward pass).
usage (GPU box): python tools/mlp_head1_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pase_amd import engine as E  # noqa: E402
from pase_amd import kernels as K  # noqa: E402
from pase_amd.engine import Act  # noqa: E402

dev = torch.device("cuda:0")
S, C, T, H = 32, 128, 32000, 64
g = torch.Generator(device=dev).manual_seed(0)
y = torch.randn(S, C, T, device=dev, generator=g)
a0, a1 = torch.rand(C, device=dev, generator=g) * 0.5, torch.rand(H, device=dev, generator=g) * 0.5
w1, b1 = torch.randn(H, C, device=dev, generator=g) * 0.1, torch.randn(H, device=dev, generator=g) * 0.1
w2, b2 = torch.randn(H, device=dev, generator=g) * 0.2, torch.randn(1, device=dev, generator=g) * 0.1
tgt = torch.randn(S, T, device=dev, generator=g)
gs = 1.0 / (S * T)


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


dy = torch.empty(S, C, T, device=dev)
acc = torch.zeros(1, dtype=torch.float64, device=dev)
sums0 = torch.zeros(C, 3, dtype=torch.float64, device=dev)
sums1 = torch.zeros(3 * H + 1, dtype=torch.float64, device=dev)
dw1 = torch.zeros(H, C, device=dev)


def fused():
    K.mlp_head1_step(y, a0, w1, b1, a1, w2, b2, tgt, None, dy, acc, sums0, sums1, dw1, S=S, C_=C, T=T, H=H,
                     loss_type=K.LOSS_L1, grad_scale=gs)


def six():
    cur = Act(y, C=C, alpha=a0)
    z, _ = E.conv_fwd(cur, w1, b1, Cout=H, taps=1, padL=0, padR=0, pad_mode=K.PAD_ZERO)
    dp = torch.empty(S, 1, T, device=dev)
    K.head1_fwd(z, w2, b2, S=S, C_=H, T=T, in_alpha=a1, target=tgt, y=None, dy=dp, loss_acc=acc, loss_type=K.LOSS_L1, grad_scale=gs)
    dz = torch.empty(S, H, T, device=dev)
    sm = torch.zeros(3 * H + 1, dtype=torch.float64, device=dev)
    K.head1_bwd(z, a1, w2, dp, dz, sm, S=S, C_=H, T=T)
    dwb = torch.zeros(H, C, device=dev)
    E.conv_wgrad(dz, cur, dwb, None, taps=1, padL=0, pad_mode=K.PAD_ZERO)
    din = E.conv_dgrad(dz, w1.view(H, C, 1), R=H, O=C, k=1, stride=1, Tin=T, padL=0, padR=0, s_red=C, s_out=1, s_k=1)
    E.act_backward(y, C=C, T=T, S=S, has_bn=False, alpha=a0, dsrc=din, dsrc_ctot=C, Tp=T)


if len(sys.argv) > 1 and sys.argv[1] == "trace":        # -DMH_TRACE build (tools/mlp_head1_ablate.sh): phase clocks of one launch
    import ctypes
    from pase_amd import _lib
    fused()
    buf = (ctypes.c_ulonglong * 32)()
    _lib.lib().pase_mlp_head1_trace_read(buf)
    names = ["copy+wait+barrier", "stage 1", "head", "stage 2 mfma", "epilogue 2", "barrier 2", "stage 3", "end barrier", "tile setup, targets"]
    order = [8, 0, 1, 2, 3, 4, 5, 6, 7]
    for wg, off in ((0, 0), (131, 16)):
        tot = sum(buf[off + i] for i in range(9))
        print("workgroup %3d: %d clocks over its tiles | " % (wg, tot) + "  ".join("%s %.1f%%" % (names[i], 100.0 * buf[off + i] / max(1, tot)) for i in order))
    sys.exit(0)
t_f = timed(fused)
if len(sys.argv) > 1 and sys.argv[1] == "fused":        # tools/mlp_head1_ablate.sh
    print("fused %.3f ms" % t_f)
    sys.exit(0)
t_6 = timed(six)
print("fused %.3f ms | six launches %.3f ms" % (t_f, t_6))

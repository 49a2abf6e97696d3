"""EXPERIMENT: what the scattered pixel-shuffle store costs.  The same split-bf16 GEMMs as the block-1 data-gradient
(M 640 = 10 phases x 64 channels, K 128, 96 x 3202 columns) and the decoder output layer (M 1280, K 768, 32 x 3202),
once with the ps = 10 pixel-shuffle store and once storing the (M, Ncols) tile rows as they are (ps = 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pase_amd import kernels as K

dev = torch.device("cuda:0")


def timeit(fn, n=10):
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


for name, S, Cin, taps, Cout, ps, q in (("blk1 dgrad", 96, 64, 2, 64, 10, 3202), ("decoder out", 32, 256, 3, 128, 10, 3202)):
    M, Kd = ps * Cout, Cin * taps
    Tin = q - taps + 1
    x = torch.randn(S, Cin, Tin, device=dev)
    wt = torch.randn(Kd, (M + 3) // 4 * 4, device=dev) * 0.05
    for mode in ("ps", "plain"):
        if mode == "ps":
            Tout = ps * q
            y = torch.zeros(S, Cout, Tout, device=dev)
            kw = dict(Cout_store=Cout, ps=ps, poff=0, Tout=Tout)
        else:
            y = torch.zeros(S, M, q, device=dev)
            kw = dict(Tout=q)
        f = lambda: K.conv_gemm(x, None, y, wt=wt, S=S, Cin=Cin, Tin=Tin, M=M, K=Kd, taps=taps, Ncols=q, stride=1,
                                tapstep=-1, padL=0, pad_mode=K.PAD_ZERO, splitk=1, **kw)
        ms = timeit(f)
        print("%-12s %-6s %.3f ms  %.1f TFLOP/s" % (name, mode, ms, 2.0 * S * q * M * Kd / ms / 1e9))

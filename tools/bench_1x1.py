"""Flat (1x1) conv_gemm timing sweep: fixed overhead vs K slope.  python tools/bench_1x1.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pase_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")


def run(M, Kd, S, T, reps=10, **kw):
    x = torch.randn(S, Kd, T, device=dev)
    w = torch.randn(M, Kd, device=dev) * 0.05
    y = torch.empty(S, M, T, device=dev)
    args = dict(S=S, Cin=Kd, Tin=T, M=M, K=Kd, taps=1, Ncols=T, Tout=T, splitk=1)
    args.update(kw)
    for _ in range(2):
        K.conv_gemm(x, w, y, **args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        K.conv_gemm(x, w, y, **args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * M * Kd * S * T
    print("M%-6d K%-5d N%dx%d %s: %.3f ms  %.1f TF/s" % (M, Kd, S, T, kw or "", ms, fl / ms / 1e9), flush=True)


for Kd in (64, 256, 1024, 4096):
    run(2560, Kd, 32, 200)
for Kd in (256, 1024):
    run(2560, Kd, 32, 256)
    run(2560, Kd, 8, 1024)
run(21504, 256, 32, 200)
run(21525, 256, 32, 200)
run(20480, 256, 32, 256)

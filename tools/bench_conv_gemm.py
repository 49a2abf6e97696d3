"""Micro-benchmark of pase_conv_gemm on the PASE+ bs32 layer shapes (GPU box only)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pase_amd import kernels as K

dev = torch.device("cuda:0")
S = 96
shapes = [  # name, Cin, Cout, k, stride, Tin, padL
    ("sinc", 1, 64, 251, 1, 32000, 125),
    ("blk1", 64, 64, 20, 10, 32000, 9),
    ("blk2", 64, 128, 11, 2, 3200, 4),
    ("blk3", 128, 128, 11, 1, 1600, 5),
    ("blk4", 128, 256, 11, 2, 1600, 4),
    ("blk5", 256, 256, 11, 1, 800, 5),
    ("blk6", 256, 512, 11, 2, 800, 4),
    ("blk7", 512, 512, 11, 2, 400, 4),
    ("lps_head", 256, 21525, 1, 1, 200, 0),
]
out = []
for name, Cin, Cout, k, st, Tin, padL in shapes:
    Sx = 32 if name == "lps_head" else S
    padR = k // 2 if k > 1 else 0
    Tout = (Tin + padL + padR - k) // st + 1
    x = torch.randn(Sx, Cin, Tin, device=dev)
    w = torch.randn(Cout, Cin * k, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    y = torch.empty(Sx, Cout, Tout, device=dev)
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev); al = torch.full((Cin,), 0.1, device=dev)
    nt = K.stat_tiles(Cout, Sx, Tout)
    stat = torch.empty(nt, Cout, 2, device=dev)
    def run():
        K.conv_gemm(x, w, y, S=Sx, Cin=Cin, Tin=Tin, M=Cout, K=Cin * k, taps=k, Ncols=Tout, Tout=Tout, bias=b,
                    in_scale=sc, in_shift=sh, in_alpha=al, stat_part=stat, stride=st, padL=padL,
                    pad_mode=K.PAD_REFLECT if k > 1 else K.PAD_ZERO)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = 2.0 * Sx * Tout * Cout * Cin * k
    rec = dict(name=name, ms=round(ms, 4), tflops=round(flops / ms / 1e9, 2), gmac=round(flops / 2e9, 2))
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_conv_gemm.json", "w"), indent=1)

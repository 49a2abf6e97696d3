import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pase_amd import kernels as K
dev = torch.device("cuda:0")
S, Cin, Cout, k, T = 384, 256, 256, 11, 800
x = torch.randn(S, Cin, T, device=dev); w = torch.randn(Cout, Cin * k, device=dev) * 0.05; y = torch.empty(S, Cout, T, device=dev)
small = torch.zeros(64 * 256 * 4, device=dev)
side = torch.cuda.Stream()
def gemm(): K.conv_gemm(x, w, y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * k, taps=k, Ncols=T, Tout=T, padL=5, pad_mode=K.PAD_REFLECT)
def run(max_wg):
    K.MAX_WG = max_wg
    gemm()
    with torch.cuda.stream(side): small.add_(1.0)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    main = torch.cuda.current_stream()
    e0.record(main); gemm(); e1.record(main)
    time.sleep(0.0007)
    with torch.cuda.stream(side):
        small.add_(1.0); e2.record(side)
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1), 3), round(e0.elapsed_time(e2), 3)
run(0)
for cap in (0, 248, 240, 224, 192, 128, 64):
    print("cap", cap, run(cap), run(cap))

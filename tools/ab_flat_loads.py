"""A/B of the flat 1x1 stage loop: next stage's loads issued in one burst behind the first iteration (product) vs spread
over the first half of the loop (-DPASE_FLAT_SPREAD build).
  container:  python tools/ab_flat_burst.py build
  GPU box:    python tools/ab_flat_burst.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "_trace", "libpase_flat_spread.so")
SO_GEN = os.path.join(ROOT, "tools", "_trace", "libpase_flat_generic.so")
if sys.argv[1:] == ["build"]:
    from pase_amd import build as B
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    for so, d in ((SO, "-DPASE_FLAT_SPREAD"), (SO_GEN, "-DPASE_FLAT_GENERIC")):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               d, "-I", B.INCLUDE, "-I", B.CSRC, "-Wno-unused-result", "-o", so] + B._sources())
        print(so)
    sys.exit(0)
import torch  # noqa: E402
from pase_amd import _lib  # noqa: E402
from pase_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
shapes = [("dgrad M256 K21525", 256, 21525, 32, 200), ("head M21525 K256", 21525, 256, 32, 200),
          ("M1920 K256", 1920, 256, 96, 200), ("M256 K1920", 256, 1920, 96, 200)]


def run(tag):
    for name, M, Kd, S, T in shapes:
        x = torch.randn(S, Kd, T, device=dev)
        w = torch.randn(M, Kd, device=dev) * 0.05
        y = torch.zeros(S, M, T, device=dev)
        wt = K.pack_wt(w, M=M, K=Kd, Cin=Kd, taps=1)
        f = lambda: K.conv_gemm(x, None, y, wt=wt, S=S, Cin=Kd, Tin=T, M=M, K=Kd, taps=1, Ncols=T, Tout=T)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-8s %-20s %.3f ms  %.1f TFLOP/s" % (tag, name, ms, 2.0 * S * T * M * Kd / ms / 1e9), flush=True)


# (the first library timed in a process runs ~8 % slower than the following ones whatever it is -- clocks / first touch:
#  one untimed pass first, then two interleaved rounds)
_lib.use_library(None, "cuda")
run("warm-up")
for _ in range(2):
    for tag, so in (("burst", None), ("spread", SO), ("generic", SO_GEN)):
        _lib.use_library(so, "cuda")
        run(tag)

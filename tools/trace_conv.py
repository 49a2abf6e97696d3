"""Per-workgroup phase timing of conv_gemm (trace build, -DPASE_TRACE).
Build (container):  python tools/trace_conv.py build
Run (GPU box):      python tools/trace_conv.py M K S T [taps stride]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.environ.get("PASE_TRACE_SO", os.path.join(ROOT, "tools", "_trace", "libpase_trace.so"))
EXTRA = os.environ.get("PASE_TRACE_DEFS", "").split()

if sys.argv[1] == "build":
    from pase_amd import build as B
    srcs = B._sources()
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DPASE_TRACE"] + EXTRA + [
           "-I", B.INCLUDE, "-I", B.CSRC, "-Wno-unused-result", "-o", SO] + srcs
    subprocess.check_call(cmd)
    print(SO)
    sys.exit(0)

import torch  # noqa: E402
from pase_amd import _lib  # noqa: E402
_lib.use_library(SO, "cuda")
from pase_amd import kernels as K  # noqa: E402

M, Cin, S, T = map(int, sys.argv[1:5])
taps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
stride = int(sys.argv[6]) if len(sys.argv) > 6 else 1
affine = len(sys.argv) > 7
dev = torch.device("cuda:0")
Kd = Cin * taps
Tout = T // stride
x = torch.randn(S, Cin, T, device=dev)
w = torch.randn(M, Kd, device=dev) * 0.05
y = torch.empty(S, M, Tout, device=dev)
args = dict(S=S, Cin=Cin, Tin=T, M=M, K=Kd, taps=taps, Ncols=Tout, Tout=Tout, stride=stride, padL=taps // 2,
            pad_mode=K.PAD_REFLECT if taps > 1 else K.PAD_ZERO, splitk=1)
if affine:
    args.update(in_scale=torch.ones(Cin, device=dev), in_shift=torch.zeros(Cin, device=dev),
                in_alpha=torch.full((Cin,), 0.1, device=dev))
for _ in range(3):
    K.conv_gemm(x, w, y, **args)
torch.cuda.synchronize()
nt = ((M + 127) // 128) * ((S * Tout + 127) // 128)
n = min(nt, 16384)
buf = np.zeros(n * 6, dtype=np.uint64)
lib = _lib.lib()
lib.pase_debug_trace.argtypes = [C.c_void_p, C.c_int]
rc = lib.pase_debug_trace(buf.ctypes.data, n)
assert rc == 0, rc
t = buf.reshape(n, 6)
ts = t[:, :5].astype(np.int64)
t0 = ts[:, 0].min()
us = (ts - t0) / 100.0      # 100 MHz
ph = np.diff(us, axis=1)
print("tiles %d; kernel span %.1f us" % (nt, us[:, 4].max()))
for i, name in enumerate(("setup+issue stage0 loads", "stage0 land+store+sync", "main loop", "epilogue+store ack")):
    print("  %-26s mean %7.2f us  p10 %7.2f  p90 %7.2f" % (name, ph[:, i].mean(), np.percentile(ph[:, i], 10),
                                                       np.percentile(ph[:, i], 90)))
print("  WG lifetime mean %.2f us" % (us[:, 4] - us[:, 0]).mean())
acc = np.zeros(n * 4, dtype=np.uint64)
lib.pase_debug_tacc.argtypes = [C.c_void_p, C.c_int]
assert lib.pase_debug_tacc(acc.ctypes.data, n) == 0
acc = acc.reshape(n, 4).astype(np.int64) / 100.0
for i, name in enumerate(("load_stage issue", "k-loop (MFMA)", "store_stage", "barrier wait")):
    print("  in-loop %-18s mean %7.2f us/WG" % (name, acc[:, i].mean()))
hw = t[:, 5]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
cu = ((hw & np.uint64(0xFFFFFFFF)).astype(np.int64) >> 8) & 0xF
se = ((hw & np.uint64(0xFFFFFFFF)).astype(np.int64) >> 13) & 0x7
sh = ((hw & np.uint64(0xFFFFFFFF)).astype(np.int64) >> 12) & 0x1
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
uk = np.unique(key)
print("  distinct CUs seen: %d; WGs per CU min/mean/max: %d / %.1f / %d" % (
    len(uk), min((key == k).sum() for k in uk), n / len(uk), max((key == k).sum() for k in uk)))
# timeline of one CU
k0 = uk[0]
idx = np.where(key == k0)[0]
idx = idx[np.argsort(us[idx, 0])]
print("  one CU timeline (start, s0issued, loop start, loop end, end) us:")
for i in idx[:12]:
    print("    blk %5d: " % i + "  ".join("%7.2f" % v for v in us[i]))
start_order = np.sort(us[:, 0])
print("  WG start times: first wave of %d WGs started by %.2f us; last WG started %.2f us" % (
    min(n, 512), start_order[min(n, 512) - 1], start_order[-1]))

"""Forward-only (eval) throughput of the PASE+ encoder, the drop-in use of README.md:31-39 of the reference:
python tools/bench_inference.py [B] [T]"""
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pase_amd.frontend import wf_builder  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32000
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    fe = wf_builder(os.path.join(ROOT, "cfg", "frontend", "PASE+.cfg")).cuda().eval()
x = (0.1 * torch.randn(B, 1, T, device="cuda")).clamp_(-1, 1)
with torch.no_grad():
    for _ in range(3):
        y = fe(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        y = fe(x)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
gf = 41.7 / 3.0 * 0.0 + 2 * (349.5 / 96.0)      # encoder forward GFLOP per sequence at T = 32000 (349.5 GMAC / 96 seq)
print(json.dumps({"B": B, "T": T, "out": list(y.shape), "ms": round(dt * 1e3, 3),
                  "utterances_per_s": round(B / dt, 1), "encoder_frames_per_s": round(B * (T // 160) / dt, 1),
                  "TFLOPs": round(gf * (T / 32000.0) * B / dt / 1e3, 1)}))

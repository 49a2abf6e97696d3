"""Time the on-device regression targets (LPS / FBanks / MFCC, short + long windows, deltas + ZNorm)
for one PASE+ batch: python tools/bench_dsp.py [B]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pase_amd import dsp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = json.load(open(os.path.join(os.path.dirname(__file__), "..", "cfg", "workers", "workers+.cfg")))
tg = dsp.DeviceTargets(cfg, device="cuda")
for n, f in tg.feats.items():
    D = next(w["num_outputs"] for w in cfg["regr"] if w["name"] == n)
    f.set_stats(torch.zeros(D), torch.ones(D))
wav = (0.1 * torch.randn(B, 1, 32000, device="cuda")).clamp_(-1, 1)
for _ in range(3):
    out = tg(wav)
torch.cuda.synchronize()
res = {}
for n, f in tg.feats.items():
    t0 = time.perf_counter()
    for _ in range(10):
        f(wav)
    torch.cuda.synchronize()
    res[n] = (time.perf_counter() - t0) / 10 * 1e3
t0 = time.perf_counter()
for _ in range(10):
    tg(wav)
torch.cuda.synchronize()
res["all"] = (time.perf_counter() - t0) / 10 * 1e3
print(json.dumps({"B": B, "ms": {k: round(v, 3) for k, v in res.items()},
                  "shapes": {k: list(v.shape) for k, v in out.items()}}))

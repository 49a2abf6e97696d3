"""Time the on-device batch producer (chunking + Reverb + additive noise + DSP targets) for one PASE+ batch:
python tools/bench_producer.py [B] [ir_len]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pase_amd import dsp, producer as P  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 24000
T = 32000
rng = np.random.RandomState(0)
pool = P.WavPool([(0.1 * rng.standard_normal(16000 * 6)).astype(np.float32) for _ in range(64)], "cuda")
irs = [np.r_[np.zeros(40), 1.0, 0.3 * rng.standard_normal(L - 41) * np.exp(-np.arange(L - 41) / (L / 6.0))] for _ in range(8)]
noises = [0.05 * rng.standard_normal(16000 * 10) for _ in range(8)]
cfg = json.load(open(os.path.join(os.path.dirname(__file__), "..", "cfg", "workers", "workers+.cfg")))
tg = dsp.DeviceTargets(cfg, device="cuda")
for n, f in tg.feats.items():
    D = next(w["num_outputs"] for w in cfg["regr"] if w["name"] == n)
    f.set_stats(torch.zeros(D), torch.ones(D))
chunker = P.DeviceChunker(pool, T, random_scale=True, rng=np.random.RandomState(1))
rv = P.DeviceReverb(irs, device="cuda")
ad = P.DeviceAdditive(noises, device="cuda")


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


x = chunker(B)["chunk"].contiguous()
res = {"chunk_gather+scale (3B crops)": timed(lambda: chunker(B)),
       "reverb all B, IR %d taps" % L: timed(lambda: rv(x.clone(), np.arange(B) % 8)),
       "additive all B": timed(lambda: ad(x.clone(), np.arange(B) % 8, np.zeros(B, int), np.full(B, 5.0))),
       "targets (lps/fbank/gtn/mfcc x2)": timed(lambda: tg(x))}
prod = P.DeviceBatchProducer(chunker, rv, 0.5, ad, 0.5, tg, rng=np.random.RandomState(2))
res["full producer (p=0.5 gates)"] = timed(lambda: prod(B))
gf = 2.0 * (T + L - 1) * L * B / 1e9
print(json.dumps({"B": B, "ms": {k: round(v, 3) for k, v in res.items()},
                  "reverb_direct_form_GFLOP": round(gf, 1),
                  "reverb_TFLOPs": round(gf / res["reverb all B, IR %d taps" % L], 2)}))

"""Per-launch breakdown of the MFMA kernels in one PASE+ bs32 step (HIP-event timed, launch order):
python tools/step_breakdown.py [out.json]"""
import contextlib
import io
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pase_amd import kernels as K  # noqa: E402
from pase_amd.trainer import trainer  # noqa: E402

dev = torch.device("cuda", 0)
fe_cfg, wk_cfg, raw = bench.load_cfgs()
torch.manual_seed(2)
with contextlib.redirect_stdout(io.StringIO()):
    tr = trainer(frontend_cfg=dict(fe_cfg), minions_cfg=wk_cfg, cfg=dict(epoch=1, bpe=100), lr_mode="poly", device=dev)
batch = bench.synthetic_batch(1234, 32, 32000, raw, dev)
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
K.GEMM_TIMER = K.GemmTimer()
NREP = 3
for _ in range(NREP):
    tr.train_step(batch)
rows = K.GEMM_TIMER.per_launch()
kernels = list(K.GEMM_TIMER.kernels)
K.GEMM_TIMER = None
n = len(rows) // NREP
agg = []
for i in range(n):
    ms = sorted(rows[i + r * n][3] for r in range(NREP))[NREP // 2]
    f, tag, fl, _ = rows[i]
    agg.append(dict(i=i, family=f, shape=tag, gflop=round(fl / 1e9, 2), ms=round(ms, 4), tflops=round(fl / ms / 1e9, 1),
                    kernel=kernels[i]))
tot = sum(a["ms"] for a in agg)
print("launch  family       ms      TF/s   GFLOP  shape")
for a in sorted(agg, key=lambda a: -a["ms"]):
    print("%3d  %-11s %7.3f  %6.1f  %7.1f  %-52s %s" % (a["i"], a["family"], a["ms"], a["tflops"], a["gflop"], a["shape"], a["kernel"]))
print("total MFMA-kernel ms/step: %.2f over %d launches" % (tot, n))
if len(sys.argv) > 1:
    json.dump(agg, open(sys.argv[1], "w"), indent=0)

"""Host-side cost of enqueueing one PASE+ bs32 step (315 launches of Python -> ctypes -> HIP), alone and beside N busy
Python neighbours -- the single-GPU proxy for "eight enqueue loops on one host" (round-5 review item 9: a data-parallel run
is one Python process per GPU; if the enqueue of a step took longer than the step, the GPU would idle and the non-collective
segments would have to be captured in hipGraphs).  The neighbours are pure CPU burners (interpreter-bound loops, as an
enqueue loop is), the GPU stays exclusive to the measured process, so the number is not polluted by eight ranks
time-slicing one device (which is what `bench.py --gpus 8` on a 1-GPU box measures: host_enqueue_ms there is back-pressure).

  python tools/host_time.py [--hogs 7] [--steps 20] [--json out.json]
"""
import argparse
import contextlib
import io
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pase_amd.trainer import trainer  # noqa: E402

HOG = "import json\nx = {i: [i] * 8 for i in range(256)}\nwhile True:\n    json.loads(json.dumps(x))\n"


def measure(tr, batch, steps):
    for _ in range(3):
        tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c0 = time.thread_time()
    for _ in range(steps):
        tr.train_step(batch)
    t1 = time.perf_counter()
    c1 = time.thread_time()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return dict(host_enqueue_ms_per_step=round((t1 - t0) / steps * 1e3, 3), host_cpu_ms_per_step=round((c1 - c0) / steps * 1e3, 3),
                step_ms=round((t2 - t0) / steps * 1e3, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hogs", type=int, default=7)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    fe_cfg, wk_cfg, raw = bench.load_cfgs()
    torch.manual_seed(2)
    with contextlib.redirect_stdout(io.StringIO()):
        tr = trainer(frontend_cfg=dict(fe_cfg), minions_cfg=wk_cfg, cfg=dict(epoch=1, bpe=1000), lr_mode="poly", device=dev)
    batch = bench.synthetic_batch(1234, 32, 32000, raw, dev)
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count()
    out = dict(host_cpus=cpus, steps=args.steps, alone=measure(tr, batch, args.steps))
    # the same with every process (this one and the neighbours) confined to as many CPUs as there are processes: the worst
    # case of a host that has no spare cores at all
    hogs = [subprocess.Popen([sys.executable, "-c", HOG]) for _ in range(args.hogs)]
    try:
        time.sleep(1.0)
        out["with_%d_busy_python_neighbours" % args.hogs] = measure(tr, batch, args.steps)
        if hasattr(os, "sched_setaffinity") and cpus > args.hogs + 1:
            few = sorted(os.sched_getaffinity(0))[:args.hogs + 1]
            for h in hogs:
                os.sched_setaffinity(h.pid, few)
            os.sched_setaffinity(0, few)
            time.sleep(0.5)
            out["same_confined_to_%d_cpus" % len(few)] = measure(tr, batch, args.steps)
    finally:
        for h in hogs:
            h.kill()
        for h in hogs:
            h.wait()
    print(json.dumps(out))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py -- PASE+ self-supervised training step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = forward (SincNet + conv stack + QRNN + dense skips) + 12 worker heads + losses +
backward + gradient all-reduce (N>1) + 13 Adam updates, on a synthetic batch already resident in
HBM: cfg/frontend/PASE+.cfg + cfg/workers/workers+.cfg, bs32 per GPU, 32 000-sample chunks
(BASELINE.json configs[2]); regression targets are N(0,1) tensors ("targets-given" mode of
SURVEY.md section 8d).  Prints ONE JSON line from rank 0.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_UTT_TRAIN = 122.0      # SURVEY.md 8(d) / BASELINE.md section 3: canonical algorithmic FLOPs (PASE+ / workers+)
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
X6_MFMA_PER_PRODUCT = 6          # split-bf16 contraction: hh + hm + mh + hl + lh + mm per fp32 product


# BASELINE.json configs[4]: the dense PASE+ encoder with two QRNN layers and norm_type 'lnorm', emb_dim 256, 64 utterances per
# GPU (/root/reference/template_scripts/run_pase_train_50h_2xQRNN_addrev_lnorm_EMB256.sh:6 names a cfg file the reference does
# not ship: built from PASE+.cfg through WaveFe's own keyword arguments, as tests/test_bench_config.py's `emb256` gate does)
VARIANTS = {"pase+": dict(fe_over={}, batch=32, gflop_per_utt=GFLOP_PER_UTT_TRAIN,
                          workload="PASE+.cfg + workers+.cfg self-supervised train step (BASELINE.json configs[2])"),
            "emb256": dict(fe_over=dict(rnn_layers=2, norm_type="lnorm"), batch=64,
                           # one more QRNN layer: + 30.20 GMAC forward per 32-utterance step, x 3 x 2
                           gflop_per_utt=GFLOP_PER_UTT_TRAIN + 6.0 * 30.20 / 32.0,
                           workload="PASE+ EMB256 variant: PASE+.cfg with rnn_layers=2, norm_type='lnorm' (LayerNorm blocks, "
                                    "InstanceNorm norm_out) + workers+.cfg, 64 utterances per GPU (BASELINE.json configs[4], one GPU)")}


def load_cfgs(variant="pase+"):
    with open(os.path.join(ROOT, "cfg", "frontend", "PASE+.cfg")) as f:
        fe = dict(json.load(f), **VARIANTS[variant]["fe_over"])
    from pase_amd.utils import strip_transforms, worker_parser
    wk = strip_transforms(worker_parser(os.path.join(ROOT, "cfg", "workers", "workers+.cfg")))
    with open(os.path.join(ROOT, "cfg", "workers", "workers+.cfg")) as f:
        raw = json.load(f)
    return fe, wk, raw


def synthetic_batch(seed, B, T, raw, device):
    g = torch.Generator(device=device).manual_seed(seed)
    batch = {k: (0.1 * torch.randn(B, 1, T, generator=g, device=device)).clamp_(-1, 1)
             for k in ("chunk", "chunk_ctxt", "chunk_rand", "cchunk")}
    for w in raw["regr"]:
        if w["name"] not in batch:
            batch[w["name"]] = torch.randn(B, w["num_outputs"], T // 160, generator=g, device=device)
    return batch


def cpu_baseline(raw, fe_cfg, seconds_budget=45.0, B=32, max_steps=2):
    """The CPU oracle (port of the reference step: oracle/pase_oracle.py) timed on the host cores on a
    bounded sample of the benchmark's own workload: full-width model, B utterances x 32 000 samples (default: the
    benchmark's batch size, SURVEY 8d "same synthetic batch, same step definition"), fwd + losses + backward + Adam,
    as many steps as fit the budget (>= 1 after one warm-up; about 20 s per step at B = 32 on 32 threads)."""
    from oracle import pase_oracle as O
    from pase_amd.pase import pase
    from pase_amd.utils import strip_transforms, worker_parser
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # torch's CPU kernels stop scaling (and with hundreds of threads on small tensors collapse) long
    # before a GPU host's full core count: use at most 32 threads and report that number as `cores`
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)
    torch.manual_seed(2)
    with contextlib.redirect_stdout(io.StringIO()):
        wk = strip_transforms(worker_parser(os.path.join(ROOT, "cfg", "workers", "workers+.cfg")))
        model = pase(frontend_cfg=dict(fe_cfg), minions_cfg=wk, cls_lst=["mi", "cmi"],
                     regr_lst=[w["name"] for w in raw["regr"]])
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, _ in model.named_parameters()]
    for n in names:
        P[n].requires_grad_(True)
    opt = torch.optim.Adam([P[n] for n in names], lr=5e-4)
    T = 32000
    batch = synthetic_batch(99, B, T, raw, torch.device("cpu"))

    def step():
        opt.zero_grad()
        so = {}
        h, chunk, preds, labels = O.pase_forward(P, fe_cfg, raw, batch, True, so)
        O.pase_losses(raw, preds, labels)["total"].backward()
        opt.step()

    tw = time.time()
    step()                                   # warm-up (also bounds the sample: a slow host stops here)
    warm = time.time() - tw
    t0 = time.time()
    n = 0
    if warm > seconds_budget:
        n, t0 = 1, tw                        # report the warm-up step itself rather than blow the budget
    else:
        while True:
            step()
            n += 1
            if time.time() - t0 > seconds_budget or n >= max_steps:
                break
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 4), "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": "oracle/pase_oracle.py (torch-CPU restatement of the reference step) full-width PASE+ + "
                      "workers+, B=%d x %d samples, %d timed steps after 1 warm-up, %d threads (host exposes %d)"
                      % (B, T, n, cores, avail)}


def cpu_targets_baseline(raw, T=32000):
    """SURVEY 8(d), mode (ii): the CPU restatement of the target transforms (oracle/dsp_oracle.py: numpy / scipy,
    what the reference's DataLoader workers run through librosa / python_speech_features / gammatone), one
    utterance, one thread."""
    import numpy as np
    from oracle import dsp_oracle as D
    x = (0.1 * np.random.RandomState(0).standard_normal(T)).astype(np.float32)
    from oracle import swipe_oracle as SW
    fns = {"lps": D.lps, "fbank": D.fbanks, "gtn": D.gammatone, "mfcc": D.mfcc,
           "prosody": lambda x_, **kw_: D.prosody(x_, SW.swipe(x_), **kw_)}
    per = {}
    t_all = 0.0
    for w in raw["regr"]:
        base = next((k for k in fns if k in w["name"]), None)
        if base is None:
            continue
        kw = dict(w.get("transform", {}))
        t0 = time.time()
        fns[base](x, **kw)
        per[w["name"]] = round(time.time() - t0, 3)
        t_all += per[w["name"]]
    return {"value": round(1.0 / t_all, 3), "unit": "utterances/s", "cores": 1, "kind": "port",
            "seconds_per_utterance": per,
            "sample": "oracle/dsp_oracle.py + swipe_oracle.py: LPS / FBANK / gammatone / MFCC (+ _long variants) / prosody of one %d-sample "
                      "utterance, single thread" % T}


def torch_rocm_baseline(raw, fe_cfg, device, B, T, steps=3):
    """SURVEY 8(d) comparator "reference module code on stock PyTorch-ROCm ops": the same torch restatement of the
    reference step (oracle/pase_oracle.py -- MIOpen / rocBLAS kernels through torch.nn.functional, autograd,
    torch.optim.Adam) timed on this GPU at the bench's own batch size.  A reported baseline like cpu_baseline;
    it is not on the product path."""
    from oracle import pase_oracle as O
    from pase_amd.pase import pase
    from pase_amd.utils import strip_transforms, worker_parser
    torch.manual_seed(2)
    with contextlib.redirect_stdout(io.StringIO()):
        wk = strip_transforms(worker_parser(os.path.join(ROOT, "cfg", "workers", "workers+.cfg")))
        model = pase(frontend_cfg=dict(fe_cfg), minions_cfg=wk, cls_lst=["mi", "cmi"],
                     regr_lst=[w["name"] for w in raw["regr"]])
    P = {k: v.detach().clone().to(device) for k, v in model.state_dict().items()}
    del model
    names = [n for n in P if P[n].is_floating_point() and "running" not in n]
    for n in names:
        P[n].requires_grad_(True)
    opt = torch.optim.Adam([P[n] for n in names], lr=5e-4)
    batch = synthetic_batch(99, B, T, raw, device)

    def step():
        opt.zero_grad()
        so = {}
        h, chunk, preds, labels = O.pase_forward(P, fe_cfg, raw, batch, True, so)
        O.pase_losses(raw, preds, labels)["total"].backward()
        opt.step()

    step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    return {"value": round(B / dt, 3), "unit": "utterances/s", "ms_per_step": round(dt * 1e3, 2),
            "kind": "port on torch ROCm ops", "sample": "oracle/pase_oracle.py on %s, B=%d x %d samples, %d timed steps "
            "after 1 warm-up (QRNN recurrence as a %d-step Python loop: torchqrnn's CUDA kernel does not exist on ROCm)"
            % (torch.cuda.get_device_name(0), B, T, steps, T // 160)}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: spawn N ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in the env, exactly what torch.distributed.run would set); rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0:                     # one rank died: the others would wait on it forever
                rc = rc or code
                for q in alive:
                    q.terminate()
    sys.exit(rc)


def main():
    # the host driver only supports dmabuf IPC: RCCL / cross-process CUDA tensors need this (set before any HIP init)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU (default: 32; 64 for --variant emb256)")
    ap.add_argument("--chunk", type=int, default=32000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the second timed leg (batch handed over as host buffers)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step in a hipGraph (trainer.capture_step) and time the replays (N=1 only)")
    ap.add_argument("--torch-gpu-baseline", action="store_true",
                    help="(default since round 6; kept for old command lines) time the torch-op restatement of the reference "
                         "step on this GPU (stock PyTorch-ROCm kernels) and report it as `torch_rocm_baseline`")
    ap.add_argument("--rccl-max-nchannels", type=int, default=0,
                    help="N > 1: NCCL_MAX_NCHANNELS for RCCL (each channel occupies a CU that the backward's kernels then do "
                         "not get); 0 = leave RCCL's default")
    ap.add_argument("--reserve-cus", type=int, default=-1,
                    help="N > 1: CUs the persistent split-bf16 GEMM grids leave free for RCCL's channel kernels (grid cap = "
                         "CUs of the device - this); -1 = 32 for N > 1 (one CU per shader engine: DESIGN.md section 6), 0 for N = 1")
    ap.add_argument("--cpu-baseline-b2", action="store_true",
                    help="time the CPU baseline on a B = 2 sample (about 25 s) instead of the benchmark's own batch size "
                         "(default: B = --batch, 1 warm-up + up to 2 timed steps, about a minute of host time)")
    ap.add_argument("--cpu-baseline-bs32", action="store_true",
                    help="the CPU baseline with 3 timed steps at the benchmark's batch size (SURVEY 8d's '>= 3 steps after 1 "
                         "warm-up': about 80 s of host time)")
    ap.add_argument("--variant", choices=sorted(VARIANTS), default="pase+",
                    help="pase+ = BASELINE.json configs[2] (the headline workload); emb256 = configs[4]'s model on one GPU "
                         "(2 x QRNN, lnorm, 64 utterances per GPU unless --batch is given)")
    ap.add_argument("--no-torch-gpu-baseline", action="store_true",
                    help="skip the stock-PyTorch-ROCm comparator (SURVEY 8d: the reference step's module code on torch ops, "
                         "same GPU, same batch size, 3 timed steps)")
    ap.add_argument("--no-capped-leg", action="store_true",
                    help="N = 1: skip the extra K steps that price the data-parallel CU reservation (n_gt1_cap_cost)")
    ap.add_argument("--producer", action="store_true",
                    help="BASELINE.json configs[3] shape: every step's batch is produced ON DEVICE inside the timed "
                         "region (random crops of a resident waveform pool, Reverb / additive-noise gating on "
                         "`chunk`, LPS / FBANK / MFCC targets from the clean chunk); default is configs[2] "
                         "(batch and targets resident in HBM)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)         # plain `python bench.py --gpus N`: spawn one rank per GPU ourselves
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU path")
    # one process per GPU over RCCL.  With fewer GPUs than ranks (a 1-GPU box) the ranks share devices and the
    # exchange falls back to gloo: a functional smoke of the N>1 code path, flagged in the JSON line, not a number.
    shared = ndev < world
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.rccl_max_nchannels > 0:       # default: RCCL's own choice (recorded in config.rccl_max_nchannels)
            os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_max_nchannels)
        # (the contract is ONE JSON line on stdout: gloo's C++ side prints "[Gloo] Rank 0 is connected to ..." to fd 1 while the
        #  group forms -- fd 1 points at stderr for the duration of the rendezvous)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if shared:
                backend = "gloo"
                dist.init_process_group("gloo")
            else:
                backend = "nccl"
                dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from pase_amd import _lib
    _lib.lib()   # fail loudly if the HIP library is missing
    from pase_amd.trainer import trainer

    fe_cfg, wk_cfg, raw = load_cfgs(args.variant)
    if args.batch <= 0:
        args.batch = VARIANTS[args.variant]["batch"]
    torch.manual_seed(2)             # train.py:376 default seed
    with contextlib.redirect_stdout(io.StringIO()):
        tr = trainer(frontend_cfg=dict(fe_cfg), minions_cfg=wk_cfg,
                     cfg=dict(fe_lr=1e-3, min_lr=5e-4, epoch=1, bpe=max(1, args.steps + args.warmup),
                              reserve_cus=(args.reserve_cus if args.reserve_cus >= 0 else (32 if world > 1 else 0))),
                     lr_mode="poly", device=dev)
    B, T = args.batch, args.chunk
    batch = synthetic_batch(1234 + rank, B, T, raw, dev)
    next_batch = lambda: batch
    if args.producer:
        import numpy as np
        from pase_amd import dsp, producer as PR
        rs = np.random.RandomState(1234 + rank)
        pool = PR.WavPool([(0.1 * rs.standard_normal(16000 * 6)).clip(-1, 1).astype(np.float32) for _ in range(64)], dev)
        irs = [np.r_[np.zeros(40), 1.0, 0.3 * rs.standard_normal(23959) * np.exp(-np.arange(23959) / 4000.0)]
               for _ in range(8)]                                     # synthetic exponentially-decaying IRs, 24 000 taps
        noises = [0.05 * rs.standard_normal(16000 * 10) for _ in range(8)]
        tg = dsp.DeviceTargets(raw, device=dev)
        for n_, f_ in tg.feats.items():
            D_ = next(w["num_outputs"] for w in raw["regr"] if w["name"] == n_)
            f_.set_stats(torch.zeros(D_), torch.ones(D_))
        prod = PR.DeviceBatchProducer(PR.DeviceChunker(pool, T, rng=rs), PR.DeviceReverb(irs, device=dev), 0.5,
                                      PR.DeviceAdditive(noises, device=dev), 0.5, tg, rng=rs)
        missing = [w["name"] for w in raw["regr"] if w["name"] not in tg.feats and w["name"] != "cchunk"]
        assert not missing, missing          # every regression target (incl. prosody: SWIPE' f0) is produced on device

        def next_batch():
            return prod(B)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.graph and world == 1 and not args.producer:
        tr.capture_step(batch)
    for _ in range(args.warmup):
        losses = tr.train_step(next_batch())
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    ev0.record()
    for _ in range(args.steps):
        losses = tr.train_step(next_batch())
    ev1.record()
    sync()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    utt_s = B * world * args.steps / dt
    total_loss = float(losses["total"])

    # ---- N > 1: what the collectives cost, from events on the two streams (two extra steps outside the timed region),
    # and the same rank's step WITHOUT any collective (the N = 1 code path on this GPU while its neighbours do the same)
    multi = None
    if world > 1:
        tr.comm_diag = True
        for _ in range(2):
            tr.train_step(next_batch())
        rep = tr.comm_report()
        tr.comm_diag = False
        sync()
        t0 = time.time()
        for _ in range(max(2, args.steps // 2)):
            tr._eager_step(next_batch(), local_only=True)
        sync()
        local_ms = (time.time() - t0) / max(2, args.steps // 2) * 1e3
        vals = torch.tensor([rep["comm_exposed_ms"], rep["comm_total_ms"], rep["host_enqueue_ms"], local_ms] if rep else
                            [0.0, 0.0, 0.0, local_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
        multi = {"comm_exposed_ms": round(float(vals[0]), 3), "comm_total_ms": round(float(vals[1]), 3),
                 "host_enqueue_ms": round(float(vals[2]), 3), "local_step_ms_no_collectives": round(float(vals[3]), 3),
                 "per_gpu_consistency": round(float(vals[3]) / ms_per_step, 4),
                 "rank0_buckets": rep["buckets"] if rep else None,
                 "rank0_backward_end_ms": rep["backward_end_ms"] if rep else None,
                 "how": "max over ranks; events on the main and the collective stream of one step (trainer.comm_report); "
                        "local_step = the same rank's eager step with no collective, all ranks busy at once; "
                        "per_gpu_consistency = local_step / ms_per_step (1.0 = the collectives and N enqueue loops cost nothing)"}

    # ---- the same K steps with the batch handed over as HOST buffers: every step's 4 waveform tensors + 9 target
    # tensors (205 MB at bs32) cross PCIe inside the timed region, through the pinned ring / copy stream of
    # pase_amd.producer.PinnedBatchFeeder (SURVEY 8d step definition; reference modules.py:16-31, pase.py:338)
    h2d = None
    if not args.producer and not args.no_h2d:
        from pase_amd.producer import PinnedBatchFeeder
        ring = []
        for j in range(3):
            hb = synthetic_batch(4000 + 17 * rank + j, B, T, raw, torch.device("cpu"))
            ring.append({k: v.pin_memory() for k, v in hb.items()})
        cnt = [0]

        def host_source():
            cnt[0] += 1
            return ring[cnt[0] % len(ring)]
        feeder = PinnedBatchFeeder(host_source, dev, depth=2)
        for _ in range(max(1, min(2, args.warmup))):
            tr.train_step(feeder.next())
        sync()
        t0 = time.time()
        for _ in range(args.steps):
            tr.train_step(feeder.next())
        sync()
        dth = time.time() - t0
        if world > 1:
            t = torch.tensor([dth], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dth = float(t.item())
        h2d = {"included": True, "value": round(B * world * args.steps / dth, 3), "unit": "utterances/s",
               "ms_per_step": round(dth / args.steps * 1e3, 3), "MB_per_step_per_gpu": round(feeder.bytes_per_batch / 1e6, 1),
               "how": "pinned host ring -> copy stream -> double-buffered device slots, overlapped with the previous step"}
        del feeder, ring

    # ---- N = 1: what the data-parallel step's CU reservation costs THIS GPU's step -- the same K steps with the encoder
    # backward's persistent grids capped at (CUs - 32), exactly the launches trainer._step_ddp caps while collectives are in
    # flight (no collective runs here: this is the price paid before a byte moves)
    capped = None
    if world == 1 and not args.producer and not args.graph and tr.reserve_cus == 0 and not args.no_capped_leg:
        saved = (tr.reserve_cus, tr._grid_cap, tr.cfg.get("reserve_scope"))
        ncu = int(torch.cuda.get_device_properties(dev).multi_processor_count)
        tr.reserve_cus, tr._grid_cap = 32, ncu - 32
        tr.cfg["reserve_scope"] = "encoder_backward"
        try:
            for _ in range(2):
                tr.train_step(batch)
            sync()
            t0 = time.time()
            for _ in range(args.steps):
                tr.train_step(batch)
            sync()
            capped_ms = (time.time() - t0) / args.steps * 1e3
        finally:
            tr.reserve_cus, tr._grid_cap = saved[0], saved[1]
            tr.cfg["reserve_scope"] = saved[2] or "step"
        capped = {"ms_per_step_capped": round(capped_ms, 3), "reserve_cus": 32, "scope": "encoder backward (what N > 1 caps)",
                  "cost_frac": round(capped_ms / ms_per_step - 1.0, 4)}

    # ---- dominant-kernel roofline, measured live: two extra steps with every MFMA-kernel launch
    # bracketed by HIP events on the launch stream (outside the timed region above)
    from pase_amd import kernels as K
    K.GEMM_TIMER = K.GemmTimer()
    # one untimed step in the timer's own layout first (with the timer on, every stream of the step is serialised onto the main
    # one: the caching allocator meets block sizes it has not seen, and a hipMalloc between a launch's start event and the launch
    # is GPU idle time inside the bracket -- round 6: a run's weight-gradient figure read 130 instead of 157 TFLOP/s for it)
    tr._eager_step(batch)
    sync()
    K.GEMM_TIMER = K.GemmTimer()
    extra = 2
    for _ in range(extra):
        tr._eager_step(batch)          # (eager even when the timed steps were graph replays: per-launch events)
    fams = K.GEMM_TIMER.summary()
    fams_pipe = K.GEMM_TIMER.summary(by_pipe=True)
    fams_kern = K.GEMM_TIMER.summary(by_kernel=True)
    K.GEMM_TIMER = None
    traffic = None
    traffic_src = None
    prof = None
    try:
        # HBM traffic cannot be counted from inside the run (PMC passes need rocprofv3): it comes from this round's
        # profile summary, and ONLY if that profile was taken of the very library that is loaded now (source digest)
        from pase_amd import build as _B
        for tag in ("r06", "r05", "r04"):   # the newest profile summary taken of the very library loaded now
            fn = os.path.join(ROOT, "profiles", "summary_%s.json" % tag)
            if os.path.exists(fn):
                with open(fn) as f:
                    cand = json.load(f)
                if cand.get("lib_digest") == _B.hip_digest():
                    prof, traffic_src = cand, "profiles/summary_%s.json" % tag
                    break
        if prof is None:
            raise ValueError("no profile of this build")
        # PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs) summed over every conv_gemm
        # instantiation, GB per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B
        # requests as 64 B)
        conv_k = ("conv_gemm_kernel<", "conv_x6c_kernel<192", "conv_x6c_kernel<128, 3, false", "conv_x6c_kernel<320", "sinc_x6_fwd_kernel")   # every pase_conv_gemm kernel
        mb = sum(row["fetch_MB_x2"] + row["write_MB"] for row in prof.get("hbm_traffic_per_step", [])
                 if row["kernel"].startswith(conv_k))
        calls = sum(k["calls"] for k in prof["step_kernel_time"]["families"] if k["kernel"].startswith(conv_k))
        if mb > 0 and calls > 0:
            traffic = round(mb / 1e3 / calls, 4)
    except Exception:
        traffic = None

    # ---- the dominant kernel INSTANTIATION (most HIP-event time over the launches that ran it), so that a reader can recompute
    # its roofline fraction in one comparison: live numbers from the events above, PMC numbers from the digest-matched profile
    dominant = None
    try:
        named = {k: v for k, v in fams_kern.items() if k}
        kname = max(named, key=lambda k: named[k]["ms"])
        v = named[kname]
        x6k = kname.startswith(("conv_x6c_kernel", "sinc_x6"))
        kpeak = PEAK_BF16_MFMA_TFLOPS / X6_MFMA_PER_PRODUCT if x6k else PEAK_F32_MFMA_TFLOPS
        dominant = {"kernel": kname, "launches_per_step": v["launches"] // extra, "ms_per_step": round(v["ms"] / extra, 3),
                    "algorithmic_gflop": round(v["flops"] / extra / 1e9, 1),
                    "achieved": round(v["flops"] / v["ms"] / 1e9, 2), "peak": round(kpeak, 1), "unit": "TFLOP/s",
                    "frac": round(v["flops"] / v["ms"] / 1e9 / kpeak, 4),
                    "how": "sum over the launches that ran this instantiation of 2*S*Ncols*M*K (the descriptors' contraction "
                           "sizes) / sum of their HIP-event durations on the launch stream (each launch's own operand packs "
                           "included), streams serialised; mfma_util / traffic_GB / profile_ms_per_step: rocprofv3 passes of "
                           "exactly this build (null otherwise)",
                    "mfma_util": None, "traffic_GB": None, "profile_ms_per_step": None, "profile_source": None}
        if prof is not None:
            for row in prof.get("sq_counters_per_step", []):
                if row["kernel"].startswith(kname):
                    dominant["mfma_util"] = row.get("mfma_util")
            for row in prof.get("hbm_traffic_per_step", []):
                if row["kernel"].startswith(kname):
                    dominant["traffic_GB"] = round((row["fetch_MB_x2"] + row["write_MB"]) / 1e3, 3)
            for row in prof["step_kernel_time"]["families"]:
                if row["kernel"].startswith(kname):
                    dominant["profile_ms_per_step"] = round(row["total_us"] / 1e3, 3)
                    dominant["profile_launches_per_step"] = row["calls"]
            dominant["profile_source"] = traffic_src
    except Exception as e:
        dominant = {"kernel": None, "error": repr(e)}

    if rank == 0:
        gflop_utt = VARIANTS[args.variant]["gflop_per_utt"]
        achieved = gflop_utt * utt_s / world / 1e3     # TFLOP/s per GPU (algorithmic)
        cg = fams.get("conv_gemm", dict(launches=1, flops=0.0, ms=1.0))
        wg = fams.get("wgrad_gemm", dict(launches=1, flops=0.0, ms=1.0))
        cg_tf = cg["flops"] / cg["ms"] / 1e9
        wg_tf = wg["flops"] / wg["ms"] / 1e9
        # the contractions run as 6 bf16 MFMAs per fp32 product (PaseConvGemm::wx6 / PaseWgrad::x6; fp32-grade result):
        # the pipe's ceiling in ALGORITHMIC (fp32-equivalent) FLOP/s is the dense bf16 peak / 6.  PASE_X6=0 puts them
        # back on the fp32 matrix pipe (157.3 TFLOP/s).
        x6 = bool(K.X6)
        peak = PEAK_BF16_MFMA_TFLOPS / X6_MFMA_PER_PRODUCT if x6 else PEAK_F32_MFMA_TFLOPS
        peak_note = ("peak = dense bf16 MFMA peak (2500) / 6 MFMAs per fp32 product = 416.7 fp32-equivalent TFLOP/s; "
                     "the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) peaks at 157.3") if x6 else \
                    "peak = v_mfma_f32_32x32x2_f32 dense peak"
        out = {
            "metric": "utterances/sec (PASE+ bs32 32k-sample chunks; encoder-frames/sec = 600 x)",
            "value": round(utt_s, 3), "unit": "utterances/s",
            "encoder_frames_per_s": round(utt_s * 3 * (T // 160), 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (contractions: fp32 operands as 3 round-to-nearest bf16 pieces, 6 bf16 MFMAs per product into two fp32 "
                      "accumulators -- unbiased, error below an fp32 fma chain's -- or the exact-fp32 MFMA, routed per launch)")
            if x6 else "f32", "data": "synthetic",
            "config": {"workload": ("PASE+.cfg + workers+.cfg train step with the batch produced on device each step: crops "
                                    "of a resident pool, Reverb(24000-tap synthetic IRs, p=0.5) + additive noise (p=0.5), "
                                    "LPS/FBANK/gammatone/MFCC targets from the clean chunk (BASELINE.json configs[3] shape)")
                       if args.producer else VARIANTS[args.variant]["workload"],
                       "batch_per_gpu": B, "global_batch": B * world, "chunk_samples": T,
                       "targets": ("lps/lps_long/fbank/fbank_long/gtn/gtn_long/mfcc/mfcc_long/prosody all computed on device from the clean chunk"
                                   if args.producer else "given (N(0,1) tensors resident in HBM)"), "parallelism": "dp%d" % world,
                       "final_total_loss": round(total_loss, 5), "inputs": "resident in HBM (see `h2d` for the "
                       "host-buffer leg)", "collective_backend": backend,
                       "hipgraph": bool(getattr(tr, "_graph", None) is not None)},
            "roofline": {"bound": "mfma", "kernel": "pase_conv_gemm launches of one step (conv_x6c_kernel / sinc_x6_fwd_kernel split-bf16 + "
                                                    "conv_gemm_kernel exact-fp32 instantiations; `by_pipe` prices each against its own "
                                                    "pipe; per-launch times include the launch's own operand packs)",
                         "achieved": round(cg_tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(cg_tf / peak, 4), "frac_of_fp32_mfma_peak": round(cg_tf / PEAK_F32_MFMA_TFLOPS, 4),
                         "peak_note": peak_note, "traffic": traffic,
                         "traffic_unit": "GB per launch (PMC FETCH_SIZE x2 + WRITE_SIZE); null when no PMC profile of "
                                         "exactly this build exists", "traffic_source": traffic_src,
                         "dominant": dominant,
                         "launches_per_step": cg["launches"] // extra,
                         "avg_launch_ms": round(cg["ms"] / cg["launches"], 4),
                         "algorithmic_gflop_per_launch": round(cg["flops"] / cg["launches"] / 1e9, 2),
                         "note": "sum of the launches' executed contraction FLOPs (2*S*Ncols*M*K from each "
                                 "descriptor) / sum of HIP-event durations on the launch stream"},
            "by_pipe": {"%s/%s" % k: {"launches_per_step": v["launches"] // extra, "ms_per_step": round(v["ms"] / extra, 3),
                                      "achieved": round(v["flops"] / v["ms"] / 1e9, 2), "unit": "TFLOP/s",
                                      "peak": round(PEAK_BF16_MFMA_TFLOPS / X6_MFMA_PER_PRODUCT if k[1] == "x6"
                                                    else PEAK_F32_MFMA_TFLOPS, 1),
                                      "frac": round(v["flops"] / v["ms"] / 1e9 /
                                                    (PEAK_BF16_MFMA_TFLOPS / X6_MFMA_PER_PRODUCT if k[1] == "x6"
                                                     else PEAK_F32_MFMA_TFLOPS), 4)}
                        for k, v in sorted(fams_pipe.items())},
            "roofline_wgrad": {"bound": "mfma", "kernel": "pase_wgrad_gemm launches (conv_x6c_kernel T-mode + wgrad_*_kernel fp32)",
                               "achieved": round(wg_tf, 2),
                               "peak": round(peak, 1), "unit": "TFLOP/s",
                               "frac": round(wg_tf / peak, 4),
                               "launches_per_step": wg["launches"] // extra,
                               "avg_launch_ms": round(wg["ms"] / wg["launches"], 4)},
            "roofline_step": {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1),
                              "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                              "frac_of_fp32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                              "note": "canonical algorithmic FLOPs (%.1f GFLOP per utterance, SURVEY 8d: minimal "
                                      "algorithm, dense skips pooled first) x utterances/s per GPU" % gflop_utt},
        }
        if h2d is not None:
            out["h2d"] = h2d
        if capped is not None:
            out["n_gt1_cap_cost"] = capped
        za = getattr(tr, "_zero_arena", None) or getattr(tr, "_graph_arena", None)
        if za is not None and za.buf is not None:
            out["config"]["zero_arena_MB"] = round(za.buf.numel() / 1e6, 1)
        if _lib.LIBRARY_OVERRIDE:
            out["library_override"] = _lib.LIBRARY_OVERRIDE      # an A/B build (PASE_LIB), not the shipped library
        if multi is not None:
            out["multi_gpu"] = multi
            out["config"]["rccl_max_nchannels"] = args.rccl_max_nchannels or "RCCL default"
            out["config"]["reserved_cus"] = tr.reserve_cus
        if shared and world > 1:
            out["invalid_as_scaling_number"] = ("%d ranks share %d GPU(s) over gloo: functional smoke of the N>1 path only"
                                                % (world, ndev))
        if world == 1 and not args.no_cpu_baseline:
            try:
                if args.cpu_baseline_b2:
                    out["cpu_baseline"] = cpu_baseline(raw, fe_cfg, seconds_budget=25.0, B=2, max_steps=20)
                elif args.cpu_baseline_bs32:
                    out["cpu_baseline"] = cpu_baseline(raw, fe_cfg, seconds_budget=600.0, B=B, max_steps=3)
                else:
                    out["cpu_baseline"] = cpu_baseline(raw, fe_cfg, B=B)
            except Exception as e:  # the baseline leg must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "utterances/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        if world == 1 and args.producer and not args.no_cpu_baseline:
            try:
                out["cpu_targets_baseline"] = cpu_targets_baseline(raw, T)
            except Exception as e:
                out["cpu_targets_baseline"] = {"value": None, "sample": "failed: %r" % (e,)}
        if world == 1 and not args.producer and not args.no_torch_gpu_baseline:
            try:
                del tr, batch
                torch.cuda.empty_cache()
                out["torch_rocm_baseline"] = torch_rocm_baseline(raw, fe_cfg, dev, B, T)
            except Exception as e:
                out["torch_rocm_baseline"] = {"value": None, "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

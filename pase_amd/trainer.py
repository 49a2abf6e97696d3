"""Mirror of pase/models/WorkerScheduler/{trainer,worker_scheduler,lr_scheduler}.py for the
`--backprop_mode base` path (README.md:129): one Adam + one LR_Scheduler + one Saver per module
(frontend + each worker: trainer.py:86-143), `_base_scheduler` = sum of weighted losses -> one
backward -> every optimizer steps (worker_scheduler.py:43-75).

MI355X-native differences (SURVEY.md section 8e -- the reference has NO multi-GPU path):
  * the step is the hand-scheduled `pase.loss_and_grads` (no autograd graph);
  * each logical optimizer owns ONE flat fp32 parameter / gradient / moment buffer, so its update is
    a single `pase_adam_step` launch and the data-parallel exchange is a handful of large RCCL
    all-reduces over the flat gradient buffers (one process per GPU; xGMI is point-to-point, so few
    large messages beat many small ones); worker-head gradients are complete before the encoder
    backward starts, so their all-reduce is issued on a side stream and overlaps it;
  * BatchNorm uses per-rank batch statistics (there is no SyncBN in the reference to match).
"""
import math
import os
import time

import torch
import torch.distributed as dist

from . import engine
from . import kernels as K
from .modules import Saver
from .pase import pase


class LR_Scheduler(object):
    """lr_scheduler.py:16-60: 'poly' lr*(1-T/N)^0.9, 'cos', 'step'; sets optimizer.param_groups[0]['lr']."""

    def __init__(self, mode, optim_name, base_lr, num_epochs, iters_per_epoch=0, lr_step=30, warmup_epochs=0):
        self.mode = mode
        self.name = optim_name
        self.lr = base_lr
        if mode == "step":
            assert lr_step
        self.lr_step = lr_step
        self.iters_per_epoch = iters_per_epoch
        self.N = num_epochs * iters_per_epoch
        self.epoch = -1
        self.warmup_iters = warmup_epochs * iters_per_epoch

    def __call__(self, optimizer, i, epoch, loss):
        T = epoch * self.iters_per_epoch + i
        if self.mode == "cos":
            lr = 0.5 * self.lr * (1 + math.cos(1.0 * T / self.N * math.pi))
        elif self.mode == "poly":
            lr = self.lr * pow((1 - 1.0 * T / self.N), 0.9)
        elif self.mode == "step":
            lr = self.lr * (0.1 ** (epoch // self.lr_step))
        else:
            raise NotImplementedError(self.mode)
        if self.warmup_iters > 0 and T < self.warmup_iters:
            lr = lr * 1.0 * T / self.warmup_iters
        self.epoch = epoch
        assert lr >= 0
        optimizer.param_groups[0]["lr"] = lr
        for g in optimizer.param_groups[1:]:
            g["lr"] = lr * 10
        return lr


class FusedAdam(object):
    """torch.optim.Adam (defaults) over one flat buffer.  Parameters are re-pointed to views of
    `flat_p` and their .grad to views of `flat_g`; state_dict() uses torch.optim.Adam's layout so
    reference-style `weights_*.ckpt` files stay interchangeable."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_buffer=None):
        self.params = [p for p in params]
        if len(self.params) == 0:
            raise ValueError("FusedAdam: no parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(n, device=dev)
        # grad_buffer: this optimizer's slice of a gradient arena shared by all optimizers (zeroed once per step)
        self.flat_g = torch.zeros(n, device=dev) if grad_buffer is None else grad_buffer
        assert self.flat_g.numel() == n
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        off = 0
        self.offsets = []
        for p in self.params:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = self.flat_g[off:off + k].view(p.shape)
            self.offsets.append((off, k))
            off += k
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_t = torch.full((1,), float(lr), device=dev)
        self._lr_host = float(lr)
        self.betas = betas
        self.eps = eps
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)]

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()

    def sync_lr(self):
        """host-side learning rate -> the device scalar the Adam kernel reads (outside any captured graph)"""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            self.lr_t.fill_(lr)
            self._lr_host = lr

    def step(self, grad_mul=1.0, tick=True):
        self.sync_lr()
        if tick:                       # (the trainer advances all 13 step counters in one multi-tensor launch)
            K.step_tick(self.step_t)
        K.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.lr_t, self.step_t,
                    beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, grad_mul=grad_mul)

    def state_dict(self):
        step = int(self.step_t.item())
        state = {}
        for i, (off, k) in enumerate(self.offsets):
            shp = self.params[i].shape
            state[i] = {"step": torch.tensor(float(step)),
                        "exp_avg": self.exp_avg[off:off + k].view(shp).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + k].view(shp).clone()}
        g = dict(self.param_groups[0])
        g["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        for i, (off, k) in enumerate(self.offsets):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_t.fill_(int(float(st["step"])))
        if sd.get("param_groups"):
            self.param_groups[0]["lr"] = sd["param_groups"][0].get("lr", self.param_groups[0]["lr"])


class trainer(object):
    """trainer.py:26-198 (constructor) + one fused training step.  `cfg` carries the train.py
    options the reference reads: epoch, batch_size, save_path, log_freq, bpe, va_bpe, fe_opt,
    fe_lr, min_opt, min_lr, lrdec_step, max_ckpts (train.py:338-451)."""

    def __init__(self, frontend=None, frontend_cfg=None, att_cfg=None, minions_cfg=None, cfg=None, cls_lst=[],
                 regr_lst=[], pretrained_ckpt=None, tensorboard=None, backprop_mode="base", lr_mode="step",
                 name="Pase_base", device=None):
        if att_cfg:
            raise NotImplementedError("pase_amd trainer: pase_attention")
        if backprop_mode != "base":
            raise NotImplementedError("pase_amd trainer: backprop_mode %r (PASE+ recipe uses 'base')" % backprop_mode)
        if len(cls_lst) == 0 and "cls" in minions_cfg:
            cls_lst = [w["name"] for w in minions_cfg["cls"]]
        if len(regr_lst) == 0 and "regr" in minions_cfg:
            regr_lst = [w["name"] for w in minions_cfg["regr"]]
        self.model = pase(frontend=frontend, frontend_cfg=frontend_cfg, minions_cfg=minions_cfg, cls_lst=cls_lst,
                          regr_lst=regr_lst, pretrained_ckpt=pretrained_ckpt, name=name)
        if device is not None:
            self.model.to(device)
        cfg = dict(cfg or {})
        self.cfg = cfg
        self.epoch = cfg.get("epoch", 1)
        self.bsize = cfg.get("batch_size", 32)
        self.save_path = cfg.get("save_path", "ckpt")
        self.log_freq = cfg.get("log_freq", 100)
        self.bpe = cfg.get("bpe", 1)
        self.va_bpe = cfg.get("va_bpe", 1)
        fe_opt, min_opt = cfg.get("fe_opt", "Adam"), cfg.get("min_opt", "Adam")
        if fe_opt.lower() != "adam" or min_opt.lower() != "adam":
            raise NotImplementedError("pase_amd trainer: only Adam (train.py:390-391 defaults)")
        fe_lr, min_lr = cfg.get("fe_lr", 0.001), cfg.get("min_lr", 0.0005)
        lrdec = cfg.get("lrdec_step", 30)
        max_ckpts = cfg.get("max_ckpts", 5)
        self.savers = []
        # ONE gradient arena for the 13 logical optimizers (frontend last, so the worker part -- final before the
        # encoder backward starts -- is one contiguous range): slices are 256-B aligned
        mods = list(self.model.classification_workers) + list(self.model.regression_workers) + [self.model.frontend]
        sizes = [sum(p.numel() for p in m.parameters()) for m in mods]
        offs, tot = [], 0
        for n_ in sizes:
            offs.append(tot)
            tot += (n_ + 63) // 64 * 64
        pdev = next(self.model.parameters()).device
        self.grad_arena = torch.zeros(tot, device=pdev)
        self._worker_grads = self.grad_arena[:offs[-1]]       # 12 worker buffers: ONE collective (87 MB at PASE+)
        self._frontend_grads = self.grad_arena[offs[-1]:]
        self._zero_arena = None
        gslice = {id(m): self.grad_arena[o:o + n_] for m, o, n_ in zip(mods, offs, sizes)}
        self.frontend_optim = FusedAdam(self.model.frontend.parameters(), lr=fe_lr,
                                        grad_buffer=gslice[id(self.model.frontend)])
        self.fe_scheduler = LR_Scheduler(lr_mode, lr_step=lrdec, optim_name="frontend", base_lr=fe_lr,
                                         num_epochs=self.epoch, iters_per_epoch=self.bpe)
        self.savers.append(Saver(self.model.frontend, self.save_path, max_ckpts=max_ckpts,
                                 optimizer=self.frontend_optim, prefix="PASE-"))
        self.cls_optim, self.cls_scheduler = {}, {}
        for worker in self.model.classification_workers:
            self.cls_optim[worker.name] = FusedAdam(worker.parameters(), lr=min_lr, grad_buffer=gslice[id(worker)])
            self.cls_scheduler[worker.name] = LR_Scheduler(lr_mode, lr_step=lrdec, optim_name=worker.name,
                                                           base_lr=min_lr, num_epochs=self.epoch,
                                                           iters_per_epoch=self.bpe)
            self.savers.append(Saver(worker, self.save_path, max_ckpts=max_ckpts,
                                     optimizer=self.cls_optim[worker.name], prefix="M-{}-".format(worker.name)))
        self.regr_optim, self.regr_scheduler = {}, {}
        for worker in self.model.regression_workers:
            self.regr_optim[worker.name] = FusedAdam(worker.parameters(), lr=min_lr, grad_buffer=gslice[id(worker)])
            self.regr_scheduler[worker.name] = LR_Scheduler(lr_mode, lr_step=lrdec, optim_name=worker.name,
                                                            base_lr=min_lr, num_epochs=self.epoch,
                                                            iters_per_epoch=self.bpe)
            self.savers.append(Saver(worker, self.save_path, max_ckpts=max_ckpts,
                                     optimizer=self.regr_optim[worker.name], prefix="M-{}-".format(worker.name)))
        self.epoch_beg = 0
        self.alphaSG = 1
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self._side = None
        self.device_targets = None
        # Data-parallel runs leave CUs to RCCL: every split-bf16 GEMM is a persistent grid of one 512-thread workgroup per
        # CU whose two waves per SIMD hold the whole register file (249 VGPRs, allocated as 256) -- a channel kernel cannot
        # share a CU with it and would only get one at a launch's tail.  cfg["reserve_cus"] (default 32 for world > 1, 0
        # otherwise) caps those grids at 256 - reserve_cus (PaseConvGemm::max_wg / PaseWgrad::max_wg).  32, not 16: workgroups
        # are dealt to the four shader engines of each XCD in turn, and with 30 workgroups per XCD two of its engines are
        # full -- a 64-workgroup kernel on another stream then waits for the GEMM's tail although 16 CUs idle (measured,
        # tools/experiments/side_probe.py: admitted at a cap of 224, not at 240); 28 per XCD leave one CU free in every engine.
        # The cap is in force only while collectives can be in flight -- from the hand-over of the first bucket (the workers'
        # gradients, before the encoder backward) to the join at the end of the step (_step_ddp); the forward and the heads'
        # backward run on all 256 CUs.  A single-process trainer with cfg reserve_cus > 0 keeps it on throughout.
        self.reserve_cus = int(self.cfg.get("reserve_cus", 32 if self.world > 1 else 0)) if hasattr(self, "cfg") else 0
        ncu = 256
        if pdev.type == "cuda":
            ncu = int(torch.cuda.get_device_properties(pdev).multi_processor_count)
        # the cap (workgroups) handed to the launches, 0 = none; it is an argument of loss_and_grads -> engine -> descriptor,
        # not process state: two trainers in one process, or an exception in a step, cannot leak it into other launches
        self._grid_cap = max(1, ncu - self.reserve_cus) if self.reserve_cus > 0 else 0
        if self.world > 1:
            self.broadcast_parameters()

    # --------------------------------------------------------------------------------------
    def optimizers(self):
        return list(self.cls_optim.values()) + list(self.regr_optim.values()) + [self.frontend_optim]

    def worker_optimizers(self):
        return list(self.cls_optim.values()) + list(self.regr_optim.values())

    def broadcast_parameters(self):
        """Rank 0's initial weights (and BN running stats) to every rank."""
        for opt in self.optimizers():
            dist.broadcast(opt.flat_p, src=0)
        for b in self.model.buffers():
            dist.broadcast(b, src=0)

    def _allreduce(self, buf):
        if buf.numel():
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    def use_device_targets(self, workers_cfg, hop=160, stats=None, device="cuda"):
        """Produce the LPS / FBanks / MFCC (+ZNorm) regression labels on the GPU from the clean chunk
        instead of the dataloader's host transforms (train.py:37-136).  `workers_cfg` is the RAW
        workers cfg (with the per-worker `transform` kwargs), `stats` the ZNorm statistics dict."""
        from .dsp import DeviceTargets
        self.device_targets = DeviceTargets(workers_cfg, hop=hop, stats=stats, device=device)
        return self.device_targets

    def _fill_targets(self, batch, device):
        missing = [n for n in self.device_targets.feats if n not in batch]
        if not missing:
            return batch
        clean = batch["cchunk"] if "cchunk" in batch else batch["chunk"]
        clean = clean.to(device if device is not None else self.device_targets.feats[missing[0]].device)
        batch = dict(batch)
        for n in missing:
            batch[n] = self.device_targets.feats[n](clean.contiguous().float())
        return batch

    def capture_step(self, example_batch):
        """Capture one training step (forward, losses, backward, 13 Adam updates: ~260 kernel launches on four
        streams) in a hipGraph for batches shaped like `example_batch` (device tensors).  train_step then copies the
        batch into the graph's static input buffers and replays: one graph launch per step instead of ~260 kernel
        launches from Python.  The learning rates and step counters are device scalars, so the captured graph stays
        valid as they change.  Single-GPU only (the data-parallel step issues its collectives eagerly)."""
        if self.world > 1:
            raise NotImplementedError("pase_amd trainer: hipGraph capture of the data-parallel step")
        # Workers that draw HOST randomness per step (SPC: random.choice of the anchor / future / past frames,
        # minions.py:614-628; Gap: np.random frame pairs, :680-681) turn those draws into slice offsets and H2D copies at
        # enqueue time: a replayed graph would reuse the frames drawn during capture for every step.  Refuse.
        from .minions import GapMinion, SPCMinion
        for w in self.model.classification_workers:
            m = getattr(w, "minion", w)
            if isinstance(m, (SPCMinion, GapMinion)) or type(w).__name__ in ("SPC", "Gap"):
                raise NotImplementedError(
                    "pase_amd trainer: worker %r draws its frames on the host every step; a captured step would freeze "
                    "them (capture is supported for the mi / cmi / regression workers, i.e. workers+.cfg)" % w.name)
        dev = self.grad_arena.device
        self._graph = None
        self._static = {k: v.detach().clone() for k, v in example_batch.items() if torch.is_tensor(v)}
        # the two warm-up steps and the capture run are REAL optimizer steps on the example batch: snapshot everything
        # they mutate (parameters, Adam moments and step counters, BatchNorm running statistics) and restore it afterwards,
        # so that capture_step() leaves the training state exactly where it found it
        snap_model = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        snap_opt = [(o.flat_p.clone(), o.exp_avg.clone(), o.exp_avg_sq.clone(), o.step_t.clone()) for o in self.optimizers()]
        cur = torch.cuda.current_stream(dev)
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):               # allocator / arena / caches reach their steady state
            for _ in range(2):
                self._eager_step(self._static, None)
        cur.wait_stream(warm)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._static_losses = self._eager_step(self._static, None)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            sd = self.model.state_dict()
            for k, v in snap_model.items():
                sd[k].copy_(v)
            for o, (fp, m1, m2, st) in zip(self.optimizers(), snap_opt):
                o.flat_p.copy_(fp)
                o.exp_avg.copy_(m1)
                o.exp_avg_sq.copy_(m2)
                o.step_t.copy_(st)
        # the graph holds raw pointers into the zero arena: the eager fallback (a batch of another shape) gets its own,
        # so a regrow there cannot free memory the graph still writes
        self._graph_arena = self._zero_arena
        self._zero_arena = None
        self._graph = graph
        return graph

    def train_step(self, batch, device=None):
        """_base_scheduler (worker_scheduler.py:43-75): zero grads, total = sum w*loss, backward,
        every optimizer steps.  Returns the loss dict (device scalars; no host sync)."""
        if getattr(self, "_graph", None) is not None and all(
                k in batch and batch[k].shape == v.shape for k, v in self._static.items()):
            for opt in self.optimizers():
                opt.sync_lr()
            for k, v in self._static.items():
                v.copy_(batch[k], non_blocking=True)
            self._graph.replay()
            return dict(self._static_losses)
        return self._eager_step(batch, device)

    def _eager_step(self, batch, device=None, local_only=False):
        """local_only (bench.py, N > 1): this rank's step with NO collective -- its weights then drift from the other
        ranks'; a measurement of the compute path under N-rank load, never part of training."""
        self.model.train()
        if self.device_targets is not None:
            batch = self._fill_targets(batch, device)
        self.grad_arena.zero_()                 # every optimizer's flat gradient buffer: one memset
        sink = engine.GradSink(direct=True)
        dev = self.grad_arena.device
        if dev.type == "cuda":
            if self._zero_arena is None:
                self._zero_arena = engine.ZeroArena()
            self._zero_arena.begin_step(dev)
            engine._ARENA = self._zero_arena
        try:
            if self.world > 1 and not local_only:
                losses = self._step_ddp(batch, sink, device)
            else:
                # (single process with cfg reserve_cus > 0: capped throughout, or -- cfg reserve_scope = "encoder_backward" --
                #  only where the data-parallel step caps it: what the reservation costs a step, measurable on one GPU)
                bwd_only = self.cfg.get("reserve_scope", "step") == "encoder_backward"
                losses = self.model.loss_and_grads(batch, sink, device, max_wg=0 if bwd_only else self._grid_cap,
                                                   encoder_backward_max_wg=self._grid_cap if bwd_only else None)
                self._step_all(1.0)
        finally:
            engine._ARENA = None
        return losses

    def _frontend_buckets(self):
        """Reverse-order buckets of the frontend gradient buffer: {tag: [(begin, end), ...]} element ranges that are
        final when engine.encoder_backward reports `tag` ("head" = W + dense skips + QRNN; then conv blocks from the
        last to the first).  Small early blocks are merged into the bucket of block 0 (one collective for the tail)."""
        if getattr(self, "_buckets", None) is not None:
            return self._buckets
        fe = self.model.frontend
        opt = self.frontend_optim
        where = {id(p): (off, off + k) for p, (off, k) in zip(opt.params, opt.offsets)}

        def ranges(params):
            r = sorted(where[id(p)] for p in params if id(p) in where)
            out = []
            for b, e in r:
                if out and out[-1][1] == b:
                    out[-1] = (out[-1][0], e)
                else:
                    out.append((b, e))
            return out
        nb = len(fe.blocks)
        block_params = [list(fe.blocks[n].parameters()) for n in range(nb)]
        in_blocks = set(id(p) for ps in block_params for p in ps)
        head = [p for p in opt.params if id(p) not in in_blocks]
        buckets = {"head": ranges(head)}
        small, merged = 1 << 20, []           # blocks under 4 MB of gradients ride with block 0
        for n in reversed(range(nb)):
            ps = block_params[n]
            if n > 0 and sum(p.numel() for p in ps) < small:
                merged += ps
                buckets[n] = []
            elif n == 0:
                buckets[0] = ranges(ps + merged)
            else:
                buckets[n] = ranges(ps)
        self._buckets = buckets
        return buckets

    def _step_ddp(self, batch, sink, device):
        """Data-parallel step: per-rank batch, sum-all-reduce of the flat gradient arena over RCCL/xGMI (mean via
        grad_mul = 1/world in the Adam kernel), bucketed in reverse-autograd order on a side stream:
          * the 12 worker buffers (87 MB of the 119 MB, one contiguous range) are final before the encoder backward
            starts: ONE collective underneath it;
          * the frontend buffer goes out as its groups complete -- head (W / dense skips / QRNN), then conv blocks
            7, 6, ... -- so only the last small bucket is exposed after the backward."""
        use_side = torch.cuda.is_available() and next(self.model.parameters()).is_cuda
        model = self.model
        if use_side and self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side
        buckets = self._frontend_buckets()
        fg = self._frontend_grads
        # comm_diag (bench.py --gpus N, tests): per bucket the moment it is handed over (main stream), when its collective
        # starts and ends (side stream), plus the end of the backward and the join -- see comm_report()
        diag = [] if (getattr(self, "comm_diag", False) and use_side) else None
        t_host0 = time.perf_counter()
        ev_begin = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True)) if diag is not None else None

        # GEMMs of the encoder backward -- everything enqueued after the workers' bucket is handed over -- leave CUs to the
        # collectives' kernels; the forward and the heads' backward keep the whole chip
        cap = self._grid_cap if use_side else 0

        def launch(tag, bufs):
            if not bufs:
                return
            if use_side:
                ready = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=diag is not None))
                side.wait_event(ready)
                with torch.cuda.stream(side):
                    if diag is not None:
                        e0 = side.record_event(torch.cuda.Event(enable_timing=True))
                    for b in bufs:
                        self._allreduce(b)
                    if diag is not None:
                        e1 = side.record_event(torch.cuda.Event(enable_timing=True))
                        diag.append((tag, sum(b.numel() for b in bufs) * 4, ready, e0, e1))
            else:                              # CPU (gloo tests): same buckets, issued in line
                for b in bufs:
                    self._allreduce(b)

        losses = model.loss_and_grads(
            batch, sink, device, before_encoder_backward=lambda: launch("workers", [self._worker_grads]),
            on_encoder_grads=lambda tag: launch(tag, [fg[b:e] for b, e in buckets.get(tag, [])]),
            encoder_backward_max_wg=cap)
        if use_side:
            if diag is not None:
                ev_bwd_end = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
            torch.cuda.current_stream().wait_stream(side)
            if diag is not None:
                ev_join = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
        self._step_all(1.0 / self.world)
        if diag is not None:
            self._comm_events = dict(begin=ev_begin, bwd_end=ev_bwd_end, join=ev_join, buckets=diag,
                                     host_enqueue_ms=(time.perf_counter() - t_host0) * 1e3)
        return losses

    def comm_report(self):
        """Timings of the most recent comm_diag step (synchronises): per bucket bytes / when it became ready / when its
        collective ran, relative to the step's first kernel (`ready_ms` is recorded on the stream that hands the bucket over:
        the main stream for the worker and head buckets, the weight-gradient side stream for the conv-block buckets --
        engine.encoder_backward joins that stream before it returns, which is what keeps ready_ms <= backward_end_ms); `comm_exposed_ms` = how long the main stream waited for the
        side stream after the last backward kernel; `comm_total_ms` = sum of the collectives' durations;
        `host_enqueue_ms` = wall time this rank's Python spent enqueueing the step (no device synchronisation inside)."""
        ev = getattr(self, "_comm_events", None)
        if ev is None:
            return None
        torch.cuda.synchronize()
        t0 = ev["begin"]
        rows = [dict(bucket=str(tag), MB=round(nb / 1e6, 2), ready_ms=round(t0.elapsed_time(rdy), 3),
                     start_ms=round(t0.elapsed_time(e0), 3), end_ms=round(t0.elapsed_time(e1), 3))
                for tag, nb, rdy, e0, e1 in ev["buckets"]]
        return dict(buckets=rows, backward_end_ms=round(t0.elapsed_time(ev["bwd_end"]), 3),
                    comm_exposed_ms=round(ev["bwd_end"].elapsed_time(ev["join"]), 3),
                    comm_total_ms=round(sum(r["end_ms"] - r["start_ms"] for r in rows), 3),
                    host_enqueue_ms=round(ev["host_enqueue_ms"], 3))

    def _step_all(self, grad_mul):
        opts = self.optimizers()
        torch._foreach_add_([o.step_t for o in opts], 1)
        for opt in opts:
            opt.step(grad_mul=grad_mul, tick=False)

    def adjust_lr(self, bidx, epoch, losses=None):
        """trainer.py:245-254 (called every log_freq iterations in the reference)."""
        lrs = {"frontend": self.fe_scheduler(self.frontend_optim, bidx, epoch, 0.0)}
        for name, sch in self.cls_scheduler.items():
            lrs[name] = sch(self.cls_optim[name], bidx, epoch, 0.0)
        for name, sch in self.regr_scheduler.items():
            lrs[name] = sch(self.regr_optim[name], bidx, epoch, 0.0)
        return lrs

    def save_epoch(self, e, step):
        """trainer.py:263-272: FE_e{e}.ckpt (bare frontend state_dict, what load_pretrained
        consumes) + one rotating Saver checkpoint per module."""
        if self.world > 1:
            # BN running statistics are per rank (local batch statistics): checkpoint rank 0's view of the model,
            # written by rank 0 only -- concurrent writers would tear the files and double-rotate the index
            for b in self.model.buffers():
                dist.broadcast(b, src=0)
        if self.rank == 0:
            os.makedirs(self.save_path, exist_ok=True)
            torch.save(self.model.frontend.state_dict(), os.path.join(self.save_path, "FE_e{}.ckpt".format(e)))
            for saver in self.savers:
                saver.save(saver.prefix[:-1], step)
        if self.world > 1:
            dist.barrier()

    def resume_training(self, device=None):
        """trainer.py:339-363: every Saver's latest checkpoint, equal steps, epoch_beg = step // bpe."""
        if self.world > 1:
            dist.barrier()          # rank 0 has finished writing; every rank then loads the same files
        steps = []
        for saver in self.savers:
            cur = saver.read_latest_checkpoint()
            if cur is None:
                return False
            steps.append(saver.load_ckpt_step(cur))
            saver.load_weights()
        if len(set(steps)) != 1:
            raise ValueError("checkpoints at different steps: %r" % steps)
        self.epoch_beg = steps[0] // self.bpe
        return True

    def _eval(self, dataloader, epoch=0, device=None):
        """Validation pass of trainer.py:282-337: `va_bpe` batches through the model in eval mode (BatchNorm running
        statistics, no gradients), every worker's loss UNWEIGHTED as the reference accumulates it (tot_loss += loss, :307,
        :316), running per-worker lists, mean per epoch.  Returns {worker: mean loss, 'total': mean} and keeps it in
        `self.last_eval` (the reference hands the lists to its tensorboard eval_logger)."""
        was_training = self.model.training
        self.model.eval()
        running = {}
        it = iter(dataloader)
        dev = torch.device(device) if device is not None else next(self.model.parameters()).device
        try:
            with torch.no_grad():
                for _bidx in range(1, self.va_bpe + 1):
                    try:
                        batch = next(it)
                    except StopIteration:
                        it = iter(dataloader)
                        batch = next(it)
                    if self.device_targets is not None:
                        batch = self._fill_targets(batch, dev)
                    h, chunk, preds, labels = self.model(batch, device=dev)
                    tot = 0.0
                    for worker in list(self.model.classification_workers) + list(self.model.regression_workers):
                        loss = float(worker.loss(preds[worker.name], labels[worker.name]))
                        running.setdefault(worker.name, []).append(loss)
                        tot += loss
                    running.setdefault("total", []).append(tot)
        finally:
            self.model.train(was_training)
        self.last_eval = {k: sum(v) / len(v) for k, v in running.items()}
        if self.rank == 0:
            print("EVAL epoch {}: ".format(epoch) + " ".join("{}={:.4f}".format(k, v) for k, v in self.last_eval.items()))
        return self.last_eval

    def train_(self, dataloader, valid_dataloader=None, device=None):
        """Epoch loop of trainer.py:200-278 without the tqdm / tensorboard / aux-supervisor side
        channels (out of scope)."""
        state = {"it": iter(dataloader)}

        def host_batch():
            try:
                return next(state["it"])
            except StopIteration:
                state["it"] = iter(dataloader)
                return next(state["it"])

        first = host_batch()
        dev = torch.device(device) if device is not None else next(self.model.parameters()).device
        feeder = None
        if dev.type == "cuda" and any(torch.is_tensor(v) and not v.is_cuda for v in first.values()):
            # host-side dataloader: copies run on a copy stream one step ahead (pinned ring, double-buffered slots)
            from .producer import PinnedBatchFeeder
            pending = [first]
            feeder = PinnedBatchFeeder(lambda: pending.pop() if pending else host_batch(), dev)
        for e in range(self.epoch_beg, self.epoch):
            for bidx in range(1, self.bpe + 1):
                if feeder is not None:
                    batch = feeder.next()
                elif first is not None:
                    batch, first = first, None
                else:
                    batch = host_batch()
                losses = self.train_step(batch, device)
                if bidx % self.log_freq == 0 or bidx >= self.bpe:
                    lrs = self.adjust_lr(bidx, e, losses)
                    if self.rank == 0:
                        print("epoch {} batch {}/{}: ".format(e, bidx, self.bpe) +
                              " ".join("{}={:.4f}".format(k, float(v)) for k, v in losses.items()) +
                              " lr_fe={:.6f}".format(lrs["frontend"]))
            if valid_dataloader is not None:
                self._eval(valid_dataloader, epoch=e, device=device)
            self.save_epoch(e, e * self.bpe + bidx)

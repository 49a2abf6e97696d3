"""Mirror of pase/models/pase.py:241-356 (class `pase`): encoder + regression / contrastive
workers, `forward(batch_dict, alpha, device) -> (h, chunk, preds, labels)`; plus the fused
forward+loss+backward schedule used by the trainer (`loss_and_grads`).
"""
import torch
import torch.nn as nn

from . import engine
from .engine import Act
from .frontend import wf_builder
from .minions import cls_worker_maker, make_samples, minion_maker
from .modules import Model


class _LossBook(object):
    """Per-step loss bookkeeping of the fused schedule: ONE float64 vector holds every worker's loss sum (each fused
    loss kernel accumulates into its slot), finalize() applies loss_weight / numel and adds the total."""
    _scale_cache = {}

    def __init__(self, names, like):
        self.names = list(names)
        self.index = {n: i for i, n in enumerate(self.names)}
        self.acc = engine._zeros((len(self.names),), like, torch.float64)
        self.scales = [0.0] * len(self.names)

    def slot(self, name):
        i = self.index[name]
        return self.acc[i:i + 1]

    def scale(self, name, s):
        self.scales[self.index[name]] = float(s)

    def finalize(self):
        key = (tuple(self.scales), str(self.acc.device))
        sc = self._scale_cache.get(key)
        if sc is None:                       # constant per model / batch shape: uploaded once
            sc = torch.tensor(self.scales, dtype=torch.float64, device=self.acc.device)
            if len(self._scale_cache) > 64:
                self._scale_cache.clear()
            self._scale_cache[key] = sc
        vec = self.acc * sc
        losses = {n: vec[i] for i, n in enumerate(self.names)}
        losses["total"] = vec.sum()
        return losses


class pase(Model):
    def __init__(self, frontend=None, frontend_cfg=None, minions_cfg=None, cls_lst=["mi", "cmi", "spc"],
                 regr_lst=["chunk", "lps", "mfcc", "prosody"], pretrained_ckpt=None, name="adversarial"):
        super().__init__(name=name)
        if minions_cfg is None or len(minions_cfg) < 1:
            raise ValueError("Please specify a stack of minions config with at least 1 minion. "
                             "GIMME SOMETHING TO DO.")
        print("pase config ==>", frontend_cfg)
        self.frontend = wf_builder(frontend_cfg)
        self.cls_lst = cls_lst
        self.reg_lst = regr_lst
        ninp = self.frontend.emb_dim
        self.regression_workers = nn.ModuleList()
        self.classification_workers = nn.ModuleList()
        self.regularizer_workers = []
        self.fwd_cchunk = False
        if "concat" in frontend_cfg.keys():
            raise NotImplementedError("pase_amd pase: 'concat' frontend cfgs")
        print("==>concat features from {} levels".format(1))
        print("==>input size for workers: {}".format(ninp))
        for type, cfg_lst in minions_cfg.items():
            for cfg in cfg_lst:
                if type == "cls":
                    cfg["num_inputs"] = ninp
                    self.classification_workers.append(cls_worker_maker(cfg, ninp))
                elif type == "regr":
                    cfg["num_inputs"] = ninp
                    self.regression_workers.append(minion_maker(cfg))
                elif type == "regu":
                    raise NotImplementedError("pase_amd pase: regularizer workers")
        if pretrained_ckpt is not None:
            self.load_pretrained(pretrained_ckpt, load_last=True)

    # ------------------------------------------------------------------------------------------
    # API-compatible forward (predictions materialised, autograd-capable)
    # ------------------------------------------------------------------------------------------
    def forward(self, x, alpha=1, device=None):
        x_ = dict((k, v) for k, v in x.items())
        if not self.fwd_cchunk:
            x_.pop("cchunk", None)
        h = self.frontend(x_, device)
        if len(h) > 1:
            assert len(h) == 2, len(h)
            h, chunk = h
        preds, labels = {}, {}
        for worker in self.regression_workers:
            preds[worker.name] = worker(chunk, alpha)
            labels[worker.name] = x[worker.name].to(device).detach()
        for worker in self.classification_workers:
            if worker.name == "spc" or worker.name == "gap":
                y, label = worker(chunk, alpha, device=device)
            else:
                y, label = worker(h, alpha, device=device)
            preds[worker.name] = y
            labels[worker.name] = label
        return h, chunk, preds, labels

    def _cls_step(self, emb, B, demb, sink, book, max_wg=0):
        """Forward + loss + backward of the contrastive / classification workers (pase.py:345-354); their
        gradient w.r.t. the embeddings is accumulated into `demb`, their loss sums into `book`."""
        E, F_ = emb.shape[1], emb.shape[2]
        chunk = emb[:B]
        h = (emb[:B], emb[B:2 * B], emb[2 * B:3 * B])
        for worker in self.classification_workers:
            mn = worker.minion
            loss = worker.loss
            if worker.name == "spc":
                # SPC reads the chunk embedding only (pase.py:346-347); frames are gathered / the
                # gradient scattered back with index plumbing, the MLP runs on the kernels
                t, ft, pt = mn.sample(F_)
                N = mn.ctxt_frames
                xin = mn.gather(chunk, t, ft, pt).contiguous()
                nb = xin.shape[0]
                label = torch.cat((torch.ones(nb // 2, 1, 1, device=emb.device),
                                   torch.zeros(nb // 2, 1, 1, device=emb.device)), dim=0)
                wctx = engine.worker_forward(list(mn.blocks), mn.W, Act(xin, C=xin.shape[1]),
                                             loss=dict(name=loss.loss_name, r=loss.r, target=label,
                                                       weight=worker.loss_weight, acc=book.slot(worker.name)),
                                             want_pred=False, max_wg=max_wg)
                dsrc = engine.worker_backward(list(mn.blocks), mn.W, wctx, wctx.dpred, sink, max_wg=max_wg)
                dx = dsrc.dense(xin.shape[1], 1)[:, :, 0]
                pos, neg = dx[:B], dx[B:]
                demb[:B, :, t] += pos[:, :E] + neg[:, :E]
                demb[:B, :, ft:ft + N] += pos[:, E:].reshape(B, E, N)
                demb[:B, :, pt - N:pt] += neg[:, E:].reshape(B, E, N)
                book.scale(worker.name, worker.loss_weight / wctx.numel)
                del wctx, dsrc
                continue
            if worker.name == "gap":
                # Gap reads two frames of the chunk embedding per item (pase.py:346-347, minions.py:672-689)
                aidx, bidx = mn.sample(B, F_)
                xin = mn.gather(chunk, aidx, bidx).contiguous()
                label = mn.labels(aidx, bidx, F_, emb.device)
                wctx = engine.worker_forward(list(mn.blocks), mn.W, Act(xin, C=xin.shape[1]),
                                             loss=dict(name=loss.loss_name, r=loss.r, target=label,
                                                       weight=worker.loss_weight, acc=book.slot(worker.name)),
                                             want_pred=False, max_wg=max_wg)
                dsrc = engine.worker_backward(list(mn.blocks), mn.W, wctx, wctx.dpred, sink, max_wg=max_wg)
                dx = dsrc.dense(xin.shape[1], 1)[:, :, 0]
                ar = torch.arange(B, device=emb.device)
                dv = demb[:B]          # (i, :, a_i) is unique per item; a_i == b_i is handled by the two statements
                dv[ar, :, torch.as_tensor(aidx, device=emb.device)] += dx[:, :E]
                dv[ar, :, torch.as_tensor(bidx, device=emb.device)] += dx[:, E:]
                book.scale(worker.name, worker.loss_weight / wctx.numel)
                del wctx, dsrc
                continue
            x_pos, x_neg = make_samples(h, worker.augment)
            xin = torch.cat((x_pos, x_neg), dim=0)
            nb = xin.shape[0]
            if worker.time_mean:
                xm = torch.empty(nb, 2 * E, 1, device=emb.device)
                engine.K.bn_act_pool(xin, xm, None, None, None, S=nb, C_=2 * E, T=F_, F=1, d=F_, o_ctot=2 * E,
                                     o_coff=0)
                win = xm
            else:
                win = xin
            Tw = win.shape[2]
            label = torch.cat((torch.ones(nb // 2, 1, Tw, device=emb.device),
                               torch.zeros(nb // 2, 1, Tw, device=emb.device)), dim=0)
            wctx = engine.worker_forward(list(mn.blocks), mn.W, Act(win, C=2 * E),
                                         loss=dict(name=loss.loss_name, r=loss.r, target=label,
                                                   weight=worker.loss_weight, acc=book.slot(worker.name)),
                                         want_pred=False, max_wg=max_wg)
            dsrc = engine.worker_backward(list(mn.blocks), mn.W, wctx, wctx.dpred, sink, max_wg=max_wg)
            dx = dsrc.dense(2 * E, Tw)
            if worker.time_mean:
                dx = (dx / F_).expand(nb, 2 * E, F_)
            # scatter back through make_samples (cls_minions.py:29-43)
            half = nb // 2
            pos, neg = dx[:half], dx[half:]
            if worker.augment:
                q = half // 2
                # pos = [h0|h1 ; h1|h0], neg = [h0|h2 ; h1|h2]
                demb[:B] += pos[:q, :E] + pos[q:, E:] + neg[:q, :E]
                demb[B:2 * B] += pos[:q, E:] + pos[q:, :E] + neg[q:, :E]
                demb[2 * B:] += neg[:q, E:] + neg[q:, E:]
            else:
                demb[:B] += pos[:, :E] + neg[:, :E]
                demb[B:2 * B] += pos[:, E:]
                demb[2 * B:] += neg[:, E:]
            book.scale(worker.name, worker.loss_weight / wctx.numel)
            del wctx, dsrc

    # ------------------------------------------------------------------------------------------
    # fused training schedule: forward + all losses + backward in one hand-scheduled pass
    # (what trainer.train_ -> model.forward -> backprop_scheduler._base_scheduler do through
    # autograd in the reference: trainer.py:229-232, worker_scheduler.py:43-75)
    # ------------------------------------------------------------------------------------------
    def loss_and_grads(self, batch, sink=None, device=None, before_encoder_backward=None, on_encoder_grads=None,
                       max_wg=0, encoder_backward_max_wg=None):
        """Returns {worker: loss_weight*loss, 'total': sum} (0-dim float64 device tensors) and
        accumulates every parameter gradient into `sink` (default: param.grad).
        max_wg: cap on the persistent grids of the step's split-bf16 GEMM launches (0 = one workgroup per CU);
        encoder_backward_max_wg: the cap for the encoder backward alone -- the data-parallel trainer leaves CUs to the
        collectives that are in flight from `before_encoder_backward` on.  Both travel down the call path into each launch
        descriptor (PaseConvGemm::max_wg / PaseWgrad::max_wg); there is no process-wide setting."""
        if sink is None:
            sink = engine.GradSink(direct=True)
        fe = self.frontend
        keys = [k for k in ("chunk", "chunk_ctxt", "chunk_rand") if k in batch]
        if len(keys) != 3:
            raise ValueError("pase_amd: the fused step needs chunk / chunk_ctxt / chunk_rand")
        x = torch.cat([batch[k] for k in keys], dim=0)
        if device is not None:
            x = x.to(device)
        emb, ectx = engine.encoder_forward(fe, x, training=fe.training, max_wg=max_wg)
        B = batch["chunk"].shape[0]
        E, F_ = emb.shape[1], emb.shape[2]
        demb = torch.zeros_like(emb)
        chunk = emb[:B]
        from .minions import MLPMinion
        group = [w for w in self.regression_workers
                 if isinstance(w, MLPMinion) and len(w.blocks) == 1 and w.blocks[0].context == 1
                 and w.W.kernel_size[0] == 1 and w.W.out_channels > 1]
        if len(group) <= 1:
            group = []
        # one float64 slot per worker for its loss sum (one zero-fill, one scale, one reduction per step instead of a
        # fill + multiply + add per worker), reported in the order regression-group, other regression, contrastive
        others = [w for w in self.regression_workers if not any(w is g for g in group)]
        book = _LossBook([w.name for w in group] + [w.name for w in others] +
                         [w.name for w in self.classification_workers], emb)
        # The contrastive workers are a few dozen 50-100-workgroup launches: they run on a side HIP stream underneath
        # the regression workers (own gradient buffer, merged after the join) instead of serialising behind them.
        side = engine.side_streams(emb, 4)
        cls_stream = side[3] if (side and engine.K.GEMM_TIMER is None and len(self.classification_workers) > 0) else None
        if cls_stream is not None:
            main = torch.cuda.current_stream()
            cls_stream.wait_event(main.record_event())
            with torch.cuda.stream(cls_stream):
                demb_cls = torch.zeros_like(emb)
                self._cls_step(emb, B, demb_cls, sink, book, max_wg)
        # one-hidden-layer MLP workers share their input: run their first layers stacked
        if group:
            tg = {w.name: (batch[w.name].to(device) if device is not None else batch[w.name]) for w in group}
            res, dx = engine.mlp_group_step(group, Act(chunk, C=E), tg, sink, accs={w.name: book.slot(w.name) for w in group},
                                             max_wg=max_wg)
            demb[:B] += dx
            for w in group:
                book.scale(w.name, w.loss_weight / res[w.name][1])
        for worker in others:
            loss = worker.loss
            tgt = batch[worker.name]
            if device is not None:
                tgt = tgt.to(device)
            wctx = engine.worker_forward(list(worker.blocks), worker.W, Act(chunk, C=E),
                                         loss=dict(name=loss.loss_name, r=loss.r, target=tgt,
                                                   weight=worker.loss_weight, acc=book.slot(worker.name)),
                                         want_pred=False, max_wg=max_wg, sink=sink)
            dsrc = engine.worker_backward(list(worker.blocks), worker.W, wctx, wctx.dpred, sink, max_wg=max_wg)
            demb[:B] += dsrc.dense(E, F_)
            book.scale(worker.name, worker.loss_weight / wctx.numel)
            del wctx, dsrc
        if cls_stream is not None:
            main.wait_event(cls_stream.record_event())
            demb += demb_cls
        else:
            self._cls_step(emb, B, demb, sink, book, max_wg)
        losses = book.finalize()
        if before_encoder_backward is not None:
            before_encoder_backward()   # all worker-head gradients are final here (DDP overlap point)
        engine.encoder_backward(fe, ectx, demb, sink, on_ready=on_encoder_grads,
                                max_wg=max_wg if encoder_backward_max_wg is None else encoder_backward_max_wg)
        return losses

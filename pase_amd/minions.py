"""Mirror of pase/models/Minions/{minions,cls_minions}.py for the workers the PASE / PASE+ configs
use: MLPMinion (:452-528), DecoderMinion (:365-449), LIM "mi" / GIM "cmi" (cls_minions.py:53-99),
minion_maker (:11-35), cls_worker_maker (cls_minions.py:10-27).  Parameter containers with the
reference's state_dict; arithmetic on the HIP kernels via pase_amd.engine.
"""
import json
import random

import numpy as np
import torch
import torch.nn as nn

from . import engine
from .engine import Act
from .modules import GDeconv1DBlock, MLPBlock, Model


class _WorkerFn(torch.autograd.Function):
    """autograd bridge for a whole Minion: forward = engine.worker_forward (prediction
    materialised), backward = engine.worker_backward."""

    @staticmethod
    def forward(ctx, mod, x, *params):
        x = x.contiguous()
        wctx = engine.worker_forward(list(mod.blocks), mod.W, Act(x, C=x.shape[1]), loss=None, want_pred=True)
        ctx.mod, ctx.wctx, ctx.params = mod, wctx, params
        ctx.in_shape = x.shape
        return wctx.pred

    @staticmethod
    def backward(ctx, dpred):
        sink = engine.GradSink(direct=False)
        need_dx = ctx.needs_input_grad[1]
        dsrc = engine.worker_backward(list(ctx.mod.blocks), ctx.mod.W, ctx.wctx, dpred.contiguous(), sink,
                                      need_dinput=need_dx)
        dx = dsrc.dense(ctx.in_shape[1], ctx.in_shape[2]).contiguous() if need_dx else None
        grads = tuple(sink.get(p) if ctx.needs_input_grad[2 + i] else None for i, p in enumerate(ctx.params))
        ctx.wctx = None
        return (None, dx) + grads


class _MinionBase(Model):
    def _run(self, x):
        params = [p for p in nn.Module.parameters(self) if p.requires_grad]
        if torch.is_grad_enabled() and (x.requires_grad or len(params) > 0):
            return _WorkerFn.apply(self, x, *params)
        wctx = engine.worker_forward(list(self.blocks), self.W, Act(x.contiguous(), C=x.shape[1]), loss=None)
        return wctx.pred


class MLPMinion(_MinionBase):
    """[Conv1d(ninp, hidden, context) -> PReLU] x hidden_layers -> Conv1d(hidden, num_outputs*r, 1)."""

    def __init__(self, num_inputs, num_outputs, dropout, dropout_time=0.0, hidden_size=256, dropin=0.0,
                 hidden_layers=2, context=1, tie_context_weights=False, skip=True, loss=None, loss_weight=1.,
                 keys=None, augment=False, r=1, name="MLPMinion", **_unused):
        super().__init__(name=name)
        assert context % 2 != 0, context
        if dropout or dropout_time or dropin or tie_context_weights:
            raise NotImplementedError("pase_amd MLPMinion: dropout / tied context weights")
        self.num_inputs = num_inputs
        self.context = context
        self.skip = skip
        self.hidden_size = hidden_size
        self.hidden_layers = hidden_layers
        self.loss = loss
        self.loss_weight = loss_weight
        self.keys = keys
        self.r = r
        self.num_outputs = num_outputs * r
        self.blocks = nn.ModuleList()
        ninp = num_inputs
        for _ in range(hidden_layers):
            self.blocks.append(MLPBlock(ninp, hidden_size, context=context))
            ninp = hidden_size
            context = 1
        self.W = nn.Conv1d(ninp, self.num_outputs, context, padding=context // 2)

    def forward(self, x, alpha=1, device=None):
        y = self._run(x)
        if self.skip:
            raise NotImplementedError("pase_amd MLPMinion: skip=True (returns hidden activations)")
        return y


class DecoderMinion(_MinionBase):
    """GDeconv1DBlock x len(fmaps) -> MLPBlock x hidden_layers -> Conv1d(hidden, num_outputs, 1)."""

    def __init__(self, num_inputs, num_outputs, dropout, dropout_time=0.0, shuffle=False, shuffle_depth=7,
                 hidden_size=256, hidden_layers=2, fmaps=[256, 256, 128, 128, 128, 64, 64],
                 strides=[2, 2, 2, 2, 2, 5], kwidths=[2, 2, 2, 2, 2, 5], norm_type=None, skip=False, loss=None,
                 loss_weight=1., keys=None, name="DecoderMinion"):
        super().__init__(name=name)
        if dropout or dropout_time or shuffle or norm_type is not None:
            raise NotImplementedError("pase_amd DecoderMinion: dropout / shuffle / norm")
        self.num_inputs = num_inputs
        self.num_outputs = num_outputs
        self.skip = skip
        self.hidden_size = hidden_size
        self.hidden_layers = hidden_layers
        self.fmaps, self.strides, self.kwidths = fmaps, strides, kwidths
        self.loss = loss
        self.loss_weight = loss_weight
        self.keys = keys
        self.blocks = nn.ModuleList()
        ninp = num_inputs
        for fmap, kw, stride in zip(fmaps, kwidths, strides):
            self.blocks.append(GDeconv1DBlock(ninp, fmap, kw, stride, norm_type=norm_type))
            ninp = fmap
        for _ in range(hidden_layers):
            self.blocks.append(MLPBlock(ninp, hidden_size))
            ninp = hidden_size
        self.W = nn.Conv1d(hidden_size, num_outputs, 1)

    def forward(self, x, alpha=1, device=None):
        y = self._run(x)
        if self.skip:
            raise NotImplementedError("pase_amd DecoderMinion: skip=True")
        return y


class SPCMinion(MLPMinion):
    """Sequence-predicting-coding worker (Minions/minions.py:575-649): one random anchor frame t per
    batch (Python `random.choice`, same RNG stream as the reference), the next `ctxt_frames` frames
    at a random future offset (positive) / a random past window (negative), flattened to
    (2B, (ctxt_frames+1)*emb, 1) and pushed through the MLP."""

    def __init__(self, num_inputs, num_outputs, dropout, hidden_size=256, hidden_layers=2, ctxt_frames=5,
                 seq_pad=16, skip=True, loss=None, loss_weight=1., keys=None, name="SPCMinion"):
        super().__init__(num_inputs=(ctxt_frames + 1) * num_inputs, num_outputs=num_outputs, dropout=dropout,
                         hidden_size=hidden_size, hidden_layers=hidden_layers, skip=skip, loss=loss,
                         loss_weight=loss_weight, keys=keys, name=name)
        self.ctxt_frames = ctxt_frames
        self.seq_pad = seq_pad

    def sample(self, T):
        """(t, future_t, past_t): minions.py:614-628, three random.choice draws in this order."""
        N, M = self.ctxt_frames, self.seq_pad + self.ctxt_frames
        t = random.choice(list(range(M + 1, T - M)))
        future_t = random.choice(list(range(t + self.seq_pad, T - N)))
        past_t = random.choice(list(range(N, t - self.seq_pad)))
        return t, future_t, past_t

    def gather(self, x, t, future_t, past_t):
        bsz, N = x.size(0), self.ctxt_frames
        future = x[:, :, future_t:future_t + N].contiguous().view(bsz, -1)
        past = x[:, :, past_t - N:past_t].contiguous().view(bsz, -1)
        current = x[:, :, t].contiguous()
        pos = torch.cat((current, future), dim=1)
        neg = torch.cat((current, past), dim=1)
        return torch.cat((pos, neg), dim=0).unsqueeze(2)

    def forward(self, x, alpha=1, device=None):
        x_full = self.gather(x, *self.sample(x.size(2)))
        y = self._run(x_full)
        if self.skip:
            raise NotImplementedError("pase_amd SPCMinion: skip=True")
        return y


class GapMinion(MLPMinion):
    """Gap worker (Minions/minions.py:651-704): two random frames per batch item (np.random.randint, same
    draws as the reference), concatenated to (B, 2*emb, 1) and pushed through the MLP to regress their
    normalised distance.  The reference builds the label with torch.LongTensor(dists) (:693), which
    truncates |a-b|/(T-1) to 0 (or 1 when the frames are the two ends); that is mirrored."""

    def sample(self, B, T):
        aidx = np.random.randint(0, T, size=B)
        bidx = np.random.randint(0, T, size=B)
        return aidx, bidx

    @staticmethod
    def labels(aidx, bidx, T, device):
        d = np.abs(aidx - bidx) / float(T - 1)
        return torch.as_tensor(np.trunc(d), dtype=torch.float32, device=device).view(-1, 1, 1)

    @staticmethod
    def gather(x, aidx, bidx):
        ar = torch.arange(x.size(0), device=x.device)
        xa = x[ar, :, torch.as_tensor(aidx, device=x.device)]
        xb = x[ar, :, torch.as_tensor(bidx, device=x.device)]
        return torch.cat((xa, xb), dim=1).unsqueeze(2)

    def forward(self, x, alpha=1, device=None):
        aidx, bidx = self.sample(x.size(0), x.size(2))
        y = self._run(self.gather(x, aidx, bidx).contiguous())
        if self.skip:
            raise NotImplementedError("pase_amd GapMinion: skip=True")
        return y, self.labels(aidx, bidx, x.size(2), x.device)


def minion_maker(cfg):
    """minions.py:11-35 (mlp / decoder; the wavernn / spc / gap / gru / regularizer types are not
    reachable from cfg/workers/workers+.cfg)."""
    if isinstance(cfg, str):
        with open(cfg, "r") as f:
            cfg = json.load(f)
    print("=" * 50)
    print("name", cfg["name"])
    print("=" * 50)
    mtype = cfg.pop("type", "mlp")
    if mtype == "mlp":
        return MLPMinion(**cfg)
    if mtype == "decoder":
        return DecoderMinion(**cfg)
    if mtype == "spc":
        return SPCMinion(**cfg)
    if mtype == "gap":
        return GapMinion(**cfg)
    raise NotImplementedError("pase_amd minion_maker: minion type {}".format(mtype))


def make_samples(x, augment):
    """cls_minions.py:29-43."""
    x_pos = torch.cat((x[0], x[1]), dim=1)
    x_neg = torch.cat((x[0], x[2]), dim=1)
    if augment:
        x_pos = torch.cat((x_pos, torch.cat((x[1], x[0]), dim=1)), dim=0)
        x_neg = torch.cat((x_neg, torch.cat((x[1], x[2]), dim=1)), dim=0)
    return x_pos, x_neg


def make_labels(y):
    """cls_minions.py:47-51 (built on the prediction's device instead of CPU + .to(device))."""
    bsz, slen = y.size(0) // 2, y.size(2)
    return torch.cat((torch.ones(bsz, 1, slen, device=y.device), torch.zeros(bsz, 1, slen, device=y.device)), dim=0)


class LIM(Model):
    """Local info-max worker "mi" (cls_minions.py:53-74)."""

    def __init__(self, cfg, emb_dim):
        super().__init__(name=cfg["name"])
        cfg["num_inputs"] = 2 * emb_dim
        self.augment = cfg.get("augment", False)
        self.minion = minion_maker(cfg)
        self.loss = self.minion.loss
        self.loss_weight = self.minion.loss_weight
        self.time_mean = False

    def forward(self, x, alpha=1, device=None):
        x_pos, x_neg = make_samples(x, self.augment)
        x = torch.cat((x_pos, x_neg), dim=0).to(device)
        if self.time_mean:
            x = torch.mean(x, dim=2, keepdim=True)
        y = self.minion(x, alpha)
        return y, make_labels(y).to(device)


class GIM(LIM):
    """Global info-max worker "cmi": as LIM on the time-averaged embedding (cls_minions.py:76-99)."""

    def __init__(self, cfg, emb_dim):
        super().__init__(cfg, emb_dim)
        self.time_mean = True


class SPC(Model):
    """cls_minions.py:101-115: SPCMinion + make_labels."""

    def __init__(self, cfg, emb_dim):
        super().__init__(name=cfg["name"])
        cfg["num_inputs"] = emb_dim
        self.minion = minion_maker(cfg)
        self.loss = self.minion.loss
        self.loss_weight = self.minion.loss_weight

    def forward(self, x, alpha=1, device=None):
        y = self.minion(x, alpha)
        return y, make_labels(y).to(device)


class Gap(Model):
    """cls_minions.py:117-131."""

    def __init__(self, cfg, emb_dim):
        super().__init__(name=cfg["name"])
        cfg["num_inputs"] = 2 * emb_dim
        self.minion = minion_maker(cfg)
        self.loss = self.minion.loss
        self.loss_weight = self.minion.loss_weight

    def forward(self, x, alpha=1, device=None):
        y, label = self.minion(x, alpha)
        return y, label.float().to(device)


def cls_worker_maker(cfg, emb_dim):
    """cls_minions.py:10-27 (spc / gap are workers.cfg-only / unshipped; not in the PASE+ path)."""
    print("=" * 50)
    print("name", cfg["name"])
    print("=" * 50)
    if cfg["name"] == "mi":
        return LIM(cfg, emb_dim)
    if cfg["name"] == "cmi":
        return GIM(cfg, emb_dim)
    if cfg["name"] == "spc":
        return SPC(cfg, emb_dim)
    if cfg["name"] == "gap":
        return Gap(cfg, emb_dim)
    return minion_maker(cfg)

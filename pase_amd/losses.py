"""Mirror of pase/losses.py:6-37 (ContextualizedLoss) on the HIP loss kernel (pase_ctx_loss).

The GAN losses of the reference file (ZAdversarialLoss / WaveAdversarialLoss, :40-213) belong to
configs no shipped PASE(+) worker file uses and are out of scope."""
import torch
import torch.nn as nn

from . import kernels as K

_TYPES = {"L1Loss": K.LOSS_L1, "MSELoss": K.LOSS_MSE, "BCEWithLogitsLoss": K.LOSS_BCE}


class _CtxLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gtruth, r, loss_type):
        pred = pred.contiguous()
        gtruth = gtruth.contiguous()
        B, M, F = pred.shape
        rr = r if (r is not None and r > 1) else 0
        if rr:
            if gtruth.shape[1] * rr != M or gtruth.shape[2] != F:
                raise ValueError("ContextualizedLoss: pred %s vs target %s with r=%d" % (tuple(pred.shape),
                                                                                      tuple(gtruth.shape), rr))
        elif gtruth.shape != pred.shape:
            raise ValueError("ContextualizedLoss: shape mismatch %s vs %s" % (tuple(pred.shape), tuple(gtruth.shape)))
        acc = torch.zeros(1, dtype=torch.float64, device=pred.device)
        dpred = torch.empty_like(pred)
        n = pred.numel()
        K.ctx_loss(pred, gtruth, dpred, acc, B=B, M=M, F=F, r_ctx=rr, label_D=gtruth.shape[1], loss_type=loss_type,
                   grad_scale=1.0 / n)
        ctx.save_for_backward(dpred)
        return (acc / n).to(torch.float32)[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None, None


class ContextualizedLoss(object):
    """criterion(pred, stack of r neighbouring target frames) with mean reduction.
    `criterion` is an nn.L1Loss / nn.MSELoss / nn.BCEWithLogitsLoss instance (as built by
    worker_parser, pase/utils.py:62-68); only its type is used -- the arithmetic is the HIP kernel."""

    def __init__(self, criterion, r=None):
        self.criterion = criterion
        self.r = r
        name = criterion if isinstance(criterion, str) else type(criterion).__name__
        if name not in _TYPES:
            raise NotImplementedError("pase_amd ContextualizedLoss: %s" % name)
        self.loss_name = name
        self.loss_type = _TYPES[name]

    def __call__(self, pred, gtruth):
        if self.r is not None:
            assert isinstance(self.r, int), type(self.r)
            assert len(gtruth.shape) == 3, gtruth.shape
        return _CtxLossFn.apply(pred, gtruth, self.r, self.loss_type)

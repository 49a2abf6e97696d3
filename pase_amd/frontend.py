"""Drop-in mirror of pase/models/frontend.py for the PASE / PASE+ encoder: `wf_builder(cfg)` and
`WaveFe` with the reference's constructor kwargs, state_dict, `forward(batch, device, mode)`,
`emb_dim`, `load_pretrained`; the arithmetic runs on the HIP kernels (pase_amd.engine).

Reference: wf_builder frontend.py:18-40; WaveFe.__init__ :116-211; forward :234-279.
"""
import json

import torch
import torch.nn as nn

from . import engine
from .modules import (FeBlock, Model, build_rnn_block, format_frontend_chunk, format_frontend_output)


def wf_builder(cfg_path):
    """str -> json.load -> dict; dict without "name" -> WaveFe(**cfg); None -> ValueError;
    a "name" key selects the alternative frontends of the reference (asppRes / Resnet50 / tdnn),
    which no shipped PASE(+) cfg uses and this engine does not implement (frontend.py:25-34)."""
    if cfg_path is not None:
        if isinstance(cfg_path, str):
            with open(cfg_path, "r") as cfg_f:
                cfg = json.load(cfg_f)
                return wf_builder(cfg)
        elif isinstance(cfg_path, dict):
            if "name" in cfg_path.keys():
                model_name = cfg_path["name"]
                if model_name in ("asppRes", "Resnet50", "tdnn"):
                    raise NotImplementedError(
                        "pase_amd: frontend %r is outside the accelerated PASE/PASE+ path" % model_name)
                raise TypeError("Unrecognized frontend type: ", model_name)
            return WaveFe(**cfg_path)
        else:
            TypeError("Unexpected config for WaveFe")   # (sic) the reference builds but never raises this
    else:
        raise ValueError("cfg cannot be None!")


class _EncoderFn(torch.autograd.Function):
    """autograd bridge: one node for the whole encoder; backward = engine.encoder_backward."""

    @staticmethod
    def forward(ctx, fe, x, *params):
        out, ectx = engine.encoder_forward(fe, x, fe.training)
        ctx.fe = fe
        ctx.ectx = ectx
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, dout):
        sink = engine.GradSink(direct=False)
        dx = engine.encoder_backward(ctx.fe, ctx.ectx, dout, sink, want_dx=ctx.needs_input_grad[1])
        grads = tuple(sink.get(p) if ctx.needs_input_grad[2 + i] else None for i, p in enumerate(ctx.params))
        ctx.ectx = None
        return (None, dx) + grads


class WaveFe(Model):
    """Convolutional front-end: SincNet -> strided conv/BN/PReLU stack -> QRNN -> 1x1 -> dense skips
    -> BatchNorm(affine=False).  Same kwargs and defaults as the reference (frontend.py:120-143)."""

    def __init__(self, num_inputs=1, sincnet=True, kwidths=[251, 10, 5, 5, 5, 5, 5, 5],
                 strides=[1, 10, 2, 1, 2, 1, 2, 2], dilations=[1, 1, 1, 1, 1, 1, 1, 1],
                 fmaps=[64, 64, 128, 128, 256, 256, 512, 512], norm_type="bnorm", pad_mode="reflect", sr=16000,
                 emb_dim=256, rnn_dim=None, activation=None, rnn_pool=False, rnn_layers=1, rnn_dropout=0,
                 rnn_type="qrnn", vq_K=None, vq_beta=0.25, vq_gamma=0.99, norm_out=False, tanh_out=False,
                 resblocks=False, denseskips=False, densemerge="sum", name="WaveFe"):
        super().__init__(name=name)
        if resblocks or tanh_out or (vq_K is not None and vq_K > 0):
            raise NotImplementedError("pase_amd WaveFe: resblocks / tanh_out / VQ are outside the PASE(+) cfgs")
        if denseskips and densemerge != "sum":
            raise NotImplementedError("pase_amd WaveFe: densemerge='concat'")
        self.num_inputs = num_inputs
        self.sincnet = sincnet
        self.kwidths = kwidths
        self.strides = strides
        self.fmaps = fmaps
        self.densemerge = densemerge
        self.denseskips_on = bool(denseskips)
        if denseskips:
            self.denseskips = nn.ModuleList()
        self.blocks = nn.ModuleList()
        assert len(kwidths) == len(strides)
        assert len(strides) == len(fmaps)
        ninp = num_inputs
        for n, (kwidth, stride, dilation, fmap) in enumerate(zip(kwidths, strides, dilations, fmaps), start=1):
            if n > 1:
                sincnet = False
            self.blocks.append(FeBlock(ninp, fmap, kwidth, stride, dilation, act=activation, pad_mode=pad_mode,
                                       norm_type=norm_type, sincnet=sincnet, sr=sr))
            if denseskips and n < len(kwidths):
                self.denseskips.append(nn.Conv1d(fmap, emb_dim, 1, bias=False))
            ninp = fmap
        if rnn_pool:
            if rnn_dim is None:
                rnn_dim = emb_dim
            self.rnn = build_rnn_block(fmap, rnn_dim // 2, rnn_layers=rnn_layers, rnn_type=rnn_type,
                                       bidirectional=True, dropout=rnn_dropout)
            self.W = nn.Conv1d(rnn_dim, emb_dim, 1)
        else:
            self.W = nn.Conv1d(fmap, emb_dim, 1)
        self.emb_dim = emb_dim
        self.rnn_pool = rnn_pool
        self.quantizer = None
        if norm_out:                      # frontend.py:206-210
            if norm_type == "bnorm":
                self.norm_out = nn.BatchNorm1d(self.emb_dim, affine=False)
            else:
                self.norm_out = nn.InstanceNorm1d(self.emb_dim)
        self.tanh_out = tanh_out

    @property
    def norm_out_mod(self):
        return getattr(self, "norm_out", None)

    def encode(self, x):
        """(S, 1, T) fp32 on the kernel device -> (S, emb_dim, T // 160)."""
        params = [p for p in nn.Module.parameters(self) if p.requires_grad]
        if torch.is_grad_enabled() and (len(params) > 0 or x.requires_grad):
            return _EncoderFn.apply(self, x, *params)
        out, _ = engine.encoder_forward(self, x, self.training)
        return out

    def forward(self, batch, device=None, mode=None):
        x, data_fmt = format_frontend_chunk(batch, device)
        y = self.encode(x)
        return format_frontend_output(y, data_fmt, mode)

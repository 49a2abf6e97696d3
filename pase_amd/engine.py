"""Hand-scheduled forward / backward of the PASE(+) encoder and workers on the HIP kernels.

This module is the MI355X-native replacement for what autograd + cuDNN/cuBLAS + torchqrnn do in
the reference: every contraction is one `pase_conv_gemm` /
`pase_wgrad_gemm` launch, every BatchNorm / PReLU is folded into the consumer's load, and the
backward pass is an explicit schedule (no autograd graph).  It reads parameters from the
reference-compatible nn.Modules in `pase_amd.frontend` / `pase_amd.minions` and accumulates
gradients straight into their `.grad` buffers (which may be views of one flat buffer).

Layout: all tensors NCT fp32; "Act" = a raw stored tensor + the per-channel affine / PReLU that the
next consumer applies on load.
"""
from dataclasses import dataclass
from typing import Optional

import torch

from . import kernels as K

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


@dataclass
class Act:
    t: torch.Tensor                       # (S, ctot, T) raw values
    C: int
    coff: int = 0
    scale: Optional[torch.Tensor] = None  # (C) on-load affine
    shift: Optional[torch.Tensor] = None
    alpha: Optional[torch.Tensor] = None  # (C) on-load PReLU slope

    @property
    def S(self):
        return self.t.shape[0]

    @property
    def ctot(self):
        return self.t.shape[1]

    @property
    def T(self):
        return self.t.shape[2]


def _new(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


class ZeroArena:
    """One buffer, ONE memset per step, for the few dozen small zero-initialised scratch tensors of a step (loss
    accumulators, per-channel sums, split weight-gradient staging): the fused trainer calls begin_step(), every
    `_zeros` below then hands out a view.  The layout repeats from step to step; a step that needs more than the
    current capacity falls back to torch.zeros for the overflow and the arena is regrown at the next begin_step.
    Views are only valid until the next begin_step()."""

    LIMIT = 1 << 30       # larger requests keep their own allocation
    TOTAL = 4 << 30       # ... and so does everything past this much per step: arena regions are not recycled within a step
                          # (split-K outputs live until their consumer has run), so the arena is as large as their SUM --
                          # bounded here; bench.py reports the size (`zero_arena_MB`)

    def __init__(self):
        self.buf = None
        self.stream = None
        self.used = 0
        self.need = 0

    def begin_step(self, device):
        self.stream = torch.cuda.current_stream(device)
        want = max(self.need, self.used)
        if self.buf is None or self.buf.device != device or self.buf.numel() < want:
            self.buf = torch.empty(max(want * 5 // 4, 1 << 20), dtype=torch.uint8, device=device)
            self.buf.zero_()
        elif self.used:
            self.buf[:self.used].zero_()
        self.used = 0
        self.need = 0

    def take(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        if nbytes > self.LIMIT or nbytes == 0:
            return None
        off = (self.used + 255) // 256 * 256
        if off + nbytes > self.TOTAL:
            return None
        self.need = max(self.need, off + nbytes)
        if self.buf is None or off + nbytes > self.buf.numel():
            self.need = off + nbytes
            self.used = off + nbytes          # keep the layout (and the regrow size) consistent
            return None
        self.used = off + nbytes
        return self.buf[off:off + nbytes].view(dtype).view(shape)


_ARENA = None          # set by the fused trainer for the duration of a step


def _zeros(shape, like, dtype=torch.float32):
    # (side streams fork from the step's main stream after begin_step's memset and join before the next one, so arena
    #  views are safe on them too)
    if _ARENA is not None and like.is_cuda and like.device == _ARENA.buf.device:
        t = _ARENA.take(tuple(shape), dtype)
        if t is not None:
            return t
    return torch.zeros(shape, device=like.device, dtype=dtype)


K.ZERO_ALLOC = _zeros


# =========================================================================================
# primitive layers
# =========================================================================================
def reflect_pads(k, stride):
    """FeBlock.forward padding rule (pase/models/modules.py:1059-1071), dilation 1."""
    if k <= 1:
        return 0, 0
    if stride > 1 or k % 2 == 0:
        return k // 2 - 1, k // 2
    return k // 2, k // 2


def conv_fwd(a: Act, w2d, bias, *, Cout, taps, stride=1, padL=0, padR=0, pad_mode=K.PAD_ZERO, want_stats=False,
             tap_major=0, tapstep=1, out=None, out_coff=0, Tout=None, max_wg=0):
    """y[s,co,t] = b + sum w[co,(ci,kk)] * a~[s,ci,t*stride + kk*tapstep - padL]."""
    S, Tin = a.S, a.T
    if Tout is None:
        Tout = (Tin + padL + padR - taps) // stride + 1
    kw = dict(S=S, Cin=a.C, Tin=Tin, M=Cout, K=a.C * taps, taps=taps, Ncols=Tout, Tout=Tout, ldw=w2d.shape[1], bias=bias,
              in_scale=a.scale, in_shift=a.shift, in_alpha=a.alpha, x_ctot=a.ctot, x_coff=a.coff, tap_major=tap_major,
              stride=stride, tapstep=tapstep, padL=padL, pad_mode=pad_mode, Cout_store=Cout)
    if out is None and not want_stats:
        return K.conv_gemm_out(a.t, w2d, (S, Cout, Tout), y_ctot=Cout, y_coff=0, **kw, max_wg=max_wg), None
    y = out if out is not None else _new((S, Cout, Tout), a.t)
    stat = K.conv_gemm(a.t, w2d, y, want_stats=want_stats, y_ctot=y.shape[1], y_coff=out_coff, **kw, max_wg=max_wg)
    return y, stat


def conv_dgrad(dy, w_nat, *, R, O, k, stride, Tin, padL, padR, s_red, s_out, s_k, max_wg=0):
    """Data-gradient of a (strided) conv in *padded* coordinates: (S, O, Tin+padL+padR).
    dXpad[s,o,u] = sum_{red,kk} W[red,o,kk] * dy[s,red,(u-kk)/stride]  (phase decomposition)."""
    S, _, Tg = dy.shape
    taps_p = -(-k // stride)
    wt = K.pack_dgrad_t(w_nat, R=R, O=O, k=k, st=stride, s_red=s_red, s_out=s_out, s_k=s_k)
    Tp = Tin + padL + padR
    Ncols = -(-Tp // stride)
    # (split-K launches -- the wide heads' data gradients, decoder layers whose tiles do not fill whole rounds -- add into a
    #  zeroed output: it comes out of the step's zero arena, K.conv_gemm_out)
    return K.conv_gemm_out(dy, None, (S, O, Tp), wt=wt, S=S, Cin=R, Tin=Tg, M=stride * O, K=R * taps_p, taps=taps_p, Ncols=Ncols,
                           Tout=Tp, stride=1, tapstep=-1, padL=0, pad_mode=K.PAD_ZERO, Cout_store=O, ps=stride, poff=0, max_wg=max_wg)


def conv_wgrad(dy, a: Act, dw2d, dbias, *, taps, stride=1, padL=0, pad_mode=K.PAD_ZERO, tap_major=0, tapstep=1,
               g_ctot=None, g_coff=0, M=None, Ncols=None, g_alpha=None, dw_col_off=0, max_wg=0, g_bwd=None, g_shape=None):
    """g_bwd (with g_shape = (S, M, T) of the gradient it stands for, dy = None): see kernels.wgrad_gemm; returns False
    when that launch has no on-load form."""
    S = a.S
    gs = tuple(dy.shape) if dy is not None else tuple(g_shape)
    M = gs[1] if M is None else M
    return K.wgrad_gemm(dy, a.t, dw2d, S=S, M=M, Tg=gs[2], Ncols=gs[2] if Ncols is None else Ncols, Cin=a.C,
                        Tz=a.T, taps=taps, ldw=dw2d.shape[1], dbias=dbias, g_ctot=gs[1] if g_ctot is None else g_ctot,
                        g_coff=g_coff, z_ctot=a.ctot, z_coff=a.coff, in_scale=a.scale, in_shift=a.shift, in_alpha=a.alpha,
                        tap_major=tap_major, stride=stride, tapstep=tapstep, padL=padL, pad_mode=pad_mode,
                        g_alpha=g_alpha, dw_col_off=dw_col_off, max_wg=max_wg, g_bwd=g_bwd)


def deconv_fwd(a: Act, w_nat, bias, *, Cout, k, stride, max_wg=0):
    """nn.ConvTranspose1d(Cin, Cout, k, stride, padding=max(0,(stride-k)//-2)) (modules.py:567-575)."""
    S, Tin = a.S, a.T
    pad = max(0, (stride - k) // -2)
    taps_p = -(-k // stride)
    wt = K.pack_dgrad_t(w_nat, R=a.C, O=Cout, k=k, st=stride, s_red=Cout * k, s_out=k, s_k=1)
    Tout = (Tin - 1) * stride - 2 * pad + k
    return K.conv_gemm_out(a.t, None, (S, Cout, Tout), wt=wt, S=S, Cin=a.C, Tin=Tin, M=stride * Cout, K=a.C * taps_p,
                           taps=taps_p, Ncols=Tin + taps_p - 1, Tout=Tout, bias=bias, in_scale=a.scale, in_shift=a.shift,
                           in_alpha=a.alpha, x_ctot=a.ctot, x_coff=a.coff, stride=1, tapstep=-1, padL=0, pad_mode=K.PAD_ZERO,
                           Cout_store=Cout, ps=stride, poff=-pad, max_wg=max_wg)


_NBT = None      # BatchNorm counters touched by the encoder forward in flight (one multi-tensor increment)


def bn_train(stat, C, count, norm, like):
    """Batch-statistics BatchNorm1d: returns (scale, shift, mean, rstd); updates running stats."""
    scale, shift, mean, rstd = (_new((C,), like) for _ in range(4))
    gamma = norm.weight if norm.affine else None
    beta = norm.bias if norm.affine else None
    if norm.momentum is None:
        # torch semantics: cumulative moving average with factor 1/num_batches_tracked (a device counter here)
        raise NotImplementedError("pase_amd: BatchNorm1d(momentum=None) (cumulative average) is not supported")
    mom = norm.momentum
    K.bn_finalize(stat, C, count, gamma, beta, norm.eps, mom, norm.running_mean, norm.running_var, scale, shift,
                  mean, rstd)
    if norm.num_batches_tracked is not None:
        if _NBT is not None:
            _NBT.append(norm.num_batches_tracked)      # committed together at the end of encoder_forward
        else:
            norm.num_batches_tracked.add_(1)
    return scale, shift, mean, rstd


def bn_eval(norm):
    rstd = torch.rsqrt(norm.running_var + norm.eps)
    if norm.affine:
        scale = norm.weight.detach() * rstd
        shift = norm.bias.detach() - norm.running_mean * scale
    else:
        scale = rstd
        shift = -norm.running_mean * scale
    return scale.contiguous(), shift.contiguous(), norm.running_mean, rstd


def norm_kind(norm):
    """'bn' (statistics ride on the consumer's on-load transform), 'in' / 'ln' (per-sample statistics: the activated
    tensor is materialised by pase_rownorm_act_fwd), None."""
    import torch.nn as nn
    if norm is None:
        return None
    if isinstance(norm, nn.BatchNorm1d):
        return "bn"
    if isinstance(norm, nn.InstanceNorm1d):
        if norm.track_running_stats:
            raise NotImplementedError("pase_amd: InstanceNorm1d(track_running_stats=True)")
        return "in"
    if isinstance(norm, nn.LayerNorm):
        return "ln"
    raise NotImplementedError("pase_amd: norm layer %r" % type(norm).__name__)


def rownorm_fwd(y, norm, kind, alpha):
    """a = PReLU(norm(y)) materialised, plus the per-group statistics for the backward."""
    S, C, T = y.shape
    out = _new((S, C, T), y)
    n = S * C if kind == "in" else S * T
    mean, rstd = _new((n,), y), _new((n,), y)
    affine = norm.elementwise_affine if kind == "ln" else norm.affine
    K.rownorm_act_fwd(y, out, norm.weight if affine else None, norm.bias if affine else None, alpha, mean, rstd,
                      S=S, C_=C, T=T, eps=float(norm.eps), mode=K.NORM_INSTANCE if kind == "in" else K.NORM_LAYER)
    return out, mean, rstd


def rownorm_backward(y, norm, kind, alpha, mean, rstd, *, dsrc=None, dsrc_ctot=None, dsrc_coff=0, Tp=None, padL=0,
                     pad_mode=K.PAD_ZERO, dpool=None, dpool_ctot=0, dpool_coff=0, pool_F=0, pool_d=1):
    """Backward of a = PReLU(norm(y)) for the per-sample norms: (dy, sums (C,3) = {dbeta, dgamma, dalpha})."""
    S, C, T = y.shape
    sums = _zeros((C, 3), y, torch.float64)
    dy = _new((S, C, T), y)
    affine = norm.elementwise_affine if kind == "ln" else norm.affine
    K.rownorm_act_bwd(y, S=S, C_=C, T=T, dsrc=dsrc, dsrc_ctot=dsrc_ctot, dsrc_coff=dsrc_coff, Tp=Tp, padL=padL,
                      pad_mode=pad_mode, dpool=dpool, dpool_ctot=dpool_ctot, dpool_coff=dpool_coff, pool_F=pool_F,
                      pool_d=pool_d, scale=norm.weight if affine else None, shift=norm.bias if affine else None,
                      alpha=alpha, mean=mean, rstd=rstd, sums=sums, dy=dy, has_bn=3 if kind == "in" else 4)
    return dy, sums


def act_backward(y, *, C, T, S, has_bn, scale=None, shift=None, alpha=None, mean=None, rstd=None, dsrc=None,
                 dsrc_ctot=None, dsrc_coff=0, Tp=None, padL=0, pad_mode=K.PAD_ZERO, dpool=None, dpool_ctot=0,
                 dpool_coff=0, pool_F=0, pool_d=1, y_ctot=None, y_coff=0, dy_out=None, defer_apply=False):
    """Backward of a = PReLU(BN(y)): returns (dy, sums) with sums (C,3) double =
    {dbeta | sum dz, dgamma, dalpha}.  has_bn: False/0 none, True/1 batch statistics, 2 frozen (eval-mode)
    statistics, whose backward is dy = scale*dz (torch's batch_norm backward with training=False).
    defer_apply: only the reduce pass runs and dy is not written; returns (apply, sums) where `apply` holds the apply pass's
    arguments for the consumer that evaluates it on load (kernels.wgrad_gemm(g_bwd=apply)) -- or for act_backward_apply()."""
    sums = _zeros((C, 3), y, torch.float64)
    kw = dict(S=S, C_=C, T=T, y_ctot=y_ctot, y_coff=y_coff, dsrc=dsrc, dsrc_ctot=dsrc_ctot, dsrc_coff=dsrc_coff, Tp=Tp, padL=padL,
              pad_mode=pad_mode, dpool=dpool, dpool_ctot=dpool_ctot, dpool_coff=dpool_coff, pool_F=pool_F,
              pool_d=pool_d, scale=scale, shift=shift, alpha=alpha, mean=mean, rstd=rstd, sums=sums, dy=None,
              has_bn=int(has_bn))
    if defer_apply:
        K.act_bwd_reduce(y, **kw)        # (dy NULL: sums only, whatever has_bn is)
        return dict(kw, y=y), sums
    kw["dy"] = dy = dy_out if dy_out is not None else _new(tuple(y.shape), y)
    K.act_bwd_reduce(y, **kw)
    if int(has_bn) == 1:             # without a BatchNorm / behind a frozen one the reduce pass has already written dy
        K.act_bwd_apply(y, **kw)
    return dy, sums


def act_backward_apply(apply):
    """materialise the dy of a deferred act_backward (its consumer had no on-load form)"""
    kw = dict(apply)
    y = kw.pop("y")
    kw["dy"] = dy = _new(tuple(y.shape), y)
    K.act_bwd_apply(y, **kw)
    return dy


_WIDE_HEAD_ELEMS = 16 * 1024 * 1024     # prediction elements above which a head fills the chip by itself
_SIDE = {}


def side_streams(like, n):
    """n side HIP streams on `like`'s device (none on the CPU emulator; PASE_SIDE_STREAMS=0 serialises
    everything on the current stream)."""
    import os
    if not like.is_cuda or os.environ.get("PASE_SIDE_STREAMS", "1") == "0":
        return []
    pool = _SIDE.setdefault(like.device.index, [])       # ONE pool per device: side_streams(x, 3) is a prefix of (x, 4)
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=like.device))
    return pool[:n]


_WSTREAM = {}
_WKEEP = {}


def wgrad_stream(like):
    """The ONE side stream weight-gradient launches go to (PASE_WGRAD_STREAM=0: none): layer n's weight gradient then runs
    beside layer n's data gradient and layer n-1's elementwise backward.  Every GEMM kernel is a persistent grid of one
    workgroup per CU, so the two never share a CU -- the second kernel's workgroups start on the CUs the first one's last
    round leaves idle (measured: step 33.4 -> 32.7 ms from the encoder blocks alone).  None while per-launch timing is on."""
    import os
    if not like.is_cuda or K.GEMM_TIMER is not None or os.environ.get("PASE_WGRAD_STREAM", "1") == "0":
        return None
    ws = _WSTREAM.get(like.device.index)
    if ws is None:
        ws = _WSTREAM[like.device.index] = torch.cuda.Stream(device=like.device)
    return ws


def on_wgrad_stream(like, fn, keep=()):
    """Run fn() (weight-gradient launches) on the weight-gradient stream, ordered after everything enqueued on the current
    stream so far; `keep`: tensors fn reads that the caller may drop before the stream has run.  The caller joins with
    join_wgrad_stream() before the gradients are consumed."""
    ws = wgrad_stream(like)
    cur = torch.cuda.current_stream() if ws is not None else None
    # only the step's main stream forks: a worker that already runs on one of the head side streams keeps its weight
    # gradients in line (nested forks bought nothing and broke hipGraph capture of the small configurations)
    if ws is None or (_ARENA is not None and _ARENA.stream is not None and cur.cuda_stream != _ARENA.stream.cuda_stream):
        fn()
        return
    ws.wait_event(cur.record_event())
    try:
        with torch.cuda.stream(ws):
            fn()
    except BaseException:
        # leave no fork behind: the forking stream waits for whatever was enqueued, the kept operands are released
        cur.wait_stream(ws)
        _WKEEP.pop((like.device.index, cur.cuda_stream), None)
        raise
    # The operands must outlive the side stream's kernels.  Not Tensor.record_stream (its deferred-free events are not
    # graph-capture safe: capture_end segfaulted on the mini configuration): hold references until the forking stream
    # joins -- after that, memory the caching allocator hands back to that stream is ordered behind the side stream's work.
    _WKEEP.setdefault((like.device.index, cur.cuda_stream), []).extend(t for t in keep if t is not None)


def join_wgrad_stream(like):
    ws = wgrad_stream(like)
    if ws is not None:
        cur = torch.cuda.current_stream()
        if _WKEEP.pop((like.device.index, cur.cuda_stream), None) is not None:      # this stream has forked since its last join
            cur.wait_stream(ws)


class GradSink:
    """Where parameter gradients go.  direct=True: accumulate straight into param.grad (the fused
    trainer zeroes one flat buffer per step and lets the wgrad kernels add into views of it);
    direct=False: collect fresh buffers (returned to autograd by the API-compat Functions)."""

    def __init__(self, direct=True):
        self.direct = direct
        self.store = {}

    def buf(self, p):
        if self.direct and p.requires_grad:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            return p.grad
        b = self.store.get(p)
        if b is None:
            b = torch.zeros_like(p)
            self.store[p] = b
        return b

    def add(self, p, value):
        self.buf(p).add_(value.reshape(p.shape))

    def add_many(self, pairs):
        """param.grad += value for several (param, value) pairs in ONE launch: pase_add_blocks takes 2-D row-major blocks with a
        row stride on either side (column / row slices of a staged concatenated gradient); anything else goes to torch
        (whose _foreach_add_ issues one strided add per non-contiguous slice)."""
        if not pairs:
            return
        blocks = []
        for p, v in pairs:
            d = self.buf(p)
            if v.dim() == 2 and d.is_contiguous() and d.numel() == v.numel() and v.shape[0] == p.shape[0]:
                blocks.append((d.view(v.shape[0], -1), v))
            else:
                blocks = None
                break
        if blocks is not None and K.add_blocks(blocks):
            return
        dst = [self.buf(p) for p, _ in pairs]
        src = [v.reshape(p.shape) for p, v in pairs]
        if len(dst) == 1:
            dst[0].add_(src[0])
        elif dst:
            torch._foreach_add_(dst, src)

    def get(self, p):
        return self.store.get(p)

    def add_cols(self, sums, ld, C, pairs):
        """param.grad[c] += sums[c*ld + col] for up to three (param or None, col) pairs, one launch."""
        K.commit_cols(sums, ld, C, [(None if p is None else self.buf(p).view(-1), col) for p, col in pairs])


# =========================================================================================
# encoder (WaveFe)
# =========================================================================================
class EncoderCtx:
    pass


def encoder_forward(fe, x, training, need_ctx=True, max_wg=0):
    """WaveFe.forward on a (S,1,T) tensor (pase/models/frontend.py:234-279 minus the dict plumbing).
    Returns (emb (S,emb_dim,F), ctx)."""
    assert x.dim() == 3 and x.shape[1] == fe.num_inputs, x.shape
    x = x.contiguous()
    S = x.shape[0]
    global _NBT
    _NBT = []
    try:
        return _encoder_forward(fe, x, training, S, max_wg)
    finally:
        if _NBT:
            torch._foreach_add_(_NBT, 1)
        _NBT = None


def _encoder_forward(fe, x, training, S, max_wg=0):
    ctx = EncoderCtx()
    ctx.x = x
    ctx.training = bool(training)
    ctx.blocks = []
    cur = Act(x, C=x.shape[1])
    for n, blk in enumerate(fe.blocks):
        rec = {}
        k, st = blk.kwidth, blk.stride
        if blk.sincnet:
            conv = blk.conv
            kk = conv.kernel_size
            filt = _new((conv.out_channels, kk), x)
            conv.n_ = conv.n_.to(x.device)
            conv.window_ = conv.window_.to(x.device)
            K.sinc_filters(conv.low_hz_, conv.band_hz_, conv.n_, conv.window_, filt, C_=conv.out_channels, Kw=kk,
                           min_low=float(conv.min_low_hz), min_band=float(conv.min_band_hz),
                           sr=float(conv.sample_rate))
            if st > 1:
                pL, pR = kk // 2 - 1, kk // 2
            else:
                pL, pR = kk // 2, kk // 2
            w2d, bias, taps = filt, None, kk
            rec["filt"] = filt
        else:
            pL, pR = reflect_pads(k, st)
            w2d, bias, taps = blk.conv.weight.view(blk.fmaps, -1), blk.conv.bias, k
        kind = norm_kind(getattr(blk, "norm", None))
        has_bn = kind == "bn"
        y, stat = conv_fwd(cur, w2d, bias, Cout=blk.fmaps, taps=taps, stride=st, padL=pL, padR=pR,
                           pad_mode=K.PAD_REFLECT, want_stats=has_bn and training, max_wg=max_wg)
        if has_bn:
            if training:
                scale, shift, mean, rstd = bn_train(stat, blk.fmaps, S * y.shape[2], blk.norm, x)
            else:
                scale, shift, mean, rstd = bn_eval(blk.norm)
        else:
            scale = shift = mean = rstd = None
        rec.update(inp=cur, y=y, scale=scale, shift=shift, mean=mean, rstd=rstd, padL=pL, padR=pR, taps=taps,
                   has_bn=has_bn, kind=kind)
        if kind in ("in", "ln"):
            # per-sample statistics: materialise a = PReLU(norm(y)); consumers load it untransformed
            amat, gmean, grstd = rownorm_fwd(y, blk.norm, kind, blk.act.weight)
            rec.update(amat=amat, mean=gmean, rstd=grstd)
            cur = Act(amat, C=blk.fmaps)
        else:
            cur = Act(y, C=blk.fmaps, scale=scale, shift=shift, alpha=blk.act.weight)
        ctx.blocks.append(rec)
    F_ = cur.T
    ctx.F = F_

    # ---- QRNN stack (frontend.py:256-259; third-party torchqrnn) -------------------------------
    ctx.rnn = []
    skips = fe.denseskips_on
    emb = fe.W.out_channels
    if skips:
        ccat = fe.W.in_channels + sum(b.fmaps for b in fe.blocks[:-1])
        acat = _new((S, ccat, F_), x)
    def pool_skips():
        # dense skips: mean-pool each block's activation to the frame rate (the ONE 1x1 GEMM over the concatenated channels
        # follows below: pool-then-project == project-then-pool for a bias-free 1x1 conv)
        off = fe.W.in_channels
        ctx.skip_off = []
        for n, blk in enumerate(fe.blocks[:-1]):
            rec = ctx.blocks[n]
            Tn = rec["y"].shape[2]
            d = Tn // F_
            if "amat" in rec:
                K.bn_act_pool(rec["amat"], acat, None, None, None, S=S, C_=blk.fmaps, T=Tn, F=F_, d=d, o_ctot=ccat,
                              o_coff=off)
            else:
                K.bn_act_pool(rec["y"], acat, rec["scale"], rec["shift"], blk.act.weight, S=S, C_=blk.fmaps, T=Tn,
                              F=F_, d=d, o_ctot=ccat, o_coff=off)
            ctx.skip_off.append((off, d))
            off += blk.fmaps

    # The pooling passes (HBM-bound, 0.5 ms per bs32 step) only feed the concatenated GEMM after the QRNN stack: on the
    # GPU they run on their own stream beside the QRNN's GEMM + scan (PASE_POOL_STREAM=0: in line, after the stack)
    pool_stream = None
    if skips and x.is_cuda and K.GEMM_TIMER is None and __import__("os").environ.get("PASE_POOL_STREAM", "1") != "0":
        # (one of the head side streams, idle during the encoder forward -- NOT a stream of its own: HIP multiplexes streams
        #  onto a few hardware queues, and one more stream put the host-buffer feeder's copy stream behind compute work:
        #  +4 ms per step in the host-buffer mode)
        ss = side_streams(x, 1)
        pool_stream = ss[0] if ss else None
    if pool_stream is not None:
        pool_stream.wait_event(torch.cuda.current_stream().record_event())
        with torch.cuda.stream(pool_stream):
            pool_skips()
    rnn_in = cur
    if fe.rnn_pool:
        layers = fe.rnn.layers
        for li, layer in enumerate(layers):
            H = layer.hidden_size
            gates, _ = conv_fwd(rnn_in, layer.linear.weight, layer.linear.bias, Cout=3 * H, taps=2, tap_major=1,
                                tapstep=-1, padL=0, padR=0, pad_mode=K.PAD_ZERO, Tout=F_, max_wg=max_wg)
            last = li == len(layers) - 1
            if last and skips:
                h_t, h_ctot = acat, ccat
            else:
                h_t, h_ctot = _new((S, H, F_), x), H
            c = _new((S, H, F_), x)
            K.qrnn_scan_fwd(gates, h_t, c, S=S, H=H, F=F_, h_ctot=h_ctot, h_coff=0)
            ctx.rnn.append(dict(inp=rnn_in, gates=gates, c=c, H=H))
            rnn_in = Act(h_t, C=H)
    elif skips:
        # no RNN but dense skips: materialise act(bn(y_last)) into the concat buffer (pool with d=1)
        K.bn_act_pool(cur.t, acat, cur.scale, cur.shift, cur.alpha, S=S, C_=cur.C, T=F_, F=F_, d=1, o_ctot=ccat,
                      o_coff=0)
        rnn_in = Act(acat, C=cur.C)

    # ---- dense skips: the pooled activations (pool_skips above) + ONE 1x1 GEMM over the concatenated channels -------
    if skips:
        if pool_stream is not None:
            torch.cuda.current_stream().wait_stream(pool_stream)
        else:
            pool_skips()
        wcat = torch.cat([fe.W.weight.view(emb, -1)] + [p.weight.view(emb, -1) for p in fe.denseskips], dim=1)
        ain = Act(acat, C=ccat)
    else:
        wcat = fe.W.weight.view(emb, -1)
        ain = rnn_in
    norm_out = fe.norm_out_mod
    yemb, stat = conv_fwd(ain, wcat, fe.W.bias, Cout=emb, taps=1,
                          want_stats=(norm_kind(norm_out) == "bn" and training), Tout=F_, max_wg=max_wg)
    ctx.out_kind = norm_kind(norm_out)
    if ctx.out_kind == "bn":
        if training:
            scale, shift, mean, rstd = bn_train(stat, emb, S * F_, norm_out, x)
        else:
            scale, shift, mean, rstd = bn_eval(norm_out)
        out = _new((S, emb, F_), x)
        K.bn_act_apply(yemb, out, scale, shift, None, S=S, C_=emb, T=F_)
        ctx.out_bn = (scale, shift, mean, rstd)
    elif ctx.out_kind is not None:          # InstanceNorm1d(emb) (frontend.py:209-210)
        out, gmean, grstd = rownorm_fwd(yemb, norm_out, ctx.out_kind, None)
        ctx.out_bn = (None, None, gmean, grstd)
    else:
        out = yemb
        ctx.out_bn = None
    ctx.ain, ctx.wcat, ctx.yemb = ain, wcat, yemb
    return out, ctx


def encoder_backward(fe, ctx, demb, sink, want_dx=False, on_ready=None, max_wg=0):
    """Accumulates d(loss)/d(param) into `sink` given demb = d(loss)/d(emb); with want_dx also returns
    d(loss)/d(input waveform) (S, num_inputs, T).  on_ready(tag): called as soon as a group of parameter gradients
    is final ON THE CURRENT STREAM -- "head" (W, dense-skip projections, QRNN), then each conv block index from the
    last to the first -- so a data-parallel trainer can start that bucket's all-reduce under the rest of the
    backward (reverse-order bucketing)."""
    demb = demb.contiguous()
    x = ctx.x
    S, F_ = x.shape[0], ctx.F
    emb = fe.W.out_channels
    bn_mode = 1 if ctx.training else 2     # eval-mode forward => frozen-statistics backward (dy = scale*dz)
    # ---- norm_out (BatchNorm1d affine=False) -----------------------------------------------------
    if ctx.out_bn is not None and ctx.out_kind == "bn":
        scale, shift, mean, rstd = ctx.out_bn
        dyemb, _ = act_backward(ctx.yemb, C=emb, T=F_, S=S, has_bn=bn_mode, scale=scale, shift=shift, mean=mean,
                                rstd=rstd, dsrc=demb)
    elif ctx.out_bn is not None:
        _, _, gmean, grstd = ctx.out_bn
        norm_out = fe.norm_out_mod
        dyemb, osums = rownorm_backward(ctx.yemb, norm_out, ctx.out_kind, None, gmean, grstd, dsrc=demb, dsrc_ctot=emb,
                                        Tp=F_)
        if norm_out.affine:
            sink.add_cols(osums, 3, emb, [(norm_out.bias, 0), (norm_out.weight, 1)])
    else:
        dyemb = demb
    # ---- W + dense-skip projections: one wgrad + one dgrad over the concatenated channels --------
    ain = ctx.ain
    ccat = ain.C
    dwcat = _zeros((emb, ccat), x)
    wb_bias = sink.buf(fe.W.bias)        # (gradient buffers are taken on the forking stream: GradSink(direct=False) allocates)
    on_wgrad_stream(x, lambda: conv_wgrad(dyemb, ain, dwcat, wb_bias, taps=1, max_wg=max_wg), keep=(dyemb, ain.t))
    cw = fe.W.in_channels
    pairs = [(fe.W.weight, dwcat[:, :cw])]
    if fe.denseskips_on:
        off = cw
        for p in fe.denseskips:
            c = p.weight.shape[1]
            pairs.append((p.weight, dwcat[:, off:off + c]))
            off += c
    on_wgrad_stream(x, lambda: sink.add_many(pairs), keep=(dwcat,))
    dacat = conv_dgrad(dyemb, ctx.wcat, R=emb, O=ccat, k=1, stride=1, Tin=F_, padL=0, padR=0, s_red=ccat, s_out=1,
                       s_k=1, max_wg=max_wg)  # (S, ccat, F)
    # gradient w.r.t. the last block's activation
    dsrc, dsrc_ctot, dsrc_coff = dacat, ccat, 0
    # ---- QRNN stack --------------------------------------------------------------------------------
    for li in reversed(range(len(ctx.rnn))):
        r = ctx.rnn[li]
        layer = fe.rnn.layers[li]
        H = r["H"]
        dgates = _new((S, 3 * H, F_), x)
        K.qrnn_scan_bwd(r["gates"], r["c"], dsrc, dgates, S=S, H=H, F=F_, dh_ctot=dsrc_ctot, dh_coff=dsrc_coff)
        inp = r["inp"]
        cin = inp.C
        # Linear over [x_t ; x_{t-1}] (tap-major columns): one wgrad per tap into the two column halves
        dwq = sink.buf(layer.linear.weight)
        # x_{t-1} tap: sum_q dG[q] x[q-1] = sum_q dG[q+1] x[q] -- shift the GRADIENT left by one (its padding is a
        # true zero; the input's would have to be a zero of the activated tensor) and stay on the 1x1 kernel
        dg_next = torch.nn.functional.pad(dgates[:, :, 1:], (0, 1))

        dbq = sink.buf(layer.linear.bias)

        def wq(dgates=dgates, dg_next=dg_next, inp=inp, dwq=dwq, dbq=dbq, cin=cin):
            conv_wgrad(dgates, inp, dwq, dbq, taps=1, padL=0, pad_mode=K.PAD_ZERO, max_wg=max_wg)
            conv_wgrad(dg_next, inp, dwq, None, taps=1, padL=0, pad_mode=K.PAD_ZERO, dw_col_off=cin, max_wg=max_wg)
        on_wgrad_stream(x, wq, keep=(dgates, dg_next, inp.t))
        # dX[s,ci,u] = sum_{o,r} Wq[o, r*cin+ci] * dG[s,o,u+r]
        wt = K.pack_dgrad_t(layer.linear.weight, R=3 * H, O=cin, k=2, st=1, s_red=2 * cin, s_out=1, s_k=cin)
        dxl = _new((S, cin, F_), x)
        K.conv_gemm(dgates, None, dxl, wt=wt, S=S, Cin=3 * H, Tin=F_, M=cin, K=3 * H * 2, taps=2, Ncols=F_, Tout=F_,
                    stride=1, tapstep=1, padL=0, pad_mode=K.PAD_ZERO, max_wg=max_wg)
        dsrc, dsrc_ctot, dsrc_coff = dxl, cin, 0
    if on_ready is not None:
        join_wgrad_stream(x)
        on_ready("head")
    # Weight gradients of the conv blocks go to the weight-gradient stream (wgrad_stream above); data-parallel, each
    # bucket is handed over ON that stream; joined at the end.
    wside = wgrad_stream(x)
    # ---- conv blocks, last to first -----------------------------------------------------------------
    dsrc_Tp, dsrc_padL, dsrc_mode = F_, 0, K.PAD_ZERO
    nb = len(fe.blocks)
    for n in reversed(range(nb)):
        blk = fe.blocks[n]
        rec = ctx.blocks[n]
        y = rec["y"]
        C, Tn = blk.fmaps, y.shape[2]
        kw = {}
        if fe.denseskips_on and n < nb - 1:
            off, d = ctx.skip_off[n]
            kw = dict(dpool=dacat, dpool_ctot=ccat, dpool_coff=off, pool_F=F_, pool_d=d)
        kind = rec.get("kind")
        if kind in ("in", "ln"):
            dy, sums = rownorm_backward(y, blk.norm, kind, blk.act.weight, rec["mean"], rec["rstd"], dsrc=dsrc,
                                        dsrc_ctot=dsrc_ctot, dsrc_coff=dsrc_coff, Tp=dsrc_Tp, padL=dsrc_padL,
                                        pad_mode=dsrc_mode, **kw)
            aff = blk.norm.elementwise_affine if kind == "ln" else blk.norm.affine
            sink.add_cols(sums, 3, C, [(blk.norm.bias if aff else None, 0), (blk.norm.weight if aff else None, 1),
                                       (blk.act.weight, 2)])
        else:
            # the SincNet layer's dy feeds its weight gradient and nothing else (no data gradient below the first layer):
            # only the reduce pass runs here and the weight-gradient launch evaluates the apply pass on load
            defer = bool(blk.sincnet and n == 0 and not want_dx and y.shape[1] == C)
            dy, sums = act_backward(y, C=C, T=Tn, S=S, has_bn=bn_mode if rec["has_bn"] else 0, scale=rec["scale"],
                                    shift=rec["shift"], alpha=blk.act.weight, mean=rec["mean"], rstd=rec["rstd"],
                                    dsrc=dsrc, dsrc_ctot=dsrc_ctot, dsrc_coff=dsrc_coff, Tp=dsrc_Tp, padL=dsrc_padL,
                                    pad_mode=dsrc_mode, defer_apply=defer, **kw)
            bn_aff = rec["has_bn"] and blk.norm.affine
            no_bn_bias = (not rec["has_bn"]) and not blk.sincnet      # bias gradient = sum dz when no norm follows
            sink.add_cols(sums, 3, C, [(blk.norm.bias if bn_aff else (blk.conv.bias if no_bn_bias else None), 0),
                                       (blk.norm.weight if bn_aff else None, 1), (blk.act.weight, 2)])
        inp = rec["inp"]
        taps = rec["taps"]
        if blk.sincnet:
            conv = blk.conv
            dfilt = _zeros((C, taps), x)
            if isinstance(dy, dict):
                if not conv_wgrad(None, inp, dfilt, None, taps=taps, stride=blk.stride, padL=rec["padL"],
                                  pad_mode=K.PAD_REFLECT, max_wg=max_wg, g_bwd=dy, g_shape=(S, C, Tn)):
                    dy = act_backward_apply(dy)
            if not isinstance(dy, dict):
                conv_wgrad(dy, inp, dfilt, None, taps=taps, stride=blk.stride, padL=rec["padL"], pad_mode=K.PAD_REFLECT,
                           max_wg=max_wg)
            dlow, dband = _new((C,), x), _new((C,), x)
            K.sinc_filters_bwd(conv.low_hz_, conv.band_hz_, conv.n_, conv.window_, dfilt, dlow, dband, C_=C, Kw=taps,
                               min_low=float(conv.min_low_hz), min_band=float(conv.min_band_hz),
                               sr=float(conv.sample_rate))
            sink.add(conv.low_hz_, dlow)
            sink.add(conv.band_hz_, dband)
        else:
            dbias = sink.buf(blk.conv.bias) if (rec["has_bn"] or rec.get("kind") in ("in", "ln")) else None
            dwblk = sink.buf(blk.conv.weight).view(C, -1)

            def wg(dy=dy, inp=inp, blk=blk, dwblk=dwblk, dbias=dbias, taps=taps, rec=rec, n=n):
                conv_wgrad(dy, inp, dwblk, dbias, taps=taps, stride=blk.stride,
                           padL=rec["padL"], pad_mode=K.PAD_REFLECT, max_wg=max_wg)
                # (the fork event also covers the per-channel sums sink.add_cols committed above: a bucket handed over on
                #  the side stream is complete)
                if on_ready is not None and wside is not None:
                    on_ready(n)
            on_wgrad_stream(x, wg, keep=(dy, inp.t))
        if on_ready is not None and (wside is None or blk.sincnet):
            if wside is not None:          # (block 0's bucket also carries the small blocks' gradients from the side stream)
                torch.cuda.current_stream().wait_stream(wside)
            on_ready(n)
        if n > 0 or want_dx:
            cin = inp.C
            w_nat = rec["filt"] if blk.sincnet else blk.conv.weight
            dsrc = conv_dgrad(dy, w_nat, R=C, O=cin, k=taps, stride=blk.stride, Tin=inp.T,
                              padL=rec["padL"], padR=rec["padR"], s_red=cin * taps, s_out=taps, s_k=1, max_wg=max_wg)
            dsrc_ctot, dsrc_coff = cin, 0
            dsrc_Tp, dsrc_padL, dsrc_mode = dsrc.shape[2], rec["padL"], K.PAD_REFLECT
    join_wgrad_stream(x)
    if not want_dx:
        return None
    # gradient w.r.t. the input waveform (saliency / adversarial use through the drop-in alias): fold the reflect
    # padding of block 0 back onto the interior (autograd of F.pad(mode='reflect')); a rare path, plain slicing
    pL, pR, T0 = ctx.blocks[0]["padL"], ctx.blocks[0]["padR"], x.shape[2]
    dx = dsrc[:, :, pL:pL + T0].clone()
    if pL > 0:
        dx[:, :, 1:pL + 1] += dsrc[:, :, :pL].flip(2)
    if pR > 0:
        dx[:, :, T0 - 1 - pR:T0 - 1] += dsrc[:, :, pL + T0:].flip(2)
    return dx


# =========================================================================================
# workers (Minions): sequential [GDeconv1DBlock | MLPBlock]* -> output conv, with fused losses
# =========================================================================================
@dataclass
class GradSrc:
    """A data-gradient tensor in (possibly padded) coordinates, as act_backward consumes it."""
    t: torch.Tensor
    ctot: int
    coff: int = 0
    Tp: int = 0
    padL: int = 0
    pad_mode: int = K.PAD_ZERO

    def dense(self, C, T):
        """(S, C, T) contiguous tensor view/copy (zero-pad mode only)."""
        assert self.pad_mode == K.PAD_ZERO
        return self.t[:, self.coff:self.coff + C, self.padL:self.padL + T]


LOSS_TYPES = {"L1Loss": K.LOSS_L1, "MSELoss": K.LOSS_MSE, "BCEWithLogitsLoss": K.LOSS_BCE}


class WorkerCtx:
    pass


def _fused_tail_ok(layers, out_conv, loss):
    """The decoder worker's pointwise tail -- [.., GDeconv1DBlock (no norm), MLPBlock(context 1)] -> Conv1d(hidden, 1, 1) with
    a plain (r = None) loss -- has a one-pass forward + backward (kernels.mlp_head1_step) when the library has the shape."""
    if loss is None or len(layers) < 2 or out_conv.out_channels != 1 or out_conv.kernel_size[0] != 1:
        return False
    if loss.get("r") not in (None, 1) or loss["name"] not in LOSS_TYPES:
        return False
    mlp, prev = layers[-1], layers[-2]
    if hasattr(mlp, "deconv") or not hasattr(prev, "deconv") or getattr(mlp, "context", 0) != 1:
        return False
    import os
    return os.environ.get("PASE_MLP_HEAD1", "1") != "0"


def worker_forward(layers, out_conv, a: Act, *, loss=None, want_pred=True, max_wg=0, sink=None):
    """Forward of a Minion.  layers: list of GDeconv1DBlock / MLPBlock containers; out_conv: the final
    nn.Conv1d(hidden, num_outputs*r, 1).  loss: None or dict(name=<nn loss name>, r=<int|None>,
    target=<tensor>, weight=<float>) -> fused loss + d(loss*weight)/d(pred).
    Returns ctx with .pred (if materialised), .loss_acc (sum of per-element losses, float64[1]),
    .numel, .dpred."""
    ctx = WorkerCtx()
    ctx.recs = []
    ctx.fused_tail = None
    cur = a
    fuse = sink is not None and _fused_tail_ok(layers, out_conv, loss)
    for blk in (layers[:-1] if fuse else layers):
        if hasattr(blk, "deconv"):
            dc = blk.deconv
            z = deconv_fwd(cur, dc.weight, dc.bias, Cout=dc.out_channels, k=blk.kwidth, stride=blk.stride, max_wg=max_wg)
            ctx.recs.append(("deconv", blk, cur, z))
            cur = Act(z, C=dc.out_channels, alpha=blk.act.weight)
        else:
            k = blk.context
            z, _ = conv_fwd(cur, blk.W.weight.view(blk.fmaps, -1), blk.W.bias, Cout=blk.fmaps, taps=k, padL=k // 2,
                            padR=k // 2, pad_mode=K.PAD_ZERO, max_wg=max_wg)
            ctx.recs.append(("conv", blk, cur, z))
            cur = Act(z, C=blk.fmaps, alpha=blk.act.weight)
    if fuse:
        mlp = layers[-1]
        H = mlp.fmaps
        if (cur.scale is None and cur.coff == 0 and cur.ctot == cur.C and cur.t.is_contiguous()
                and K.mlp_head1_supported(S=cur.S, C_=cur.C, T=cur.T, H=H)):
            # forward, loss and backward of the tail in one pass over the deconvolution's output: its data gradient (dz of
            # the last GDeconv1DBlock), the tail's parameter gradients straight into the sink's buffers
            B, T, C = cur.S, cur.T, cur.C
            ctx.last = cur
            ctx.head1 = False
            ctx.numel = B * T
            ctx.dpred = None
            ctx.loss_acc = loss["acc"] if loss.get("acc") is not None else _zeros((1,), cur.t, torch.float64)
            ctx.pred = _new((B, 1, T), cur.t) if want_pred else None
            dz = _new((B, C, T), cur.t)
            sums0 = _zeros((C, 3), cur.t, torch.float64)
            sums1 = _zeros((3 * H + 1,), cur.t, torch.float64)
            K.mlp_head1_step(cur.t, cur.alpha, mlp.W.weight.view(H, C), mlp.W.bias, mlp.act.weight,
                             out_conv.weight.view(-1), out_conv.bias, loss["target"].contiguous(), ctx.pred, dz,
                             ctx.loss_acc, sums0, sums1, sink.buf(mlp.W.weight).view(H, C), S=B, C_=C, T=T, H=H,
                             loss_type=LOSS_TYPES[loss["name"]], grad_scale=float(loss.get("weight", 1.0)) / ctx.numel,
                             max_wg=max_wg)
            ctx.fused_tail = (mlp, dz, sums0, sums1)
            return ctx
        # (no one-pass form for this shape: the tail's layers one by one)
        blk = layers[-1]
        k = blk.context
        z, _ = conv_fwd(cur, blk.W.weight.view(blk.fmaps, -1), blk.W.bias, Cout=blk.fmaps, taps=k, padL=k // 2,
                        padR=k // 2, pad_mode=K.PAD_ZERO, max_wg=max_wg)
        ctx.recs.append(("conv", blk, cur, z))
        cur = Act(z, C=blk.fmaps, alpha=blk.act.weight)
    ctx.last = cur
    nout = out_conv.out_channels
    B, T = cur.S, cur.T
    ctx.numel = B * nout * T
    ctx.pred = None
    ctx.dpred = None
    ctx.loss_acc = None
    w2d = out_conv.weight.view(nout, -1)
    if out_conv.kernel_size[0] != 1:
        raise NotImplementedError("pase_amd worker: output conv with context > 1")
    if loss is not None:
        ltype = LOSS_TYPES[loss["name"]]
        r = loss.get("r")
        gscale = float(loss.get("weight", 1.0)) / ctx.numel
        ctx.loss_acc = loss["acc"] if loss.get("acc") is not None else _zeros((1,), cur.t, torch.float64)
        ctx.dpred = _new((B, nout, T), cur.t)
    if nout == 1 and cur.scale is None and cur.coff == 0 and cur.ctot == cur.C:
        # single-output head: streaming kernel with the loss fused
        if want_pred:
            ctx.pred = _new((B, 1, T), cur.t)
        if loss is not None:
            if r not in (None, 1):
                raise NotImplementedError("pase_amd worker: r-context loss on a 1-output head")
            K.head1_fwd(cur.t, w2d.view(-1), out_conv.bias, S=B, C_=cur.C, T=T, in_alpha=cur.alpha,
                        target=loss["target"].contiguous(), y=ctx.pred, dy=ctx.dpred, loss_acc=ctx.loss_acc,
                        loss_type=ltype, grad_scale=gscale)
        else:
            K.head1_fwd(cur.t, w2d.view(-1), out_conv.bias, S=B, C_=cur.C, T=T, in_alpha=cur.alpha, y=ctx.pred)
        ctx.head1 = True
        return ctx
    ctx.head1 = False
    if loss is not None and ltype == K.LOSS_MSE and r not in (None, 1):
        # projection GEMM with the r-context MSE fused in the epilogue (prediction never stored
        # unless asked for)
        if want_pred:
            ctx.pred = _new((B, nout, T), cur.t)
        tgt = loss["target"].contiguous()
        K.conv_gemm(cur.t, w2d, ctx.pred, S=B, Cin=cur.C, Tin=T, M=nout, K=cur.C, taps=1, Ncols=T, Tout=T,
                    bias=out_conv.bias, in_scale=cur.scale, in_shift=cur.shift, in_alpha=cur.alpha,
                    x_ctot=cur.ctot, x_coff=cur.coff, epilogue=K.EPI_MSE_CTX, label=tgt, grad_out=ctx.dpred,
                    loss_acc=ctx.loss_acc, grad_scale=2.0 * gscale, r_ctx=r, label_D=tgt.shape[1], max_wg=max_wg)
        return ctx
    pred, _ = conv_fwd(cur, w2d, out_conv.bias, Cout=nout, taps=1, Tout=T, max_wg=max_wg)
    ctx.pred = pred
    if loss is not None:
        tgt = loss["target"].contiguous()
        rr = r if r not in (None, 1) else 0
        K.ctx_loss(pred, tgt, ctx.dpred, ctx.loss_acc, B=B, M=nout, F=T, r_ctx=rr, label_D=tgt.shape[1],
                   loss_type=ltype, grad_scale=gscale)
    return ctx


def worker_backward(layers, out_conv, ctx, dpred, sink, need_dinput=True, max_wg=0):
    """Backward of a Minion from dpred = d(loss)/d(pred).  Returns GradSrc for the worker input."""
    cur = ctx.last
    B, T = cur.S, cur.T
    nout = out_conv.out_channels
    x = cur.t
    if getattr(ctx, "fused_tail", None) is not None:
        mlp, dz, sums0, sums1 = ctx.fused_tail
        H = mlp.fmaps
        sink.add_cols(sums1, 3, H, [(out_conv.weight, 0), (mlp.act.weight, 1), (mlp.W.bias, 2)])
        sink.add_cols(sums1[H * 3:], 1, 1, [(out_conv.bias, 0)])
        psums, pcols = sums0, (2, 0)            # (dalpha, sum dz) columns for the deconvolution below
        have_dz = True
    elif ctx.head1:
        C = cur.C
        sums = _zeros((C * 3 + 1,), x, torch.float64)
        dz = _new((B, C, T), x)
        K.head1_bwd(cur.t, cur.alpha, out_conv.weight.view(-1), dpred.contiguous(), dz, sums, S=B, C_=C, T=T)
        sink.add_cols(sums, 3, C, [(out_conv.weight, 0)])
        sink.add_cols(sums[C * 3:], 1, 1, [(out_conv.bias, 0)])
        psums, pcols = sums, (1, 2)             # (dalpha, sum dz) columns for the layer below
        have_dz = True
    else:
        dpred = dpred.contiguous()
        dw_out, db_out = sink.buf(out_conv.weight).view(nout, -1), sink.buf(out_conv.bias)
        on_wgrad_stream(x, lambda: conv_wgrad(dpred, cur, dw_out, db_out, taps=1, max_wg=max_wg), keep=(dpred, cur.t))
        dsrc = GradSrc(conv_dgrad(dpred, out_conv.weight, R=nout, O=cur.C, k=1, stride=1, Tin=T, padL=0, padR=0,
                                  s_red=cur.C, s_out=1, s_k=1, max_wg=max_wg), ctot=cur.C, Tp=T)
        have_dz = False
    for kind, blk, inp, z in reversed(ctx.recs):
        C, Tz = z.shape[1], z.shape[2]
        if not have_dz:
            dz, sums = act_backward(z, C=C, T=Tz, S=B, has_bn=False, alpha=blk.act.weight, dsrc=dsrc.t,
                                    dsrc_ctot=dsrc.ctot, dsrc_coff=dsrc.coff, Tp=dsrc.Tp, padL=dsrc.padL,
                                    pad_mode=dsrc.pad_mode)
            psums, pcols = sums, (2, 0)
        have_dz = False
        # PReLU slope gradient + the bias gradient (= sum dz) of this layer's conv, one launch
        sink.add_cols(psums, 3, C, [(blk.act.weight, pcols[0]), ((blk.deconv if kind == "deconv" else blk.W).bias, pcols[1])])
        if kind == "deconv":
            dc = blk.deconv
            k, st = blk.kwidth, blk.stride
            pad = max(0, (st - k) // -2)
            cin = inp.C
            # dW[ci, co, kk] = sum_{s,t} act(in)[s,ci,t] * dz[s,co,t*st + kk - pad]
            if inp.scale is not None:
                raise NotImplementedError("deconv wgrad with an affine on-load input")
            dw_dc = sink.buf(dc.weight).view(cin, -1)

            def wd(inp=inp, dz=dz, dw_dc=dw_dc, cin=cin, C=C, Tz=Tz, k=k, st=st, pad=pad):
                K.wgrad_gemm(inp.t, dz, dw_dc, S=B, M=cin, Tg=inp.T, Ncols=inp.T, Cin=C,
                             Tz=Tz, taps=k, ldw=C * k, g_ctot=inp.ctot, g_coff=inp.coff, stride=st, tapstep=1, padL=pad,
                             pad_mode=K.PAD_ZERO, g_alpha=inp.alpha, max_wg=max_wg)
            on_wgrad_stream(x, wd, keep=(inp.t, dz))
            last = blk is layers[0]
            if need_dinput or not last:
                din, _ = conv_fwd(Act(dz, C=C), dc.weight.view(cin, -1), None, Cout=cin, taps=k, stride=st, padL=pad,
                                  padR=pad, pad_mode=K.PAD_ZERO, Tout=inp.T, max_wg=max_wg)
                dsrc = GradSrc(din, ctot=cin, Tp=inp.T)
        else:
            k = blk.context
            cin = inp.C
            dw_blk = sink.buf(blk.W.weight).view(C, -1)
            on_wgrad_stream(x, lambda dz=dz, inp=inp, dw_blk=dw_blk, k=k: conv_wgrad(
                dz, inp, dw_blk, None, taps=k, padL=k // 2, pad_mode=K.PAD_ZERO, max_wg=max_wg),
                keep=(dz, inp.t))
            last = blk is layers[0]
            if need_dinput or not last:
                din = conv_dgrad(dz, blk.W.weight, R=C, O=cin, k=k, stride=1, Tin=inp.T, padL=k // 2, padR=k // 2,
                                 s_red=cin * k, s_out=k, s_k=1, max_wg=max_wg)
                dsrc = GradSrc(din, ctot=cin, Tp=din.shape[2], padL=k // 2)
    if len(ctx.recs) == 0 and ctx.head1:
        raise NotImplementedError("1-output worker without hidden layers")
    join_wgrad_stream(x)
    return dsrc if need_dinput else None


def mlp_group_step(workers, a: Act, targets, sink, accs=None, max_wg=0):
    """Forward + loss + backward of several one-hidden-layer MLP regression workers that read the
    SAME input (the chunk embedding): their first layers run as ONE stacked GEMM (9 x (256->256) ->
    one 2304-row launch instead of nine 100-workgroup launches), their input gradient as ONE dgrad,
    their first-layer weight gradients as ONE wgrad.  Heads (hidden -> num_outputs*r, fused r-context
    MSE) stay per worker.  Returns ({name: (loss_acc, numel)}, d(input) as a dense (B, Cin, F) tensor).
    Reference: MLPMinion.forward (Minions/minions.py:512-528) x N + ContextualizedLoss."""
    B, F_, cin = a.S, a.T, a.C
    x = a.t
    hs = [w.blocks[0].fmaps for w in workers]
    offs = [sum(hs[:i]) for i in range(len(hs))]
    htot = sum(hs)
    w1cat = torch.cat([w.blocks[0].W.weight.view(h, -1) for w, h in zip(workers, hs)], dim=0)
    b1cat = torch.cat([w.blocks[0].W.bias for w in workers], dim=0)
    z_all, _ = conv_fwd(a, w1cat, b1cat, Cout=htot, taps=1, Tout=F_, max_wg=max_wg)
    dz_all = _new((B, htot, F_), x)
    out = {}

    def head(w, h, off):
        """one worker's head: projection + fused loss, weight / data gradients, PReLU backward into dz_all"""
        blk, oc = w.blocks[0], w.W
        nout = oc.out_channels
        loss = w.loss
        cur = Act(z_all, C=h, coff=off, alpha=blk.act.weight)
        numel = B * nout * F_
        gscale = float(w.loss_weight) / numel
        acc = accs[w.name] if accs is not None else _zeros((1,), x, torch.float64)
        dpred = _new((B, nout, F_), x)
        tgt = targets[w.name].contiguous()
        w2d = oc.weight.view(nout, -1)
        r = loss.r
        if loss.loss_name == "MSELoss" and r not in (None, 1):
            K.conv_gemm(z_all, w2d, None, S=B, Cin=h, Tin=F_, M=nout, K=h, taps=1, Ncols=F_, Tout=F_, bias=oc.bias,
                        in_alpha=cur.alpha, x_ctot=htot, x_coff=off, epilogue=K.EPI_MSE_CTX, label=tgt,
                        grad_out=dpred, loss_acc=acc, grad_scale=2.0 * gscale, r_ctx=r, label_D=tgt.shape[1], max_wg=max_wg)
        else:
            pred, _ = conv_fwd(cur, w2d, oc.bias, Cout=nout, taps=1, Tout=F_, max_wg=max_wg)
            K.ctx_loss(pred, tgt, dpred, acc, B=B, M=nout, F=F_, r_ctx=(r if r not in (None, 1) else 0),
                       label_D=tgt.shape[1], loss_type=LOSS_TYPES[loss.loss_name], grad_scale=gscale)
        out[w.name] = (acc, numel)
        # head backward (in line: forking the wide heads' weight gradients to the weight-gradient stream measured +0.9 ms per
        # step -- the narrow heads already run underneath the wide ones)
        conv_wgrad(dpred, cur, sink.buf(oc.weight).view(nout, -1), sink.buf(oc.bias), taps=1, max_wg=max_wg)
        dA = conv_dgrad(dpred, oc.weight, R=nout, O=h, k=1, stride=1, Tin=F_, padL=0, padR=0, s_red=h, s_out=1, s_k=1, max_wg=max_wg)
        _, sums = act_backward(z_all, C=h, T=F_, S=B, has_bn=False, alpha=blk.act.weight, dsrc=dA, dsrc_ctot=h, Tp=F_,
                               y_ctot=htot, y_coff=off, dy_out=dz_all)
        sink.add_cols(sums, 3, h, [(blk.act.weight, 2), (blk.W.bias, 0)])

    # The wide heads (LPS: 21 525 rows) fill the chip on their own; the narrow ones (84 ... 840 rows) launch
    # 10-100 workgroups each and would leave most of the 256 CUs idle if serialised behind each other: they go to
    # side HIP streams and run underneath the wide ones (fork after the stacked first layer, join before the stacked
    # backward).  Every head writes disjoint slices (its own parameter gradients, its channel range of dz_all).
    items = list(zip(workers, hs, offs))
    narrow = [it for it in items if it[0].W.out_channels * F_ * B < _WIDE_HEAD_ELEMS]
    side = side_streams(x, 3) if (len(narrow) > 1 and K.GEMM_TIMER is None) else []
    if side:
        main = torch.cuda.current_stream()
        fork = main.record_event()
        for i, it in enumerate(narrow):
            st = side[i % len(side)]
            if i < len(side):
                st.wait_event(fork)
            with torch.cuda.stream(st):
                head(*it)
        for it in items:
            if not any(it is n for n in narrow):
                head(*it)
        for st in side[:len(narrow)]:
            main.wait_event(st.record_event())
    else:
        for it in items:
            head(*it)
    # stacked first layer: one wgrad, one dgrad
    dw1 = _zeros((htot, cin), x)

    conv_wgrad(dz_all, a, dw1, None, taps=1, max_wg=max_wg)
    sink.add_many([(w.blocks[0].W.weight, dw1[off:off + h]) for w, h, off in zip(workers, hs, offs)])
    dx = conv_dgrad(dz_all, w1cat, R=htot, O=cin, k=1, stride=1, Tin=F_, padL=0, padR=0, s_red=cin, s_out=1, s_k=1, max_wg=max_wg)
    return out, dx

"""CPU oracle: a functional, plain-torch-fp32 restatement of the reference PASE / PASE+ hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; nothing under pase_amd/ does.  It is the checker, never the product.

Parity pinning: this restatement is pinned against the LIVE reference modules imported from
/root/reference (through oracle/ref_shim.py) by tests/test_oracle_pins.py when the reference tree
is present, and against golden vectors generated from the live reference by
oracle/make_golden.py and committed under tests/golden/ (those travel to the GPU box).
The QRNN is third-party (salesforce/pytorch-qrnn, un-vendored and un-pinned, requirements.txt:16);
its semantics are restated from the published upstream algorithm => that sub-path's parity is
"unpinned" in the sense of SURVEY.md section 8c (no reference test, checkpoint or vector exercises it).

Everything is a function of (params: dict name -> tensor, cfg: dict) with the reference's
state_dict names; gradients come from torch.autograd on these functions.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

WAVEFE_DEFAULTS = dict(
    num_inputs=1, sincnet=True, kwidths=[251, 10, 5, 5, 5, 5, 5, 5], strides=[1, 10, 2, 1, 2, 1, 2, 2],
    fmaps=[64, 64, 128, 128, 256, 256, 512, 512], norm_type="bnorm", sr=16000, emb_dim=256, rnn_dim=None,
    rnn_pool=False, rnn_layers=1, norm_out=False, denseskips=False)


def full_cfg(cfg):
    c = dict(WAVEFE_DEFAULTS)
    c.update(cfg)
    return c


# ---------------------------------------------------------------------------------------------
# SincConv_fast  (pase/models/modules.py:818-934)
# ---------------------------------------------------------------------------------------------
def sinc_constants(K=251, sr=16000):
    """window_ and n_ exactly as modules.py:868-876."""
    n_lin = torch.linspace(0, (K / 2) - 1, steps=int(K / 2))
    window = 0.54 - 0.46 * torch.cos(2 * math.pi * n_lin / K)
    n = (K - 1) / 2.0
    n_ = 2 * math.pi * torch.arange(-n, 0).view(1, -1) / sr
    return window, n_


def sinc_init(out_channels=64, sr=16000, min_low=50, min_band=50):
    """mel-spaced initial (low_hz_, band_hz_) of modules.py:852-866."""
    to_mel = lambda hz: 2595 * np.log10(1 + hz / 700)
    to_hz = lambda mel: 700 * (10 ** (mel / 2595) - 1)
    hz = to_hz(np.linspace(to_mel(30), to_mel(sr / 2 - (min_low + min_band)), out_channels + 1))
    return torch.Tensor(hz[:-1]).view(-1, 1), torch.Tensor(np.diff(hz)).view(-1, 1)


def sinc_filters(low_hz_, band_hz_, K=251, sr=16000, min_low=50, min_band=50):
    """(C,1,K) band-pass bank, modules.py:895-915."""
    window, n_ = sinc_constants(K, sr)
    window, n_ = window.to(low_hz_), n_.to(low_hz_)          # device AND dtype (an fp64 evaluation uses the same constants)
    low = min_low + torch.abs(low_hz_)
    high = torch.clamp(low + min_band + torch.abs(band_hz_), min_low, sr / 2)
    band = (high - low)[:, 0]
    left = ((torch.sin(torch.matmul(high, n_)) - torch.sin(torch.matmul(low, n_))) / (n_ / 2)) * window
    centre = 2 * band.view(-1, 1)
    bp = torch.cat([left, centre, torch.flip(left, dims=[1])], dim=1)
    bp = bp / (2 * band[:, None])
    return bp.view(low_hz_.shape[0], 1, K)


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
def batch_norm(x, prefix, P, training, stats_out=None, affine=True, eps=1e-5, momentum=0.1):
    """nn.BatchNorm1d over (B,C,T).  In training mode uses batch stats and reports the updated
    running stats into stats_out (functional: P is not mutated)."""
    rm, rv = P[prefix + ".running_mean"], P[prefix + ".running_var"]
    w = P[prefix + ".weight"] if affine else None
    b = P[prefix + ".bias"] if affine else None
    if training:
        rm2, rv2 = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, momentum, eps)
        if stats_out is not None:
            stats_out[prefix + ".running_mean"] = rm2
            stats_out[prefix + ".running_var"] = rv2
        return y
    return F.batch_norm(x, rm, rv, w, b, False, momentum, eps)


def prelu(x, alpha):
    return F.prelu(x, alpha)


def fe_pad(x, k, stride):
    """FeBlock.forward reflect padding, modules.py:1059-1071 (dilation 1)."""
    if k <= 1:
        return x
    if stride > 1 or k % 2 == 0:
        P = (k // 2 - 1, k // 2)
    else:
        P = (k // 2, k // 2)
    return F.pad(x, P, mode="reflect")


def qrnn_layer(x, W, b):
    """salesforce/pytorch-qrnn QRNNLayer(window=2) + ForgetMult on (B,C,T) input; returns (B,H,T)."""
    xm1 = torch.cat([torch.zeros_like(x[:, :, :1]), x[:, :, :-1]], 2)
    src = torch.cat([x, xm1], 1)                     # channel order [x_t ; x_{t-1}]
    Y = torch.einsum("ok,bkt->bot", W, src) + b[None, :, None]
    Z, Fg, O = Y.chunk(3, dim=1)
    Z, Fg, O = torch.tanh(Z), torch.sigmoid(Fg), torch.sigmoid(O)
    cs = []
    c = None
    for t in range(x.shape[2]):
        ct = Fg[:, :, t] * Z[:, :, t]
        if c is not None:
            ct = ct + (1 - Fg[:, :, t]) * c
        cs.append(ct)
        c = ct
    C = torch.stack(cs, 2)
    return O * C


# ---------------------------------------------------------------------------------------------
# WaveFe.forward  (pase/models/frontend.py:234-279)
# ---------------------------------------------------------------------------------------------
def encoder_forward(P, cfg, x, training=True, stats_out=None, taps=None):
    """x (B,1,T) -> (B,emb,T/160).  `taps` (dict) optionally receives intermediate activations."""
    cfg = full_cfg(cfg)
    h = x
    skips = []
    nb = len(cfg["kwidths"])
    nt = cfg["norm_type"]
    if nt not in ("bnorm", "lnorm", "inorm", "affinorm", None):
        raise NotImplementedError(nt)
    for n in range(nb):
        k, st = cfg["kwidths"][n], cfg["strides"][n]
        pre = "blocks.%d." % n
        if n == 0 and cfg["sincnet"]:
            kk = k + 1 if k % 2 == 0 else k
            filt = sinc_filters(P[pre + "conv.low_hz_"], P[pre + "conv.band_hz_"], kk, cfg["sr"])
            pad = (kk // 2 - 1, kk // 2) if st > 1 else (kk // 2, kk // 2)
            h = F.conv1d(F.pad(h, pad, mode="reflect"), filt, stride=st)
        else:
            h = F.conv1d(fe_pad(h, k, st), P[pre + "conv.weight"], P[pre + "conv.bias"], stride=st)
        if taps is not None:
            taps["conv%d" % n] = h
        # build_norm_layer / forward_norm (modules.py:77-109)
        if nt == "bnorm":
            h = batch_norm(h, pre + "norm", P, training, stats_out)
        elif nt == "lnorm":               # nn.LayerNorm(C) on the (B, T, C) transpose
            h = F.layer_norm(h.transpose(1, 2), (h.shape[1],), P[pre + "norm.weight"], P[pre + "norm.bias"],
                             1e-5).transpose(1, 2)
        elif nt == "inorm":               # nn.InstanceNorm1d(C, affine=False)
            h = F.instance_norm(h, eps=1e-5)
        elif nt == "affinorm":
            h = F.instance_norm(h, weight=P[pre + "norm.weight"], bias=P[pre + "norm.bias"], eps=1e-5)
        h = prelu(h, P[pre + "act.weight"])
        if cfg["denseskips"] and n + 1 < nb:
            skips.append(F.conv1d(h, P["denseskips.%d.weight" % n]))
    if cfg["rnn_pool"]:
        for l in range(cfg["rnn_layers"]):
            h = qrnn_layer(h, P["rnn.layers.%d.linear.weight" % l], P["rnn.layers.%d.linear.bias" % l])
        if taps is not None:
            taps["rnn"] = h
    y = F.conv1d(h, P["W.weight"], P["W.bias"])
    for s in skips:                                   # fuse_skip, densemerge='sum' (frontend.py:213-232)
        d = s.shape[2] // y.shape[2]
        if d > 1:
            s = s[:, :, :y.shape[2] * d]
            s = s.view(s.shape[0], s.shape[1], s.shape[2] // d, d).mean(3)
        y = y + s
    if taps is not None:
        taps["pre_norm"] = y
    if cfg["norm_out"]:                   # frontend.py:206-210
        if nt == "bnorm":
            y = batch_norm(y, "norm_out", P, training, stats_out, affine=False)
        else:
            y = F.instance_norm(y, eps=1e-5)
    return y


def encoder_forward_batch(P, cfg, batch, training=True, stats_out=None):
    """dict batch -> (tuple of embeddings, chunk embedding)  (modules.py:16-43)."""
    keys = [k for k in ["chunk", "chunk_ctxt", "chunk_rand", "cchunk"] if k in batch]
    x = torch.cat([batch[k] for k in keys], 0)
    y = encoder_forward(P, cfg, x, training, stats_out)
    h = torch.chunk(y, len(keys), 0)
    return h, h[0]


def select_output(h, mode=None):
    """modules.py:62-74."""
    if mode == "avg_norm":
        return h - h.mean(2, keepdim=True)
    if mode == "avg_concat":
        return torch.cat((h, h.mean(2, keepdim=True).repeat(1, 1, h.shape[-1])), 1)
    if mode == "avg_norm_concat":
        g = h.mean(2, keepdim=True)
        return torch.cat((h - g, g.repeat(1, 1, h.shape[-1])), 1)
    return h


# ---------------------------------------------------------------------------------------------
# workers  (pase/models/Minions/minions.py, cls_minions.py) and losses (pase/losses.py)
# ---------------------------------------------------------------------------------------------
def contextualize_r(t, r):
    """ContextualizedLoss.contextualize_r (losses.py:14-31): (B,D,F) -> (B,D*r,F), channel d*r+j
    holds frame t+j-r//2 (zero outside)."""
    if r is None:
        return t
    p = F.pad(t, (r // 2, r // 2))
    B = t.shape[0]
    return torch.cat([p[:, :, i:i + r].contiguous().view(B, -1).unsqueeze(2) for i in range(p.size(2) - (r - 1))], 2)


def ctx_loss(pred, target, loss_name, r=None):
    tg = contextualize_r(target, r)
    if loss_name == "MSELoss":
        return F.mse_loss(pred, tg)
    if loss_name == "L1Loss":
        return F.l1_loss(pred, tg)
    if loss_name == "BCEWithLogitsLoss":
        return F.binary_cross_entropy_with_logits(pred, tg)
    raise ValueError(loss_name)


def mlp_minion(P, pre, x, hidden_layers=1):
    """MLPMinion.forward (minions.py:512-528): [Conv1d(k=1) -> PReLU] x hidden_layers -> Conv1d(k=1)."""
    h = x
    for i in range(hidden_layers):
        h = F.conv1d(h, P[pre + "blocks.%d.W.weight" % i], P[pre + "blocks.%d.W.bias" % i])
        h = prelu(h, P[pre + "blocks.%d.act.weight" % i])
    return F.conv1d(h, P[pre + "W.weight"], P[pre + "W.bias"])


def decoder_minion(P, pre, x, strides, kwidths, hidden_layers=1):
    """DecoderMinion.forward (minions.py:420-449): GDeconv1DBlock x len(strides) -> MLPBlock x
    hidden_layers -> Conv1d(hidden, num_outputs, 1)."""
    h = x
    nb = len(strides)
    for i, (st, k) in enumerate(zip(strides, kwidths)):
        pad = max(0, (st - k) // -2)
        h = F.conv_transpose1d(h, P[pre + "blocks.%d.deconv.weight" % i], P[pre + "blocks.%d.deconv.bias" % i],
                               stride=st, padding=pad)
        if (st % 2 != 0 and k % 2 == 0) or (st % 2 == 0 and k % 2 != 0):
            h = h[:, :, :-1]
        h = prelu(h, P[pre + "blocks.%d.act.weight" % i])
    for j in range(hidden_layers):
        i = nb + j
        h = F.conv1d(h, P[pre + "blocks.%d.W.weight" % i], P[pre + "blocks.%d.W.bias" % i])
        h = prelu(h, P[pre + "blocks.%d.act.weight" % i])
    return F.conv1d(h, P[pre + "W.weight"], P[pre + "W.bias"])


def make_samples(h, augment):
    """cls_minions.py:29-43."""
    pos = torch.cat((h[0], h[1]), 1)
    neg = torch.cat((h[0], h[2]), 1)
    if augment:
        pos = torch.cat((pos, torch.cat((h[1], h[0]), 1)), 0)
        neg = torch.cat((neg, torch.cat((h[1], h[2]), 1)), 0)
    return pos, neg


def spc_samples(x, ctxt_frames=5, seq_pad=16):
    """SPCMinion.forward sampling + gathering (Minions/minions.py:606-640): three `random.choice`
    draws (anchor t, future start, past end), returns the (2B, (N+1)*C, 1) MLP input."""
    import random
    N, M = ctxt_frames, seq_pad + ctxt_frames
    T, bsz = x.size(2), x.size(0)
    t = random.choice(list(range(M + 1, T - M)))
    future_t = random.choice(list(range(t + seq_pad, T - N)))
    past_t = random.choice(list(range(N, t - seq_pad)))
    future = x[:, :, future_t:future_t + N].contiguous().view(bsz, -1)
    past = x[:, :, past_t - N:past_t].contiguous().view(bsz, -1)
    current = x[:, :, t].contiguous()
    pos = torch.cat((current, future), 1)
    neg = torch.cat((current, past), 1)
    return torch.cat((pos, neg), 0).unsqueeze(2)


def gap_samples(x):
    """GapMinion.forward sampling (Minions/minions.py:672-693): two np.random.randint draws of size B, the
    frame pair concatenated on channels, label = LongTensor(|a-b|/(T-1)) (truncation as written)."""
    import numpy as np
    B, T = x.size(0), x.size(2)
    aidx = np.random.randint(0, T, size=B)
    bidx = np.random.randint(0, T, size=B)
    xa = torch.stack([x[i, :, int(a)] for i, a in enumerate(aidx)], 0)
    xb = torch.stack([x[i, :, int(b)] for i, b in enumerate(bidx)], 0)
    dists = torch.tensor([float(int(abs(int(a) - int(b)) / (T - 1))) for a, b in zip(aidx, bidx)])
    return torch.cat((xa, xb), 1).unsqueeze(2), dists.view(-1, 1, 1).to(x.device)


def make_labels(y):
    """cls_minions.py:47-51."""
    bsz, slen = y.size(0) // 2, y.size(2)
    return torch.cat((torch.ones(bsz, 1, slen, device=y.device), torch.zeros(bsz, 1, slen, device=y.device)), 0)


def pase_forward(P, fe_cfg, workers_cfg, batch, training=True, stats_out=None):
    """pase.forward (pase/models/pase.py:310-356) for mlp / decoder regression workers and the
    mi (LIM) / cmi (GIM) contrastive workers.  Returns (h, chunk, preds, labels)."""
    xb = {k: v for k, v in batch.items() if k != "cchunk"}      # no regularizer workers: cchunk not encoded
    h, chunk = encoder_forward_batch({k[len("frontend."):]: v for k, v in P.items() if k.startswith("frontend.")},
                                     fe_cfg, xb, training, stats_out)
    preds, labels = {}, {}
    for i, w in enumerate(workers_cfg.get("regr", [])):
        pre = "regression_workers.%d." % i
        if w.get("type", "mlp") == "decoder":
            preds[w["name"]] = decoder_minion(P, pre, chunk, w["strides"], w["kwidths"], w.get("hidden_layers", 2))
        else:
            preds[w["name"]] = mlp_minion(P, pre, chunk, w.get("hidden_layers", 2))
        labels[w["name"]] = batch[w["name"]]
    for i, w in enumerate(workers_cfg.get("cls", [])):
        pre = "classification_workers.%d.minion." % i
        if w["name"] == "gap":
            x, lab = gap_samples(chunk)
            preds[w["name"]] = mlp_minion(P, pre, x, w.get("hidden_layers", 2))
            labels[w["name"]] = lab
            continue
        if w["name"] == "spc":
            x = spc_samples(chunk, w.get("ctxt_frames", 5), w.get("seq_pad", 16))
        else:
            pos, neg = make_samples(h, w.get("augment", False))
            x = torch.cat((pos, neg), 0)
            if w["name"] == "cmi":
                x = x.mean(2, keepdim=True)
            elif w["name"] != "mi":
                raise NotImplementedError(w["name"])
        y = mlp_minion(P, pre, x, w.get("hidden_layers", 2))
        preds[w["name"]] = y
        labels[w["name"]] = make_labels(y)
    return h, chunk, preds, labels


def pase_losses(workers_cfg, preds, labels):
    """_base_scheduler's loss dict (WorkerScheduler/worker_scheduler.py:43-62): loss_weight * loss,
    plus 'total'."""
    losses = {}
    tot = 0
    for grp in ("cls", "regr"):
        for w in workers_cfg.get(grp, []):
            l = w.get("loss_weight", 1.0) * ctx_loss(preds[w["name"]], labels[w["name"]], w["loss"], w.get("r"))
            losses[w["name"]] = l
            tot = tot + l
    losses["total"] = tot
    return losses


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update (defaults), returns new (p, m, v)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v

"""Host-side mirror of the pieces of pase/models/modules.py that the PASE / PASE+ path uses
(reference file:line in each docstring).  These classes are *parameter containers with the
reference's names, shapes, registration order and initialisers* -- so `state_dict()` round-trips
with reference checkpoints and `torch.manual_seed(s)` gives the same initial weights -- but their
arithmetic is done by the HIP kernels through `pase_amd.engine`, never by torch ops.
"""
import json
import math
import os

import numpy as np
import torch
import torch.nn as nn


# ----------------------------------------------------------------------------------------------
# batch plumbing: pase/models/modules.py:16-43,62-74
# ----------------------------------------------------------------------------------------------
def format_frontend_chunk(batch, device="cpu"):
    """dict -> cat([chunk, chunk_ctxt, chunk_rand(, cchunk)], 0), data_fmt = #parts;
    tensor -> (tensor, 0).  (modules.py:16-31; note the reference's `'chunk_ctxt' and
    'chunk_rand' in batch` only tests for 'chunk_rand')."""
    if type(batch) == dict:
        if "chunk_rand" in batch:
            keys = ["chunk", "chunk_ctxt", "chunk_rand", "cchunk"]
            parts = [batch[k] for k in keys if k in batch]
            x = torch.cat(parts, dim=0).to(device)
            return x, len(parts)
        return batch["chunk"].to(device), 1
    return batch, 0


def select_output(h, mode=None):
    """modules.py:62-74."""
    if mode == "avg_norm":
        return h - torch.mean(h, dim=2, keepdim=True)
    if mode == "avg_concat":
        g = torch.mean(h, dim=2, keepdim=True).repeat(1, 1, h.shape[-1])
        return torch.cat((h, g), dim=1)
    if mode == "avg_norm_concat":
        g = torch.mean(h, dim=2, keepdim=True)
        h = h - g
        return torch.cat((h, g.repeat(1, 1, h.shape[-1])), dim=1)
    return h


def format_frontend_output(y, data_fmt, mode):
    """modules.py:33-43."""
    if data_fmt > 1:
        embedding = torch.chunk(y, data_fmt, dim=0)
        return embedding, embedding[0]
    if data_fmt == 1:
        return y, y
    return select_output(y, mode=mode)


# ----------------------------------------------------------------------------------------------
# checkpoints: Saver (modules.py:151-301) and Model (modules.py:304-373)
# ----------------------------------------------------------------------------------------------
class Saver(object):
    """weights_{prefix}{name}-{step}.ckpt = {'step','state_dict'[,'optimizer']} + a JSON index
    `{prefix}checkpoints` = {'latest': [...], 'current': ...} with max_ckpts rotation."""

    def __init__(self, model, save_path, max_ckpts=5, optimizer=None, prefix=""):
        self.model = model
        self.save_path = save_path
        self.ckpt_path = os.path.join(save_path, "{}checkpoints".format(prefix))
        self.max_ckpts = max_ckpts
        self.optimizer = optimizer
        self.prefix = prefix

    def _index(self):
        if os.path.exists(self.ckpt_path):
            with open(self.ckpt_path, "r") as f:
                return json.load(f)
        return {"latest": [], "current": []}

    def save(self, model_name, step, best_val=False):
        os.makedirs(self.save_path, exist_ok=True)
        model_path = "{}-{}.ckpt".format(model_name, step)
        if best_val:
            model_path = "best_" + model_path
        model_path = "{}{}".format(self.prefix, model_path)
        ckpts = self._index()
        latest = ckpts["latest"]
        if len(latest) > 0 and self.max_ckpts is not None and len(latest) > self.max_ckpts:
            # modules.py:181-191: the oldest entry leaves the index only when its file could be removed
            fn = os.path.join(self.save_path, "weights_" + latest[0])
            try:
                os.remove(fn)
                latest = latest[1:]
            except FileNotFoundError:
                print("ERROR: ckpt is not there?")
        latest.append(model_path)
        ckpts["latest"] = latest
        ckpts["current"] = model_path
        with open(self.ckpt_path, "w") as f:
            json.dump(ckpts, f, indent=2)
        st = {"step": step, "state_dict": self.model.state_dict()}
        if self.optimizer is not None:
            st["optimizer"] = self.optimizer.state_dict()
        torch.save(st, os.path.join(self.save_path, "weights_" + model_path))

    def read_latest_checkpoint(self):
        if not os.path.exists(self.ckpt_path):
            return None
        with open(self.ckpt_path, "r") as f:
            ckpts = json.load(f)
        cur = ckpts.get("current")
        return cur if cur else None

    def load_weights(self):
        cur = self.read_latest_checkpoint()
        if cur is None:
            return False
        st = torch.load(os.path.join(self.save_path, "weights_" + cur), map_location="cpu")
        self.model.load_state_dict(st["state_dict"] if "state_dict" in st else st)
        if self.optimizer is not None and "optimizer" in st:
            self.optimizer.load_state_dict(st["optimizer"])
        return True

    def load_ckpt_step(self, curr_ckpt):
        ckpt = torch.load(os.path.join(self.save_path, "weights_" + curr_ckpt), map_location="cpu")
        return ckpt["step"]

    def load_pretrained_ckpt(self, ckpt_file, load_last=False, load_opt=True, verbose=True):
        """modules.py:267-301: keep keys that exist with the same shape; without load_last the LAST
        TWO checkpoint keys are dropped; raise ValueError if matched != model key count."""
        model_dict = self.model.state_dict()
        st_dict = torch.load(ckpt_file, map_location=lambda storage, loc: storage)
        pt_dict = st_dict["state_dict"] if "state_dict" in st_dict else st_dict
        all_keys = list(pt_dict.keys())
        allowed = all_keys[:] if load_last else all_keys[:-2]
        allowed_set = set(allowed)
        pt_dict = {k: v for k, v in pt_dict.items()
                   if k in model_dict and k in allowed_set and v.size() == model_dict[k].size()}
        if verbose:
            print("Current Model keys: ", len(model_dict))
            print("Current Pt keys: ", len(pt_dict))
            print("Loading matching keys: ", list(pt_dict.keys()))
        if len(pt_dict) != len(model_dict):
            raise ValueError("WARNING: LOADING DIFFERENT NUM OF KEYS")
        model_dict.update(pt_dict)
        self.model.load_state_dict(model_dict)
        for k in model_dict.keys():
            if k not in allowed_set:
                print("WARNING: {} weights not loaded from pt ckpt".format(k))
        if self.optimizer is not None and "optimizer" in st_dict and load_opt:
            self.optimizer.load_state_dict(st_dict["optimizer"])


class NeuralBlock(nn.Module):
    def __init__(self, name="NeuralBlock"):
        super().__init__()
        self.name = name

    def describe_params(self):
        pp = sum(p.numel() for p in self.parameters())
        print("-" * 10)
        print(self)
        print("Num params: ", pp)
        print("-" * 10)
        return pp


class Model(NeuralBlock):
    """modules.py:304-373."""

    def __init__(self, max_ckpts=5, name="BaseModel"):
        super().__init__(name=name)
        self.optim = None
        self.max_ckpts = max_ckpts

    def save(self, save_path, step, best_val=False, saver=None):
        if not hasattr(self, "saver") and saver is None:
            self.saver = Saver(self, save_path, optimizer=self.optim, prefix=self.name + "-",
                               max_ckpts=self.max_ckpts)
        (self.saver if saver is None else saver).save(self.name, step, best_val=best_val)

    def load(self, save_path):
        if os.path.isdir(save_path):
            if not hasattr(self, "saver"):
                self.saver = Saver(self, save_path, optimizer=self.optim, prefix=self.name + "-",
                                   max_ckpts=self.max_ckpts)
            self.saver.load_weights()
        else:
            print("Loading ckpt from ckpt: ", save_path)
            self.load_pretrained(save_path)

    def load_pretrained(self, ckpt_path, load_last=False, verbose=True):
        Saver(self, ".", optimizer=self.optim).load_pretrained_ckpt(ckpt_path, load_last, verbose=verbose)

    def parameters(self, recurse=True):
        return filter(lambda p: p.requires_grad, super().parameters(recurse))

    def get_total_params(self):
        return sum(p.numel() for p in self.parameters())

    def describe_params(self):
        pp = 0
        if hasattr(self, "blocks"):
            for b in self.blocks:
                pp += b.describe_params()
        else:
            print("Warning: did not find a list of blocks...")
            print("Just printing all params calculation.")
        total = self.get_total_params()
        print("{} total params: {}".format(self.name, total))
        return total


# ----------------------------------------------------------------------------------------------
# parameter containers
# ----------------------------------------------------------------------------------------------
class SincConv_fast(nn.Module):
    """Learnable band-pass bank, parameters and constant buffers exactly as
    pase/models/modules.py:818-876 (mel-spaced init 30 Hz .. sr/2-100 Hz; half Hamming window on
    linspace(0, K/2-1, K//2) divided by K; n_ = 2*pi*arange(-(K-1)/2, 0)/sr)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding="VALID", pad_mode="reflect",
                 sample_rate=16000, min_low_hz=50, min_band_hz=50):
        super().__init__()
        if in_channels != 1:
            raise ValueError("SincConv only support one input channel (here, in_channels = {%i})" % in_channels)
        self.out_channels = out_channels
        self.kernel_size = kernel_size + 1 if kernel_size % 2 == 0 else kernel_size
        self.stride = stride
        self.padding = padding
        self.pad_mode = pad_mode
        self.sample_rate = sample_rate
        self.min_low_hz = min_low_hz
        self.min_band_hz = min_band_hz
        to_mel = lambda hz: 2595 * np.log10(1 + hz / 700)
        to_hz = lambda mel: 700 * (10 ** (mel / 2595) - 1)
        high_hz = sample_rate / 2 - (min_low_hz + min_band_hz)
        hz = to_hz(np.linspace(to_mel(30), to_mel(high_hz), out_channels + 1))
        self.low_hz_ = nn.Parameter(torch.Tensor(hz[:-1]).view(-1, 1))
        self.band_hz_ = nn.Parameter(torch.Tensor(np.diff(hz)).view(-1, 1))
        n_lin = torch.linspace(0, (self.kernel_size / 2) - 1, steps=int(self.kernel_size / 2))
        self.window_ = (0.54 - 0.46 * torch.cos(2 * math.pi * n_lin / self.kernel_size)).contiguous()
        n = (self.kernel_size - 1) / 2.0
        self.n_ = (2 * math.pi * torch.arange(-n, 0).view(1, -1) / sample_rate).contiguous()


class FeBlock(NeuralBlock):
    """conv (+reflect pad) -> BatchNorm1d -> PReLU(init 0) container (modules.py:1014-1077)."""

    def __init__(self, num_inputs, fmaps, kwidth, stride, dilation, pad_mode="reflect", act=None, norm_type=None,
                 sincnet=False, sr=16000, name="FeBlock"):
        super().__init__(name=name)
        if act is not None and act != "prelu":
            raise NotImplementedError("pase_amd FeBlock: only PReLU activations (PASE/PASE+ cfgs)")
        if dilation != 1:
            raise NotImplementedError("pase_amd FeBlock: dilation != 1")
        if pad_mode != "reflect":
            raise NotImplementedError("pase_amd FeBlock: pad_mode != reflect")
        self.num_inputs = num_inputs
        self.fmaps = fmaps
        self.kwidth = kwidth
        self.stride = stride
        self.dilation = dilation
        self.pad_mode = pad_mode
        self.sincnet = sincnet
        if sincnet:
            assert num_inputs == 1, num_inputs
            self.conv = SincConv_fast(1, fmaps, kwidth, sample_rate=sr, padding="SAME", stride=stride,
                                      pad_mode=pad_mode)
        else:
            self.conv = nn.Conv1d(num_inputs, fmaps, kwidth, stride, dilation=dilation)
        # build_norm_layer (pase/models/modules.py:77-96); the spectral / weight-norm reparametrisations are not built
        if norm_type == "bnorm":
            self.norm = nn.BatchNorm1d(fmaps)
        elif norm_type == "lnorm":
            self.norm = nn.LayerNorm(fmaps)
        elif norm_type == "inorm":
            self.norm = nn.InstanceNorm1d(fmaps, affine=False)
        elif norm_type == "affinorm":
            self.norm = nn.InstanceNorm1d(fmaps, affine=True)
        elif norm_type is None:
            self.norm = None
        else:
            raise NotImplementedError("pase_amd FeBlock: norm_type %r (snorm / bsnorm / wnorm reparametrise the conv "
                                      "weight; outside the PASE(+) cfgs)" % norm_type)
        self.act = nn.PReLU(fmaps, init=0)


class QRNNLayer(nn.Module):
    """Parameter container of torchqrnn.QRNNLayer(window=2, output_gate=True): one
    nn.Linear(window*in, 3*hidden) (state_dict key `linear.{weight,bias}`)."""

    def __init__(self, input_size, hidden_size, window=2):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.window = window
        self.linear = nn.Linear(window * input_size, 3 * hidden_size)


class QRNN(nn.Module):
    """Parameter container of torchqrnn.QRNN (state_dict prefix `layers.{i}.linear.*`)."""

    def __init__(self, input_size, hidden_size, num_layers=1, dropout=0, window=2, use_cuda=True):
        super().__init__()
        if dropout:
            raise NotImplementedError("pase_amd QRNN: dropout between layers")
        if window != 2:
            raise NotImplementedError("pase_amd QRNN: window != 2")
        self.layers = nn.ModuleList(
            [QRNNLayer(input_size if l == 0 else hidden_size, hidden_size, window) for l in range(num_layers)])


def build_rnn_block(in_size, rnn_size, rnn_layers, rnn_type, bidirectional=True, dropout=0, use_cuda=True):
    """modules.py:45-60: qrnn ignores `bidirectional` and doubles rnn_size instead."""
    if rnn_type.lower() == "qrnn":
        if bidirectional:
            print("WARNING: QRNN ignores bidirectional flag")
            rnn_size = 2 * rnn_size
        return QRNN(in_size, rnn_size, rnn_layers, dropout=dropout, window=2, use_cuda=use_cuda)
    raise TypeError("Unrecognized rnn type: ", rnn_type)


class MLPBlock(NeuralBlock):
    """Conv1d(ninp, fmaps, context) -> PReLU(fmaps) container (modules.py:527-556); din/dout = 0."""

    def __init__(self, ninp, fmaps, din=0, dout=0, context=1, tie_context_weights=False, name="MLPBlock", **_):
        super().__init__(name=name)
        assert context % 2 != 0, context
        if tie_context_weights or din > 0 or dout > 0:
            raise NotImplementedError("pase_amd MLPBlock: dropout / tied context weights")
        self.ninp = ninp
        self.fmaps = fmaps
        self.context = context
        self.W = nn.Conv1d(ninp, fmaps, context, padding=context // 2)
        self.act = nn.PReLU(fmaps)


class GDeconv1DBlock(NeuralBlock):
    """ConvTranspose1d(ninp, fmaps, k, stride, padding=max(0,(stride-k)//-2)) -> PReLU(init 0)
    container (modules.py:558-589)."""

    def __init__(self, ninp, fmaps, kwidth, stride=4, norm_type=None, act=None, bias=True, name="GDeconv1DBlock"):
        super().__init__(name=name)
        if norm_type is not None or act is not None:
            raise NotImplementedError("pase_amd GDeconv1DBlock: norm / non-PReLU activation")
        pad = max(0, (stride - kwidth) // -2)
        self.deconv = nn.ConvTranspose1d(ninp, fmaps, kwidth, stride=stride, padding=pad, bias=bias)
        self.norm = None
        self.act = nn.PReLU(fmaps, init=0)
        self.kwidth = kwidth
        self.stride = stride
        # the reference trims one sample when exactly one of (stride, kwidth) is odd (:584-586)
        self.trim = (stride % 2 != 0 and kwidth % 2 == 0) or (stride % 2 == 0 and kwidth % 2 != 0)
        if self.trim:
            raise NotImplementedError("pase_amd GDeconv1DBlock: odd/even stride-kwidth trimming")

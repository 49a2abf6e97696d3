"""Import shim that lets the *live* reference (santi-pdp/pase under /root/reference) be
imported in this container, so it can pin the oracle restatement (oracle/pase_oracle.py) and
generate the golden vectors under tests/golden/.

TEST INFRASTRUCTURE ONLY.  Nothing under pase_amd/ may import this file.  /root/reference does
not exist on the GPU box, so only `oracle/make_golden.py` (run here, output committed) and the
`-m "not gpu"` pinning tests (skipped when the reference is absent) use it.

What is stubbed and why:
  * torchvision       - imported at pase/models/frontend.py:6, never used by WaveFe.
  * soundfile         - imported at pase/models/pase.py:15 (via pase.utils / transforms chain).
  * torchqrnn.QRNN    - third-party, un-vendored, un-pinned (requirements.txt:16
                        git+https://github.com/salesforce/pytorch-qrnn).  Restated below from the
                        published upstream semantics (QRNNLayer window=2, ForgetMult with
                        h_0 = f_0*z_0); call sites pase/models/modules.py:12,52-53.
"""
import sys
import types
import importlib.machinery

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


class _QRNNLayer(nn.Module):
    """salesforce/pytorch-qrnn QRNNLayer(input, hidden, window=2, output_gate=True), CPU path."""

    def __init__(self, input_size, hidden_size, window=2):
        super().__init__()
        assert window in (1, 2)
        self.window = window
        self.hidden_size = hidden_size
        self.linear = nn.Linear(window * input_size, 3 * hidden_size)

    def forward(self, X, hidden=None):
        # X: (T, B, C)
        if self.window == 2:
            Xm1 = torch.cat([X[:1] * 0, X[:-1]], 0)
            source = torch.cat([X, Xm1], 2)
        else:
            source = X
        Y = self.linear(source)
        Z, F, O = Y.chunk(3, dim=2)
        Z = torch.tanh(Z)
        F = torch.sigmoid(F)
        # ForgetMult: h_t = f_t * z_t + (1 - f_t) * h_{t-1}, h_{-1} absent (=> h_0 = f_0 z_0)
        fz = F * Z
        hs = []
        prev = hidden
        for t in range(X.size(0)):
            h = fz[t]
            if prev is not None:
                h = h + (1 - F[t]) * prev
            hs.append(h)
            prev = h
        C = torch.stack(hs)
        H = torch.sigmoid(O) * C
        return H, C[-1:]


class QRNN(nn.Module):
    """salesforce/pytorch-qrnn QRNN(input_size, hidden_size, num_layers, dropout, window, use_cuda)."""

    def __init__(self, input_size, hidden_size, num_layers=1, dropout=0, window=2, use_cuda=True,
                 **kw):
        super().__init__()
        self.layers = nn.ModuleList([
            _QRNNLayer(input_size if l == 0 else hidden_size, hidden_size, window=window)
            for l in range(num_layers)])
        self.dropout = dropout

    def forward(self, x, hidden=None):
        nh = []
        for i, layer in enumerate(self.layers):
            x, hn = layer(x, None if hidden is None else hidden[i])
            nh.append(hn)
            if self.dropout and i < len(self.layers) - 1:
                x = torch.nn.functional.dropout(x, self.dropout, self.training)
        return x, torch.cat(nh, 0)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _legacy_stft(orig):
    """torch.stft as torch<=1.7 defaulted: called WITHOUT return_complex it returned a real (..., 2) tensor.
    pase/transforms.py:467-469 (LPS) relies on that; torch 2.x rejects the call.  Same arithmetic
    (centred, reflect-padded, rectangular window of win_length zero-padded to n_fft), legacy return layout."""
    def stft(input, n_fft, hop_length=None, win_length=None, window=None, center=True, pad_mode="reflect",
             normalized=False, onesided=None, return_complex=None):
        if return_complex is not None:
            return orig(input, n_fft, hop_length, win_length, window, center, pad_mode, normalized, onesided,
                        return_complex)
        if window is None:
            window = torch.ones(win_length if win_length is not None else n_fft, dtype=input.dtype, device=input.device)
        return torch.view_as_real(orig(input, n_fft, hop_length, win_length, window, center, pad_mode, normalized,
                                       onesided, True))
    stft._pase_legacy = True
    return stft


def _librosa_delta(data, width=9, order=1, axis=-1, mode="interp", **kw):
    """librosa 0.6.3 feature.delta (the version requirements.txt:3 pins): a Savitzky-Golay filter,
    `scipy.signal.savgol_filter(data, width, deriv=order, axis=axis, mode=mode, polyorder=order)`.
    librosa itself is absent here; scipy (the function it forwards to) is live."""
    import scipy.signal
    return scipy.signal.savgol_filter(data, width, deriv=order, axis=axis, mode=mode, polyorder=order, **kw)


def _sf_read(filename, *a, **k):
    """soundfile.read stand-in for the parity tests' float32 .wav fixtures: (float64 samples, rate)."""
    import numpy as np
    import scipy.io.wavfile
    rate, x = scipy.io.wavfile.read(filename)
    assert x.dtype == np.float32, "the fixtures are IEEE-float wavs (no PCM scaling to restate)"
    return x.astype(np.float64), rate


class _Compose(object):
    """torchvision.transforms.Compose: apply in order."""

    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def install_transforms():
    """On top of install(): make `import pase.transforms` / `import pase.dataset` work, so that the chunkers,
    distortions, ZNorm, LPS and DictCollater run LIVE (pase/transforms.py:4-29, pase/dataset.py:7 imports).
    Third-party packages that are absent are stubbed; the ones whose arithmetic the pinned classes need get
    working stand-ins built on the live scipy / torch functions they forward to:
      torchvision.transforms.Compose (plain composition), soundfile.read (float wav fixtures through
      scipy.io.wavfile), librosa.feature.delta (scipy savgol, librosa 0.6.3 definition), legacy torch.stft.
    gammatone / pysptk / python_speech_features / ahoproc_tools / torchaudio stay inert: the classes that call
    them (Gammatone, Prosody, FBanks, MFCC) remain 'parity unpinned' (SURVEY 8c)."""
    install()
    g = _stub("gammatone")
    g.gtgram = _stub("gammatone.gtgram", gtgram=None)
    _stub("pysptk", swipe=None)
    _stub("python_speech_features", logfbank=None)
    lb = _stub("librosa")
    lb.feature = _stub("librosa.feature", delta=_librosa_delta)
    tv = sys.modules["torchvision"]
    tv.transforms = _stub("torchvision.transforms", Compose=_Compose)
    ah = _stub("ahoproc_tools")
    ah.interpolate = _stub("ahoproc_tools.interpolate", interpolation=None)
    ah.io = _stub("ahoproc_tools.io")
    _stub("torchaudio")
    sys.modules["soundfile"].read = _sf_read
    if not getattr(torch.stft, "_pase_legacy", False):
        torch.stft = _legacy_stft(torch.stft)


def install():
    """Make `import pase.models.frontend` etc. work from /root/reference."""
    import os
    if not os.path.isdir(REFERENCE_ROOT):
        raise FileNotFoundError(REFERENCE_ROOT)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.models = _stub("torchvision.models")
    for name in ("soundfile", "tensorboardX"):
        if name not in sys.modules:
            _stub(name, SummaryWriter=object)
    if "torchqrnn" not in sys.modules:
        _stub("torchqrnn", QRNN=QRNN)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

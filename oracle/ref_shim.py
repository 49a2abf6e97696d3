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

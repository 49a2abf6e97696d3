"""WaveFe on the HIP kernels vs the CPU oracle: forward, every parameter gradient, BatchNorm running
statistics, eval mode, dict input / output modes.  dev='emu' runs the same kernel sources on the CPU
SIMT emulator (mini widths); dev='gpu' additionally checks the full-width PASE / PASE+ configs against
the golden vectors generated from the live reference."""
import os

import numpy as np
import pytest
import torch

from oracle import pase_oracle as O
from util import (GOLD, MINI_FE, MINI_FE_PLAIN, assert_close, is_noise_grad, load_cfg, oracle_params, quiet,
                  randomize_affine, seed_all)


def _build(cfg, dev, seed=0):
    from pase_amd.frontend import wf_builder
    seed_all(seed)
    fe = quiet(wf_builder, dict(cfg))
    randomize_affine(fe)
    return fe.to(dev)


@pytest.mark.parametrize("cfg", [MINI_FE, MINI_FE_PLAIN, dict(MINI_FE, rnn_layers=2)], ids=["plus", "plain", "2xqrnn"])
def test_mini_train_forward_backward(dev, cfg):
    fe = _build(cfg, dev)
    P = oracle_params(fe)
    x = torch.randn(6, 1, 1600) * 0.3
    fe.train()
    y = fe(x.to(dev))
    so = {}
    yo = O.encoder_forward(P, cfg, x, True, so)
    assert y.shape == yo.shape
    assert_close(y, yo, rtol=1e-4, atol=1e-4, what="forward")
    g = torch.randn_like(yo)
    (yo * g).sum().backward()
    (y * g.to(dev)).sum().backward()
    for n, p in fe.named_parameters():
        if is_noise_grad(n):
            continue
        ref = P[n].grad
        assert_close(p.grad, ref, rtol=1e-3, atol=1e-4 * max(1.0, float(ref.abs().max())), what=n)
    sd = fe.state_dict()
    for k, v in so.items():
        assert_close(sd[k], v, rtol=1e-5, atol=1e-6, what=k)
    assert int(sd["blocks.1.norm.num_batches_tracked"]) == 1


def test_dense_skips_with_a_small_downstream_stride_product(dev):
    """the SincNet layer's deferred dy (weight gradient evaluating the BatchNorm + PReLU backward on load) exists
    only for pooled dense-skip branches with pool_d >= 16.  A frontend whose downstream stride product is below 16 -- here
    strides [1, 2, 2, 2]: block 0 pools by 8 -- on the one-channel split-bf16 plan (>= 32 taps) must fall back to the
    materialised dy instead of failing with -13, and still match the oracle."""
    from pase_amd import kernels as K
    cfg = dict(kwidths=[33, 11, 11, 11], strides=[1, 2, 2, 2], fmaps=[4, 4, 6, 6], emb_dim=12, rnn_dim=10, denseskips=True,
               norm_out=True, rnn_pool=True, rnn_layers=1)
    fe = _build(cfg, dev)
    P = oracle_params(fe)
    x = torch.randn(3, 1, 320) * 0.3
    fe.train()
    seen = []
    real = K.wgrad_gemm

    def spy(*a, **kw):
        r = real(*a, **kw)
        if kw.get("g_bwd") is not None:
            seen.append((r, K.LAST_WGRAD_KIND, kw["g_bwd"].get("pool_d")))
        return r
    K.wgrad_gemm = spy
    try:
        y = fe(x.to(dev))
        yo = O.encoder_forward(P, cfg, x, True, {})
        assert_close(y, yo, rtol=1e-4, atol=1e-4, what="forward")
        g = torch.randn_like(yo)
        (yo * g).sum().backward()
        (y * g.to(dev)).sum().backward()
    finally:
        K.wgrad_gemm = real
    # the on-load form was asked for on the one-channel plan and refused (pool_d 8), nothing was enqueued by that call
    assert seen == [(False, 5, 8)], seen
    for n, p in fe.named_parameters():
        if is_noise_grad(n):
            continue
        ref = P[n].grad
        assert_close(p.grad, ref, rtol=1e-3, atol=1e-4 * max(1.0, float(ref.abs().max())), what=n)


def test_mini_eval_modes_and_dict_input(dev):
    cfg = MINI_FE
    fe = _build(cfg, dev, seed=3)
    with torch.no_grad():      # non-trivial running stats
        for n, b in fe.named_buffers():
            if n.endswith("running_mean"):
                b.normal_(0, 0.1)
            if n.endswith("running_var"):
                b.uniform_(0.5, 1.5)
    P = oracle_params(fe)
    fe.eval()
    x = torch.randn(2, 1, 1760) * 0.3       # not a multiple of the 160 hop after the strides
    with torch.no_grad():
        yo = O.encoder_forward(P, cfg, x, False)
        for mode in (None, "avg_norm", "avg_concat", "avg_norm_concat"):
            y = fe(x.to(dev), mode=mode)
            assert_close(y, O.select_output(yo, mode), rtol=1e-4, atol=1e-4, what=str(mode))
        batch = {k: torch.randn(2, 1, 1600) * 0.3 for k in ("chunk", "chunk_ctxt", "chunk_rand")}
        h, chunk = fe({k: v.to(dev) for k, v in batch.items()}, device=dev)
        ho, co = O.encoder_forward_batch(P, cfg, batch, False)
        assert len(h) == 3
        for a, b in zip(h, ho):
            assert_close(a, b, rtol=1e-4, atol=1e-4)
        assert_close(chunk, co, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag,cfgfile", [("pase_plus", "frontend/PASE+.cfg"), ("pase", "frontend/PASE.cfg")])
def test_full_width_golden(dev, tag, cfgfile):
    """Full-width PASE / PASE+ encoder vs vectors produced by the live reference (fp32 |err| <= 1e-4
    on the normalised embedding: BASELINE.json north_star tolerance)."""
    if dev.type == "cpu":
        pytest.skip("full-width configs are GPU-only (emulator covers the mini configs)")
    from pase_amd.frontend import wf_builder
    g = np.load(os.path.join(GOLD, "wavefe_%s.npz" % tag))
    cfg = load_cfg(cfgfile)
    seed_all(int(g["seed"]))
    fe = quiet(wf_builder, dict(cfg)).to(dev)
    x = torch.tensor(g["x"]).to(dev)
    fe.train()
    y = fe(x)
    assert tuple(y.shape) == g["y_train"].shape
    assert_close(y, g["y_train"], rtol=0, atol=1e-4, what="train forward")
    (y * torch.tensor(g["g"]).to(dev)).sum().backward()
    names = [str(s) for s in g["grad_names"]]
    params = dict(fe.named_parameters())
    keep = [i for i, n in enumerate(names) if not is_noise_grad(n)]
    gsq = torch.tensor([float((params[names[i]].grad.double() ** 2).sum()) for i in keep])
    assert_close(gsq.sqrt(), np.sqrt(g["grad_sq"][keep]), rtol=2e-3, atol=1e-5, what="grad norms")
    fe.eval()
    with torch.no_grad():
        assert_close(fe(x), g["y_eval"], rtol=0, atol=1e-4, what="eval forward")
        assert_close(fe(x, mode="avg_norm"), g["y_avg_norm"], rtol=0, atol=1e-4)


def test_readme_shapes(dev):
    """BASELINE.json configs[0] / README.md:31-39: (1,1,100000) -> (1,256,625) for PASE+.cfg and
    (1,100,625) for PASE.cfg (SURVEY.md headline fact 6)."""
    if dev.type == "cpu":
        pytest.skip("full-width configs are GPU-only")
    from pase_amd.frontend import wf_builder
    for cfgfile, emb in (("frontend/PASE+.cfg", 256), ("frontend/PASE.cfg", 100)):
        fe = quiet(wf_builder, load_cfg(cfgfile)).to(dev).eval()
        with torch.no_grad():
            y = fe(torch.randn(1, 1, 100000, device=dev))
        assert tuple(y.shape) == (1, emb, 625)
        assert bool(torch.isfinite(y).all())


@pytest.mark.parametrize("cfg", [MINI_FE, MINI_FE_PLAIN], ids=["plus", "plain"])
def test_eval_mode_backward_frozen_bn_and_input_gradient(dev, cfg):
    """fe.eval(); loss(fe(x)).backward(): BatchNorm with FROZEN statistics backpropagates dy = scale*dz (fine-tuning
    through a frozen frontend).  In this mode the conv biases in front of a BatchNorm and W.bias have NON-ZERO
    gradients, so the bias-gradient kernel path (excluded from the train-mode comparisons by is_noise_grad) is
    checked here -- every parameter, no exclusions -- together with the gradient w.r.t. the input waveform."""
    fe = _build(cfg, dev, seed=5)
    with torch.no_grad():
        for n, b in fe.named_buffers():
            if n.endswith("running_mean"):
                b.normal_(0, 0.1)
            if n.endswith("running_var"):
                b.uniform_(0.5, 1.5)
    P = oracle_params(fe)
    fe.eval()
    x = (torch.randn(4, 1, 1600) * 0.3).requires_grad_(True)
    xd = x.detach().to(dev).requires_grad_(True)
    y = fe(xd)
    yo = O.encoder_forward(P, cfg, x, False)
    assert_close(y, yo, rtol=1e-4, atol=1e-4, what="eval forward")
    g = torch.randn_like(yo)
    (yo * g).sum().backward()
    (y * g.to(dev)).sum().backward()
    for n, p in fe.named_parameters():
        ref = P[n].grad
        assert float(ref.abs().max()) > 0, n
        assert_close(p.grad, ref, rtol=1e-3, atol=1e-4 * max(1.0, float(ref.abs().max())), what=n)
    assert_close(xd.grad, x.grad, rtol=1e-3, atol=1e-4 * max(1.0, float(x.grad.abs().max())), what="d/d waveform")
    sd = fe.state_dict()
    for k in sd:
        if "running" in k:
            assert_close(sd[k], P[k], rtol=0, atol=0, what=k)     # eval forward leaves the statistics alone


def test_train_mode_input_gradient(dev):
    fe = _build(MINI_FE, dev, seed=6)
    P = oracle_params(fe)
    fe.train()
    x = (torch.randn(3, 1, 1600) * 0.3).requires_grad_(True)
    xd = x.detach().to(dev).requires_grad_(True)
    g = torch.randn(3, 12, 10)
    (O.encoder_forward(P, MINI_FE, x, True, {}) * g).sum().backward()
    (fe(xd) * g.to(dev)).sum().backward()
    assert_close(xd.grad, x.grad, rtol=1e-3, atol=1e-4 * max(1.0, float(x.grad.abs().max())), what="d/d waveform")


@pytest.mark.parametrize("norm_type", ["lnorm", "inorm", "affinorm"])
def test_per_sample_norm_types_forward_backward(dev, norm_type):
    """norm_type 'lnorm' / 'inorm' / 'affinorm' (modules.py:77-109; norm_out becomes an InstanceNorm1d,
    frontend.py:206-210): the 2xQRNN dense-skip encoder of BASELINE.json configs[4] at mini width, forward and
    every parameter gradient vs the oracle (itself pinned to the live reference for these norm types)."""
    cfg = dict(MINI_FE, rnn_layers=2, norm_type=norm_type)
    fe = _build(cfg, dev, seed=8)
    with torch.no_grad():
        for n, p in fe.named_parameters():
            if n.endswith("norm.weight"):
                p.uniform_(0.5, 1.5)
            elif n.endswith("norm.bias"):
                p.normal_(0, 0.2)
    P = oracle_params(fe)
    x = torch.randn(4, 1, 3200) * 0.3
    fe.train()
    y = fe(x.to(dev))
    yo = O.encoder_forward(P, cfg, x, True, {})
    assert_close(y, yo, rtol=1e-4, atol=1e-4, what="forward")
    g = torch.randn_like(yo)
    (yo * g).sum().backward()
    (y * g.to(dev)).sum().backward()
    for n, p in fe.named_parameters():
        ref = P[n].grad
        # instance norms cancel the conv bias in front of them exactly like BatchNorm (round-off-only gradients);
        # LayerNorm (statistics over channels) does not
        if norm_type != "lnorm" and is_noise_grad(n):
            continue
        assert_close(p.grad, ref, rtol=1e-3, atol=1e-4 * max(1.0, float(ref.abs().max())), what=n)
    fe.eval()                      # per-sample statistics: eval == train arithmetic
    with torch.no_grad():
        assert_close(fe(x.to(dev)), O.encoder_forward(P, cfg, x, False), rtol=1e-4, atol=1e-4, what="eval forward")

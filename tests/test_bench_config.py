"""Parity gate AT THE BENCHMARK CONFIGURATION (BASELINE.json configs[2]): PASE+.cfg + workers+.cfg, B = 32
utterances x 32 000 samples (96 sequences through the encoder), on the HIP path vs the oracle restatement
(oracle/pase_oracle.py) evaluated with stock torch fp32 ops on the same GPU, from the same state_dict and
the same batch.  This is the only place the 3-workgroup/CU instantiations, the split-K heuristics, the
32-bit offset guards and BatchNorm statistics over 96 x 32 000 samples are checked against the reference
algorithm (worker_scheduler.py:43-75, trainer.py:229-232):

  * embedding |err| <= 1e-4, all 13 losses 1e-4 relative,
  * ELEMENT-WISE gradients of every parameter that has a non-noise gradient,
  * 10 Adam steps on fresh batches: total-loss curves track.
"""
import contextlib
import io
import json
import os

import pytest
import torch

from oracle import pase_oracle as O
from util import ROOT, assert_close, is_noise_grad

pytestmark = pytest.mark.gpu

B, T = 32, 32000


def _cfgs():
    from pase_amd.utils import strip_transforms, worker_parser
    with open(os.path.join(ROOT, "cfg", "frontend", "PASE+.cfg")) as f:
        fe = json.load(f)
    with contextlib.redirect_stdout(io.StringIO()):
        wk = strip_transforms(worker_parser(os.path.join(ROOT, "cfg", "workers", "workers+.cfg")))
    with open(os.path.join(ROOT, "cfg", "workers", "workers+.cfg")) as f:
        raw = json.load(f)
    return fe, wk, raw


def _batch(seed, raw, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    batch = {k: (0.1 * torch.randn(B, 1, T, generator=g, device=dev)).clamp_(-1, 1)
             for k in ("chunk", "chunk_ctxt", "chunk_rand", "cchunk")}
    for w in raw["regr"]:
        if w["name"] not in batch:
            batch[w["name"]] = torch.randn(B, w["num_outputs"], T // 160, generator=g, device=dev)
    return batch


@pytest.fixture(scope="module", params=[True, False], ids=["x6", "fp32pipe"])
def setup(request):
    """Both matrix pipes are gated: the split-bf16 contraction (the shipped default) and the exact-fp32 MFMA pipe
    (K.X6 False) run the same two tests against the same comparator."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pase_amd import _lib
    _lib.use_library(None, "cuda")
    _lib.lib()
    from pase_amd import kernels as K
    saved = K.X6
    K.X6 = request.param
    try:
        yield _make_setup()
    finally:
        K.X6 = saved


def _make_setup():
    from pase_amd.trainer import trainer
    dev = torch.device("cuda:0")
    fe, wk, raw = _cfgs()
    torch.manual_seed(2)
    with contextlib.redirect_stdout(io.StringIO()):
        tr = trainer(frontend_cfg=dict(fe), minions_cfg=wk, cfg=dict(fe_lr=1e-3, min_lr=5e-4, epoch=1, bpe=10 ** 6),
                     device=dev)
    # move the PReLU slopes / BN affines off their init values (0, 1, 0) so every backward term is live
    g = torch.Generator().manual_seed(123)
    with torch.no_grad():
        for n, p in tr.model.named_parameters():
            if n.endswith("norm.weight"):
                p.copy_(torch.empty(p.shape).uniform_(0.7, 1.3, generator=g))
            elif n.endswith("norm.bias"):
                p.copy_(torch.empty(p.shape).normal_(0, 0.1, generator=g))
            elif n.endswith("act.weight") and n.startswith("frontend."):
                p.copy_(torch.empty(p.shape).uniform_(0.02, 0.3, generator=g))
    P = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
    names = [n for n, _ in tr.model.named_parameters()]
    for n in names:
        P[n].requires_grad_(True)
    return dict(tr=tr, P=P, names=names, fe=fe, raw=raw, dev=dev)


def _oracle_step(P, fe, raw, batch, opts=None):
    if opts is not None:
        for o in opts:
            o.zero_grad()
    so = {}
    h, chunk, preds, labels = O.pase_forward(P, fe, raw, batch, True, so)
    lo = O.pase_losses(raw, preds, labels)
    lo["total"].backward()
    emb = torch.cat([t.detach() for t in h], 0)
    del h, chunk, preds, labels
    if opts is not None:
        for o in opts:
            o.step()
    with torch.no_grad():
        for k, v in so.items():
            P["frontend." + k].copy_(v)
    return {k: float(v) for k, v in lo.items()}, emb


def test_bs32_embedding_losses_and_elementwise_grads(setup):
    from pase_amd import engine
    tr, P, fe, raw, dev = setup["tr"], setup["P"], setup["fe"], setup["raw"], setup["dev"]
    batch = _batch(4321, raw, dev)
    m = tr.model
    m.train()
    # -- HIP: embedding (train-mode encoder), then the fused loss + backward (no optimizer step) ------------
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.cat([batch[k] for k in ("chunk", "chunk_ctxt", "chunk_rand")], 0)
    emb, _ = engine.encoder_forward(m.frontend, x, training=True, need_ctx=False)
    emb = emb.clone()
    with torch.no_grad():
        for k, v in m.state_dict().items():
            v.copy_(sd0[k])
    for opt in tr.optimizers():
        opt.zero_grad()
    lf = m.loss_and_grads(batch)
    lf = {k: float(v) for k, v in lf.items()}
    # -- oracle on the same GPU (stock torch ops, fp32, autograd) ---------------------------------------------
    P0 = {k: v.detach().clone() for k, v in P.items()}
    lo, emb_ref = _oracle_step(P, fe, raw, batch)
    with torch.no_grad():                    # keep P at the pre-step state for the curve test
        for k in P:
            if not P[k].requires_grad:
                P[k].copy_(P0[k])
    assert_close(emb, emb_ref, rtol=0, atol=1e-4, what="embedding (96,256,200)")
    assert len(lo) == 13
    for k, v in lo.items():
        assert abs(lf[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, lf[k], v)
    # -- element-wise gradients -------------------------------------------------------------------------------
    checked = 0
    worst = (0.0, None)
    bad = []
    stats = []
    for n, p in m.named_parameters():
        if is_noise_grad(n):
            continue
        ref = P[n].grad
        gmax = float(ref.abs().max())
        err = float((p.grad - ref).abs().max())
        rel2 = float((p.grad - ref).double().norm() / max(1e-30, float(ref.double().norm())))
        worst = max(worst, (rel2, n))
        # Tolerances (relative L2 per tensor; every element within 2 % of the tensor's largest gradient -- measured
        # worst 0.9 % on blocks.4.conv.weight):
        #   * weight tensors: 3e-3.  Both sides sum 19 200 ... 3 072 000 products per element in fp32 in different
        #     orders (ours: MFMA k-order + split-K atomics; comparator: MIOpen / rocBLAS);
        #   * per-channel reductions (BN gamma / beta, PReLU slopes, biases): 1e-2 -- ONE scalar per channel summed
        #     over up to 3M positions; ours accumulates in fp64, the comparator (torch's fp32 batch_norm / prelu
        #     backward) does not, measured 2e-6 absolute on gradients of 1e-3;
        #   * the two SincNet vectors: 1e-2 -- d/d(low_hz, band_hz) contracts the 251-tap filter gradient with sin/cos
        #     derivatives of alternating sign (cancellation).
        # A sign, permutation or missing-term error shows up at O(1), three orders above these.
        per_channel = n.endswith(("norm.weight", "norm.bias", "act.weight", ".bias", "low_hz_", "band_hz_"))
        tol2 = 1e-2 if per_channel else 3e-3
        if not (err <= 2e-2 * gmax + 1e-9 and rel2 <= tol2):
            bad.append((n, "max|err| %.3e of max|g| %.3e" % (err, gmax), "relL2 %.3e" % rel2))
        stats.append((rel2, n))
        checked += 1
    print("worst relative L2 gradient error:", worst)
    for r, n in sorted(stats, reverse=True)[:12]:
        print("   relL2 %.3e  %s" % (r, n))
    for b in bad:
        print("   OUT OF TOLERANCE", b)
    assert not bad, "%d tensors out of tolerance, first: %r" % (len(bad), bad[:3])
    assert checked >= 100, checked
    for n in setup["names"]:
        P[n].grad = None


def test_bs32_ten_adam_steps_track(setup):
    """Loss curves track (north_star): 10 steps, fresh batch each, Adam fe 1e-3 / workers 5e-4 on both sides."""
    tr, P, fe, raw, dev, names = (setup[k] for k in ("tr", "P", "fe", "raw", "dev", "names"))
    opts = [torch.optim.Adam([P[n]], lr=1e-3 if n.startswith("frontend.") else 5e-4) for n in names]
    p0 = {n: p.detach().clone() for n, p in tr.model.named_parameters()}
    ours, ref = [], []
    for s in range(10):
        batch = _batch(500 + s, raw, dev)
        ours.append(float(tr.train_step(batch)["total"]))
        lo, _ = _oracle_step(P, fe, raw, batch, opts)
        ref.append(lo["total"])
        del batch
    rel = [abs(a - b) / abs(b) for a, b in zip(ours, ref)]
    print("hip  ", ours)
    print("torch", ref)
    assert rel[0] <= 1e-5, rel
    assert max(rel) <= 2e-3, rel
    assert ours[-1] < ours[0]              # and it trains
    # parameters after 10 Adam steps: Adam normalises every gradient to a +-lr step, so an element whose gradient is
    # round-off-sized (dense-skip and decoder weights early in training) moves by lr per step in a direction both
    # implementations pick by round-off -- a per-element bound says nothing there.  What must agree is the UPDATE as
    # a whole: its direction (cosine >= 0.95 per tensor; measured >= 0.98) and its length (within 5 %).
    worst = (1.0, None)
    for n, p in tr.model.named_parameters():
        if is_noise_grad(n):
            continue
        da, db = (p.detach() - p0[n]).double().flatten(), (P[n].detach() - p0[n]).double().flatten()
        cos = float((da * db).sum() / (da.norm() * db.norm()).clamp_min(1e-30))
        worst = min(worst, (cos, n))
        assert cos >= 0.95, (n, cos)
        ratio = float(da.norm() / db.norm().clamp_min(1e-30))
        assert 0.95 <= ratio <= 1.05, (n, ratio)
    print("smallest cosine between the two 10-step updates:", worst)

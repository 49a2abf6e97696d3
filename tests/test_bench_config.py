"""Parity gates AT FULL SIZE for the BASELINE.json configurations, on the HIP path vs the oracle restatement
(oracle/pase_oracle.py) evaluated with stock torch fp32 ops on the same GPU, from the same state_dict and
the same batch:

  pase+    configs[2] (the benchmark): PASE+.cfg + workers+.cfg, 32 utterances x 32 000 samples -- both matrix pipes
  pase     configs[1]: PASE.cfg + workers.cfg (emb 100, no QRNN / skips; decoder, MLP regressors, SPC / MI / CMI),
           32 x 16 000 (/root/reference/cfg/workers/workers.cfg; the SPC worker's frames come from Python's `random`
           stream, seeded identically on both sides)
  emb256   configs[4]: the dense PASE+ encoder with norm_type 'lnorm' and two QRNN layers (modules.py:77-109,
           template_scripts/run_pase_train_50h_2xQRNN_addrev_lnorm_EMB256.sh), 64 x 32 000: LayerNorm / InstanceNorm
           kernels and 2x the activation size of the benchmark (the 32-bit offset guards)

This is the only place the full-width instantiations, the split-K heuristics, the routing between the matrix pipes,
the offset guards and the batch statistics over 96 ... 192 long sequences are checked against the reference
algorithm (worker_scheduler.py:43-75, trainer.py:229-232):

  * embedding |err| <= 1e-4, all 13 losses 1e-4 relative,
  * gradients of every parameter that has a non-noise gradient, against an fp64 evaluation of the same step,
  * 10 Adam steps on fresh batches: total-loss curves track.
"""
import contextlib
import io
import json
import os
import random

import pytest
import torch

from oracle import pase_oracle as O
from util import ROOT, assert_close, is_noise_grad

pytestmark = pytest.mark.gpu

VARIANTS = {
    "pase+": dict(fe="PASE+.cfg", fe_over={}, workers="workers+.cfg", B=32, T=32000),
    "pase": dict(fe="PASE.cfg", fe_over={}, workers="workers.cfg", B=32, T=16000),
    "emb256": dict(fe="PASE+.cfg", fe_over=dict(rnn_layers=2, norm_type="lnorm"), workers="workers+.cfg", B=64, T=32000),
}


def _cfgs(variant="pase+"):
    from pase_amd.utils import strip_transforms, worker_parser
    v = VARIANTS[variant]
    with open(os.path.join(ROOT, "cfg", "frontend", v["fe"])) as f:
        fe = dict(json.load(f), **v["fe_over"])
    with contextlib.redirect_stdout(io.StringIO()):
        wk = strip_transforms(worker_parser(os.path.join(ROOT, "cfg", "workers", v["workers"])))
    with open(os.path.join(ROOT, "cfg", "workers", v["workers"])) as f:
        raw = json.load(f)
    return fe, wk, raw


def _batch(seed, raw, dev, B=32, T=32000):
    g = torch.Generator(device=dev).manual_seed(seed)
    batch = {k: (0.1 * torch.randn(B, 1, T, generator=g, device=dev)).clamp_(-1, 1)
             for k in ("chunk", "chunk_ctxt", "chunk_rand", "cchunk")}
    for w in raw["regr"]:
        if w["name"] not in batch:
            batch[w["name"]] = torch.randn(B, w["num_outputs"], T // 160, generator=g, device=dev)
    return batch


def _noise(variant, n):
    """parameters whose gradient is analytically zero (round-off on both sides)"""
    if variant == "emb256":      # LayerNorm over channels does NOT cancel a per-channel conv bias; InstanceNorm norm_out does W's
        return n == "frontend.W.bias"
    return is_noise_grad(n)


@pytest.fixture(scope="module", params=[("pase+", True), ("pase+", False), ("pase", True), ("pase", False), ("emb256", True)],
                ids=["x6", "fp32pipe", "pase-cfg1-x6", "pase-cfg1-fp32pipe", "emb256-lnorm-2xqrnn-bs64-x6"])
def setup(request):
    """The benchmark configuration is gated on both matrix pipes -- the split-bf16 contraction (the shipped default) and
    the exact-fp32 MFMA pipe (K.X6 False) run the same two tests against the same comparator -- the other two BASELINE
    configurations on the default."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pase_amd import _lib
    _lib.use_library(None, "cuda")
    _lib.lib()
    from pase_amd import kernels as K
    variant, x6 = request.param
    saved = K.X6
    K.X6 = x6
    try:
        yield _make_setup(variant)
    finally:
        K.X6 = saved
        torch.cuda.empty_cache()


def _make_setup(variant="pase+"):
    from pase_amd.trainer import trainer
    dev = torch.device("cuda:0")
    fe, wk, raw = _cfgs(variant)
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
    return dict(tr=tr, P=P, names=names, fe=fe, raw=raw, dev=dev, variant=variant, B=VARIANTS[variant]["B"],
                T=VARIANTS[variant]["T"])


def _oracle_step(P, fe, raw, batch, opts=None, seed=None):
    if seed is not None:
        random.seed(seed)          # the SPC worker's frame draws (minions.py:614-628)
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
    variant = setup["variant"]
    batch = _batch(4321, raw, dev, setup["B"], setup["T"])
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
    random.seed(77)
    lf = m.loss_and_grads(batch)
    lf = {k: float(v) for k, v in lf.items()}
    # -- oracle on the same GPU (stock torch ops, fp32, autograd) ---------------------------------------------
    P0 = {k: v.detach().clone() for k, v in P.items()}
    lo, emb_ref = _oracle_step(P, fe, raw, batch, seed=77)
    with torch.no_grad():                    # keep P at the pre-step state for the curve test
        for k in P:
            if not P[k].requires_grad:
                P[k].copy_(P0[k])
    assert_close(emb, emb_ref, rtol=0, atol=1e-4, what="embedding %s" % (tuple(emb.shape),))
    assert len(lo) == {"pase+": 13, "pase": 8, "emb256": 13}[variant]
    for k, v in lo.items():
        assert abs(lf[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, lf[k], v)
    # -- the same step in fp64 (same torch restatement, same constants): the truth both fp32 evaluations are measured by ---
    ref32 = {n: P[n].grad.detach().double().clone() for n in setup["names"]}
    # -- how far does an EQUALLY VALID fp32 evaluation land from that one?  The comparator again, with one layer's weights
    # moved by one unit in the last place (block 1: its output changes by ~1e-7 relative, what a different summation order
    # does).  Round 4: routing block 1's forward to the split-bf16 kernel (error against fp64 2.2e-7 instead of 6.2e-7 on
    # that layer, checked element-wise) moved the gradients of blocks 1-4 from 2.5e-3 to 5.5e-3 ... 7.8e-3 away from fp64:
    # the PReLU kinks and the batch statistics over 3M positions amplify ANY last-bit change of an early layer that much,
    # so the gate measures the comparator's own spread instead of assuming it is the 1.5x of a single draw.
    ref32_alt = []
    wname = "frontend.blocks.1.conv.weight"
    for k in range(2):
        gen = torch.Generator(device="cpu").manual_seed(900 + k)
        w0 = P[wname].detach().clone()
        sign = (torch.randint(0, 2, tuple(w0.shape), generator=gen).float() * 2 - 1).to(w0.device)
        with torch.no_grad():
            P[wname].mul_(1 + sign * 2.0 ** -23)
        for n in setup["names"]:
            P[n].grad = None
        _oracle_step(P, fe, raw, batch, seed=77)
        ref32_alt.append({n: P[n].grad.detach().double().clone() for n in setup["names"]})
        with torch.no_grad():
            P[wname].copy_(w0)
            for kk in P:
                if not P[kk].requires_grad:
                    P[kk].copy_(P0[kk])
    P64 = {k: (v.detach().double() if v.is_floating_point() else v.detach().clone()) for k, v in P0.items()}
    for n in setup["names"]:
        P64[n].requires_grad_(True)
    _oracle_step(P64, fe, raw, {k: v.double() for k, v in batch.items()}, seed=77)
    # -- element-wise gradients -------------------------------------------------------------------------------
    # Per tensor, relative L2 against the fp64 evaluation:  ours <= 1.5 x (torch fp32 ops: the worst of three equally
    # valid evaluations, see above) + floor.  At these sizes
    # the fp32 comparator itself is 1e-3 ... 3e-3 away from fp64 on the encoder (19 200 ... 3 072 000 products per
    # element summed in fp32, BatchNorm statistics over 3M positions) -- measured (tools/f64_probe.py, PASE.cfg
    # bs32): torch fp32 2.6e-3 ... 3.5e-3, split-bf16 pipe 2.1e-3 ... 2.7e-3, exact-fp32 pipe 0.5e-3 ... 1.3e-3 -- so a
    # bound on |ours - torch fp32| alone measures the comparator.  Floors: 5e-4 (weights), 1.5e-3 (one scalar per channel:
    # BN gamma / beta, PReLU slopes, biases, the two SincNet vectors) for tensors where all three sit at round-off level.
    # A sign, permutation or missing-term error shows up at O(1).
    checked = 0
    worst = (0.0, None)
    bad = []
    stats = []
    skipped = []
    for n, p in m.named_parameters():
        if _noise(variant, n):
            continue
        t64 = P64[n].grad
        den = max(1e-300, float(t64.norm()))
        e_ours = float((p.grad.double() - t64).norm()) / den
        e_ref = float((ref32[n] - t64).norm()) / den
        e_ref_one = e_ref
        e_ref = max([e_ref] + [float((alt[n] - t64).norm()) / den for alt in ref32_alt])
        worst = max(worst, (e_ours, n))
        per_channel = n.endswith(("norm.weight", "norm.bias", "act.weight", ".bias", "low_hz_", "band_hz_"))
        floor = 1.5e-3 if per_channel else 5e-4
        setup.setdefault("eref", {})[n] = e_ref
        if e_ref >= 0.5:
            # the fp64 gradient of this tensor is (numerically) zero -- e.g. a worker whose hidden units are all dead at
            # this random state -- and BOTH fp32 evaluations are pure round-off relative to it: nothing to compare
            skipped.append(n)
            continue
        # (no absolute cap: the two SincNet vectors are 2.5e-2 ... 6.7e-2 away from fp64 in torch fp32 AND here -- the
        #  cancellation in d/d(low_hz, band_hz) -- and agree with each other to 1e-4 of that)
        if not e_ours <= 1.5 * e_ref + floor:
            bad.append((n, "relL2 vs fp64: ours %.3e, torch fp32 %.3e (as is %.3e)" % (e_ours, e_ref, e_ref_one)))
        stats.append((e_ours, n, e_ref))
        checked += 1
    del P64, ref32, ref32_alt
    print("worst relative L2 gradient error:", worst)
    for r, n, rr in sorted(stats, reverse=True)[:12]:
        print("   relL2 vs fp64: ours %.3e  torch fp32 %.3e  %s" % (r, rr, n))
    for b in bad:
        print("   OUT OF TOLERANCE", b)
    assert not bad, "%d tensors out of tolerance, first: %r" % (len(bad), bad[:3])
    assert checked >= {"pase+": 100, "pase": 50, "emb256": 100}[variant], checked
    assert len(skipped) <= 4, skipped
    setup["dead"] = set(skipped)          # their Adam updates are +-lr steps in round-off directions on both sides
    for n in setup["names"]:
        P[n].grad = None


def test_bs32_ten_adam_steps_track(setup):
    """Loss curves track (north_star): 10 steps, fresh batch each, Adam fe 1e-3 / workers 5e-4 on both sides."""
    tr, P, fe, raw, dev, names = (setup[k] for k in ("tr", "P", "fe", "raw", "dev", "names"))
    opts = [torch.optim.Adam([P[n]], lr=1e-3 if n.startswith("frontend.") else 5e-4) for n in names]
    p0 = {n: p.detach().clone() for n, p in tr.model.named_parameters()}
    ours, ref = [], []
    variant = setup["variant"]
    for s in range(10):
        batch = _batch(500 + s, raw, dev, setup["B"], setup["T"])
        random.seed(900 + s)
        ours.append(float(tr.train_step(batch)["total"]))
        lo, _ = _oracle_step(P, fe, raw, batch, opts, seed=900 + s)
        ref.append(lo["total"])
        del batch
    rel = [abs(a - b) / abs(b) for a, b in zip(ours, ref)]
    print("hip  ", ours)
    print("torch", ref)
    # benchmark configuration: measured <= 1.4e-3 over the 10 steps.  The other two (half the samples per BatchNorm
    # statistic / LayerNorm + twice the sequences) measured 3.1e-3 at step 5: two fp32 evaluations whose round-off Adam
    # turns into +-lr steps drift apart that fast; the first step, before any update, agrees to 1e-5 everywhere
    assert rel[0] <= 1e-5, rel
    # (round 6: gates at what was measured + a margin -- 1.4e-3 -> 2e-3, 3.1e-3 -> 4e-3)
    assert max(rel) <= (2e-3 if variant == "pase+" else 4e-3), rel
    assert ours[-1] < ours[0]              # and it trains
    # parameters after 10 Adam steps: Adam normalises every gradient to a +-lr step, so an element whose gradient is
    # round-off-sized (dense-skip and decoder weights early in training) moves by lr per step in a direction both
    # implementations pick by round-off -- a per-element bound says nothing there.  What must agree is the UPDATE as
    # a whole: its direction (cosine >= 0.95 per tensor; measured >= 0.98) and its length (within 5 %).
    # A tensor whose step-0 gradient the torch fp32 comparator itself only knows to a few per cent (relative L2 against fp64
    # >= 2e-2: the cancellation-dominated SincNet vectors, workers whose hidden units are nearly all dead at this state) is
    # noise-dominated -- Adam turns that into +-lr steps in round-off directions on both sides; measured on emb256: the cmi
    # worker's hidden weight, cosine -0.006 between two fp32 evaluations -- and single-element tensors (a head's scalar bias)
    # have no direction at all: both are reported, not gated.
    # Gates = measured worst case minus a margin (round 4, both pipes, several boxes; atomics make the runs differ):
    #   pase+ (benchmark)  cos >= 0.986, length within 3.8 %   -> 0.95, 5 %
    #   pase (configs[1])  cos >= 0.951, length within 6.5 %   -> 0.92, 8 %   (half the samples per BatchNorm statistic)
    #   emb256 (configs[4]) cos >= 0.895 (first dense-skip weight), length within 4.2 % -> 0.85, 8 %
    COS = {"pase+": 0.95, "pase": 0.92, "emb256": 0.85}[variant]
    LEN = {"pase+": 0.05, "pase": 0.08, "emb256": 0.08}[variant]
    rows, ungated = [], []
    eref = setup.get("eref", {})
    for n, p in tr.model.named_parameters():
        if _noise(variant, n) or n in setup.get("dead", ()):
            continue
        da, db = (p.detach() - p0[n]).double().flatten(), (P[n].detach() - p0[n]).double().flatten()
        cos = float((da * db).sum() / (da.norm() * db.norm()).clamp_min(1e-30))
        ratio = float(da.norm() / db.norm().clamp_min(1e-30))
        (ungated if (eref.get(n, 0.0) >= 2e-2 or p.numel() == 1) else rows).append((cos, ratio, n))
    print("smallest cosines between the two 10-step updates:", sorted(rows)[:6])
    print("update-length ratios farthest from 1:", sorted(rows, key=lambda r: -abs(r[1] - 1.0))[:6])
    print("not gated (noise-dominated step-0 gradient or a single element):", sorted(ungated)[:8])
    assert len(ungated) <= 8, ungated
    for cos, ratio, n in rows:
        assert cos >= COS, (n, cos)
        assert 1.0 - LEN <= ratio <= 1.0 + LEN, (n, ratio)


def _voiced_pool(rs, n, T):
    """harmonic complexes with gliding f0 (70-280 Hz) + noise floor, an unvoiced noise stretch and a silent stretch each: signals
    on which SWIPE' decides voiced AND unvoiced frames (white noise alone is unvoiced throughout: the prosody target would be a
    constant row and its gate empty)"""
    import numpy as np
    t = np.arange(T) / 16000.0
    out = []
    for _ in range(n):
        fa, fb = rs.uniform(70, 280, size=2)
        f0 = fa + (fb - fa) * t / t[-1]
        ph = 2 * np.pi * np.cumsum(f0) / 16000.0
        x = 0.1 * sum(np.sin(k * ph) / k for k in range(1, 12)) + 0.003 * rs.standard_normal(T)
        g0 = rs.randint(T // 4, T // 2)
        x[g0:g0 + T // 8] = 0.02 * rs.standard_normal(T // 8)
        x[:T // 40] = 0.0
        out.append(x.clip(-1, 1).astype(np.float32))
    return out


def test_bs32_producer_mode_step(tmp_path):
    """BASELINE.json configs[3], the part one GPU can check: a bs32 step whose batch comes from the on-device producer
    (pase_amd/producer.py: crops + reverb / additive-noise chain + DSP regression targets, the data side of
    /root/reference/pase/dataset.py:430-520 + train.py:37-136) instead of given tensors.  Comparator: the SAME crops
    through the numpy / scipy restatement of the target transforms (oracle/dsp_oracle.py) -- the prosody target from the
    ORACLE's own SWIPE' contour (oracle/swipe_oracle.py, fp64), not from the device tracker's -- and the torch restatement of
    the step (oracle/pase_oracle.py).  Three gates:
      (a) targets: every regression target within 2e-3 of its range; prosody on the frames whose 9-frame delta window has
          the same voicing decisions on both sides (agreement rate asserted inline), log-f0 rows within the 1.5 % f0
          tolerance of tests/test_dsp.py, the f0-independent rows (voiced flag, energy, zero crossings) as tight as the others;
      (b) losses: the HIP step on the producer's own batch against the comparator on the oracle's targets;
      (c) gradients of EVERY parameter tensor, both sides on the same (oracle) labels, judged against an fp64 evaluation of
          the step like the other full-size gates: ours <= 1.5 x torch-fp32 + floor."""
    import numpy as np
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import dsp_oracle as D
    from oracle import swipe_oracle as SW
    from pase_amd import _lib, dsp, producer as PR
    from pase_amd.trainer import trainer
    _lib.use_library(None, "cuda")
    _lib.lib()
    dev = torch.device("cuda:0")
    fe, wk, raw = _cfgs("pase+")
    Bp, Tp = 32, 32000
    rs = np.random.RandomState(77)
    pool = PR.WavPool(_voiced_pool(rs, 40, 16000 * 5), dev)
    irs = [np.r_[np.zeros(40), 1.0, 0.3 * rs.standard_normal(7959) * np.exp(-np.arange(7959) / 1500.0)] for _ in range(4)]
    noises = [0.05 * rs.standard_normal(16000 * 6) for _ in range(4)]
    tg = dsp.DeviceTargets(raw, device=dev)
    stats = {}
    for n_, f_ in tg.feats.items():
        D_ = next(w["num_outputs"] for w in raw["regr"] if w["name"] == n_)
        stats[n_] = (rs.standard_normal(D_).astype(np.float32) * 0.1, (0.5 + rs.random_sample(D_)).astype(np.float32))
        f_.set_stats(torch.from_numpy(stats[n_][0]), torch.from_numpy(stats[n_][1]))
    prod = PR.DeviceBatchProducer(PR.DeviceChunker(pool, Tp, rng=rs), PR.DeviceReverb(irs, device=dev), 0.5,
                                  PR.DeviceAdditive(noises, device=dev), 0.5, tg, rng=rs)
    batch = prod(Bp)
    assert batch["chunk"].shape == (Bp, 1, Tp) and not torch.equal(batch["chunk"], batch["cchunk"])
    # ---- (a) regression targets of the same clean crops on the CPU ---------------------------------------------------
    f0_dev = tg.feats["prosody"].tracker(batch["cchunk"]).cpu().numpy()
    fns = {"lps": D.lps, "fbank": D.fbanks, "gtn": D.gammatone, "mfcc": D.mfcc}
    clean = batch["cchunk"][:, 0].cpu().numpy()
    ref_batch = {k: batch[k] for k in ("chunk", "chunk_ctxt", "chunk_rand", "cchunk")}
    for w in raw["regr"]:
        name = w["name"]
        if name == "cchunk":
            continue
        kw = dict(w.get("transform", {}))
        rows, f0_or = [], []
        for b in range(Bp):
            if "prosody" in name:
                f0_or.append(SW.swipe(clean[b].astype(np.float64))[:f0_dev.shape[1]])
                X = D.prosody(clean[b], f0_or[-1], **kw)
            else:
                X = next(fn for k_, fn in fns.items() if k_ in name)(clean[b], **kw)
            rows.append(D.znorm(np.asarray(X, dtype=np.float64), stats[name][0], stats[name][1]))
        ref = torch.from_numpy(np.stack(rows)).float().to(dev)
        assert ref.shape == batch[name].shape, (name, ref.shape, batch[name].shape)
        ref_batch[name] = ref
        span = float(ref.max() - ref.min())
        err = (batch[name] - ref).abs()
        if "prosody" not in name:
            bad = float((err > 2e-3 * span).float().mean())
            # (log-power features of near-silent bins amplify round-off: a handful of elements may leave the band)
            assert bad <= 1e-4, (name, bad, float(err.max()), span)
            continue
        # prosody: the device SWIPE' against the oracle's, then the target on the frames both decide alike
        f0_or = np.stack(f0_or)
        Fp = ref.shape[2]
        v_o, v_d = f0_or > 0, f0_dev > 0
        agree = float((v_o == v_d).mean())
        both = v_o & v_d
        rel = np.abs(f0_dev[both] - f0_or[both]) / f0_or[both]
        print("prosody: voicing agreement %.4f, voiced on both %.3f of the frames, f0 within 1.5 %% on %.4f of those, "
              "median rel err %.2e" % (agree, float(both.mean()), float((rel <= 0.015).mean()), float(np.median(rel))))
        assert both.mean() > 0.3 and (~v_o).mean() > 0.1, "the crops must have voiced AND unvoiced frames"
        assert agree >= 0.99, agree
        assert (rel <= 0.015).mean() >= 0.99 and np.median(rel) < 1e-3
        # frames whose 9-frame Savitzky-Golay window (deltas) sees the same voicing decisions on both sides
        same = (v_o == v_d)[:, :Fp].astype(np.float64)
        win = np.stack([np.convolve(r_, np.ones(9), mode="same") for r_ in same]) >= 9 - 1e-9
        win[:, :4] = False
        win[:, -4:] = False        # (the 'interp' edge fit uses the first / last 9 frames)
        for b in range(Bp):
            win[b, :4] = win[b, -4:] = bool(same[b, :9].all()) and bool(same[b, -9:].all())
        wmask = torch.from_numpy(win).to(dev)
        assert float(wmask.float().mean()) >= 0.9
        nrow = ref.shape[1] // 3                            # [lf0, voiced, energy, zcr] x (static, delta, delta-delta)
        istd = torch.from_numpy(1.0 / stats[name][1]).to(dev)
        for r_ in range(ref.shape[1]):
            e = err[:, r_][wmask]
            if r_ % nrow == 0:
                # log-f0 (interpolated across unvoiced stretches from the neighbouring voiced frames): |d log f0| <= log(1.015)
                # on >= 99 % of the compared frames, in z-normalised units
                tol = float(np.log(1.015)) * float(istd[r_]) * (1.0 if r_ == 0 else 0.6)
                assert float((e <= tol).float().mean()) >= 0.99, (r_, float(e.max()), tol)
            else:
                assert float((e > 2e-3 * span).float().mean()) <= 1e-3, (r_, float(e.max()), span)
    # ---- one step on both sides -------------------------------------------------------------------------------------
    torch.manual_seed(2)
    with contextlib.redirect_stdout(io.StringIO()):
        tr = trainer(frontend_cfg=dict(fe), minions_cfg=wk, cfg=dict(fe_lr=1e-3, min_lr=5e-4, epoch=1, bpe=10 ** 6),
                     device=dev)
    m = tr.model
    m.train()
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    for n in names:
        P[n].requires_grad_(True)
    # (b) losses: the producer's own batch (device targets) against the comparator on the oracle's targets
    for opt in tr.optimizers():
        opt.zero_grad()
    lf = {k: float(v) for k, v in m.loss_and_grads(batch).items()}
    P0 = {k: v.detach().clone() for k, v in P.items()}
    lo, _ = _oracle_step(P, fe, raw, ref_batch)
    assert len(lo) == 13
    rel_l = {k: abs(lf[k] - v) / max(1.0, abs(v)) for k, v in lo.items()}
    print("producer-mode losses, relative difference:", {k: "%.1e" % v for k, v in rel_l.items()})
    for k, v in rel_l.items():
        # (the targets carry the DSP kernels' 1e-4-level differences into the losses; prosody additionally the frames on
        #  which the two SWIPE' evaluations decide differently: <= 1 % of them, each worth O(1) in the voiced-flag row)
        assert v <= (3e-2 if k in ("prosody", "total") else 1e-3), (k, lf[k], lo[k])
    assert rel_l["total"] <= 2e-3, rel_l["total"]
    # (c) gradients of every tensor on the SAME labels, judged against fp64 (as test_bs32_embedding_losses_and_elementwise_grads)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            v.copy_(sd0[k])
    for opt in tr.optimizers():
        opt.zero_grad()
    m.loss_and_grads(ref_batch)
    ref32 = {n: P[n].grad.detach().double().clone() for n in names}
    P64 = {k: (v.detach().double() if v.is_floating_point() else v.detach().clone()) for k, v in P0.items()}
    for n in names:
        P64[n].requires_grad_(True)
    _oracle_step(P64, fe, raw, {k: v.double() for k, v in ref_batch.items()})
    bad, checked, skipped, stats_ = [], 0, [], []
    for n, p in m.named_parameters():
        if is_noise_grad(n):
            continue
        t64 = P64[n].grad
        den = max(1e-300, float(t64.norm()))
        e_ours = float((p.grad.double() - t64).norm()) / den
        e_ref = float((ref32[n] - t64).norm()) / den
        if e_ref >= 0.5:
            skipped.append(n)
            continue
        per_channel = n.endswith(("norm.weight", "norm.bias", "act.weight", ".bias", "low_hz_", "band_hz_"))
        floor = 1.5e-3 if per_channel else 5e-4
        stats_.append((e_ours, n, e_ref))
        if not e_ours <= 1.5 * e_ref + floor:
            bad.append((n, "relL2 vs fp64: ours %.3e, torch fp32 %.3e" % (e_ours, e_ref)))
        checked += 1
    for r, n, rr in sorted(stats_, reverse=True)[:8]:
        print("   relL2 vs fp64: ours %.3e  torch fp32 %.3e  %s" % (r, rr, n))
    assert not bad, "%d tensors out of tolerance, first: %r" % (len(bad), bad[:3])
    assert checked >= 100 and len(skipped) <= 4, (checked, skipped)

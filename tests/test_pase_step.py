"""The full self-supervised step (encoder + workers + losses + backward + Adam) on the HIP kernels
vs the CPU oracle / the live-reference golden step."""
import os

import numpy as np
import pytest
import torch

from oracle import pase_oracle as O
from util import (GOLD, MINI_FE, assert_close, is_noise_grad, load_cfg, mini_workers, oracle_params, quiet,
                  randomize_affine, seed_all, synthetic_batch, with_losses)


def _mini_model(dev, seed=0):
    from pase_amd.pase import pase
    seed_all(seed)
    m = quiet(pase, frontend_cfg=dict(MINI_FE), minions_cfg=with_losses(mini_workers()), cls_lst=["mi", "cmi"],
              regr_lst=["cchunk", "lps", "prosody"])
    randomize_affine(m)
    return m.to(dev)


def _mini_batch(seed=1, B=2, T=1600):
    g = torch.Generator().manual_seed(seed)
    batch = {k: torch.randn(B, 1, T, generator=g) * 0.3 for k in ("chunk", "chunk_ctxt", "chunk_rand", "cchunk")}
    batch["lps"] = torch.randn(B, 5, T // 160, generator=g)
    batch["prosody"] = torch.randn(B, 3, T // 160, generator=g)
    return batch


def _check_grads(model, P, rtol=1e-3):
    for n, p in model.named_parameters():
        if is_noise_grad(n):
            continue
        ref = P[n].grad
        assert_close(p.grad, ref, rtol=rtol, atol=1e-4 * max(1e-2, float(ref.abs().max())), what=n)


def test_fused_step_losses_and_grads(dev):
    m = _mini_model(dev)
    P = oracle_params(m)
    batch = _mini_batch()
    raw = mini_workers()
    h, chunk, preds, labels = O.pase_forward(P, MINI_FE, raw, batch, True)
    lo = O.pase_losses(raw, preds, labels)
    lo["total"].backward()
    m.train()
    lf = m.loss_and_grads({k: v.to(dev) for k, v in batch.items()})
    for k, v in lo.items():
        assert abs(float(lf[k]) - float(v)) <= 1e-4 * max(1.0, abs(float(v))), (k, float(lf[k]), float(v))
    _check_grads(m, P)


@pytest.mark.parametrize("one_pass", [True, False], ids=["one-pass-tail", "six-launches"])
def test_fused_step_with_the_decoder_tail_at_its_real_width(dev, monkeypatch, one_pass):
    """cchunk's last deconvolution 128 channels wide and its MLPBlock 64: the shape whose pointwise tail has the one-pass
    forward + backward (pase_mlp_head1_step); same oracle comparison with the tail run layer by layer (PASE_MLP_HEAD1=0)."""
    from pase_amd import kernels as K
    from pase_amd.pase import pase
    monkeypatch.setenv("PASE_MLP_HEAD1", "1" if one_pass else "0")
    raw = mini_workers()
    raw["regr"][0].update(fmaps=[10, 8, 128], hidden_size=64)
    seed_all(11)
    wk = mini_workers()
    wk["regr"][0].update(fmaps=[10, 8, 128], hidden_size=64)
    m = quiet(pase, frontend_cfg=dict(MINI_FE), minions_cfg=with_losses(wk), cls_lst=["mi", "cmi"],
              regr_lst=["cchunk", "lps", "prosody"])
    randomize_affine(m)
    m = m.to(dev)
    P = oracle_params(m)
    batch = _mini_batch(seed=12, T=1600)
    h, chunk, preds, labels = O.pase_forward(P, MINI_FE, raw, batch, True)
    lo = O.pase_losses(raw, preds, labels)
    lo["total"].backward()
    m.train()
    n0 = K.MLP_HEAD1_CALLS
    lf = m.loss_and_grads({k: v.to(dev) for k, v in batch.items()})
    assert K.MLP_HEAD1_CALLS == n0 + (1 if one_pass else 0)
    for k, v in lo.items():
        assert abs(float(lf[k]) - float(v)) <= 1e-4 * max(1.0, abs(float(v))), (k, float(lf[k]), float(v))
    _check_grads(m, P)


def test_api_compat_forward_and_autograd(dev):
    """pase.forward(batch) -> (h, chunk, preds, labels) + worker.loss(...) + .backward(): the
    reference's own calling convention (trainer.py:229, worker_scheduler.py:43-75)."""
    m = _mini_model(dev, seed=4)
    P = oracle_params(m)
    batch = _mini_batch(seed=5)
    raw = mini_workers()
    h, chunk, preds, labels = O.pase_forward(P, MINI_FE, raw, batch, True)
    O.pase_losses(raw, preds, labels)["total"].backward()
    m.train()
    h2, chunk2, preds2, labels2 = m({k: v.to(dev) for k, v in batch.items()}, device=dev)
    assert set(preds2) == set(preds)
    tot = 0
    for w in list(m.classification_workers) + list(m.regression_workers):
        assert_close(preds2[w.name], preds[w.name], rtol=1e-4, atol=1e-4, what="pred " + w.name)
        assert_close(labels2[w.name], labels[w.name], rtol=0, atol=0, what="label " + w.name)
        tot = tot + w.loss_weight * w.loss(preds2[w.name], labels2[w.name])
    tot.backward()
    _check_grads(m, P)


def test_trainer_three_adam_steps(dev):
    """loss curve + parameters after 3 fused steps track the oracle + torch.optim.Adam."""
    from pase_amd.trainer import trainer
    seed_all(0)
    tr = quiet(trainer, frontend_cfg=dict(MINI_FE), minions_cfg=with_losses(mini_workers()),
               cfg=dict(fe_lr=1e-3, min_lr=5e-4, epoch=2, bpe=10), lr_mode="poly", device=dev)
    m = tr.model
    P = oracle_params(m)
    names = [n for n, _ in m.named_parameters()]
    opts = [torch.optim.Adam([P[n]], lr=1e-3 if n.startswith("frontend.") else 5e-4) for n in names]
    raw = mini_workers()
    for step in range(3):
        batch = _mini_batch(seed=10 + step)
        for o in opts:
            o.zero_grad()
        so = {}
        h, chunk, preds, labels = O.pase_forward(P, MINI_FE, raw, batch, True, so)
        lo = O.pase_losses(raw, preds, labels)
        lo["total"].backward()
        for o in opts:
            o.step()
        with torch.no_grad():
            for k, v in so.items():
                P["frontend." + k].copy_(v)
        lf = tr.train_step({k: v.to(dev) for k, v in batch.items()})
        assert abs(float(lf["total"]) - float(lo["total"])) <= 2e-4 * abs(float(lo["total"])), step
    for n, p in m.named_parameters():
        if not is_noise_grad(n):
            assert_close(p, P[n], rtol=0, atol=3e-4, what=n)
    sd = tr.frontend_optim.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    lrs = tr.adjust_lr(5, 0)
    assert abs(lrs["frontend"] - 1e-3 * (1 - 5 / 20) ** 0.9) < 1e-12


# What "fp32-grade" means at step level (round-2 review item): the live reference's OWN fp32 gradients of this step are
# compared with the same step evaluated by the live reference in fp64 (tests/golden/*_grads_f64.npz, oracle/make_golden.py
# `grads64`), and so are ours.  Per tensor e_ref = relL2(reference fp32, fp64), e_ours = relL2(HIP path, fp64).
# Measured (round 3, tests/golden step, B = 2): median over the 120 tensors 2.7e-6 (reference) / 3.2e-6 (split-bf16) /
# 3.3e-6 (fp32 MFMA); on PASE.cfg + workers.cfg every tensor of both pipes is within 1.5x of the reference's own error.
# What remains on PASE+ are a few encoder tensors at 2e-4 ... 1e-3: freshly initialised PReLU slopes are 0 (a ReLU), so an
# activation whose pre-activation differs from the fp64 value in the last bit AND straddles zero flips its backward mask --
# one flipped element moves a per-channel sum over 300 ... 19 200 signed terms by 1e-3 ... 1e-2 of itself.  Every fp32
# evaluation order has such flips (which elements flip is luck: the CPU reference has none on this batch, the k-ordered
# fp32 MFMA chain a few at 1e-4, the split-bf16 forward a few at 1e-3; `PASE_X6_ONLY=fwd|bwd` shows they come from the
# forward convolutions alone, whose outputs are CLOSER to fp64 than the fp32 chain's: tests/test_conv_x6c.py).  Hence
#   e_ours <= 1.5 * e_ref + FLOOR,  FLOOR = a handful of mask flips; a sign / permutation / missing-term error is O(1),
# and the round-2 single-accumulator kernels (systematic error, 3e-3 ... 5e-3 on ALL encoder tensors) fail it;
# both pipes must meet THE SAME bound and the split pipe's MEDIAN distance to fp64 must not exceed the fp32 pipe's.
GRAD_FLOOR_WEIGHT = 1.5e-3       # relL2, weight tensors
GRAD_FLOOR_PER_CHANNEL = 3e-3    # relL2, one scalar per channel (BN gamma / beta, PReLU slopes, biases, SincNet vectors)
# The "*_perturbed" goldens (round-3 review item 2) are the same two live-reference steps with the BatchNorm affines and the
# PReLU slopes moved off their init values (oracle/make_golden.py:perturb_affine == util.randomize_affine, same seed).  They
# exercise the affine / slope gradient paths with non-trivial values, but they do NOT remove the discrete events: a PReLU
# with slope a still has a kink of height 1 - a (0.6 ... 0.95 here), and the L1 loss of the decoder has a sign.  Measured
# (round 4): on the perturbed PASE+ golden the LIVE REFERENCE's own fp32 step is 5.5e-3 away from its fp64 self on
# blocks.2.norm.bias and 4.4e-3 on blocks.2.conv.weight -- more than at init -- while the HIP path is at 2e-6 there; on the
# perturbed PASE.cfg golden the split pipe has one flip in the upper encoder (1.4e-4 on the blocks below it) and the fp32 pipe
# one in the decoder (1.4e-3 on deconv.weight), each absent on the other pipe.  They are gated like the init-state goldens.
# The "*_smooth" goldens are the TIGHT gate: same perturbed affines, every PReLU slope exactly 1 (identity: no kink anywhere;
# every contraction, BatchNorm, scan, loss and the slope gradients themselves are still evaluated).  There the live
# reference's fp32 step is within 1.2e-6 (median) / 4.5e-5 (worst regular tensor) of its fp64 self, and every tensor of both
# pipes must be within 1.5 x that + 2e-5; tensors whose fp64 gradient is analytically zero there (BatchNorm betas feeding
# another BatchNorm through a linear map: |truth| < 1e-9) are skipped like the other noise gradients.
GRAD_FLOOR_SMOOTH = 2e-5
_PIPE_ERR = {}                 # (gold, pipe) -> {name: relL2 vs fp64}, to compare the two pipes with each other


@pytest.mark.parametrize("x6", [True, False], ids=["x6", "fp32pipe"])
@pytest.mark.parametrize("gold,fe,wk", [("pase_plus_step.npz", "frontend/PASE+.cfg", "workers/workers+.cfg"),
                                        ("pase_step_cfg2.npz", "frontend/PASE.cfg", "workers/workers.cfg"),
                                        ("pase_plus_step_perturbed.npz", "frontend/PASE+.cfg", "workers/workers+.cfg"),
                                        ("pase_step_cfg2_perturbed.npz", "frontend/PASE.cfg", "workers/workers.cfg"),
                                        ("pase_plus_step_smooth.npz", "frontend/PASE+.cfg", "workers/workers+.cfg"),
                                        ("pase_step_cfg2_smooth.npz", "frontend/PASE.cfg", "workers/workers.cfg"),
                                        ("pase_plus_step_bs32_smooth.npz", "frontend/PASE+.cfg", "workers/workers+.cfg"),
                                        ("pase_plus_step_bs32_perturbed.npz", "frontend/PASE+.cfg", "workers/workers+.cfg"),
                                        ("pase_step_cfg2_bs32_smooth.npz", "frontend/PASE.cfg", "workers/workers.cfg"),
                                        ("pase_plus_step_emb256_bs32_smooth.npz", "frontend/PASE+.cfg", "workers/workers+.cfg")],
                         ids=["plus", "cfg2", "plus-perturbed", "cfg2-perturbed", "plus-smooth", "cfg2-smooth", "plus-bs32-smooth",
                              "plus-bs32-perturbed", "cfg2-bs32-smooth", "plus-emb256-bs32-smooth"])
def test_full_width_golden_step(dev, gold, fe, wk, x6):
    """Full-width one step vs the live reference's trainer step: PASE+.cfg + workers+.cfg (12 workers)
    and PASE.cfg + workers.cfg (decoder, r-less regressors, SPC / LIM / GIM); on the split-bf16 pipe (the default)
    and on the exact-fp32 matrix pipe.  `plus-bs32-smooth` / `plus-bs32-perturbed` are the same judge on live-reference steps
    AT THE BENCHMARK'S OWN SIZE (32 utterances x 32 000 samples, BASELINE.json configs[2]; fp32 and fp64 runs of /root/reference
    in the build container, oracle/make_golden.py:gen_bs32), `cfg2-bs32-smooth` on BASELINE.json configs[1] at its full size
    (PASE.cfg + workers.cfg, 32 x 16 000), `plus-emb256-bs32-smooth` on configs[4]'s model (PASE+.cfg with rnn_layers = 2 and
    norm_type = 'lnorm': LayerNorm blocks, InstanceNorm norm_out, two QRNN layers) at 32 x 32 000: embedding and prediction combs, the 13 losses, gradient and post-Adam norms,
    and every parameter's sampled gradients against the reference's fp64 step at 1.5 x the reference's own fp32 error + 2e-5."""
    if dev.type == "cpu":
        pytest.skip("full-width step is GPU-only")
    if not os.path.exists(os.path.join(GOLD, gold)):
        pytest.fail("golden %s is missing (python oracle/make_golden.py ...)" % gold)
    from pase_amd import kernels as K
    saved = K.X6
    K.X6 = x6
    try:
        _full_width_golden_step(dev, gold, fe, wk, x6)
    finally:
        K.X6 = saved


def _full_width_golden_step(dev, gold, fe, wk, x6):
    import random
    from pase_amd.trainer import trainer
    g = np.load(os.path.join(GOLD, gold))
    raw = load_cfg(wk)
    seed_all(int(g["seed"]))
    fe_cfg = load_cfg(fe)
    if "fe_over" in g.files:        # WaveFe keyword arguments on top of the cfg file (the emb256 variant: rnn_layers, norm_type)
        import json
        fe_cfg = dict(fe_cfg, **json.loads(str(g["fe_over"])))
    tr = quiet(trainer, frontend_cfg=fe_cfg, minions_cfg=with_losses(load_cfg(wk)),
               cfg=dict(fe_lr=1e-3, min_lr=5e-4, epoch=1, bpe=10), lr_mode="poly", device=dev)
    batch = synthetic_batch(int(g["seed"]) + 1, int(g["B"]), int(g["T"]), raw["regr"])
    batch = {k: v.to(dev) for k, v in batch.items()}
    m = tr.model
    perturbed = "perturbed" in gold or "smooth" in gold
    smooth = "smooth" in gold
    if perturbed:
        from util import randomize_affine
        randomize_affine(m, smooth=smooth)   # the draw the live reference model got before its step (make_golden.perturb_affine)
    # same starting point as the live reference: per-tensor checksums of the state_dict (initial weights + perturbation)
    sd_now = m.state_dict()
    for k_, s_, q_ in zip((str(s) for s in g["param_names"]), g["param_sum"], g["param_sq"]):
        v_ = sd_now[k_].double()
        assert abs(float(v_.sum()) - s_) <= 1e-6 * max(1.0, abs(s_)) and abs(float((v_ ** 2).sum()) - q_) <= 1e-6 * max(1.0, q_), k_
    # API-compat forward first (train mode, same batch) for the prediction tensors
    m.train()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    random.seed(int(g["seed"]) + 2)
    h, chunk, preds, labels = m(batch, device=dev)
    if "compact" in g.files:
        # benchmark-size golden (oracle/make_golden.py:gen_bs32): the tensors are 6 ... 550 MB, the file holds strided combs
        # of them (comb_index) and their sums -- the embedding to the north-star tolerance at every sampled element
        from util import comb_index
        have = sorted(k_[:-5] for k_ in g.files if k_.endswith("_comb"))
        assert "chunk_emb" in have and len(have) >= 5, have
        for key in have:
            t, rtol = (chunk, 0.0) if key == "chunk_emb" else (preds[key[len("pred_"):]], 1e-4)
            flat = t.detach().reshape(-1)
            assert flat.numel() == int(g[key + "_numel"]), key
            idx = torch.as_tensor(comb_index(flat.numel(), g[key + "_comb"].size), device=dev)
            assert_close(flat[idx], g[key + "_comb"], rtol=rtol, atol=1e-4, what=key + " (comb)")
            sq = float((flat.double() ** 2).sum())
            assert abs(sq - float(g[key + "_sq"])) <= 1e-4 * max(1.0, float(g[key + "_sq"])), (key, sq, float(g[key + "_sq"]))
    else:
        assert_close(chunk, g["chunk_emb"], rtol=0, atol=1e-4, what="chunk embedding")
        assert_close(preds["mi"], g["pred_mi"], rtol=1e-4, atol=1e-4)
        assert_close(preds["cmi"], g["pred_cmi"], rtol=1e-4, atol=1e-4)
        if "pred_mfcc" in g.files:
            assert_close(preds["mfcc"], g["pred_mfcc"], rtol=1e-4, atol=1e-4)
        assert_close(preds["cchunk"][:, :, :400], g["pred_cchunk_head"], rtol=1e-4, atol=1e-4)
        if "pred_spc" in g.files:
            assert_close(preds["spc"], g["pred_spc"], rtol=1e-4, atol=1e-4)
    del h, chunk, preds, labels
    with torch.no_grad():      # undo the running-stat update of the extra forward
        for k, v in m.state_dict().items():
            v.copy_(sd0[k])
    random.seed(int(g["seed"]) + 2)
    losses = tr.train_step(batch)
    gl = dict(zip([str(s) for s in g["loss_names"]], g["loss_values"]))
    for k, v in gl.items():
        assert abs(float(losses[k]) - v) <= 1e-4 * max(1.0, abs(v)), (k, float(losses[k]), v)
    names = [str(s) for s in g["grad_names"]]
    params = dict(m.named_parameters())
    zero_names = set()
    if smooth:      # gradients that are analytically zero with identity activations: Adam turns their round-off into +-lr steps
        g64z = np.load(os.path.join(GOLD, gold.replace(".npz", "_grads_f64.npz")))
        zero_names = {str(n_) for n_, a_ in zip(g64z["grad_names"], g64z["grad_absmax"]) if float(a_) < 1e-9}
    # (LayerNorm over channels does not cancel a per-channel conv bias as BatchNorm does; the InstanceNorm norm_out cancels W's)
    noise = (lambda n_: n_ == "frontend.W.bias") if "emb256" in gold else is_noise_grad
    keep = [i for i, n in enumerate(names) if not noise(n) and n not in zero_names]
    gsq = torch.tensor([float((params[names[i]].grad.double() ** 2).sum()) for i in keep])
    assert_close(gsq.sqrt(), np.sqrt(g["grad_sq"][keep]), rtol=5e-3, atol=1e-6, what="grad norms")
    psq = torch.tensor([float((params[names[i]].detach().double() ** 2).sum()) for i in keep])
    assert_close(psq.sqrt(), np.sqrt(g["post_sq"][keep]), rtol=1e-4, atol=1e-5, what="post-Adam parameter norms")
    # ELEMENT-WISE gradients: every parameter, sampled at grad_sample_index(numel) -- the live reference's fp32
    # gradients (oracle/make_golden.py:gen_pase_step_grads) and the live reference's fp64 gradients of the same step
    from util import grad_sample_index
    gg = np.load(os.path.join(GOLD, gold.replace(".npz", "_grads.npz")))
    g64 = np.load(os.path.join(GOLD, gold.replace(".npz", "_grads_f64.npz")))
    assert [str(s) for s in gg["grad_names"]] == [str(s) for s in g64["grad_names"]]
    offs = gg["grad_offsets"]
    checked = 0
    stats, bad, mine, direct = [], [], {}, []
    for i, n in enumerate(str(s) for s in gg["grad_names"]):
        if noise(n) or n in zero_names:
            continue
        ref32 = torch.as_tensor(gg["grad_values"][offs[i]:offs[i + 1]]).double()
        truth = torch.as_tensor(g64["grad_values"][offs[i]:offs[i + 1]]).double()
        idx = torch.as_tensor(grad_sample_index(params[n].numel(), int(gg["n_samples"])), device=dev)
        got = params[n].grad.detach().reshape(-1)[idx].cpu().double()
        tn = float(truth.norm().clamp_min(1e-30))
        e_ref = float((ref32 - truth).norm()) / tn
        e_ours = float((got - truth).norm()) / tn
        gmax = float(g64["grad_absmax"][i])
        emax = float((got - truth).abs().max()) / max(gmax, 1e-30)
        per_channel = n.endswith(("norm.weight", "norm.bias", "act.weight", ".bias", "low_hz_", "band_hz_"))
        floor = GRAD_FLOOR_SMOOTH if smooth else (GRAD_FLOOR_PER_CHANNEL if per_channel else GRAD_FLOOR_WEIGHT)
        mine[n] = e_ours
        e_dir = float((got - ref32).norm()) / tn          # distance between the two fp32 evaluations themselves
        stats.append((e_ours, e_ref, emax, n))
        direct.append((e_dir, n))
        # a sign / permutation / missing-term error is O(1) in both measures
        if not (e_ours <= 1.5 * e_ref + floor and emax <= 10 * (1.5 * e_ref + floor)):
            bad.append("%s: relL2 vs fp64 %.3e (reference fp32: %.3e), max|err|/max|g| %.3e" % (n, e_ours, e_ref, emax))
        checked += 1
    stats.sort(reverse=True)
    print("gradients vs the fp64 reference step [%s, %s]: worst tensors (ours, reference fp32, max|err|/max|g|, name)"
          % (gold, "split-bf16" if x6 else "fp32 MFMA"))
    for st_ in stats[:10]:
        print("   %.3e  %.3e  %.3e  %s" % st_)
    direct.sort(reverse=True)
    print("   HIP path vs the live reference's fp32 gradients directly (relL2 / |fp64 truth|): median %.3e; worst %s"
          % (direct[len(direct) // 2][0], ", ".join("%.2e %s" % d_ for d_ in direct[:6])))
    med = sorted(s_[0] for s_ in stats)[len(stats) // 2]
    print("   median relL2 ours %.3e, reference fp32 %.3e" % (med, sorted(s_[1] for s_ in stats)[len(stats) // 2]))
    assert not bad, "%d tensors out of tolerance: %s" % (len(bad), "; ".join(bad[:4]))
    assert checked >= (100 if "plus" in gold else 40) - (8 if smooth else 0), checked
    if smooth:         # no mask flips to excuse: the median distance to fp64 is the reference's own, within 1.5x
        assert med <= 1.5 * sorted(s_[1] for s_ in stats)[len(stats) // 2] + 1e-6, med
    _PIPE_ERR[(gold, x6)] = mine
    other = _PIPE_ERR.get((gold, not x6))
    if other is not None:      # both pipes ran in this session: the split pipe is no farther from fp64 than the fp32 pipe
        ex6, ef32 = (mine, other) if x6 else (other, mine)
        ratio = float(np.median([ex6[n] / max(ef32[n], 1e-30) for n in ex6]))
        print("   split-bf16 vs fp32-MFMA distance to fp64: median ratio %.2f" % ratio)
        assert ratio <= 1.25, ratio


def _mini_workers_cfg2():
    """workers.cfg-shaped (BASELINE.json configs[1]): decoder, r-less MLP regressors, spc / mi / cmi."""
    return {"regr": [
        {"num_outputs": 1, "dropout": 0, "hidden_layers": 1, "name": "cchunk", "type": "decoder", "hidden_size": 6,
         "fmaps": [10, 8, 6], "strides": [4, 4, 10], "kwidths": [30, 30, 30], "loss": "L1Loss"},
        {"num_outputs": 9, "dropout": 0, "hidden_size": 9, "hidden_layers": 1, "name": "lps", "loss": "MSELoss",
         "skip": False},
        {"num_outputs": 4, "dropout": 0, "hidden_size": 7, "hidden_layers": 1, "name": "prosody", "loss": "MSELoss",
         "skip": False}],
        "cls": [
        {"num_outputs": 1, "dropout": 0, "hidden_size": 8, "hidden_layers": 1, "name": "spc", "type": "spc",
         "loss": "BCEWithLogitsLoss", "skip": False},
        {"num_outputs": 1, "dropout": 0, "hidden_size": 8, "hidden_layers": 1, "name": "mi",
         "loss": "BCEWithLogitsLoss", "skip": False},
        {"num_outputs": 1, "dropout": 0, "hidden_size": 8, "hidden_layers": 1, "name": "cmi",
         "loss": "BCEWithLogitsLoss", "skip": False}]}


def test_config2_shaped_step_with_spc(dev):
    """PASE.cfg-shaped encoder (no QRNN / skips) + workers.cfg-shaped heads incl. the SPC worker, whose
    frame sampling consumes Python's `random` stream exactly like the reference (minions.py:614-628)."""
    import random
    from pase_amd.pase import pase
    from util import MINI_FE_PLAIN
    seed_all(7)
    m = quiet(pase, frontend_cfg=dict(MINI_FE_PLAIN), minions_cfg=with_losses(_mini_workers_cfg2()),
              cls_lst=["spc", "mi", "cmi"], regr_lst=["cchunk", "lps", "prosody"])
    randomize_affine(m)
    m = m.to(dev)
    P = oracle_params(m)
    B, T = 2, 8000
    g = torch.Generator().manual_seed(3)
    batch = {k: torch.randn(B, 1, T, generator=g) * 0.3 for k in ("chunk", "chunk_ctxt", "chunk_rand", "cchunk")}
    batch["lps"] = torch.randn(B, 9, T // 160, generator=g)
    batch["prosody"] = torch.randn(B, 4, T // 160, generator=g)
    raw = _mini_workers_cfg2()
    random.seed(11)
    h, chunk, preds, labels = O.pase_forward(P, MINI_FE_PLAIN, raw, batch, True)
    lo = O.pase_losses(raw, preds, labels)
    lo["total"].backward()
    m.train()
    random.seed(11)
    lf = m.loss_and_grads({k: v.to(dev) for k, v in batch.items()})
    for k, v in lo.items():
        assert abs(float(lf[k]) - float(v)) <= 1e-4 * max(1.0, abs(float(v))), (k, float(lf[k]), float(v))
    _check_grads(m, P)
    # API-compatible path draws the same frames under the same seed
    m.zero_grad()
    random.seed(11)
    h2, chunk2, preds2, labels2 = m({k: v.to(dev) for k, v in batch.items()}, device=dev)
    assert_close(preds2["spc"], preds["spc"], rtol=1e-3, atol=1e-3)


def test_step_with_gap_worker(dev):
    """The Gap worker (cls_minions.py:117-131, minions.py:651-704; no cfg ships with the reference): frame
    pairs drawn from numpy's global RNG in the reference's order, label truncation as written."""
    from pase_amd.pase import pase
    from util import MINI_FE_PLAIN

    def workers():
        return {"regr": [{"num_outputs": 9, "dropout": 0, "hidden_size": 9, "hidden_layers": 1, "name": "lps",
                          "context": 1, "r": 3, "loss": "MSELoss", "skip": False}],
                "cls": [{"num_outputs": 1, "dropout": 0, "hidden_size": 10, "hidden_layers": 1, "name": "gap",
                         "type": "gap", "loss": "MSELoss", "skip": False, "loss_weight": 2.0},
                        {"num_outputs": 1, "dropout": 0, "hidden_size": 8, "hidden_layers": 1, "name": "mi",
                         "loss": "BCEWithLogitsLoss", "skip": False}]}
    seed_all(9)
    m = quiet(pase, frontend_cfg=dict(MINI_FE_PLAIN), minions_cfg=with_losses(workers()), cls_lst=["gap", "mi"],
              regr_lst=["lps"])
    randomize_affine(m)
    m = m.to(dev)
    P = oracle_params(m)
    B, T = 3, 3200
    g = torch.Generator().manual_seed(5)
    batch = {k: torch.randn(B, 1, T, generator=g) * 0.3 for k in ("chunk", "chunk_ctxt", "chunk_rand")}
    batch["lps"] = torch.randn(B, 9, T // 160, generator=g)
    raw = workers()
    np.random.seed(21)
    h, chunk, preds, labels = O.pase_forward(P, MINI_FE_PLAIN, raw, batch, True)
    lo = O.pase_losses(raw, preds, labels)
    lo["total"].backward()
    m.train()
    np.random.seed(21)
    lf = m.loss_and_grads({k: v.to(dev) for k, v in batch.items()})
    for k, v in lo.items():
        assert abs(float(lf[k]) - float(v)) <= 1e-4 * max(1.0, abs(float(v))), (k, float(lf[k]), float(v))
    _check_grads(m, P)
    np.random.seed(21)
    h2, chunk2, preds2, labels2 = m({k: v.to(dev) for k, v in batch.items()}, device=dev)
    assert_close(preds2["gap"], preds["gap"], rtol=1e-3, atol=1e-3)
    assert_close(labels2["gap"], labels["gap"], rtol=0, atol=0)


def test_gap_worker_vs_live_reference(dev):
    """The Gap worker against the LIVE reference's output (cls_minions.py:117-131 -> minions.py:651-704, run in
    oracle/live_transforms.py with the legacy LongTensor adapter): same parameters, same numpy seed -> same frame
    pairs, prediction and truncated labels."""
    from pase_amd.minions import cls_worker_maker
    g = np.load(os.path.join(GOLD, "transforms_live.npz"))
    cfg = {"num_outputs": 1, "dropout": 0, "hidden_size": 16, "hidden_layers": 1, "name": "gap", "type": "gap",
           "loss": "MSELoss", "skip": False}
    w = quiet(cls_worker_maker, with_losses({"cls": [cfg]})["cls"][0], 12)
    sd = {str(n): torch.from_numpy(g["gap_p_" + str(n)]) for n in g["gap_param_names"]}
    assert set(sd) == set(k for k, _ in w.named_parameters())
    w.load_state_dict(sd)
    w = w.to(dev)
    np.random.seed(701)
    with torch.no_grad():
        y, lab = w(torch.from_numpy(g["gap_x"]).to(dev), 1, device=dev)
    assert_close(y, g["gap_y"], rtol=1e-4, atol=1e-5, what="gap prediction")
    assert_close(lab, g["gap_label"], rtol=0, atol=0, what="gap label")


def test_fused_step_config5_shaped_lnorm_2xqrnn(dev):
    """BASELINE.json configs[4] shape at mini width: dense-skip encoder with norm_type='lnorm', two QRNN layers,
    InstanceNorm norm_out + the workers: fused step losses and every gradient vs the oracle."""
    from pase_amd.pase import pase
    fe_cfg = dict(MINI_FE, rnn_layers=2, norm_type="lnorm")
    seed_all(12)
    m = quiet(pase, frontend_cfg=dict(fe_cfg), minions_cfg=with_losses(mini_workers()), cls_lst=["mi", "cmi"],
              regr_lst=["cchunk", "lps", "prosody"])
    randomize_affine(m)
    m = m.to(dev)
    P = oracle_params(m)
    batch = _mini_batch(seed=13)
    raw = mini_workers()
    h, chunk, preds, labels = O.pase_forward(P, fe_cfg, raw, batch, True)
    lo = O.pase_losses(raw, preds, labels)
    lo["total"].backward()
    m.train()
    lf = m.loss_and_grads({k: v.to(dev) for k, v in batch.items()})
    for k, v in lo.items():
        assert abs(float(lf[k]) - float(v)) <= 1e-4 * max(1.0, abs(float(v))), (k, float(lf[k]), float(v))
    for n, p in m.named_parameters():
        if n == "frontend.W.bias":          # cancelled by the InstanceNorm norm_out
            continue
        ref = P[n].grad
        assert_close(p.grad, ref, rtol=1e-3, atol=1e-4 * max(1e-2, float(ref.abs().max())), what=n)


def test_trainer_epoch_loop_checkpoints_and_resume(dev, tmp_path):
    """trainer.train_ (trainer.py:200-278): epoch loop over a host-side loader (on the GPU the batches travel through
    the pinned feeder), poly LR at the log points, FE_e{e}.ckpt + Saver checkpoints per epoch, resume_training."""
    from pase_amd.trainer import trainer
    seed_all(3)
    cfg = dict(fe_lr=1e-3, min_lr=5e-4, epoch=2, bpe=2, log_freq=1, save_path=str(tmp_path / "ck"))
    tr = quiet(trainer, frontend_cfg=dict(MINI_FE), minions_cfg=with_losses(mini_workers()), cfg=cfg, lr_mode="poly",
               device=dev)
    loader = [_mini_batch(seed=20 + i) for i in range(3)]          # shorter than epoch * bpe: the iterator restarts
    w0 = tr.model.frontend.W.weight.detach().clone()
    quiet(tr.train_, loader, device=dev)
    assert not torch.equal(w0, tr.model.frontend.W.weight.detach())
    assert os.path.exists(os.path.join(cfg["save_path"], "FE_e1.ckpt"))
    assert int(tr.frontend_optim.step_t.item()) == 4
    tr2 = quiet(trainer, frontend_cfg=dict(MINI_FE), minions_cfg=with_losses(mini_workers()), cfg=cfg, lr_mode="poly",
                device=dev)
    assert quiet(tr2.resume_training, dev) and tr2.epoch_beg == 2
    for (n, p), (_, q) in zip(tr.model.named_parameters(), tr2.model.named_parameters()):
        assert torch.equal(p.detach().cpu(), q.detach().cpu()), n


def test_trainer_eval_pass_matches_the_oracle(dev, tmp_path):
    """trainer._eval (trainer.py:282-337): eval-mode forward (BatchNorm running statistics), unweighted per-worker losses
    averaged over va_bpe validation batches, parameters / running statistics untouched, training mode restored; and
    train_ runs it once per epoch when a validation loader is given."""
    from pase_amd.trainer import trainer
    seed_all(5)
    cfg = dict(fe_lr=1e-3, min_lr=5e-4, epoch=1, bpe=1, va_bpe=3, log_freq=1, save_path=str(tmp_path / "ck"))
    tr = quiet(trainer, frontend_cfg=dict(MINI_FE), minions_cfg=with_losses(mini_workers()), cfg=cfg, lr_mode="poly",
               device=dev)
    m = tr.model
    randomize_affine(m)
    m.train()
    quiet(tr.train_step, {k: v.to(dev) for k, v in _mini_batch(seed=50).items()})     # running statistics off their init
    val = [_mini_batch(seed=60 + i) for i in range(2)]          # shorter than va_bpe: the iterator restarts
    P = oracle_params(m)
    raw = mini_workers()
    want = {}
    for i in range(3):
        b = val[i % 2]
        h, chunk, preds, labels = O.pase_forward(P, MINI_FE, raw, b, False)
        lo = {}
        for grp in ("cls", "regr"):          # UNWEIGHTED, as trainer._eval sums them (trainer.py:305-317)
            for w in raw[grp]:
                lo[w["name"]] = float(O.ctx_loss(preds[w["name"]], labels[w["name"]], w["loss"], w.get("r")))
        lo["total"] = sum(lo.values())
        for k, v in lo.items():
            want.setdefault(k, []).append(float(v))
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    got = quiet(tr._eval, val, 0, dev)
    assert m.training
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd0[k]), k
    assert set(got) == set(want)
    for k in want:
        ref = sum(want[k]) / len(want[k])
        assert abs(got[k] - ref) <= 2e-4 * max(1.0, abs(ref)), (k, got[k], ref)
    quiet(tr.train_, [_mini_batch(seed=70)], val, dev)
    assert tr.last_eval is not None and "total" in tr.last_eval


@pytest.mark.gpu
def test_captured_step_matches_eager_steps():
    """trainer.capture_step: the hipGraph replay of the fused step (side streams, arenas, multi-tensor commits, Adam
    with device-scalar lr / step) tracks an identical trainer stepping eagerly, over steps with changing batches and a
    learning-rate change between them."""
    from pase_amd.trainer import trainer
    dev = torch.device("cuda:0")

    def make():
        seed_all(0)
        return quiet(trainer, frontend_cfg=dict(MINI_FE), minions_cfg=with_losses(mini_workers()),
                     cfg=dict(fe_lr=1e-3, min_lr=5e-4, epoch=2, bpe=10), lr_mode="poly", device=dev)
    a, b = make(), make()
    ex = {k: v.to(dev) for k, v in _mini_batch(seed=30).items()}
    snap = {k: v.clone() for k, v in b.model.state_dict().items()}
    b.capture_step(ex)
    # capture runs warm-up steps on the example batch and must leave the training state where it found it
    for k, v in b.model.state_dict().items():
        assert torch.equal(v, snap[k]), k
    for opt in b.optimizers():
        assert int(opt.step_t.item()) == 0 and float(opt.exp_avg.abs().max()) == 0.0
    for step in range(4):
        batch = {k: v.to(dev) for k, v in _mini_batch(seed=40 + step).items()}
        if step == 2:
            a.adjust_lr(5, 0)
            b.adjust_lr(5, 0)
        la = a.train_step(batch)
        lb = b.train_step(batch)
        assert abs(float(la["total"]) - float(lb["total"])) <= 1e-5 * abs(float(la["total"])), step
    for (n, p), (_, q) in zip(a.model.named_parameters(), b.model.named_parameters()):
        if not is_noise_grad(n):
            assert_close(q, p, rtol=0, atol=2e-5, what=n)
    assert int(b.frontend_optim.step_t.item()) == 4
    # a batch of another shape falls back to the eager step on its OWN zero arena (the graph keeps raw pointers into its)
    small = {k: v[:1].contiguous() for k, v in batch.items()}
    b.train_step(small)
    assert b._zero_arena is not b._graph_arena
    lb2 = b.train_step(batch)                      # and the graph still replays
    assert torch.isfinite(lb2["total"])


@pytest.mark.gpu
def test_capture_refuses_workers_with_host_randomness():
    """SPC / Gap draw their frames with `random` / `numpy.random` on the host at every step (minions.py:614-628, 680-681):
    a captured step would freeze the draw of the capture run."""
    from pase_amd.trainer import trainer
    from util import MINI_FE_PLAIN
    dev = torch.device("cuda:0")
    seed_all(0)
    tr = quiet(trainer, frontend_cfg=dict(MINI_FE_PLAIN), minions_cfg=with_losses(_mini_workers_cfg2()),
               cfg=dict(fe_lr=1e-3, min_lr=5e-4, epoch=1, bpe=10), lr_mode="poly", device=dev)
    g = torch.Generator().manual_seed(3)
    batch = {k: (torch.randn(2, 1, 8000, generator=g) * 0.3).to(dev) for k in ("chunk", "chunk_ctxt", "chunk_rand", "cchunk")}
    batch["lps"] = torch.randn(2, 9, 50, generator=g).to(dev)
    batch["prosody"] = torch.randn(2, 4, 50, generator=g).to(dev)
    with pytest.raises(NotImplementedError):
        tr.capture_step(batch)

"""Generate the golden vectors under tests/golden/ by running the LIVE reference
(/root/reference, imported through oracle/ref_shim.py) on seeded inputs.

Run in the build container only (the reference tree does not exist on the GPU box):
    python oracle/make_golden.py
The .npz files are small and committed; tests compare both the oracle restatement
(oracle/pase_oracle.py) and the HIP implementation against them.

Weights are NOT stored: the reference builds its modules under torch.manual_seed(seed) and the
mirrors in pase_amd/ create the same nn.Conv1d / nn.Linear / ... in the same order, so the same seed
reproduces the same initial weights (checked by tests/test_oracle_pins.py against the checksums
stored here).
"""
import contextlib
import io
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = ref_shim.REFERENCE_ROOT


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def param_checksums(sd):
    names = list(sd.keys())
    sums = np.array([float(sd[k].double().sum()) for k in names])
    sq = np.array([float((sd[k].double() ** 2).sum()) for k in names])
    return names, sums, sq


def gen_sinc():
    from pase.models.modules import SincConv_fast
    m = SincConv_fast(1, 64, 251, padding="SAME")
    x = torch.zeros(1, 1, 400)
    m(x)
    np.savez(os.path.join(GOLD, "sinc_init.npz"), low_hz_=m.low_hz_.detach().numpy(),
             band_hz_=m.band_hz_.detach().numpy(), filters=m.filters.detach().numpy(),
             window_=m.window_.numpy(), n_=m.n_.numpy())
    # perturbed parameters (negative values, clamping at sr/2) + gradient of a seeded functional
    seed_all(5)
    with torch.no_grad():
        m.low_hz_.mul_(torch.empty_like(m.low_hz_).uniform_(-1.2, 1.2))
        m.band_hz_.mul_(torch.empty_like(m.band_hz_).uniform_(-30.0, 30.0))
    x = torch.randn(2, 1, 700) * 0.5
    y = m(x)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    np.savez(os.path.join(GOLD, "sinc_perturbed.npz"), low_hz_=m.low_hz_.detach().numpy(),
             band_hz_=m.band_hz_.detach().numpy(), filters=m.filters.detach().numpy(), x=x.numpy(),
             y=y.detach().numpy(), g=g.numpy(), dlow=m.low_hz_.grad.numpy(), dband=m.band_hz_.grad.numpy())


def gen_wavefe(tag, cfg_file, seed, S, T):
    from pase.models.frontend import wf_builder
    seed_all(seed)
    m = quiet(wf_builder, os.path.join(REF, "cfg", "frontend", cfg_file))
    names, sums, sq = param_checksums(m.state_dict())
    seed_all(seed + 1)
    x = torch.randn(S, 1, T) * 0.1
    m.train()
    y_tr = m(x)
    g = torch.randn_like(y_tr)
    (y_tr * g).sum().backward()
    gnames = [n for n, p in m.named_parameters()]
    gsum = np.array([float(p.grad.double().sum()) for n, p in m.named_parameters()])
    gsq = np.array([float((p.grad.double() ** 2).sum()) for n, p in m.named_parameters()])
    rm = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
    m.eval()
    with torch.no_grad():
        y_ev = m(x)
        y_avg = m(x, mode="avg_norm")
    np.savez(os.path.join(GOLD, "wavefe_%s.npz" % tag), seed=seed, S=S, T=T, param_names=np.array(names),
             param_sum=sums, param_sq=sq, x=x.numpy(), y_train=y_tr.detach().numpy(), g=g.numpy(),
             grad_names=np.array(gnames), grad_sum=gsum, grad_sq=gsq, y_eval=y_ev.numpy(), y_avg_norm=y_avg.numpy(),
             running_names=np.array(list(rm.keys())), running_sum=np.array([float(v.double().sum()) for v in rm.values()]))


def synthetic_batch(seed, B, T, workers_cfg):
    """chunk/chunk_ctxt/chunk_rand/cchunk ~ 0.1 N(0,1) clamp +-1; targets ~ N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    batch = {k: (0.1 * torch.randn(B, 1, T, generator=g)).clamp_(-1, 1)
             for k in ("chunk", "chunk_ctxt", "chunk_rand", "cchunk")}
    for w in workers_cfg["regr"]:
        if w["name"] not in batch:
            batch[w["name"]] = torch.randn(B, w["num_outputs"], T // 160, generator=g)
    return batch


def perturb_affine(module, seed=123, smooth=False):
    """BatchNorm affines and PReLU slopes moved off their init values -- the SAME seeded draw, in named_parameters order,
    as tests/util.py:randomize_affine applies to the HIP model.  At init every encoder PReLU slope is 0 (a ReLU): a forward
    value that differs from the fp64 one in the last bit can flip a backward mask, which hides everything below ~1e-3 in the
    gradient comparison; with slopes in [0.05, 0.4] a flip changes a gradient element by at most (1 - slope) of a value that
    is itself ~0, so two fp32 evaluations of the step agree to ~1e-5 and the gate can be that tight."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("norm.weight"):
                p.copy_(torch.empty(p.shape).uniform_(0.5, 1.5, generator=g))
            elif n.endswith("norm.bias"):
                p.copy_(torch.empty(p.shape).normal_(0, 0.2, generator=g))
            elif n.endswith("act.weight"):
                p.copy_(torch.empty(p.shape).uniform_(0.05, 0.4, generator=g))
                if smooth:
                    # slope exactly 1: PReLU is the identity, the step has NO kink left (the perturbed draw above still has
                    # one of height 1 - slope at every activation: measured, the reference's own fp32 step is 5.5e-3 away from
                    # its fp64 self on blocks.2.norm.bias there).  Every contraction, BatchNorm, scan, loss and the slope
                    # gradient itself are still evaluated; what is left of discrete events is the sign of the L1 loss.
                    p.fill_(1.0)


def _ref_step(seed, B, T, fe_name, wk_name, double=False, perturb=False, fe_over=None):
    """One reference training step (trainer.py:229-232 -> worker_scheduler._base_scheduler) of a
    frontend cfg + workers cfg on a seeded synthetic batch.  Returns (model, param checksums, losses,
    chunk, preds); _base_scheduler has stepped the optimizers, so `.grad` holds the step's gradients
    and the parameters are post-Adam."""
    from pase.models.pase import pase
    from pase.utils import worker_parser
    from pase.models.WorkerScheduler.worker_scheduler import backprop_scheduler
    import torch.optim as optim
    with open(os.path.join(REF, "cfg", "frontend", fe_name)) as f:
        fe_cfg = dict(json.load(f), **(fe_over or {}))      # (fe_over: WaveFe keyword arguments on top of the cfg file)
    minions_cfg = quiet(worker_parser, os.path.join(REF, "cfg", "workers", wk_name))
    for _t, lst in minions_cfg.items():
        for c in lst:
            c.pop("transform", None)          # train.py:64
    with open(os.path.join(REF, "cfg", "workers", wk_name)) as f:
        raw_cfg = json.load(f)
    seed_all(seed)
    model = quiet(pase, frontend_cfg=fe_cfg, minions_cfg=minions_cfg,
                  cls_lst=[w["name"] for w in raw_cfg["cls"]], regr_lst=[w["name"] for w in raw_cfg["regr"]])
    if perturb:
        perturb_affine(model, smooth=(perturb == "smooth"))
    names, sums, sq = param_checksums(model.state_dict())
    batch = synthetic_batch(seed + 1, B, T, raw_cfg)
    if double:
        # the same step in fp64: the fp32 initial weights and the fp32 batch converted exactly (the reference modules
        # are dtype-agnostic; cls_minions.make_labels builds fp32 labels, which BCEWithLogitsLoss needs in the
        # prediction's dtype -> default dtype fp64 for the duration of the step)
        model = model.double()
        for mod in model.modules():        # SincConv_fast keeps n_ / window_ as plain attributes, not buffers
            for attr in ("n_", "window_"):
                if torch.is_tensor(getattr(mod, attr, None)):
                    setattr(mod, attr, getattr(mod, attr).double())
        batch = {k: v.double() for k, v in batch.items()}
        torch.set_default_dtype(torch.float64)
    fe_opt = optim.Adam(model.frontend.parameters(), lr=1e-3)
    cls_opt = {w.name: optim.Adam(w.parameters(), lr=5e-4) for w in model.classification_workers}
    regr_opt = {w.name: optim.Adam(w.parameters(), lr=5e-4) for w in model.regression_workers}
    sched = backprop_scheduler(model, mode="base")
    model.train()
    random.seed(seed + 2)          # the SPC worker draws its frames from Python's `random`
    h, chunk, preds, labels = model.forward(batch, 1, "cpu")
    try:
        losses, _ = sched(preds, labels, cls_opt, regr_opt, fe_opt, device="cpu")
    finally:
        torch.set_default_dtype(torch.float32)
    return model, (names, sums, sq), losses, chunk, preds


def gen_pase_step(seed, B, T, fe_name="PASE+.cfg", wk_name="workers+.cfg", out="pase_plus_step.npz", perturb=False):
    """losses, gradient norms, post-Adam norms of one reference step."""
    model, (names, sums, sq), losses, chunk, preds = _ref_step(seed, B, T, fe_name, wk_name, perturb=perturb)
    gnames = [n for n, p in model.named_parameters()]
    gsq = np.array([float((p.grad.double() ** 2).sum()) for n, p in model.named_parameters()])
    gsum = np.array([float(p.grad.double().sum()) for n, p in model.named_parameters()])
    post_sq = np.array([float((p.detach().double() ** 2).sum()) for n, p in model.named_parameters()])
    post_sum = np.array([float(p.detach().double().sum()) for n, p in model.named_parameters()])
    extra = {}
    if "mfcc" in preds:
        extra["pred_mfcc"] = preds["mfcc"].detach().numpy()
    if "spc" in preds:
        extra["pred_spc"] = preds["spc"].detach().numpy()
    np.savez(os.path.join(GOLD, out), seed=seed, B=B, T=T, param_names=np.array(names),
             param_sum=sums, param_sq=sq, loss_names=np.array(list(losses.keys())),
             loss_values=np.array([float(v) for v in losses.values()]), chunk_emb=chunk.detach().numpy(),
             grad_names=np.array(gnames), grad_sq=gsq, grad_sum=gsum, post_sq=post_sq, post_sum=post_sum,
             pred_mi=preds["mi"].detach().numpy(), pred_cmi=preds["cmi"].detach().numpy(),
             pred_cchunk_head=preds["cchunk"].detach().numpy()[:, :, :400], **extra)


GRAD_SAMPLES = 2048


def grad_sample_index(numel, n=GRAD_SAMPLES):
    """Flat indices at which element-wise gradients are stored: every element of small tensors, an
    evenly strided comb (odd stride so it walks all rows / columns / taps) of large ones."""
    if numel <= n:
        return np.arange(numel)
    st = numel // n
    st += (st % 2 == 0)
    return (np.arange(n) * st) % numel


def gen_pase_step_grads(seed, B, T, fe_name="PASE+.cfg", wk_name="workers+.cfg", out="pase_plus_step_grads.npz",
                        double=False, perturb=False):
    """ELEMENT-WISE reference gradients of the same step as gen_pase_step (same seeds => same step): for
    every parameter, the gradient at grad_sample_index(numel) flat positions, plus each tensor's max |grad|.
    double=True: the same step evaluated by the live reference in fp64 (same fp32 initial weights, same fp32 batch):
    the TRUTH both fp32 implementations (the reference's own and the HIP path) are measured against."""
    model, _cs, losses, chunk, preds = _ref_step(seed, B, T, fe_name, wk_name, double=double, perturb=perturb)
    gnames, vals, offs, gmax = [], [], [0], []
    for n, p in model.named_parameters():
        g = p.grad.detach().reshape(-1).numpy()
        idx = grad_sample_index(g.size)
        gnames.append(n)
        vals.append(g[idx].astype(np.float64 if double else np.float32))
        offs.append(offs[-1] + idx.size)
        gmax.append(float(np.abs(g).max()))
    np.savez(os.path.join(GOLD, out), seed=seed, B=B, T=T, grad_names=np.array(gnames),
             grad_values=np.concatenate(vals), grad_offsets=np.array(offs), grad_absmax=np.array(gmax),
             n_samples=GRAD_SAMPLES, loss_total=float(losses["total"]))


def gen_perturbed(only=None):
    """The two full-width steps again with BN affines / PReLU slopes off their init values (perturb_affine): fp32 step,
    element-wise fp32 gradients and the fp64 truth; and the "smooth" variant (slopes = 1: no kink) -- the TIGHT live-reference
    gradient gate (tests/test_pase_step.py::test_full_width_golden_step[*-smooth])."""
    for seed, fe, wk, stem, mode in ((2, "PASE+.cfg", "workers+.cfg", "pase_plus_step_perturbed", True),
                                     (4, "PASE.cfg", "workers.cfg", "pase_step_cfg2_perturbed", True),
                                     (2, "PASE+.cfg", "workers+.cfg", "pase_plus_step_smooth", "smooth"),
                                     (4, "PASE.cfg", "workers.cfg", "pase_step_cfg2_smooth", "smooth")):
        if only and only not in stem:
            continue
        gen_pase_step(seed=seed, B=2, T=8000, fe_name=fe, wk_name=wk, out=stem + ".npz", perturb=mode)
        gen_pase_step_grads(seed=seed, B=2, T=8000, fe_name=fe, wk_name=wk, out=stem + "_grads.npz", perturb=mode)
        gen_pase_step_grads(seed=seed, B=2, T=8000, fe_name=fe, wk_name=wk, out=stem + "_grads_f64.npz", double=True,
                            perturb=mode)


def comb_index(numel, n):
    """n evenly strided flat positions (odd stride) of a tensor with numel elements -- the embedding / prediction samples of
    the benchmark-size golden (the tensors themselves are 6 ... 550 MB)."""
    if numel <= n:
        return np.arange(numel)
    st = numel // n
    st += (st % 2 == 0)
    return (np.arange(n) * st) % numel


def gen_bs32(mode="smooth", seed=2, B=32, T=32000, stem=None, only=None, fe_name="PASE+.cfg", wk_name="workers+.cfg", fe_over=None):
    """ONE live-reference step at the BENCHMARK's own size (BASELINE.json configs[2]: PASE+.cfg + workers+.cfg, 32 utterances x
    32 000 samples), fp32 and fp64 -- round-5 review: the bs32 gates compared the HIP path with the oracle port only; this is
    the reference-generated anchor at that size.  Stored compactly (a few MB): parameter checksums, the 13 losses, per-tensor
    checksums + a 65 536-sample comb of the embedding, combs of four prediction tensors, gradient / post-Adam norms and the
    2 048-sample gradient combs of every parameter in fp32 and fp64 (same files / same keys as the B = 2 goldens, so
    tests/test_pase_step.py judges them with the same code).  About 17 GB (fp32) / 35 GB (fp64) of host memory and ~20 min
    on 8 cores: run in the build container, `python oracle/make_golden.py bs32 [smooth|perturbed] [f32|f64]`."""
    stem = stem or ("pase_plus_step_bs32_%s" % mode)
    perturb = "smooth" if mode == "smooth" else True
    if only in (None, "f32"):
        model, (names, sums, sq), losses, chunk, preds = _ref_step(seed, B, T, fe_name, wk_name, perturb=perturb, fe_over=fe_over)
        gnames = [n for n, p in model.named_parameters()]
        gsq = np.array([float((p.grad.double() ** 2).sum()) for n, p in model.named_parameters()])
        gsum = np.array([float(p.grad.double().sum()) for n, p in model.named_parameters()])
        post_sq = np.array([float((p.detach().double() ** 2).sum()) for n, p in model.named_parameters()])
        post_sum = np.array([float(p.detach().double().sum()) for n, p in model.named_parameters()])
        combs = {}
        for key, t in [("chunk_emb", chunk)] + [("pred_" + k_, preds[k_]) for k_ in ("mi", "cmi", "mfcc", "cchunk", "lps", "spc")
                                                if k_ in preds]:
            flat = t.detach().reshape(-1)
            idx = comb_index(flat.numel(), 65536 if key == "chunk_emb" else 16384)
            combs[key + "_comb"] = flat[torch.as_tensor(idx)].numpy()
            combs[key + "_sum"] = float(flat.double().sum())
            combs[key + "_sq"] = float((flat.double() ** 2).sum())
            combs[key + "_numel"] = flat.numel()
        np.savez(os.path.join(GOLD, stem + ".npz"), seed=seed, B=B, T=T, compact=1, fe_over=json.dumps(fe_over or {}),
                 param_names=np.array(names), param_sum=sums,
                 param_sq=sq, loss_names=np.array(list(losses.keys())), loss_values=np.array([float(v) for v in losses.values()]),
                 grad_names=np.array(gnames), grad_sq=gsq, grad_sum=gsum, post_sq=post_sq, post_sum=post_sum, **combs)
        _save_grad_comb(model, losses, os.path.join(GOLD, stem + "_grads.npz"), seed, B, T, False)
        del model, chunk, preds, losses
    if only in (None, "f64"):
        model, _cs, losses, chunk, preds = _ref_step(seed, B, T, fe_name, wk_name, double=True, perturb=perturb, fe_over=fe_over)
        _save_grad_comb(model, losses, os.path.join(GOLD, stem + "_grads_f64.npz"), seed, B, T, True)


def _save_grad_comb(model, losses, path, seed, B, T, double):
    gnames, vals, offs, gmax = [], [], [0], []
    for n, p in model.named_parameters():
        g = p.grad.detach().reshape(-1).numpy()
        idx = grad_sample_index(g.size)
        gnames.append(n)
        vals.append(g[idx].astype(np.float64 if double else np.float32))
        offs.append(offs[-1] + idx.size)
        gmax.append(float(np.abs(g).max()))
    np.savez(path, seed=seed, B=B, T=T, grad_names=np.array(gnames), grad_values=np.concatenate(vals),
             grad_offsets=np.array(offs), grad_absmax=np.array(gmax), n_samples=GRAD_SAMPLES, loss_total=float(losses["total"]))


if __name__ == "__main__":
    ref_shim.install()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    if sys.argv[1:2] == ["bs32"]:          # the benchmark-size golden (minutes, tens of GB): bs32 [smooth|perturbed] [f32|f64]
        gen_bs32(sys.argv[2] if len(sys.argv) > 2 else "smooth", only=sys.argv[3] if len(sys.argv) > 3 else None)
        sys.exit(0)
    if sys.argv[1:2] == ["emb256"]:        # BASELINE.json configs[4]'s model (2 x QRNN, lnorm; built through WaveFe's kwargs) at 32 x 32 000
        gen_bs32(sys.argv[2] if len(sys.argv) > 2 else "smooth", fe_over=dict(rnn_layers=2, norm_type="lnorm"),
                 stem="pase_plus_step_emb256_bs32_%s" % (sys.argv[2] if len(sys.argv) > 2 else "smooth"),
                 only=sys.argv[3] if len(sys.argv) > 3 else None)
        sys.exit(0)
    if sys.argv[1:2] == ["cfg2bs32"]:      # BASELINE.json configs[1] at its full size: PASE.cfg + workers.cfg, 32 x 16 000
        gen_bs32(sys.argv[2] if len(sys.argv) > 2 else "smooth", seed=4, B=32, T=16000, fe_name="PASE.cfg", wk_name="workers.cfg",
                 stem="pase_step_cfg2_bs32_%s" % (sys.argv[2] if len(sys.argv) > 2 else "smooth"),
                 only=sys.argv[3] if len(sys.argv) > 3 else None)
        sys.exit(0)
    if sys.argv[1:2] == ["perturbed"]:     # only the perturbed-slope steps (optionally: only files whose stem contains argv[2])
        gen_perturbed(sys.argv[2] if len(sys.argv) > 2 else None)
        sys.exit(0)
    if sys.argv[1:] == ["grads"]:          # only the element-wise gradient file
        gen_pase_step_grads(seed=2, B=2, T=8000)
        sys.exit(0)
    if sys.argv[1:] == ["grads64"]:        # only the fp64-truth gradient files
        gen_pase_step_grads(seed=2, B=2, T=8000, out="pase_plus_step_grads_f64.npz", double=True)
        gen_pase_step_grads(seed=4, B=2, T=8000, fe_name="PASE.cfg", wk_name="workers.cfg",
                            out="pase_step_cfg2_grads_f64.npz", double=True)
        gen_pase_step_grads(seed=4, B=2, T=8000, fe_name="PASE.cfg", wk_name="workers.cfg",
                            out="pase_step_cfg2_grads.npz")
        sys.exit(0)
    gen_sinc()
    gen_wavefe("pase_plus", "PASE+.cfg", seed=2, S=3, T=8000)
    gen_wavefe("pase", "PASE.cfg", seed=3, S=3, T=8000)
    gen_pase_step(seed=2, B=2, T=8000)
    gen_pase_step(seed=4, B=2, T=8000, fe_name="PASE.cfg", wk_name="workers.cfg", out="pase_step_cfg2.npz")
    gen_pase_step_grads(seed=2, B=2, T=8000)
    gen_pase_step_grads(seed=2, B=2, T=8000, out="pase_plus_step_grads_f64.npz", double=True)
    gen_pase_step_grads(seed=4, B=2, T=8000, fe_name="PASE.cfg", wk_name="workers.cfg",
                        out="pase_step_cfg2_grads_f64.npz", double=True)
    gen_pase_step_grads(seed=4, B=2, T=8000, fe_name="PASE.cfg", wk_name="workers.cfg", out="pase_step_cfg2_grads.npz")
    gen_perturbed()
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))

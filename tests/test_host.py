"""Host-side / boundary tests that need no kernel execution (`not gpu`): the C-ABI library exports
every symbol include/pase_amd.h declares, struct layouts match, the reference's Python surface
(wf_builder / WaveFe / load_pretrained / worker_parser / Saver / LR schedule) behaves like the
reference's."""
import ctypes
import json
import os
import re

import pytest
import torch

from util import ROOT, load_cfg, quiet


def test_header_symbols_exported_by_hip_library():
    from pase_amd import build, kernels
    so = build.build_hip()          # hipcc cross-compile; no GPU needed
    lib = ctypes.CDLL(so)
    hdr = open(os.path.join(ROOT, "include", "pase_amd.h")).read()
    names = set(re.findall(r"\b(pase_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(lib, n), "libpase_hip.so does not export %s" % n
    kernels.declare(lib)            # ABI struct sizes verified inside
    code = open(so, "rb").read()
    assert b"gfx950" in code


def test_no_cpu_fallback():
    from pase_amd import _lib, kernels
    _lib.use_library(None, "cuda")
    with pytest.raises(Exception):
        kernels._ptr(torch.zeros(4))      # CPU tensor with the product library -> loud failure


def test_product_never_imports_oracle():
    for dirpath, _d, files in os.walk(os.path.join(ROOT, "pase_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f


def test_wf_builder_surface():
    from pase_amd.frontend import WaveFe, wf_builder
    with pytest.raises(ValueError):
        wf_builder(None)
    with pytest.raises(TypeError):
        wf_builder({"name": "nonsense"})
    fe = quiet(wf_builder, os.path.join(ROOT, "cfg", "frontend", "PASE+.cfg"))
    assert isinstance(fe, WaveFe) and fe.emb_dim == 256
    keys = list(fe.state_dict().keys())
    assert keys[0] == "denseskips.0.weight" and keys[-1] == "norm_out.num_batches_tracked"
    sd = fe.state_dict()
    assert tuple(sd["blocks.0.conv.low_hz_"].shape) == (64, 1)
    assert tuple(sd["blocks.7.conv.weight"].shape) == (512, 512, 11)
    assert tuple(sd["rnn.layers.0.linear.weight"].shape) == (1536, 1024)
    assert tuple(sd["W.weight"].shape) == (256, 512, 1)
    assert sum(p.numel() for p in fe.parameters()) == 7832896         # SURVEY.md section 8a [probe]
    fe2 = quiet(wf_builder, load_cfg("frontend/PASE.cfg"))
    assert fe2.emb_dim == 100 and sum(p.numel() for p in fe2.parameters()) == 5818020


def test_load_pretrained_semantics(tmp_path):
    from pase_amd.frontend import wf_builder
    cfg = dict(kwidths=[31, 20, 11], strides=[1, 10, 2], fmaps=[4, 4, 6], emb_dim=5, norm_out=True)
    a = quiet(wf_builder, dict(cfg))
    b = quiet(wf_builder, dict(cfg))
    ck = str(tmp_path / "FE_e0.ckpt")
    torch.save(a.state_dict(), ck)
    quiet(b.load_pretrained, ck, load_last=True, verbose=False)
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k])
    with pytest.raises(ValueError):       # load_last=False drops the last two keys -> count mismatch
        quiet(b.load_pretrained, ck, load_last=False, verbose=False)
    torch.save({"state_dict": a.state_dict(), "step": 3}, ck)
    quiet(b.load_pretrained, ck, load_last=True, verbose=False)


def test_worker_parser_and_model_params():
    from pase_amd.losses import ContextualizedLoss
    from pase_amd.pase import pase
    from pase_amd.utils import strip_transforms, worker_parser
    cfg = strip_transforms(worker_parser(os.path.join(ROOT, "cfg", "workers", "workers+.cfg")))
    assert [w["name"] for w in cfg["regr"]][:3] == ["cchunk", "lps", "lps_long"]
    assert all(isinstance(w["loss"], ContextualizedLoss) for g in cfg.values() for w in g)
    assert cfg["regr"][1]["loss"].r == 7 and cfg["regr"][0]["loss"].r is None
    m = quiet(pase, frontend_cfg=load_cfg("frontend/PASE+.cfg"), minions_cfg=cfg, cls_lst=["mi", "cmi"],
              regr_lst=[w["name"] for w in load_cfg("workers/workers+.cfg")["regr"]])
    n_workers = sum(p.numel() for w in list(m.regression_workers) + list(m.classification_workers)
                    for p in w.parameters())
    assert n_workers == 21842710
    assert sum(p.numel() for p in m.parameters()) == 29675606


def test_saver_and_lr_scheduler(tmp_path):
    from pase_amd.frontend import wf_builder
    from pase_amd.modules import Saver
    from pase_amd.trainer import LR_Scheduler
    fe = quiet(wf_builder, dict(kwidths=[31, 20], strides=[1, 10], fmaps=[4, 4], emb_dim=5))
    sv = Saver(fe, str(tmp_path), max_ckpts=2, prefix="PASE-")
    for step in (10, 20, 30, 40):
        sv.save("PASE", step)
    idx = json.load(open(os.path.join(str(tmp_path), "PASE-checkpoints")))
    assert idx["current"] == "PASE-PASE-40.ckpt"
    assert sv.read_latest_checkpoint() == "PASE-PASE-40.ckpt"
    assert sv.load_ckpt_step("PASE-PASE-40.ckpt") == 40
    assert sv.load_weights()

    class Opt:
        param_groups = [{"lr": 0.0}]
    sch = LR_Scheduler("poly", "frontend", 1e-3, num_epochs=4, iters_per_epoch=100)
    lr = sch(Opt, 50, 1, 0.0)
    assert abs(lr - 1e-3 * (1 - 150 / 400) ** 0.9) < 1e-15 and Opt.param_groups[0]["lr"] == lr


def test_ddp_frontend_buckets_partition_the_gradient_buffer():
    """trainer._frontend_buckets: the reverse-order all-reduce buckets cover every element of the frontend gradient
    buffer exactly once, the head bucket holds W / dense skips / QRNN, and block 0's bucket carries the merged small
    blocks (no GPU, no process group needed)."""
    import contextlib
    import io
    from pase_amd import _lib, build
    from pase_amd.trainer import trainer
    from util import MINI_FE, mini_workers, with_losses
    _lib.use_library(build.build_emu(), "cpu")
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            tr = trainer(frontend_cfg=dict(MINI_FE), minions_cfg=with_losses(mini_workers()), cfg=dict(epoch=1, bpe=2))
        bk = tr._frontend_buckets()
        n = tr.frontend_optim.flat_g.numel()
        cover = [0] * n
        for tag, rs in bk.items():
            for b, e in rs:
                for i in range(b, e):
                    cover[i] += 1
        assert min(cover) == 1 and max(cover) == 1
        assert set(bk) == {"head"} | set(range(len(tr.model.frontend.blocks)))
        names = {id(p): k for k, p in tr.model.frontend.named_parameters()}
        head_elems = sum(e - b for b, e in bk["head"])
        want = sum(p.numel() for p in tr.frontend_optim.params if not names[id(p)].startswith("blocks."))
        assert head_elems == want
    finally:
        _lib.use_library(None, "cuda")

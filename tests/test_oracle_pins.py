"""Pin the CPU oracle (oracle/pase_oracle.py) against (a) the golden vectors generated from the LIVE
reference by oracle/make_golden.py and (b), when /root/reference is present, the live reference
itself on fresh random cases.  Also pins the weight-initialisation parity of the pase_amd mirrors
(same seed -> same initial weights as the reference).  CPU only (`not gpu`)."""
import os

import numpy as np
import pytest
import torch

from oracle import pase_oracle as O
from util import GOLD, assert_close, is_noise_grad, load_cfg, oracle_params, quiet, seed_all, synthetic_batch, with_losses


def _npz(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def test_sinc_known_answers():
    """SURVEY.md section 8c known-answer pins of the deterministic SincConv_fast initialisation."""
    low, band = O.sinc_init()
    assert_close(low[:3, 0], torch.tensor([30.0, 58.68235, 88.49165]), rtol=1e-6, atol=1e-4)
    assert_close(band[:3, 0], torch.tensor([28.68235, 29.80930, 30.98054]), rtol=1e-6, atol=1e-4)
    assert abs(float(low[-1]) - 7574.873) < 1e-2 and abs(float(band[-1]) - 325.127) < 1e-2
    f = O.sinc_filters(low, band)
    assert f.shape == (64, 1, 251)
    assert torch.equal(f, torch.flip(f, dims=[2]))
    assert float(f[0, 0, 125]) == 1.0
    assert abs(float(f[0, 0, 124]) - 0.99871814) < 1e-6
    assert abs(float(f[63, 0, 0]) - 0.0018330044) < 1e-7
    assert abs(float(f.sum()) - 6.6129384) < 1e-3


def test_sinc_golden():
    g = _npz("sinc_init.npz")
    low, band = O.sinc_init()
    assert_close(low, g["low_hz_"], rtol=0, atol=0)
    assert_close(band, g["band_hz_"], rtol=0, atol=0)
    assert_close(O.sinc_filters(low, band), g["filters"], rtol=1e-6, atol=1e-7)
    w, n_ = O.sinc_constants()
    assert_close(w, g["window_"], rtol=0, atol=0)
    assert_close(n_, g["n_"], rtol=0, atol=0)
    p = _npz("sinc_perturbed.npz")
    low = torch.tensor(p["low_hz_"], requires_grad=True)
    band = torch.tensor(p["band_hz_"], requires_grad=True)
    f = O.sinc_filters(low, band)
    assert_close(f, p["filters"], rtol=1e-6, atol=1e-7)
    y = torch.nn.functional.conv1d(torch.nn.functional.pad(torch.tensor(p["x"]), (125, 125), mode="reflect"), f)
    assert_close(y, p["y"], rtol=1e-5, atol=1e-5)
    (y * torch.tensor(p["g"])).sum().backward()
    assert_close(low.grad, p["dlow"], rtol=1e-4, atol=1e-6)
    assert_close(band.grad, p["dband"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag,cfgfile", [("pase_plus", "frontend/PASE+.cfg"), ("pase", "frontend/PASE.cfg")])
def test_encoder_golden(tag, cfgfile):
    """oracle WaveFe forward/backward == live reference on the seeded golden case; the pase_amd
    mirror initialises to the same weights under the same seed."""
    from pase_amd.frontend import wf_builder
    g = _npz("wavefe_%s.npz" % tag)
    cfg = load_cfg(cfgfile)
    seed_all(int(g["seed"]))
    fe = quiet(wf_builder, dict(cfg))
    sd = fe.state_dict()
    assert list(sd.keys()) == [str(s) for s in g["param_names"]]
    assert_close(torch.tensor([float(v.double().sum()) for v in sd.values()]), g["param_sum"], rtol=1e-7, atol=1e-6)
    assert_close(torch.tensor([float((v.double() ** 2).sum()) for v in sd.values()]), g["param_sq"], rtol=1e-7,
                 atol=1e-6)
    P = oracle_params(fe)
    x = torch.tensor(g["x"])
    so = {}
    y = O.encoder_forward(P, cfg, x, True, so)
    assert_close(y, g["y_train"], rtol=1e-4, atol=1e-4, what="train fwd")
    (y * torch.tensor(g["g"])).sum().backward()
    names = [str(s) for s in g["grad_names"]]
    keep = [i for i, n in enumerate(names) if not is_noise_grad(n)]
    gsq = torch.tensor([float((P[names[i]].grad.double() ** 2).sum()) for i in keep])
    assert_close(gsq.sqrt(), np.sqrt(g["grad_sq"][keep]), rtol=2e-3, atol=1e-5, what="grad norms")
    for k, v in so.items():
        P[k] = v
    with torch.no_grad():
        ye = O.encoder_forward(P, cfg, x, False)
    assert_close(ye, g["y_eval"], rtol=1e-4, atol=1e-4, what="eval fwd")
    assert_close(O.select_output(ye, "avg_norm"), g["y_avg_norm"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("gold,fe,wk", [("pase_plus_step.npz", "frontend/PASE+.cfg", "workers/workers+.cfg"),
                                        ("pase_step_cfg2.npz", "frontend/PASE.cfg", "workers/workers.cfg"),
                                        ("pase_plus_step_perturbed.npz", "frontend/PASE+.cfg", "workers/workers+.cfg"),
                                        ("pase_step_cfg2_perturbed.npz", "frontend/PASE.cfg", "workers/workers.cfg"),
                                        ("pase_plus_step_smooth.npz", "frontend/PASE+.cfg", "workers/workers+.cfg"),
                                        ("pase_step_cfg2_smooth.npz", "frontend/PASE.cfg", "workers/workers.cfg")],
                         ids=["plus", "cfg2", "plus-perturbed", "cfg2-perturbed", "plus-smooth", "cfg2-smooth"])
def test_pase_step_golden(gold, fe, wk):
    """oracle full step (all workers, losses, grads) == live reference trainer step, for PASE+.cfg +
    workers+.cfg (BASELINE configs[2]) and PASE.cfg + workers.cfg incl. the SPC worker (configs[1])."""
    import random
    from pase_amd.pase import pase
    g = _npz(gold)
    fe_cfg = load_cfg(fe)
    raw = load_cfg(wk)
    seed_all(int(g["seed"]))
    model = quiet(pase, frontend_cfg=dict(fe_cfg), minions_cfg=with_losses(load_cfg(wk)),
                  cls_lst=[w["name"] for w in raw["cls"]], regr_lst=[w["name"] for w in raw["regr"]])
    if "perturbed" in gold or "smooth" in gold:      # BN affines / PReLU slopes off init, the draw of make_golden.perturb_affine
        from util import randomize_affine
        randomize_affine(model, smooth="smooth" in gold)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(s) for s in g["param_names"]]
    assert_close(torch.tensor([float((v.double() ** 2).sum()) for v in sd.values()]), g["param_sq"], rtol=1e-7,
                 atol=1e-6, what="init parity")
    P = oracle_params(model)
    B, T = int(g["B"]), int(g["T"])
    batch = synthetic_batch(int(g["seed"]) + 1, B, T, raw["regr"])
    random.seed(int(g["seed"]) + 2)
    h, chunk, preds, labels = O.pase_forward(P, fe_cfg, raw, batch, True)
    assert_close(chunk, g["chunk_emb"], rtol=1e-4, atol=1e-4, what="chunk embedding")
    assert_close(preds["mi"], g["pred_mi"], rtol=1e-4, atol=1e-4)
    assert_close(preds["cmi"], g["pred_cmi"], rtol=1e-4, atol=1e-4)
    assert_close(preds["mfcc"], g["pred_mfcc"], rtol=1e-4, atol=1e-4)
    if "pred_spc" in g.files:
        assert_close(preds["spc"], g["pred_spc"], rtol=1e-4, atol=1e-4)
    assert_close(preds["cchunk"][:, :, :400], g["pred_cchunk_head"], rtol=1e-4, atol=1e-4)
    losses = O.pase_losses(raw, preds, labels)
    gl = dict(zip([str(s) for s in g["loss_names"]], g["loss_values"]))
    for k, v in gl.items():
        assert abs(float(losses[k]) - v) <= 1e-4 * max(1.0, abs(v)), (k, float(losses[k]), v)
    losses["total"].backward()
    names = [str(s) for s in g["grad_names"]]
    keep = [i for i, n in enumerate(names) if not is_noise_grad(n)]
    gsq = torch.tensor([float((P[names[i]].grad.double() ** 2).sum()) for i in keep])
    assert_close(gsq.sqrt(), np.sqrt(g["grad_sq"][keep]), rtol=5e-3, atol=1e-6, what="grad norms")


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="live reference tree not present")
@pytest.mark.parametrize("norm_type", ["bnorm", "lnorm", "inorm", "affinorm"])
def test_oracle_vs_live_reference_random_cfg(norm_type):
    """incl. the norm_type variants of build_norm_layer / forward_norm (modules.py:77-109) and the InstanceNorm
    norm_out they imply (frontend.py:206-210): BASELINE.json configs[4] is an 'lnorm'-style 2xQRNN variant."""
    from oracle import ref_shim
    ref_shim.install()
    from pase.models.frontend import wf_builder as ref_builder
    cfg = dict(kwidths=[51, 20, 11, 11, 11, 11, 11, 11], strides=[1, 10, 2, 1, 2, 1, 2, 2],
               fmaps=[8, 8, 12, 12, 16, 16, 20, 20], emb_dim=24, rnn_dim=20, denseskips=True, norm_out=True,
               rnn_pool=True, rnn_layers=2, norm_type=norm_type)
    seed_all(11)
    ref = quiet(ref_builder, dict(cfg))
    P = oracle_params(ref)
    x = torch.randn(4, 1, 3200) * 0.2
    ref.train()
    yr = ref(x)
    yo = O.encoder_forward(P, cfg, x, True)
    assert_close(yo, yr, rtol=1e-5, atol=1e-5)
    gsel = torch.randn_like(yr)
    (yr * gsel).sum().backward()
    (yo * gsel).sum().backward()
    for n, p in ref.named_parameters():
        if not is_noise_grad(n) or norm_type == "lnorm":      # LayerNorm does not cancel the conv bias
            assert_close(P[n].grad, p.grad, rtol=1e-3, atol=1e-4, what=n)


def test_dsp_oracle_stft_matches_torch_stft():
    """LPS calls torch.stft(wav, n_fft, hop, win) (transforms.py:465): the oracle's rectangular-window
    centred STFT is pinned against this image's torch.stft with the same positional arguments."""
    from oracle import dsp_oracle as O
    g = torch.Generator().manual_seed(0)
    wav = torch.randn(4000, generator=g) * 0.1
    for n_fft, hop, win in ((2048, 160, 400), (2048, 160, 512), (256, 160, 100)):
        ref = torch.stft(wav, n_fft, hop, win, return_complex=True).numpy()
        got = O.stft_rect(wav.numpy(), n_fft, hop, win)
        np.testing.assert_allclose(got, ref, atol=2e-4)
        lps_ref = 10 * torch.log10(torch.view_as_real(torch.stft(wav, n_fft, hop, win, return_complex=True))
                                   .norm(2, dim=2)[:, :len(wav) // hop] ** 2 + 10e-20).numpy()
        np.testing.assert_allclose(O.lps(wav.numpy(), n_fft, hop, win, der_order=0), lps_ref, atol=2e-3)


def test_dsp_oracle_mel_banks_are_well_formed():
    """Unpinned third-party restatements (librosa / python_speech_features are not installed): check the
    published invariants -- Slaney bank rows integrate to ~2/bandwidth * triangle area (area-normalised),
    HTK bank rows peak at 1 -- and that host-side bases equal the oracle's."""
    from oracle import dsp_oracle as O
    from pase_amd import dsp
    fb = O.psf_get_filterbanks(40, 512, 16000)
    assert fb.shape == (40, 257) and abs(fb.max() - 1.0) < 1e-12 and (fb >= 0).all()
    np.testing.assert_allclose(dsp.psf_mel_filterbank(40, 512, 16000), fb, atol=1e-6)
    mel = O.librosa_mel(16000, 400)
    assert mel.shape == (128, 201) and (mel >= 0).all()
    np.testing.assert_allclose(dsp.slaney_mel_filterbank(16000, 400), mel, atol=1e-7)
    import scipy.fftpack
    x = np.random.default_rng(1).normal(size=(128, 5))
    np.testing.assert_allclose(dsp.dct2_ortho(13, 128).astype(np.float64) @ x,
                               scipy.fftpack.dct(x, axis=0, type=2, norm="ortho")[:13], atol=1e-5)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="live reference tree not present")
def test_checkpoint_wire_format_round_trip_with_live_reference(tmp_path):
    """SURVEY (f)-4: a checkpoint written by pase_amd's Saver loads into the REFERENCE WaveFe through the
    reference's own load_pretrained (modules.py:267-301), and a reference checkpoint loads into pase_amd --
    same keys, same shapes, same tensors (PASE+-shaped encoder incl. QRNN and dense skips)."""
    from oracle import ref_shim
    ref_shim.install()
    from pase.models.frontend import wf_builder as ref_builder
    import pase_amd.frontend as mine
    from pase_amd.modules import Saver
    cfg = dict(kwidths=[51, 20, 11, 11, 11, 11, 11, 11], strides=[1, 10, 2, 1, 2, 1, 2, 2],
               fmaps=[8, 8, 12, 12, 16, 16, 20, 20], emb_dim=24, rnn_dim=20, denseskips=True, norm_out=True,
               rnn_pool=True, rnn_layers=1)
    seed_all(5)
    a = quiet(mine.wf_builder, dict(cfg))
    seed_all(6)
    ref = quiet(ref_builder, dict(cfg))
    assert list(a.state_dict().keys()) == list(ref.state_dict().keys())
    # pase_amd -> reference: Saver file ('state_dict' wrapper, trainer.py:267-272 convention)
    sv = Saver(a, str(tmp_path), max_ckpts=2, prefix="PASE-")
    quiet(sv.save, "PASE", 3)
    ck = [f for f in os.listdir(tmp_path) if f.endswith(".ckpt") and "weights" in f]
    assert ck, os.listdir(tmp_path)
    quiet(ref.load_pretrained, os.path.join(str(tmp_path), ck[0]), load_last=True, verbose=False)
    for k, v in a.state_dict().items():
        assert torch.equal(ref.state_dict()[k], v), k
    # reference -> pase_amd: bare state_dict file
    seed_all(7)
    ref2 = quiet(ref_builder, dict(cfg))
    p2 = os.path.join(str(tmp_path), "ref.ckpt")
    torch.save(ref2.state_dict(), p2)
    quiet(a.load_pretrained, p2, load_last=True, verbose=False)
    for k, v in ref2.state_dict().items():
        assert torch.equal(a.state_dict()[k], v), k
    # and the two then compute the same thing on CPU through the oracle parameters
    x = torch.randn(2, 1, 1600) * 0.2
    ref2.eval()
    with torch.no_grad():
        yr = ref2(x)
    yo = O.encoder_forward(oracle_params(a), cfg, x, False)
    assert_close(yo, yr, rtol=1e-5, atol=1e-5)

"""Pins of the transform / dataset / Gap oracles against the LIVE reference classes (SURVEY.md 8c, rows a16, a19,
a20, a25, a26, a27): pase/transforms.py {MIChunkWav, SingleChunkWav, Reverb, SimpleAdditive, SimpleAdditiveShift,
Clipping, BandDrop, Downsample, PCompose via config_distortions, LPS, ZNorm}, pase/dataset.py DictCollater and the
Gap worker run here through oracle/ref_shim.install_transforms() (oracle/live_transforms.py); the numpy / scipy
restatements the device kernels are tested against must reproduce them from the same random draws.  Skipped when
/root/reference is absent (GPU box): there the committed golden tests/golden/transforms_live.npz stands in."""
import os

import numpy as np
import pytest
import torch

from oracle import dsp_oracle as D
from oracle import producer_oracle as O
from oracle import ref_shim

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transforms_live.npz")
needs_ref = pytest.mark.skipif(not os.path.isdir(ref_shim.REFERENCE_ROOT), reason="live reference not present")


@pytest.fixture(scope="module")
def live():
    from oracle import live_transforms
    return live_transforms.run()


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@needs_ref
def test_committed_golden_is_what_the_live_reference_produces(live, gold):
    assert set(gold.files) == set(live.keys())
    for k in gold.files:
        a, b = gold[k], np.asarray(live[k])
        if a.dtype.kind in "US":
            assert (a == b).all(), k
        else:
            np.testing.assert_allclose(b, a, rtol=1e-6, atol=1e-7, err_msg=k)


def _check_chunkers(g):
    T = g["mi0_chunk"].shape[0]
    for ci, (a, b) in enumerate(g["chunk_pairs"]):
        for j, (k, src) in enumerate((("chunk", a), ("chunk_ctxt", a), ("chunk_rand", b))):
            want = g["mi%d_%s" % (ci, k)]
            got = O.norm_and_scale(O.select_chunk(g["wav%d" % src], T, int(g["mi%d_beg" % ci][j])), g["mi%d_scale" % ci][j])
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7, err_msg="MIChunkWav %d %s" % (ci, k))


def _irs(g, max_len):
    return [O.prepare_ir(g["irs%d" % i], max_len) for i in range(2)]


def _check_distortions(g):
    clean = g["clean"]
    irs = _irs(g, int(g["reverb_max_len"]))
    for ci in range(2):
        ir, pm = irs[int(g["reverb%d_ir" % ci])]
        np.testing.assert_allclose(O.reverb(clean, ir, pm), g["reverb%d" % ci], rtol=1e-5, atol=1e-6, err_msg="Reverb")
    for ci in range(3):
        nidx, nbeg, snr = g["additive%d_draw" % ci]
        got = O.additive(clean, g["noises%d" % int(nidx)], int(nbeg), float(snr))
        np.testing.assert_allclose(got, g["additive%d" % ci], rtol=1e-5, atol=1e-6, err_msg="SimpleAdditive %d" % ci)
    for ci in range(3):
        si, sbeg, shift, iri, snr = g["overlap%d_draw" % ci]
        ir, pm = irs[int(iri)]
        got = O.overlap(clean, g["speech%d" % int(si)], int(sbeg), int(shift), float(snr), ir, pm)
        np.testing.assert_allclose(got, g["overlap%d" % ci], rtol=1e-5, atol=1e-6, err_msg="SimpleAdditiveShift %d" % ci)
        np.testing.assert_allclose(O.overlap_label(len(clean), int(shift), 160), g["overlap%d_label" % ci], atol=1e-7)
    np.testing.assert_array_equal(O.clipping(clean, float(g["clipping_cf"])), g["clipping"])
    np.testing.assert_allclose(O.fir_filter_distortion(clean, g["bandrop0"]), g["bandrop"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.fir_filter_distortion(clean, g["downsample0"]), g["downsample"], rtol=1e-5, atol=1e-6)


def oracle_chain(g, ci):
    """config_distortions order (transforms.py:83-141): Reverb, SimpleAdditiveShift, SimpleAdditive, Clipping,
    BandDrop, Downsample, each behind its PCompose gate (a draw of -1 / 0 = gated off)."""
    x = g["clean"].copy()
    rv, osrc, obeg, oshift, osnr, aidx, abeg, asnr, clip, bd, ds = g["chain%d_draw" % ci]
    irs = _irs(g, 24000)           # config_distortions builds Reverb with the default max_reverb_len
    label = np.zeros(len(x) // 160, dtype=np.float32)
    if rv >= 0:
        x = O.reverb(x, *irs[int(rv)])
    if osrc >= 0:
        x = O.overlap(x, g["speech%d" % int(osrc)], int(obeg), int(oshift), float(osnr))
        label = O.overlap_label(len(x), int(oshift), 160)
    if aidx >= 0:
        x = O.additive(x, g["noises%d" % int(aidx)], int(abeg), float(asnr))
    if clip > 0:
        x = O.clipping(x, float(clip))
    if bd >= 0:
        x = O.fir_filter_distortion(x, g["bandrop0"])
    if ds >= 0:
        x = O.fir_filter_distortion(x, g["downsample0"])
    return x, label


def _check_chain(g):
    assert list(g["chain_order"]) == ["Reverb", "SimpleAdditiveShift", "SimpleAdditive", "Clipping", "BandDrop",
                                      "Downsample"]
    n_on = 0
    for ci in range(4):
        x, label = oracle_chain(g, ci)
        np.testing.assert_allclose(x, g["chain%d" % ci], rtol=2e-5, atol=2e-6, err_msg="chain %d" % ci)
        np.testing.assert_allclose(label, g["chain%d_label" % ci], atol=1e-7)
        d = g["chain%d_draw" % ci]
        n_on += int(d[0] >= 0) + int(d[1] >= 0) + int(d[5] >= 0) + int(d[8] > 0) + int(d[9] >= 0) + int(d[10] >= 0)
    assert n_on >= 10          # the seeds exercise most gates


def _check_lps_znorm(g):
    clean = g["clean"]
    for nm, win in (("lps", 400), ("lps_long", 512)):
        got = D.lps(clean, n_fft=2048, hop=160, win=win, der_order=2)
        want = g[nm]
        assert got.shape == want.shape == (3075, len(clean) // 160)
        # same fp32 torch.stft underneath; log10 of near-zero bins amplifies the last ulp
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-3, err_msg=nm)
    np.testing.assert_allclose(D.znorm(g["lps"], g["znorm_mean"], g["znorm_std"]), g["lps_znorm"], rtol=1e-6, atol=1e-6)


def _check_gap(g):
    from oracle import pase_oracle as PO
    x = torch.from_numpy(g["gap_x"])
    np.random.seed(701)
    xin, lab = PO.gap_samples(x)
    P = {"m." + str(n): torch.from_numpy(g["gap_p_" + str(n)]) for n in g["gap_param_names"]}
    y = PO.mlp_minion(P, "m.minion.", xin, 1)
    np.testing.assert_allclose(y.numpy(), g["gap_y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(lab.numpy(), g["gap_label"])
    assert g["gap_label"].max() == 1.0 and g["gap_label"].min() == 0.0     # the LongTensor truncation is exercised


def _check_collate(g):
    """DictCollater layout (dataset.py:21-89): 1-D -> (B,1,T), 2-D -> (B,D,F); non-batching keys dropped."""
    keys = set(str(k) for k in g["collate_keys"])
    assert keys == {"chunk", "chunk_ctxt", "chunk_rand", "cchunk", "overlap", "lps"}
    B = 3
    for k in ("chunk", "chunk_ctxt", "chunk_rand", "cchunk"):
        assert g["collate_" + k].shape == (B, 1, 1600)
    assert g["collate_overlap"].shape == (B, 1, 10) and g["collate_lps"].shape == (B, 3075, 10)
    for ci in range(B):
        np.testing.assert_array_equal(g["collate_chunk"][ci, 0], g["mi%d_chunk" % ci])
        np.testing.assert_array_equal(g["collate_chunk_rand"][ci, 0], g["mi%d_chunk_rand" % ci])


@pytest.mark.parametrize("which", ["golden", "live"])
def test_oracles_reproduce_the_reference(which, gold, request):
    """Runs against the committed golden everywhere, and against a fresh live run when the reference is present."""
    if which == "live":
        if not os.path.isdir(ref_shim.REFERENCE_ROOT):
            pytest.skip("live reference not present")
        g = request.getfixturevalue("live")
    else:
        g = gold
    _check_chunkers(g)
    _check_distortions(g)
    _check_chain(g)
    _check_lps_znorm(g)
    _check_gap(g)
    _check_collate(g)

"""On-device batch producer (chunking, norm_and_scale, Reverb, SimpleAdditive) vs the numpy/scipy oracle."""
import numpy as np
import pytest
import torch

from oracle import producer_oracle as O
from pase_amd import producer as P


def _pool(rng, lens):
    return [(0.3 * rng.standard_normal(n)).astype(np.float32) for n in lens]


def test_chunker_matches_select_chunk(dev):
    rng = np.random.RandomState(0)
    wavs = _pool(rng, [5000, 1300, 900, 2400])        # 900 <= T: reflect-padded from 0
    T = 1200
    pool = P.WavPool(wavs, dev)
    ch = P.DeviceChunker(pool, T, random_scale=True, rng=np.random.RandomState(1))
    src, beg, scale = ch.draw(5)
    out = ch(src=src, beg=beg, scale=scale)
    for k, name in enumerate(("chunk", "chunk_ctxt", "chunk_rand")):
        assert out[name].shape == (5, 1, T)
        for b in range(5):
            want = O.norm_and_scale(O.select_chunk(wavs[src[k, b]], T, int(beg[k, b])), scale[k, b])
            np.testing.assert_allclose(out[name][b, 0].cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    assert (src[0] == src[1]).all() and (src[2] != src[0]).all()


@pytest.mark.parametrize("T,irlens", [(700, [90, 300, 1]), (2300, [1100, 40])])
def test_reverb_matches_scipy(dev, T, irlens):
    rng = np.random.RandomState(2)
    B = 4
    x = (0.2 * rng.standard_normal((B, 1, T))).astype(np.float32)
    irs = []
    for L in irlens:
        ir = rng.standard_normal(L) * np.exp(-np.arange(L) / max(L / 5.0, 1.0))
        ir[min(7, L - 1)] = 2.5                       # a clear direct-path peak at a non-zero delay
        irs.append(ir)
    rv = P.DeviceReverb(irs, max_reverb_len=1000, device=dev)
    idx = np.array([0, -1, len(irs) - 1, 1 % len(irs)])
    got = rv(torch.from_numpy(x.copy()).to(dev), idx).cpu().numpy()
    for b in range(B):
        if idx[b] < 0:
            np.testing.assert_array_equal(got[b, 0], x[b, 0])
            continue
        ir, pm = O.prepare_ir(irs[idx[b]], 1000)
        want = O.reverb(x[b, 0], ir, pm)
        np.testing.assert_allclose(got[b, 0], want, rtol=2e-4, atol=2e-5 * np.abs(want).max())


def test_filter_distortions_and_clipping(dev):
    rng = np.random.RandomState(9)
    B, T = 4, 1800
    x = (0.2 * rng.standard_normal((B, 1, T))).astype(np.float32)
    filts = [np.sinc(np.arange(-50, 51) / 2.0) * np.hamming(101), rng.standard_normal(64)]   # odd and even lengths
    fd = P.DeviceFilter(filts, device=dev)
    idx = np.array([0, 1, -1, 0])
    got = fd(torch.from_numpy(x.copy()).to(dev), idx).cpu().numpy()
    for b in range(B):
        want = x[b, 0] if idx[b] < 0 else O.fir_filter_distortion(x[b, 0], filts[idx[b]])
        np.testing.assert_allclose(got[b, 0], want, rtol=2e-4, atol=2e-5 * np.abs(want).max())
    cf = np.array([0.3, 0.0, 0.5, 0.1], dtype=np.float32)
    got = P.DeviceClipping()(torch.from_numpy(x.copy()).to(dev), cf).cpu().numpy()
    for b in range(B):
        want = x[b, 0] if cf[b] <= 0 else O.clipping(x[b, 0], cf[b])
        np.testing.assert_array_equal(got[b, 0], want)


def test_overlap_speech_with_reverberated_interferer(dev):
    rng = np.random.RandomState(10)
    B, T = 4, 1600
    x = (0.2 * rng.standard_normal((B, 1, T))).astype(np.float32)
    speech = _pool(rng, [5000, 700, 2600])
    irs = [np.r_[0.0, 0.0, 0.3, 1.0, 0.4 * rng.standard_normal(120) * np.exp(-np.arange(120) / 30.0)]]
    rv = P.DeviceReverb(irs, device=dev)
    ov = P.DeviceOverlap(P.WavPool(speech, dev), reverb=rv)
    src = np.array([0, 1, -1, 2])
    shift = np.array([300, 1000, 0, 0])
    beg = np.array([1234, 0, 0, 17])          # utterance 1: file (700) longer than T - shift (600): crop from `beg`
    beg[1] = 50
    snr = np.array([5.0, 7.5, 10.0, 10.0], dtype=np.float32)
    got, label = ov(torch.from_numpy(x.copy()).to(dev), src, beg, shift, snr, ir_idx=np.zeros(B, int))
    got = got.cpu().numpy()
    ir, pm = O.prepare_ir(irs[0])
    for b in range(B):
        want = x[b, 0] if src[b] < 0 else O.overlap(x[b, 0], speech[src[b]], int(beg[b]), int(shift[b]), float(snr[b]),
                                                    ir, pm)
        np.testing.assert_allclose(got[b, 0], want, rtol=2e-4, atol=2e-5 * np.abs(want).max())
    assert label.shape == (B, T // 160) and float(label[2].sum()) == 0.0 and float(label[3].min()) == 1.0


def test_additive_matches_reference_formula(dev):
    rng = np.random.RandomState(3)
    B, T = 5, 1500
    x = (0.2 * rng.standard_normal((B, 1, T))).astype(np.float32)
    noises = [0.05 * rng.standard_normal(4000), 0.1 * rng.standard_normal(900), np.zeros(2000)]
    ad = P.DeviceAdditive(noises, device=dev)
    idx = np.array([0, 1, 2, -1, 0])
    beg = np.array([100, 0, 10, 0, 2499])
    snr = np.array([0.0, 5.0, 10.0, 5.0, 10.0], dtype=np.float32)
    got = ad(torch.from_numpy(x.copy()).to(dev), idx, beg, snr).cpu().numpy()
    for b in range(B):
        want = x[b, 0] if idx[b] < 0 else O.additive(x[b, 0], noises[idx[b]], int(beg[b]), float(snr[b]))
        np.testing.assert_allclose(got[b, 0], want, rtol=1e-5, atol=1e-6)


def test_batch_producer_contract(dev):
    """dataset.__getitem__ + DictCollater layout: (B, 1, T) waveforms, cchunk = clean chunk, distortions on
    `chunk` only, energy preserved by both distortions."""
    rng = np.random.RandomState(4)
    pool = P.WavPool(_pool(rng, [4000, 5000, 3000]), dev)
    T = 1600
    prod = P.DeviceBatchProducer(
        P.DeviceChunker(pool, T, rng=np.random.RandomState(5)),
        reverb=P.DeviceReverb([np.r_[0.0, 1.0, 0.5 * rng.standard_normal(200) * np.exp(-np.arange(200) / 40.0)]],
                              device=dev), reverb_p=1.0,
        additive=P.DeviceAdditive([0.1 * rng.standard_normal(6000)], device=dev), additive_p=1.0,
        rng=np.random.RandomState(6))
    batch = prod(4)
    assert set(batch) == {"chunk", "chunk_ctxt", "chunk_rand", "cchunk"}
    for v in batch.values():
        assert v.shape == (4, 1, T) and v.dtype == torch.float32
    e_clean = (batch["cchunk"] ** 2).sum(-1)
    e_dist = (batch["chunk"] ** 2).sum(-1)
    assert not torch.allclose(batch["chunk"], batch["cchunk"])
    # additive renormalises to the (reverberated) input energy; reverb scales the FULL convolution to the clean
    # energy, so the trimmed chunk keeps most of it
    assert (e_dist <= e_clean * 1.001).all() and (e_dist >= 0.5 * e_clean).all()


@pytest.mark.gpu
def test_reverb_full_size_24000_taps():
    """The BASELINE configs[3] size: 32 000-sample chunks, impulse responses truncated at 24 000 taps."""
    rng = np.random.RandomState(8)
    T, L, B = 32000, 24000, 3
    x = (0.1 * rng.standard_normal((B, 1, T))).astype(np.float32)
    irs = [np.r_[np.zeros(123), 1.0, 0.2 * rng.standard_normal(L + 500) * np.exp(-np.arange(L + 500) / 4000.0)],
           np.r_[0.0, 0.0, 1.0, 0.5 * rng.standard_normal(3000) * np.exp(-np.arange(3000) / 600.0)]]
    rv = P.DeviceReverb(irs, device="cuda")
    idx = np.array([0, 1, 0])
    got = rv(torch.from_numpy(x.copy()).cuda(), idx).cpu().numpy()
    for b in range(B):
        ir, pm = O.prepare_ir(irs[idx[b]])
        want = O.reverb(x[b, 0], ir, pm)
        np.testing.assert_allclose(got[b, 0], want, rtol=1e-3, atol=1e-4 * np.abs(want).max())


# ------------------------------------------------------------------------------------------------------------
# device producers vs the LIVE reference's outputs (tests/golden/transforms_live.npz, written by
# oracle/live_transforms.py from pase/transforms.py classes run under recorded random draws)
# ------------------------------------------------------------------------------------------------------------
import os  # noqa: E402

_G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transforms_live.npz"))


def _close(got, want, what, rtol=2e-4):
    np.testing.assert_allclose(got, want, rtol=rtol, atol=2e-5 * max(np.abs(want).max(), 1e-3), err_msg=what)


def test_live_reference_chunker(dev):
    g = _G
    wavs = [g["wav%d" % i] for i in range(3)]
    T = g["mi0_chunk"].shape[0]
    pairs = g["chunk_pairs"]
    ch = P.DeviceChunker(P.WavPool(wavs, dev), T, random_scale=True)
    src = np.stack([pairs[:, 0], pairs[:, 0], pairs[:, 1]], 0)
    beg = np.stack([g["mi%d_beg" % ci] for ci in range(3)], 1)
    scale = np.stack([g["mi%d_scale" % ci] for ci in range(3)], 1)
    out = ch(src=src, beg=beg, scale=scale)
    for ci in range(3):
        for k in ("chunk", "chunk_ctxt", "chunk_rand"):
            _close(out[k][ci, 0].cpu().numpy(), g["mi%d_%s" % (ci, k)], "MIChunkWav %d %s" % (ci, k), rtol=1e-6)


def _clean(dev, n):
    return torch.from_numpy(np.repeat(_G["clean"][None, None, :], n, 0).copy()).to(dev)


def test_live_reference_distortions(dev):
    g = _G
    irs = [g["irs0"], g["irs1"]]
    rv = P.DeviceReverb(irs, max_reverb_len=int(g["reverb_max_len"]), device=dev)
    got = rv(_clean(dev, 2), np.array([int(g["reverb0_ir"]), int(g["reverb1_ir"])])).cpu().numpy()
    for ci in range(2):
        _close(got[ci, 0], g["reverb%d" % ci], "Reverb %d" % ci)
    ad = P.DeviceAdditive([g["noises0"], g["noises1"]], device=dev)
    dr = np.stack([g["additive%d_draw" % ci] for ci in range(3)], 0)
    got = ad(_clean(dev, 3), dr[:, 0].astype(int), dr[:, 1].astype(int), dr[:, 2].astype(np.float32)).cpu().numpy()
    for ci in range(3):
        _close(got[ci, 0], g["additive%d" % ci], "SimpleAdditive %d" % ci)
    ov = P.DeviceOverlap(P.WavPool([g["speech0"], g["speech1"]], dev), reverb=rv)
    dr = np.stack([g["overlap%d_draw" % ci] for ci in range(3)], 0)
    got, lab = ov(_clean(dev, 3), dr[:, 0].astype(int), dr[:, 1].astype(int), dr[:, 2].astype(int),
                  dr[:, 4].astype(np.float32), ir_idx=dr[:, 3].astype(int))
    for ci in range(3):
        _close(got[ci, 0].cpu().numpy(), g["overlap%d" % ci], "SimpleAdditiveShift %d" % ci)
        np.testing.assert_allclose(lab[ci].cpu().numpy(), g["overlap%d_label" % ci], atol=1e-6)
    got = P.DeviceClipping()(_clean(dev, 1), np.array([float(g["clipping_cf"])])).cpu().numpy()
    np.testing.assert_array_equal(got[0, 0], g["clipping"])
    got = P.DeviceFilter([g["bandrop0"]], device=dev)(_clean(dev, 1), np.array([0])).cpu().numpy()
    _close(got[0, 0], g["bandrop"], "BandDrop")
    got = P.DeviceFilter([g["downsample0"]], device=dev)(_clean(dev, 1), np.array([0])).cpu().numpy()
    _close(got[0, 0], g["downsample"], "Downsample")


def test_live_reference_full_distortion_chain(dev):
    """config_distortions (transforms.py:38-146) -> PCompose gating -> six distortions in the reference's order, four
    utterances with the reference's own random decisions: DeviceBatchProducer.apply_chain on one batch."""
    g = _G
    B = 4
    prod = P.DeviceBatchProducer(
        None, reverb=P.DeviceReverb([g["irs0"], g["irs1"]], device=dev),
        overlap=P.DeviceOverlap(P.WavPool([g["speech0"], g["speech1"]], dev)),
        additive=P.DeviceAdditive([g["noises0"], g["noises1"]], device=dev), clipping=P.DeviceClipping(),
        bandrop=P.DeviceFilter([g["bandrop0"]], device=dev), downsample=P.DeviceFilter([g["downsample0"]], device=dev))
    dr = np.stack([g["chain%d_draw" % ci] for ci in range(B)], 0)
    d = dict(reverb_ir=dr[:, 0].astype(int), ov_src=dr[:, 1].astype(int), ov_beg=dr[:, 2].astype(int),
             ov_shift=dr[:, 3].astype(int), ov_snr=dr[:, 4].astype(np.float32), add_idx=dr[:, 5].astype(int),
             add_beg=dr[:, 6].astype(int), add_snr=dr[:, 7].astype(np.float32), clip=dr[:, 8].astype(np.float32),
             bandrop=dr[:, 9].astype(int), downsample=dr[:, 10].astype(int))
    batch = prod.apply_chain({"chunk": _clean(dev, B)}, d)
    for ci in range(B):
        _close(batch["chunk"][ci, 0].cpu().numpy(), g["chain%d" % ci], "chain %d" % ci, rtol=5e-4)
        np.testing.assert_allclose(batch["overlap"][ci, 0].cpu().numpy(), g["chain%d_label" % ci], atol=1e-6)


def test_live_reference_dictcollater_layout(dev):
    """The producer's batch dict has DictCollater's layout (dataset.py:21-89) for the keys it emits."""
    g = _G
    rng = np.random.RandomState(4)
    pool = P.WavPool([g["wav0"], g["wav1"], g["wav2"]], dev)
    prod = P.DeviceBatchProducer(P.DeviceChunker(pool, 1600, rng=rng), overlap=P.DeviceOverlap(P.WavPool([g["speech0"]], dev)),
                                 overlap_p=1.0, rng=rng)
    batch = prod(3)
    for k in ("chunk", "chunk_ctxt", "chunk_rand", "cchunk"):
        assert tuple(batch[k].shape) == g["collate_" + k].shape
    assert tuple(batch["overlap"].shape) == g["collate_overlap"].shape          # (B, 1, F), as DictCollater collates it


@pytest.mark.gpu
def test_pinned_batch_feeder_delivers_every_batch_intact():
    """Host -> HBM leg (the asynchronous form of the reference's `.to(device)`, modules.py:16-31 / pase.py:338): 12
    distinct host batches (pinned and pageable sources mixed) through the double-buffered copy stream, each consumed by
    a kernel on the compute stream while the next copy is in flight; every delivered batch must equal its source."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    src = []
    for i in range(12):
        b = {"chunk": torch.randn(8, 1, 32000, generator=g), "lps": torch.randn(8, 3075, 200, generator=g)}
        if i % 2 == 0:
            b = {k: v.pin_memory() for k, v in b.items()}
        src.append(b)
    it = iter(src)
    feeder = P.PinnedBatchFeeder(lambda: next(it, src[-1]), dev, depth=2)
    sums = []
    for i in range(11):
        d = feeder.next()
        # some work on the compute stream that reads the slot while the following copy runs
        sums.append((d["chunk"].double().sum() + d["lps"].double().sum()).clone())
        y = d["lps"] @ d["lps"].transpose(1, 2)          # keep the compute stream busy
        del y
    torch.cuda.synchronize()
    for i, s in enumerate(sums):
        want = src[i]["chunk"].double().sum() + src[i]["lps"].double().sum()
        assert abs(float(s) - float(want)) <= 1e-6 * abs(float(want)) + 1e-6, i
    assert feeder.bytes_per_batch == 8 * 32000 * 4 + 8 * 3075 * 200 * 4


@pytest.mark.gpu
def test_pinned_feeder_variable_batches_and_end_of_data():
    """A batch of another shape (partial last batch) gets its own slot buffers instead of being broadcast into the old
    ones, non-tensor entries pass through, and a source that runs dry while prefetching raises StopIteration only when
    the missing batch would be handed out."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    src = [{"chunk": torch.randn(8, 1, 4000, generator=g), "uttname": ["u%d" % i] * 8} for i in range(3)]
    src.append({"chunk": torch.randn(3, 1, 4000, generator=g), "uttname": ["last"] * 3})      # partial batch
    it = iter(src)
    feeder = P.PinnedBatchFeeder(lambda: next(it), dev, depth=2)
    got = []
    for i in range(4):
        d = feeder.next()
        assert d["uttname"] == src[i]["uttname"]
        got.append(d["chunk"].clone())
    torch.cuda.synchronize()
    for i in range(4):
        assert got[i].shape == src[i]["chunk"].shape
        assert torch.equal(got[i].cpu(), src[i]["chunk"])
    with pytest.raises(StopIteration):
        feeder.next()


def test_from_config_accepts_the_reference_distortion_schema(dev, tmp_path):
    """DeviceBatchProducer.from_config takes the dict train.py --dtrans_cfg loads (pase/transforms.py:38-146): same keyword
    names / defaults / gating, files read from the configured roots, unknown keywords rejected like the reference's
    function signature would, transforms outside this engine's scope refused loudly."""
    import json
    import wave
    rs = np.random.RandomState(0)
    root = tmp_path
    (root / "irs").mkdir()
    (root / "noises").mkdir()
    (root / "filts").mkdir()
    irs = []
    for i in range(3):
        ir = rs.randn(900) * np.exp(-np.arange(900) / 150.0)
        ir[5 + i] = 4.0
        np.save(str(root / "irs" / ("IR_%d.npy" % i)), ir)
        irs.append(ir)
    for i in range(2):
        x = (rs.randn(9000) * 0.2 * 32767).astype(np.int16)
        with wave.open(str(root / "noises" / ("n%d.wav" % i)), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.tobytes())
    np.save(str(root / "filts" / "bd0.npy"), np.hamming(61) * np.sinc(0.3 * (np.arange(61) - 30)))
    cfg = {"reverb_irfiles": ["IR_0.npy", "IR_1.npy", "IR_2.npy"], "reverb_fmt": "npy", "reverb_data_root": str(root / "irs"),
           "reverb_p": 1, "noises_dir": [str(root / "noises")], "noises_snrs": [0, 5, 10], "noises_p": 0.5,
           "bandrop_irfiles": ["bd0.npy"], "bandrop_data_root": str(root / "filts"), "bandrop_p": 0.4,
           "clip_factors": [0.3, 0.5], "clip_p": 0.25, "overlap_p": 0.25}          # overlap_dir missing -> gated off
    cfg = json.loads(json.dumps(cfg))
    g = _G
    rng = np.random.RandomState(4)
    pool = P.WavPool([g["wav0"], g["wav1"], g["wav2"]], dev)
    prod = P.DeviceBatchProducer.from_config(P.DeviceChunker(pool, 1600, rng=rng), cfg, rng=rng, device=dev)
    assert prod.reverb is not None and prod.reverb.n == 3 and prod.reverb_p == 1
    assert prod.additive is not None and len(prod.additive.noises) == 2 and prod.additive_p == 0.5
    assert prod.bandrop is not None and prod.bandrop_p == 0.4 and prod.downsample is None and prod.overlap is None
    assert prod.clipping.clip_factors == [0.3, 0.5] and prod.clip_p == 0.25
    assert prod.skipped == ["Codec2"]           # on by default in the reference (p = 0.3): listed, never silently absent
    # the IRs are prepared as Reverb.load_IR does: divided by their maximum
    want = (irs[1] / np.abs(np.max(irs[1]))).astype(np.float32)
    o, n = int(prod.reverb.irs.off[1]), int(prod.reverb.irs.len[1])
    np.testing.assert_allclose(prod.reverb.irs.pool[o:o + n].cpu().numpy(), want, rtol=1e-6)
    batch = prod(4)
    assert tuple(batch["chunk"].shape) == (4, 1, 1600) and torch.isfinite(batch["chunk"]).all()
    assert not torch.equal(batch["chunk"], batch["cchunk"])          # reverb_p = 1: every chunk is distorted
    with pytest.raises(TypeError):
        P.DeviceBatchProducer.from_config(P.DeviceChunker(pool, 1600, rng=rng), dict(cfg, no_such_option=1), device=dev)
    with pytest.raises(NotImplementedError):
        P.DeviceBatchProducer.from_config(P.DeviceChunker(pool, 1600, rng=rng), dict(cfg, speed_range=[0.9, 1.1]), device=dev)
    with pytest.raises(FileNotFoundError):
        P.DeviceBatchProducer.from_config(P.DeviceChunker(pool, 1600, rng=rng), dict(cfg, reverb_irfiles=["nope.npy"]),
                                          device=dev)
    # the shipped cfg (files absent in this image) builds with seeded synthetic pools when asked to
    import os
    shipped = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfg", "distortions", "PASE+distortions.cfg")
    if os.path.exists(shipped):
        with open(shipped) as f:
            sc = json.load(f)
        with pytest.raises(NotImplementedError):          # it enables the Chopper (needs a VAD): refused unless told to skip
            P.DeviceBatchProducer.from_config(P.DeviceChunker(pool, 1600, rng=rng), sc, device=dev, synthetic_ok=True)
        prod2 = P.DeviceBatchProducer.from_config(P.DeviceChunker(pool, 1600, rng=rng), sc, rng=rng, device=dev,
                                                  synthetic_ok=True, unsupported="skip")
        assert "Chopper" in prod2.skipped and "Codec2" in prod2.skipped
        assert prod2.reverb.n == len(sc["reverb_irfiles"])
        assert tuple(prod2(2)["chunk"].shape) == (2, 1, 1600)

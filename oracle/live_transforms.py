"""Run the LIVE reference transforms / collater / Gap worker (pase/transforms.py, pase/dataset.py,
pase/models/Minions) on seeded synthetic inputs and record inputs, the random draws they made, and outputs.

TEST INFRASTRUCTURE ONLY (build container: /root/reference must exist).  Two consumers:
  * `python oracle/live_transforms.py` writes tests/golden/transforms_live.npz (committed; travels to the GPU
    box, where the device producers are compared with it);
  * tests/test_transform_pins.py calls `run()` to pin oracle/producer_oracle.py, oracle/dsp_oracle.lps / znorm and
    oracle/pase_oracle.gap_samples against the live classes, and to check the committed golden is reproducible.

How the draws are recovered: every live call is made under fixed seeds of `random`, `numpy.random` and `torch`;
the same seeds are then replayed and the draws repeated in the order the reference code makes them (cited per
case).  The pin test proves the replay right: oracle(draws) == live output.
"""
import os
import pickle
import random
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "transforms_live.npz")
T_CHUNK = 1600
HOP = 160


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def _write_wav(path, x, rate=16000):
    import scipy.io.wavfile
    scipy.io.wavfile.write(path, rate, np.asarray(x, dtype=np.float32))


def fixtures():
    """Synthetic waveforms / IRs / noises / filters (float32 where the reference reads float wavs)."""
    rs = np.random.RandomState(20240)
    fx = {}
    fx["wavs"] = [(0.3 * rs.standard_normal(n)).astype(np.float32) for n in (5000, 2100, 1300)]   # 1300 < T_CHUNK: padded
    irs = []
    for L, peak in ((300, 7), (90, 3)):      # (a peak at tap 0 crashes the reference's own shift(): xs[:-0])
        ir = rs.standard_normal(L) * np.exp(-np.arange(L) / (L / 5.0))
        ir[peak] = 2.5
        irs.append(ir)
    fx["irs"] = irs
    fx["noises"] = [(0.05 * rs.standard_normal(4000)).astype(np.float32), (0.1 * rs.standard_normal(900)).astype(np.float32)]
    fx["speech"] = [(0.2 * rs.standard_normal(n)).astype(np.float32) for n in (5000, 700)]
    fx["bandrop"] = [np.sinc(np.arange(-50, 51) / 2.0) * np.hamming(101)]
    fx["downsample"] = [rs.standard_normal(64)]
    return fx


def run():
    """Returns a flat dict of numpy arrays (inputs, draws, live outputs)."""
    ref_shim.install_transforms()
    import pase.transforms as TR
    from pase.dataset import DictCollater
    fx = fixtures()
    out = {}
    for i, w in enumerate(fx["wavs"]):
        out["wav%d" % i] = w
    for k in ("irs", "noises", "speech", "bandrop", "downsample"):
        for i, v in enumerate(fx[k]):
            out["%s%d" % (k, i)] = np.asarray(v)
    tmp = tempfile.mkdtemp(prefix="pase_live_")
    ir_dir, n_dir, s_dir, f_dir = (os.path.join(tmp, d) for d in ("ir", "noise", "speech", "filt"))
    for d in (ir_dir, n_dir, s_dir, f_dir):
        os.makedirs(d)
    ir_files = []
    for i, ir in enumerate(fx["irs"]):
        np.save(os.path.join(ir_dir, "ir%d.npy" % i), ir)
        ir_files.append("ir%d.npy" % i)
    for i, n in enumerate(fx["noises"]):
        _write_wav(os.path.join(n_dir, "n%d.wav" % i), n)
    for i, n in enumerate(fx["speech"]):
        _write_wav(os.path.join(s_dir, "s%d.wav" % i), n)
    np.save(os.path.join(f_dir, "bd0.npy"), fx["bandrop"][0])
    np.save(os.path.join(f_dir, "ds0.npy"), fx["downsample"][0])

    # ---- (a19) MIChunkWav (transforms.py:388-436) with random_scale: draws = np.random.randint per crop of a file
    # longer than the chunk (select_chunk :350), in the order raw, raw_ctxt(= raw), raw_rand; then torch.rand(1) per
    # norm_and_scale (:148-151) in the order chunk, chunk_ctxt, chunk_rand
    wavs = fx["wavs"]
    pairs = [(0, 1), (2, 0), (1, 2)]
    out["chunk_pairs"] = np.array(pairs)
    for ci, (a, b) in enumerate(pairs):
        seed_all(100 + ci)
        pkg = TR.MIChunkWav(T_CHUNK, random_scale=True)({"raw": torch.from_numpy(wavs[a].copy()),
                                                          "raw_rand": torch.from_numpy(wavs[b].copy())})
        for k in ("chunk", "chunk_ctxt", "chunk_rand"):
            out["mi%d_%s" % (ci, k)] = pkg[k].numpy()
        seed_all(100 + ci)
        begs = [int(np.random.randint(0, len(wavs[j]) - T_CHUNK)) if len(wavs[j]) > T_CHUNK else 0 for j in (a, a, b)]
        scales = [float(torch.rand(1)) for _ in range(3)]
        out["mi%d_beg" % ci] = np.array(begs)
        out["mi%d_scale" % ci] = np.array(scales, dtype=np.float32)
        assert int(pkg["chunk_beg_i"]) == begs[0]

    # the clean chunk every distortion below starts from
    seed_all(7)
    clean = TR.SingleChunkWav(T_CHUNK, random_scale=False)({"raw": torch.from_numpy(wavs[0].copy())})["chunk"]
    out["clean"] = clean.numpy().copy()

    def pkg0():
        return {"chunk": clean.clone(), "chunk_beg_i": 0, "chunk_end_i": T_CHUNK, "dec_resolution": 1}

    # ---- (a27) Reverb.__call__ (:1071-1103): draw = random.choice(ir_idxs) (:1066)
    for ci in range(2):
        seed_all(200 + ci)
        rv = TR.Reverb(list(ir_files), ir_fmt="npy", data_root=ir_dir, max_reverb_len=200)
        out["reverb%d" % ci] = rv(pkg0())["chunk"].numpy()
        seed_all(200 + ci)
        out["reverb%d_ir" % ci] = np.array(random.choice(list(range(len(ir_files)))))
    out["reverb_max_len"] = np.array(200)

    # ---- (a27) SimpleAdditive.__call__ (:1633-1675): draws = np.random.randint(len(noises)) (:1614),
    # np.random.randint(0, len(noise) - T) when the noise is longer than the chunk (:1653), random.choice(snr) (:1658)
    noise_files = sorted(os.listdir(n_dir))
    for ci in range(3):
        seed_all(300 + ci)
        ad = TR.SimpleAdditive(n_dir, snr_levels=[0, 5, 10])
        order = [os.path.basename(f) for f in ad.noises]          # glob order
        out["additive%d" % ci] = ad(pkg0())["chunk"].numpy()
        seed_all(300 + ci)
        ni = int(np.random.randint(0, len(order)))
        nidx = noise_files.index(order[ni])
        L = len(fx["noises"][nidx])
        nbeg = int(np.random.randint(0, L - T_CHUNK)) if L > T_CHUNK else 0
        snr = random.choice([0, 5, 10])
        out["additive%d_draw" % ci] = np.array([nidx, nbeg, snr])

    # ---- SimpleAdditiveShift.__call__ (:1714-1766) with a reverberated interferer and the 'overlap' label at the hop
    # rate: draws = np.random.randint(0, int(.75 T)) shift (:1719), sample_noise (:1614), np.random.randint crop
    # (:1730), noise_transform -> Reverb's random.choice, random.choice(snr) (:1754)
    sp_files = sorted(os.listdir(s_dir))
    for ci in range(3):
        seed_all(400 + ci)
        rvn = TR.Reverb(list(ir_files), ir_fmt="npy", data_root=ir_dir, max_reverb_len=200)
        ov = TR.SimpleAdditiveShift(s_dir, snr_levels=[5, 7.5, 10], noise_transform=rvn)
        order = [os.path.basename(f) for f in ov.noises]
        p = pkg0()
        p["overlap"] = torch.zeros(T_CHUNK // HOP)
        p["dec_resolution"] = HOP
        res = ov(p)
        out["overlap%d" % ci] = res["chunk"].numpy()
        out["overlap%d_label" % ci] = res["overlap"].numpy()
        seed_all(400 + ci)
        shift = int(np.random.randint(0, int(0.75 * T_CHUNK)))
        si = sp_files.index(order[int(np.random.randint(0, len(order)))])
        Ls, need = len(fx["speech"][si]), T_CHUNK - shift
        sbeg = int(np.random.randint(0, Ls - need)) if Ls > need else 0
        iri = random.choice(list(range(len(ir_files))))
        snr = random.choice([5, 7.5, 10])
        out["overlap%d_draw" % ci] = np.array([si, sbeg, shift, iri, snr], dtype=np.float64)

    # ---- Clipping (:1514-1535): random.choice(clip_factors); BandDrop / Downsample (:1113-1300): random.choice(idx)
    seed_all(500)
    out["clipping"] = TR.Clipping([0.3, 0.4, 0.5])(pkg0())["chunk"].numpy()
    seed_all(500)
    out["clipping_cf"] = np.array(random.choice([0.3, 0.4, 0.5]))
    seed_all(501)
    out["bandrop"] = TR.BandDrop(["bd0.npy"], filt_fmt="npy", data_root=f_dir)(pkg0())["chunk"].numpy()
    seed_all(502)
    out["downsample"] = TR.Downsample(["ds0.npy"], filt_fmt="npy", data_root=f_dir)(pkg0())["chunk"].numpy()

    # ---- the whole chain in config_distortions order (:38-146) under PCompose's Bernoulli gating (:208-237):
    # one random.random() per transform, BEFORE that transform's own draws
    probs = dict(reverb_p=0.7, overlap_p=0.6, noises_p=0.7, clip_p=0.5, bandrop_p=0.6, downsample_p=0.5)
    for ci in range(4):
        seed_all(600 + ci)
        chain = TR.config_distortions(reverb_irfiles=list(ir_files), reverb_fmt="npy", reverb_data_root=ir_dir,
                                      overlap_dir=s_dir, overlap_list=None, overlap_snrs=[5, 7.5, 10],
                                      overlap_reverb=False, noises_dir=n_dir, noises_snrs=[0, 5, 10],
                                      clip_factors=[0.3, 0.4, 0.5], bandrop_irfiles=["bd0.npy"], bandrop_fmt="npy",
                                      bandrop_data_root=f_dir, downsample_irfiles=["ds0.npy"], downsample_fmt="npy",
                                      downsample_data_root=f_dir, codec2_p=0.0, **probs)
        names = [t.__class__.__name__ for t in chain.transforms]
        p = pkg0()
        p["overlap"] = torch.zeros(T_CHUNK // HOP)
        p["dec_resolution"] = HOP
        seed_all(600 + ci)
        res = chain(p)
        out["chain%d" % ci] = res["chunk"].numpy()
        out["chain%d_label" % ci] = res["overlap"].numpy()
        # replay
        seed_all(600 + ci)
        n_order = [os.path.basename(f) for f in chain.transforms[names.index("SimpleAdditive")].noises]
        s_order = [os.path.basename(f) for f in chain.transforms[names.index("SimpleAdditiveShift")].noises]
        d = dict(reverb_ir=-1, ov_src=-1, ov_beg=0, ov_shift=0, ov_snr=0.0, add_idx=-1, add_beg=0, add_snr=0.0, clip=0.0,
                 bandrop=-1, downsample=-1)
        for name, prob in zip(names, chain.probs):
            if not (random.random() < prob):
                continue
            if name == "Reverb":
                d["reverb_ir"] = random.choice(list(range(len(ir_files))))
            elif name == "SimpleAdditiveShift":
                d["ov_shift"] = int(np.random.randint(0, int(0.75 * T_CHUNK)))
                d["ov_src"] = sp_files.index(s_order[int(np.random.randint(0, len(s_order)))])
                Ls, need = len(fx["speech"][d["ov_src"]]), T_CHUNK - d["ov_shift"]
                d["ov_beg"] = int(np.random.randint(0, Ls - need)) if Ls > need else 0
                d["ov_snr"] = random.choice([5, 7.5, 10])
            elif name == "SimpleAdditive":
                d["add_idx"] = noise_files.index(n_order[int(np.random.randint(0, len(n_order)))])
                L = len(fx["noises"][d["add_idx"]])
                d["add_beg"] = int(np.random.randint(0, L - T_CHUNK)) if L > T_CHUNK else 0
                d["add_snr"] = random.choice([0, 5, 10])
            elif name == "Clipping":
                d["clip"] = random.choice([0.3, 0.4, 0.5])
            elif name == "BandDrop":
                d["bandrop"] = random.choice([0])
            elif name == "Downsample":
                d["downsample"] = random.choice([0])
            else:
                raise AssertionError(name)
        out["chain%d_draw" % ci] = np.array([d[k] for k in ("reverb_ir", "ov_src", "ov_beg", "ov_shift", "ov_snr", "add_idx",
                                                             "add_beg", "add_snr", "clip", "bandrop", "downsample")],
                                            dtype=np.float64)
    out["chain_order"] = np.array(names)

    # ---- (a20) LPS (:439-487) and (a25) ZNorm (:183-205) on the clean chunk
    for nm, kw in (("lps", dict(n_fft=2048, hop=HOP, win=400)), ("lps_long", dict(n_fft=2048, hop=HOP, win=512))):
        out[nm] = TR.LPS(name=nm, **kw)({"chunk": clean.clone()})[nm].numpy()
    rs = np.random.RandomState(5)
    stats = {"lps": {"mean": torch.from_numpy(rs.standard_normal(3075).astype(np.float32)),
                     "std": torch.from_numpy((0.5 + rs.random_sample(3075)).astype(np.float32))}}
    sp = os.path.join(tmp, "stats.pkl")
    with open(sp, "wb") as f:
        pickle.dump(stats, f)
    out["znorm_mean"], out["znorm_std"] = stats["lps"]["mean"].numpy(), stats["lps"]["std"].numpy()
    out["lps_znorm"] = TR.ZNorm(sp)({"chunk": clean.clone(), "lps": torch.from_numpy(out["lps"].copy())})["lps"].numpy()

    # ---- (a26) DictCollater (dataset.py:21-89) on per-utterance packages as the dataset emits them (:482-513)
    pkgs = []
    for ci in range(3):
        p = {k: torch.from_numpy(out["mi%d_%s" % (ci, k)]) for k in ("chunk", "chunk_ctxt", "chunk_rand")}
        p["cchunk"] = p["chunk"].clone()
        p["overlap"] = torch.zeros(T_CHUNK // HOP)
        p["lps"] = torch.from_numpy(out["lps"][:, :T_CHUNK // HOP].copy())
        p["uttname"] = "utt%d.wav" % ci
        p["dec_resolution"] = HOP
        pkgs.append(p)
    batch = DictCollater()(pkgs)
    out["collate_keys"] = np.array(sorted(batch.keys()))
    for k, v in batch.items():
        out["collate_" + k] = v.numpy()

    # ---- (a16) Gap worker (cls_minions.py:117-131 -> Minions/minions.py:651-704): np.random.randint(0, T, B) twice
    from pase.models.Minions.cls_minions import cls_worker_maker
    seed_all(700)
    cfg = {"num_outputs": 1, "dropout": 0, "hidden_size": 16, "hidden_layers": 1, "name": "gap", "type": "gap",
           "loss": "MSELoss", "skip": False}
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        gap = cls_worker_maker(dict(cfg), 12)
    x = torch.randn(8, 12, 3)       # T = 3: |a - b| = T - 1 (label 1 after the truncation) happens
    seed_all(701)
    # legacy-torch adapter: minions.py:689 divides two LongTensors (integer division on the reference's torch 1.x,
    # true division today) and :693 builds a LongTensor from the list of 0-dim results, which torch 2.x rejects for
    # float elements.  |a-b|/(T-1) lies in [0, 1], so truncating the true quotient equals the legacy integer quotient.
    _LT = torch.LongTensor

    def _long_tensor(v):
        if isinstance(v, (list, tuple)) and len(v) and torch.is_tensor(v[0]):
            return torch.tensor([int(e) for e in v], dtype=torch.long)
        return _LT(v)
    torch.LongTensor = _long_tensor
    try:
        y, lab = gap(x, 1, device="cpu")
    finally:
        torch.LongTensor = _LT
    seed_all(701)
    out["gap_aidx"] = np.random.randint(0, 3, size=8)
    out["gap_bidx"] = np.random.randint(0, 3, size=8)
    out["gap_x"] = x.numpy()
    out["gap_y"] = y.detach().numpy()
    out["gap_label"] = lab.numpy()
    out["gap_param_names"] = np.array([n for n, _ in gap.named_parameters()])
    for n, p in gap.named_parameters():
        out["gap_p_" + n] = p.detach().numpy()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return out


if __name__ == "__main__":
    res = run()
    np.savez(GOLD, **res)
    print(GOLD, os.path.getsize(GOLD), len(res), "arrays")

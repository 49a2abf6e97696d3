"""On-device batch producer: the dataset / transform side of the training step (SURVEY.md section 8 rows a19,
a26, a27; (f)-2, (f)-3), batched on the GPU over HBM-resident waveforms.

Mirrors, per batch instead of per utterance:
  * MIChunkWav / SingleChunkWav (pase/transforms.py:295-436): random fixed-size crops of (utterance,
    same-context utterance, random other utterance), reflect padding of short files, optional
    norm_and_scale (:148-151);
  * the dataset's package layout + DictCollater (pase/dataset.py:21-89,428-513): every waveform key is a
    (B, 1, T) tensor, `cchunk` is the clean copy of `chunk` taken BEFORE the distortions (:496);
  * Reverb (:1001-1103) and SimpleAdditive (:1590-1680) applied to `chunk` with the reference's Bernoulli
    gating (PCompose, :208-237).
Random DECISIONS (utterance ids, crop starts, which IR / noise / SNR, the U(0,1) scales) are drawn on the host
with numpy exactly where the reference draws them; every `__call__` also accepts them explicitly so parity
tests can replay the reference's draws.  The arithmetic runs in pase_amd/csrc/producer.hip.
"""
import numpy as np
import torch

from . import kernels as K


def _dev_i32(a, device):
    return torch.as_tensor(np.asarray(a, dtype=np.int32), device=device)


class WavPool(object):
    """Waveforms (1-D float arrays) concatenated in one HBM buffer."""

    def __init__(self, wavs, device="cuda"):
        self.device = torch.device(device)
        lens = [int(len(w)) for w in wavs]
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64) if lens else np.zeros(0, np.int64)
        self.lens_host = np.asarray(lens, dtype=np.int64)
        flat = np.concatenate([np.asarray(w, dtype=np.float32).reshape(-1) for w in wavs]) if lens else np.zeros(0, np.float32)
        self.pool = torch.from_numpy(flat).to(self.device)
        self.off = torch.from_numpy(offs).to(self.device)
        self.len = torch.from_numpy(np.asarray(lens, dtype=np.int32)).to(self.device)

    def __len__(self):
        return len(self.lens_host)


class DeviceChunker(object):
    """MIChunkWav over a resident pool: __call__ -> {'chunk','chunk_ctxt','chunk_rand'} as (B, 1, T)."""

    def __init__(self, pool, chunk_size, random_scale=False, rng=None):
        self.pool, self.T, self.random_scale = pool, int(chunk_size), random_scale
        self.rng = rng if rng is not None else np.random

    def draw(self, B):
        """The reference's draws: dataset __getitem__ picks the utterance and a different random one
        (dataset.py:482-484); select_chunk draws np.random.randint(0, len - T) per crop (transforms.py:350)."""
        n = len(self.pool)
        utt = self.rng.randint(0, n, size=B)
        # a different utterance, uniformly (random.choice over the other indices, dataset.py:482-484)
        rand = self.rng.randint(0, max(n - 1, 1), size=B)
        rand = np.where(n > 1, rand + (rand >= utt), utt)
        src = np.stack([utt, utt, rand], 0)                   # chunk, chunk_ctxt (same context), chunk_rand
        L = self.pool.lens_host[src]
        beg = np.where(L > self.T, (self.rng.random_sample(src.shape) * np.maximum(L - self.T, 1)).astype(np.int64), 0)
        scale = self.rng.random_sample(src.shape).astype(np.float32) if self.random_scale else None
        return src, beg, scale

    def __call__(self, B=None, src=None, beg=None, scale=None):
        if src is None:
            src, beg, scale = self.draw(B)
        src = np.asarray(src)
        N = src.size
        out = torch.empty(N, self.T, device=self.pool.device)
        K.chunk_gather(self.pool.pool, self.pool.off, self.pool.len, _dev_i32(src.reshape(-1), self.pool.device),
                       _dev_i32(np.asarray(beg).reshape(-1), self.pool.device), out, N=N, T=self.T)
        if scale is not None:
            K.peak_scale(out, torch.as_tensor(np.asarray(scale, dtype=np.float32).reshape(-1), device=out.device),
                         N=N, T=self.T)
        Bn = src.shape[1]
        out = out.view(3, Bn, 1, self.T)
        return {"chunk": out[0], "chunk_ctxt": out[1], "chunk_rand": out[2]}


class DeviceReverb(object):
    """Reverb (pase/transforms.py:1001-1103) for a batch; IRs truncated to max_reverb_len and divided by
    |max| at load time like load_IR (:1039-1042)."""

    def __init__(self, irs, max_reverb_len=24000, device="cuda"):
        prepared, pmax = [], []
        for ir in irs:
            ir = np.asarray(ir, dtype=np.float64).reshape(-1)[:max_reverb_len]
            if np.max(ir) > 0:
                ir = ir / np.abs(np.max(ir))
            prepared.append(ir.astype(np.float32))
            pmax.append(int(np.argmax(np.abs(ir))))
        self.irs = WavPool(prepared, device)
        self.pmax = _dev_i32(pmax, device)
        self.max_len = int(max(len(i) for i in prepared))
        self.n = len(prepared)

    def __call__(self, chunk, ir_idx):
        """chunk (B, 1, T) modified in place; ir_idx[b] = IR index, -1 = leave utterance b clean."""
        B, _, T = chunk.shape
        full = torch.empty(B, T + self.max_len - 1, device=chunk.device)
        energies = torch.empty(2 * B, dtype=torch.float64, device=chunk.device)
        K.reverb(chunk, self.irs.pool, self.irs.off, self.irs.len, self.pmax, _dev_i32(ir_idx, chunk.device), full,
                 energies, B=B, T=T, max_ir_len=self.max_len)
        return chunk


class DeviceFilter(object):
    """BandDrop / Downsample (pase/transforms.py:1113-1300): FIR with a filter divided by |max| at load time
    (:1140,1241), delay compensation by round(len / 2) (Python's round, as written), trim, energy matched on the
    trimmed signal."""

    def __init__(self, filters, device="cuda"):
        prepared, shifts = [], []
        for f in filters:
            f = np.asarray(f, dtype=np.float64).reshape(-1)
            f = f / np.abs(np.max(f))
            prepared.append(f.astype(np.float32))
            shifts.append(int(round(f.shape[0] / 2)))
        self.filters = WavPool(prepared, device)
        self.shift = _dev_i32(shifts, device)
        self.max_len = int(max(len(f) for f in prepared))
        self.n = len(prepared)

    def __call__(self, chunk, filt_idx):
        B, _, T = chunk.shape
        full = torch.empty(B, T + self.max_len - 1, device=chunk.device)
        energies = torch.empty(2 * B, dtype=torch.float64, device=chunk.device)
        K.fir_distort(chunk, self.filters.pool, self.filters.off, self.filters.len, self.shift,
                      _dev_i32(filt_idx, chunk.device), full, energies, B=B, T=T, max_ir_len=self.max_len,
                      trimmed_energy=1)
        return chunk


BandDrop = Downsample = DeviceFilter


class DeviceClipping(object):
    """Clipping (pase/transforms.py:1514-1535)."""

    def __init__(self, clip_factors=(0.3, 0.4, 0.5)):
        self.clip_factors = list(clip_factors)

    def __call__(self, chunk, factor):
        """factor[b] = clip factor drawn for utterance b, <= 0 = leave untouched"""
        B, _, T = chunk.shape
        K.clip(chunk, torch.as_tensor(np.asarray(factor, dtype=np.float32), device=chunk.device), B=B, T=T)
        return chunk


class DeviceAdditive(object):
    """SimpleAdditive (pase/transforms.py:1590-1680) for a batch of resident noises."""

    def __init__(self, noises, snr_levels=(0, 5, 10), device="cuda"):
        self.noises = WavPool(noises, device)
        self.snr_levels = list(snr_levels)

    def __call__(self, chunk, noise_idx, noise_beg, snr):
        B, _, T = chunk.shape
        K.add_noise(chunk, self.noises.pool, self.noises.off, self.noises.len, _dev_i32(noise_idx, chunk.device),
                    _dev_i32(noise_beg, chunk.device),
                    torch.as_tensor(np.asarray(snr, dtype=np.float32), device=chunk.device), B=B, T=T)
        return chunk


class DeviceOverlap(object):
    """SimpleAdditiveShift (pase/transforms.py:1684-1766): another utterance, cropped to T - shift samples,
    optionally reverberated (overlap_reverb), front-padded by `shift` and mixed in at a drawn SNR with the
    energy renormalisation of SimpleAdditive.  Also returns the reference's `overlap` label (fraction of
    overlapped samples per hop)."""

    def __init__(self, speech_pool, snr_levels=(5, 7.5, 10), reverb=None):
        self.pool, self.snr_levels, self.reverb = speech_pool, list(snr_levels), reverb

    def __call__(self, chunk, src, beg, shift, snr, ir_idx=None, hop=160):
        """src[b] < 0 leaves utterance b untouched; ir_idx: IR per utterance for the interfering crop (or None)."""
        B, _, T = chunk.shape
        dev = chunk.device
        src = np.asarray(src)
        sh = np.where(src >= 0, np.asarray(shift), T)
        noise = torch.zeros(B, 1, T, device=dev)
        K.overlap_gather(self.pool.pool, self.pool.off, self.pool.len, _dev_i32(src, dev), _dev_i32(beg, dev),
                         _dev_i32(sh, dev), noise, B=B, T=T)
        if self.reverb is not None and ir_idx is not None:
            # Reverb of the (T - shift)-sample crop == Reverb of its front-padded version with the pre-echo that
            # leaks in front of `shift` removed (the reference pads after reverberating)
            self.reverb(noise, np.where(src >= 0, np.asarray(ir_idx), -1))
            K.zero_front(noise, _dev_i32(sh, dev), B=B, T=T)
        off = torch.arange(B, dtype=torch.int64, device=dev) * T
        K.add_noise(chunk, noise.view(-1), off, torch.full((B,), T, dtype=torch.int32, device=dev),
                    _dev_i32(np.where(src >= 0, np.arange(B), -1), dev), _dev_i32(np.zeros(B), dev),
                    torch.as_tensor(np.asarray(snr, dtype=np.float32), device=dev), B=B, T=T)
        t = torch.arange(T, device=dev)[None, :]
        mask = (t >= torch.as_tensor(sh, device=dev)[:, None]).float()
        return chunk, mask.view(B, T // hop, hop).mean(2)


class DeviceBatchProducer(object):
    """dataset.__getitem__ + DictCollater for one batch: chunks, clean copy, gated distortions, and (when a
    DeviceTargets is attached) the regression labels computed from the clean chunk."""

    def __init__(self, chunker, reverb=None, reverb_p=0.5, additive=None, additive_p=0.5, targets=None, rng=None,
                 clipping=None, clip_p=0.2, bandrop=None, bandrop_p=0.35, downsample=None, downsample_p=0.25,
                 overlap=None, overlap_p=0.1):
        self.chunker, self.reverb, self.additive, self.targets = chunker, reverb, additive, targets
        self.reverb_p, self.additive_p = reverb_p, additive_p
        self.clipping, self.clip_p = clipping, clip_p
        self.bandrop, self.bandrop_p = bandrop, bandrop_p
        self.downsample, self.downsample_p = downsample, downsample_p
        self.overlap, self.overlap_p = overlap, overlap_p
        self.rng = rng if rng is not None else np.random

    # ---- the reference's distortion-config schema ---------------------------------------------------------------
    @classmethod
    def from_config(cls, chunker, cfg, targets=None, rng=None, device="cuda", synthetic_ok=False, sr=16000,
                    unsupported="raise"):
        """Build the producer from the dict `train.py --dtrans_cfg` loads (cfg/distortions/*.cfg) with the keyword
        names, defaults and gating rules of pase/transforms.py:38-146 config_distortions: a transform is active when
        its probability is > 0 AND its file list / directory is given; order reverb, overlap-speech, noises, clip,
        band-drop, downsample (draw_chain / apply_chain).  Files are read with the reference's own loaders' rules
        (load_IR :1028-1044: 'npy' | 'imp' / 'txt' | 'wav' | 'mat'; noises and overlap speech: every .wav under the
        directories / the .scp list).  `synthetic_ok=True` substitutes seeded synthetic pools for entries whose files
        do not exist (this image ships the cfgs, not the corpora): exponentially decaying IRs of the configured count,
        coloured noise, band-limited FIRs -- the schema and gating are exercised, the audio is not the reference's.
        Transforms this engine does not implement (speed / resample / chop) raise
        NotImplementedError when the cfg enables them (`unsupported="skip"` drops them instead); either way the names
        of everything dropped are in `producer.skipped`.  Codec2 (on by default in the reference, p = 0.3, through the
        optional pycodec2 package) is always listed there."""
        import os
        c = dict(reverb_irfiles=None, reverb_fmt="imp", reverb_data_root=".", reverb_p=0.5, overlap_dir=None,
                 overlap_list=None, overlap_snrs=[0, 5, 10], overlap_reverb=False, overlap_p=0.5, noises_dir=None,
                 noises_snrs=[0, 5, 10], noises_p=0.5, speed_range=None, speed_p=0.5, resample_factors=[], resample_p=0.5,
                 bandrop_irfiles=[], bandrop_fmt="npy", bandrop_data_root=".", bandrop_p=0.5, downsample_irfiles=[],
                 downsample_fmt="npy", downsample_data_root=".", downsample_p=0.5, clip_factors=[], clip_p=0.5,
                 chop_factors=[], max_chops=5, chop_p=0.5, codec2_p=0.3, codec2_kbps=1600, codec2_cachedir=None,
                 codec2_cache=False, reverb_cache=False, noises_cache=False, report=False)
        unknown = set(cfg) - set(c)
        if unknown:
            raise TypeError("config_distortions() got unexpected keyword arguments %s" % sorted(unknown))
        c.update(cfg)
        skipped = []

        def drop(name, hard=True):
            if hard and unsupported != "skip":
                raise NotImplementedError("pase_amd producer: %s is enabled by the distortion cfg and not implemented "
                                          "(from_config(..., unsupported='skip') drops it)" % name)
            skipped.append(name)
        if c["speed_p"] > 0. and c["speed_range"] is not None:
            drop("SpeedChange")
        if c["resample_p"] > 0. and len(c["resample_factors"]) > 0:
            drop("Resample")
        if c["chop_p"] > 0. and len(c["chop_factors"]) > 0:
            drop("Chopper")
        if c["codec2_p"] > 0.:
            drop("Codec2", hard=False)
        srng = np.random.RandomState(1234)

        def load_arr(path, fmt):
            if fmt == "npy":
                return np.load(path)
            if fmt in ("imp", "txt"):
                return np.loadtxt(path)
            if fmt == "wav":
                import wave
                with wave.open(path, "rb") as w:
                    raw = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
                return raw.astype(np.float64) / 32768.0
            if fmt == "mat":
                from scipy.io import loadmat
                return loadmat(path, squeeze_me=True, struct_as_record=False)["risp_imp"]
            raise TypeError("Unrecognized IR format: %s" % fmt)

        def files(names, root, fmt, synth):
            out = []
            for n in names:
                path = os.path.join(root, n)
                if os.path.exists(path):
                    out.append(np.asarray(load_arr(path, fmt), dtype=np.float64).reshape(-1))
                elif synthetic_ok:
                    out.append(synth())
                else:
                    raise FileNotFoundError(path)
            return out

        def synth_ir():
            n = int(srng.randint(4000, 24000))
            ir = srng.randn(n) * np.exp(-np.arange(n) / (0.15 * n))
            ir[int(srng.randint(0, 200))] = 3.0 * np.abs(ir).max()          # a direct path
            return ir

        def synth_fir():
            n = 2 * int(srng.randint(30, 120)) + 1
            t = np.arange(n) - n // 2
            fc = srng.uniform(0.1, 0.45)
            return np.sinc(2 * fc * t) * np.hamming(n)

        def wavs_under(dirs, lst):
            dirs = [dirs] if isinstance(dirs, str) else list(dirs)
            names = []
            if lst is not None and not os.path.exists(lst) and not synthetic_ok:
                # the reference's SimpleAdditiveShift opens the list and fails (transforms.py): a typo in the cfg must not
                # silently train on another overlap pool
                raise FileNotFoundError("overlap_list %r does not exist" % (lst,))
            if lst is not None and os.path.exists(lst):
                with open(lst) as f:
                    names = [os.path.join(dirs[0], ln.strip()) for ln in f if ln.strip()]
                missing = [n for n in names if not os.path.exists(n)]
                if missing and not synthetic_ok:
                    raise FileNotFoundError("%d files of overlap_list %r do not exist (first: %s)" % (len(missing), lst, missing[0]))
            else:
                for d in dirs:
                    if os.path.isdir(d):
                        for root, _, fs in os.walk(d):
                            names += [os.path.join(root, f) for f in sorted(fs) if f.endswith(".wav")]
            pool = [np.asarray(load_arr(n, "wav"), dtype=np.float32) for n in names if os.path.exists(n)]
            if not pool:
                if not synthetic_ok:
                    raise FileNotFoundError("no .wav under %s" % (dirs,))
                for _ in range(8):
                    n = int(srng.randint(3 * sr, 8 * sr))
                    x = np.cumsum(srng.randn(n)) * 0.01 + 0.3 * srng.randn(n)       # coloured noise
                    pool.append((x / np.abs(x).max() * 0.5).astype(np.float32))
            return pool

        kw = dict(targets=targets, rng=rng)
        reverb = None
        if c["reverb_irfiles"] is not None:          # the reference builds Reverb whenever the list is given (:84-88)
            reverb = DeviceReverb(files(c["reverb_irfiles"], c["reverb_data_root"], c["reverb_fmt"], synth_ir), device=device)
        if c["reverb_p"] > 0. and reverb is not None:
            kw.update(reverb=reverb, reverb_p=c["reverb_p"])
        if c["overlap_p"] > 0. and c["overlap_dir"] is not None:
            pool = WavPool(wavs_under(c["overlap_dir"], c["overlap_list"]), device)
            kw.update(overlap=DeviceOverlap(pool, c["overlap_snrs"], reverb=reverb if c["overlap_reverb"] else None),
                      overlap_p=c["overlap_p"])
        if c["noises_p"] > 0. and c["noises_dir"] is not None:
            kw.update(additive=DeviceAdditive(wavs_under(c["noises_dir"], None), c["noises_snrs"], device=device),
                      additive_p=c["noises_p"])
        if c["clip_p"] > 0. and len(c["clip_factors"]) > 0:
            kw.update(clipping=DeviceClipping(c["clip_factors"]), clip_p=c["clip_p"])
        if c["bandrop_p"] > 0. and c["bandrop_irfiles"] is not None and len(c["bandrop_irfiles"]) > 0:
            kw.update(bandrop=DeviceFilter(files(c["bandrop_irfiles"], c["bandrop_data_root"], c["bandrop_fmt"], synth_fir),
                                           device=device), bandrop_p=c["bandrop_p"])
        if c["downsample_p"] > 0. and len(c["downsample_irfiles"]) > 0:
            kw.update(downsample=DeviceFilter(files(c["downsample_irfiles"], c["downsample_data_root"],
                                                    c["downsample_fmt"], synth_fir), device=device),
                      downsample_p=c["downsample_p"])
        prod = cls(chunker, **kw)
        prod.skipped = skipped
        return prod

    def draw_chain(self, B, T):
        """The random decisions of one batch's distortion chain, per utterance: PCompose draws one Bernoulli per
        transform (pase/transforms.py:221-229), then the transform draws its own parameters.  A gated-off transform
        is encoded as index -1 (clip factor 0)."""
        rng = self.rng
        d = {}
        if self.reverb is not None:
            gate = rng.random_sample(B) < self.reverb_p
            d["reverb_ir"] = np.where(gate, rng.randint(0, self.reverb.n, size=B), -1)
        if self.overlap is not None:
            gate = rng.random_sample(B) < self.overlap_p
            src = rng.randint(0, len(self.overlap.pool), size=B)
            shift = rng.randint(0, int(0.75 * T), size=B)
            nl = self.overlap.pool.lens_host[src]
            need = T - shift
            d["ov_src"] = np.where(gate, src, -1)
            d["ov_shift"] = shift
            d["ov_beg"] = np.where(nl > need, (rng.random_sample(B) * np.maximum(nl - need, 1)).astype(np.int64), 0)
            d["ov_snr"] = np.asarray(self.overlap.snr_levels, dtype=np.float32)[rng.randint(0, len(self.overlap.snr_levels), size=B)]
            d["ov_ir"] = rng.randint(0, self.overlap.reverb.n, size=B) if self.overlap.reverb is not None else None
        if self.additive is not None:
            gate = rng.random_sample(B) < self.additive_p
            idx = rng.randint(0, len(self.additive.noises), size=B)
            nl = self.additive.noises.lens_host[idx]
            d["add_idx"] = np.where(gate, idx, -1)
            d["add_beg"] = np.where(nl > T, (rng.random_sample(B) * np.maximum(nl - T, 1)).astype(np.int64), 0)
            d["add_snr"] = np.asarray(self.additive.snr_levels, dtype=np.float32)[rng.randint(0, len(self.additive.snr_levels), size=B)]
        if self.clipping is not None:
            gate = rng.random_sample(B) < self.clip_p
            cf = np.asarray(self.clipping.clip_factors, dtype=np.float32)[rng.randint(0, len(self.clipping.clip_factors), size=B)]
            d["clip"] = np.where(gate, cf, 0.0)
        for name, filt, prob in (("bandrop", self.bandrop, self.bandrop_p), ("downsample", self.downsample, self.downsample_p)):
            if filt is not None:
                gate = rng.random_sample(B) < prob
                d[name] = np.where(gate, rng.randint(0, filt.n, size=B), -1)
        return d

    def apply_chain(self, batch, d):
        """config_distortions order (pase/transforms.py:83-141): reverb, overlap-speech, noises, clip, [chop], bandrop,
        downsample, on `chunk` only, with the decisions `d` of draw_chain (or replayed from the reference)."""
        chunk = batch["chunk"]
        if self.reverb is not None and "reverb_ir" in d:
            self.reverb(chunk, d["reverb_ir"])
        if self.overlap is not None and "ov_src" in d:
            _, ov = self.overlap(chunk, d["ov_src"], d["ov_beg"], d["ov_shift"], d["ov_snr"], d.get("ov_ir"))
            batch["overlap"] = ov.unsqueeze(1)          # (B, 1, F): DictCollater turns a 1-D label into (1, 1, F) per item
        if self.additive is not None and "add_idx" in d:
            self.additive(chunk, d["add_idx"], d["add_beg"], d["add_snr"])
        if self.clipping is not None and "clip" in d:
            self.clipping(chunk, d["clip"])
        for name, filt in (("bandrop", self.bandrop), ("downsample", self.downsample)):
            if filt is not None and name in d:
                filt(chunk, d[name])
        return batch

    def __call__(self, B):
        batch = self.chunker(B)
        batch = {k: v.contiguous() for k, v in batch.items()}
        batch["cchunk"] = batch["chunk"].clone()                       # dataset.py:496, before the distortions
        T = batch["chunk"].shape[-1]
        self.apply_chain(batch, self.draw_chain(B, T))
        if self.targets is not None:
            batch.update(self.targets(batch["cchunk"]))
        return batch


class PinnedBatchFeeder(object):
    """Host -> HBM leg of the step for host-produced batches (what `format_frontend_chunk`'s `.to(device)`,
    pase/models/modules.py:16-31, and the label `.to(device)` of pase/models/pase.py:338 do synchronously in the
    reference): a ring of page-locked host slots, `depth` device slots and ONE copy stream.  `next()` hands out
    the device batch whose copy was issued a step earlier (the compute stream waits on its event, the host
    never blocks) and immediately issues the copy of the following batch, so the 205 MB / step of PASE+ bs32
    travel over PCIe (SDMA engines, no CUs) underneath the previous step's kernels.

    `source` is any callable returning a dict of CPU tensors (a DataLoader iterator's `next`, or the bench's
    synthetic generator).  A device slot is recycled only after the compute stream has consumed it
    (per-slot 'free' event recorded when the NEXT batch is requested)."""

    def __init__(self, source, device="cuda", depth=2):
        if depth < 2:
            raise ValueError("PinnedBatchFeeder needs depth >= 2 (one slot in flight, one being consumed)")
        self.source = source
        self.device = torch.device(device)
        self.depth = int(depth)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.host = [None] * self.depth          # pinned staging slots
        self.dev = [None] * self.depth
        self.ready = [None] * self.depth         # copy finished
        self.free = [None] * self.depth          # compute finished with the slot
        self.sig = [None] * self.depth           # {key: (shape, dtype)} the slot's buffers were allocated for
        self.dev_extra = [None] * self.depth     # non-tensor entries of the batch in the slot
        self.i = 0
        self.bytes_per_batch = 0
        self._issue(0)

    def _issue(self, slot):
        try:
            batch = self.source()
        except StopIteration:
            # the loader ran dry while prefetching one batch past the last one handed out: remember it, raise on the
            # next() that would hand that slot out
            self.ready[slot] = None
            self.dev_extra[slot] = StopIteration
            return
        # non-tensor entries (utterance ids, lengths as python objects ...) pass through untouched
        extra = {k: v for k, v in batch.items() if not torch.is_tensor(v)}
        batch = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        self.dev_extra[slot] = extra
        sig = {k: (tuple(v.shape), v.dtype) for k, v in batch.items()}
        if self.dev[slot] is None or self.sig[slot] != sig:
            # first use of the slot, or a batch of another layout (partial last batch, variable chunk length, a key
            # that comes and goes): new device / staging buffers for the slot -- never a silent broadcast into old ones
            if self.free[slot] is not None:
                self.free[slot].synchronize()
            self.dev[slot] = {k: torch.empty(v.shape, dtype=v.dtype, device=self.device) for k, v in batch.items()}
            self.host[slot] = {}
            self.sig[slot] = sig
            self.bytes_per_batch = sum(v.numel() * v.element_size() for v in batch.values())
        hs, ds = self.host[slot], self.dev[slot]
        if self.ready[slot] is not None:
            self.ready[slot].synchronize()       # the previous copy OUT of this slot's pinned staging has finished
        src = {}
        for k, v in batch.items():
            if v.is_pinned():                    # a loader that fills page-locked memory: DMA straight from it
                src[k] = v
            else:                                # pageable -> pinned staging (host memcpy)
                if k not in hs:
                    hs[k] = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                hs[k].copy_(v)
                src[k] = hs[k]
        with torch.cuda.stream(self.copy_stream):
            if self.free[slot] is not None:
                self.copy_stream.wait_event(self.free[slot])
            for k in src:
                ds[k].copy_(src[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.ready[slot] = ev

    def next(self):
        slot = self.i % self.depth
        if self.dev_extra[slot] is StopIteration:
            raise StopIteration
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.ready[slot])
        out = dict(self.dev[slot])
        out.update(self.dev_extra[slot])
        # the slot handed out the call before is free once everything enqueued so far has run
        prev = (self.i - 1) % self.depth
        if self.i > 0:
            fe = torch.cuda.Event()
            fe.record(cur)
            self.free[prev] = fe
        self.i += 1
        self._issue(self.i % self.depth)
        return out

    __call__ = next

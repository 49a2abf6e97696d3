"""CPU oracle for the batch producer (TEST INFRASTRUCTURE ONLY): numpy / scipy restatement of
pase/transforms.py select_chunk (:309-356), norm_and_scale (:148-151), Reverb.__call__ (:1071-1103) with
load_IR's preparation (:1039-1042), SimpleAdditive.__call__ (:1633-1675), SimpleAdditiveShift (:1714-1766),
Clipping (:1514-1535), BandDrop / Downsample (:1162-1196, :1256-1296), one utterance at a time and with the
random decisions passed in.

PINNED: tests/test_transform_pins.py runs the LIVE classes (oracle/ref_shim.install_transforms() stubs the absent
third-party imports of pase/transforms.py) under recorded random draws -- alone and chained through
config_distortions / PCompose -- and requires these functions to reproduce their outputs; the same live outputs
are committed as tests/golden/transforms_live.npz (oracle/live_transforms.py) for the GPU box."""
import numpy as np
import scipy.signal


def select_chunk(wav, T, beg):
    wav = np.asarray(wav, dtype=np.float32)
    if len(wav) <= T:
        P = T - len(wav)
        return np.pad(wav, (0, P), mode="reflect")
    return wav[beg:beg + T]


def norm_and_scale(wav, u):
    return wav / np.max(np.abs(wav)) * np.float32(u)


def prepare_ir(ir, max_reverb_len=24000):
    ir = np.asarray(ir, dtype=np.float64)[:max_reverb_len]
    if np.max(ir) > 0:
        ir = ir / np.abs(np.max(ir))
    return ir, int(np.argmax(np.abs(ir)))


def shift(xs, n):
    e = np.empty_like(xs)
    if n >= 0:
        e[:n] = 0.0
        e[n:] = xs[:-n] if n > 0 else xs     # (the reference's xs[:-0] is empty: n == 0 is defined here as no shift)
    else:
        e[n:] = 0.0
        e[:n] = xs[-n:]
    return e


def reverb(wav, ir, p_max):
    ir = ir.astype(np.float32)
    wav = np.asarray(wav).reshape(-1)
    Ex = np.dot(wav, wav)
    wav = wav.astype(np.float32)
    rev = scipy.signal.convolve(wav, ir, mode="full").reshape(-1)
    Er = np.dot(rev, rev)
    rev = shift(rev, -p_max)
    ratio = np.sqrt(Ex / Er) if Er > 0 else 1.0
    return (ratio * rev[:wav.shape[0]]).astype(np.float32)


def additive(wav, noise_full, n_beg, snr):
    wav = np.asarray(wav, dtype=np.float32).reshape(-1)
    sel = np.asarray(noise_full, dtype=np.float64)
    if len(sel) < len(wav):
        sel = np.concatenate([sel, np.zeros(len(wav) - len(sel))])
    T = len(wav)
    noise = sel[n_beg:n_beg + T].astype(np.float32)
    Ex, En = np.dot(wav, wav), np.dot(noise, noise)
    if not En > 0:
        return wav
    Kf = np.sqrt(Ex / ((10 ** (snr / 10.)) * En))
    noisy = wav + Kf * noise
    return (np.sqrt(Ex / (np.dot(noisy, noisy) + 1e-14)) * noisy).astype(np.float32)


def fir_filter_distortion(wav, filt):
    """BandDrop / Downsample.__call__ (pase/transforms.py:1162-1196, 1256-1296)."""
    filt = np.asarray(filt, dtype=np.float64)
    filt = (filt / np.abs(np.max(filt))).astype(np.float32)
    wav = np.asarray(wav).reshape(-1)
    Ex = np.dot(wav, wav)
    wav = wav.astype(np.float32)
    sig = scipy.signal.convolve(wav, filt, mode="full").reshape(-1)
    sig = shift(sig, -round(filt.shape[0] / 2))
    sig = sig[:wav.shape[0]]
    Ef = np.dot(sig, sig)
    ratio = np.sqrt(Ex / Ef) if Ef > 0 else 1.0
    return (ratio * sig).astype(np.float32)


def clipping(wav, cf):
    wav = np.asarray(wav, dtype=np.float32)
    clip = np.maximum(wav, cf * np.min(wav))
    return np.minimum(clip, cf * np.max(wav))


def overlap(wav, speech, n_beg, shift_n, snr, ir=None, p_max=0):
    """SimpleAdditiveShift.__call__ (pase/transforms.py:1714-1766) with the draws passed in."""
    wav = np.asarray(wav, dtype=np.float32).reshape(-1)
    T = len(wav) - shift_n
    sel = np.asarray(speech, dtype=np.float64)
    if len(sel) < T:
        sel = np.concatenate([sel, np.zeros(T - len(sel))])
        n_beg = 0
    noise = sel[n_beg:n_beg + T].astype(np.float32)
    if ir is not None:
        noise = reverb(noise, ir, p_max)
    pad_len = len(wav) - len(noise)
    noise = np.concatenate([np.zeros(pad_len, dtype=np.float32), noise])
    Ex, En = np.dot(wav, wav), np.dot(noise, noise)
    Kf = np.sqrt(Ex / ((10 ** (snr / 10.)) * En)) if En > 0 else 1.0
    noisy = wav + Kf * noise
    return (np.sqrt(Ex / (np.dot(noisy, noisy) + 1e-14)) * noisy).astype(np.float32)


def overlap_label(T, shift_n, hop):
    """The 'overlap' label SimpleAdditiveShift writes (pase/transforms.py:1738-1748): per-sample mask
    [zeros(pad_len) | ones(len(noise))] with pad_len = shift, averaged over hop-sized blocks."""
    mask = np.concatenate([np.zeros(shift_n, dtype=np.float32), np.ones(T - shift_n, dtype=np.float32)])
    return mask[:(T // hop) * hop].reshape(-1, hop).mean(1)

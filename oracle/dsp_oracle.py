"""CPU oracle for the regression-target transforms (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

numpy/scipy restatement of pase/transforms.py LPS (:439-487), FBanks (:489-548), MFCC (:671-722) and
ZNorm (:183-205) on one utterance, following the reference's own order of operations.

Pinning status:
  * LPS and ZNorm: PINNED against the LIVE pase.transforms.LPS / ZNorm classes (tests/test_transform_pins.py;
    oracle/ref_shim.py adapts the pre-complex torch.stft calling convention the reference uses, which torch 2.x
    rejects, and routes the absent librosa 0.6.3 `feature.delta` to the scipy call it forwards to); the STFT is
    additionally pinned against this image's torch.stft -- tests/test_oracle_pins.py.
  * deltas: scipy.signal.savgol_filter IS what librosa 0.6.3 feature.delta calls
    (width=9, polyorder=order, deriv=order, mode='interp'); scipy is installed, so it is used directly.
  * DCT: scipy.fftpack.dct(type=2, norm='ortho') used directly (what librosa.feature.mfcc calls).
  * python_speech_features 0.6 (logfbank) and librosa 0.6.3 (stft/mel/power_to_db) are pinned
    dependencies of the reference that are NOT installed in this image (requirements.txt:3,7):
    their published algorithms are restated below -- PARITY UNPINNED for the mel filter banks,
    psf framing/pre-emphasis and power_to_db.
"""
import math

import numpy as np
import scipy.fftpack
import scipy.signal


def delta(X, order):
    """librosa.feature.delta(X, order=order): width 9, axis -1, mode 'interp'."""
    return scipy.signal.savgol_filter(X, 9, deriv=order, polyorder=order, axis=-1, mode="interp")


def with_deltas(X, der_order):
    if der_order <= 0:
        return X
    return np.concatenate([X] + [delta(X, n) for n in range(1, der_order + 1)])


def stft_rect(wav, n_fft, hop, win):
    """Legacy torch.stft(wav, n_fft, hop, win): window=None -> ones(win) zero-padded to n_fft centred;
    center=True with reflect padding of n_fft//2; onesided.  Returns complex (n_fft/2+1, frames)."""
    x = np.pad(np.asarray(wav, dtype=np.float64), n_fft // 2, mode="reflect")
    nfr = 1 + (len(x) - n_fft) // hop
    left = (n_fft - win) // 2
    w = np.zeros(n_fft)
    w[left:left + win] = 1.0
    frames = np.stack([x[t * hop:t * hop + n_fft] * w for t in range(nfr)], axis=1)
    return np.fft.rfft(frames, axis=0)


def lps(wav, n_fft=2048, hop=160, win=400, der_order=2):
    max_frames = len(wav) // hop
    X = np.abs(stft_rect(wav, n_fft, hop, win))[:, :max_frames]
    X = 10 * np.log10(X ** 2 + 10e-20)
    return with_deltas(X.astype(np.float32), der_order)


def psf_get_filterbanks(nfilt, nfft, samplerate):
    """python_speech_features.base.get_filterbanks (HTK mel, floor()-quantised bin edges)."""
    hz2mel = lambda hz: 2595 * np.log10(1 + hz / 700.)
    mel2hz = lambda mel: 700 * (10 ** (mel / 2595.0) - 1)
    melpoints = np.linspace(hz2mel(0), hz2mel(samplerate / 2), nfilt + 2)
    bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)
    fbank = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(0, nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fbank[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fbank[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fbank


def psf_logfbank(signal, samplerate, winlen, winstep, nfilt, nfft, preemph=0.97):
    """python_speech_features.base.logfbank -> fbank -> sigproc.{preemphasis,framesig,powspec}."""
    signal = np.asarray(signal, dtype=np.float64)
    signal = np.append(signal[0], signal[1:] - preemph * signal[:-1])
    # sigproc.round_half_up on the lengths
    frame_len = int(math.floor(winlen * samplerate + 0.5))
    frame_step = int(math.floor(winstep * samplerate + 0.5))
    slen = len(signal)
    numframes = 1 if slen <= frame_len else 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))
    padlen = int((numframes - 1) * frame_step + frame_len)
    padsignal = np.concatenate((signal, np.zeros((padlen - slen,))))
    idx = np.arange(frame_len)[None, :] + (np.arange(numframes) * frame_step)[:, None]
    frames = padsignal[idx]                                  # rectangular window (winfunc = ones)
    pspec = 1.0 / nfft * np.square(np.absolute(np.fft.rfft(frames, nfft)))
    feat = np.dot(pspec, psf_get_filterbanks(nfilt, nfft, samplerate).T)
    feat = np.where(feat == 0, np.finfo(float).eps, feat)
    return np.log(feat)


def fbanks(wav, n_filters=40, n_fft=512, hop=160, win=400, rate=16000, der_order=2):
    X = psf_logfbank(wav, rate, float(win) / rate, float(hop) / rate, n_filters, n_fft).T
    expected = len(wav) // hop
    X = with_deltas(X, der_order).astype(np.float32)
    if X.shape[1] < expected:
        X = np.concatenate([X, np.repeat(X[:, -1:], expected - X.shape[1], axis=1)], axis=1)
    return X


def librosa_mel(sr, n_fft, n_mels=128):
    """librosa.filters.mel(sr, n_fft, n_mels=128, fmin=0, fmax=sr/2, htk=False, norm=1) (0.6.3)."""
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = float(f)
        return min_log_mel + np.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        out = f_sp * m
        log_t = m >= min_log_mel
        out[log_t] = min_log_hz * np.exp(logstep * (m[log_t] - min_log_mel))
        return out

    fftfreqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def librosa_mfcc(y, sr, n_mfcc, n_fft, hop_length):
    """librosa.feature.mfcc -> melspectrogram(power=2) -> stft(hann periodic, center, reflect) ->
    power_to_db(ref=1, amin=1e-10, top_db=80) -> dct(type 2, ortho)[:n_mfcc]."""
    y = np.asarray(y, dtype=np.float64)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    nfr = 1 + (len(yp) - n_fft) // hop_length
    win = scipy.signal.get_window("hann", n_fft, fftbins=True)
    frames = np.stack([yp[t * hop_length:t * hop_length + n_fft] * win for t in range(nfr)], axis=1)
    S = np.abs(np.fft.rfft(frames, axis=0)) ** 2
    mel = np.dot(librosa_mel(sr, n_fft), S)
    log_spec = 10.0 * np.log10(np.maximum(1e-10, mel))
    log_spec -= 10.0 * np.log10(np.maximum(1e-10, 1.0))
    log_spec = np.maximum(log_spec, log_spec.max() - 80.0)
    return scipy.fftpack.dct(log_spec, axis=0, type=2, norm="ortho")[:n_mfcc]


def mfcc(wav, hop=160, order=13, win=400, der_order=2):
    max_frames = len(wav) // hop
    m = librosa_mfcc(wav, 16000, order, win, hop)[:, :max_frames]
    return with_deltas(m, der_order).astype(np.float32)


def znorm(X, mean, std):
    return (X - np.asarray(mean)[:, None]) / np.asarray(std)[:, None]


# ---- Gammatone (pase/transforms.py:550-613; gammatone package absent: restated, PARITY UNPINNED) -------------
def gt_erb_space(low_freq, high_freq, num):
    """gammatone.filters.erb_space (centre frequencies from high to low)."""
    ear_q, min_bw = 9.26449, 24.7
    return (-ear_q * min_bw + np.exp((np.arange(1, num + 1) / num) * (-np.log(high_freq + ear_q * min_bw)
                                                                     + np.log(low_freq + ear_q * min_bw)))
            * (high_freq + ear_q * min_bw))


def gt_make_erb_filters(fs, centre_freqs, width=1.0):
    """gammatone.filters.make_erb_filters."""
    T = 1 / fs
    ear_q, min_bw, order = 9.26449, 24.7, 1
    erb = width * ((centre_freqs / ear_q) ** order + min_bw ** order) ** (1 / order)
    B = 1.019 * 2 * np.pi * erb
    arg = 2 * centre_freqs * np.pi * T
    vec = np.exp(2j * arg)
    A0, A2, B0 = T, 0, 1
    B1 = -2 * np.cos(arg) / np.exp(B * T)
    B2 = np.exp(-2 * B * T)
    rt_pos, rt_neg = np.sqrt(3 + 2 ** 1.5), np.sqrt(3 - 2 ** 1.5)
    common = -T * np.exp(-(B * T))
    k11 = np.cos(arg) + rt_pos * np.sin(arg)
    k12 = np.cos(arg) - rt_pos * np.sin(arg)
    k13 = np.cos(arg) + rt_neg * np.sin(arg)
    k14 = np.cos(arg) - rt_neg * np.sin(arg)
    A11, A12, A13, A14 = common * k11, common * k12, common * k13, common * k14
    gain_arg = np.exp(1j * arg - B * T)
    gain = np.abs((vec - gain_arg * k11) * (vec - gain_arg * k12) * (vec - gain_arg * k13) * (vec - gain_arg * k14)
                  * (T * np.exp(B * T) / (-1 / np.exp(B * T) + 1 + vec * (1 - np.exp(B * T)))) ** 4)
    allfilts = np.ones_like(centre_freqs)
    return np.column_stack([A0 * allfilts, A11, A12, A13, A14, A2 * allfilts, B0 * allfilts, B1, B2, gain])


def gt_erb_filterbank(wave, coefs):
    """gammatone.filters.erb_filterbank: four scipy.signal.lfilter passes per channel, then / gain."""
    output = np.zeros((coefs.shape[0], wave.shape[0]))
    gain = coefs[:, 9]
    As = [coefs[:, (0, k, 5)] for k in (1, 2, 3, 4)]
    Bs = coefs[:, 6:9]
    for idx in range(coefs.shape[0]):
        y = wave
        for A in As:
            y = scipy.signal.lfilter(A[idx], Bs[idx], y)
        output[idx, :] = y / gain[idx]
    return output


def gtgram(wave, fs, window_time, hop_time, channels, f_min):
    """gammatone.gtgram.gtgram: sqrt of the mean squared filter output per window."""
    xe = np.power(gt_erb_filterbank(np.asarray(wave, dtype=np.float64),
                                    gt_make_erb_filters(fs, gt_erb_space(f_min, fs / 2, channels))), 2)
    nwin = int(math.floor(window_time * fs + 0.5))
    hop_samples = int(math.floor(hop_time * fs + 0.5))
    columns = 1 + int(np.floor((xe.shape[1] - nwin) / hop_samples))
    y = np.zeros((channels, columns))
    for c in range(columns):
        y[:, c] = np.sqrt(xe[:, c * hop_samples + np.arange(nwin)].mean(axis=1))
    return y


def gammatone(wav, f_min=500, n_channels=40, hop=160, win=400, der_order=2, rate=16000):
    gtn = np.log(gtgram(np.asarray(wav, dtype=np.float32), rate, float(win) / rate, float(hop) / rate, n_channels,
                        f_min) + 1e-10)
    gtn = with_deltas(gtn, der_order).astype(np.float32)
    expected = len(wav) // hop
    if gtn.shape[1] < expected:
        gtn = np.concatenate([gtn, np.repeat(gtn[:, -1:], expected - gtn.shape[1], axis=1)], axis=1)
    return gtn


# ---------------------------------------------------------------------------------------------
# Prosody (pase/transforms.py:919-999): rows [lf0, uv, egy, zcr] + deltas.
# PARITY UNPINNED for every third-party piece (none is installed here): pysptk.swipe (the f0 tracker; f0 is an INPUT
# of these functions), ahoproc_tools.interpolate.interpolation (un-pinned git dependency, requirements.txt:17;
# restated from the published package, its quirk of flagging the last voiced frame before a gap as unvoiced kept),
# librosa 0.6.3 feature.rmse / zero_crossing_rate / util.frame (restated from the published source).
# ---------------------------------------------------------------------------------------------
def ahoproc_interpolation(signal, unvoiced_symbol):
    """ahoproc_tools.interpolate.interpolation: returns (interpolated signal, voiced flag)."""
    signal = np.asarray(signal, dtype=np.float64)
    tb = [None, None]
    fb = [None, None]
    prev = signal[0]
    isig = signal.copy()
    uv = np.ones(signal.shape, dtype=np.int8)
    for t in range(1, signal.shape[0]):
        if signal[t] > unvoiced_symbol and prev <= unvoiced_symbol and tb == [None, None]:
            isig[:t] = signal[t]                      # leading unvoiced stretch: first voiced value
            uv[:t] = 0
        elif signal[t] <= unvoiced_symbol and prev > unvoiced_symbol:
            tb[0], fb[0] = t - 1, prev
        elif signal[t] > unvoiced_symbol and prev <= unvoiced_symbol:
            tb[1], fb[1] = t, signal[t]
            n = tb[1] - tb[0]
            isig[tb[0]:tb[1]] = fb[0] + np.arange(n) * ((fb[1] - fb[0]) / n)
            uv[tb[0]:tb[1]] = 0
            tb, fb = [None, None], [None, None]
        prev = signal[t]
    if tb[0] is not None:
        isig[tb[0]:] = fb[0]                          # trailing unvoiced stretch: last voiced value
        uv[tb[0]:] = 0
    if np.all(isig <= unvoiced_symbol):
        uv = np.zeros(signal.shape, dtype=np.int8)
    return isig, uv


def librosa_frame(y, frame_length, hop_length):
    n = 1 + (len(y) - frame_length) // hop_length
    return np.stack([y[i * hop_length:i * hop_length + frame_length] for i in range(n)], 1)     # (frame_length, n)


def librosa_rmse(y, frame_length, hop_length, pad_mode="reflect"):
    y = np.pad(np.asarray(y), int(frame_length // 2), mode=pad_mode)
    x = librosa_frame(y, frame_length, hop_length)
    return np.sqrt(np.mean(np.abs(x) ** 2, axis=0, keepdims=True))


def librosa_zero_crossing_rate(y, frame_length, hop_length, threshold=1e-10):
    y = np.pad(np.asarray(y), int(frame_length // 2), mode="edge")
    x = librosa_frame(y, frame_length, hop_length).copy()
    x[np.abs(x) <= threshold] = 0
    sign = np.signbit(x)
    cross = np.concatenate([np.zeros((1, x.shape[1]), dtype=bool), sign[1:] != sign[:-1]], 0)    # pad=False
    return np.mean(cross, axis=0, keepdims=True)


def prosody(wav, f0, hop=160, win=320, f0_min=60, f0_max=300, der_order=2):
    """Prosody.__call__ (transforms.py:931-990) given the tracker's f0 contour (Hz, 0 = unvoiced)."""
    wav = np.asarray(wav)
    max_frames = wav.shape[0] // hop
    f0 = np.asarray(f0, dtype=np.float64)
    if len(f0) < max_frames:
        pad = max_frames - len(f0)
        f0 = np.concatenate((f0, f0[-pad:]), axis=0)
    lf0 = np.log(f0 + 1e-10)
    lf0, uv = ahoproc_interpolation(lf0, -1)
    lf0 = lf0.astype(np.float32)[None, :max_frames]
    uv = uv.astype(np.float32)[None, :max_frames]
    if uv.sum() == 0:
        lf0 = np.ones_like(uv) * np.float32(np.log(f0_min))
    zcr = librosa_zero_crossing_rate(wav, win, hop).astype(np.float32)[:, :max_frames]
    egy = librosa_rmse(wav, win, hop, pad_mode="constant").astype(np.float32)[:, :max_frames]
    return with_deltas(np.concatenate((lf0, uv, egy, zcr), 0), der_order)

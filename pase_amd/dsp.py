"""On-device regression targets: mirrors of the host transforms LPS / FBanks / MFCC (+ deltas, ZNorm)
of pase/transforms.py (:439-487, :489-548, :671-722, :183-205), batched on the GPU.

`pase_frame_prep` lays the (padded, optionally pre-emphasised) waveform out hop-major, which turns the
hop-strided framing into a stride-1 conv with `hop` input channels; every spectrum then is ONE
`pase_conv_gemm` launch with a DFT basis as the weight (window, centring and -- for MFCC -- the Hann
taper folded into the basis) and a power / log-power post-op in the epilogue; mel and DCT projections
are 1x1 launches; deltas + ZNorm are one `pase_delta_znorm` launch.  Nothing runs on the
host per step: the basis matrices are built once with numpy.

Third-party arithmetic the reference delegates to (not installed here; restated from the published
algorithms, parity UNPINNED -- SURVEY.md section 8c): legacy torch.stft (rectangular window centred in
n_fft, reflect centre padding) for LPS; python_speech_features 0.6 `logfbank` for FBanks; librosa 0.6.3
`feature.mfcc` (periodic Hann, Slaney mel bank with area normalisation, power_to_db top_db=80, DCT-II
ortho) and `feature.delta` (Savitzky-Golay width 9) for MFCC / deltas; detly/gammatone `gtgram` for Gammatone;
pysptk.swipe (SWIPE' f0), ahoproc_tools `interpolation` and librosa `rmse` / `zero_crossing_rate` for Prosody.
"""
import math

import numpy as np
import torch

from . import kernels as K


def _dft_basis(n_fft, taps, offset, window=None):
    """(2*(n_fft/2+1), taps): rows (2f, 2f+1) = window[n] * (cos, -sin)(2 pi f (n+offset) / n_fft)."""
    f = np.arange(n_fft // 2 + 1, dtype=np.float64)[:, None]
    n = np.arange(taps, dtype=np.float64)[None, :] + offset
    ang = 2.0 * np.pi * f * n / n_fft
    w = np.ones(taps) if window is None else np.asarray(window, dtype=np.float64)
    basis = np.empty((2 * (n_fft // 2 + 1), taps), dtype=np.float64)
    basis[0::2] = np.cos(ang) * w[None, :]
    basis[1::2] = -np.sin(ang) * w[None, :]
    return basis.astype(np.float32)


def _hop_major(basis, hop):
    """(M, win) frame basis -> (M, hop * taps) weight of the stride-1 conv over the hop-major layout:
    W[f, r, dq] = basis[f, dq*hop + r] (zero past the window)."""
    M, win = basis.shape
    taps = (win + hop - 1) // hop
    w = np.zeros((M, taps * hop), dtype=np.float32)
    w[:, :win] = basis
    return np.ascontiguousarray(w.reshape(M, taps, hop).transpose(0, 2, 1)).reshape(M, hop * taps), taps


def _spectrum(wav, weight, taps, hop, nframes, padL, pad_mode, preemph, post_op, post_scale, post_eps=0.0):
    """frames x DFT basis -> (B, bins, nframes) power / log-power spectrum."""
    B, _, T = wav.shape
    Q = nframes + taps - 1
    xq = torch.empty(B, hop, Q, device=wav.device)
    K.frame_prep(wav, xq, B=B, T=T, hop=hop, Q=Q, padL=padL, pad_mode=pad_mode, preemph=preemph)
    bins = weight.shape[0] // 2
    out = torch.empty(B, bins, nframes, device=wav.device)
    K.conv_gemm(xq, weight, out, S=B, Cin=hop, Tin=Q, M=2 * bins, K=hop * taps, taps=taps, Ncols=nframes,
                Tout=nframes, Cout_store=bins, y_ctot=bins, post_op=post_op, post_scale=post_scale,
                post_eps=post_eps, splitk=1)
    return out


def savgol_delta_coefs(order_max=2, width=9):
    """(order_max+1, 9, 9) table for pase_delta_znorm.  With polyorder == deriv == k the k-th derivative
    of the least-squares polynomial is constant over the window, so scipy's mode='interp' edge fit
    uses the same 9 weights as the interior filter (shifted window): every pos row is identical."""
    j = np.arange(width, dtype=np.float64) - (width - 1) / 2.0
    tab = np.zeros((order_max + 1, width, width), dtype=np.float64)
    for k in range(1, order_max + 1):
        A = np.vander(j, k + 1, increasing=True)          # (9, k+1)
        coef = np.linalg.pinv(A)[k] * math.factorial(k)    # d^k/dt^k of the fit = k! * a_k
        tab[k, :, :] = coef[None, :]
    return tab.astype(np.float32)


def psf_mel_filterbank(nfilt, nfft, sr, lowfreq=0.0, highfreq=None):
    """python_speech_features.base.get_filterbanks."""
    highfreq = highfreq or sr / 2
    hz2mel = lambda hz: 2595.0 * np.log10(1 + hz / 700.0)
    mel2hz = lambda mel: 700.0 * (10 ** (mel / 2595.0) - 1)
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    b = np.floor((nfft + 1) * mel2hz(melpoints) / sr)
    fb = np.zeros([nfilt, nfft // 2 + 1])
    for jj in range(nfilt):
        for i in range(int(b[jj]), int(b[jj + 1])):
            fb[jj, i] = (i - b[jj]) / (b[jj + 1] - b[jj])
        for i in range(int(b[jj + 1]), int(b[jj + 2])):
            fb[jj, i] = (b[jj + 2] - i) / (b[jj + 2] - b[jj + 1])
    return fb.astype(np.float32)


def slaney_mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=1)."""
    fmax = fmax or sr / 2.0

    def hz_to_mel(f):
        f = np.asanyarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
        min_log_mel = min_log_hz / f_sp
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asanyarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
        min_log_mel = min_log_hz / f_sp
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


def dct2_ortho(n_out, n_in):
    """rows of scipy.fftpack.dct(type=2, norm='ortho') restricted to the first n_out coefficients."""
    k = np.arange(n_out, dtype=np.float64)[:, None]
    n = np.arange(n_in, dtype=np.float64)[None, :]
    m = 2.0 * np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in))
    m[0] *= math.sqrt(1.0 / (4 * n_in))
    m[1:] *= math.sqrt(1.0 / (2 * n_in))
    return m.astype(np.float32)


class _Feature(object):
    def __init__(self, name, der_order, device):
        self.name = name
        self.der_order = der_order
        self.device = torch.device(device)
        self.coef = torch.from_numpy(savgol_delta_coefs(2)).to(self.device)
        self.mean = self.istd = None

    def set_stats(self, mean, std):
        """ZNorm (transforms.py:183-205): per output channel (x - mean) / std, fused into the delta kernel."""
        self.mean = torch.as_tensor(mean, dtype=torch.float32).reshape(-1).contiguous().to(self.device)
        self.istd = (1.0 / torch.as_tensor(std, dtype=torch.float32).reshape(-1)).contiguous().to(self.device)

    def _finish(self, base, F, Fo):
        B, D = base.shape[0], base.shape[1]
        out = torch.empty(B, (self.der_order + 1) * D, Fo, device=base.device)
        K.delta_znorm(base, self.coef, self.mean, self.istd, out, B=B, D=D, F=F, Fo=Fo, order=self.der_order,
                      x_ctot=D, x_coff=0)
        return out


class LPS(_Feature):
    """10 log10(|STFT|^2 + 1e-19) with the legacy torch.stft(wav, n_fft, hop, win) semantics."""

    def __init__(self, n_fft=2048, hop=160, win=400, der_order=2, name="lps", device="cuda"):
        super().__init__(name, der_order, device)
        self.n_fft, self.hop, self.win = n_fft, hop, win
        off = (n_fft - win) // 2
        self.padL = n_fft // 2 - off
        w, self.taps = _hop_major(_dft_basis(n_fft, win, off), hop)
        self.basis = torch.from_numpy(w).to(self.device)

    def __call__(self, wav):
        """wav (B, 1, T) -> (B, (der_order+1)*(n_fft/2+1), T//hop)"""
        B, _, T = wav.shape
        F = T // self.hop
        lps = _spectrum(wav, self.basis, self.taps, self.hop, F, self.padL, K.PAD_REFLECT, 0.0, K.POST_LOGPOW,
                        10.0 / math.log(10.0), 10e-20)
        return self._finish(lps, F, F)


class FBanks(_Feature):
    """python_speech_features.logfbank (pre-emphasis 0.97, rectangular frames, |rfft|^2 / nfft, HTK mel
    triangles, log) -> deltas -> replicate-pad to T//hop frames."""

    def __init__(self, n_filters=40, n_fft=512, hop=160, win=400, rate=16000, der_order=2, name="fbank",
                 device="cuda"):
        super().__init__(name, der_order, device)
        self.n_filters, self.n_fft, self.hop, self.win, self.rate = n_filters, n_fft, hop, win, rate
        # numpy.fft.rfft(frames, NFFT) truncates longer frames
        w, self.taps = _hop_major(_dft_basis(n_fft, min(win, n_fft), 0), hop)
        self.basis = torch.from_numpy(w).to(self.device)
        self.mel = torch.from_numpy(psf_mel_filterbank(n_filters, n_fft, rate)).to(self.device)

    def __call__(self, wav):
        B, _, T = wav.shape
        L, st = self.win, self.hop
        nf = 1 if T <= L else 1 + int(math.ceil((1.0 * T - L) / st))
        Fo = T // st
        bins = self.n_fft // 2 + 1
        pspec = _spectrum(wav, self.basis, self.taps, st, nf, 0, K.PAD_ZERO, 0.97, K.POST_POW, 1.0 / self.n_fft)
        feat = torch.empty(B, self.n_filters, nf, device=wav.device)
        K.conv_gemm(pspec, self.mel, feat, S=B, Cin=bins, Tin=nf, M=self.n_filters, K=bins, taps=1, Ncols=nf, Tout=nf,
                    post_op=K.POST_LOG, post_scale=1.0, post_eps=float(np.finfo(float).eps), splitk=1)
        return self._finish(feat, nf, max(Fo, nf))


class MFCC(_Feature):
    """librosa.feature.mfcc(y, sr, n_mfcc=order, n_fft=win, hop_length=hop)[:, :T//hop] -> deltas."""

    def __init__(self, n_fft=2048, hop=160, order=13, sr=16000, win=400, der_order=2, name="mfcc", device="cuda"):
        super().__init__(name, der_order, device)
        self.n_fft, self.hop, self.order, self.sr = win, hop, order, 16000    # (sic) n_fft := win, :679-683
        n = self.n_fft
        hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)            # periodic ('fftbins') Hann
        w, self.taps = _hop_major(_dft_basis(n, n, 0, hann), hop)
        self.basis = torch.from_numpy(w).to(self.device)
        self.mel = torch.from_numpy(slaney_mel_filterbank(self.sr, n, 128)).to(self.device)
        self.dct = torch.from_numpy(dct2_ortho(order, 128)).to(self.device)

    def __call__(self, wav):
        B, _, T = wav.shape
        n, st = self.n_fft, self.hop
        nf = 1 + T // st                       # centred STFT
        F = T // st
        bins = n // 2 + 1
        power = _spectrum(wav, self.basis, self.taps, st, nf, n // 2, K.PAD_REFLECT, 0.0, K.POST_POW, 1.0)
        mel = torch.empty(B, 128, nf, device=wav.device)
        K.conv_gemm(power, self.mel, mel, S=B, Cin=bins, Tin=nf, M=128, K=bins, taps=1, Ncols=nf, Tout=nf, splitk=1)
        db = torch.empty_like(mel)
        umax = torch.empty(B, dtype=torch.int32, device=wav.device)
        K.power_to_db(mel, db, umax, per_utt=128 * nf, B=B, amin=1e-10, ref_db=0.0, top_db=80.0)
        mfcc = torch.empty(B, self.order, F, device=wav.device)
        K.conv_gemm(db, self.dct, mfcc, S=B, Cin=128, Tin=nf, M=self.order, K=128, taps=1, Ncols=F, Tout=F, splitk=1)
        return self._finish(mfcc, F, F)


def erb_space(low_freq, high_freq, num):
    """gammatone.filters.erb_space: `num` centre frequencies from high_freq down to low_freq on the ERB scale."""
    ear_q, min_bw = 9.26449, 24.7
    frac = np.arange(1, num + 1) / float(num)
    return -ear_q * min_bw + np.exp(frac * (-np.log(high_freq + ear_q * min_bw) + np.log(low_freq + ear_q * min_bw))) * (
        high_freq + ear_q * min_bw)


def make_erb_filters(fs, centre_freqs, width=1.0):
    """gammatone.filters.make_erb_filters (Slaney 1993, Auditory Toolbox MakeERBFilters): rows
    [A0, A11, A12, A13, A14, A2, B0, B1, B2, gain] of the four cascaded second-order sections."""
    T = 1.0 / fs
    ear_q, min_bw, order = 9.26449, 24.7, 1
    cf = np.asarray(centre_freqs, dtype=np.float64)
    erb = width * ((cf / ear_q) ** order + min_bw ** order) ** (1.0 / order)
    B = 1.019 * 2 * np.pi * erb
    arg = 2 * cf * np.pi * T
    vec = np.exp(2j * arg)
    A0, A2, B0 = T, 0.0, 1.0
    B1 = -2 * np.cos(arg) / np.exp(B * T)
    B2 = np.exp(-2 * B * T)
    rt_pos, rt_neg = np.sqrt(3 + 2 ** 1.5), np.sqrt(3 - 2 ** 1.5)
    common = -T * np.exp(-(B * T))
    k11 = np.cos(arg) + rt_pos * np.sin(arg)
    k12 = np.cos(arg) - rt_pos * np.sin(arg)
    k13 = np.cos(arg) + rt_neg * np.sin(arg)
    k14 = np.cos(arg) - rt_neg * np.sin(arg)
    gain_arg = np.exp(1j * arg - B * T)
    gain = np.abs((vec - gain_arg * k11) * (vec - gain_arg * k12) * (vec - gain_arg * k13) * (vec - gain_arg * k14)
                  * (T * np.exp(B * T) / (-1 / np.exp(B * T) + 1 + vec * (1 - np.exp(B * T)))) ** 4)
    ones = np.ones_like(cf)
    return np.column_stack([A0 * ones, common * k11, common * k12, common * k13, common * k14, A2 * ones, B0 * ones,
                            B1, B2, gain])


class Gammatone(_Feature):
    """gammatone.gtgram.gtgram(wav, rate, win/rate, hop/rate, n_channels, f_min) -> log(. + 1e-10) -> deltas ->
    replicate-pad (pase/transforms.py:550-613).  Third-party arithmetic restated from the published Slaney ERB
    filter design (package absent: parity unpinned)."""

    def __init__(self, f_min=500, n_channels=40, hop=160, win=400, der_order=2, rate=16000, name="gtn", device="cuda"):
        super().__init__(name, der_order, device)
        self.hop, self.win, self.C, self.rate = hop, win, n_channels, rate
        self.erb = torch.from_numpy(make_erb_filters(rate, erb_space(f_min, rate / 2.0, n_channels))).to(self.device)
        self.g = math.gcd(win, hop)

    def __call__(self, wav, blocks=None, g=None):
        """`blocks` / `g`: block sums shared with another Gammatone of the same filterbank (DeviceTargets)."""
        B, _, T = wav.shape
        if blocks is None:
            g = self.g
            blocks = self.block_sums(wav, g)
        ncol = 1 + (T - self.win) // self.hop
        gt = torch.empty(B, self.C, ncol, device=wav.device)
        K.gammatone_frames(blocks, gt, rows=B * self.C, T=T, g=g, nwin=self.win, hop=self.hop, ncol=ncol, eps=1e-10)
        return self._finish(gt, ncol, max(T // self.hop, ncol))

    def block_sums(self, wav, g):
        B, _, T = wav.shape
        blocks = torch.empty(B * self.C, (T + g - 1) // g, device=wav.device)
        K.gammatone_blocks(wav, self.erb, blocks, B=B, C_=self.C, T=T, g=g)
        return blocks


# ---------------------------------------------------------------------------------------------------------------
# SWIPE' f0 tracker (A. Camacho 2007/2008; what pysptk.swipe -- SPTK's swipe.c -- computes for the Prosody target,
# pase/transforms.py:948-952).  PARITY UNPINNED: the third-party tracker is not installed; the published algorithm
# (swipep.m) with swipe.c's constants is implemented: 1/96-octave candidates, ERB-scale sqrt-magnitude loudness every
# 0.1 ERB, Hann windows of the power-of-two sizes nearest 8 periods at 50 % overlap, prime-harmonic cosine kernels,
# linear interpolation in time, lambda-weighted combination of window sizes, parabolic refinement on a 1/768-octave
# grid, strength threshold 0.3, unvoiced frames reported as f0 = 0.
# Device mapping: per window size the spectrogram is ONE DFT-basis pase_conv_gemm launch (Hann window folded into
# the basis, |.| post-op), the natural-cubic-spline resampling to the ERB frequencies and the kernel / tail-energy
# inner products are 1x1 pase_conv_gemm launches against host-built matrices; pase_swipe_accumulate / _pick finish.
# ---------------------------------------------------------------------------------------------------------------
def _primes_upto(n):
    return [q for q in range(2, n + 1) if all(q % r for r in range(2, int(q ** 0.5) + 1))]


def natural_spline_matrix(n, xq):
    """(len(xq), n) matrix of natural-cubic-spline interpolation from the uniform grid 0..n-1 (in grid units) to the
    query abscissae xq; rows of queries outside [0, n-1] are zero (interp1(..., 'spline', 0))."""
    A = np.zeros((n, n))
    Bm = np.zeros((n, n))
    A[0, 0] = A[n - 1, n - 1] = 1.0                       # natural ends: second derivative 0
    for i in range(1, n - 1):
        A[i, i - 1], A[i, i], A[i, i + 1] = 1.0, 4.0, 1.0
        Bm[i, i - 1], Bm[i, i], Bm[i, i + 1] = 6.0, -12.0, 6.0
    M2 = np.linalg.solve(A, Bm)                           # second derivatives as a linear map of the samples
    xq = np.asarray(xq, dtype=np.float64)
    E = np.zeros((len(xq), n))
    inside = (xq >= 0) & (xq <= n - 1)
    i0 = np.clip(np.floor(xq).astype(int), 0, n - 2)
    u = xq - i0                                           # in [0, 1]
    eye = np.eye(n)
    for r in np.where(inside)[0]:
        i, t = i0[r], u[r]
        a, b = 1.0 - t, t
        E[r] = a * eye[i] + b * eye[i + 1] + ((a ** 3 - a) * M2[i] + (b ** 3 - b) * M2[i + 1]) / 6.0
    return E


class SwipeTracker(object):
    """f0 = SwipeTracker(...)(wav): wav (B, 1, T) -> (B, T // hop + 1) Hz, 0 on unvoiced frames."""

    DLOG2P, DERBS, POLYV = 1.0 / 96.0, 0.1, 1.0 / 12.0 / 64.0

    def __init__(self, fs=16000, hop=160, f0_min=60.0, f0_max=300.0, threshold=0.3, device="cuda"):
        self.fs, self.hop, self.fmin, self.fmax, self.st = fs, hop, float(f0_min), float(f0_max), float(threshold)
        self.device = torch.device(device)
        log2pc = np.arange(np.log2(self.fmin), np.log2(self.fmax), self.DLOG2P)
        pc = 2.0 ** log2pc
        self.NC = len(pc)
        logws = np.round(np.log2(8.0 * fs / np.array([self.fmin, self.fmax]))).astype(int)
        wss = 2 ** np.arange(logws[0], logws[1] - 1, -1)
        pO = 8.0 * fs / wss
        d = 1.0 + log2pc - np.log2(8.0 * fs / wss[0])
        erb = lambda hz: 21.4 * np.log10(1.0 + hz / 229.0)
        ferbs = (10.0 ** (np.arange(erb(pc.min() / 4.0), erb(fs / 2.0), self.DERBS) / 21.4) - 1.0) * 229.0
        dev = self.device
        self.windows = []
        for i, w in enumerate(int(v) for v in wss):
            ii = i + 1
            dn = max(1, int(round(8.0 * 0.5 * fs / pO[i])))
            if len(wss) == 1:
                j, k = np.arange(len(pc)), np.array([], dtype=np.int64)
            elif ii == len(wss):
                j = np.where(d - ii > -1)[0]
                k = np.where(d[j] - ii < 0)[0]
            elif ii == 1:
                j = np.where(d - ii < 1)[0]
                k = np.where(d[j] - ii > 0)[0]
            else:
                j = np.where(np.abs(d - ii) < 1)[0]
                k = np.arange(len(j))
            ferbs = ferbs[int(np.argmax(ferbs > pc[j[0]] / 4.0)):]
            mu = np.ones(len(j))
            mu[k] = 1.0 - np.abs(d[j[k]] - ii)
            E = natural_spline_matrix(w // 2 + 1, ferbs * w / float(fs))
            Kmat = np.zeros((len(j), len(ferbs)))
            tail = np.zeros((len(j), len(ferbs)))
            start = 0
            for r, p_ in enumerate(pc[j]):
                start += int(np.argmax(ferbs[start:] > p_ / 4.0))
                fj = ferbs[start:]
                nh = int(np.fix(fj[-1] / p_ - 0.75))
                if nh == 0:
                    raise ValueError("SWIPE': candidate %.1f Hz has no harmonic below fs/2" % p_)
                q = fj / p_
                kk = np.zeros(len(fj))
                for h in [1] + _primes_upto(nh):
                    a = np.abs(q - h)
                    pk = a < 0.25
                    kk[pk] = np.cos(2.0 * np.pi * q[pk])
                    v = (0.25 < a) & (a < 0.75)
                    kk[v] += np.cos(2.0 * np.pi * q[v]) / 2.0
                kk *= np.sqrt(1.0 / fj)
                kk /= np.linalg.norm(kk[kk > 0])
                Kmat[r, start:] = kk
                tail[r, start:] = 1.0
            n = np.arange(1, w + 1, dtype=np.float64)
            hann = 0.5 * (1.0 - np.cos(2.0 * np.pi * n / (w + 1)))          # MATLAB hanning(w)
            basis, taps = _hop_major(_dft_basis(w, w, 0, window=hann), dn)
            t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
            self.windows.append(dict(ws=w, dn=dn, taps=taps, basis=t32(basis), E=t32(E), K=t32(Kmat), tail=t32(tail),
                                     mu=t32(mu), cand=torch.from_numpy(j.astype(np.int32)).to(dev), nj=len(j),
                                     nerb=len(ferbs), bins=w // 2 + 1))

    def __call__(self, wav, return_strength=False):
        B, _, T = wav.shape
        F = T // self.hop + 1                               # t = 0 : dt : len/fs
        dev = wav.device
        S = torch.zeros(B, self.NC, F, device=dev)
        wav = wav.contiguous()

        def gemm(x, wgt, M, Kd, nfr, post):
            y = torch.empty(B, M, nfr, device=dev)
            K.conv_gemm(x, wgt, y, S=B, Cin=Kd, Tin=nfr, M=M, K=Kd, taps=1, Ncols=nfr, Tout=nfr, post_op=post, splitk=1)
            return y
        for w in self.windows:
            ws, dn = w["ws"], w["dn"]
            nfr = (T + ws + dn - (ws - dn)) // dn
            X = _spectrum(wav, w["basis"], w["taps"], dn, nfr, ws // 2, K.PAD_ZERO, 0.0, K.POST_MAG, 1.0)
            L = gemm(X, w["E"], w["nerb"], w["bins"], nfr, K.POST_SQRTPOS)
            L2 = gemm(X, w["E"], w["nerb"], w["bins"], nfr, K.POST_RELU)
            num = gemm(L, w["K"], w["nj"], w["nerb"], nfr, K.POST_NONE)
            den2 = gemm(L2, w["tail"], w["nj"], w["nerb"], nfr, K.POST_NONE)
            K.swipe_accumulate(num, den2, w["mu"], w["cand"], S, B=B, nj=w["nj"], nfr=nfr, NC=self.NC, F=F,
                               frames_per_out=float(self.hop) / float(dn))
        f0 = torch.empty(B, F, device=dev)
        strength = torch.empty(B, F, device=dev) if return_strength else None
        K.swipe_pick(S, f0, strength, B=B, NC=self.NC, F=F, log2_fmin=float(np.log2(self.fmin)), dlog2p=self.DLOG2P,
                     polyv=self.POLYV, st=self.st)
        return (f0, strength) if return_strength else f0


class Prosody(_Feature):
    """Prosody target (pase/transforms.py:919-999): rows [lf0, uv, energy, zcr] (+ deltas, ZNorm).  The energy and
    zero-crossing rows and the lf0 interpolation / voiced flag run on the device (pase_zcr_rms, pase_lf0_interp);
    the f0 contour itself (Hz per hop, 0 = unvoiced: what pysptk.swipe(..., otype='f0') returns) comes from `f0`
    -- a (B, F) tensor the caller provides, or the device tracker when one is attached (`tracker(wav) -> f0`)."""

    def __init__(self, hop=160, win=320, f0_min=60, f0_max=300, der_order=2, sr=16000, name="prosody", device="cuda",
                 tracker=None):
        super().__init__(name, der_order, device)
        self.hop, self.win, self.f0_min, self.f0_max, self.sr = hop, win, f0_min, f0_max, sr
        self.tracker = tracker

    def __call__(self, wav, f0=None):
        """wav (B, 1, T), f0 (B, >= T//hop) -> (B, 4*(der_order+1), T//hop)"""
        B, _, T = wav.shape
        F = T // self.hop
        if f0 is None:
            if self.tracker is None:
                raise ValueError("pase_amd Prosody: no f0 contour given and no device tracker attached")
            f0 = self.tracker(wav)
        f0 = f0.contiguous().float()
        if f0.shape[1] < F:       # transforms.py:951-953: a short contour repeats its tail
            f0 = torch.cat((f0, f0[:, f0.shape[1] - (F - f0.shape[1]):]), 1).contiguous()
        base = torch.empty(B, 4, F, device=wav.device)
        K.lf0_interp(f0, base, B=B, Fin=f0.shape[1], F=F, out_ctot=4, out_coff=0, f0_min=float(self.f0_min))
        K.zcr_rms(wav.contiguous(), base, B=B, T=T, F=F, hop=self.hop, win=self.win, out_ctot=4, out_coff=2)
        return self._finish(base, F, F)


class DeviceTargets(object):
    """What train.py:make_transforms (train.py:37-136) composes from the worker names -- LPS / FBanks /
    MFCC (+ their *_long variants via the per-worker `transform` kwargs) followed by ZNorm -- as a
    batched on-device producer: targets(cchunk) -> {worker name: (B, D, T//hop)}."""

    def __init__(self, workers_cfg, hop=160, stats=None, device="cuda"):
        self.feats = {}
        for w in workers_cfg.get("regr", []):
            name = w["name"]
            kw = dict(w.get("transform", {}))
            if "lps" in name:
                f = LPS(hop=hop, name=name, device=device, **kw)
            elif "fbank" in name:
                f = FBanks(hop=hop, name=name, device=device, **kw)
            elif "mfcc" in name:
                f = MFCC(hop=hop, name=name, device=device, **kw)
            elif "gtn" in name:
                f = Gammatone(hop=hop, name=name, device=device, **kw)
            elif "prosody" in name:
                f = Prosody(hop=hop, name=name, device=device, **kw)
                f.tracker = SwipeTracker(fs=f.sr, hop=hop, f0_min=f.f0_min, f0_max=f.f0_max, device=device)
            else:
                continue      # cchunk: the waveform itself
            if stats is not None and name in stats:
                f.set_stats(stats[name]["mean"], stats[name]["std"])
            self.feats[name] = f

    def __call__(self, cchunk):
        out = {}
        # Gammatone features that share a filterbank (gtn / gtn_long differ only in the window) share the IIR pass
        gts = [f for f in self.feats.values() if isinstance(f, Gammatone)]
        shared = {}
        if len(gts) > 1 and all((f.C, f.rate) == (gts[0].C, gts[0].rate) and torch.equal(f.erb, gts[0].erb) for f in gts):
            g = 0
            for f in gts:
                g = math.gcd(g, f.g)
            blocks = gts[0].block_sums(cchunk, g)
            shared = {f.name: (blocks, g) for f in gts}
        for name, f in self.feats.items():
            out[name] = f(cchunk, *shared[name]) if name in shared else f(cchunk)
        return out


class TargetStats(object):
    """ZNorm statistics with make_trainset_statistics.py's definition (:96-101): over all collected utterances,
    mean = mean_u(mean_t x), std = std_u(std_t x) (unbiased, torch.std) -- per feature channel.  Accumulates the
    per-utterance moments batch by batch on the device instead of concatenating the features of an epoch."""

    def __init__(self):
        self.means, self.stds = {}, {}

    def update(self, targets):
        for k, v in targets.items():          # v: (B, D, F) un-normalised feature
            self.means.setdefault(k, []).append(v.mean(dim=2))
            self.stds.setdefault(k, []).append(v.std(dim=2))

    def finalize(self):
        stats = {}
        for k in self.means:
            m, sd = torch.cat(self.means[k]), torch.cat(self.stds[k])
            stats[k] = {"mean": m.mean(dim=0).cpu(), "std": sd.std(dim=0).cpu()}
        return stats

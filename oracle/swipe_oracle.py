"""CPU oracle of the f0 tracker behind the Prosody target: SWIPE' (A. Camacho, "SWIPE: a sawtooth waveform inspired
pitch estimator for speech and music", PhD thesis / JASA 124(3) 2008), as `pysptk.swipe(x, fs, hopsize, min, max,
threshold=0.3, otype='f0')` computes it (pase/transforms.py:948-952).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: pysptk 0.1.16 wraps SPTK's swipe.c (K. Gorman's C port of Camacho's
swipep.m), a third-party dependency that is not installed here and has no vectors in the reference repo.  This file
restates the PUBLISHED algorithm (swipep.m) with swipe.c's constants: dlog2p = 1/96 octave candidates, dERBs = 0.1,
50 % window overlap, Hann windows of the power-of-two sizes nearest 8 periods, prime-harmonic cosine kernels with a
1/sqrt(f) envelope, pitch strength = normalised inner product with the sqrt-magnitude ERB spectrum, linear
interpolation in time, lambda-weighted combination across window sizes, parabolic refinement on a 1/768-octave grid,
strength threshold st; unvoiced frames are reported as f0 = 0.  Details where swipe.c is known to differ from
swipep.m in round-off only (its own spline and FFT) are not modelled.
"""
import numpy as np
from scipy.interpolate import CubicSpline

DLOG2P = 1.0 / 96.0
DERBS = 0.1
POLYV = 1.0 / 12.0 / 64.0      # 1/768 octave
WOVERLAP = 0.5


def hz2erbs(hz):
    return 21.4 * np.log10(1.0 + np.asarray(hz, dtype=np.float64) / 229.0)


def erbs2hz(erbs):
    return (10.0 ** (np.asarray(erbs, dtype=np.float64) / 21.4) - 1.0) * 229.0


def primes_upto(n):
    return [p for p in range(2, n + 1) if all(p % q for q in range(2, int(p ** 0.5) + 1))]


def candidates(fmin, fmax):
    log2pc = np.arange(np.log2(fmin), np.log2(fmax), DLOG2P)
    return log2pc, 2.0 ** log2pc


def window_sizes(fs, fmin, fmax):
    logws = np.round(np.log2(8.0 * fs / np.array([fmin, fmax], dtype=np.float64))).astype(int)
    ws = 2 ** np.arange(logws[0], logws[1] - 1, -1)
    return ws, 8.0 * fs / ws


def hanning(n):
    """MATLAB hanning(n): 0.5 (1 - cos(2 pi k / (n + 1))), k = 1..n."""
    k = np.arange(1, n + 1, dtype=np.float64)
    return 0.5 * (1.0 - np.cos(2.0 * np.pi * k / (n + 1)))


def kernel_matrix(f, pc_sel):
    """pitchStrengthAllCandidates' kernels as a matrix (len(pc_sel), len(f)) with zeros below each candidate's first
    bin k(j) (first f > pc/4), and k(j) itself."""
    Kmat = np.zeros((len(pc_sel), len(f)))
    kidx = np.zeros(len(pc_sel), dtype=np.int64)
    start = 0
    for j, pc in enumerate(pc_sel):
        start = start + int(np.argmax(f[start:] > pc / 4.0))
        kidx[j] = start
        fj = f[start:]
        n = int(np.fix(fj[-1] / pc - 0.75))
        if n == 0:
            Kmat[j] = np.nan
            continue
        q = fj / pc
        k = np.zeros(len(fj))
        for i in [1] + primes_upto(n):
            a = np.abs(q - i)
            pk = a < 0.25
            k[pk] = np.cos(2.0 * np.pi * q[pk])
            v = (0.25 < a) & (a < 0.75)
            k[v] = k[v] + np.cos(2.0 * np.pi * q[v]) / 2.0
        k = k * np.sqrt(1.0 / fj)
        k = k / np.linalg.norm(k[k > 0])
        Kmat[j, start:] = k
    return Kmat, kidx


def plan(fs, fmin, fmax):
    """Everything that does not depend on the signal (shared with the device implementation's host setup):
    candidates, window sizes, and per window size: hop, candidate subset + mu weights, ERB frequency subset, the
    natural-cubic-spline interpolation matrix (bins -> ERB frequencies), the kernel matrix and the tail mask."""
    log2pc, pc = candidates(fmin, fmax)
    ws, pO = window_sizes(fs, fmin, fmax)
    d = 1.0 + log2pc - np.log2(8.0 * fs / ws[0])
    ferbs = erbs2hz(np.arange(hz2erbs(pc.min() / 4.0), hz2erbs(fs / 2.0), DERBS))
    per_ws = []
    for i, w in enumerate(ws):
        dn = max(1, int(round(8.0 * (1.0 - WOVERLAP) * fs / pO[i])))
        ii = i + 1
        if len(ws) == 1:
            j = np.arange(len(pc))
            k = np.array([], dtype=np.int64)
        elif ii == len(ws):
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
        fbins = np.arange(w // 2 + 1) * fs / float(w)
        E = CubicSpline(fbins, np.eye(len(fbins)), bc_type="natural", extrapolate=False)(ferbs)
        E = np.nan_to_num(E, nan=0.0)                      # interp1(..., 'spline', 0): zero outside the grid
        Kmat, kidx = kernel_matrix(ferbs, pc[j])
        tail = (np.arange(len(ferbs))[None, :] >= kidx[:, None]).astype(np.float64)
        per_ws.append(dict(ws=int(w), dn=dn, j=j, mu=mu, ferbs=ferbs.copy(), E=E, K=Kmat, tail=tail))
    return dict(pc=pc, log2pc=log2pc, per_ws=per_ws)


def refine(S_col, pc, log2pc, st):
    """argmax + parabolic refinement of one frame's strengths -> (f0 or 0, strength)."""
    if not np.any(np.isfinite(S_col)):
        return 0.0, np.nan
    i = int(np.nanargmax(S_col))
    s = S_col[i]
    if not (s >= st):
        return 0.0, s
    if i == 0 or i == len(pc) - 1:
        return pc[i], s
    I = [i - 1, i, i + 1]
    tc = 1.0 / pc[I]
    ntc = (tc / tc[1] - 1.0) * 2.0 * np.pi
    c = np.polyfit(ntc, S_col[I], 2)
    grid = np.arange(log2pc[I[0]], log2pc[I[2]] + 1e-12, POLYV)
    ftc = 1.0 / 2.0 ** grid
    nftc = (ftc / tc[1] - 1.0) * 2.0 * np.pi
    vals = np.polyval(c, nftc)
    k = int(np.argmax(vals))
    return 2.0 ** (log2pc[I[0]] + k * POLYV), vals[k]


def swipe(x, fs=16000, hopsize=160, fmin=60.0, fmax=300.0, st=0.3, return_strength=False):
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    dt = hopsize / float(fs)
    nt = int(np.floor(len(x) / float(fs) / dt + 1e-9)) + 1
    t = np.arange(nt) * dt
    pl = plan(fs, fmin, fmax)
    pc, log2pc = pl["pc"], pl["log2pc"]
    S = np.zeros((len(pc), nt))
    for w in pl["per_ws"]:
        ws, dn = w["ws"], w["dn"]
        xzp = np.concatenate([np.zeros(ws // 2), x, np.zeros(dn + ws // 2)])
        nfr = (len(xzp) - (ws - dn)) // dn
        win = hanning(ws)
        frames = np.stack([xzp[i * dn:i * dn + ws] * win for i in range(nfr)], 1)
        X = np.abs(np.fft.rfft(frames, axis=0))                     # (ws/2+1, nfr)
        ti = np.arange(nfr) * dn / float(fs)
        L2 = np.maximum(0.0, w["E"] @ X)                            # loudness^2 at the ERB frequencies
        L = np.sqrt(L2)
        num = w["K"] @ L
        nrm = np.sqrt(w["tail"] @ L2)
        nrm[nrm == 0] = np.inf
        Si = num / nrm                                              # (len(j), nfr)
        Sit = np.stack([np.interp(t, ti, row, left=np.nan, right=np.nan) for row in Si], 0) if nfr > 1 \
            else np.full((Si.shape[0], nt), np.nan)
        S[w["j"]] += w["mu"][:, None] * Sit
    f0 = np.zeros(nt)
    strength = np.zeros(nt)
    for jf in range(nt):
        f0[jf], strength[jf] = refine(S[:, jf], pc, log2pc, st)
    return (f0, strength) if return_strength else f0

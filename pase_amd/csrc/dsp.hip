// dsp.hip -- HBM-bound pieces of the on-device regression targets:
// the spectra themselves are DFT-basis convolutions on pase_conv_gemm (PASE_POST_POW / _LOGPOW / _LOG);
// here: librosa-style delta features (Savitzky-Golay, width 9, mode='interp') fused with the ZNorm
// of pase/transforms.py:183-205, and librosa.power_to_db's per-utterance top_db clamp.
//
// Reference arithmetic (third-party, not installed here; restated in oracle/dsp_oracle.py):
//   librosa 0.6.3 feature.delta(X, width=9, order=n) = scipy.signal.savgol_filter(X, 9, polyorder=n,
//     deriv=n, axis=-1, mode='interp')                       (transforms.py:475-477, :528-530, :709-711)
//   librosa.power_to_db(S, ref=1, amin=1e-10, top_db=80): 10 log10(max(amin,S)) clipped at max-80.
#include "hip_compat.h"
#include "pase_amd.h"

namespace {

constexpr int NT = 256;

// out[b, k*D + d, t] = (delta_k(x[b, d, :])[t] - mean[k*D+d]) * istd[k*D+d],  k = 0..order
// coef: (order+1, 9, 9): row (k, pos, j) = weight of sample j of a 9-window when the output sits at
// window position pos; pos == 4 is the interior filter, pos < 4 / > 4 the 'interp' edge fits.
__global__ void __launch_bounds__(NT) delta_znorm_kernel(const float* x, const float* coef, const float* mean,
                                                         const float* istd, float* out, int B, int D, int F, int Fo,
                                                         int order, int x_ctot, int x_coff) {
    const long total = (long)B * D * Fo;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int t = (int)(i % Fo);
        const int d = (int)((i / Fo) % D);
        const int b = (int)(i / ((long)Fo * D));
        const float* row = x + ((size_t)b * x_ctot + x_coff + d) * (size_t)F;
        // window start and position of t inside it (replicate-pad columns t >= F copy column F-1)
        const int tt = t < F ? t : F - 1;
        int w0 = tt - 4, pos = 4;
        if (F >= 9) {
            if (w0 < 0) { pos = tt; w0 = 0; }
            if (w0 + 9 > F) { w0 = F - 9; pos = tt - w0; }
        }
        float xs[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int u = w0 + j;
            xs[j] = (u >= 0 && u < F) ? row[u] : 0.f;
        }
        for (int k = 0; k <= order; ++k) {
            float v;
            if (k == 0) {
                v = row[tt];
            } else {
                const float* c = coef + ((size_t)k * 9 + pos) * 9;
                v = 0.f;
#pragma unroll
                for (int j = 0; j < 9; ++j) v = fmaf(c[j], xs[j], v);
            }
            const int ch = k * D + d;
            if (mean) v = (v - mean[ch]) * istd[ch];
            out[((size_t)b * (order + 1) * D + ch) * (size_t)Fo + t] = v;
        }
    }
}

// Framed energy and zero-crossing rate of the Prosody target (pase/transforms.py:967-978):
//   librosa.feature.rmse(y, frame_length=win, hop_length=hop, center=True, pad_mode='constant'):
//       sqrt(mean(x^2)) over frames of the signal ZERO-padded by win/2 on both sides;
//   librosa.feature.zero_crossing_rate(y, frame_length=win, hop_length=hop, center=True):
//       signal EDGE-padded by win/2; |y| <= 1e-10 counts as +0; a crossing = np.signbit differs between
//       consecutive samples INSIDE the frame (the frame's first sample compares with nothing: pad=False);
//       rate = crossings / win.
// out (B, out_ctot, F): row out_coff = energy, row out_coff + 1 = zcr (the reference's [.., egy, zcr] order).
// One wave per frame, lanes stride over the window.
__global__ void __launch_bounds__(NT) zcr_rms_kernel(const float* x, float* out, int B, int T, int F, int hop, int win,
                                                     int out_ctot, int out_coff) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long fr = (long)blockIdx.x * (NT / 64) + wave;
    if (fr >= (long)B * F) return;
    const int b = (int)(fr / F), f = (int)(fr % F);
    const float* xr = x + (size_t)b * T;
    const int u0 = f * hop - win / 2;
    float e = 0.f;
    int zc = 0;
    for (int i = lane; i < win; i += 64) {
        const int u = u0 + i;
        const float v = (u >= 0 && u < T) ? xr[u] : 0.f;                 // zero padding (energy)
        e = fmaf(v, v, e);
        if (i > 0) {
            float a = xr[min(max(u - 1, 0), T - 1)], c = xr[min(max(u, 0), T - 1)];   // edge padding (zcr)
            a = fabsf(a) <= 1e-10f ? 0.f : a;
            c = fabsf(c) <= 1e-10f ? 0.f : c;
            zc += ((a < 0.f) != (c < 0.f)) ? 1 : 0;       // np.signbit after the threshold (no -0 / NaN left)
        }
    }
    e = pase_wave_sum64(e);
    float z = pase_wave_sum64((float)zc);
    if (lane == 0) {
        float* o = out + ((size_t)b * out_ctot + out_coff) * (size_t)F + f;
        o[0] = sqrtf(e / (float)win);
        o[F] = z / (float)win;
    }
}

// log-f0 contour of the Prosody target (pase/transforms.py:948-961): lf0 = log(f0 + 1e-10) (f0 = 0 on unvoiced
// frames), then ahoproc_tools.interpolate.interpolation(lf0, -1): unvoiced stretches (lf0 <= -1) are bridged
// linearly between their voiced neighbours, a leading stretch takes the first voiced value, a trailing one the
// last voiced value; uv = 1 on voiced frames; an all-unvoiced chunk gets lf0 = log(f0_min), uv = 0.
// One thread per utterance (F = 200 frames, sequential by nature).  out rows: out_coff = lf0, out_coff+1 = uv.
__global__ void lf0_interp_kernel(const float* f0, float* out, int B, int Fin, int F, int out_ctot, int out_coff,
                                  float f0_min) {
    // Fin = frames of the tracker's contour (the interpolation runs over ALL of them, transforms.py:955), F <= Fin =
    // frames kept (:956-957 truncate afterwards; :958-960 then test the truncated voiced flags)
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* f = f0 + (size_t)b * Fin;
    float* lf = out + ((size_t)b * out_ctot + out_coff) * (size_t)F;
    float* uv = lf + F;
    const double us = -1.0;
    auto L = [&](int t) { return log((double)f[t] + 1e-10); };
    auto put = [&](int i, double v, float flag) { if (i < F) { lf[i] = (float)v; uv[i] = flag; } };
    bool any_voiced = false;
    for (int t = 0; t < Fin; ++t) { const double v = L(t); any_voiced = any_voiced || v > us; put(t, v, 1.f); }
    int tb0 = -1;              // tbound[0]
    double fb0 = 0.0;          // fbound[0]
    bool have_b0 = false;      // tbound != [None, None]
    double prev = L(0);
    for (int t = 1; t < Fin; ++t) {
        const double cur = L(t);
        if (cur > us && prev <= us && !have_b0) {
            // leading unvoiced stretch: constant first voiced value
            for (int i = 0; i < t; ++i) put(i, cur, 0.f);
        } else if (cur <= us && prev > us) {
            tb0 = t - 1; fb0 = prev; have_b0 = true;
        } else if (cur > us && prev <= us) {
            const double slope = (cur - fb0) / (double)(t - tb0);
            for (int i = tb0; i < t; ++i) put(i, fb0 + (double)(i - tb0) * slope, 0.f);
            have_b0 = false; tb0 = -1;
        }
        prev = cur;
    }
    if (have_b0) for (int i = tb0; i < Fin; ++i) put(i, fb0, 0.f);
    float uvsum = 0.f;
    if (!any_voiced) for (int t = 0; t < F; ++t) uv[t] = 0.f;
    for (int t = 0; t < F; ++t) uvsum += uv[t];
    if (uvsum == 0.f) for (int t = 0; t < F; ++t) lf[t] = logf(f0_min);
}

// ---- SWIPE' pitch tracker (the f0 contour behind the Prosody target, pase/transforms.py:948-952: pysptk.swipe) ----
// The spectra, the ERB-scale loudness and the kernel inner products are pase_conv_gemm launches (pase_amd/dsp.py:
// SwipeTracker); these two kernels do the rest.
// (1) per window size: pitch strength S_i = (K . L) / sqrt(tail . L^2) at that window's own frame rate (0 when the
//     loudness above the candidate's first bin is zero), interpolated linearly in time to the output frame times
//     t = f * hop / fs and accumulated, weighted by mu[c], into the rows cand_index[c] of S (B, NC, F).
__global__ void __launch_bounds__(NT) swipe_accumulate_kernel(const float* num, const float* den2, const float* mu,
                                                              const int* cand, float* S, int B, int nj, int nfr, int NC,
                                                              int F, float frames_per_out) {
    const long total = (long)B * nj * F;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int f = (int)(i % F);
        const int c = (int)((i / F) % nj);
        const int b = (int)(i / ((long)F * nj));
        const float pos = (float)f * frames_per_out;          // output time in units of this window's hop
        int i0 = (int)floorf(pos);
        float w = pos - (float)i0;
        if (i0 >= nfr - 1) { i0 = nfr - 2; w = pos - (float)i0; }
        const float* nr = num + ((size_t)b * nj + c) * (size_t)nfr;
        const float* dr = den2 + ((size_t)b * nj + c) * (size_t)nfr;
        const float d0 = dr[i0], d1 = dr[i0 + 1];
        const float s0 = d0 > 0.f ? nr[i0] / sqrtf(d0) : 0.f;
        const float s1 = d1 > 0.f ? nr[i0 + 1] / sqrtf(d1) : 0.f;
        float v = s0 + w * (s1 - s0);
        if (w > 1.f || pos < 0.f) v = 0.f;                    // outside the analysed span (interp1's NaN): no support
        S[((size_t)b * NC + cand[c]) * (size_t)F + f] += mu[c] * v;
    }
}

// (2) per output frame: strongest candidate, strength threshold, parabolic refinement on a 1/768-octave grid
//     (swipep.m's polyfit through the three strengths around the maximum, in normalised period units).
__global__ void __launch_bounds__(NT) swipe_pick_kernel(const float* S, float* f0, float* strength, int B, int NC, int F,
                                                        float log2_fmin, float dlog2p, float polyv, float st) {
    const long total = (long)B * F;
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const int f = (int)(i % F), b = (int)(i / F);
    const float* col = S + (size_t)b * NC * (size_t)F + f;
    int im = 0;
    float sm = col[0];
    for (int c = 1; c < NC; ++c) {
        const float v = col[(size_t)c * F];
        if (v > sm) { sm = v; im = c; }
    }
    double pitch = 0.0, sbest = (double)sm;
    if (sm >= st) {
        if (im == 0 || im == NC - 1) {
            pitch = exp2((double)log2_fmin + (double)im * (double)dlog2p);
        } else {
            const double l0 = (double)log2_fmin + (double)(im - 1) * (double)dlog2p;
            const double tc1 = exp2(-(l0 + (double)dlog2p));
            double xs[3], ys[3];
            for (int k = 0; k < 3; ++k) {
                xs[k] = (exp2(-(l0 + k * (double)dlog2p)) / tc1 - 1.0) * 6.283185307179586;
                ys[k] = (double)col[(size_t)(im - 1 + k) * F];
            }
            // interpolating parabola through the three points (what polyfit(.,.,2) returns for three samples)
            const double d01 = xs[0] - xs[1], d02 = xs[0] - xs[2], d12 = xs[1] - xs[2];
            const double a0 = ys[0] / (d01 * d02), a1 = -ys[1] / (d01 * d12), a2 = ys[2] / (d02 * d12);
            const int ng = (int)floor(2.0 * (double)dlog2p / (double)polyv + 1e-9) + 1;
            int kb = 0;
            double vb = -1e300;
            for (int k = 0; k < ng; ++k) {
                const double x = (exp2(-(l0 + k * (double)polyv)) / tc1 - 1.0) * 6.283185307179586;
                const double v = a0 * (x - xs[1]) * (x - xs[2]) + a1 * (x - xs[0]) * (x - xs[2]) + a2 * (x - xs[0]) * (x - xs[1]);
                if (v > vb) { vb = v; kb = k; }
            }
            pitch = exp2(l0 + kb * (double)polyv);
            sbest = vb;
        }
    }
    f0[i] = (float)pitch;
    if (strength) strength[i] = (float)sbest;
}

// Framing prologue shared by every spectral target: the waveform, padded (reflect / zero) by padL on the
// left and optionally pre-emphasised (python_speech_features.sigproc.preemphasis: y[0] = x[0],
// y[n] = x[n] - c x[n-1]), is laid out hop-major -- y[b][r][q] = xpad[q*hop + r] -- so that a frame of
// `win` samples at hop `hop` becomes a stride-1 conv over q with `hop` input channels and
// ceil(win/hop) taps: the strided STFT runs on the same sliding-window implicit GEMM as the encoder.
__global__ void __launch_bounds__(NT) frame_prep_kernel(const float* x, float* y, int T, int hop, int Q, int padL,
                                                        int reflect, float coeff, long total) {
    const int HQ = hop * Q;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int b = (int)(i / HQ);
        const int rq = (int)(i - (long)b * HQ);
        const int q = rq / hop, r = rq - q * hop;     // consecutive threads walk the waveform: coalesced reads
        int u = q * hop + r - padL;
        float v = 0.f;
        if (reflect) {
            if (u < 0) u = -u;
            if (u >= T) u = 2 * (T - 1) - u;
        }
        if (u >= 0 && u < T) {
            const float* xb = x + (long)b * T;
            v = xb[u];
            if (coeff != 0.f && u > 0) v -= coeff * xb[u - 1];
        }
        y[((long)b * hop + r) * Q + q] = v;
    }
}

__device__ __forceinline__ unsigned f2ord(float f) {   // order-preserving float -> uint
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

// y = 10 log10(max(amin, x)) - ref_db ; per-utterance running max (ordered-uint atomicMax)
__global__ void __launch_bounds__(NT) power_to_db_kernel(const float* x, float* y, unsigned* umax, long per_utt,
                                                         int B, float amin, float ref_db) {
    __shared__ unsigned sh[NT / 64];
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * per_utt;
    float* yb = y + (size_t)b * per_utt;
    float mx = -3.0e38f;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < per_utt; i += (long)gridDim.x * NT) {
        const float v = 10.f * log10f(fmaxf(amin, xb[i])) - ref_db;
        yb[i] = v;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = f2ord(mx);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned u = sh[0];
        for (int w = 1; w < NT / 64; ++w) u = max(u, sh[w]);
        atomicMax(umax + b, u);
    }
}

__global__ void __launch_bounds__(NT) clamp_top_db_kernel(float* y, const unsigned* umax, long per_utt, int B,
                                                          float top_db) {
    const int b = blockIdx.y;
    const float lo = ord2f(umax[b]) - top_db;
    float* yb = y + (size_t)b * per_utt;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < per_utt; i += (long)gridDim.x * NT)
        yb[i] = fmaxf(yb[i], lo);
}

}  // namespace

extern "C" int pase_delta_znorm(const float* x, const float* coef, const float* mean, const float* istd, float* out,
                                int B, int D, int F, int Fo, int order, int x_ctot, int x_coff, void* stream) {
    if (order < 0 || order > 2 || Fo < F) return -2;
    const long total = (long)B * D * Fo;
    if (total <= 0) return 0;
    long blocks = (total + NT - 1) / NT;
    if (blocks > 4096) blocks = 4096;
    PASE_LAUNCH(delta_znorm_kernel, dim3((unsigned)blocks), dim3(NT), (hipStream_t)stream, x, coef, mean, istd, out, B, D,
                F, Fo, order, x_ctot, x_coff);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_zcr_rms(const float* x, float* out, int B, int T, int F, int hop, int win, int out_ctot,
                            int out_coff, void* stream) {
    if (B <= 0 || F <= 0) return 0;
    if (win < 2 || hop < 1 || T < 1 || out_coff + 2 > out_ctot) return -2;
    const long frames = (long)B * F;
    PASE_LAUNCH(zcr_rms_kernel, dim3((unsigned)((frames + NT / 64 - 1) / (NT / 64))), dim3(NT), (hipStream_t)stream, x, out,
                B, T, F, hop, win, out_ctot, out_coff);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_lf0_interp(const float* f0, float* out, int B, int Fin, int F, int out_ctot, int out_coff,
                               float f0_min, void* stream) {
    if (B <= 0 || F <= 0) return 0;
    if (out_coff + 2 > out_ctot || Fin < F) return -2;
    PASE_LAUNCH(lf0_interp_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), (hipStream_t)stream, f0, out, B, Fin, F,
                out_ctot, out_coff, f0_min);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_swipe_accumulate(const float* num, const float* den2, const float* mu, const int* cand, float* S, int B,
                                     int nj, int nfr, int NC, int F, float frames_per_out, void* stream) {
    const long total = (long)B * nj * F;
    if (total <= 0) return 0;
    if (nfr < 2) return -2;
    long blocks = (total + NT - 1) / NT;
    if (blocks > 65535) blocks = 65535;
    PASE_LAUNCH(swipe_accumulate_kernel, dim3((unsigned)blocks), dim3(NT), (hipStream_t)stream, num, den2, mu, cand, S, B,
                nj, nfr, NC, F, frames_per_out);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_swipe_pick(const float* S, float* f0, float* strength, int B, int NC, int F, float log2_fmin,
                               float dlog2p, float polyv, float st, void* stream) {
    const long total = (long)B * F;
    if (total <= 0) return 0;
    if (NC < 1) return -2;
    PASE_LAUNCH(swipe_pick_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), (hipStream_t)stream, S, f0, strength,
                B, NC, F, log2_fmin, dlog2p, polyv, st);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_power_to_db(const float* x, float* y, unsigned* umax_scratch, long per_utt, int B, float amin,
                                float ref_db, float top_db, void* stream) {
    if (per_utt <= 0 || B <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(umax_scratch, 0, sizeof(unsigned) * (size_t)B, st);
    if (e != hipSuccess) return (int)e;
    long bx = (per_utt + NT - 1) / NT;
    if (bx > 64) bx = 64;
    PASE_LAUNCH(power_to_db_kernel, dim3((unsigned)bx, (unsigned)B), dim3(NT), st, x, y, umax_scratch, per_utt, B, amin,
                ref_db);
    if (top_db > 0.f)
        PASE_LAUNCH(clamp_top_db_kernel, dim3((unsigned)bx, (unsigned)B), dim3(NT), st, y, umax_scratch, per_utt, B, top_db);
    PASE_CHECK_LAUNCH();
    return 0;
}

// ---- Gammatone filterbank energies (gammatone.gtgram, Slaney's ERB filter design) ---------------------------
// One thread per (utterance, channel): four cascaded second-order sections in fp64 (scipy.signal.lfilter's
// transposed direct form II, as gammatone.filters.erb_filterbank applies them), squared output accumulated
// into sums over blocks of `g` samples; a window of nwin = k*g samples is then a sum of k block sums.
// coef (C, 10) doubles: A0, A11, A12, A13, A14, A2, B0, B1, B2, gain (gammatone.filters.make_erb_filters).
// The recurrence is split in time: a thread owns `bps` blocks and first runs the filters over the `warm` samples
// before its segment to rebuild the state (the 4th-order gammatone impulse response decays like
// t^3 exp(-2 pi 1.019 ERB t): after 2048 samples at 16 kHz the truncated history is below exp(-24) even for
// a 50 Hz channel, i.e. far under fp32 resolution of the log-energy output).
__global__ void __launch_bounds__(64) gammatone_blocks_kernel(const float* x, const double* coef, float* blocks,
                                                              int B, int C, int T, int g, int nblk, int bps, int nseg,
                                                              int warm) {
    const int i0 = blockIdx.x * 64 + threadIdx.x;
    if (i0 >= B * C * nseg) return;
    const int sidx = i0 / (B * C);
    const int i = i0 - sidx * (B * C);
    const int b = i / C, ch = i - b * C;
    const double* k = coef + (size_t)ch * 10;
    const double A0 = k[0], A2 = k[5], B0 = k[6], B1 = k[7] / k[6], B2 = k[8] / k[6], gain = k[9];
    const double b0[4] = {A0 / gain / B0, A0 / B0, A0 / B0, A0 / B0};
    const double b1[4] = {k[1] / gain / B0, k[2] / B0, k[3] / B0, k[4] / B0};
    const double b2[4] = {A2 / gain / B0, A2 / B0, A2 / B0, A2 / B0};
    double z0[4] = {0, 0, 0, 0}, z1[4] = {0, 0, 0, 0};
    const float* xb = x + (size_t)b * T;
    float* out = blocks + (size_t)i * nblk;
    const int q0 = sidx * bps, q1 = min(nblk, q0 + bps);
    for (int n = max(0, q0 * g - warm); n < q0 * g; ++n) {   // state warm-up, nothing accumulated
        double v = (double)xb[n];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const double y = b0[st] * v + z0[st];
            z0[st] = b1[st] * v - B1 * y + z1[st];
            z1[st] = b2[st] * v - B2 * y;
            v = y;
        }
    }
    for (int q = q0; q < q1; ++q) {
        double acc = 0.0;
        const int n1 = min(T, (q + 1) * g);
        for (int n = q * g; n < n1; ++n) {
            double v = (double)xb[n];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const double y = b0[st] * v + z0[st];
                z0[st] = b1[st] * v - B1 * y + z1[st];
                z1[st] = b2[st] * v - B2 * y;
                v = y;
            }
            acc += v * v;
        }
        out[q] = (float)acc;
    }
}

// frame c = log(sqrt(mean of nwin squared samples from c*hop) + eps): kblk block sums from block c*hblk
__global__ void __launch_bounds__(NT) gammatone_frames_kernel(const float* blocks, float* out, long total, int nblk,
                                                              int ncol, int kblk, int hblk, float inv_nwin, float eps) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % ncol);
        const long row = i / ncol;
        const float* bl = blocks + row * nblk + (long)c * hblk;
        float s = 0.f;
        for (int j = 0; j < kblk; ++j) s += bl[j];
        out[i] = logf(sqrtf(s * inv_nwin) + eps);
    }
}

extern "C" int pase_frame_prep(const float* x, float* y, int B, int T, int hop, int Q, int padL, int pad_mode,
                               float preemph, void* stream) {
    const long total = (long)B * hop * Q;
    if (total <= 0) return 0;
    if (pad_mode == PASE_PAD_REFLECT && (padL >= T || (long)Q * hop - padL > 2L * T - 1)) return -3;
    long blocks = (total + NT - 1) / NT;
    if (blocks > 8192) blocks = 8192;
    PASE_LAUNCH(frame_prep_kernel, dim3((unsigned)blocks), dim3(NT), (hipStream_t)stream, x, y, T, hop, Q, padL,
                pad_mode == PASE_PAD_REFLECT ? 1 : 0, preemph, total);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_gammatone_blocks(const float* x, const double* coef, float* blocks, int B, int C, int T, int g,
                                     void* stream) {
    if (B <= 0 || C <= 0 || T <= 0) return 0;
    if (g < 1) return -2;
    const int nblk = (T + g - 1) / g;
    const int warm = 2048;
    int bps = (2000 + g - 1) / g;                         // ~2000-sample segments
    if (bps < 1) bps = 1;
    const int nseg = (nblk + bps - 1) / bps;
    PASE_LAUNCH(gammatone_blocks_kernel, dim3((unsigned)(((long)B * C * nseg + 63) / 64)), dim3(64), (hipStream_t)stream,
                x, coef, blocks, B, C, T, g, nblk, bps, nseg, warm);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_gammatone_frames(const float* blocks, float* out, int rows, int T, int g, int nwin, int hop,
                                     int ncol, float eps, void* stream) {
    if (rows <= 0 || ncol <= 0) return 0;
    if (g < 1 || (nwin % g) || (hop % g) || (long)(ncol - 1) * hop + nwin > T) return -2;
    const int nblk = (T + g - 1) / g;
    const long total = (long)rows * ncol;
    long blocks_n = (total + NT - 1) / NT;
    if (blocks_n > 4096) blocks_n = 4096;
    PASE_LAUNCH(gammatone_frames_kernel, dim3((unsigned)blocks_n), dim3(NT), (hipStream_t)stream, blocks, out, total,
                nblk, ncol, nwin / g, hop / g, 1.0f / (float)nwin, eps);
    PASE_CHECK_LAUNCH();
    return 0;
}

// producer.hip -- on-device batch producer: chunking, peak-normalise + random scale, reverberation and
// additive noise.  The reference does all of this per utterance
// on DataLoader CPU workers with numpy / scipy (pase/transforms.py:309-436 SingleChunkWav / MIChunkWav,
// :148-151 norm_and_scale, :1071-1103 Reverb.__call__, :1633-1675 SimpleAdditive.__call__); here the
// waveforms, impulse responses and noises are resident in HBM and one launch handles the whole batch.
// Random decisions (which utterance, where the crop starts, which IR / noise / SNR) stay on the host, as
// small index arrays: they are the reference's numpy / random draws, not arithmetic.
//
// All kernels are HBM- or VALU-bound elementwise / FIR work; nothing here is reshaped into a GEMM
// except the FIR, which is a register-tiled sliding window (8 outputs per lane, operands in LDS).
#include "hip_compat.h"
#include "pase_amd.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    v = pase_wave_sum64d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < NT / 64; ++i) t += sh[i];
    return t;
}
__device__ __forceinline__ float block_max_f(float v, float* sh) {
    for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = sh[0];
    for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, sh[i]);
    return t;
}

// select_chunk (transforms.py:309-356): chunk n = wav[src[n]][beg[n] : beg[n]+T]; a waveform not longer than
// T is right-padded by reflection (F.pad(..., (0, P), mode='reflect'): index len-2-(t-len)).
__global__ void __launch_bounds__(NT) chunk_gather_kernel(const float* pool, const long long* off, const int* len,
                                                          const int* src, const int* beg, float* out, int T) {
    const int n = blockIdx.y;
    const int u = src[n];
    const float* w = pool + off[u];
    const int L = len[u];
    const int b0 = L <= T ? 0 : beg[n];
    for (int t = blockIdx.x * NT + threadIdx.x; t < T; t += gridDim.x * NT) {
        int i = b0 + t;
        if (i >= L) i = 2 * (L - 1) - i;      // reflect (valid while P < len, as torch requires)
        if (i < 0) i = 0;
        out[(size_t)n * T + t] = w[i];
    }
}

// SimpleAdditiveShift's interfering-speech crop (transforms.py:1718-1750): out[b, t] = 0 for t < shift[b], else
// wav_{src[b]}[beg[b] + t - shift[b]] (zero past the end of a short file): the crop of length T - shift, front-padded.
__global__ void __launch_bounds__(NT) overlap_gather_kernel(const float* pool, const long long* off, const int* len,
                                                            const int* src, const int* beg, const int* shift,
                                                            float* out, int T) {
    const int n = blockIdx.y;
    const int u = src[n];
    if (u < 0) return;
    const float* w = pool + off[u];
    const int L = len[u], b0 = beg[n], sh = shift[n];
    for (int t = blockIdx.x * NT + threadIdx.x; t < T; t += gridDim.x * NT) {
        const int i = b0 + t - sh;
        out[(size_t)n * T + t] = (t >= sh && i < L) ? w[i] : 0.f;
    }
}

// zero the first shift[b] samples (the reference front-pads AFTER reverberating the crop)
__global__ void __launch_bounds__(NT) zero_front_kernel(float* x, const int* shift, int T) {
    const int n = blockIdx.y;
    const int sh = min(shift[n], T);
    for (int t = blockIdx.x * NT + threadIdx.x; t < sh; t += gridDim.x * NT) x[(size_t)n * T + t] = 0.f;
}

// norm_and_scale (transforms.py:148-151): x / max|x| * u, one block per chunk
__global__ void __launch_bounds__(NT) peak_scale_kernel(float* x, const float* u, int T) {
    __shared__ float sh[NT / 64];
    float* row = x + (size_t)blockIdx.x * T;
    float m = 0.f;
    for (int t = threadIdx.x; t < T; t += NT) m = fmaxf(m, fabsf(row[t]));
    m = block_max_f(m, sh);
    const float uu = u[blockIdx.x];
    for (int t = threadIdx.x; t < T; t += NT) row[t] = row[t] / m * uu;
}

// row energies sum x^2 (double)
__global__ void __launch_bounds__(NT) row_energy_kernel(const float* x, int T, double* e) {
    __shared__ double sh[NT / 64];
    const float* row = x + (size_t)blockIdx.x * T;
    double s = 0.0;
    for (int t = threadIdx.x; t < T; t += NT) s += (double)row[t] * (double)row[t];
    s = block_sum_d(s, sh);
    if (threadIdx.x == 0) e[blockIdx.x] = s;
}

// ---- Reverb: full = x * h ('full' convolution, scipy.signal.convolve), Er = sum full^2 -----------------
// Block = 2048 consecutive outputs of one utterance (8 per lane); per stage 512 taps of h and the matching
// 2048+511 input samples are staged in LDS (zero outside the signal).  Lane owns outputs 8*tid .. 8*tid+7: moving
// to the next tap slides its 8-sample window by one, i.e. ONE new LDS read per 8 FMAs (+ a broadcast read of
// the tap).  The window base 8*tid would hit 4 banks only, so the staged signal is stored skewed
// (i -> i + i/8).
constexpr int FIR_TILE = 2048, FIR_KC = 512, FIR_OPT = 8;
constexpr int FIR_XS = FIR_TILE + FIR_KC;           // staged samples (one spare)
__device__ __forceinline__ int skew(int i) { return i + (i >> 3); }

__global__ void __launch_bounds__(NT) fir_full_kernel(const float* x, const float* irs, const int* ir_len,
                                                      const long long* ir_off, const int* ir_idx, float* full,
                                                      double* Er, int T, int full_stride) {
    __shared__ float xs[FIR_XS + FIR_XS / 8 + 8];
    __shared__ float hs[FIR_KC];
    __shared__ double sh[NT / 64];
    const int b = blockIdx.y;
    const int ii = ir_idx[b];
    if (ii < 0) return;                              // this utterance is not reverberated
    const int L = ir_len[ii];
    const int nfull = T + L - 1;
    const int n0 = blockIdx.x * FIR_TILE;
    if (n0 >= nfull) return;
    const float* h = irs + ir_off[ii];
    const float* xb = x + (size_t)b * T;
    const int tid = threadIdx.x;
    float acc[FIR_OPT];
#pragma unroll
    for (int o = 0; o < FIR_OPT; ++o) acc[o] = 0.f;
    // taps k >= n0 + TILE never touch this tile (x index would be negative); taps k < n0 - T + 1 neither
    const int k_hi = min(L, n0 + FIR_TILE);
    int k_lo = n0 - (T - 1);
    if (k_lo < 0) k_lo = 0;
    k_lo = (k_lo / FIR_KC) * FIR_KC;
    for (int k0 = k_lo; k0 < k_hi; k0 += FIR_KC) {
        __syncthreads();
        // staged sample j <-> x index n0 - k0 - (KC-1) + j ; output o at tap kk reads j = o - kk + KC-1
        const int xbase = n0 - k0 - (FIR_KC - 1);
        for (int j = tid; j < FIR_XS; j += NT) {
            const int xi = xbase + j;
            xs[skew(j)] = (xi >= 0 && xi < T) ? xb[xi] : 0.f;
        }
        for (int j = tid; j < FIR_KC; j += NT) hs[j] = (k0 + j < L) ? h[k0 + j] : 0.f;
        __syncthreads();
        // window w[o] = staged[8 tid + o - kk + KC-1]
        float w[FIR_OPT];
        const int jb = FIR_OPT * tid + FIR_KC - 1;
#pragma unroll
        for (int o = 0; o < FIR_OPT; ++o) w[o] = xs[skew(jb + o)];
        for (int kk = 0; kk < FIR_KC; kk += FIR_OPT) {
#pragma unroll
            for (int r = 0; r < FIR_OPT; ++r) {
                const float hk = hs[kk + r];
                // at tap kk+r the window is w[(o - r) mod 8] for output o, with slot (8 - r) % 8 .. freshly read
#pragma unroll
                for (int o = 0; o < FIR_OPT; ++o) acc[o] = fmaf(hk, w[(o - r + FIR_OPT) % FIR_OPT], acc[o]);
                // slide: the sample leaving on the right (output 7's) is replaced by the new leftmost one
                const int jn = jb - (kk + r) - 1;
                w[(FIR_OPT - 1 - r + FIR_OPT) % FIR_OPT] = jn >= 0 ? xs[skew(jn)] : 0.f;
            }
        }
    }
    double e = 0.0;
#pragma unroll
    for (int o = 0; o < FIR_OPT; ++o) {
        const int n = n0 + FIR_OPT * tid + o;
        if (n < nfull) {
            full[(size_t)b * full_stride + n] = acc[o];
            e += (double)acc[o] * (double)acc[o];
        }
    }
    e = block_sum_d(e, sh);
    if (tid == 0) atomicAdd(Er + b, e);
}

// rev = Eratio * shift(full, -p_max)[:T]  (transforms.py:1088-1098); utterances with ir_idx < 0 are left as is.
// trimmed_energy (BandDrop / Downsample, :1275-1289): the energy ratio is taken on the shifted, trimmed signal,
// so this pass writes it unscaled and fir_rescale_kernel applies the ratio afterwards.
__global__ void __launch_bounds__(NT) reverb_finish_kernel(float* x, const float* full, const int* ir_len,
                                                           const int* ir_pmax, const int* ir_idx, const double* Ex,
                                                           const double* Er, int T, int full_stride,
                                                           int trimmed_energy) {
    const int b = blockIdx.y;
    const int ii = ir_idx[b];
    if (ii < 0) return;
    const int nfull = T + ir_len[ii] - 1;
    const int p = ir_pmax[ii];
    const float ratio = trimmed_energy ? 1.f : (Er[b] > 0.0 ? (float)sqrt(Ex[b] / Er[b]) : 1.f);
    for (int t = blockIdx.x * NT + threadIdx.x; t < T; t += gridDim.x * NT) {
        const int n = t + p;
        x[(size_t)b * T + t] = n < nfull ? ratio * full[(size_t)b * full_stride + n] : 0.f;
    }
}

__global__ void __launch_bounds__(NT) fir_rescale_kernel(float* x, const int* ir_idx, const double* Ex, int T) {
    __shared__ double sh[NT / 64];
    const int b = blockIdx.x;
    if (ir_idx[b] < 0) return;
    float* row = x + (size_t)b * T;
    double e = 0.0;
    for (int t = threadIdx.x; t < T; t += NT) e += (double)row[t] * (double)row[t];
    e = block_sum_d(e, sh);
    const float ratio = e > 0.0 ? (float)sqrt(Ex[b] / e) : 1.f;
    for (int t = threadIdx.x; t < T; t += NT) row[t] *= ratio;
}

// Clipping.__call__ (transforms.py:1514-1535): clamp to [cf * min(x), cf * max(x)]; cf[b] <= 0 leaves b untouched
__global__ void __launch_bounds__(NT) clip_kernel(float* x, const float* cf, int T) {
    __shared__ float sh[NT / 64];
    const int b = blockIdx.x;
    const float f = cf[b];
    if (!(f > 0.f)) return;
    float* row = x + (size_t)b * T;
    float mx = -3.4e38f, mn = 3.4e38f;
    for (int t = threadIdx.x; t < T; t += NT) { mx = fmaxf(mx, row[t]); mn = fminf(mn, row[t]); }
    mx = block_max_f(mx, sh);
    mn = -block_max_f(-mn, sh);
    const float lo = f * mn, hi = f * mx;
    for (int t = threadIdx.x; t < T; t += NT) row[t] = fminf(fmaxf(row[t], lo), hi);
}

// SimpleAdditive.__call__ (transforms.py:1633-1675): noise crop, K = sqrt(Ex / (10^(snr/10) En)),
// noisy = wav + K noise renormalised to the clean energy (norm_energy, eps 1e-14).  One block per utterance.
__global__ void __launch_bounds__(NT) add_noise_kernel(float* x, const float* npool, const long long* noff,
                                                       const int* nlen, const int* nidx, const int* nbeg,
                                                       const float* snr, int T) {
    __shared__ double sh[NT / 64];
    const int b = blockIdx.x;
    const int ni = nidx[b];
    if (ni < 0) return;
    float* row = x + (size_t)b * T;
    const float* nz = npool + noff[ni];
    const int NL = nlen[ni], nb = nbeg[b];
    auto noise_at = [&](int t) { const int i = nb + t; return i < NL ? nz[i] : 0.f; };   // short noises are zero-padded
    double ex = 0.0, en = 0.0;
    for (int t = threadIdx.x; t < T; t += NT) {
        const float w = row[t], n = noise_at(t);
        ex += (double)w * w;
        en += (double)n * n;
    }
    ex = block_sum_d(ex, sh);
    en = block_sum_d(en, sh);
    if (!(en > 0.0)) return;                         // silent noise: the chunk is returned unchanged
    const float K = (float)sqrt(ex / (pow(10.0, (double)snr[b] / 10.0) * en));
    double eo = 0.0;
    for (int t = threadIdx.x; t < T; t += NT) {
        const float v = row[t] + K * noise_at(t);
        eo += (double)v * v;
    }
    eo = block_sum_d(eo, sh);
    const float g = (float)sqrt(ex / (eo + 1e-14));
    for (int t = threadIdx.x; t < T; t += NT) row[t] = g * (row[t] + K * noise_at(t));
}

}  // namespace

extern "C" int pase_chunk_gather(const float* pool, const long long* off, const int* len, const int* src,
                                 const int* beg, float* out, int N, int T, void* stream) {
    if (N <= 0 || T <= 0) return 0;
    PASE_LAUNCH(chunk_gather_kernel, dim3((unsigned)((T + NT * 4 - 1) / (NT * 4)), (unsigned)N), dim3(NT),
                (hipStream_t)stream, pool, off, len, src, beg, out, T);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_overlap_gather(const float* pool, const long long* off, const int* len, const int* src,
                                   const int* beg, const int* shift, float* out, int B, int T, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    PASE_LAUNCH(overlap_gather_kernel, dim3((unsigned)((T + NT * 4 - 1) / (NT * 4)), (unsigned)B), dim3(NT),
                (hipStream_t)stream, pool, off, len, src, beg, shift, out, T);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_zero_front(float* x, const int* shift, int B, int T, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    PASE_LAUNCH(zero_front_kernel, dim3((unsigned)((T + NT * 4 - 1) / (NT * 4)), (unsigned)B), dim3(NT),
                (hipStream_t)stream, x, shift, T);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_peak_scale(float* x, const float* u, int N, int T, void* stream) {
    if (N <= 0 || T <= 0) return 0;
    PASE_LAUNCH(peak_scale_kernel, dim3((unsigned)N), dim3(NT), (hipStream_t)stream, x, u, T);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_fir_distort(float* x, const float* irs, const long long* ir_off, const int* ir_len,
                                const int* ir_shift, const int* ir_idx, float* full, double* energies, int B, int T,
                                int max_ir_len, int trimmed_energy, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (max_ir_len < 1) return -2;
    hipStream_t st = (hipStream_t)stream;
    const int full_stride = T + max_ir_len - 1;
    double* Ex = energies;
    double* Er = energies + B;
    if (hipMemsetAsync(Er, 0, sizeof(double) * (size_t)B, st) != hipSuccess) return -1;
    PASE_LAUNCH(row_energy_kernel, dim3((unsigned)B), dim3(NT), st, (const float*)x, T, Ex);
    PASE_LAUNCH(fir_full_kernel, dim3((unsigned)((full_stride + FIR_TILE - 1) / FIR_TILE), (unsigned)B), dim3(NT), st,
                (const float*)x, irs, ir_len, ir_off, ir_idx, full, Er, T, full_stride);
    PASE_LAUNCH(reverb_finish_kernel, dim3((unsigned)((T + NT * 4 - 1) / (NT * 4)), (unsigned)B), dim3(NT), st, x,
                (const float*)full, ir_len, ir_shift, ir_idx, (const double*)Ex, (const double*)Er, T, full_stride,
                trimmed_energy);
    if (trimmed_energy)
        PASE_LAUNCH(fir_rescale_kernel, dim3((unsigned)B), dim3(NT), st, x, ir_idx, (const double*)Ex, T);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_reverb(float* x, const float* irs, const long long* ir_off, const int* ir_len,
                           const int* ir_pmax, const int* ir_idx, float* full, double* energies, int B, int T,
                           int max_ir_len, void* stream) {
    return pase_fir_distort(x, irs, ir_off, ir_len, ir_pmax, ir_idx, full, energies, B, T, max_ir_len, 0, stream);
}

extern "C" int pase_clip(float* x, const float* factor, int B, int T, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    PASE_LAUNCH(clip_kernel, dim3((unsigned)B), dim3(NT), (hipStream_t)stream, x, factor, T);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_add_noise(float* x, const float* npool, const long long* noff, const int* nlen,
                              const int* nidx, const int* nbeg, const float* snr, int B, int T, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    PASE_LAUNCH(add_noise_kernel, dim3((unsigned)B), dim3(NT), (hipStream_t)stream, x, npool, noff, nlen, nidx, nbeg,
                snr, T);
    PASE_CHECK_LAUNCH();
    return 0;
}

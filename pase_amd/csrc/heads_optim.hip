// heads_optim.hip -- single-output worker heads with fused losses, Sinc filter synthesis and its
// gradient, the transposed weight pack used by every data-gradient, and the fused Adam update.
#include "hip_compat.h"
#include "pase_amd.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    v = pase_wave_sum64d(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += sh[w];
    return t;
}

// ---- single-output head:  y[s,t] = b + sum_c w[c] * act(z[s,c,t])  + fused loss ------------------
// Replaces the final nn.Conv1d(hidden, 1, 1) of DecoderMinion / MLPMinion (Minions/minions.py:431,
// :510) fused with nn.L1Loss (cchunk worker, cfg/workers/workers+.cfg:3-14) or
// nn.BCEWithLogitsLoss (mi / cmi workers, :112-134) as wrapped by ContextualizedLoss(r=None)
// (pase/losses.py:33-37).  A 1-row GEMM would waste the MFMA tile; this is a streaming reduction
// over channels with consecutive lanes on consecutive time steps.
__global__ void __launch_bounds__(NT) head1_fwd_kernel(const float* z, const float* in_scale, const float* in_shift,
                                                       const float* in_alpha, const float* w, const float* bias,
                                                       const float* target, float* y, float* dy, double* loss_acc,
                                                       int S, int C, int T, int loss_type, float grad_scale) {
    __shared__ double sh[NT / 64];
    const long total = (long)S * T;
    double lsum = 0.0;
    const bool small = total < 0x7fffffffL;              // uniform: 32-bit index arithmetic (a fifth of the 64-bit division)
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int s = small ? (int)((unsigned)i / (unsigned)T) : (int)(i / T);
        const int t = (int)(i - (long)s * T);
        const float* zp = z + (size_t)s * C * T + t;
        float acc = bias ? bias[0] : 0.f;
        // channels are T floats apart: eight independent loads in flight, accumulated in channel order
        int c = 0;
        for (; c + 8 <= C; c += 8) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = zp[(size_t)(c + e) * T];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float u = v[e];
                if (in_scale) u = u * in_scale[c + e] + in_shift[c + e];
                if (in_alpha) u = u > 0.f ? u : u * in_alpha[c + e];
                acc = fmaf(w[c + e], u, acc);
            }
        }
        for (; c < C; ++c) {
            float v = zp[(size_t)c * T];
            if (in_scale) v = v * in_scale[c] + in_shift[c];
            if (in_alpha) v = v > 0.f ? v : v * in_alpha[c];
            acc = fmaf(w[c], v, acc);
        }
        if (y) y[i] = acc;
        if (loss_type != PASE_LOSS_NONE) {
            const float tg = target[i];
            float l, g;
            if (loss_type == PASE_LOSS_L1) {
                const float d = acc - tg;
                l = fabsf(d);
                g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            } else if (loss_type == PASE_LOSS_MSE) {
                const float d = acc - tg;
                l = d * d;
                g = 2.f * d;
            } else {  // BCE with logits: max(x,0) - x*y + log1p(exp(-|x|))
                l = fmaxf(acc, 0.f) - acc * tg + log1pf(expf(-fabsf(acc)));
                g = 1.f / (1.f + expf(-acc)) - tg;
            }
            lsum += (double)l;
            if (dy) dy[i] = g * grad_scale;
        }
    }
    if (loss_type != PASE_LOSS_NONE) {
        lsum = block_sum_d(lsum, sh);
        if (threadIdx.x == 0) atomicAdd(loss_acc, lsum);
    }
}

// backward of the head: dz[s,c,t] = w[c]*dy[s,t]*prelu'(z);  dw[c] = sum dy*act(z);  db = sum dy;
// dalpha[c] = sum w[c]*dy*z*[z<=0]; sums = (C,3) {dw, dalpha, sum dz} followed by db at [3C].  One block per (c, chunk of s*t): reductions stay per channel.
__global__ void __launch_bounds__(NT) head1_bwd_kernel(const float* z, const float* in_alpha, const float* w,
                                                       const float* dy, float* dz, double* sums, int S, int C, int T,
                                                       int chunks) {
    __shared__ double sh[NT / 64];
    const int c = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const long total = (long)S * T;
    const long per = (total + chunks - 1) / chunks;
    const long i0 = ch * per, i1 = (i0 + per < total) ? i0 + per : total;
    const float wc = w[c];
    const float al = in_alpha ? in_alpha[c] : 1.f;
    double s_w = 0.0, s_a = 0.0, s_b = 0.0, s_z = 0.0;
    // (s, t) of element i walked incrementally: one 64-bit division per thread instead of two per element (they were most
    // of this kernel's instructions), and sum dz -- the hidden layer's conv-bias gradient -- taken in the same pass
    long i = i0 + threadIdx.x;
    int s = i < i1 ? (int)(i / T) : 0;
    int t = i < i1 ? (int)(i - (long)s * T) : 0;
    for (; i < i1; i += NT) {
        const size_t o = ((size_t)s * C + c) * (size_t)T + t;
        const float zv = z[o];
        const float g = dy[i];
        const float act = zv > 0.f ? zv : zv * al;
        const float dact = wc * g;
        const float dzv = zv > 0.f ? dact : dact * al;
        dz[o] = dzv;
        s_w += (double)(g * act);
        if (!(zv > 0.f)) s_a += (double)(dact * zv);
        s_b += (double)g;
        s_z += (double)dzv;
        t += NT;
        while (t >= T) {
            t -= T;
            ++s;
        }
    }
    s_w = block_sum_d(s_w, sh);
    s_a = block_sum_d(s_a, sh);
    s_b = block_sum_d(s_b, sh);
    s_z = block_sum_d(s_z, sh);
    if (threadIdx.x == 0) {
        atomicAdd(sums + (size_t)c * 3 + 0, s_w);
        atomicAdd(sums + (size_t)c * 3 + 1, s_a);
        atomicAdd(sums + (size_t)c * 3 + 2, s_z);
        if (c == 0) atomicAdd(sums + (size_t)C * 3, s_b);
    }
}

// ---- generic elementwise loss on a materialised prediction (API-compat path) ---------------------
// ContextualizedLoss.__call__ (pase/losses.py:33-37): target gathered with the r-context stacking
// of contextualize_r (:14-31); r_ctx <= 1 means plain (pred, target) of equal shape.
__global__ void __launch_bounds__(NT) ctx_loss_kernel(const float* pred, const float* label, float* dpred,
                                                      double* loss_acc, int B, int M, int F, int r_ctx, int label_D,
                                                      int loss_type, float grad_scale) {
    __shared__ double sh[NT / 64];
    const long total = (long)B * M * F;
    const int half = r_ctx / 2;
    double lsum = 0.0;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int t = (int)(i % F);
        const int m = (int)((i / F) % M);
        const int b = (int)(i / ((long)F * M));
        float tg;
        if (r_ctx > 1) {
            const int d = m / r_ctx, j = m - d * r_ctx;
            const int tt = t + j - half;
            tg = (tt >= 0 && tt < F) ? label[((size_t)b * label_D + d) * (size_t)F + tt] : 0.f;
        } else {
            tg = label[i];
        }
        const float x = pred[i];
        float l, g;
        if (loss_type == PASE_LOSS_L1) {
            const float d = x - tg;
            l = fabsf(d);
            g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        } else if (loss_type == PASE_LOSS_MSE) {
            const float d = x - tg;
            l = d * d;
            g = 2.f * d;
        } else {
            l = fmaxf(x, 0.f) - x * tg + log1pf(expf(-fabsf(x)));
            g = 1.f / (1.f + expf(-x)) - tg;
        }
        lsum += (double)l;
        if (dpred) dpred[i] = g * grad_scale;
    }
    lsum = block_sum_d(lsum, sh);
    if (threadIdx.x == 0) atomicAdd(loss_acc, lsum);
}

// ---- Sinc band-pass filter bank (SincConv_fast.forward, pase/models/modules.py:881-915) ----------
//   low = min_low + |low_|; high = clamp(low + min_band + |band_|, min_low, sr/2); band = high-low
//   left[n] = (sin(high*n_[n]) - sin(low*n_[n])) / (n_[n]/2) * window[n];  centre = 2*band
//   filt = [left, centre, flip(left)] / (2*band)
// n_ and window_ are the module's constant buffers (modules.py:868-876), computed host-side with the
// reference's own fp32 expressions (quirks: linspace(0, K/2-1, K//2), division by K not K-1).
__global__ void __launch_bounds__(NT) sinc_filters_kernel(const float* low_hz_, const float* band_hz_,
                                                          const float* n_, const float* window_, float* filt, int C,
                                                          int Kw, float min_low, float min_band, float sr) {
    const int c = blockIdx.x;
    const int half = (Kw - 1) / 2;
    const float low = min_low + fabsf(low_hz_[c]);
    float high = low + min_band + fabsf(band_hz_[c]);
    high = fminf(fmaxf(high, min_low), sr * 0.5f);
    const float band = high - low;
    for (int k = threadIdx.x; k < Kw; k += NT) {
        float v;
        if (k == half) {
            v = 2.f * band;
        } else {
            const int n = k < half ? k : Kw - 1 - k;
            const float nn = n_[n];
            v = ((sinf(high * nn) - sinf(low * nn)) / (nn / 2.f)) * window_[n];
        }
        filt[(size_t)c * Kw + k] = v / (2.f * band);
    }
}

// gradient of the above w.r.t. (low_hz_, band_hz_) given dF (C, Kw): one block per filter
__global__ void __launch_bounds__(NT) sinc_filters_bwd_kernel(const float* low_hz_, const float* band_hz_,
                                                              const float* n_, const float* window_,
                                                              const float* dfilt, float* dlow, float* dband, int C,
                                                              int Kw, float min_low, float min_band, float sr) {
    __shared__ double sh[NT / 64];
    const int c = blockIdx.x;
    const int half = (Kw - 1) / 2;
    const float lraw = low_hz_[c], braw = band_hz_[c];
    const float low = min_low + fabsf(lraw);
    const float hpre = low + min_band + fabsf(braw);
    const float high = fminf(fmaxf(hpre, min_low), sr * 0.5f);
    const bool pass = hpre >= min_low && hpre <= sr * 0.5f;   // torch.clamp passes grad inside [min, max]
    const double band = (double)high - (double)low;
    double g_h = 0.0, g_l = 0.0;
    for (int n = threadIdx.x; n < half; n += NT) {
        const double gF = (double)dfilt[(size_t)c * Kw + n] + (double)dfilt[(size_t)c * Kw + (Kw - 1 - n)];
        const double nn = (double)n_[n];
        const double u = (2.0 / nn) * (double)window_[n];
        const double sh_ = sin((double)high * nn), sl_ = sin((double)low * nn);
        const double num = (sh_ - sl_) * u;
        // F = num / (2*band), band = high - low
        g_h += gF * (cos((double)high * nn) * nn * u / (2.0 * band) - num / (2.0 * band * band));
        g_l += gF * (-cos((double)low * nn) * nn * u / (2.0 * band) + num / (2.0 * band * band));
    }
    g_h = block_sum_d(g_h, sh);
    g_l = block_sum_d(g_l, sh);
    if (threadIdx.x == 0) {
        const double gh = pass ? g_h : 0.0;
        const float sl = lraw > 0.f ? 1.f : (lraw < 0.f ? -1.f : 0.f);
        const float sb = braw > 0.f ? 1.f : (braw < 0.f ? -1.f : 0.f);
        dlow[c] = (float)((g_l + gh) * sl);
        dband[c] = (float)(gh * sb);
    }
}

// ---- transposed weight pack for data-gradients / transposed convolutions -------------------------
//   dst[(p*O + o), (r*? ...)]:  dst[(p, o), (red, j)] = src[red, o, p + st*j]  (0 beyond k)
// with arbitrary source strides, so nn.Conv1d (out,in,k), nn.ConvTranspose1d (in,out,k) and the
// tap-major QRNN Linear (3H, 2*Cin) all use the same kernel.
__global__ void __launch_bounds__(NT) pack_dgrad_kernel(const float* src, float* dst, int R, int O, int k, int st,
                                                        int taps_p, long s_red, long s_out, long s_k) {
    const long total = (long)st * O * R * taps_p;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int j = (int)(i % taps_p);
        const int red = (int)((i / taps_p) % R);
        const long po = i / ((long)taps_p * R);
        const int o = (int)(po % O), ph = (int)(po / O);
        const int kk = ph + st * j;
        dst[i] = kk < k ? src[red * s_red + o * s_out + kk * s_k] : 0.f;
    }
}

// K-major variant for pase_conv_gemm's `wt` operand: dst[(red*taps_p + j)*ldt + (p*O + o)] = src[red, o, p + st*j];
// consecutive threads walk the (p, o) index, so writes are coalesced; pad columns up to ldt are zero-filled.
__global__ void __launch_bounds__(NT) pack_dgrad_t_kernel(const float* src, float* dst, int R, int O, int k, int st,
                                                          int taps_p, long s_red, long s_out, long s_k, int ldt) {
    const long total = (long)R * taps_p * ldt;
    const int M = st * O;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int m = (int)(i % ldt);
        const long kr = i / ldt;
        const int j = (int)(kr % taps_p);
        const int red = (int)(kr / taps_p);
        float v = 0.f;
        if (m < M) {
            const int ph = m / O, o = m - ph * O;
            const int kk = ph + st * j;
            if (kk < k) v = src[red * s_red + o * s_out + kk * s_k];
        }
        dst[i] = v;
    }
}

// Parameter-gradient commit: g_k[c] += (float)sums[c*ld + col_k] for up to three (buffer, column) pairs in one launch
// (the fp64 per-channel sums of pase_act_bwd_reduce / pase_head1_bwd -> dbeta / dgamma / dalpha / dbias buffers).
__global__ void __launch_bounds__(NT) commit_cols_kernel(const double* sums, int ld, int C, float* g0, int c0, float* g1,
                                                         int c1, float* g2, int c2) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    const double* row = sums + (size_t)c * ld;
    if (g0) g0[c] += (float)row[c0];
    if (g1) g1[c] += (float)row[c1];
    if (g2) g2[c] += (float)row[c2];
}

// Parameter-gradient commit of STAGED weight gradients: dst_k[r, c] += src_k[r, c] for up to 16 row-major blocks in ONE launch
// (the concatenated W + dense-skip weight gradient -> its eight parameters' .grad buffers: column slices; the stacked first
// layers of the MLP heads -> nine parameters: row slices).  torch's _foreach_add_ falls back to one strided add per slice.
__global__ void __launch_bounds__(NT) add_blocks_kernel(PaseAddBlocks p) {
    const PaseAddBlock b = p.seg[blockIdx.y];
    const long n = (long)b.rows * b.width;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const int r = (int)(i / b.width), c = (int)(i - (long)r * b.width);
        b.dst[(size_t)r * b.dst_ld + c] += b.src[(size_t)r * b.src_ld + c];
    }
}

// ---- Adam (torch.optim.Adam defaults; WorkerScheduler/trainer.py:91,111,134) ----------------------
// One launch per logical optimizer over its flat parameter / gradient / moment buffers.  `step` and
// `lr` live in device memory so a captured hipGraph replays correctly as they change.
__global__ void __launch_bounds__(NT) adam_kernel(float* p, const float* g, float* m, float* v, long n,
                                                  const float* lr_p, const int* step_p, float beta1, float beta2,
                                                  float eps, float grad_mul) {
    const float lr = lr_p[0];
    const int step = step_p[0];
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    const float step_size = lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float gi = g[i] * grad_mul;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

__global__ void step_tick_kernel(int* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1;
}

}  // namespace

static unsigned grid_for(long total) {
    long b = (total + NT - 1) / NT;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

extern "C" int pase_head1_fwd(const float* z, const float* in_scale, const float* in_shift, const float* in_alpha,
                              const float* w, const float* bias, const float* target, float* y, float* dy,
                              double* loss_acc, int S, int C, int T, int loss_type, float grad_scale, void* stream) {
    if (loss_type != PASE_LOSS_NONE && (!target || !loss_acc)) return -2;
    const long total = (long)S * T;
    if (total <= 0) return 0;
    PASE_LAUNCH(head1_fwd_kernel, dim3(grid_for(total)), dim3(NT), (hipStream_t)stream, z, in_scale, in_shift,
                in_alpha, w, bias, target, y, dy, loss_acc, S, C, T, loss_type, grad_scale);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_head1_bwd(const float* z, const float* in_alpha, const float* w, const float* dy, float* dz,
                              double* sums, int S, int C, int T, void* stream) {
    const long total = (long)S * T;
    if (total <= 0 || C <= 0) return 0;
    int chunks = (int)((total + 16383) / 16384);
    if (chunks < 1) chunks = 1;
    PASE_LAUNCH(head1_bwd_kernel, dim3((unsigned)(C * chunks)), dim3(NT), (hipStream_t)stream, z, in_alpha, w, dy, dz,
                sums, S, C, T, chunks);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_ctx_loss(const float* pred, const float* label, float* dpred, double* loss_acc, int B, int M,
                             int F, int r_ctx, int label_D, int loss_type, float grad_scale, void* stream) {
    const long total = (long)B * M * F;
    if (total <= 0) return 0;
    if (loss_type == PASE_LOSS_NONE || !loss_acc) return -2;
    PASE_LAUNCH(ctx_loss_kernel, dim3(grid_for(total)), dim3(NT), (hipStream_t)stream, pred, label, dpred, loss_acc, B,
                M, F, r_ctx, label_D, loss_type, grad_scale);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_sinc_filters(const float* low_hz_, const float* band_hz_, const float* n_, const float* window_,
                                 float* filt, int C, int Kw, float min_low, float min_band, float sr, void* stream) {
    if (C <= 0) return 0;
    if ((Kw & 1) == 0) return -2;
    PASE_LAUNCH(sinc_filters_kernel, dim3(C), dim3(NT), (hipStream_t)stream, low_hz_, band_hz_, n_, window_, filt, C,
                Kw, min_low, min_band, sr);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_sinc_filters_bwd(const float* low_hz_, const float* band_hz_, const float* n_,
                                     const float* window_, const float* dfilt, float* dlow, float* dband, int C,
                                     int Kw, float min_low, float min_band, float sr, void* stream) {
    if (C <= 0) return 0;
    PASE_LAUNCH(sinc_filters_bwd_kernel, dim3(C), dim3(NT), (hipStream_t)stream, low_hz_, band_hz_, n_, window_, dfilt,
                dlow, dband, C, Kw, min_low, min_band, sr);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_pack_dgrad(const float* src, float* dst, int R, int O, int k, int st, long s_red, long s_out,
                               long s_k, void* stream) {
    const int taps_p = (k + st - 1) / st;
    const long total = (long)st * O * R * taps_p;
    if (total <= 0) return 0;
    PASE_LAUNCH(pack_dgrad_kernel, dim3(grid_for(total)), dim3(NT), (hipStream_t)stream, src, dst, R, O, k, st,
                taps_p, s_red, s_out, s_k);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_pack_dgrad_t(const float* src, float* dst, int R, int O, int k, int st, long s_red, long s_out,
                                 long s_k, int ldt, void* stream) {
    const int taps_p = (k + st - 1) / st;
    if (ldt < st * O || (ldt & 3)) return -4;
    const long total = (long)R * taps_p * ldt;
    if (total <= 0) return 0;
    PASE_LAUNCH(pack_dgrad_t_kernel, dim3(grid_for(total)), dim3(NT), (hipStream_t)stream, src, dst, R, O, k, st,
                taps_p, s_red, s_out, s_k, ldt);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_commit_cols(const double* sums, int ld, int C, float* g0, int c0, float* g1, int c1, float* g2,
                                int c2, void* stream) {
    if (C <= 0) return 0;
    PASE_LAUNCH(commit_cols_kernel, dim3((unsigned)((C + NT - 1) / NT)), dim3(NT), (hipStream_t)stream, sums, ld, C, g0,
                c0, g1, c1, g2, c2);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_add_blocks(const PaseAddBlocks* d, void* stream) {
    if (d->n <= 0) return 0;
    if (d->n > 16) return -2;
    long nmax = 0;
    for (int k = 0; k < d->n; ++k) {
        const PaseAddBlock& b = d->seg[k];
        if (b.rows < 0 || b.width < 0 || b.src_ld < b.width || b.dst_ld < b.width || !b.src || !b.dst) return -2;
        const long n = (long)b.rows * b.width;
        nmax = n > nmax ? n : nmax;
    }
    if (nmax == 0) return 0;
    long gx = (nmax + NT - 1) / NT;
    if (gx > 1024) gx = 1024;
    PASE_LAUNCH(add_blocks_kernel, dim3((unsigned)gx, (unsigned)d->n), dim3(NT), (hipStream_t)stream, *d);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_adam_step(float* p, const float* g, float* m, float* v, long n, const float* lr,
                              const int* step, float beta1, float beta2, float eps, float grad_mul, void* stream) {
    if (n <= 0) return 0;
    PASE_LAUNCH(adam_kernel, dim3(grid_for(n)), dim3(NT), (hipStream_t)stream, p, g, m, v, n, lr, step, beta1, beta2,
                eps, grad_mul);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_step_tick(int* step, void* stream) {
    PASE_LAUNCH(step_tick_kernel, dim3(1), dim3(64), (hipStream_t)stream, step);
    PASE_CHECK_LAUNCH();
    return 0;
}

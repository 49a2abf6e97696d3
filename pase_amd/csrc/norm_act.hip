// norm_act.hip -- the HBM-bound kernels around the contractions: BatchNorm statistics finalisation,
// dense-skip mean-pooling, normalise(+PReLU) materialisation, and the BatchNorm+PReLU backward
// (reduce + apply) with the reflect-pad fold and the pooled dense-skip gradient merged in.
//
// Reference semantics: nn.BatchNorm1d in training mode (pase/models/modules.py:79 via
// build_norm_layer, applied in FeBlock.forward :1072-1074; frontend.py:206-210 norm_out with
// affine=False), nn.PReLU (modules.py:111-113), fuse_skip's view(...).mean(3)
// (pase/models/frontend.py:213-232) and F.pad(mode='reflect') (modules.py:1061-1071), whose
// autograd backward adds the mirrored edge gradients back onto the interior.
//
// Roofline: every kernel here streams each tensor once with consecutive lanes on consecutive
// addresses; they are priced against HBM bandwidth (bytes = 4 * elements touched).
#include "act_bwd.h"
#include "hip_compat.h"
#include "pase_amd.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    // sh: >= NT/64 doubles of LDS
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

// ---- E1: finalise batch statistics -> on-load affine (scale, shift), running stats -------------
__global__ void __launch_bounds__(NT) bn_finalize_kernel(const float* stat_part, int ntiles, int C, double count,
                                                         const float* gamma, const float* beta, float eps,
                                                         float momentum, float* running_mean, float* running_var,
                                                         float* scale, float* shift, float* mean_out,
                                                         float* rstd_out) {
    __shared__ double sh[NT / 64];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += NT) {
        const float* q = stat_part + ((size_t)t * C + c) * 2;
        s1 += (double)q[0];
        s2 += (double)q[1];
    }
    s1 = block_sum_d(s1, sh);
    s2 = block_sum_d(s2, sh);
    if (threadIdx.x == 0) {
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double g = gamma ? (double)gamma[c] : 1.0;
        const double b = beta ? (double)beta[c] : 0.0;
        scale[c] = (float)(g * rstd);
        shift[c] = (float)(b - mean * g * rstd);
        mean_out[c] = (float)mean;
        rstd_out[c] = (float)rstd;
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
        }
    }
}

// ---- E2: dense-skip pooling  P[s, c, f] = mean_i act(bn(y[s, c, f*d + i])) ----------------------
// G lanes cooperate on one output (16 lanes x 64-B segments for the 160-sample windows of the 786 MB block-0 tensor)
template <int G>
__global__ void __launch_bounds__(NT) bn_act_pool_kernel(const float* y, float* out, const float* scale,
                                                         const float* shift, const float* alpha, int S, int C,
                                                         int T, int F, int d, int o_ctot, int o_coff) {
    // (a (row, chunk) grid without the per-thread 64-bit divisions measured SLOWER -- 143 vs 112 us on the 786 MB layer:
    //  consecutive workgroups then read rows 128 KB apart instead of consecutive memory)
    const long nout = (long)S * C * F;
    const long gid = ((long)blockIdx.x * NT + threadIdx.x) / G;
    const int gl = threadIdx.x % G;
    const bool ok = gid < nout;
    const long g = ok ? gid : 0;
    const int f = (int)(g % F);
    const long sc_ = g / F;
    const int c = (int)(sc_ % C);
    const int s = (int)(sc_ / C);
    const float a = scale ? scale[c] : 1.f, b = shift ? shift[c] : 0.f;
    const float al = alpha ? alpha[c] : 1.f;
    const float* row = y + ((size_t)s * C + c) * (size_t)T + (size_t)f * d;
    float acc = 0.f;
    if (ok)
        for (int i = gl; i < d; i += G) {
            float v = row[i] * a + b;
            v = v > 0.f ? v : v * al;
            acc += v;
        }
    if (G > 1) {
#pragma unroll
        for (int m = 1; m < G; m <<= 1) acc += __shfl_xor(acc, m);
    }
    if (ok && gl == 0) out[((size_t)s * o_ctot + o_coff + c) * (size_t)F + f] = acc / (float)d;
}

// ---- E3: materialise  out = act(bn(y))  (public outputs: the embedding; API-compat activations) --
__global__ void __launch_bounds__(NT) bn_act_apply_kernel(const float* y, float* out, const float* scale,
                                                          const float* shift, const float* alpha, int C, int T,
                                                          long total) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)((i / T) % C);
        float v = y[i];
        if (scale) v = v * scale[c] + shift[c];
        if (alpha) v = v > 0.f ? v : v * alpha[c];
        out[i] = v;
    }
}

// ---- E4 / E5 work decomposition -------------------------------------------------------------------
// A unit of work is a SEGMENT: `seg_len` consecutive time steps of one (sequence, channel) row.  Long rows
// (T >= 2048) are cut into segments of <= 8192 elements handled by a whole 256-thread block; short rows (the
// 200 ... 1600-frame layers, 49 152 rows each) are handled one row per WAVE, four rows per block, so that no
// barrier and no cross-wave reduction sits behind a 200-element loop.  Per-thread partial sums are fp32 over
// <= 32 elements (4 independent chains, loads of all four in flight), widened to fp64 for the lane / wave /
// global reduction.
struct ActBwdGrid { int waves_per_seg, segs_per_row, seg_len; long nseg; };

__device__ __forceinline__ void act_bwd_locate(const PaseActBwd& p, const ActBwdGrid& g, int& row, int& t0, int& t1,
                                               int& lane_id, int& nlanes) {
    if (g.waves_per_seg == 4) {                 // block per segment
        const long seg = blockIdx.x;
        row = (int)(seg / g.segs_per_row);
        const int k = (int)(seg % g.segs_per_row);
        t0 = k * g.seg_len;
        t1 = min(p.T, t0 + g.seg_len);
        lane_id = threadIdx.x;
        nlanes = NT;
    } else {                                    // wave per row
        const long seg = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
        row = seg < g.nseg ? (int)seg : -1;
        t0 = 0;
        t1 = p.T;
        lane_id = threadIdx.x & 63;
        nlanes = 64;
    }
}

// ---- E4: reduce pass.  sums[c] = { sum dz, sum dz*xhat, sum dA*z*[z<=0] } ------------------------
__global__ void __launch_bounds__(NT) act_bwd_reduce_kernel(PaseActBwd p, ActBwdGrid g) {
    const unsigned pmagic = act_pool_magic(p);
    __shared__ double sh[NT / 64];
    int row, t0, t1, lid, nl;
    act_bwd_locate(p, g, row, t0, t1, lid, nl);
    const bool live = row >= 0;
    const int s = live ? row / p.C : 0, c = live ? row % p.C : 0;
    const float a = p.scale ? p.scale[c] : 1.f, b = p.shift ? p.shift[c] : 0.f;
    const float al = p.alpha ? p.alpha[c] : 1.f;
    const float mean = p.mean ? p.mean[c] : 0.f, rstd = p.rstd ? p.rstd[c] : 1.f;
    const float* yrow = p.y + ((size_t)s * p.y_ctot + p.y_coff + c) * (size_t)p.T;
    // without a BatchNorm (has_bn 0: dy = dz) or behind a FROZEN one (has_bn 2, eval-mode statistics: dy = scale * dz)
    // dy does not depend on the sums: written here, no apply pass (one read of y / dA less)
    float* drow = (p.has_bn != 1 && p.dy) ? p.dy + ((size_t)s * p.y_ctot + p.y_coff + c) * (size_t)p.T : nullptr;
    const float dmul = p.has_bn == 2 ? a : 1.f;
    float f_dz[4] = {0.f, 0.f, 0.f, 0.f}, f_dzx[4] = {0.f, 0.f, 0.f, 0.f}, f_da[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        // all eight loads of a round first (as in the apply pass), then the arithmetic: with the loads inside the per-element
        // branch the pass ran one memory latency per element and thread -- 1.1 TB/s on the 786 MB SincNet output against the
        // apply pass's 4.7 TB/s over the same two tensors (profiles/bench_r04_kernel_stats.csv).  Elements past the end read as
        // y = 0, dA = 0: they add nothing to the three sums.
        for (int tb = t0 + lid; tb < t1; tb += 4 * nl) {
            float yv[4], dA[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = tb + u * nl;
                yv[u] = t < t1 ? yrow[t] : 0.f;
                dA[u] = t < t1 ? grad_post_act(p, s, c, t, pmagic) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = tb + u * nl;
                const float z = yv[u] * a + b;
                const float dz = z > 0.f ? dA[u] : dA[u] * al;
                if (drow && t < t1) drow[t] = dz * dmul;
                const float xhat = (yv[u] - mean) * rstd;
                f_dz[u] += dz;
                f_dzx[u] = fmaf(dz, xhat, f_dzx[u]);
                // (select on the PRODUCT: 0 * z is NaN for z = +Inf, and the reference's prelu backward adds nothing there;
                //  the padded tail has dA = 0 and a finite z = b)
                f_da[u] += z > 0.f ? 0.f : dA[u] * z;
            }
        }
    }
    double s_dz = ((double)f_dz[0] + (double)f_dz[1]) + ((double)f_dz[2] + (double)f_dz[3]);
    double s_dzx = ((double)f_dzx[0] + (double)f_dzx[1]) + ((double)f_dzx[2] + (double)f_dzx[3]);
    double s_da = ((double)f_da[0] + (double)f_da[1]) + ((double)f_da[2] + (double)f_da[3]);
    if (g.waves_per_seg == 4) {
        s_dz = block_sum_d(s_dz, sh);
        s_dzx = block_sum_d(s_dzx, sh);
        s_da = block_sum_d(s_da, sh);
        if (threadIdx.x != 0) return;
    } else {
        s_dz = pase_wave_sum64d(s_dz);
        s_dzx = pase_wave_sum64d(s_dzx);
        s_da = pase_wave_sum64d(s_da);
        if ((threadIdx.x & 63) != 0 || !live) return;
    }
    atomicAdd(p.sums + (size_t)c * 3 + 0, s_dz);
    atomicAdd(p.sums + (size_t)c * 3 + 1, s_dzx);
    atomicAdd(p.sums + (size_t)c * 3 + 2, s_da);
}

// ---- E5: apply pass.  dy = scale * (dz - mean(dz) - xhat * mean(dz*xhat))   (BN)  or  dy = dz ----
__global__ void __launch_bounds__(NT) act_bwd_apply_kernel(PaseActBwd p, ActBwdGrid g) {
    const unsigned pmagic = act_pool_magic(p);
    int row, t0, t1, lid, nl;
    act_bwd_locate(p, g, row, t0, t1, lid, nl);
    if (row < 0) return;
    const int s = row / p.C, c = row % p.C;
    const ActBwdRow rc = act_bwd_row(p, c);
    const float* yrow = p.y + ((size_t)s * p.y_ctot + p.y_coff + c) * (size_t)p.T;
    float* drow = p.dy + ((size_t)s * p.y_ctot + p.y_coff + c) * (size_t)p.T;
    for (int tb = t0 + lid; tb < t1; tb += 4 * nl) {
        float yv[4], dA[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tb + u * nl;
            yv[u] = t < t1 ? yrow[t] : 0.f;
            dA[u] = t < t1 ? grad_post_act(p, s, c, t, pmagic) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tb + u * nl;
            if (t < t1) drow[t] = act_bwd_dy(rc, p.has_bn, yv[u], dA[u]);
        }
    }
}

// =====================================================================================================
// Per-sample normalisations: nn.InstanceNorm1d (statistics per (sequence, channel) over time; norm_type 'inorm' /
// 'affinorm', and WaveFe.norm_out when norm_type != 'bnorm', pase/models/frontend.py:206-210) and nn.LayerNorm(C)
// applied on the transposed tensor (statistics per (sequence, time step) over channels; norm_type 'lnorm',
// pase/models/modules.py:85-86,98-105).  Their scale / shift depend on the sample, so they cannot ride on the
// consumer's per-channel on-load transform like BatchNorm: the activated tensor a = PReLU(gamma * xhat + beta) is
// materialised once (consumers then load it untransformed) together with the group statistics for the backward.
// Not on the benchmark path (PASE+.cfg is 'bnorm'): written for clarity, two passes over the group.
// =====================================================================================================
constexpr int LN_TT = 64;      // time steps per block in layer mode (lanes along time: coalesced rows)

__device__ __forceinline__ float block_sum_f(float v, float* sh) {
    v = pase_wave_sum64(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += sh[w];
    return t;
}

// mode 0: instance (block = one (s, c) row);  mode 1: layer (block = LN_TT time steps of one sequence)
__global__ void __launch_bounds__(NT) rownorm_act_fwd_kernel(const float* y, float* out, const float* gamma,
                                                             const float* beta, const float* alpha, float* mean_out,
                                                             float* rstd_out, int S, int C, int T, float eps, int mode) {
    __shared__ float sh[NT / 64];
    __shared__ float sm[LN_TT][NT / 64 + 1], sq[LN_TT][NT / 64 + 1];
    __shared__ float s_mean[LN_TT], s_rstd[LN_TT];
    if (mode == 0) {
        const int row = blockIdx.x;
        const int c = row % C;
        const float* yr = y + (size_t)row * T;
        float* orow = out + (size_t)row * T;
        float a1 = 0.f;
        for (int t = threadIdx.x; t < T; t += NT) a1 += yr[t];
        const float mean = block_sum_f(a1, sh) / (float)T;
        float a2 = 0.f;
        for (int t = threadIdx.x; t < T; t += NT) { const float d = yr[t] - mean; a2 = fmaf(d, d, a2); }
        const float var = block_sum_f(a2, sh) / (float)T;          // biased, like torch
        const float rstd = 1.f / sqrtf(var + eps);
        if (threadIdx.x == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        for (int t = threadIdx.x; t < T; t += NT) {
            float v = (yr[t] - mean) * rstd * g + b;
            if (alpha) v = v > 0.f ? v : v * alpha[c];
            orow[t] = v;
        }
        return;
    }
    const int tiles = (T + LN_TT - 1) / LN_TT;
    const int s = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * LN_TT;
    const int tl = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int t = t0 + tl;
    const bool ok = t < T;
    const float* yb = y + (size_t)s * C * T + t;
    float a1 = 0.f;
    for (int c = cg; c < C; c += NT / 64) a1 += ok ? yb[(size_t)c * T] : 0.f;
    sm[tl][cg] = a1;
    __syncthreads();
    if (cg == 0) {
        float m = 0.f;
        for (int w = 0; w < NT / 64; ++w) m += sm[tl][w];
        s_mean[tl] = m / (float)C;
    }
    __syncthreads();
    const float mean = s_mean[tl];
    float a2 = 0.f;
    for (int c = cg; c < C; c += NT / 64) { const float d = ok ? yb[(size_t)c * T] - mean : 0.f; a2 = fmaf(d, d, a2); }
    sq[tl][cg] = a2;
    __syncthreads();
    if (cg == 0) {
        float v = 0.f;
        for (int w = 0; w < NT / 64; ++w) v += sq[tl][w];
        const float rstd = 1.f / sqrtf(v / (float)C + eps);
        s_rstd[tl] = rstd;
        if (ok) { mean_out[(size_t)s * T + t] = mean; rstd_out[(size_t)s * T + t] = rstd; }
    }
    __syncthreads();
    const float rstd = s_rstd[tl];
    float* ob = out + (size_t)s * C * T + t;
    for (int c = cg; c < C; c += NT / 64) {
        if (!ok) continue;
        float v = (yb[(size_t)c * T] - mean) * rstd * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
        if (alpha) v = v > 0.f ? v : v * alpha[c];
        ob[(size_t)c * T] = v;
    }
}

// Backward of a = PReLU(gamma * xhat + beta), xhat = (y - mean_g) * rstd_g over group g (p.has_bn: 3 = instance,
// 4 = layer; p.scale / p.shift carry gamma / beta, p.mean / p.rstd the group statistics):
//   dz = dA * prelu'(z);  dxh = dz * gamma;  dy = rstd_g * (dxh - mean_g(dxh) - xhat * mean_g(dxh * xhat))
//   sums[c] += { sum dz (dbeta), sum dz * xhat (dgamma), sum dA * z * [z <= 0] (dalpha) }
__global__ void __launch_bounds__(NT) rownorm_act_bwd_kernel(PaseActBwd p) {
    const unsigned pmagic = act_pool_magic(p);
    __shared__ float sh[NT / 64];
    __shared__ float r1[LN_TT][NT / 64 + 1], r2[LN_TT][NT / 64 + 1];
    __shared__ float m1s[LN_TT], m2s[LN_TT];
    if (p.has_bn == 3) {
        const int row = blockIdx.x;
        const int s = row / p.C, c = row % p.C;
        const float g = p.scale ? p.scale[c] : 1.f, b = p.shift ? p.shift[c] : 0.f;
        const float al = p.alpha ? p.alpha[c] : 1.f;
        const float mean = p.mean[row], rstd = p.rstd[row];
        const float* yr = p.y + ((size_t)s * p.y_ctot + p.y_coff + c) * (size_t)p.T;
        float* dr = p.dy + ((size_t)s * p.y_ctot + p.y_coff + c) * (size_t)p.T;
        float s1 = 0.f, s2 = 0.f, sa = 0.f;
        for (int t = threadIdx.x; t < p.T; t += NT) {
            const float xh = (yr[t] - mean) * rstd;
            const float z = xh * g + b;
            const float dA = grad_post_act(p, s, c, t, pmagic);
            const float dz = z > 0.f ? dA : dA * al;
            s1 += dz;
            s2 = fmaf(dz, xh, s2);
            if (!(z > 0.f)) sa = fmaf(dA, z, sa);
        }
        s1 = block_sum_f(s1, sh);
        s2 = block_sum_f(s2, sh);
        sa = block_sum_f(sa, sh);
        if (threadIdx.x == 0) {
            atomicAdd(p.sums + (size_t)c * 3 + 0, (double)s1);
            atomicAdd(p.sums + (size_t)c * 3 + 1, (double)s2);
            atomicAdd(p.sums + (size_t)c * 3 + 2, (double)sa);
        }
        const float m1 = s1 * g / (float)p.T, m2 = s2 * g / (float)p.T;
        for (int t = threadIdx.x; t < p.T; t += NT) {
            const float xh = (yr[t] - mean) * rstd;
            const float z = xh * g + b;
            const float dA = grad_post_act(p, s, c, t, pmagic);
            const float dz = z > 0.f ? dA : dA * al;
            dr[t] = rstd * (dz * g - m1 - xh * m2);
        }
        return;
    }
    const int tiles = (p.T + LN_TT - 1) / LN_TT;
    const int s = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * LN_TT;
    const int tl = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int t = t0 + tl;
    const bool ok = t < p.T;
    const float mean = ok ? p.mean[(size_t)s * p.T + t] : 0.f, rstd = ok ? p.rstd[(size_t)s * p.T + t] : 0.f;
    float a1 = 0.f, a2 = 0.f;
    for (int c = cg; c < p.C; c += NT / 64) {
        float dz = 0.f, xh = 0.f, da = 0.f;
        if (ok) {
            const float yv = p.y[((size_t)s * p.y_ctot + p.y_coff + c) * (size_t)p.T + t];
            const float g = p.scale ? p.scale[c] : 1.f, b = p.shift ? p.shift[c] : 0.f;
            xh = (yv - mean) * rstd;
            const float z = xh * g + b;
            const float dA = grad_post_act(p, s, c, t, pmagic);
            dz = z > 0.f ? dA : dA * (p.alpha ? p.alpha[c] : 1.f);
            if (!(z > 0.f)) da = dA * z;
            a1 = fmaf(dz, g, a1);
            a2 = fmaf(dz * g, xh, a2);
        }
        // per-channel parameter gradients: reduce this wave's 64 time steps, one atomic per (block, channel)
        const float q0 = pase_wave_sum64(dz), q1 = pase_wave_sum64(dz * xh), q2 = pase_wave_sum64(da);
        if (tl == 0) {
            atomicAdd(p.sums + (size_t)c * 3 + 0, (double)q0);
            atomicAdd(p.sums + (size_t)c * 3 + 1, (double)q1);
            atomicAdd(p.sums + (size_t)c * 3 + 2, (double)q2);
        }
    }
    r1[tl][cg] = a1;
    r2[tl][cg] = a2;
    __syncthreads();
    if (cg == 0) {
        float u = 0.f, v = 0.f;
        for (int w = 0; w < NT / 64; ++w) { u += r1[tl][w]; v += r2[tl][w]; }
        m1s[tl] = u / (float)p.C;
        m2s[tl] = v / (float)p.C;
    }
    __syncthreads();
    const float m1 = m1s[tl], m2 = m2s[tl];
    for (int c = cg; c < p.C; c += NT / 64) {
        if (!ok) continue;
        const size_t o = ((size_t)s * p.y_ctot + p.y_coff + c) * (size_t)p.T + t;
        const float g = p.scale ? p.scale[c] : 1.f, b = p.shift ? p.shift[c] : 0.f;
        const float xh = (p.y[o] - mean) * rstd;
        const float z = xh * g + b;
        const float dA = grad_post_act(p, s, c, t, pmagic);
        const float dz = z > 0.f ? dA : dA * (p.alpha ? p.alpha[c] : 1.f);
        p.dy[o] = rstd * (dz * g - m1 - xh * m2);
    }
}

}  // namespace

extern "C" int pase_bn_finalize(const float* stat_part, int ntiles, int C, double count, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean,
                                float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out,
                                void* stream) {
    if (C <= 0) return 0;
    PASE_LAUNCH(bn_finalize_kernel, dim3(C), dim3(NT), (hipStream_t)stream, stat_part, ntiles, C, count, gamma,
                beta, eps, momentum, running_mean, running_var, scale, shift, mean_out, rstd_out);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_bn_act_pool(const float* y, float* out, const float* scale, const float* shift,
                                const float* alpha, int S, int C, int T, int F, int d, int o_ctot, int o_coff,
                                void* stream) {
    if (F * d > T) return -2;
    const long nout = (long)S * C * F;
    if (nout <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (d >= 512) {      // (one wave per output only pays for long windows: 160-sample windows run 4 outputs per wave)
        const long blocks = (nout * 64 + NT - 1) / NT;
        PASE_LAUNCH((bn_act_pool_kernel<64>), dim3((unsigned)blocks), dim3(NT), st, y, out, scale, shift, alpha, S, C, T, F, d, o_ctot, o_coff);
    } else if (d >= 16) {
        const long blocks = (nout * 16 + NT - 1) / NT;
        PASE_LAUNCH((bn_act_pool_kernel<16>), dim3((unsigned)blocks), dim3(NT), st, y, out, scale, shift, alpha, S, C, T, F, d, o_ctot, o_coff);
    } else {
        const long blocks = (nout + NT - 1) / NT;
        PASE_LAUNCH((bn_act_pool_kernel<1>), dim3((unsigned)blocks), dim3(NT), st, y, out, scale, shift, alpha, S, C, T, F, d, o_ctot, o_coff);
    }
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_bn_act_apply(const float* y, float* out, const float* scale, const float* shift,
                                 const float* alpha, int S, int C, int T, void* stream) {
    const long total = (long)S * C * T;
    if (total <= 0) return 0;
    long blocks = (total + NT - 1) / NT;
    if (blocks > 256 * 16) blocks = 256 * 16;
    PASE_LAUNCH(bn_act_apply_kernel, dim3((unsigned)blocks), dim3(NT), (hipStream_t)stream, y, out, scale, shift,
                alpha, C, T, total);
    PASE_CHECK_LAUNCH();
    return 0;
}

static ActBwdGrid act_bwd_grid(const PaseActBwd& p) {
    ActBwdGrid g;
    const long rows = (long)p.S * p.C;
    if (p.T >= 2048) {
        g.waves_per_seg = 4;
        g.segs_per_row = (p.T + 8191) / 8192;
        g.seg_len = (p.T + g.segs_per_row - 1) / g.segs_per_row;
        g.nseg = rows * g.segs_per_row;
    } else {
        g.waves_per_seg = 1;
        g.segs_per_row = 1;
        g.seg_len = p.T;
        g.nseg = rows;
    }
    return g;
}

static unsigned act_bwd_blocks(const ActBwdGrid& g) {
    return (unsigned)(g.waves_per_seg == 4 ? g.nseg : (g.nseg + NT / 64 - 1) / (NT / 64));
}

extern "C" int pase_act_bwd_reduce(const PaseActBwd* d, void* stream) {
    const PaseActBwd p = *d;
    if (!p.sums || !p.y) return -2;
    const long rows = (long)p.S * p.C;
    if (rows <= 0 || p.T <= 0) return 0;
    const ActBwdGrid g = act_bwd_grid(p);
    if (g.nseg >= 0x7fffffffL) return -8;
    PASE_LAUNCH(act_bwd_reduce_kernel, dim3(act_bwd_blocks(g)), dim3(NT), (hipStream_t)stream, p, g);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_act_bwd_apply(const PaseActBwd* d, void* stream) {
    const PaseActBwd p = *d;
    if (!p.dy || !p.y || (p.has_bn == 1 && !p.sums)) return -2;
    const long rows = (long)p.S * p.C;
    if (rows <= 0 || p.T <= 0) return 0;
    const ActBwdGrid g = act_bwd_grid(p);
    if (g.nseg >= 0x7fffffffL) return -8;
    PASE_LAUNCH(act_bwd_apply_kernel, dim3(act_bwd_blocks(g)), dim3(NT), (hipStream_t)stream, p, g);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_rownorm_act_fwd(const float* y, float* out, const float* gamma, const float* beta, const float* alpha,
                                    float* mean_out, float* rstd_out, int S, int C, int T, float eps, int mode,
                                    void* stream) {
    if (S <= 0 || C <= 0 || T <= 0) return 0;
    if (mode != 0 && mode != 1) return -2;
    const long blocks = mode == 0 ? (long)S * C : (long)S * ((T + LN_TT - 1) / LN_TT);
    if (blocks >= 0x7fffffffL) return -8;
    PASE_LAUNCH(rownorm_act_fwd_kernel, dim3((unsigned)blocks), dim3(NT), (hipStream_t)stream, y, out, gamma, beta, alpha,
                mean_out, rstd_out, S, C, T, eps, mode);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_rownorm_act_bwd(const PaseActBwd* d, void* stream) {
    const PaseActBwd p = *d;
    if (!p.sums || !p.y || !p.dy || !p.mean || !p.rstd) return -2;
    if (p.has_bn != 3 && p.has_bn != 4) return -2;
    if (p.S <= 0 || p.C <= 0 || p.T <= 0) return 0;
    const long blocks = p.has_bn == 3 ? (long)p.S * p.C : (long)p.S * ((p.T + LN_TT - 1) / LN_TT);
    if (blocks >= 0x7fffffffL) return -8;
    PASE_LAUNCH(rownorm_act_bwd_kernel, dim3((unsigned)blocks), dim3(NT), (hipStream_t)stream, p);
    PASE_CHECK_LAUNCH();
    return 0;
}

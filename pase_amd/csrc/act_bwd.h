// act_bwd.h -- element functions of the BatchNorm + PReLU backward, shared by norm_act.hip's reduce / apply passes and by
// the kernels that evaluate the apply pass ON LOAD of their gradient operand (sinc_x6.hip's weight gradient: the SincNet
// layer's dy is consumed by nothing else, so it is never materialised).  Not part of the ABI.
#pragma once
#include "hip_compat.h"
#include "pase_amd.h"

// ---- gradient w.r.t. the post-activation tensor, assembled from its producers -------------------
//   * dsrc: data-gradient written by conv_gemm in *padded* coordinates (length Tp, left pad padL);
//           reflect padding folds the mirrored edges back (autograd of F.pad(mode='reflect')),
//   * dpool: gradient of the mean-pooled dense-skip branch, broadcast back over its d inputs.
// pool_magic: ceil(2^32 / pool_d) (0: divide) -- t / pool_d as one v_mul_hi_u32 instead of a ~20-instruction integer
// division per element (fewer instructions; measured no change in the passes' duration: they are memory-bound)
__device__ __forceinline__ unsigned act_pool_magic(const PaseActBwd& p) {
    if (!p.dpool || p.pool_d <= 1) return 0u;
    if ((unsigned long long)p.T * (unsigned)p.pool_d >= 0x100000000ULL) return 0u;       // exactness domain of the multiply
    return (unsigned)((0x100000000ULL + (unsigned)p.pool_d - 1u) / (unsigned)p.pool_d);
}

__device__ __forceinline__ float grad_post_act(const PaseActBwd& p, int s, int c, int t, unsigned pool_magic) {
    float v = 0.f;
    if (p.dsrc) {
        const float* row = p.dsrc + ((size_t)s * p.dsrc_ctot + p.dsrc_coff + c) * (size_t)p.Tp;
        const int i = t + p.padL;
        if (i < p.Tp) v = row[i];
        if (p.pad_mode == PASE_PAD_REFLECT) {
            const int padR = p.Tp - p.T - p.padL;
            // (only the first padL + 1 and the last padR + 1 steps receive a mirrored contribution)
            if (t <= p.padL || t >= p.T - 1 - padR) {
                if (t >= 1 && t <= p.padL) v += row[p.padL - t];
                if (t >= p.T - 1 - padR && t <= p.T - 2) v += row[p.padL + 2 * (p.T - 1) - t];
            }
        }
    }
    if (p.dpool) {
        const int f = pool_magic ? (int)(((unsigned long long)(unsigned)t * pool_magic) >> 32) : (p.pool_d > 1 ? t / p.pool_d : t);
        if (f < p.pool_F) v += p.dpool[((size_t)s * p.dpool_ctot + p.dpool_coff + c) * (size_t)p.pool_F + f] * p.pool_inv;
    }
    return v;
}

// per-(channel) constants of the apply pass and its element function:  dy = scale * (dz - mean(dz) - xhat * mean(dz * xhat))
// with batch statistics (has_bn 1), scale * dz behind frozen statistics (2), dz without a norm (0)
struct ActBwdRow { float a, b, al, mean, rstd, m1, m2; };

__device__ __forceinline__ ActBwdRow act_bwd_row(const PaseActBwd& p, int c) {
    ActBwdRow r;
    r.a = p.scale ? p.scale[c] : 1.f;
    r.b = p.shift ? p.shift[c] : 0.f;
    r.al = p.alpha ? p.alpha[c] : 1.f;
    r.mean = p.mean ? p.mean[c] : 0.f;
    r.rstd = p.rstd ? p.rstd[c] : 1.f;
    r.m1 = 0.f;
    r.m2 = 0.f;
    if (p.has_bn == 1) {
        const double n = (double)p.S * (double)p.T;
        r.m1 = (float)(p.sums[(size_t)c * 3 + 0] / n);
        r.m2 = (float)(p.sums[(size_t)c * 3 + 1] / n);
    }
    return r;
}

__device__ __forceinline__ float act_bwd_dy(const ActBwdRow& r, int has_bn, float y, float dA) {
    const float z = y * r.a + r.b;
    const float dz = z > 0.f ? dA : dA * r.al;
    if (has_bn == 1) {
        const float xhat = (y - r.mean) * r.rstd;
        return r.a * (dz - r.m1 - xhat * r.m2);
    }
    return has_bn == 2 ? r.a * dz : dz;
}

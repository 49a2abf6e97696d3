// VERDICT r04 item 8 (exploratory): can a THREE-MFMA split carry an fp32 product on gfx950?
// Two fp16 pieces per operand (11 + 11 significand bits; h = rn16(x), l = rn16(x - h)), per-matrix power-of-two pre-scaling so
// that the low piece of an average element stays a NORMAL fp16, and hh -> accH, hl + lh -> accS on v_mfma_f32_32x32x16_f16 (the
// same two-accumulator separation that made the six-MFMA bf16 form unbiased: the pipe aligns the 16 products and C to the
// largest exponent and drops what falls below).  Against an fp64 host reference, on the same operands:
//   w0  bf16 x 3 pieces, 6 MFMAs, H + S accumulators, round-to-nearest pieces   (what ships)
//   w1  fp16 x 2 pieces, 3 MFMAs: hh | hl + lh
//   w2  fp16 x 2 pieces, 4 MFMAs: hh | hl + lh + ll                              (what the dropped term costs)
//   w3  fp32 MFMA k-ordered chain                                                (the exact-fp32 pipe)
//   w4  as w1 with NO pre-scaling (operands as they are)
// Data sets: mixed-sign ~N(0,1) x 0.2 N(0,1); same-signed (bias shows as mean(err / sum|ab|) != 0); wide dynamic range
// (every element times 2^-u, u uniform in [0, 16): the pre-scaling cannot keep every low piece normal).
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/experiments/f16x3_probe tools/experiments/f16x3_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_h(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ unsigned rne_bf16(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u & 0xffff0000u;
}
__device__ __forceinline__ void split_bf(const float (&x)[8], u32x4 (&o)[3]) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = x[i];
    for (int s = 0; s < 3; ++s) {
        unsigned b[8];
        for (int i = 0; i < 8; ++i) {
            b[i] = rne_bf16(r[i]);
            r[i] -= __uint_as_float(b[i]);
        }
        for (int i = 0; i < 4; ++i) o[s][i] = (b[2 * i] >> 16) | b[2 * i + 1];
    }
}
__device__ __forceinline__ void split_h(const float (&x)[8], float scale, f16x8& h, f16x8& l) {
    for (int i = 0; i < 8; ++i) {
        const float v = x[i] * scale;          // (power of two: exact unless it overflows / underflows fp32)
        const _Float16 hi = (_Float16)v;       // round to nearest even
        h[i] = hi;
        l[i] = (_Float16)(v - (float)hi);
    }
}

// A (M x K), B (N x K) row-major, contracted along K; out[v] (M x N).  sa / sb: power-of-two pre-scales of the fp16 variants.
__global__ void probe(const float* A, const float* B, float* out, int M, int N, int K, float sa, float sb) {
    const int lane = threadIdx.x, fr = lane & 31, fk = lane >> 5;
    const int mt = blockIdx.x, nt = blockIdx.y;
    const float* ar = A + (size_t)(mt * 32 + fr) * K;
    const float* br = B + (size_t)(nt * 32 + fr) * K;
    f32x16 z;
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    f32x16 h0 = z, s0 = z, h1 = z, s1 = z, h2 = z, s2 = z, v3 = z, h4 = z, s4 = z;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float xa[8], xb[8];
        for (int e = 0; e < 8; ++e) { xa[e] = ar[k0 + fk * 8 + e]; xb[e] = br[k0 + fk * 8 + e]; }
        u32x4 ga[3], gb[3];
        split_bf(xa, ga); split_bf(xb, gb);
        s0 = mfma_bf(ga[1], gb[1], s0); s0 = mfma_bf(ga[0], gb[2], s0); s0 = mfma_bf(ga[2], gb[0], s0);
        s0 = mfma_bf(ga[0], gb[1], s0); s0 = mfma_bf(ga[1], gb[0], s0); h0 = mfma_bf(ga[0], gb[0], h0);
        f16x8 ah, al, bh, bl;
        split_h(xa, sa, ah, al); split_h(xb, sb, bh, bl);
        h1 = mfma_h(ah, bh, h1); s1 = mfma_h(ah, bl, s1); s1 = mfma_h(al, bh, s1);
        h2 = mfma_h(ah, bh, h2); s2 = mfma_h(al, bl, s2); s2 = mfma_h(ah, bl, s2); s2 = mfma_h(al, bh, s2);
        f16x8 ch, cl, dh, dl;
        split_h(xa, 1.f, ch, cl); split_h(xb, 1.f, dh, dl);
        h4 = mfma_h(ch, dh, h4); s4 = mfma_h(ch, dl, s4); s4 = mfma_h(cl, dh, s4);
        for (int s = 0; s < 8; ++s) {
            const float a1 = ar[k0 + 2 * s + fk], b1 = br[k0 + 2 * s + fk];
            v3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, v3, 0, 0, 0);
        }
    }
    const size_t MN = (size_t)M * N;
    const float un = 1.f / (sa * sb);
    for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, col = nt * 32 + fr;
        const size_t o = (size_t)row * N + col;
        out[0 * MN + o] = h0[r] + s0[r];
        out[1 * MN + o] = (h1[r] + s1[r]) * un;
        out[2 * MN + o] = (h2[r] + s2[r]) * un;
        out[3 * MN + o] = v3[r];
        out[4 * MN + o] = h4[r] + s4[r];
    }
}

static float pow2_scale(const std::vector<float>& v, int target_exp) {   // 2^s with max|v| * 2^s in [2^(target-1), 2^target)
    float m = 0.f;
    for (float x : v) m = fmaxf(m, fabsf(x));
    int e;
    frexpf(m, &e);                       // m = f * 2^e, f in [0.5, 1)
    return ldexpf(1.f, target_exp - e);
}

int main() {
    const int M = 128, N = 256;
    const int Ks[] = {256, 2816, 16384};
    const char* names[5] = {"w0 bf16x3 6 MFMA", "w1 f16x2 3 MFMA", "w2 f16x2 4 MFMA", "w3 fp32 MFMA", "w4 f16x2 unscaled"};
    const char* sets[3] = {"mixed", "pos  ", "wide "};
    for (int set = 0; set < 3; ++set)
        for (int K : Ks) {
            std::vector<float> A((size_t)M * K), B((size_t)N * K);
            srand(7);
            auto rnd = [&]() {
                float s = 0;
                for (int i = 0; i < 4; ++i) s += (float)rand() / RAND_MAX - 0.5f;
                return s * 1.7f;
            };
            auto wide = [&]() { return set == 2 ? ldexpf(1.f, -(rand() % 16)) : 1.f; };
            for (auto& v : A) { v = rnd() * wide(); if (set == 1) v = fabsf(v); }
            for (auto& v : B) { v = rnd() * 0.2f * wide(); if (set == 1) v = fabsf(v); }
            // fp16 max 65504: with K terms accumulated in fp32 inside the MFMA only the OPERANDS must fit: scale max|x| to < 2^12
            // (headroom for the 16-term block sums is not needed: products are formed in a wider format)
            const float sa = pow2_scale(A, 12), sb = pow2_scale(B, 12);
            float *dA, *dB, *dO;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, (size_t)5 * M * N * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            probe<<<dim3(M / 32, N / 32), 64>>>(dA, dB, dO, M, N, K, sa, sb);
            std::vector<float> O((size_t)5 * M * N);
            hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
            std::vector<double> ref((size_t)M * N), sab((size_t)M * N);
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < N; ++j) {
                    double s = 0, sa2 = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)A[(size_t)i * K + k] * B[(size_t)j * K + k]; s += p; sa2 += fabs(p); }
                    ref[(size_t)i * N + j] = s; sab[(size_t)i * N + j] = sa2;
                }
            for (int v = 0; v < 5; ++v) {
                double se = 0, se2 = 0, sr2 = 0, sn = 0;
                for (size_t o = 0; o < ref.size(); ++o) {
                    const double e = (double)O[v * ref.size() + o] - ref[o];
                    se += e; se2 += e * e; sr2 += ref[o] * ref[o]; sn += e / sab[o];
                }
                const double n = (double)ref.size();
                printf("%s K=%5d  %-18s relL2 %.3e  mean/rms %+.3f  mean(err / sum|ab|) %+.3e\n", sets[set], K, names[v], sqrt(se2 / sr2),
                       (se / n) / sqrt(se2 / n), sn / n);
            }
            hipFree(dA); hipFree(dB); hipFree(dO);
        }
    return 0;
}

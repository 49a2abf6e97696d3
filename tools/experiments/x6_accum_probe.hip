// Which accumulation scheme makes the split-bf16 ("x6") contraction unbiased on gfx950?
// v_mfma_f32_32x32x16_bf16 aligns its 16 products and C to the largest exponent and drops low bits toward -inf
// (tools/experiments/mfma_round_probe.hip), so small split terms (hm, mh, hl, lh, mm) added to a large accumulator
// lose bits systematically.  Variants, all on the same random operands, against an fp64 host reference:
//   v0  all six terms into one accumulator, smallest first           (the round-2 kernels)
//   v1  hh -> accH ; the other five -> accS ; result = accH + accS
//   v2  hh -> accH ; hm, mh -> accM ; hl, lh, mm -> accL ; result = accH + (accM + accL)
//   v3  fp32 MFMA (v_mfma_f32_32x32x2_f32) k-ordered chain, for scale
//   v4  as v1 but operands split with round-to-nearest pieces (v_cvt_pk_bf16_f32)
// One wave per 32x32 output tile; K is a run-time multiple of 16.
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/experiments/x6_accum_probe tools/experiments/x6_accum_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split_trunc(const float (&x)[8], u32x4 (&o)[3]) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = x[i];
    for (int s = 0; s < 3; ++s) {
        unsigned b[8];
        for (int i = 0; i < 8; ++i) {
            b[i] = __float_as_uint(r[i]) & 0xffff0000u;
            r[i] -= __uint_as_float(b[i]);
        }
        for (int i = 0; i < 4; ++i) o[s][i] = (b[2 * i] >> 16) | b[2 * i + 1];
    }
}
__device__ __forceinline__ unsigned rne_bf16(float f) {   // round-to-nearest-even to bf16, as fp32 bits
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u & 0xffff0000u;
}
__device__ __forceinline__ void split_rne(const float (&x)[8], u32x4 (&o)[3]) {
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

// A (M x K) row-major, B (N x K) row-major (both contracted along K), out[v] (M x N)
__global__ void probe(const float* A, const float* B, float* out, int M, int N, int K) {
    const int lane = threadIdx.x, fr = lane & 31, fk = lane >> 5;
    const int mt = blockIdx.x, nt = blockIdx.y;
    const float* ar = A + (size_t)(mt * 32 + fr) * K;
    const float* br = B + (size_t)(nt * 32 + fr) * K;
    f32x16 z;
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    f32x16 v0 = z, h1 = z, s1 = z, h2 = z, m2 = z, l2 = z, v3 = z, h4 = z, s4 = z;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float xa[8], xb[8];
        for (int e = 0; e < 8; ++e) { xa[e] = ar[k0 + fk * 8 + e]; xb[e] = br[k0 + fk * 8 + e]; }
        u32x4 fa[3], fb[3], ga[3], gb[3];
        split_trunc(xa, fa); split_trunc(xb, fb);
        split_rne(xa, ga); split_rne(xb, gb);
        // v0: mm, hl, lh, hm, mh, hh
        v0 = mfma_bf(fa[1], fb[1], v0); v0 = mfma_bf(fa[0], fb[2], v0); v0 = mfma_bf(fa[2], fb[0], v0);
        v0 = mfma_bf(fa[0], fb[1], v0); v0 = mfma_bf(fa[1], fb[0], v0); v0 = mfma_bf(fa[0], fb[0], v0);
        // v1
        s1 = mfma_bf(fa[1], fb[1], s1); s1 = mfma_bf(fa[0], fb[2], s1); s1 = mfma_bf(fa[2], fb[0], s1);
        s1 = mfma_bf(fa[0], fb[1], s1); s1 = mfma_bf(fa[1], fb[0], s1); h1 = mfma_bf(fa[0], fb[0], h1);
        // v2
        l2 = mfma_bf(fa[1], fb[1], l2); l2 = mfma_bf(fa[0], fb[2], l2); l2 = mfma_bf(fa[2], fb[0], l2);
        m2 = mfma_bf(fa[0], fb[1], m2); m2 = mfma_bf(fa[1], fb[0], m2); h2 = mfma_bf(fa[0], fb[0], h2);
        // v4
        s4 = mfma_bf(ga[1], gb[1], s4); s4 = mfma_bf(ga[0], gb[2], s4); s4 = mfma_bf(ga[2], gb[0], s4);
        s4 = mfma_bf(ga[0], gb[1], s4); s4 = mfma_bf(ga[1], gb[0], s4); h4 = mfma_bf(ga[0], gb[0], h4);
        // v3: fp32 MFMA, k order: lane half fk holds k = 2 s + fk
        for (int s = 0; s < 8; ++s) {
            const float a1 = ar[k0 + 2 * s + fk], b1 = br[k0 + 2 * s + fk];
            v3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, v3, 0, 0, 0);
        }
    }
    const size_t MN = (size_t)M * N;
    for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, col = nt * 32 + fr;
        const size_t o = (size_t)row * N + col;
        out[0 * MN + o] = v0[r];
        out[1 * MN + o] = h1[r] + s1[r];
        out[2 * MN + o] = h2[r] + (m2[r] + l2[r]);
        out[3 * MN + o] = v3[r];
        out[4 * MN + o] = h4[r] + s4[r];
    }
}

int main() {
    const int M = 128, N = 256;
    const int Ks[] = {64, 256, 2048, 16384};
    const char* names[5] = {"v0 one acc", "v1 H+S", "v2 H+M+L", "v3 fp32 mfma", "v4 H+S rne-split"};
    for (int positive = 1; positive >= 0; --positive)
        for (int K : Ks) {
            std::vector<float> A((size_t)M * K), B((size_t)N * K);
            srand(7);
            auto rnd = [&]() {   // roughly normal
                float s = 0;
                for (int i = 0; i < 4; ++i) s += (float)rand() / RAND_MAX - 0.5f;
                return s * 1.7f;
            };
            for (auto& v : A) { v = rnd(); if (positive) v = fabsf(v); }
            for (auto& v : B) { v = rnd() * 0.2f; if (positive) v = fabsf(v); }
            float *dA, *dB, *dO;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, (size_t)5 * M * N * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            probe<<<dim3(M / 32, N / 32), 64>>>(dA, dB, dO, M, N, K);
            std::vector<float> O((size_t)5 * M * N);
            hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
            std::vector<double> ref((size_t)M * N), sab((size_t)M * N);
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < N; ++j) {
                    double s = 0, sa = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)A[(size_t)i * K + k] * B[(size_t)j * K + k]; s += p; sa += fabs(p); }
                    ref[(size_t)i * N + j] = s; sab[(size_t)i * N + j] = sa;
                }
            for (int v = 0; v < 5; ++v) {
                double se = 0, se2 = 0, sr2 = 0, sn = 0;
                for (size_t o = 0; o < ref.size(); ++o) {
                    const double e = (double)O[v * ref.size() + o] - ref[o];
                    se += e; se2 += e * e; sr2 += ref[o] * ref[o]; sn += e / sab[o];
                }
                const double n = (double)ref.size();
                printf("%s K=%5d  %-18s relL2 %.3e  mean/rms %+.3f  mean(err / sum|ab|) %+.3e\n", positive ? "pos  " : "mixed", K,
                       names[v], sqrt(se2 / sr2), (se / n) / sqrt(se2 / n), sn / n);
            }
            hipFree(dA); hipFree(dB); hipFree(dO);
        }
    return 0;
}

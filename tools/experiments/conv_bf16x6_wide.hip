// EXPERIMENT — not part of libpase_hip.so, not built by pase_amd.build.
//
// (wide variant of conv_bf16x6_hybrid.hip: 128 x 256 workgroup tile, each wave 64 x 128 = 2 x 4 MFMA tiles, ONE
// workgroup per CU -- half the fragment reads and B splits per MFMA.)
// Follow-up of gemm_bf16_split.hip: the "hybrid" form a bf16x6 variant of pase_conv_gemm would take, on the shape
// of encoder block 5 (Cin 256 -> Cout 256, 11 taps, stride 1, 96 sequences):
//   * the weights are split ONCE (the per-step pack kernel) into three bf16 planes stored in fragment order,
//     so the weight slab goes HBM -> registers -> LDS as plain 16-byte chunks (no VALU);
//   * the activations stay what pase_conv_gemm stages today: fp32 sliding-window spans [channel row][time] in LDS
//     (so the on-load BatchNorm / PReLU, reflect padding and segment logic of the loader carry over unchanged);
//     each wave splits its B fragment into hi / mid / lo when it reads it (8 ds_read_b32 + ~44 VALU per fragment);
//   * a 16-deep MFMA step covers 4 channel rows x 4 taps (11 taps padded to 12 with zero weights), the lane halves
//     (k-groups) taking the even / odd rows, so every LDS offset is [per-lane constant] + immediate;
//   * accumulators / epilogue are those of the fp32 kernel (the C layout of 32x32x16_bf16 equals 32x32x2_f32).
//
//   y[s][co][t] = sum_{ci,tap} W[co][ci][tap] x[s][ci][t + tap]     (valid convolution, Tin = Tout + taps - 1)
//
//   hipcc --offload-arch=gfx950 -O3 -o conv_bf16x6_hybrid conv_bf16x6_hybrid.hip && ./conv_bf16x6_hybrid
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

__host__ __device__ inline float synth(unsigned long long i, unsigned seed) {
    unsigned long long z = i * 0x9E3779B97F4A7C15ULL + seed * 0xD1B54A32D192ED03ULL + 0x632BE59BD9B4E019ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (float)((double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0);
}

constexpr int TAPS = 11, TAPS_P = 12, NB = TAPS_P / 4, CB = 4;
constexpr int BM = 128, BN = 256, NJ = 4, SPANP = 272;          // span = 255 + 12 = 267 -> 272
constexpr int CHUNKS = NB * 2 * BM;            // 16-byte chunks of one split plane of one stage (768)
constexpr int XSLOTS = (CB * SPANP + 255) / 256;                 // 5

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = synth(i, seed);
}

__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&out)[3]) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = x[i];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        unsigned b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            b[i] = __float_as_uint(r[i]) & 0xffff0000u;
            r[i] -= __uint_as_float(b[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[s][i] = __builtin_amdgcn_perm(b[2 * i + 1], b[2 * i], 0x07060302u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[2][i] = __builtin_amdgcn_perm(__float_as_uint(r[2 * i + 1]), __float_as_uint(r[2 * i]), 0x07060302u);
}

__global__ void pack_kernel(u32x4* Ap, int M, int Cin) {
    const int ncg = Cin / CB;
    const size_t n = (size_t)ncg * CHUNKS * (M / BM);
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
        const int ml = (int)(c % BM);
        const int fk = (int)((c / BM) % 2);
        const int b = (int)((c / (2 * BM)) % NB);
        const int cg = (int)((c / CHUNKS) % ncg);
        const int rt = (int)(c / ((size_t)CHUNKS * ncg));
        const int m = rt * BM + ml;
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ci = CB * cg + fk + 2 * (i >> 2), tap = 4 * b + (i & 3);
            x[i] = tap < TAPS ? synth(((unsigned long long)m * Cin + ci) * TAPS + tap, 1u) : 0.f;
        }
        u32x4 o[3];
        split8(x, o);
#pragma unroll
        for (int s = 0; s < 3; ++s) Ap[(size_t)s * n + c] = o[s];
    }
}

__global__ __launch_bounds__(256, 1) void conv_x6_kernel(const u32x4* __restrict__ Ap, const float* __restrict__ X,
                                                         float* __restrict__ Y, int M, int Cin, int S, int Tout) {
    __shared__ __attribute__((aligned(16))) u32x4 As[2][3][CHUNKS];
    __shared__ __attribute__((aligned(16))) float Xs[2][XSLOTS * 256];
    const int Tin = Tout + TAPS - 1;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1, fr = lane & 31, fk = lane >> 5;
    const int n_rt = M / BM, tps = Tout / BN;
    const int rt = blockIdx.x % n_rt, ct = blockIdx.x / n_rt;
    const int s = ct / tps, t0 = (ct % tps) * BN;
    const int ncg = Cin / CB;
    const size_t plane = (size_t)ncg * CHUNKS * n_rt;
    const u32x4* ap = Ap + (size_t)rt * ncg * CHUNKS;
    const float* xb = X + (size_t)s * Cin * Tin + t0;

    u32x4 areg[9];
    float xreg[XSLOTS];
    auto gload = [&](int cg) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int c = t + 256 * j;
            areg[j] = ap[(size_t)(c / CHUNKS) * plane + (size_t)cg * CHUNKS + (c % CHUNKS)];
        }
#pragma unroll
        for (int j = 0; j < XSLOTS; ++j) {
            const int e = min(t + 256 * j, CB * SPANP - 1);
            const int row = e / SPANP, pos = e % SPANP;
            const int tt = min(t0 + pos, Tin - 1) - t0;
            xreg[j] = xb[(size_t)(cg * CB + row) * Tin + tt];
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int c = t + 256 * j;
            As[buf][c / CHUNKS][c % CHUNKS] = areg[j];
        }
#pragma unroll
        for (int j = 0; j < XSLOTS; ++j) Xs[buf][t + 256 * j] = xreg[j];
    };

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int cur) {
        const float* x0 = &Xs[cur][fk * SPANP + wn * 128 + fr];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            u32x4 a[3][2], bb[NJ][3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                for (int i = 0; i < 2; ++i) a[sp][i] = As[cur][sp][(b * 2 + fk) * BM + wm * 64 + i * 32 + fr];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float xv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[i] = x0[(2 * (i >> 2)) * SPANP + j * 32 + 4 * b + (i & 3)];
                split8(xv, bb[j]);
            }
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int pi = 0; pi < 6; ++pi)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, a[PA[pi]][i]), __builtin_bit_cast(bf16x8, bb[j][PB[pi]]),
                            acc[i][j], 0, 0, 0);
        }
    };

    gload(0);
    stash(0);
    __syncthreads();
    for (int cg = 0; cg + 1 < ncg; ++cg) {
        gload(cg + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(cg & 1);
        stash((cg & 1) ^ 1);
        __syncthreads();
    }
    compute((ncg - 1) & 1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = t0 + wn * 128 + j * 32 + (lane & 31);
                Y[((size_t)s * M + row) * Tout + col] = acc[i][j][r];
            }
}

int main() {
    const int S = 96, Cin = 256, M = 256, Tout = 768, Tin = Tout + TAPS - 1;
    float *X, *Y;
    u32x4* Ap;
    const size_t n_chunks = (size_t)(Cin / CB) * CHUNKS * (M / BM);
    CHECK(hipMalloc(&X, (size_t)S * Cin * Tin * 4));
    CHECK(hipMalloc(&Y, (size_t)S * M * Tout * 4));
    CHECK(hipMalloc(&Ap, 3 * n_chunks * 16));
    fill_kernel<<<2048, 256>>>(X, (size_t)S * Cin * Tin, 2u);
    pack_kernel<<<1024, 256>>>(Ap, M, Cin);
    CHECK(hipDeviceSynchronize());
    const dim3 grid((unsigned)((M / BM) * S * (Tout / BN))), block(256);
    for (int i = 0; i < 3; ++i) conv_x6_kernel<<<grid, block>>>(Ap, X, Y, M, Cin, S, Tout);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 10;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) conv_x6_kernel<<<grid, block>>>(Ap, X, Y, M, Cin, S, Tout);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0, ms_pack = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) pack_kernel<<<1024, 256>>>(Ap, M, Cin);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms_pack, e0, e1));
    ms_pack /= reps;
    double num = 0, den = 0, num32 = 0;
    for (int i = 0; i < 256; ++i) {
        const int s = i % S, m = (int)(((unsigned)i * 2654435761u) % (unsigned)M), tt = (int)(((unsigned)i * 40503u + 17u) % (unsigned)Tout);
        double ref = 0;
        float ref32 = 0;
        for (int ci = 0; ci < Cin; ++ci)
            for (int tap = 0; tap < TAPS; ++tap) {
                const float w = synth(((unsigned long long)m * Cin + ci) * TAPS + tap, 1u);
                const float x = synth(((unsigned long long)s * Cin + ci) * Tin + tt + tap, 2u);
                ref += (double)w * (double)x;
                ref32 = fmaf(w, x, ref32);
            }
        float y;
        CHECK(hipMemcpy(&y, Y + ((size_t)s * M + m) * Tout + tt, 4, hipMemcpyDeviceToHost));
        num += ((double)y - ref) * ((double)y - ref);
        num32 += ((double)ref32 - ref) * ((double)ref32 - ref);
        den += ref * ref;
    }
    const double gflop = 2.0 * S * Tout * (double)M * Cin * TAPS / 1e9;
    printf("{\"kernel\": \"conv_bf16x6_hybrid\", \"shape\": \"S96 Cin256 Cout256 taps11 Tout768\", \"ms\": %.4f, "
           "\"fp32_equiv_tflops\": %.1f, \"mfma_rate_tflops_incl_tap_padding\": %.1f, \"pack_ms\": %.4f, "
           "\"rel_l2_vs_f64\": %.3e, \"fp32_fma_chain_rel_l2_vs_f64\": %.3e}\n",
           ms, gflop / ms, gflop / ms * TAPS_P / TAPS, ms_pack, sqrt(num / den), sqrt(num32 / den));
    return 0;
}

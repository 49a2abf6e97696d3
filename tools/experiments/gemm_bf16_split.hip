// EXPERIMENT — not part of libpase_hip.so, not built by pase_amd.build.
//
// Question (round-1 review, item 10): can the fp32 contractions of the training step run on the bf16 matrix
// pipe by splitting each fp32 operand into bf16 pieces on load, and what does it cost in accuracy?
//
//   x = hi + mid + lo   (three truncated bf16 pieces carry all 24 mantissa bits exactly)
//   NS = 3: a*b ~= hh + hm + mh + hl + lh + mm   (6 bf16 MFMAs per k-step; dropped terms <= 3 * 2^-24 |ab|)
//   NS = 2: a*b ~= hh + hm + mh                  (3 MFMAs; 16-bit mantissa, error ~2^-16 |ab|)
//   NS = 1: plain bf16                           (the ceiling of this kernel structure)
//
// One plain GEMM  C[M][N] = At[K][M]^T * X[K][N]  (both operands K-major fp32 in HBM, like pase_conv_gemm's
// flat path), 128 x 128 workgroup tile, 4 waves x (2 x 2) v_mfma_f32_32x32x16_bf16, BK = 16, split in the
// loader, double-buffered LDS image [split][k/8][row][8 bf16] (one ds_read_b128 per fragment).
//
//   hipcc --offload-arch=gfx950 -O3 -o gemm_bf16_split gemm_bf16_split.hip && ./gemm_bf16_split
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
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

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = synth(i, seed);
}

__device__ __forceinline__ int swz(int r) { return r ^ ((r >> 3) & 3); }

template <int NS>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const float* __restrict__ At,
                                                            const float* __restrict__ X, float* __restrict__ C,
                                                            int M, int N, int K) {
    constexpr int BM = 128, BN = 128, BK = 16;
    __shared__ __attribute__((aligned(16))) unsigned short sm[2][2][NS][2][128][8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int tiles_m = M / BM;
    const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;

    // loader role: operand, k-group of 8, half of it (4 rows of k), a quad of 4 tile columns
    const int q = t & 31, h = (t >> 5) & 1, g = (t >> 6) & 1, op = t >> 7;
    const int ld = op ? N : M;
    const float* src = (op ? X + (size_t)bn * BN : At + (size_t)bm * BM) + (size_t)(g * 8 + h * 4) * ld + q * 4;
    float4 v[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(src + (size_t)(kt * BK + i) * ld);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = reinterpret_cast<const float*>(&v[i])[j];
            const int row = swz(q * 4 + j);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                unsigned b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    b[i] = __float_as_uint(x[i]) & 0xffff0000u;
                    x[i] -= __uint_as_float(b[i]);          // exact
                }
                uint2 w;
                w.x = __builtin_amdgcn_perm(b[1], b[0], 0x07060302u);
                w.y = __builtin_amdgcn_perm(b[3], b[2], 0x07060302u);
                *reinterpret_cast<uint2*>(&sm[buf][op][s][g][row][h * 4]) = w;
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    gload(0);
    stash(0);
    __syncthreads();
    const int kg = lane >> 5, lr = lane & 31;
    auto compute = [&](int cur) {
        bf16x8 a[NS][2], b[NS][2];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[s][i] = *reinterpret_cast<const bf16x8*>(&sm[cur][0][s][kg][swz(wm * 64 + i * 32 + lr)][0]);
                b[s][i] = *reinterpret_cast<const bf16x8*>(&sm[cur][1][s][kg][swz(wn * 64 + i * 32 + lr)][0]);
            }
        // smallest terms first
        constexpr int NP = NS == 3 ? 6 : (NS == 2 ? 3 : 1);
        constexpr int PA[6] = {NS == 3 ? 1 : 0, NS == 3 ? 0 : 1, NS == 3 ? 2 : 0, 0, 1, 0};
        constexpr int PB[6] = {NS == 3 ? 1 : (NS == 2 ? 1 : 0), NS == 3 ? 2 : 0, 0, 1, 0, 0};
#pragma unroll
        for (int pi = 0; pi < NP; ++pi)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[pi]][i], b[PB[pi]][j], acc[i][j], 0, 0, 0);
    };
    // one basic block per stage: next stage's loads, this stage's MFMAs, next stage's split + LDS writes
    for (int kt = 0; kt + 1 < nk; ++kt) {
        gload(kt + 1);
        compute(kt & 1);
        stash((kt & 1) ^ 1);
        __syncthreads();
    }
    compute((nk - 1) & 1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = bn * BN + wn * 64 + j * 32 + (lane & 31);
                C[(size_t)row * N + col] = acc[i][j][r];
            }
}

template <int NS>
static void run(const char* name, const float* At, const float* X, float* C, int M, int N, int K,
                const std::vector<int>& sm, const std::vector<int>& sn, const std::vector<double>& ref,
                const std::vector<float>& ref32) {
    const dim3 grid((unsigned)((M / 128) * (N / 128))), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) gemm_split_kernel<NS><<<grid, block>>>(At, X, C, M, N, K);
    CHECK(hipDeviceSynchronize());
    const int reps = 10;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) gemm_split_kernel<NS><<<grid, block>>>(At, X, C, M, N, K);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double num = 0, den = 0, worst = 0, num32 = 0;
    for (size_t i = 0; i < sm.size(); ++i) {
        float c;
        CHECK(hipMemcpy(&c, C + (size_t)sm[i] * N + sn[i], 4, hipMemcpyDeviceToHost));
        const double d = (double)c - ref[i];
        num += d * d;
        den += ref[i] * ref[i];
        const double d32 = (double)ref32[i] - ref[i];
        num32 += d32 * d32;
        if (fabs(d) > worst) worst = fabs(d);
    }
    const double tf = 2.0 * M * N * (double)K / (ms * 1e-3) / 1e12;
    printf("{\"kernel\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"ms\": %.4f, \"fp32_equiv_tflops\": %.1f, "
           "\"rel_l2_vs_f64\": %.3e, \"max_abs_err\": %.3e, \"fp32_fma_chain_rel_l2_vs_f64\": %.3e}\n",
           name, M, N, K, ms, tf, sqrt(num / den), worst, sqrt(num32 / den));
    fflush(stdout);
}

int main() {
    const int shapes[3][3] = {{512, 19200, 5632}, {4096, 4096, 4096}, {256, 6400, 21504}};
    for (int si = 0; si < 3; ++si) {
        const int M = shapes[si][0], N = shapes[si][1], K = shapes[si][2];
        float *At, *X, *C;
        CHECK(hipMalloc(&At, (size_t)K * M * 4));
        CHECK(hipMalloc(&X, (size_t)K * N * 4));
        CHECK(hipMalloc(&C, (size_t)M * N * 4));
        fill_kernel<<<2048, 256>>>(At, (size_t)K * M, 1u);
        fill_kernel<<<2048, 256>>>(X, (size_t)K * N, 2u);
        CHECK(hipDeviceSynchronize());
        std::vector<int> sm, sn;
        std::vector<double> ref;
        std::vector<float> ref32;
        for (int i = 0; i < 192; ++i) {
            const int m = (int)(((unsigned)i * 2654435761u) % (unsigned)M), n = (int)(((unsigned)i * 40503u + 17u) % (unsigned)N);
            double s = 0;
            float s32 = 0;
            for (int k = 0; k < K; ++k) {
                const float a = synth((unsigned long long)k * M + m, 1u), x = synth((unsigned long long)k * N + n, 2u);
                s += (double)a * (double)x;
                s32 = fmaf(a, x, s32);
            }
            sm.push_back(m);
            sn.push_back(n);
            ref.push_back(s);
            ref32.push_back(s32);
        }
        run<1>("bf16x1", At, X, C, M, N, K, sm, sn, ref, ref32);
        run<2>("bf16x3", At, X, C, M, N, K, sm, sn, ref, ref32);
        run<3>("bf16x6", At, X, C, M, N, K, sm, sn, ref, ref32);
        CHECK(hipFree(At));
        CHECK(hipFree(X));
        CHECK(hipFree(C));
    }
    return 0;
}

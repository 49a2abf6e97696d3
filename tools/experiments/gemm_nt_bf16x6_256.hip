// EXPERIMENT — not part of libpase_hip.so, not built by pase_amd.build.
//
// The "next rung" for the split-bf16 contraction: 256 x 256 workgroup tile, 4 waves x (4 x 4)
// v_mfma_f32_32x32x16_bf16 tiles (256 accumulator registers per lane -> AGPRs, ONE wave per SIMD), so a 16-deep step
// is 24 fragment reads per 96 MFMAs (the shipped kernels: 12-22 reads + up to 88 VALU per 24 MFMAs).
// Problem = the 1x1 weight-gradient shape: C[M][N] = sum_k G[m][k] * Z[n][k], both operands fp32 and contiguous along
// the reduction; both are split into three bf16 pieces by the staging threads (float4 = half a fragment).
//
//   hipcc --offload-arch=gfx950 -O3 -o gemm_nt_bf16x6_256 gemm_nt_bf16x6_256.hip && ./gemm_nt_bf16x6_256
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

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = synth(i, seed);
}

// four values -> three pieces x two dwords (half a fragment)
__device__ __forceinline__ void split4(const float4 v, unsigned (&o)[3][2]) {
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        unsigned b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b[i] = __float_as_uint(r[i]) & 0xffff0000u;
            r[i] -= __uint_as_float(b[i]);
        }
        o[s][0] = __builtin_amdgcn_perm(b[1], b[0], 0x07060302u);
        o[s][1] = __builtin_amdgcn_perm(b[3], b[2], 0x07060302u);
    }
}

constexpr int BT = 256;          // tile rows = tile columns
constexpr int BK = 16;           // reduction positions per stage
constexpr int OPB = 2 * 3 * BT;  // 16-byte chunks of one operand in one buffer: [k-group][plane][row]

__global__ __launch_bounds__(256, 1) void gemm_nt_kernel(const float* __restrict__ G, const float* __restrict__ Z,
                                                         float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) u32x4 Ls[2][2][OPB];     // [buffer][operand][chunk]: 96 KB
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fk = lane >> 5;
    const int tiles_m = M / BT;
    const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;

    // staging: thread -> rows r0 + 64 i (i < 4) of both operands, time steps k4 .. k4+3 = half (t & 1) of k-group (t & 3) >> 1
    const int k4 = (t & 3) * 4, r0 = t >> 2;
    const float* gsrc = G + (size_t)(bm * BT + r0) * K + k4;
    const float* zsrc = Z + (size_t)(bn * BT + r0) * K + k4;
    float4 greg[4], zreg[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            greg[i] = *reinterpret_cast<const float4*>(gsrc + (size_t)(64 * i) * K + kt * BK);
            zreg[i] = *reinterpret_cast<const float4*>(zsrc + (size_t)(64 * i) * K + kt * BK);
        }
    };
    const int sub = ((t & 3) >> 1) * 3 * BT, halfb = (t & 1) * 8;
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned o[3][2];
            split4(greg[i], o);
            unsigned char* d = reinterpret_cast<unsigned char*>(&Ls[buf][0][sub + r0 + 64 * i]) + halfb;
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) *reinterpret_cast<uint2*>(d + pz * BT * 16) = make_uint2(o[pz][0], o[pz][1]);
            split4(zreg[i], o);
            d = reinterpret_cast<unsigned char*>(&Ls[buf][1][sub + r0 + 64 * i]) + halfb;
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) *reinterpret_cast<uint2*>(d + pz * BT * 16) = make_uint2(o[pz][0], o[pz][1]);
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    gload(0);
    stash(0);
    __syncthreads();
    auto compute = [&](int cur) __attribute__((always_inline)) {
        const u32x4* aL = &Ls[cur][0][fk * 3 * BT + wm * 128 + fr];
        const u32x4* bL = &Ls[cur][1][fk * 3 * BT + wn * 128 + fr];
        // products ordered by A plane so one plane of A fragments is live at a time: (2,0) (1,1) (1,0) (0,2) (0,1) (0,0)
        u32x4 fb[3][4];
#pragma unroll
        for (int pz = 0; pz < 3; ++pz)
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[pz][j] = bL[pz * BT + 32 * j];
        constexpr int PA[6] = {2, 1, 1, 0, 0, 0}, PB[6] = {0, 1, 0, 2, 1, 0};
        u32x4 fa[4];
#pragma unroll
        for (int pi = 0; pi < 6; ++pi) {
            if (pi == 0 || PA[pi] != PA[pi - 1]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = aL[PA[pi] * BT + 32 * i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                        __builtin_bit_cast(bf16x8, fb[PB[pi]][j]),
                                                                        acc[i][j], 0, 0, 0);
        }
    };
    // one basic block per stage: next stage's loads, this stage's 96 MFMAs, next stage's split + LDS writes
    for (int kt = 0; kt + 1 < nk; ++kt) {
        gload(kt + 1);
        __builtin_amdgcn_sched_barrier(0);      // the loads stay in front of the stage's MFMAs
        compute(kt & 1);
        stash((kt & 1) ^ 1);
#ifdef EXP_SGB
#pragma unroll
        for (int q = 0; q < 96; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);
        }
#endif
        __syncthreads();
    }
    compute((nk - 1) & 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bm * BT + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = bn * BT + wn * 128 + j * 32 + (lane & 31);
                C[(size_t)row * N + col] = acc[i][j][r];
            }
}

int main() {
    const int shapes[3][3] = {{4096, 4096, 4096}, {21504, 256, 6400}, {8192, 8192, 2048}};
    for (int si = 0; si < 3; ++si) {
        const int M = shapes[si][0], N = shapes[si][1], K = shapes[si][2];
        float *G, *Z, *C;
        CHECK(hipMalloc(&G, (size_t)M * K * 4));
        CHECK(hipMalloc(&Z, (size_t)N * K * 4));
        CHECK(hipMalloc(&C, (size_t)M * N * 4));
        fill_kernel<<<2048, 256>>>(G, (size_t)M * K, 1u);
        fill_kernel<<<2048, 256>>>(Z, (size_t)N * K, 2u);
        CHECK(hipDeviceSynchronize());
        const dim3 grid((unsigned)((M / BT) * (N / BT))), block(256);
        for (int i = 0; i < 3; ++i) gemm_nt_kernel<<<grid, block>>>(G, Z, C, M, N, K);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        const int reps = 10;
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) gemm_nt_kernel<<<grid, block>>>(G, Z, C, M, N, K);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        double num = 0, den = 0, num32 = 0;
        for (int i = 0; i < 128; ++i) {
            const int m = (int)(((unsigned)i * 2654435761u) % (unsigned)M), n = (int)(((unsigned)i * 40503u + 17u) % (unsigned)N);
            double ref = 0;
            float ref32 = 0;
            for (int k = 0; k < K; ++k) {
                const float a = synth((unsigned long long)m * K + k, 1u), b = synth((unsigned long long)n * K + k, 2u);
                ref += (double)a * (double)b;
                ref32 = fmaf(a, b, ref32);
            }
            float c;
            CHECK(hipMemcpy(&c, C + (size_t)m * N + n, 4, hipMemcpyDeviceToHost));
            num += ((double)c - ref) * ((double)c - ref);
            num32 += ((double)ref32 - ref) * ((double)ref32 - ref);
            den += ref * ref;
        }
        printf("{\"kernel\": \"gemm_nt_bf16x6_256\", \"M\": %d, \"N\": %d, \"K\": %d, \"workgroups\": %d, \"ms\": %.4f, "
               "\"fp32_equiv_tflops\": %.1f, \"rel_l2_vs_f64\": %.3e, \"fp32_fma_chain_rel_l2_vs_f64\": %.3e}\n",
               M, N, K, (M / BT) * (N / BT), ms, 2.0 * M * N * (double)K / ms / 1e9, sqrt(num / den), sqrt(num32 / den));
        fflush(stdout);
        CHECK(hipFree(G));
        CHECK(hipFree(Z));
        CHECK(hipFree(C));
    }
    return 0;
}

// mfma_dep_probe.hip -- how much does v_mfma_f32_32x32x16_bf16 lose when consecutive instructions share an accumulator,
// and what do ds_read_b128 / global_load_dwordx4 waits inside the stream cost?  One wave per SIMD (4 waves per CU, one
// workgroup per CU), s_memtime around a loop of 24-MFMA "steps" in several orders:
//   0  all 24 on ONE accumulator                      1  pairs: (S0 S0)(S1 S1) ...          2  alternate two accumulators
//   3  round robin over 8 accumulators                4  the order hipcc emitted for conv_x6c's step (see below)
//   5  as 3, plus 12 ds_read_b128 per step issued ONE STEP ahead (fragments double-buffered)
//   6  as 3, plus 12 ds_read_b128 per step issued right before their first use (lgkmcnt waits inside the step)
//   7  two halves software-pipelined: the six reads of the NEXT half are issued in front of the 12 MFMAs of this one
//   8  as 7 with the six reads spread between the MFMAs (one read per two MFMAs)
//   9  as 8, plus THREE global_load_dwordx4 per step (the A fragments of conv_x6c's compute waves: 1 KB per instruction,
//      L2-resident, prefetched two steps ahead)         10  as 8, plus three more ds_read_b128 per step instead (A through LDS)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_dep_probe mfma_dep_probe.hip ; run: ./mfma_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ void __launch_bounds__(256, 1) probe(unsigned long long* out, float* sink, int steps, const u32x4* gsrc = nullptr) {
    __shared__ u32x4 lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    u32x4 a[3], b[12], bn[12];
    for (int i = 0; i < 3; ++i) a[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    for (int i = 0; i < 12; ++i) b[i] = lds[lane + 64 * i];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 24; ++i) { acc[0] = mf(a[i % 3], b[i % 12], acc[0]); SB(); }
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 24; ++i) { acc[(i / 2) % 8] = mf(a[i % 3], b[i % 12], acc[(i / 2) % 8]); SB(); }
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 24; ++i) { acc[(i % 2) + 2 * ((i / 6) % 4)] = mf(a[i % 3], b[i % 12], acc[(i % 2) + 2 * ((i / 6) % 4)]); SB(); }
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 24; ++i) { acc[i % 8] = mf(a[i % 3], b[i % 12], acc[i % 8]); SB(); }
        } else if (MODE == 4) {
            // accumulator sequence of the compiled conv_x6c step: 0 0 1 1 1 0 1 0 1 4 0 4(...) -- approximated: 0 0 1 1 1 0 1 0 1 5 0 4 | 2 2 3 3 2 3 2 3 2 3 6 7
            constexpr int SEQ[24] = {0, 0, 1, 1, 1, 0, 1, 0, 1, 5, 0, 4, 2, 2, 3, 3, 2, 3, 2, 3, 2, 3, 6, 7};
#pragma unroll
            for (int i = 0; i < 24; ++i) { acc[SEQ[i]] = mf(a[i % 3], b[i % 12], acc[SEQ[i]]); SB(); }
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 12; ++i) bn[i] = lds[((lane + 64 * i + 7 * s) & 4095)];
#pragma unroll
            for (int i = 0; i < 24; ++i) acc[i % 8] = mf(a[i % 3], b[i % 12], acc[i % 8]);
#pragma unroll
            for (int i = 0; i < 12; ++i) b[i] = bn[i];
        } else if (MODE == 7 || MODE == 8) {
            // cA holds the fragments of this step's first half (loaded during the previous step's second half)
            u32x4* cA = b;          // b[0..5]
            u32x4* cB = b + 6;      // b[6..11]
            if (s == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) cA[i] = lds[(lane + 64 * i) & 4095];
            }
            if (MODE == 7) {
#pragma unroll
                for (int i = 0; i < 6; ++i) cB[i] = lds[((lane + 64 * (i + 6) + 7 * s) & 4095)];
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i % 4] = mf(a[i % 3], cA[i % 6], acc[i % 4]);
                SB();
#pragma unroll
                for (int i = 0; i < 6; ++i) cA[i] = lds[((lane + 64 * i + 7 * s + 7) & 4095)];
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[4 + i % 4] = mf(a[i % 3], cB[i % 6], acc[4 + i % 4]);
                SB();
            } else {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    if ((i & 1) == 0) cB[i / 2] = lds[((lane + 64 * (i / 2 + 6) + 7 * s) & 4095)];
                    acc[i % 4] = mf(a[i % 3], cA[i % 6], acc[i % 4]);
                    SB();
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    if ((i & 1) == 0) cA[i / 2] = lds[((lane + 64 * (i / 2) + 7 * s + 7) & 4095)];
                    acc[4 + i % 4] = mf(a[i % 3], cB[i % 6], acc[4 + i % 4]);
                    SB();
                }
            }
        } else if (MODE == 6) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 c[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) c[i] = lds[((lane + 64 * (i + 6 * h) + 7 * s) & 4095)];
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[(i % 4) + 4 * h] = mf(a[i % 3], c[i % 6], acc[(i % 4) + 4 * h]);
                SB();
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int j = 0; j < 8; ++j) sum += acc[j][0] + acc[j][7];
    if (sum == 123.456f) sink[0] = sum;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
__global__ void __launch_bounds__(256, 1) probe_a(unsigned long long* out, float* sink, int steps, const u32x4* gsrc) {
    __shared__ u32x4 lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    u32x4 cA[6], cB[6], a0[3], a1[3], a2[3];
    const u32x4* gp = gsrc + ((size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 64) * 192 + lane;
    for (int i = 0; i < 3; ++i) { a0[i] = gp[64 * i]; a1[i] = gp[192 + 64 * i]; }
    for (int i = 0; i < 6; ++i) cA[i] = lds[(lane + 64 * i) & 4095];
    int s = 0;
    auto step = [&](const u32x4 (&cur)[3], u32x4 (&nxt)[3]) __attribute__((always_inline)) {
        if (MODE == 9) {
#pragma unroll
            for (int i = 0; i < 3; ++i) nxt[i] = gp[(size_t)((s + 2) & 63) * 192 + 64 * i];
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) nxt[i] = lds[((lane + 64 * (i + 13) + 5 * s) & 4095)];
        }
        __builtin_amdgcn_sched_group_barrier(MODE == 9 ? 0x020 : 0x100, 3, 0);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if ((i & 1) == 0) cB[i / 2] = lds[((lane + 64 * (i / 2 + 6) + 7 * s) & 4095)];
            acc[i % 4] = mf(cur[i % 3], cA[i % 6], acc[i % 4]);
            SB();
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if ((i & 1) == 0) cA[i / 2] = lds[((lane + 64 * (i / 2) + 7 * s + 7) & 4095)];
            acc[4 + i % 4] = mf(cur[i % 3], cB[i % 6], acc[4 + i % 4]);
            SB();
        }
        ++s;
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < steps / 3; ++it) {
        step(a0, a2);
        step(a1, a0);
        step(a2, a1);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int j = 0; j < 8; ++j) sum += acc[j][0] + acc[j][7];
    if (sum == 123.456f) sink[0] = sum;
    if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) * steps / (3 * (steps / 3));
}

int main() {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * 8);
    hipMalloc(&sink, 4);
    const int steps = 2000;
    std::vector<unsigned long long> h(256);
    u32x4* gsrc;
    hipMalloc(&gsrc, (size_t)256 * 4 * 64 * 192 * 16);
    hipMemset(gsrc, 0x3f, (size_t)256 * 4 * 64 * 192 * 16);
    auto run = [&](auto kern, const char* name) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, sink, steps, gsrc);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, sink, steps, gsrc);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        printf("%-70s %8.1f clock ticks per 24-MFMA step (768 = pipe-bound at 32 cycles per MFMA if the counter runs at the shader clock)\n",
               name, s / 256 / steps);
    };
    run(probe<0>, "0 one accumulator (every MFMA depends on the previous one)");
    run(probe<1>, "1 pairs on the same accumulator");
    run(probe<2>, "2 two accumulators alternating (distance 2)");
    run(probe<3>, "3 round robin over 8 accumulators");
    run(probe<4>, "4 conv_x6c's compiled order");
    run(probe<5>, "5 round robin + 12 ds_read_b128 per step, one step ahead");
    run(probe<6>, "6 two halves, each: 6 ds_read_b128 then 12 MFMAs on 4 accumulators");
    run(probe<7>, "7 two halves, the next half's 6 reads issued in front of this half's MFMAs");
    run(probe<8>, "8 as 7, one read per two MFMAs");
    run(probe_a<9>, "9 as 8 + three global_load_dwordx4 per step (A fragments from L2, two steps ahead)");
    run(probe_a<10>, "10 as 8 + three more ds_read_b128 per step (A fragments through LDS)");
    return 0;
}

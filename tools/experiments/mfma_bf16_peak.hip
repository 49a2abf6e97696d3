// EXPERIMENT: sustained issue rate of v_mfma_f32_32x32x16_bf16 on this box (what "100 %" means for the split-bf16 kernels).
//   NACC independent accumulators per wave, WPS waves per SIMD, ITER x NACC MFMAs per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.5f); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs, int iters) {
    float* out;
    hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<wgs, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<wgs, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 4 * iters * NACC * 2.0 * 32 * 32 * 16;
    printf("{\"nacc\": %d, \"workgroups\": %d, \"iters\": %d, \"ms\": %.3f, \"bf16_tflops\": %.0f, \"fp32_equiv_x6_tflops\": %.1f}\n", NACC, wgs, iters, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 6);
    hipFree(out);
}
int main() {
    run<4>(256, 20000); run<4>(512, 20000); run<8>(256, 10000); run<16>(256, 5000); run<4>(256, 200000); run<4>(1024, 20000);
    return 0;
}

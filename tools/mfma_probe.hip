// mfma_probe.hip -- ceilings for the conv_gemm wave structure on gfx950 (diagnostic tool, GPU box only)
//   P0: 4 independent 32x32x2 f32 MFMA chains, operands in registers
//   P1: P0 + operands re-read from LDS every k-step (ds_read_b32 x4), no global traffic
//   P2: P1 + per-stage barrier + register->LDS restaging of constant data (no global loads)
//   P3: P2 + global loads prefetch (coalesced) per stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int MODE>
__global__ void __launch_bounds__(256, 2) probe(const float* __restrict__ g, float* out, int stages, int ksteps) {
    __shared__ float As[2][48][129];
    __shared__ float Xs[2][3072];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fk = lane >> 5;
    for (int i = tid; i < 2 * 48 * 129; i += 256) (&As[0][0][0])[i] = 0.001f * (i % 7);
    for (int i = tid; i < 2 * 3072; i += 256) (&Xs[0][0])[i] = 0.002f * (i % 5);
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float ra0 = 0.5f + lane, ra1 = 0.25f, rb0 = 1.5f, rb1 = 0.75f;
    float areg[32], xreg[12];
    for (int i = 0; i < 32; ++i) areg[i] = 0.f;
    for (int i = 0; i < 12; ++i) xreg[i] = 0.f;
    const float* gp = g + (size_t)blockIdx.x * 4096 + tid;
    for (int s = 0; s < stages; ++s) {
        const int cur = s & 1;
        if (MODE >= 3) {
#pragma unroll
            for (int i = 0; i < 32; ++i) areg[i] = gp[(size_t)((s * 32 + i) & 15) * 256];
#pragma unroll
            for (int i = 0; i < 12; ++i) xreg[i] = gp[(size_t)((s * 12 + i) & 15) * 256 + 64];
        }
        const float* as_ = &As[cur][fk][wm * 64 + fr];
        const float* xs_ = &Xs[cur][(wn * 64 + fr) * 2 + fk];
        int xo = fk, j = 0;
        const int tbe = 11 + (stages & 1), D = 100 + (stages & 3), jc = (tbe - fk + 1) / 2 - 1, xo_lim = 2800 - fr;
        (void)xo_lim; (void)jc; (void)D; (void)tbe; (void)j; (void)xo;
        for (int ks = 0; ks < ksteps; ++ks) {
            float a0 = ra0, a1 = ra1, b0 = rb0, b1 = rb1;
            if (MODE >= 1 && MODE < 4) { a0 = as_[ks * 2 * 129]; a1 = as_[ks * 2 * 129 + 32]; b0 = xs_[ks * 2]; b1 = xs_[ks * 2 + 64]; }
            if (MODE >= 4) {
                // conv_gemm-style per-lane offset walk: clamp, two conditional jumps, wrap counter
                const int ka = min(ks * 2, 46) * 129;
                const int xoc = min(xo, xo_lim);
                a0 = as_[ka]; a1 = as_[ka + 32]; b0 = xs_[xoc]; b1 = xs_[xoc + 64];
                const int endj = (j == tbe - 1) ? D : 0;
                xo += 2 + ((j == jc) ? D : 0) + endj;
                if (MODE >= 5) xo &= 1023;
                j = (j == tbe - 1) ? 0 : j + 1;
            }
            acc[0][0] = MFMA(a0, b0, acc[0][0]);
            acc[0][1] = MFMA(a0, b1, acc[0][1]);
            acc[1][0] = MFMA(a1, b0, acc[1][0]);
            acc[1][1] = MFMA(a1, b1, acc[1][1]);
        }
        if (MODE >= 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kl = (tid & 31) + 32 * h;
                if (kl < 48)
#pragma unroll
                    for (int i = 0; i < 16; ++i) As[cur ^ 1][kl][(tid >> 5) + 8 * i] = areg[h * 16 + i] + 0.001f;
            }
#pragma unroll
            for (int t = 0; t < 12; ++t) Xs[cur ^ 1][tid + 256 * t] = xreg[t] + 0.002f;
            __syncthreads();
        }
    }
    float sum = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int MODE>
void run(const char* name, const float* g, float* out, int blocks) {
    const int stages = 64, ksteps = 24;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, g, out, stages, ksteps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, g, out, stages, ksteps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flop = (double)blocks * 4 * stages * ksteps * 4 * 4096.0;
    printf("%s blocks=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
}

int main() {
    float *g, *out;
    hipMalloc(&g, 64u << 20); hipMemset(g, 0, 64u << 20);
    hipMalloc(&out, 16u << 20);
    for (int blocks : {256, 512, 2048}) {
        run<0>("P0 regs only      ", g, out, blocks);
        run<1>("P1 +LDS reads     ", g, out, blocks);
        run<2>("P2 +restage+barrier", g, out, blocks);
        run<3>("P3 +global prefetch", g, out, blocks);
        run<4>("P4 P3 w/ offset walk", g, out, blocks);
    }
    return 0;
}

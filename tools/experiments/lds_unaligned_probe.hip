// Does ds_read_b128 work at 2-byte alignment on gfx950 (ROCm runs LDS in unaligned access mode), and how fast?
// A wgrad / Sinc B fragment of the split-bf16 contraction is 8 consecutive bf16 of a plane starting at an arbitrary
// sample: with unaligned 16-byte LDS reads the planes need no per-tap (Toeplitz) expansion.
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/experiments/lds_unaligned_probe tools/experiments/lds_unaligned_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) U4u { unsigned x, y, z, w; };

template <int OFF2>   // byte offset of the lane's read = lane * 16 + 2 * OFF2  (OFF2 = 0: aligned)
__global__ void probe(unsigned* out, unsigned long long* cycles, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384 + 64];
    for (int i = threadIdx.x; i < 16384 + 64; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned a = (unsigned)(size_t)&lds[((wave * 8 + j) * 64 + lane) * 8 + OFF2];   // LDS byte address
            u32x4 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    // one verified read
    const unsigned a1 = (unsigned)(size_t)&lds[lane * 8 + OFF2];
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a1) : "memory");
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 0] = v.x;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 1] = v.y;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 2] = v.z;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 3] = v.w + (acc & 0);
}

template <int OFF2>
void run() {
    unsigned* dout; unsigned long long* dcyc;
    hipMalloc(&dout, 256 * 4 * 4); hipMalloc(&dcyc, 8);
    probe<OFF2><<<1, 256>>>(dout, dcyc, 1000);
    unsigned out[1024]; unsigned long long cyc;
    if (hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost) != hipSuccess) { printf("off %d bytes: FAULT\n", 2 * OFF2); return; }
    hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) {
            const unsigned want = (unsigned)(l * 8 + OFF2 + e) & 0xffff;
            const unsigned got = (out[l * 4 + e / 2] >> (16 * (e & 1))) & 0xffff;
            bad += want != got;
        }
    printf("offset %2d bytes: %s, %.1f cycles per wave ds_read_b128 (4 waves, 8 reads per iteration)\n", 2 * OFF2, bad ? "WRONG DATA" : "ok",
           (double)cyc / (1000.0 * 8));
}

int main() {
    run<0>(); run<1>(); run<2>(); run<3>(); run<4>(); run<5>(); run<6>(); run<7>();
    return 0;
}

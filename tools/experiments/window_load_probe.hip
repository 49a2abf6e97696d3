// window_load_probe.hip -- what does one global_load_dwordx4 per lane cost when the 64 lanes read OVERLAPPING 16-byte
// windows (the Toeplitz columns of a weight gradient out of bf16 planes)?  One wave per SIMD, every lane issues N loads from
// an L2-resident buffer; cycles per wave-level load instruction (issue-to-issue, 8 loads in flight):
//   stride 16 B (disjoint, aligned)  /  stride 4 B (dword-aligned windows)  /  stride 2 B (2-byte-aligned windows)
//   /  stride 2 B rounded down to 4 (what the two-copy planes give)  /  6 rows x 11 shifts (a real 11-tap tile)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) U16 { u32x4 v; };

template <int MODE>
__global__ void __launch_bounds__(256, 1) probe(const char* buf, unsigned long long* out, unsigned* sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t off;
    if (MODE == 0) off = (size_t)lane * 16;
    else if (MODE == 1) off = (size_t)lane * 4;
    else if (MODE == 2) off = (size_t)lane * 2;
    else if (MODE == 3) off = (size_t)(lane * 2) & ~(size_t)3;
    else off = (size_t)(lane / 11) * 1664 + (size_t)(lane % 11) * 2;          // 6 channel rows, 11 shifts each
    const char* p = buf + (size_t)(blockIdx.x * 4 + wave) * 65536 + off;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = reinterpret_cast<const U16*>(p + (size_t)((i * 8 + k) & 31) * 128)->v;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc[0] == 0x12345678u) sink[0] = acc[1];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    char* buf;
    unsigned long long* d;
    unsigned* sink;
    hipMalloc(&buf, (size_t)256 * 4 * 65536 + 4096);
    hipMemset(buf, 1, (size_t)256 * 4 * 65536 + 4096);
    hipMalloc(&d, 256 * 8);
    hipMalloc(&sink, 4);
    const int iters = 2000;
    std::vector<unsigned long long> h(256);
    auto run = [&](auto kern, const char* name) {
        for (int r = 0; r < 2; ++r) {
            hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, buf, d, sink, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        printf("%-64s %7.1f ticks per wave-level global_load_dwordx4 (4 waves per CU issuing)\n", name, s / 256 / iters / 8);
    };
    run(probe<0>, "lanes 16 B apart (aligned, disjoint)");
    run(probe<1>, "lanes 4 B apart (overlapping dword-aligned windows)");
    run(probe<2>, "lanes 2 B apart (overlapping 2-byte-aligned windows)");
    run(probe<3>, "lanes 2 B apart rounded down to 4 (two-copy planes)");
    run(probe<4>, "6 rows x 11 shifts of 2 B (an 11-tap column tile)");
    return 0;
}

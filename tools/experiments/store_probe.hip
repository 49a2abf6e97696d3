// store_probe.hip -- what does the epilogue's store phase cost?  Each workgroup (4 waves, one per SIMD, as the compute waves
// of conv_x6c) writes 128 x 128 fp32 tiles of a (rows x cols) matrix, each wave a 32 x 128 block, then "works" for `gap`
// clocks (s_sleep: the main loop of the next tile), as a persistent grid of `grid` workgroups.
//   mode 0: global_store_dword  as the MFMA accumulator layout gives them (lanes 0-31: 32 consecutive columns of row r,
//           lanes 32-63: of row r + 4; 64 instructions per wave)
//   mode 1: global_store_dwordx4, row-major (lanes 0-31: 128 consecutive columns of row r, lanes 32-63: row r + 1; 16 per wave)
//   mode 2: global_store_dwordx2 (lanes 0-31: 64 columns of row r, 32-63: row r + 1 ...; 32 per wave)
// Reported: clocks from the first store's issue to the last store's issue (what the issuing wave loses), per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256, 1) probe(float* y, unsigned long long* out, int rows, int cols, int pitch, int gap,
                                                 int stagger) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nct = cols / 128, ntiles = (rows / 128) * nct;
    unsigned long long acc = 0;
    int n = 0;
    for (int i = 0; i < (int)((blockIdx.x >> 3) & 15) * stagger; ++i) __builtin_amdgcn_s_sleep(8);
    for (int item = blockIdx.x; item < ntiles; item += gridDim.x) {
        const int mt = item / nct, nt = item - mt * nct;
        char* base = reinterpret_cast<char*>(y) + ((size_t)(mt * 128 + wave * 32) * pitch + (size_t)nt * 128) * 4;
        const float v = (float)item;
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (MODE == 0) {
            const unsigned lo = (unsigned)((4 * (lane >> 5)) * pitch + (lane & 31)) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                char* row = base + (size_t)((r & 3) + 8 * (r >> 2)) * pitch * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<float*>(row + lo + j * 128) = v;
            }
        } else if (MODE == 1) {
            const unsigned lo = (unsigned)((lane >> 5) * pitch + 4 * (lane & 31)) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                char* row = base + (size_t)(2 * r) * pitch * 4;
                *reinterpret_cast<f32x4*>(row + lo) = f32x4{v, v, v, v};
            }
        } else {
            const unsigned lo = (unsigned)((lane >> 5) * pitch + 2 * (lane & 31)) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                char* row = base + (size_t)(2 * r) * pitch * 4;
#pragma unroll
                for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x2*>(row + lo + j * 256) = f32x2{v, v};
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        acc += t1 - t0;
        ++n;
        for (int i = 0; i < gap / 512; ++i) __builtin_amdgcn_s_sleep(8);
    }
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = acc;
        out[2 * blockIdx.x + 1] = n;
    }
}

int main() {
    const int rows = 16384, cols = 6400;
    float* y;
    unsigned long long* d;
    hipMalloc(&d, 256 * 16);
    std::vector<unsigned long long> h(512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int pitch : {6400, 6404}) {
        hipMalloc(&y, (size_t)rows * pitch * 4);
        auto run = [&](auto kern, const char* name, int grid, int gap, int stagger) {
            float ms = 0;
            for (int r = 0; r < 2; ++r) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, y, d, rows, cols, pitch, gap, stagger);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
            double s = 0, n = 0;
            for (int b = 0; b < grid; ++b) { s += (double)h[2 * b]; n += (double)h[2 * b + 1]; }
            const double bytes = (double)rows * cols * 4 * ((double)n / ((rows / 128) * (cols / 128)));
            printf("pitch %5d %-22s grid %3d gap %6d stagger %2d: %8.0f clocks of store issue per tile, %7.3f ms, %6.2f TB/s\n",
                   pitch, name, grid, gap, stagger, s / n, ms, bytes / ms * 1e-9);
        };
        for (int grid : {256, 8}) {
            for (int gap : {0, 40000}) {
                for (int stagger : {0, 5}) {
                    if (gap == 0 && stagger) continue;
                    run(probe<0>, "dword (MFMA layout)", grid, gap, stagger);
                    run(probe<2>, "dwordx2 row-major", grid, gap, stagger);
                    run(probe<1>, "dwordx4 row-major", grid, gap, stagger);
                }
            }
        }
        hipFree(y);
    }
    return 0;
}

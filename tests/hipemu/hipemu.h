// hipemu.h -- a tiny SIMT emulator so the *same* HIP kernel sources under pase_amd/csrc/ can be
// compiled with g++ and executed on the CPU at toy sizes.
//
// TEST INFRASTRUCTURE ONLY.  The product (pase_amd/_lib.py) never loads the emulated library; it
// exists because the build container has no GPU and GPU minutes are scarce: indexing / tiling /
// fragment-layout bugs are caught here first (tests/test_emu_*.py, `-m "not gpu"`), and the same
// tests then run against the real gfx950 library under `-m gpu`.
//
// Model: one workgroup = N fibers (ucontext) multiplexed on ONE OS thread; __syncthreads() and the
// wave-collective ops (MFMA, shuffles) are fiber barriers.  Workgroups of a grid are distributed
// over a small pool of OS threads, so global atomics are real atomics.  `__shared__` becomes
// `static thread_local` (one copy per OS thread == per resident workgroup).
//
// MFMA semantics follow /opt/skills/guides/cdna_hip_programming.md §3:
//   v_mfma_f32_32x32x2_f32:  A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//                            D: col=l&31, row=(reg&3)+8*(reg>>2)+4*(l>>5), k-ordered fmaf chain.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace hipemu {

struct uint3_ { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct BlockRunner;
struct ThreadCtx {
    uint3_ tid, bid;
    dim3 bdim, gdim;
    int flat, lane, wave;
    BlockRunner* runner;
};
extern thread_local ThreadCtx* cur;

void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void sync_block();
void sync_wave();
uint64_t* wave_buf(int slot);  // 64-entry exchange buffer of the calling fiber's wave (slot 0/1)

}  // namespace hipemu

using hipemu::dim3;
#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::cur->bid)
#define blockDim (hipemu::cur->bdim)
#define gridDim (hipemu::cur->gdim)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3

inline void __syncthreads() { hipemu::sync_block(); }

// ---------------------------------------------------------------- vector types
typedef float f32x16 __attribute__((vector_size(64)));
typedef float f32x4 __attribute__((vector_size(16)));
typedef uint32_t u32x4 __attribute__((vector_size(16)));
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

// ---------------------------------------------------------------- wave collectives
template <class T>
inline T emu_xchg(T v, int src_lane_of_me) {
    static_assert(sizeof(T) <= 8, "emu_xchg");
    uint64_t* b = hipemu::wave_buf(0);
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    b[hipemu::cur->lane] = bits;
    hipemu::sync_wave();
    uint64_t rb = b[src_lane_of_me & 63];
    hipemu::sync_wave();
    T r;
    memcpy(&r, &rb, sizeof(T));
    return r;
}
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return emu_xchg(v, hipemu::cur->lane ^ mask); }
template <class T>
inline T __shfl_down(T v, int d, int width = 64) {
    int l = hipemu::cur->lane;
    int src = l + d;
    if ((l % width) + d >= width) src = l;
    return emu_xchg(v, src);
}
template <class T>
inline T __shfl(T v, int src, int width = 64) {
    int l = hipemu::cur->lane;
    return emu_xchg(v, (l / width) * width + (src % width));
}

inline f32x16 emu_mfma_32x32x2(float a, float b, f32x16 c) {
    uint64_t* ba = hipemu::wave_buf(0);
    uint64_t* bb = hipemu::wave_buf(1);
    int l = hipemu::cur->lane;
    uint32_t ua, ub;
    memcpy(&ua, &a, 4);
    memcpy(&ub, &b, 4);
    ba[l] = ua;
    bb[l] = ub;
    hipemu::sync_wave();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            uint32_t xa = (uint32_t)ba[row + 32 * k], xb = (uint32_t)bb[col + 32 * k];
            float fa, fb;
            memcpy(&fa, &xa, 4);
            memcpy(&fb, &xb, 4);
            acc = fmaf(fa, fb, acc);
        }
        c[r] = acc;
    }
    hipemu::sync_wave();
    return c;
}

// v_mfma_f32_32x32x16_bf16: lane l holds A[l&31][8*(l>>5)+e] and B[8*(l>>5)+e][l&31], e = 0..7 (bf16, low half of
// dword 0 first).  The 16 products of an output element are summed exactly (double) and added to c once.
inline f32x16 emu_mfma_bf16_32x32x16(u32x4 a, u32x4 b, f32x16 c) {
    uint64_t* ba = hipemu::wave_buf(0);
    uint64_t* bb = hipemu::wave_buf(1);
    const int l = hipemu::cur->lane;
    uint32_t la[64][4], lb[64][4];
    for (int h = 0; h < 2; ++h) {
        ba[l] = (uint64_t)a[2 * h] | ((uint64_t)a[2 * h + 1] << 32);
        bb[l] = (uint64_t)b[2 * h] | ((uint64_t)b[2 * h + 1] << 32);
        hipemu::sync_wave();
        for (int j = 0; j < 64; ++j) {
            la[j][2 * h] = (uint32_t)ba[j];
            la[j][2 * h + 1] = (uint32_t)(ba[j] >> 32);
            lb[j][2 * h] = (uint32_t)bb[j];
            lb[j][2 * h + 1] = (uint32_t)(bb[j] >> 32);
        }
        hipemu::sync_wave();
    }
    auto bf = [](const uint32_t (&v)[4], int e) {
        const uint32_t bits = ((v[e >> 1] >> (16 * (e & 1))) & 0xffffu) << 16;
        float f;
        memcpy(&f, &bits, 4);
        return (double)f;
    };
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double acc = 0.0;
        for (int kg = 0; kg < 2; ++kg)
            for (int e = 0; e < 8; ++e) acc += bf(la[row + 32 * kg], e) * bf(lb[col + 32 * kg], e);
        c[r] = (float)((double)c[r] + acc);
    }
    return c;
}

// ---------------------------------------------------------------- atomics
inline float atomicAdd(float* p, float v) {
    uint32_t* u = (uint32_t*)p;
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
    for (;;) {
        float f;
        memcpy(&f, &old, 4);
        f += v;
        uint32_t nw;
        memcpy(&nw, &f, 4);
        if (__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            float r;
            memcpy(&r, &old, 4);
            return r;
        }
    }
}
inline double atomicAdd(double* p, double v) {
    uint64_t* u = (uint64_t*)p;
    uint64_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
    for (;;) {
        double f;
        memcpy(&f, &old, 8);
        f += v;
        uint64_t nw;
        memcpy(&nw, &f, 8);
        if (__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            double r;
            memcpy(&r, &old, 8);
            return r;
        }
    }
}
inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

// ---------------------------------------------------------------- math shims
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __expf(float x) { return expf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline double rsqrt(double x) { return 1.0 / sqrt(x); }
#include <algorithm>
using std::min;
using std::max;

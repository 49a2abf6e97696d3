// hip_compat.h -- the one place that differs between the real gfx950 build (hipcc) and the
// CPU SIMT-emulator build used by the `-m "not gpu"` kernel tests (tests/hipemu, -DPASE_HIPEMU).
// Kernel sources include only this header.
#pragma once

#ifdef PASE_HIPEMU
#include "hipemu.h"
#define PASE_LAUNCH(kernel, grid, block, stream, ...) \
    hipemu::launch(grid, block, [=]() { kernel(__VA_ARGS__); })
__device__ __forceinline__ f32x16 pase_mfma_32x32x2(float a, float b, f32x16 c) {
    return emu_mfma_32x32x2(a, b, c);
}
__device__ __forceinline__ f32x16 pase_mfma_bf16_32x32x16(u32x4 a, u32x4 b, f32x16 c) {
    return emu_mfma_bf16_32x32x16(a, b, c);
}
__device__ __forceinline__ unsigned pase_pack_hi16(unsigned lo_src, unsigned hi_src) {
    return (lo_src >> 16) | (hi_src & 0xffff0000u);
}
__device__ __forceinline__ int pase_uniform(int v) { return v; }
#define PASE_LAUNDER(x) ((void)0)
#define PASE_SCHED_BARRIER() ((void)0)
#define PASE_SGB(mask, n) ((void)0)
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define PASE_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
// v_mfma_f32_32x32x2_f32: exact fp32 (k-ordered fmaf chain), 64 cycles/SIMD, 157 TFLOP/s chip peak.
// Fragment layout (cdna_hip_programming.md §3): A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
// D: col=l&31, row=(reg&3)+8*(reg>>2)+4*(l>>5).
__device__ __forceinline__ f32x16 pase_mfma_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_32x32x16_bf16 (8 passes, 16x the fp32 MFMA rate): A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31],
// e = the lane's eight bf16 (low half of dword 0 first); D layout as above.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pase_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 pase_mfma_bf16_32x32x16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pase_bf16x8, a),
                                                   __builtin_bit_cast(pase_bf16x8, b), c, 0, 0, 0);
}
// {lo_src[31:16], hi_src[31:16]} -> one dword of two bf16 (v_perm_b32)
__device__ __forceinline__ unsigned pase_pack_hi16(unsigned lo_src, unsigned hi_src) {
    return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u);
}
// value known to be identical across the wave (e.g. threadIdx.x / 64): make it an SGPR so branches
// on it are scalar (cdna_hip_programming.md T20: threadIdx-derived values are divergent to hipcc)
__device__ __forceinline__ int pase_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// make the compiler forget what it knows about x (blocks loop-invariant hoisting of values derived from it)
#define PASE_LAUNDER(x) asm volatile("" : "+v"(x))
// pin the instruction order across this point (software-pipelined ds_read -> MFMA loops)
#define PASE_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// ask the scheduler for `n` instructions of class `mask` next (0x8 MFMA, 0x2 VALU, 0x4 SALU, 0x100 DS read)
#define PASE_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif

// All lanes of this wave have executed everything before this point (LDS exchanged between the lanes of ONE wave needs no
// workgroup barrier: a wave's LDS operations execute in order; the compiler just must not move them across)
#ifdef PASE_HIPEMU
#define PASE_SLEEP(n) ((void)0)
#else
#define PASE_SLEEP(n) __builtin_amdgcn_s_sleep(n)     // n x 64 clocks
#endif
#ifdef PASE_HIPEMU
__device__ __forceinline__ void pase_wave_sync() { (void)__shfl_xor(0, 1); }
#else
__device__ __forceinline__ void pase_wave_sync() { __builtin_amdgcn_wave_barrier(); }
#endif

#ifdef PASE_HIPEMU
__device__ __forceinline__ int pase_wave_all(int pred) {
    int v = pred ? 1 : 0;
    for (int m = 1; m < 64; m <<= 1) v &= __shfl_xor(v, m);
    return v;
}
// sum over the 32 lanes sharing (lane>>5); the result is only guaranteed in lane 31 of each half
__device__ __forceinline__ float pase_half_sum_lane31(float v) {
    for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
#else
__device__ __forceinline__ int pase_wave_all(int pred) { return __all(pred); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float pase_dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(moved);
}
// DPP tree (no LDS crossbar): quad xor 1, xor 2, row_half_mirror, row_mirror, then row_bcast15 into
// rows 1 and 3 -> lanes 16-31 / 48-63 hold the sum of their 32-lane half (lane 31 / 63 is the reader)
__device__ __forceinline__ float pase_half_sum_lane31(float v) {
    v = pase_dpp_add<0xB1, 0xf>(v);
    v = pase_dpp_add<0x4E, 0xf>(v);
    v = pase_dpp_add<0x141, 0xf>(v);
    v = pase_dpp_add<0x140, 0xf>(v);
    v = pase_dpp_add<0x142, 0xa>(v);
    return v;
}
#endif

#define PASE_CHECK_LAUNCH()                      \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

// runs of consecutive floats at a 4-byte-aligned address (global_store_dwordx4 / x2: ROCm runs the memory pipeline in
// unaligned access mode; the packed types tell the compiler not to assume more than dword alignment)
struct __attribute__((packed, aligned(4))) pase_f4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) pase_f2u { float x, y; };
__device__ __forceinline__ void pase_store_run4(float* d, const float (&v)[4]) {
    *reinterpret_cast<pase_f4u*>(d) = pase_f4u{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void pase_store_run2(float* d, float a, float b) {
    *reinterpret_cast<pase_f2u*>(d) = pase_f2u{a, b};
}

// x = hi + mid + lo with three truncated bf16 pieces (8 + 8 + 8 mantissa bits: exact for normal x);
// eight values -> three fragments of eight bf16 each (element e of a fragment = piece of x[e])
__device__ __forceinline__ void pase_split_bf16x3(const float (&x)[8], u32x4 (&out)[3]) {
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
        for (int i = 0; i < 4; ++i) out[s][i] = pase_pack_hi16(b[2 * i], b[2 * i + 1]);
    }
    // last piece: the pack takes the upper halves of the remainders as they are
#pragma unroll
    for (int i = 0; i < 4; ++i) out[2][i] = pase_pack_hi16(__float_as_uint(r[2 * i]), __float_as_uint(r[2 * i + 1]));
}

// Round-to-nearest pieces (the form the x6c kernels use): hi = bf16_rne(x), mid = bf16_rne(x - hi), lo = bf16_rne(x - hi -
// mid); both remainders are exact in fp32.  Unlike the truncated pieces above, the dropped part of a product (ml + lm + ll
// and the rounding of lo) has no preferred sign, so sums of same-signed products carry no systematic error
// (tools/experiments/x6_accum_probe.hip: mean error / sum|ab| -4e-8 truncated, -4e-9 rounded).  One v_cvt_pk_bf16_f32 per
// pair and level; same VALU count as the truncating split.
#ifdef PASE_HIPEMU
__device__ __forceinline__ unsigned pase_cvt_pk_bf16(float lo, float hi) {
    auto rne = [](float f) {
        unsigned u = __float_as_uint(f);
        if ((u & 0x7f800000u) == 0x7f800000u) return u >> 16;         // inf / nan: keep the top half
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    };
    return rne(lo) | (rne(hi) << 16);
}
#else
typedef __bf16 pase_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pase_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pase_cvt_pk_bf16(float lo, float hi) {
    const pase_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pase_bf16x2));
}
#endif
// (Non-finite values: hi = bf16(Inf) = Inf and the remainders Inf - Inf are NaN, so an infinite operand makes every product
//  it enters NaN where the fp32 pipe gives +-Inf.  Zeroing the lower pieces of a non-finite value does NOT restore +-Inf: the
//  OTHER operand's mid / lo pieces have arbitrary signs (round-to-nearest remainders) and zeros, so mid * Inf is -+Inf or NaN
//  and the six-term sum is NaN again for about half of all weights (tried in round 4, tests/test_conv_x6c.py
//  ::test_infinite_activation_stays_non_finite_where_the_reference_is).  The contract is the footprint: exactly the outputs
//  that are non-finite in fp32 arithmetic are non-finite here, every other output is unaffected.)
__device__ __forceinline__ void pase_split_bf16x3_rne(const float (&x)[8], u32x4 (&out)[3]) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = x[i];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned pk = pase_cvt_pk_bf16(r[2 * i], r[2 * i + 1]);
            out[s][i] = pk;
            r[2 * i] -= __uint_as_float(pk << 16);
            r[2 * i + 1] -= __uint_as_float(pk & 0xffff0000u);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[2][i] = pase_cvt_pk_bf16(r[2 * i], r[2 * i + 1]);
}

// nine consecutive values -> the pieces of the two 8-element windows starting at element 0 and element 1:
//   w0[pz] = pieces pz of v[0..7], w1[pz] = pieces pz of v[1..8]   (pase_split_bf16x3_rne's arithmetic on both pairings;
//   used where a fragment must start at ANY element: the second window makes every start dword-aligned)
__device__ __forceinline__ void pase_split_two_windows(const float (&v)[9], u32x4 (&w0)[3], u32x4 (&w1)[3]) {
    float r[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] = v[i];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        unsigned pe[4], po[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pe[i] = pase_cvt_pk_bf16(r[2 * i], r[2 * i + 1]);          // (0,1) (2,3) (4,5) (6,7)
            po[i] = pase_cvt_pk_bf16(r[2 * i + 1], r[2 * i + 2]);      // (1,2) (3,4) (5,6) (7,8)
            w0[s][i] = pe[i];
            w1[s][i] = po[i];
        }
        if (s < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[2 * i] -= __uint_as_float(pe[i] << 16);
                r[2 * i + 1] -= __uint_as_float(pe[i] & 0xffff0000u);
            }
            r[8] -= __uint_as_float(po[3] & 0xffff0000u);
        }
    }
}

// four values -> three pieces x two dwords (half a fragment)
__device__ __forceinline__ void pase_split_bf16x3_quad(const float (&x)[4], unsigned (&out)[3][2]) {
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = x[i];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        unsigned b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b[i] = __float_as_uint(r[i]) & 0xffff0000u;
            r[i] -= __uint_as_float(b[i]);
        }
        out[s][0] = pase_pack_hi16(b[0], b[1]);
        out[s][1] = pase_pack_hi16(b[2], b[3]);
    }
    out[2][0] = pase_pack_hi16(__float_as_uint(r[0]), __float_as_uint(r[1]));
    out[2][1] = pase_pack_hi16(__float_as_uint(r[2]), __float_as_uint(r[3]));
}

__device__ __forceinline__ float pase_wave_sum32(float v) {
    // sum over the 32 lanes that share (lane>>5); result valid in every lane of the half-wave
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}
__device__ __forceinline__ float pase_wave_sum64(float v) {
    v = pase_wave_sum32(v);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ double pase_wave_sum64d(double v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
#include <type_traits>
#include <utility>
template <int... Is, class F>
__host__ __device__ __forceinline__ void pase_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__host__ __device__ __forceinline__ void pase_static_for(F&& f) {
    pase_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

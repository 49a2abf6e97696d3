// mlp_head1.hip -- the pointwise tail of the waveform decoder, forward AND backward in one pass over its input.
//
//   a0 = PReLU(y, alpha0)            y: (S, 128, T) raw output of the last GDeconv1DBlock (norm_type None: no statistics
//   y1 = W1 a0 + b1                     stand between the layers, pase/models/modules.py:558-589)
//   a1 = PReLU(y1, alpha1)           MLPBlock(128 -> 64, context 1)            (modules.py:527-556)
//   pred = w2 . a1 + b2              DecoderMinion.W = Conv1d(64, 1, 1)        (Minions/minions.py:416-417, 446)
//   loss = L(pred, target)           nn.L1Loss via ContextualizedLoss(r=None)  (pase/losses.py:33-37)
//   and their autograd: dw2, db2, dalpha1, dW1, db1, dalpha0, sum dy (the deconvolution's bias gradient), dy = dL/dy.
//
// What it replaces in the bs32 step (32 x 128 x 32000 = 524 MB per tensor): the 64-row 1x1 convolution, head1_fwd, head1_bwd,
// the 1x1 weight gradient, the 128-row 1x1 data gradient and the PReLU backward pass over y -- six launches that write and
// re-read y1 (262 MB), dy1 (262 MB) and dA0 (524 MB) and read y three times.  Here a workgroup takes a tile of 128 time
// steps x all 128 channels into LDS once and walks three small GEMMs over it on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: the arithmetic of the launches it replaces), wave w owning the tile's time steps 32 w .. 32 w + 31.
// Every product is oriented so that an accumulator tile has TIME STEPS in its rows (registers) and FEATURES in its columns
// (lanes): the per-feature sums of the backward (dw2, dalpha1, db1; the layer below's bias gradient and dalpha0) are then
// in-lane sums over a lane's 16 time steps -- two registers per statistic and tile instead of a register per (row, tile),
// which is what a first build with features in the rows needed (352 live accumulators / partial sums: 499 spilled VGPRs).
//   1. Y1^T (32 x 64) = A0^T (32 x 128) W1^T          A fragments = PReLU(y) read from the LDS tile
//      head: pred of a time step = sum over the 64 hidden lanes (xor-shuffles), loss, dpred; dy1 in place and into LDS
//   2. dA0^T (32 x 128) = dY1^T (32 x 64) W1           A fragments = this wave's own dy1 rows from LDS (no barrier)
//      -> dy = dA0 * PReLU'(y): a lane holds 4 consecutive time steps of its channel per register quad (16-byte stores)
//   3. dW1^T (128 x 64) += A0 (128 x P) dY1^T (P x 64)  contraction over the tile's 128 time steps behind one barrier, wave w
//      owning channels 32 w .. 32 w + 31; accumulated across the workgroup's tiles, one atomic flush.
// 384 MFMAs per wave and tile; HBM traffic = one read of y, one write of dy.
#include <type_traits>

#include "hip_compat.h"
#include "pase_amd.h"

namespace {

constexpr int MH_NT = 256;
constexpr int MH_C = 128, MH_H = 64, MH_P = 128;
constexpr int MH_PY = MH_P + 1;       // floats per channel row of the y tile (odd: rows and columns both walk all banks)
constexpr int MH_PW = MH_C + 1;       // floats per hidden row of W1
constexpr int MH_PD = MH_H + 1;       // floats per time step of the transposed dy1 tile
constexpr int MH_UB = 4;              // MFMA steps whose fragments are read ahead together

// (tools/mlp_head1_ablate.sh: timing builds that leave one phase out -- never defined in the product build)
#ifdef MH_ABL_NO1
#define MH_MFMA1(a, b, c) mh_fake(a, b, c)
#else
#define MH_MFMA1(a, b, c) pase_mfma_32x32x2(a, b, c)
#endif
#ifdef MH_ABL_NO2
#define MH_MFMA2(a, b, c) mh_fake(a, b, c)
#else
#define MH_MFMA2(a, b, c) pase_mfma_32x32x2(a, b, c)
#endif
#ifdef MH_ABL_NO3
#define MH_MFMA3(a, b, c) mh_fake(a, b, c)
#else
#define MH_MFMA3(a, b, c) pase_mfma_32x32x2(a, b, c)
#endif
__device__ __forceinline__ f32x16 mh_fake(float a, float b, f32x16 c) {      // keeps the operand reads alive, one VALU instead of an MFMA
    c[0] = fmaf(a, b, c[0]);
    return c;
}

// -DMH_TRACE (tools/mlp_head1_ablate.sh trace): shader-clock stamps at the phase boundaries of wave 0 of workgroups 0 and 131,
// summed over the workgroup's tiles: slot i = clocks from stamp i's predecessor to stamp i
#ifdef MH_TRACE
__device__ unsigned long long g_mh_trace[2 * 16];
#define MH_TRACE_BEGIN() unsigned long long mh_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mh_last = clock64()
#define MH_STAMP(i)                                   \
    do {                                              \
        const unsigned long long mh_now = clock64();  \
        mh_t[i] += mh_now - mh_last;                  \
        mh_last = mh_now;                             \
    } while (0)
#define MH_TRACE_END()                                                                            \
    do {                                                                                          \
        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 131))                           \
            for (int i = 0; i < 10; ++i) g_mh_trace[(blockIdx.x ? 16 : 0) + i] = mh_t[i];         \
    } while (0)
#else
#define MH_TRACE_BEGIN() ((void)0)
#define MH_STAMP(i) ((void)0)
#define MH_TRACE_END() ((void)0)
#endif

__device__ __forceinline__ int mh_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// one dword per lane straight from global memory into LDS (global_load_lds_dword: destination = wave-uniform LDS address +
// 4 * lane; nothing passes through VGPRs) and the wait for this wave's copies
#ifdef PASE_HIPEMU
__device__ __forceinline__ void mh_load_lds4(const float* src, float* lds_wave_base, int lane) { lds_wave_base[lane] = *src; }
__device__ __forceinline__ void mh_dma_wait() {}
#else
__device__ __forceinline__ void mh_load_lds4(const float* src, float* lds_wave_base, int) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ void mh_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// LOSS: the loss of the head, a compile-time constant -- as a run-time switch inside the unrolled head the compiler laid all
// three losses (the BCE's exp / log1p expansions included) out sixteen times: 4 600 instructions, 10 k clocks per tile
template <int LOSS>
__global__ void __launch_bounds__(MH_NT) mlp_head1_kernel(PaseMlpHead1 p, int tiles_per_seq, long ntiles) {
    __shared__ float Ys[MH_C * MH_PY];
    __shared__ float W1s[MH_H * MH_PW];
    __shared__ float Ds[MH_P * MH_PD];
    __shared__ float a0s[MH_C];

    const int tid = threadIdx.x, lane = tid & 63, wave = pase_uniform(tid >> 6);
    const int l31_k = lane & 31, half_k = lane >> 5;
    const int p0w = 32 * wave;                        // this wave's first time step inside a tile (stages 1 and 2)

    for (int i = tid; i < MH_H * MH_C; i += MH_NT) W1s[(i >> 7) * MH_PW + (i & 127)] = p.w1[i];
    for (int i = tid; i < MH_C; i += MH_NT) a0s[i] = p.alpha0 ? p.alpha0[i] : 1.f;
    // per-lane constants: hidden rows 32 t + l31 (stages 1 / head), channels 32 c + l31 (stage 2), channel 32 w + l31 (stage 3)
    float b1v[2], a1v[2], w2v[2], a0v[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int h = 32 * t + l31_k;
        b1v[t] = p.b1 ? p.b1[h] : 0.f;
        a1v[t] = p.alpha1 ? p.alpha1[h] : 1.f;
        w2v[t] = p.w2[h];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) a0v[c] = p.alpha0 ? p.alpha0[32 * c + l31_k] : 1.f;
    const float a0_mine = p.alpha0 ? p.alpha0[32 * wave + l31_k] : 1.f;
    const float b2 = p.b2 ? p.b2[0] : 0.f;
    const bool vec4 = (p.T & 3) == 0 && (((unsigned long long)(size_t)p.dy) & 15) == 0;      // uniform: 16-byte dy stores
    __syncthreads();

    f32x16 acc3[2];                                   // dW1^T tiles (channels 32 w .. in the rows, hidden 32 t + l31), all tiles
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[t][r] = 0.f;
    float st_dw2[2] = {0.f, 0.f}, st_da1[2] = {0.f, 0.f}, st_db1[2] = {0.f, 0.f};      // hidden row 32 t + l31, this half's steps
    float st_dy0[4] = {0.f, 0.f, 0.f, 0.f}, st_da0[4] = {0.f, 0.f, 0.f, 0.f};           // channel 32 c + l31
    double lsum = 0.0, s_db2 = 0.0;
    float pf0 = 0.f, pf1 = 0.f;       // L2 prefetch of the next tile (stage 3)

    MH_TRACE_BEGIN();
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        MH_STAMP(8);
        const int s = (int)((unsigned)tile / (unsigned)tiles_per_seq);        // (32-bit: the 64-bit scalar division is a ~130-
        const int t0 = ((int)tile - s * tiles_per_seq) * MH_P;                //  instruction routine, once more for the prefetch)
        const int nv = min(MH_P, p.T - t0);                                   // live time steps of the tile
        // LDS indices = a per-lane base + a compile-time constant (the pitches are odd: left to the compiler, "(row) * 129 +
        // col" with the row depending on the half or the wave became ~40 hoisted address registers, most of them spilled).
        // The bases are recomputed per tile from laundered lane ids (a dozen integer operations): hoisted out of the tile loop
        // they were live across all of it and the register allocator parked them in scratch, a reload with s_waitcnt
        // vmcnt(0) -- which also waits for the previous tile's dy stores -- at every use.
        int l31 = l31_k, half = half_k;
        PASE_LAUNDER(l31);
        PASE_LAUNDER(half);
        const int ys_pos = half * MH_PY + p0w + l31;              // Ys[k][p0w + l31],        k = 2 j + half:  + 2 j * MH_PY
        const int w1_row = l31 * MH_PW + half;                    // W1s[32 t + l31][k]:                        + 32 t * MH_PW + 2 j
        const int ds_own = (p0w + l31) * MH_PD + half;            // Ds[p0w + l31][k]:                          + 2 j
        const int w1_col = half * MH_PW + l31;                    // W1s[k][32 c + l31]:                        + 2 j * MH_PW + 32 c
        const int ds_st = (p0w + 4 * half) * MH_PD + l31;         // Ds[p0w + mh_row(r, half)][32 t + l31]:     + mh_row(r, 0) * MH_PD + 32 t
        const int ys_ch = l31 * MH_PY + p0w + 4 * half;           // Ys[32 c + l31][p0w + 4 half + ..]:         + 32 c * MH_PY + 8 g + e
        const int ys_row = (32 * wave + l31) * MH_PY + half;      // Ys[32 w + l31][q],       q = 2 j + half:  + 2 j
        const int ds_col = half * MH_PD + l31;                    // Ds[q][32 t + l31]:                         + 2 j * MH_PD + 32 t
        // ---- targets of this lane's 16 time steps (in flight under the tile copy).  Every global address below is a
        //      wave-uniform base plus a 32-bit lane offset: with 64-bit per-lane pointers the compiler hoisted ~50 address
        //      pairs out of the tile loop and spilled them (170 VGPRs, a scratch reload -- s_waitcnt vmcnt(0) -- every few
        //      MFMAs) ---------------------------------------------------------------------------------------------------
        const size_t ob = (size_t)s * p.T + t0 + p0w;                          // (uniform)
        const unsigned hq = 4u * (unsigned)half;                               // mh_row(r, half) = hq + a constant of r
        float tgv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) tgv[r] = 0.f;
        if (LOSS != PASE_LOSS_NONE) {
            const float* tgb = p.target + ob;
            if (nv == MH_P) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tgv[r] = tgb[hq + (unsigned)mh_row(r, 0)];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (p0w + mh_row(r, half) < nv) tgv[r] = tgb[hq + (unsigned)mh_row(r, 0)];
            }
        }
        // ---- the y tile: rows of 128 consecutive floats by LDS DMA (two 64-lane copies per channel row, wave w the rows
        //      32 w ..), time steps past the end of a ragged tile are zeros ------------------------------------------------
        {
            const float* yb = p.y + ((size_t)s * MH_C + 32 * wave) * (size_t)p.T + t0;      // (uniform)
#ifdef MH_ABL_NOCOPY
            if (tile != (long)blockIdx.x) {
            } else
#endif
            if (nv == MH_P) {       // uniform
#pragma unroll 8
                for (int rr = 0; rr < 32; ++rr) {
                    mh_load_lds4(yb + (size_t)rr * p.T + lane, &Ys[(32 * wave + rr) * MH_PY], lane);
                    mh_load_lds4(yb + (size_t)rr * p.T + 64 + lane, &Ys[(32 * wave + rr) * MH_PY + 64], lane);
                }
                mh_dma_wait();
#ifndef PASE_HIPEMU
                asm volatile("" ::"v"(pf0), "v"(pf1));      // (the prefetch loads' destination registers stay theirs until here)
#endif
            } else {
                for (int rr = 0; rr < 32; ++rr)
                    for (int q = lane; q < MH_P; q += 64) Ys[(32 * wave + rr) * MH_PY + q] = q < nv ? yb[(size_t)rr * p.T + q] : 0.f;
            }
        }
        __syncthreads();
        MH_STAMP(0);

        // ---- 1. Y1^T = A0^T W1^T for this wave's 32 time steps: rows = steps, columns = hidden 32 t + l31 ---------------
        f32x16 acc1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
        // (one wave per SIMD: nothing but this wave's own MFMAs covers an LDS read -- the fragments of the next MH_UB steps are
        //  read in front of this block's MFMAs, in all three products)
        {
            float yc[MH_UB], lc[MH_UB], bc[MH_UB][2];
            auto ld = [&](int blk, float (&yy)[MH_UB], float (&ll)[MH_UB], float (&bb)[MH_UB][2]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < MH_UB; ++u) {
                    const int j = blk * MH_UB + u;
                    yy[u] = Ys[ys_pos + 2 * j * MH_PY];
                    ll[u] = a0s[half + 2 * j];
#pragma unroll
                    for (int t = 0; t < 2; ++t) bb[u][t] = W1s[w1_row + 32 * t * MH_PW + 2 * j];
                }
            };
            ld(0, yc, lc, bc);
#pragma unroll
            for (int blk = 0; blk < MH_C / 2 / MH_UB; ++blk) {
                float yn[MH_UB], ln[MH_UB], bn[MH_UB][2];
                if (blk + 1 < MH_C / 2 / MH_UB) ld(blk + 1, yn, ln, bn);
#pragma unroll
                for (int u = 0; u < MH_UB; ++u) {
                    const float neg = yc[u] * lc[u];
                    const float a = yc[u] > 0.f ? yc[u] : neg;
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc1[t] = MH_MFMA1(a, bc[u][t], acc1[t]);
                }
                if (blk + 1 < MH_C / 2 / MH_UB) {
#pragma unroll
                    for (int u = 0; u < MH_UB; ++u) {
                        yc[u] = yn[u];
                        lc[u] = ln[u];
                        bc[u][0] = bn[u][0];
                        bc[u][1] = bn[u][1];
                    }
                }
            }
        }
        MH_STAMP(1);
        // ---- head: register r = time step p0w + mh_row(r, half); the 64 hidden values of a step sit in the 32 lanes of
        //      this half (two row tiles): pred = xor-shuffle sum ------------------------------------------------------------
        float dpred[16];
        {
            float part[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                part[r] = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float y1 = acc1[t][r] + b1v[t];
                    const float neg = y1 * a1v[t];
                    part[r] = fmaf(w2v[t], y1 > 0.f ? y1 : neg, part[r]);
                }
            }
            // the five exchange rounds over all 16 values at once (16 independent exchanges in flight per round)
#ifndef MH_ABL_NOHEAD
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) {
                float o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = __shfl_xor(part[r], m);
#pragma unroll
                for (int r = 0; r < 16; ++r) part[r] += o[r];
            }
#endif
            float lt = 0.f, dt = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pred = part[r] + b2;
                const bool ok = p0w + mh_row(r, half) < nv;
                const float tg = tgv[r];
                float l = 0.f, g = 0.f;
                if constexpr (LOSS == PASE_LOSS_L1) {
                    const float d = pred - tg;
                    l = fabsf(d);
                    g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                } else if constexpr (LOSS == PASE_LOSS_MSE) {
                    const float d = pred - tg;
                    l = d * d;
                    g = 2.f * d;
                } else if constexpr (LOSS == PASE_LOSS_BCE_LOGITS) {
                    l = fmaxf(pred, 0.f) - pred * tg + log1pf(expf(-fabsf(pred)));
                    g = 1.f / (1.f + expf(-pred)) - tg;
                }
                const float dp = ok ? g * p.grad_scale : 0.f;
                dpred[r] = dp;
                lt += ok ? l : 0.f;
                dt += dp;
                part[r] = pred;
            }
            if (p.pred && l31 == 0) {       // (inference-style callers only: the training step does not ask for it)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (p0w + mh_row(r, half) < nv) (p.pred + ob)[hq + (unsigned)mh_row(r, 0)] = part[r];
            }
            if (l31 == 0) {
                lsum += (double)lt;
                s_db2 += (double)dt;
            }
        }
        // dy1 in place of y1; the per-hidden-row sums; dy1 rows (time step major) into LDS for stages 2 and 3
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y1 = acc1[t][r] + b1v[t];
                const float a1 = y1 > 0.f ? y1 : y1 * a1v[t];
                const float da1 = w2v[t] * dpred[r];
                const float dy1 = y1 > 0.f ? da1 : da1 * a1v[t];
                st_dw2[t] = fmaf(dpred[r], a1, st_dw2[t]);
                st_da1[t] = fmaf(da1, fminf(y1, 0.f), st_da1[t]);
                st_db1[t] += dy1;
                Ds[ds_st + mh_row(r, 0) * MH_PD + 32 * t] = dy1;
            }
        pase_wave_sync();                 // (the rows this wave reads next are the rows it has just written)
        MH_STAMP(2);

        // ---- 2. dA0^T = dY1^T W1 for the same 32 time steps: rows = steps, columns = channels 32 c + l31 -----------------
        f32x16 acc2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[c][r] = 0.f;
        {
            float ac[MH_UB], bc[MH_UB][4];
            auto ld = [&](int blk, float (&aa)[MH_UB], float (&bb)[MH_UB][4]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < MH_UB; ++u) {
                    const int j = blk * MH_UB + u;
                    aa[u] = Ds[ds_own + 2 * j];
#pragma unroll
                    for (int c = 0; c < 4; ++c) bb[u][c] = W1s[w1_col + 2 * j * MH_PW + 32 * c];
                }
            };
            ld(0, ac, bc);
#pragma unroll
            for (int blk = 0; blk < MH_H / 2 / MH_UB; ++blk) {
                float an[MH_UB], bn[MH_UB][4];
                if (blk + 1 < MH_H / 2 / MH_UB) ld(blk + 1, an, bn);
#pragma unroll
                for (int u = 0; u < MH_UB; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc2[c] = MH_MFMA2(ac[u], bc[u][c], acc2[c]);
                if (blk + 1 < MH_H / 2 / MH_UB) {
#pragma unroll
                    for (int u = 0; u < MH_UB; ++u) {
                        ac[u] = an[u];
#pragma unroll
                        for (int c = 0; c < 4; ++c) bc[u][c] = bn[u][c];
                    }
                }
            }
        }
        MH_STAMP(3);
        // whole-tile form (uniform: every step live, rows 16-byte aligned) without range tests or a scalar twin of the store; a
        // channel block's 16 y values are read together in front of its arithmetic (read per register quad they were one LDS
        // latency per quad: the stamps had this epilogue at 13 % of a tile)
        auto epilogue2 = [&](auto fast_tag) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float* dyb = p.dy + ((size_t)s * MH_C + 32 * c) * (size_t)p.T + t0 + p0w;      // (uniform)
                const unsigned lo = (unsigned)l31 * (unsigned)p.T + hq;                     // this lane's row, this half's steps
                const float* yrow = &Ys[ys_ch + 32 * c * MH_PY];
                float yq[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) yq[r] = yrow[8 * (r >> 2) + (r & 3)];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int q = 8 * g + 4 * half;                           // registers 4 g .. 4 g + 3 = steps q .. q + 3
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float yv = yq[4 * g + e];
                        const float dA = acc2[c][4 * g + e];                  // (steps past the end: dpred = 0 -> dA = 0)
                        o[e] = yv > 0.f ? dA : dA * a0v[c];
                        st_dy0[c] += o[e];
                        st_da0[c] = fmaf(dA, fminf(yv, 0.f), st_da0[c]);      // dA * y where y <= 0 (one use of the comparison)
                    }
                    float* dst = dyb + (lo + 8u * (unsigned)g);
#ifdef MH_ABL_NOSTORE
                    if (o[0] == 12345.678f)
#endif
                    if constexpr (FAST) {
                        f32x4 v4;
                        v4[0] = o[0]; v4[1] = o[1]; v4[2] = o[2]; v4[3] = o[3];
                        *reinterpret_cast<f32x4*>(dst) = v4;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (p0w + q + e < nv) dst[e] = o[e];
                    }
                }
                PASE_SCHED_BARRIER();   // (one channel block at a time: hoisted together, the 64 comparisons' lane masks were
                                        //  parked in spilled SGPRs -- 380 v_writelane / v_readlane with their hazard nops)
            }
        };
        if (vec4 && nv == MH_P) epilogue2(std::true_type{});
        else epilogue2(std::false_type{});
        MH_STAMP(4);
        __syncthreads();                                               // every wave's dy1 is in Ds
        MH_STAMP(5);

        // ---- 3. dW1^T += A0 dY1^T over the tile's 128 time steps; this wave's 32 channels ----------------------------
        // The next tile's copy cannot start before this stage has read Ys (single-buffered) and was 13-16 % of a tile, exposed:
        // one dword of every 128-byte line of this wave's rows of the NEXT tile is requested here, so that the copy at the top
        // of the next tile finds its lines in L2.  (Plain loads the compiler can see; their values are only "used" by the empty
        // asm behind the next tile's wait.)
#ifndef MH_ABL_NOPF
        if (tile + gridDim.x < ntiles) {
            const long nt = tile + gridDim.x;
            const int s2 = (int)((unsigned)nt / (unsigned)tiles_per_seq);
            const int t2 = ((int)nt - s2 * tiles_per_seq) * MH_P;
            const float* yb2 = p.y + ((size_t)s2 * MH_C + 32 * wave) * (size_t)p.T + t2;      // (uniform)
            const unsigned col = 32u * ((unsigned)lane & 3u);
            if ((int)col < p.T - t2) {
                pf0 = yb2[((unsigned)lane >> 2) * (unsigned)p.T + col];
                pf1 = yb2[(16u + ((unsigned)lane >> 2)) * (unsigned)p.T + col];
            }
        }
#endif
        {
            float yc[MH_UB], bc[MH_UB][2];
            auto ld = [&](int blk, float (&yy)[MH_UB], float (&bb)[MH_UB][2]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < MH_UB; ++u) {
                    const int j = blk * MH_UB + u;
                    yy[u] = Ys[ys_row + 2 * j];
#pragma unroll
                    for (int t = 0; t < 2; ++t) bb[u][t] = Ds[ds_col + 2 * j * MH_PD + 32 * t];
                }
            };
            ld(0, yc, bc);
#pragma unroll
            for (int blk = 0; blk < MH_P / 2 / MH_UB; ++blk) {
                float yn[MH_UB], bn[MH_UB][2];
                if (blk + 1 < MH_P / 2 / MH_UB) ld(blk + 1, yn, bn);
#pragma unroll
                for (int u = 0; u < MH_UB; ++u) {
                    const float neg = yc[u] * a0_mine;
                    const float a = yc[u] > 0.f ? yc[u] : neg;
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc3[t] = MH_MFMA3(a, bc[u][t], acc3[t]);
                }
                if (blk + 1 < MH_P / 2 / MH_UB) {
#pragma unroll
                    for (int u = 0; u < MH_UB; ++u) {
                        yc[u] = yn[u];
                        bc[u][0] = bn[u][0];
                        bc[u][1] = bn[u][1];
                    }
                }
            }
        }
        MH_STAMP(6);
        __syncthreads();                                               // the next tile overwrites Ys / Ds
        MH_STAMP(7);
    }

    MH_TRACE_END();
    const int l31 = l31_k, half = half_k;
    // ---- flush: dW1 (+=), the per-row sums (doubles, += : PaseActBwd::sums / pase_head1_bwd layouts), loss --------------
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 32 * wave + mh_row(r, half), h = 32 * t + l31;
            atomicAdd(p.dw1 + (size_t)h * MH_C + ch, acc3[t][r]);
        }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int h = 32 * t + l31;
        const float v0 = st_dw2[t] + __shfl_xor(st_dw2[t], 32), v1 = st_da1[t] + __shfl_xor(st_da1[t], 32);
        const float v2 = st_db1[t] + __shfl_xor(st_db1[t], 32);
        if (half == 0) {
            atomicAdd(p.sums1 + (size_t)h * 3 + 0, (double)v0);
            atomicAdd(p.sums1 + (size_t)h * 3 + 1, (double)v1);
            atomicAdd(p.sums1 + (size_t)h * 3 + 2, (double)v2);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = 32 * c + l31;
        const float v0 = st_dy0[c] + __shfl_xor(st_dy0[c], 32), v2 = st_da0[c] + __shfl_xor(st_da0[c], 32);
        if (half == 0) {
            atomicAdd(p.sums0 + (size_t)ch * 3 + 0, (double)v0);
            atomicAdd(p.sums0 + (size_t)ch * 3 + 2, (double)v2);
        }
    }
    lsum = pase_wave_sum64d(lsum);
    s_db2 = pase_wave_sum64d(s_db2);
    if (lane == 0) {
        if (p.loss_acc) atomicAdd(p.loss_acc, lsum);
        atomicAdd(p.sums1 + (size_t)MH_H * 3, s_db2);
    }
}

}  // namespace

#ifdef MH_TRACE
extern "C" int pase_mlp_head1_trace_read(unsigned long long* host) {
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mh_trace), sizeof(unsigned long long) * 32);
}
#endif

extern "C" int pase_mlp_head1_supported(const PaseMlpHead1* d) {
    return (d->C == MH_C && d->H == MH_H && d->S > 0 && d->T > 0) ? 1 : 0;
}

extern "C" int pase_mlp_head1_step(const PaseMlpHead1* d, void* stream) {
    const PaseMlpHead1 p = *d;
    if (p.S <= 0 || p.T <= 0) return 0;
    if (!pase_mlp_head1_supported(d)) return -11;
    if ((long)p.C * p.T >= (1L << 29)) return -8;            // 32-bit byte offsets inside one sequence
    if ((long)p.S * ((p.T + MH_P - 1) / MH_P) >= 0x7fffffffL) return -8;      // 32-bit tile index
    if (!p.y || !p.w1 || !p.w2 || !p.dy || !p.dw1 || !p.sums0 || !p.sums1) return -2;
    if (p.loss_type != PASE_LOSS_NONE && (!p.target || !p.loss_acc)) return -2;
    const int tps = (p.T + MH_P - 1) / MH_P;
    const long ntiles = (long)p.S * tps;
    long nwg = p.max_wg > 0 ? p.max_wg : 256;
    if (nwg > ntiles) nwg = ntiles;
    switch (p.loss_type) {
        case PASE_LOSS_NONE: PASE_LAUNCH(mlp_head1_kernel<PASE_LOSS_NONE>, dim3((unsigned)nwg), dim3(MH_NT), (hipStream_t)stream, p, tps, ntiles); break;
        case PASE_LOSS_L1: PASE_LAUNCH(mlp_head1_kernel<PASE_LOSS_L1>, dim3((unsigned)nwg), dim3(MH_NT), (hipStream_t)stream, p, tps, ntiles); break;
        case PASE_LOSS_MSE: PASE_LAUNCH(mlp_head1_kernel<PASE_LOSS_MSE>, dim3((unsigned)nwg), dim3(MH_NT), (hipStream_t)stream, p, tps, ntiles); break;
        case PASE_LOSS_BCE_LOGITS: PASE_LAUNCH(mlp_head1_kernel<PASE_LOSS_BCE_LOGITS>, dim3((unsigned)nwg), dim3(MH_NT), (hipStream_t)stream, p, tps, ntiles); break;
        default: return -2;
    }
    PASE_CHECK_LAUNCH();
    return 0;
}

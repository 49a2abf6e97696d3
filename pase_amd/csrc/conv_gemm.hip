// conv_gemm.hip -- the one MFMA kernel family behind every dense contraction of the PASE+ step.
//
//   Y[s, row, pos] = sum_{ci,kk} A[row, (ci,kk)] * act(bn(X[s, ci, q*stride + kk*tapstep - padL]))
//
// i.e. an implicit-GEMM 1-D convolution over NCT fp32 tensors (reference layout: (batch, channels,
// time), pase/models/modules.py:1058-1077 FeBlock.forward, :527-556 MLPBlock, :558-589
// GDeconv1DBlock, frontend.py:182,195 1x1 convs, third-party torchqrnn Linear) with
//   * the previous layer's BatchNorm affine + PReLU applied ON LOAD (activations are stored once,
//     raw, and never re-written normalised),
//   * reflect / zero padding resolved in the loader (modules.py:1061-1071: asymmetric reflect pad),
//   * a store map that is either plain (Conv1d) or a pixel-shuffle (ConvTranspose1d and every
//     strided dgrad: rows = (phase, channel), pos = q*ps + phase + poff),
//   * optional per-output-channel partial sums (sum, sum of squares) for the following BatchNorm
//     (training-mode batch statistics, modules.py:79 / frontend.py:208), deterministic (no atomics),
//   * optional fused r-context MSE epilogue (pase/losses.py:6-37 ContextualizedLoss): the
//     (B, D*r, F) prediction/target pair of a regression worker is never materialised.
//
// gfx950 mapping: 256 threads = 4 waves; block tile BM x BN x 16; each wave owns a 64x64 sub-tile as
// 2x2 v_mfma_f32_32x32x2_f32 tiles (exact fp32, 64 accumulator VGPRs).  A/B K-tiles are gathered
// global -> registers -> LDS (double-buffered, one barrier per K-tile); fp32 MFMA needs only one
// operand dword per lane per 64-cycle instruction, so the im2col gather + ds_read_b32 fragment reads
// stay far below the LDS / L1 limits and the kernel is bound by the MFMA pipe.
// Two shapes: <128,128> (waves 2x2) and <64,256> (waves 1x4) for the 64-row layers.
#include "hip_compat.h"
#include "pase_amd.h"

namespace {

constexpr int BK = 16;
constexpr int NTHREADS = 256;

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    // bijective XCD-aware remap (cdna_hip_programming.md T1): blocks that run on one XCD (bid % 8)
    // get a contiguous range of tile ids, so row-tiles sharing a B panel hit the same L2.
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int BM, int BN>
__global__ void __launch_bounds__(NTHREADS) conv_gemm_kernel(PaseConvGemm p) {
    constexpr int WAVES_M = BM / 64;
    constexpr int WAVES_N = BN / 64;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    constexpr int A_PER_T = BM * BK / NTHREADS;  // 8 or 4
    constexpr int B_PER_T = BN * BK / NTHREADS;  // 8 or 16
    constexpr int B_KSTEP = NTHREADS / BN;       // 2 or 1
    constexpr int LDA = BM + 1;                  // +1: conflict-free transposed ds_write
    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][BN];
    __shared__ float red[WAVES_N][BM][2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    const int ntot = p.S * p.Ncols;
    const int n_row_tiles = (p.M + BM - 1) / BM;
    const int n_col_tiles = (ntot + BN - 1) / BN;
    const int tile = xcd_swizzle(blockIdx.x, n_row_tiles * n_col_tiles);
    const int mt = tile % n_row_tiles;
    const int nt = tile / n_row_tiles;
    const int m0 = mt * BM;
    const int n0 = nt * BN;

    // ---- per-thread loader state -------------------------------------------------------------
    // B (im2col gather): fixed column j, k-rows kr0 + i*B_KSTEP
    const int bj = tid % BN;
    const int bkr0 = tid / BN;
    const int bn_ = n0 + bj;
    const bool bcol_ok = bn_ < ntot;
    const int bs = bcol_ok ? bn_ / p.Ncols : 0;
    const int bq = bcol_ok ? bn_ % p.Ncols : 0;
    const int ubase = bq * p.stride - p.padL;
    const float* xcol = p.x + ((size_t)bs * p.x_ctot + p.x_coff) * (size_t)p.Tin;
    // A: k-col = tid % BK, rows tid / BK + i * (NTHREADS / BK)
    const int akc = tid % BK;
    const int ar0 = tid / BK;

    float areg[A_PER_T], breg[B_PER_T];

    auto load_tile = [&](int k0) {
        // ---- A tile: W[m][k], K contiguous
        {
            const int kf = k0 + akc;
            const bool kok = kf < p.K;
#pragma unroll
            for (int i = 0; i < A_PER_T; ++i) {
                const int m = m0 + ar0 + i * (NTHREADS / BK);
                areg[i] = (kok && m < p.M) ? p.w[(size_t)m * p.ldw + kf] : 0.f;
            }
        }
        // ---- B tile: gather with padding + on-load transform
        {
            int kf = k0 + bkr0;
            int ci, kk;
            if (p.tap_major) { kk = kf / p.Cin; ci = kf - kk * p.Cin; }
            else             { ci = kf / p.taps; kk = kf - ci * p.taps; }
#pragma unroll
            for (int i = 0; i < B_PER_T; ++i) {
                float v = 0.f;
                if (bcol_ok && kf < p.K) {
                    int u = ubase + kk * p.tapstep;
                    if (p.pad_mode == PASE_PAD_REFLECT) {
                        if (u < 0) u = -u;
                        if (u >= p.Tin) u = 2 * (p.Tin - 1) - u;
                    }
                    if (u >= 0 && u < p.Tin) {
                        v = xcol[(size_t)ci * p.Tin + u];
                        if (p.in_scale) v = v * p.in_scale[ci] + p.in_shift[ci];
                        if (p.in_alpha) v = v > 0.f ? v : v * p.in_alpha[ci];
                    }
                }
                breg[i] = v;
                // advance (ci,kk) by B_KSTEP flat k positions
                kf += B_KSTEP;
                if (p.tap_major) { ci += B_KSTEP; while (ci >= p.Cin) { ci -= p.Cin; ++kk; } }
                else             { kk += B_KSTEP; while (kk >= p.taps) { kk -= p.taps; ++ci; } }
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) As[buf][akc][ar0 + i * (NTHREADS / BK)] = areg[i];
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) Bs[buf][bkr0 + i * B_KSTEP][bj] = breg[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int fr = lane & 31;   // fragment row/col within a 32-wide MFMA tile
    const int fk = lane >> 5;   // k within the K=2 step
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);   // global loads in flight under the MFMAs
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int kb = ks * 2 + fk;
            const float a0 = As[cur][kb][wm * 64 + fr];
            const float a1 = As[cur][kb][wm * 64 + 32 + fr];
            const float b0 = Bs[cur][kb][wn * 64 + fr];
            const float b1 = Bs[cur][kb][wn * 64 + 32 + fr];
            acc[0][0] = pase_mfma_32x32x2(a0, b0, acc[0][0]);
            acc[0][1] = pase_mfma_32x32x2(a0, b1, acc[0][1]);
            acc[1][0] = pase_mfma_32x32x2(a1, b0, acc[1][0]);
            acc[1][1] = pase_mfma_32x32x2(a1, b1, acc[1][1]);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------
    // D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int rbase = 4 * (lane >> 5);
    int cs[2], cq[2];
    bool cok[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn * 64 + b * 32 + fr;
        cok[b] = n < ntot;
        cs[b] = cok[b] ? n / p.Ncols : 0;
        cq[b] = cok[b] ? n % p.Ncols : 0;
    }

    if (p.epilogue == PASE_EPI_STORE) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + rbase;  // row in tile
                const int m = m0 + ml;
                const bool mok = m < p.M;
                const int ph = mok ? m / p.Cout_store : 0;
                const int co = mok ? m - ph * p.Cout_store : 0;
                const float bv = (mok && p.bias) ? p.bias[co] : 0.f;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float v = acc[a][b][r] + bv;
                    const int pos = cq[b] * p.ps + ph + p.poff;
                    const bool ok = mok && cok[b] && pos >= 0 && pos < p.Tout;
                    if (ok) {
                        p.y[((size_t)cs[b] * p.y_ctot + p.y_coff + co) * (size_t)p.Tout + pos] = v;
                        s1 += v;
                        s2 += v * v;
                    }
                }
                if (p.stat_part) {   // uniform branch
                    s1 = pase_wave_sum32(s1);
                    s2 = pase_wave_sum32(s2);
                    if (fr == 0) { red[wn][ml][0] = s1; red[wn][ml][1] = s2; }
                }
            }
        }
        if (p.stat_part) {
            __syncthreads();
            // one partial (sum, sumsq) per (column tile, output row); rows are channels here
            for (int ml = tid; ml < BM; ml += NTHREADS) {
                const int m = m0 + ml;
                if (m < p.M) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int w = 0; w < WAVES_N; ++w) { s1 += red[w][ml][0]; s2 += red[w][ml][1]; }
                    float* dst = p.stat_part + ((size_t)nt * p.M + m) * 2;
                    dst[0] = s1;
                    dst[1] = s2;
                }
            }
        }
    } else {  // PASE_EPI_MSE_CTX: rows m = d*r + j, columns (b, t); target = label[b, d, t + j - r/2]
        float lsum = 0.f;
        const int half = p.r_ctx / 2;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                const bool mok = m < p.M;
                const int d = mok ? m / p.r_ctx : 0;
                const int j = mok ? m - d * p.r_ctx : 0;
                const float bv = (mok && p.bias) ? p.bias[m] : 0.f;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (mok && cok[b]) {
                        const float pred = acc[a][b][r] + bv;
                        const int tt = cq[b] + j - half;
                        float tgt = 0.f;
                        if (tt >= 0 && tt < p.Ncols)
                            tgt = p.label[((size_t)cs[b] * p.label_D + d) * (size_t)p.Ncols + tt];
                        const float diff = pred - tgt;
                        lsum += diff * diff;
                        const size_t o = ((size_t)cs[b] * p.M + m) * (size_t)p.Ncols + cq[b];
                        if (p.y) p.y[o] = pred;
                        if (p.grad_out) p.grad_out[o] = diff * p.grad_scale;
                    }
                }
            }
        }
        lsum = pase_wave_sum64(lsum);
        if (lane == 0) red[0][wave][0] = lsum;
        __syncthreads();
        if (tid == 0) {
            const double t = (double)red[0][0][0] + (double)red[0][1][0] + (double)red[0][2][0] + (double)red[0][3][0];
            atomicAdd(p.loss_acc, t);
        }
    }
}

}  // namespace

extern "C" int pase_conv_gemm(const PaseConvGemm* d, void* stream) {
    const PaseConvGemm p = *d;
    if (p.M <= 0 || p.K <= 0 || p.S <= 0 || p.Ncols <= 0) return 0;
    if (p.epilogue == PASE_EPI_MSE_CTX && (!p.label || !p.loss_acc || p.r_ctx < 1)) return -2;
    if (p.pad_mode == PASE_PAD_REFLECT && (p.padL >= p.Tin)) return -3;
    const long ntot = (long)p.S * p.Ncols;
    hipStream_t st = (hipStream_t)stream;
    const bool narrow = (p.tile_hint == 64) || (p.tile_hint == 0 && p.M <= 64);
    if (narrow) {
        const long tiles = ((p.M + 63) / 64) * ((ntot + 255) / 256);
        PASE_LAUNCH((conv_gemm_kernel<64, 256>), dim3((unsigned)tiles), dim3(NTHREADS), st, p);
    } else {
        const long tiles = ((p.M + 127) / 128) * ((ntot + 127) / 128);
        PASE_LAUNCH((conv_gemm_kernel<128, 128>), dim3((unsigned)tiles), dim3(NTHREADS), st, p);
    }
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_conv_gemm_stat_tiles(int M, int S, int Ncols, int tile_hint) {
    const long ntot = (long)S * Ncols;
    const bool narrow = (tile_hint == 64) || (tile_hint == 0 && M <= 64);
    return (int)(narrow ? (ntot + 255) / 256 : (ntot + 127) / 128);
}

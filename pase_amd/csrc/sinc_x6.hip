// sinc_x6.hip -- the SincNet layer (ONE input channel, up to 256 taps, at most 64 filters) on the bf16 matrix pipe:
//
//   forward          y[s, m, q]  = sum_kk filt[m, kk] * x[s, q + kk - padL]                       (SincConv_fast's F.conv1d,
//                                                                                                 pase/models/modules.py:932)
//   weight gradient  dfilt[m, kk] += sum_{s,q} g[s, m, q] * x[s, q + kk - padL]                   (autograd's conv1d wgrad)
//
// Same arithmetic as conv_x6c.hip: every fp32 operand is the sum of three round-to-nearest bf16 pieces, a product is
// hh + (mm + hl + lh + hm + mh) on v_mfma_f32_32x32x16_bf16 with the hh sum and the small terms in separate accumulators.
//
// Why its own kernels.  With one input channel the contraction index of the GEMM is the TAP, and an MFMA operand fragment
// (eight consecutive k of one column) is eight consecutive SAMPLES x[u .. u + 7]: a sliding window.  The LDS image is the
// "window image"  W[plane][w] = 16-byte chunk of the pieces of x[u0 + w .. u0 + w + 7]:  the fragment of (column c, tap
// group g, half fk) is window c + 16 g + 8 fk -- one aligned, conflict-free ds_read_b128, consecutive lanes = consecutive
// chunks -- and ONE staging pass (504 windows from 511 samples, each converted a handful of times) serves all 16 k-groups of
// a 64 x 256 tile.  conv_x6c.hip's channel-minor image cannot express that (its k axis is channels), and its 128-row tile
// would spend half of every MFMA on zero rows: the round-3 build ran both launches on the exact-fp32 pipe (98 / 107 TFLOP/s).
//   Tile 64 x 256, 256 threads = 4 waves as 2 (rows) x 2 (column halves), wave tile 32 x 128 with two accumulators per B
//   tile (128 VGPRs); two workgroups per CU so that one's staging / epilogue runs under the other's MFMAs.
//   Weight gradient: BOTH operands are staged (g is converted on the fly: each element exactly once, the 64 x 256 tile is the
//   whole problem), contraction over positions in stages of 64, split over the grid, fp32 atomics into dfilt.
#include <type_traits>

#include "sinc_x6.h"
#include "act_bwd.h"

namespace {

constexpr int SX_NT = 256;
constexpr int SX_BN = 256;                    // columns (forward: positions; weight gradient: taps) per tile
constexpr int SX_KGMAX = 16;                  // forward: k-groups of 16 taps
constexpr int SX_NW = SX_BN + 16 * SX_KGMAX - 8;      // 504 windows of a forward tile
constexpr int SW_POS = 64;                    // weight gradient: positions per stage (4 k-groups)
constexpr int SW_NW = SX_BN + SW_POS;         // 320 windows of a weight-gradient stage
constexpr int SW_ACH = (SW_POS / 8) * 64;     // chunks per plane of the g image: [8 position octets][64 rows]
constexpr int SW_RP = SW_POS + 4;             // on-load form: floats per row of the raw y / dA tiles (16-byte aligned rows, a
                                              // bank shift of four per row for the 16-byte reads of 16 rows x 4 parts)

// one dword per lane straight from global memory into LDS (global_load_lds_dword: destination = wave-uniform LDS address +
// 4 * lane for the active lanes; no VGPR holds the data while it is in flight)
#ifdef PASE_HIPEMU
__device__ __forceinline__ void sx_load_lds4(const float* src, float* lds_wave_base, int lane) { lds_wave_base[lane] = *src; }
#else
__device__ __forceinline__ void sx_load_lds4(const float* src, float* lds_wave_base, int) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
#endif
// the wave's own LDS DMA has landed (each wave converts exactly the rows it copied: no barrier).  The compiler's own
// bookkeeping would put this wait in front of the first read of the raw tiles as well; it is spelled out because the CPU
// emulator's lanes are fibers that only meet at wave-level exchanges: there it is the point where they do.
#ifdef PASE_HIPEMU
__device__ __forceinline__ void sx_dma_wait() { hipemu::sync_wave(); }
#else
__device__ __forceinline__ void sx_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

__device__ __forceinline__ int sx_reflect(int u, int T) {
    u = max(u, -u);
    return min(u, 2 * (T - 1) - u);
}

// ================================================================================================================
// forward
// ================================================================================================================
__global__ void __launch_bounds__(SX_NT, 2) sinc_x6_fwd_kernel(PaseConvGemm p, PaseSincPlan pl) {
    constexpr int RED_CHUNKS = (int)(sizeof(float) * 2 * 64 * 2 / 16);
    __shared__ __attribute__((aligned(16))) u32x4 Ws[3 * SX_NW + RED_CHUNKS];
    float (*red)[64][2] = reinterpret_cast<float (*)[64][2]>(&Ws[3 * SX_NW]);

    const int tid = threadIdx.x, lane = tid & 63, wave = pase_uniform(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 31, fk = lane >> 5;
    // persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ... (two workgroups per CU; launched one tile per workgroup,
    // 12 000 workgroups of ~18 us, a third of the kernel's time was between workgroups -- in-kernel stamps against its duration)
    const int ntiles = p.S * pl.tiles_per_seq;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int s = tile / pl.tiles_per_seq;
    const int q0 = (tile - s * pl.tiles_per_seq) * SX_BN;
    const float* xrow = p.x + ((size_t)s * p.x_ctot + p.x_coff) * (size_t)p.Tin;

    // ---- window image: thread -> windows 2 t, 2 t + 1 from the nine samples starting at q0 - padL + 2 t --------------
    for (int t = tid; 2 * t < pl.nwin; t += SX_NT) {
        float v[9];
        const int u0 = q0 - p.padL + 2 * t;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            int u = u0 + e;
            if (p.pad_mode == PASE_PAD_REFLECT) u = sx_reflect(u, p.Tin);
            v[e] = (u >= 0 && u < p.Tin) ? xrow[u] : 0.f;
        }
        u32x4 w0[3], w1[3];
        pase_split_two_windows(v, w0, w1);
#pragma unroll
        for (int pz = 0; pz < 3; ++pz) {
            Ws[pz * SX_NW + 2 * t] = w0[pz];
            if (2 * t + 1 < SX_NW) Ws[pz * SX_NW + 2 * t + 1] = w1[pz];
        }
    }

    // ---- filters: fragment-ordered pack [32-row tile][k-group][plane][lane], prefetched two k-groups ahead --------------
    const u32x4* ap = reinterpret_cast<const u32x4*>(p.wx6) + (size_t)wm * pl.n_kg * 192 + lane;
    auto load_a = [&](u32x4 (&a)[3], int g) __attribute__((always_inline)) {
        const int gg = min(g, pl.n_kg - 1);
#pragma unroll
        for (int pz = 0; pz < 3; ++pz) a[pz] = ap[(size_t)gg * 192 + pz * 64];
    };
    f32x16 accH[4], accS[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accH[j][r] = 0.f;
            accS[j][r] = 0.f;
        }
    u32x4 a0[3], a1[3], a2[3];
    load_a(a0, 0);
    load_a(a1, 1);
    __syncthreads();

    const int bcol = wn * 128 + fr + 8 * fk;
    auto mfma_step = [&](const u32x4 (&a)[3], int g) __attribute__((always_inline)) {
        constexpr int PZA[5] = {1, 0, 2, 0, 1}, PZB[5] = {1, 2, 0, 1, 0};       // mm, hl, lh, hm, mh -> accS; hh -> accH
        const u32x4* xb = &Ws[bcol + 16 * g];
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
            u32x4 b0[3], b1[3];
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) {
                b0[pz] = xb[pz * SX_NW + 32 * jp];
                b1[pz] = xb[pz * SX_NW + 32 * (jp + 1)];
            }
#pragma unroll
            for (int pi = 0; pi < 5; ++pi) {
                accS[jp] = pase_mfma_bf16_32x32x16(a[PZA[pi]], b0[PZB[pi]], accS[jp]);
                accS[jp + 1] = pase_mfma_bf16_32x32x16(a[PZA[pi]], b1[PZB[pi]], accS[jp + 1]);
            }
            accH[jp] = pase_mfma_bf16_32x32x16(a[0], b0[0], accH[jp]);
            accH[jp + 1] = pase_mfma_bf16_32x32x16(a[0], b1[0], accH[jp + 1]);
        }
    };
    for (int g = 0; g < pl.n_kg; g += 3) {
        load_a(a2, g + 2);
        mfma_step(a0, g);
        if (g + 1 < pl.n_kg) {
            load_a(a0, g + 3);
            mfma_step(a1, g + 1);
        }
        if (g + 2 < pl.n_kg) {
            load_a(a1, g + 4);
            mfma_step(a2, g + 2);
        }
    }

    // ---- epilogue: store, BatchNorm partial sums per (tile, row) ------------------------------------------------------
    // D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    // Round 5 (in-kernel stamps: this epilogue was 58 % of a workgroup's time, 27-31 k clocks against 16-21 k for the 384-MFMA
    // loop): the bias load sat inside the row loop behind a branch, so every use of its register -- one per store, each in a
    // block of its own behind the range tests -- carried s_waitcnt vmcnt(0), which, the counter being in order, also waited for
    // the previous STORE: 64 stores, one completion latency each (whether or not the layer has a bias).  Now the 16 bias
    // values are loaded in front of the first store, whole tiles take a path without range tests (one basic block), and an
    // address is a wave-uniform base plus one 32-bit lane offset.
    const int rbase = wm * 32 + 4 * fk;
    float bvs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = rbase + (r & 3) + 8 * (r >> 2);
        bvs[r] = (p.bias && m < p.M) ? p.bias[m] : 0.f;
    }
    float* ybase = p.y + ((size_t)s * p.y_ctot + p.y_coff + wm * 32) * (size_t)p.Tout + q0 + wn * 128;      // (uniform)
    const unsigned lane_off = (unsigned)(4 * fk) * (unsigned)p.Tout + (unsigned)fr;
    const bool whole = p.M == 64 && q0 + SX_BN <= p.Ncols;                                                  // (uniform)
    auto store_rows = [&](auto whole_tag) __attribute__((always_inline)) {
        constexpr bool WHOLE = decltype(whole_tag)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = (r & 3) + 8 * (r >> 2);                           // row inside the wave's 32 (plus 4 fk: lane_off)
            const int m = rbase + mr;
            const bool mok = WHOLE || m < p.M;
            float* yrow = ybase + (size_t)mr * p.Tout;                       // (uniform)
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = accH[j][r] + accS[j][r] + bvs[r];
                if (WHOLE || (mok && q0 + wn * 128 + j * 32 + fr < p.Ncols)) {
                    yrow[lane_off + 32u * (unsigned)j] = v;
                    s1 += v;
                    s2 += v * v;
                }
            }
            if (p.stat_part) {      // uniform
                s1 = pase_half_sum_lane31(s1);
                s2 = pase_half_sum_lane31(s2);
                if (fr == 31) {
                    red[wn][m][0] = s1;
                    red[wn][m][1] = s2;
                }
            }
        }
    };
    if (whole) store_rows(std::true_type{});
    else store_rows(std::false_type{});
    if (p.stat_part) {
        __syncthreads();
        if (tid < 64 && tid < p.M) {
            float* dst = p.stat_part + ((size_t)tile * p.M + tid) * 2;
            dst[0] = red[0][tid][0] + red[1][tid][0];
            dst[1] = red[0][tid][1] + red[1][tid][1];
        }
    }
    __syncthreads();      // the next tile's window image / partial sums overwrite this one's
  }
}

// filt (K-major fp32 pack wt[kk * ldwt + m]) -> fragment-ordered bf16 planes [32-row tile (2)][k-group][plane][lane]:
// lane = (fk, row): element e = tap 16 g + 8 fk + e of filter 32 rt + row; zero past the taps / past M
// (wt == NULL: straight from the filters as the reference stores them, w[m * ldw + kk])
__global__ void sinc_x6_pack_kernel(const float* __restrict__ wt, u32x4* __restrict__ out, int M, int ldwt, int taps, int n_kg,
                                    const float* __restrict__ w, int ldw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * n_kg * 64) return;
    const int lane = idx & 63, g = (idx >> 6) % n_kg, rt = (idx >> 6) / n_kg;
    const int fk = lane >> 5, m = rt * 32 + (lane & 31);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kk = 16 * g + 8 * fk + e;
        v[e] = (kk < taps && m < M) ? (wt ? wt[(size_t)kk * ldwt + m] : w[(size_t)m * ldw + kk]) : 0.f;
    }
    u32x4 o[3];
    pase_split_bf16x3_rne(v, o);
#pragma unroll
    for (int pz = 0; pz < 3; ++pz) out[((size_t)(rt * n_kg + g) * 3 + pz) * 64 + lane] = o[pz];
}

// ================================================================================================================
// weight gradient
// ================================================================================================================
// AB: the gradient operand is not read from p.g but evaluated while it is staged as the APPLY pass of the BatchNorm + PReLU
// backward `ab` (act_bwd.h: the same element functions as norm_act.hip's act_bwd_apply_kernel; ab.sums complete, i.e. the
// reduce pass enqueued earlier on this stream).  The SincNet layer's dy (786 MB at bs32) has no other consumer -- the first
// layer needs no data gradient -- so the apply pass over it (read y, read dA, write dy) and this kernel's read of dy become
// one read of y and dA.
//   The plain form keeps the next stage's 16 gradient values per thread in registers under this stage's MFMAs (the kernel
//   sits at its 256-register budget: two workgroups per CU).  Twice that (y and dA) does not fit -- a register build of this
//   form spilled 80 VGPRs and ran 2.8 ms against 1.2 ms for apply pass + plain launch -- so the raw 64 x 64 tiles of y and of
//   the padded data gradient travel by LDS DMA (one 256-byte row per instruction, each wave the 16 rows it converts itself:
//   no barrier between the copy and its use, only the wave's own vmcnt) and are turned into dy, split and written to the
//   g image at the top of the next stage.  The pooled dense-skip gradient (one or two frames per run of 16 positions) is
//   prefetched into two registers; the reflect-padding fold touches the first and last stage of a sequence only and is
//   read there directly.
//   HB: ab.has_bn as a compile-time constant (the element function's norm switch sat inside the unrolled conversion: with the
//   per-element range tests, 127 exec-mask regions per stage); whole stages (64 live rows, 64 live positions) take a path
//   without range tests.
template <bool AB, int HB>
__global__ void __launch_bounds__(SX_NT, 2) sinc_x6_wgrad_kernel(PaseWgrad p, PaseSincPlan pl, PaseActBwd ab) {
    // one LDS image per stage: g [plane][position octet c (8)][row (64)] and the window image of x [plane][320]
    __shared__ __attribute__((aligned(16))) u32x4 Ws[3 * SW_ACH + 3 * SW_NW];
    // (objects of their own: the compiler tracks pending LDS DMA and would wait for it in front of every read of Ws otherwise)
    __shared__ __attribute__((aligned(16))) float Yr[AB ? 64 * SW_RP : 4];
    __shared__ __attribute__((aligned(16))) float Dr[AB ? 64 * SW_RP : 4];
    __shared__ __attribute__((aligned(16))) float Xr[AB ? SW_NW + 8 + 56 : 4];      // raw samples of the window image (328, in 64s)
    u32x4* As = Ws;
    u32x4* Bs = Ws + 3 * SW_ACH;

    const int tid = threadIdx.x, lane = tid & 63, wave = pase_uniform(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 31, fk = lane >> 5;
    // this workgroup's stages: a contiguous range of the flattened (sequence, 64-position stage) index
    const long total = (long)p.S * pl.tiles_per_seq;
    const long per = (total + gridDim.x - 1) / gridDim.x;
    const long st_begin = (long)blockIdx.x * per;
    const long st_end = st_begin + per < total ? st_begin + per : total;
    if (st_begin >= st_end) return;

    // staging roles: g -- thread -> (row = tid >> 2, 16 positions 16 (tid & 3) ..);  x -- thread -> windows 2 tid, 2 tid + 1
    const int grow = tid >> 2, gpart = tid & 3;
    const bool grow_ok = grow < p.M;
    const float g_al = (!AB && p.g_alpha && grow_ok) ? p.g_alpha[grow] : 1.f;
    ActBwdRow rc = {};
    unsigned pmagic = 0u;
    if (AB) {
        rc = act_bwd_row(ab, grow_ok ? grow : 0);
        pmagic = act_pool_magic(ab);
    }
    auto frame_of = [&](int t) __attribute__((always_inline)) {
        return pmagic ? (int)(((unsigned long long)(unsigned)t * pmagic) >> 32) : (ab.pool_d > 1 ? t / ab.pool_d : t);
    };
    float gv[AB ? 1 : 16], xv[AB ? 1 : 9];
    float dp0 = 0.f, dp1 = 0.f;       // AB: pooled-branch gradient of the first / last frame of the staged run
    int f_first = 0;
    int s_ld = 0, q0_ld = 0;          // AB: (sequence, first position) of the stage whose raw tiles are in flight / in LDS
    auto load_stage = [&](long st) __attribute__((always_inline)) {
        const int s = (int)(st / pl.tiles_per_seq);
        const int q0 = (int)(st - (long)s * pl.tiles_per_seq) * SW_POS;
        if constexpr (AB) {
            s_ld = s;
            q0_ld = q0;
            const int nq = p.Ncols - q0;                                   // live positions of the stage (>= 1)
#pragma unroll 4
            for (int rr = 0; rr < 16; ++rr) {
                const int row = wave * 16 + rr;                            // (wave-uniform)
                if (row >= p.M) break;
                const float* yrw = ab.y + ((size_t)s * ab.y_ctot + ab.y_coff + row) * (size_t)ab.T + q0;
                if (lane < nq) sx_load_lds4(yrw + lane, &Yr[row * SW_RP], lane);
                if (ab.dsrc) {
                    const float* drw = ab.dsrc + ((size_t)s * ab.dsrc_ctot + ab.dsrc_coff + row) * (size_t)ab.Tp + ab.padL + q0;
                    if (lane < nq) sx_load_lds4(drw + lane, &Dr[row * SW_RP], lane);
                }
            }
            if (ab.dpool && grow_ok) {
                const int qa = q0 + 16 * gpart;
                const float* prow = ab.dpool + ((size_t)s * ab.dpool_ctot + ab.dpool_coff + grow) * (size_t)ab.pool_F;
                f_first = frame_of(qa);
                const int f_last = frame_of(min(qa + 15, ab.T - 1));
                dp0 = f_first < ab.pool_F ? prow[f_first] : 0.f;          // (times 1 / d where it is used: no wait here)
                dp1 = f_last < ab.pool_F ? prow[f_last] : 0.f;
            }
            // raw samples x[q0 - padL + i], i < 328 (windows 0 .. 319 need eight more), chunks of 64 shared out over the waves
            const float* xrow = p.z + ((size_t)s * p.z_ctot + p.z_coff) * (size_t)p.Tz;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ch = wave + 4 * c;                                // (wave-uniform)
                if (64 * ch >= SW_NW + 8) break;
                int u = q0 - p.padL + 64 * ch + lane;
                if (p.pad_mode == PASE_PAD_REFLECT) u = sx_reflect(u, p.Tz);
                if (64 * ch + lane < SW_NW + 8 && u >= 0 && u < p.Tz) sx_load_lds4(xrow + u, &Xr[64 * ch], lane);
            }
        } else {
            const float* grw = p.g + ((size_t)s * p.g_ctot + p.g_coff + (grow_ok ? grow : 0)) * (size_t)p.Tg;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int q = q0 + 16 * gpart + e;
                gv[e] = (grow_ok && q < p.Ncols) ? grw[q] : 0.f;
            }
            if (2 * tid < SW_NW) {
                const float* xrow = p.z + ((size_t)s * p.z_ctot + p.z_coff) * (size_t)p.Tz;
                const int u0 = q0 - p.padL + 2 * tid;
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    int u = u0 + e;
                    if (p.pad_mode == PASE_PAD_REFLECT) u = sx_reflect(u, p.Tz);
                    xv[e] = (u >= 0 && u < p.Tz) ? xrow[u] : 0.f;
                }
            }
        }
    };
    auto store_stage = [&]() __attribute__((always_inline)) {
        const int padR = ab.Tp - ab.T - ab.padL;
        // does this stage touch the steps that receive a mirrored contribution of the reflect padding?
        const bool fold = AB && ab.dsrc && ab.pad_mode == PASE_PAD_REFLECT &&
                          (q0_ld <= ab.padL || q0_ld + SW_POS - 1 >= ab.T - 1 - padR);
        if constexpr (AB) {
            if (fold && grow_ok) {
                // the mirrored edge gradients, added into the raw tile in place (each thread patches the 16 values it reads
                // itself; two stages per sequence, a rolled loop: nothing of it is in the other stages' code path)
                const float* row = ab.dsrc + ((size_t)s_ld * ab.dsrc_ctot + ab.dsrc_coff + grow) * (size_t)ab.Tp;
                float* dl = &Dr[grow * SW_RP + 16 * gpart];
#pragma unroll 1
                for (int e = 0; e < 16; ++e) {
                    const int q = q0_ld + 16 * gpart + e;
                    if (q >= p.Ncols) break;
                    float a = dl[e];
                    if (q >= 1 && q <= ab.padL) a += row[ab.padL - q];
                    if (q >= ab.T - 1 - padR && q <= ab.T - 2) a += row[ab.padL + 2 * (ab.T - 1) - q];
                    dl[e] = a;
                }
            }
        }
        const bool whole = AB && p.M == 64 && q0_ld + SW_POS <= p.Ncols;       // (uniform)
        auto convert_g = [&](auto whole_tag) __attribute__((always_inline)) {
        constexpr bool WHOLE = decltype(whole_tag)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v[8];
            if constexpr (AB) {
                const f32x4* yl = reinterpret_cast<const f32x4*>(&Yr[grow * SW_RP + 16 * gpart + 8 * h]);
                const f32x4* dl = reinterpret_cast<const f32x4*>(&Dr[grow * SW_RP + 16 * gpart + 8 * h]);
                const f32x4 y0 = yl[0], y1 = yl[1], d0 = dl[0], d1 = dl[1];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int q = q0_ld + 16 * gpart + 8 * h + e;
                    const bool ok = WHOLE || (grow_ok && q < p.Ncols);
                    const float yv = e < 4 ? y0[e & 3] : y1[e & 3];
                    // dA in grad_post_act's order: padded data gradient (+ mirrored edges, patched in above), pooled branch
                    float dA = ab.dsrc ? (e < 4 ? d0[e & 3] : d1[e & 3]) : 0.f;
                    if (ab.dpool) dA += (frame_of(q) == f_first ? dp0 : dp1) * ab.pool_inv;
                    // (positions past Ncols / rows past M must come out as zeros, not as -scale * (m1 + xhat m2))
                    v[e] = ok ? act_bwd_dy(rc, HB, yv, dA) : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = gv[8 * h + e];
                    v[e] = (p.g_alpha && t < 0.f) ? t * g_al : t;
                }
            }
            u32x4 o[3];
            pase_split_bf16x3_rne(v, o);
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) As[pz * SW_ACH + (2 * gpart + h) * 64 + grow] = o[pz];
        }
        };
        if (whole) convert_g(std::true_type{});
        else convert_g(std::false_type{});
        if (2 * tid < SW_NW) {
            u32x4 w0[3], w1[3];
            if constexpr (AB) {
                float xw[9];
                const int u0 = q0_ld - p.padL + 2 * tid;
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    int u = u0 + e;
                    // samples outside the sequence were not copied (zero padding; with reflect padding the ones a single
                    // reflection does not bring back inside: sequences shorter than the window image).  Their LDS words are
                    // whatever the previous stage left -- possibly NaN / Inf patterns -- and 0 * NaN would reach dfilt through
                    // dead positions: select 0 exactly as the plain path does
                    if (p.pad_mode == PASE_PAD_REFLECT) u = sx_reflect(u, p.Tz);
                    xw[e] = (u >= 0 && u < p.Tz) ? Xr[2 * tid + e] : 0.f;
                }
                pase_split_two_windows(xw, w0, w1);
            } else {
                pase_split_two_windows(xv, w0, w1);
            }
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) {
                Bs[pz * SW_NW + 2 * tid] = w0[pz];
                Bs[pz * SW_NW + 2 * tid + 1] = w1[pz];
            }
        }
    };

    f32x16 accH[4], accS[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accH[j][r] = 0.f;
            accS[j][r] = 0.f;
        }
    load_stage(st_begin);
    if (AB) {                 // (the plain form's first loads are waited for by their first use)
        sx_dma_wait();
        __syncthreads();
    }
    for (long st = st_begin; st < st_end; ++st) {
        store_stage();
        __syncthreads();
        if (st + 1 < st_end) load_stage(st + 1);          // in flight under this stage's MFMAs
        constexpr int PZA[5] = {1, 0, 2, 0, 1}, PZB[5] = {1, 2, 0, 1, 0};
#pragma unroll
        for (int kg = 0; kg < SW_POS / 16; ++kg) {
            u32x4 a[3];
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) a[pz] = As[pz * SW_ACH + (2 * kg + fk) * 64 + wm * 32 + fr];
            const u32x4* xb = &Bs[wn * 128 + fr + 16 * kg + 8 * fk];
#pragma unroll
            for (int jp = 0; jp < 4; jp += 2) {
                u32x4 b0[3], b1[3];
#pragma unroll
                for (int pz = 0; pz < 3; ++pz) {
                    b0[pz] = xb[pz * SW_NW + 32 * jp];
                    b1[pz] = xb[pz * SW_NW + 32 * (jp + 1)];
                }
#pragma unroll
                for (int pi = 0; pi < 5; ++pi) {
                    accS[jp] = pase_mfma_bf16_32x32x16(a[PZA[pi]], b0[PZB[pi]], accS[jp]);
                    accS[jp + 1] = pase_mfma_bf16_32x32x16(a[PZA[pi]], b1[PZB[pi]], accS[jp + 1]);
                }
                accH[jp] = pase_mfma_bf16_32x32x16(a[0], b0[0], accH[jp]);
                accH[jp + 1] = pase_mfma_bf16_32x32x16(a[0], b1[0], accH[jp + 1]);
            }
        }
        if (AB) sx_dma_wait();                              // this wave's copies of the next stage's raw tiles have landed ...
        __syncthreads();                                    // every wave has read the image: the next stage may overwrite it
    }

    // ---- += into the caller-zeroed dfilt (other workgroups hold the other position ranges) ---------------------------
    const int rbase = wm * 32 + 4 * fk;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = wn * 128 + j * 32 + fr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = rbase + (r & 3) + 8 * (r >> 2);
            if (m < p.M && kk < p.taps) atomicAdd(p.dw + (size_t)m * p.ldw + kk, accH[j][r] + accS[j][r]);
        }
    }
}

}  // namespace

bool pase_sinc_x6_plan(const PaseConvGemm& p, PaseSincPlan& pl) {
    if (p.Cin != 1 || p.x_ctot < 1 || p.taps < 32 || p.taps > 16 * SX_KGMAX || p.M > 64 || p.M < 1) return false;
    if (p.stride != 1 || p.tapstep != 1 || p.tap_major || p.ps != 1 || p.poff != 0) return false;
    if (p.epilogue != PASE_EPI_STORE || p.post_op != PASE_POST_NONE || p.splitk > 1) return false;
    if (p.in_scale || p.in_alpha) return false;
    if (p.pad_mode == PASE_PAD_REFLECT && p.padL >= p.Tin) return false;
    if (p.Cout_store != p.M) return false;
    if (p.Tout >= (1 << 28)) return false;                  // the epilogue's 32-bit lane offsets (4 rows of the output)
    pl.n_kg = (p.taps + 15) / 16;
    pl.nwin = SX_BN + 16 * pl.n_kg - 8;
    pl.tiles_per_seq = (p.Ncols + SX_BN - 1) / SX_BN;
    pl.pack_bytes = (long)2 * pl.n_kg * 192 * 16;
    if ((long)p.S * pl.tiles_per_seq >= 0x7fffffffL) return false;
    return true;
}

int pase_sinc_x6_pack(const PaseConvGemm& p, const PaseSincPlan& pl, hipStream_t st) {
    const int total = 2 * pl.n_kg * 64;
    PASE_LAUNCH(sinc_x6_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), st, p.wt,
                reinterpret_cast<u32x4*>(const_cast<void*>(p.wx6)), p.M, p.ldwt, p.taps, pl.n_kg, p.w, p.ldw);
    PASE_CHECK_LAUNCH();
    return 0;
}

int pase_sinc_x6_launch(const PaseConvGemm& p, const PaseSincPlan& pl, hipStream_t st) {
    const long ntiles = (long)p.S * pl.tiles_per_seq;
    long nwg = p.max_wg > 0 ? 2L * p.max_wg : 512;          // two workgroups per CU
    if (nwg > ntiles) nwg = ntiles;
    PASE_LAUNCH(sinc_x6_fwd_kernel, dim3((unsigned)nwg), dim3(SX_NT), st, p, pl);
    PASE_CHECK_LAUNCH();
    return 0;
}

bool pase_sinc_x6_wgrad_plan(const PaseWgrad& w, PaseSincPlan& pl) {
    if (w.Cin != 1 || w.taps < 32 || w.taps > SX_BN || w.M > 64 || w.M < 1) return false;
    if (w.stride != 1 || w.tapstep != 1 || w.tap_major || w.dbias) return false;
    if (w.in_scale || w.in_alpha) return false;
    if (w.pad_mode == PASE_PAD_REFLECT && w.padL >= w.Tz) return false;
    pl.n_kg = SW_POS / 16;
    pl.nwin = SW_NW;
    pl.tiles_per_seq = (w.Ncols + SW_POS - 1) / SW_POS;       // stages per sequence
    pl.pack_bytes = 16;                                         // no pack: both operands are converted while they are staged
    return true;
}

// the one-channel plan with its gradient operand evaluated on load (sinc_x6_wgrad_kernel<true>): ab describes the same
// (S, M, Ncols) tensor the plain launch would read from w.g
bool pase_sinc_x6_wgrad_act_bwd_ok(const PaseWgrad& w, const PaseActBwd& ab) {
    if (ab.S != w.S || ab.C != w.M || ab.T != w.Ncols || !ab.y) return false;
    if (ab.has_bn < 0 || ab.has_bn > 2 || (ab.has_bn == 1 && !ab.sums)) return false;
    if (w.g_alpha || w.dbias) return false;
    if (ab.dpool && ab.pool_d < 16) return false;        // a staged run of 16 positions must touch at most two pooled frames
    if (ab.dsrc && (ab.padL < 0 || ab.Tp < ab.T + ab.padL)) return false;      // every step has its own column in the padded gradient
    return true;
}

int pase_sinc_x6_wgrad_launch(const PaseWgrad& w, const PaseSincPlan& pl, hipStream_t st, const PaseActBwd* ab) {
    const long total = (long)w.S * pl.tiles_per_seq;
    // two workgroups per CU; at least 8 stages per workgroup (the atomic flush of a 64 x 256 tile costs about two stages)
    long nwg = w.max_wg > 0 ? 2L * w.max_wg : 512;
    if (w.splitk > 0) nwg = w.splitk;
    if (nwg > (total + 7) / 8) nwg = (total + 7) / 8;
    if (nwg < 1) nwg = 1;
    if (ab && ab->has_bn == 1) {
        PASE_LAUNCH((sinc_x6_wgrad_kernel<true, 1>), dim3((unsigned)nwg), dim3(SX_NT), st, w, pl, *ab);
    } else if (ab && ab->has_bn == 2) {
        PASE_LAUNCH((sinc_x6_wgrad_kernel<true, 2>), dim3((unsigned)nwg), dim3(SX_NT), st, w, pl, *ab);
    } else if (ab) {
        PASE_LAUNCH((sinc_x6_wgrad_kernel<true, 0>), dim3((unsigned)nwg), dim3(SX_NT), st, w, pl, *ab);
    } else {
        PASE_LAUNCH((sinc_x6_wgrad_kernel<false, 0>), dim3((unsigned)nwg), dim3(SX_NT), st, w, pl, PaseActBwd{});
    }
    PASE_CHECK_LAUNCH();
    return 0;
}

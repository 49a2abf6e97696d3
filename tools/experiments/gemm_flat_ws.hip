// NOT BUILT.  Experiment kept for the record: producer / consumer persistent variant of the flat 1x1 GEMM.
// Measured equal to the single-role flat kernel on long reductions (116 vs 115 TFLOP/s, K = 21 525) and 3-17 % slower on
// the K = 256 heads; it was therefore not adopted.  It was wired into pase_conv_gemm behind PASE_FLAT_WS and passed the
// flat-path tests of tests/test_conv_gemm.py on the emulator and on the GPU.
// gemm_flat_ws.hip -- producer / consumer ("wave-specialised") variant of the flat 1x1 path of pase_conv_gemm.
//
//   Y[s, m, q] = bias[m] + sum_k wt[k, m] * act(X[s, k, q])          (taps = 1, stride = 1, no padding)
//
// i.e. the worker heads (Conv1d(256 -> num_outputs * r, 1) with the fused r-context MSE, pase/models/Minions/
// minions.py:510 + pase/losses.py:6-37), their data-gradients (K = 21 525), the stacked first layers and the
// concatenated dense-skip / W projection (pase/models/frontend.py:182,195,262).
//
// Why: in conv_gemm.hip every wave alternates between an MFMA phase and a stage-turnover phase (LDS stores, barrier,
// next stage's global loads), and the two workgroups of a CU do so in lock-step, which leaves the matrix pipe idle a
// third of the time (rocprofv3: MFMA busy 0.48-0.56 on the flat instantiation).  Here a 512-thread workgroup splits
// its 8 waves by ROLE: waves 0-3 only read operand fragments from LDS and issue MFMAs; waves 4-7 only move data
// (global -> registers -> on-load PReLU / affine -> LDS), one stage ahead.  One s_barrier per stage hands a stage
// over in both directions (B_n: "stage n is in LDS" and "the consumers are done with stage n-1"); the workgroup is
// persistent over (tile, split-K slice) work items, so the first stage of the next tile lands in LDS while the
// consumers run the epilogue of the current one.
//
// Tile 128 x 128, stage = 32 k-rows: A slab [32][128] (+4 pad) and X slab [32][128], double-buffered = 66.6 KB,
// <= 128 VGPRs: two workgroups (4 waves per SIMD: two consumers, two producers) per CU.
#include <cstdlib>
#include <type_traits>

#include "hip_compat.h"
#include "pase_amd.h"

namespace {

constexpr int WS_THREADS = 512;
constexpr int KS = 32;
constexpr int BM = 128, BN = 128, LDA = BM + 4;

struct alignas(16) F4 { float x, y, z, w; };

struct WsPlan {
    int n_row_tiles, n_col_tiles, ntiles, splitk, G, g_per, nitems;
    unsigned ncols_magic, rctx_magic;
};

__device__ __forceinline__ unsigned div_magic(unsigned e, unsigned magic) {
    return magic ? (unsigned)(((unsigned long long)e * magic) >> 32) : e;
}

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// XFM: 0 no on-load transform, 1 PReLU slope only, 2 affine + slope.  EPI: 0 store (bias / split-K atomics /
// BatchNorm partial sums), 1 fused r-context MSE.
template <int XFM, int EPI>
__global__ void __launch_bounds__(WS_THREADS, 2) gemm_flat_ws_kernel(PaseConvGemm p, WsPlan pl) {
    __shared__ __attribute__((aligned(16))) float As[2][KS][LDA];
    __shared__ __attribute__((aligned(16))) float Xs[2][KS][BN];
    __shared__ float red[2][BM][2];            // BatchNorm partials across the two column waves (never aliased)

    const int tid = threadIdx.x;
    const bool producer = pase_uniform(tid >> 8) != 0;
    const int t = tid & 255;
    const int ntot = p.S * p.Ncols;
    const bool want_stats = EPI == 0 && p.stat_part != nullptr;      // uniform

    // both roles walk the same list of work items and stages, so their barrier counts match by construction
    auto item_decode = [&](int w, int& m0, int& n0, int& nt, int& split, int& g_begin, int& g_end) __attribute__((always_inline)) {
        split = w / pl.ntiles;
        const int tile = xcd_swizzle(w - split * pl.ntiles, pl.ntiles);
        const int mt = tile % pl.n_row_tiles;
        nt = tile / pl.n_row_tiles;
        m0 = mt * BM;
        n0 = nt * BN;
        g_begin = split * pl.g_per;
        g_end = min(pl.G, g_begin + pl.g_per);
    };

    if (producer) {
        // ---- loader waves: thread -> slab rows r0 + 8 i (i < 4), 4 consecutive floats at column 4 * c4 -------------
        const int r0 = t >> 5, c4 = (t & 31) * 4;
        F4 areg[4], xreg[4];
        float ps[4], ph[4], pa[4];
        int buf = 0;
        bool have = false;                     // registers hold a loaded stage
        int xoff = 0;
        bool xok = false;
        unsigned a_col = 0;
        int kcur = 0;
        auto locate = [&](int m0, int n0) __attribute__((always_inline)) {
            a_col = (unsigned)min(m0 + c4, p.ldwt - 4);
            const unsigned n = (unsigned)(n0 + c4);
            int s = (int)div_magic(n, pl.ncols_magic);
            int u = (int)n - s * p.Ncols;
            if (u < 0) { --s; u += p.Ncols; }
            xok = (int)n < ntot;
            xoff = xok ? (s * p.x_ctot + p.x_coff) * p.Tin + u : p.x_coff * p.Tin;
        };
        auto load = [&](int g) __attribute__((always_inline)) {
            kcur = g * KS;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = min(kcur + r0 + 8 * i, p.K - 1);
                areg[i] = *reinterpret_cast<const F4*>(p.wt + ((unsigned)k * (unsigned)p.ldwt + a_col));
                xreg[i] = *reinterpret_cast<const F4*>(p.x + (unsigned)(xoff + k * p.Tin));
                if (XFM == 2) { ps[i] = p.in_scale[k]; ph[i] = p.in_shift[k]; }
                if (XFM >= 1) pa[i] = p.in_alpha ? p.in_alpha[k] : 1.f;
            }
        };
        auto store = [&](int b) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + 8 * i;
                F4 a = areg[i], x = xreg[i];
                if (kcur + r >= p.K) a = F4{0.f, 0.f, 0.f, 0.f};             // ragged last stage: zero weights
                if (XFM == 2) {
                    x.x = fmaf(x.x, ps[i], ph[i]); x.y = fmaf(x.y, ps[i], ph[i]);
                    x.z = fmaf(x.z, ps[i], ph[i]); x.w = fmaf(x.w, ps[i], ph[i]);
                }
                if (XFM >= 1) {
                    x.x = x.x > 0.f ? x.x : x.x * pa[i]; x.y = x.y > 0.f ? x.y : x.y * pa[i];
                    x.z = x.z > 0.f ? x.z : x.z * pa[i]; x.w = x.w > 0.f ? x.w : x.w * pa[i];
                }
                if (!xok) x = F4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<F4*>(&As[b][r][c4]) = a;
                *reinterpret_cast<F4*>(&Xs[b][r][c4]) = x;
            }
        };
        // software pipeline over the flattened (item, stage) sequence: registers run one stage ahead of LDS
        int w = blockIdx.x;
        int m0, n0, nt, split, g, g_end;
        bool more = w < pl.nitems;
        if (more) {
            item_decode(w, m0, n0, nt, split, g, g_end);
            locate(m0, n0);
            load(g);
            have = true;
        }
        while (have) {
            store(buf);
            // advance to the next stage of the sequence and start its loads before handing this one over
            ++g;
            const bool item_done = g >= g_end;
            if (item_done) {
                w += gridDim.x;
                more = w < pl.nitems;
                if (more) {
                    item_decode(w, m0, n0, nt, split, g, g_end);
                    locate(m0, n0);
                }
            }
            have = more;
            if (have) load(g);
            __syncthreads();                                     // B_n
            if (item_done && want_stats) __syncthreads();        // the consumers' epilogue barrier
            buf ^= 1;
        }
        return;
    }

    // ---- MFMA waves ----------------------------------------------------------------------------------------------
    const int lane = t & 63;
    const int wave = pase_uniform(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fk = lane >> 5;
    int buf = 0;
    for (int w = blockIdx.x; w < pl.nitems; w += gridDim.x) {
        int m0, n0, nt, split, g_begin, g_end;
        item_decode(w, m0, n0, nt, split, g_begin, g_end);
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int g = g_begin; g < g_end; ++g) {
            __syncthreads();                                     // B_n: stage in LDS
            const float* aL = &As[buf][fk][wm * 64 + fr];
            const float* x0L = &Xs[buf][fk][wn * 64 + fr];
            float pa0 = aL[0], pa1 = aL[32], pb0 = x0L[0], pb1 = x0L[32], qa0, qa1, qb0, qb1;
            pase_static_for<KS / 4>([&](auto it_tag) __attribute__((always_inline)) {
                constexpr int it = decltype(it_tag)::value;
                constexpr int k1 = 2 * it + 1, k2 = (2 * it + 2 < KS / 2) ? 2 * it + 2 : 2 * it + 1;
                qa0 = aL[k1 * 2 * LDA]; qa1 = aL[k1 * 2 * LDA + 32]; qb0 = x0L[k1 * 2 * BN]; qb1 = x0L[k1 * 2 * BN + 32];
                PASE_SCHED_BARRIER();
                acc[0][0] = pase_mfma_32x32x2(pa0, pb0, acc[0][0]);
                acc[0][1] = pase_mfma_32x32x2(pa0, pb1, acc[0][1]);
                acc[1][0] = pase_mfma_32x32x2(pa1, pb0, acc[1][0]);
                acc[1][1] = pase_mfma_32x32x2(pa1, pb1, acc[1][1]);
                PASE_SCHED_BARRIER();
                pa0 = aL[k2 * 2 * LDA]; pa1 = aL[k2 * 2 * LDA + 32]; pb0 = x0L[k2 * 2 * BN]; pb1 = x0L[k2 * 2 * BN + 32];
                PASE_SCHED_BARRIER();
                acc[0][0] = pase_mfma_32x32x2(qa0, qb0, acc[0][0]);
                acc[0][1] = pase_mfma_32x32x2(qa0, qb1, acc[0][1]);
                acc[1][0] = pase_mfma_32x32x2(qa1, qb0, acc[1][0]);
                acc[1][1] = pase_mfma_32x32x2(qa1, qb1, acc[1][1]);
                PASE_SCHED_BARRIER();
            });
            buf ^= 1;
        }

        // ---- epilogue (D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) --------------------
        const int rbase = m0 + wm * 64 + 4 * (lane >> 5);
        int cs[2], cq[2];
        bool cok[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned n = (unsigned)(n0 + wn * 64 + b * 32 + fr);
            cok[b] = (int)n < ntot;
            int s = (int)div_magic(n, pl.ncols_magic);
            int u = (int)n - s * p.Ncols;
            if (u < 0) { --s; u += p.Ncols; }
            cs[b] = cok[b] ? s : 0;
            cq[b] = cok[b] ? u : 0;
        }
        if (EPI == 0) {
            const float* biasp = (p.bias && split == 0) ? p.bias : nullptr;
            int cbase[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) cbase[b] = (cs[b] * p.y_ctot + p.y_coff) * p.Tout + cq[b];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float bvs[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bvs[r] = 0.f;
                if (biasp) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
                        if (m < p.M) bvs[r] = biasp[m];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
                    const bool mok = m < p.M;
                    const int rowoff = m * p.Tout;
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        float v = acc[a][b][r] + bvs[r];
                        if (p.post_op == PASE_POST_LOG) v = p.post_scale * logf(v == 0.f ? p.post_eps : v);
                        if (p.post_op == PASE_POST_RELU) v = fmaxf(v, 0.f);
                        if (p.post_op == PASE_POST_SQRTPOS) v = sqrtf(fmaxf(v, 0.f));
                        if (mok && cok[b]) {
                            float* dst = p.y + (unsigned)(cbase[b] + rowoff);
                            if (pl.splitk > 1) atomicAdd(dst, v);
                            else *dst = v;
                            s1 += v;
                            s2 += v * v;
                        }
                    }
                    if (want_stats) {
                        s1 = pase_half_sum_lane31(s1);
                        s2 = pase_half_sum_lane31(s2);
                        if (fr == 31) {
                            red[wn][m - m0][0] = s1;
                            red[wn][m - m0][1] = s2;
                        }
                    }
                }
            }
            if (want_stats) {
                __syncthreads();                                  // (matched by the producers)
                if (t < BM && m0 + t < p.M) {
                    float* dst = p.stat_part + ((size_t)nt * p.M + m0 + t) * 2;
                    dst[0] = red[0][t][0] + red[1][t][0];
                    dst[1] = red[0][t][1] + red[1][t][1];
                }
            }
        } else {
            // rows m = d * r + j, columns (b, t); target = label[b, d, t + j - r/2].  Label / bias loads of a 32-row
            // block are issued together, ahead of the stores (see conv_gemm.hip mse_rows)
            float lsum = 0.f;
            const int half = p.r_ctx / 2;
            int lbase[2], obase[2], tb[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                tb[b] = cq[b] - half;
                lbase[b] = cs[b] * p.label_D * p.Ncols + tb[b];
                obase[b] = cs[b] * p.M * p.Ncols + cq[b];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {            // 8 rows at a time keeps the kernel under 128 VGPRs
                    float tg[8][2], bvs[8];
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int r = h * 8 + rr;
                        const int m = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
                        const bool mok = m < p.M;
                        const int d = (int)div_magic((unsigned)m, pl.rctx_magic);
                        const int jj = m - d * p.r_ctx;
                        bvs[rr] = (mok && p.bias) ? p.bias[m] : 0.f;
                        const int lrow = d * p.Ncols + jj;
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            tg[rr][b] = 0.f;
                            if (mok && cok[b] && (unsigned)(tb[b] + jj) < (unsigned)p.Ncols)
                                tg[rr][b] = p.label[(unsigned)(lbase[b] + lrow)];
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int r = h * 8 + rr;
                        const int m = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
                        const bool mok = m < p.M;
                        const int orow = m * p.Ncols;
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            if (mok && cok[b]) {
                                const float pred = acc[a][b][r] + bvs[rr];
                                const float diff = pred - tg[rr][b];
                                lsum += diff * diff;
                                const unsigned o = (unsigned)(obase[b] + orow);
                                if (p.y) p.y[o] = pred;
                                if (p.grad_out) p.grad_out[o] = diff * p.grad_scale;
                            }
                        }
                    }
                }
            }
            lsum = pase_wave_sum64(lsum);
            if (lane == 0) atomicAdd(p.loss_acc, (double)lsum);
        }
    }
}

unsigned magic_of(int d) {
    return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned long long)d);
}

}  // namespace

// Called by pase_conv_gemm for flat-eligible launches (conv_gemm.hip decides; splitk already resolved by its plan).
// Returns -100 when the launch is outside what this variant covers (the caller then uses the generic kernel).
extern "C" int pase_gemm_flat_ws(const PaseConvGemm* d, int splitk, void* stream) {
    const PaseConvGemm p = *d;
    if (p.taps != 1 || p.stride != 1 || p.padL != 0 || p.ps != 1 || p.poff != 0) return -100;
    if ((p.Ncols % 4) || (p.Tin % 4) || (((unsigned long long)(size_t)p.x) % 16) || p.Ncols != p.Tin) return -100;
    if (p.post_op == PASE_POST_POW || p.post_op == PASE_POST_LOGPOW || p.post_op == PASE_POST_MAG) return -100;
    if (p.epilogue == PASE_EPI_STORE && p.Tout != p.Ncols) return -100;
    if (p.epilogue == PASE_EPI_STORE && p.Cout_store != p.M) return -100;
    if (p.in_scale && !p.in_alpha) return -100;
    WsPlan pl;
    pl.n_row_tiles = (p.M + BM - 1) / BM;
    const long ntot = (long)p.S * p.Ncols;
    pl.n_col_tiles = (int)((ntot + BN - 1) / BN);
    pl.ntiles = pl.n_row_tiles * pl.n_col_tiles;
    pl.G = (p.K + KS - 1) / KS;
    if (splitk < 1) splitk = 1;
    if (splitk > pl.G) splitk = pl.G;
    pl.g_per = (pl.G + splitk - 1) / splitk;
    pl.splitk = (pl.G + pl.g_per - 1) / pl.g_per;
    if (pl.splitk > 1 && (p.stat_part || p.epilogue != PASE_EPI_STORE || p.post_op != PASE_POST_NONE)) return -100;
    pl.nitems = pl.ntiles * pl.splitk;
    pl.ncols_magic = magic_of(p.Ncols);
    pl.rctx_magic = magic_of(p.r_ctx);
    // 2 workgroups per CU x 256 CUs (PASE_WS_SLOTS: test hook that forces several work items per workgroup)
    static const int slots = [] { const char* e = getenv("PASE_WS_SLOTS"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
    const dim3 grid((unsigned)(pl.nitems < slots ? pl.nitems : slots)), block(WS_THREADS);
    hipStream_t st = (hipStream_t)stream;
    const int xfm = p.in_scale ? 2 : (p.in_alpha ? 1 : 0);
    if (p.epilogue == PASE_EPI_MSE_CTX) {
        if (xfm == 0) PASE_LAUNCH((gemm_flat_ws_kernel<0, 1>), grid, block, st, p, pl);
        else if (xfm == 1) PASE_LAUNCH((gemm_flat_ws_kernel<1, 1>), grid, block, st, p, pl);
        else PASE_LAUNCH((gemm_flat_ws_kernel<2, 1>), grid, block, st, p, pl);
    } else {
        if (xfm == 0) PASE_LAUNCH((gemm_flat_ws_kernel<0, 0>), grid, block, st, p, pl);
        else if (xfm == 1) PASE_LAUNCH((gemm_flat_ws_kernel<1, 0>), grid, block, st, p, pl);
        else PASE_LAUNCH((gemm_flat_ws_kernel<2, 0>), grid, block, st, p, pl);
    }
    PASE_CHECK_LAUNCH();
    return 0;
}

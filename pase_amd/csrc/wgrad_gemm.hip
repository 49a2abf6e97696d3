// wgrad_gemm.hip -- weight-gradient contraction on v_mfma_f32_32x32x2_f32.
//
//   dW[m, j=(ci,kk)] += sum_{s,q} G[s, m, q] * act(bn(Z[s, ci, q*stride + kk*tapstep - padL]))
//   dbias[m]         += sum_{s,q} G[s, m, q]                      (extra all-ones column j == Kw)
//
// Replaces autograd's conv1d / conv_transpose1d / linear weight+bias gradients for every layer the
// reference builds from nn.Conv1d, nn.ConvTranspose1d and torchqrnn's nn.Linear
// (pase/models/modules.py:1047-1051, :543, :571-575; frontend.py:182,195; Minions/minions.py:510),
// driven by `tot_loss.backward()` in WorkerScheduler/worker_scheduler.py:67.
// For nn.Conv1d G is dY and Z the layer input; for nn.ConvTranspose1d the roles swap (G = layer
// input at the low rate, with its PReLU applied on load via g_alpha; Z = dY at the high rate) and the
// result lands directly in the (in, out, k) weight layout.
//
// GEMM view: M x Nw x Kred with Kred = S*Ncols (19 200 ... 3 072 000): the reduction is the long
// axis, so the grid is (row tiles x col tiles x split-K) and partial tiles are combined with fp32
// global atomics into a caller-zeroed dW.
//
// gfx950 mapping: 4 waves, 2x2 32x32x2 MFMA tiles per wave (as conv_gemm.hip).  Per stage the block
// takes a chunk of 32 consecutive time steps of one sequence and stages, with coalesced row loads,
// (a) the [BM x 32] slab of G and (b) the raw sliding-window SPANS (31*stride + taps samples) of the
// few input channels that the tile's (ci,kk) columns touch -- not an im2col tile.  Each lane's two
// B columns map to fixed LDS offsets (channel row + tap), so the MFMA loop is
// `Zs[off_j + kq*stride]`: no index arithmetic, no per-element gathers.
#include "hip_compat.h"
#include "pase_amd.h"
#include "conv_x6c.h"
#include "sinc_x6.h"

namespace {

constexpr int NTHREADS = 256;
constexpr int BKQ = 32;                 // reduction positions per stage
constexpr int ZS_ONES = 320;            // region of 1.0f (bias column / padding columns)
// staged span floats per stage = ZPT * 256; two instantiations: ZPT = 9 (2304 floats: every k > 1 layer
// of PASE+; fits 2 waves/SIMD) and ZPT = 19 (4864 floats: 1x1 layers with 128 input channels per tile,
// the stride-10 block-1 layer; 1 wave/SIMD)
constexpr int ZPT_SMALL = 9, ZPT_LARGE = 19;

struct WgradPlan {
    int bias_rowsum;   // dbias from row sums of the G slab in column-tile 0 (no extra all-ones column tile)
    int zshift;        // ZV: samples between the aligned load address and the span start (0..3)
    int gvec, flat, SPANW, chunks_per_seq, n_chunks, kt_per_split, n_row_tiles, n_col_tiles;
    unsigned span_magic, ncols_magic;
};

__device__ __forceinline__ unsigned div_magic(unsigned e, unsigned magic) {
    // e / d with magic = ceil(2^32 / d); magic == 0 encodes d == 1 (2^32 does not fit)
    return magic ? (unsigned)(((unsigned long long)e * magic) >> 32) : e;
}

// ZV = 1: the spans are staged with 16-byte loads (ZPT = float4 slots per thread): every channel row of the slab is
// SPANW rounded up to whole float4s, read from the 16-byte-aligned address `zshift` samples in front of the span.
// That is the large-span case (the stride-10 block-1 layer: 14 channels x 330 samples per stage) at a quarter of the
// load / LDS-store instructions and 14 fewer offset registers, which is what lets it run two workgroups per CU.
// X6 = 1: split-bf16 contraction (PaseWgrad::x6): a stage is two 16-deep v_mfma_f32_32x32x16_bf16 steps over its 32 time
// steps.  The G slab is split into its three bf16 pieces when it is staged (LDS image [8 time steps][plane][row][8 bf16]:
// one ds_read_b128 per A fragment); the spans stay fp32 in LDS and each wave splits its B fragment (the lane's 8
// consecutive time steps of its (channel, tap) column) when it reads it.
template <int BM, int BN, int ZPT, int ZV = 0, int X6 = 0>
__global__ void __launch_bounds__(NTHREADS, ((ZPT <= ZPT_SMALL || ZV) ? 2 : 1)) wgrad_gemm_kernel(PaseWgrad p, WgradPlan pl) {
    constexpr int ZW = ZV ? 4 : 1;                     // floats per slot
    constexpr int ZS_DATA = ZPT * NTHREADS * ZW;
    constexpr int ZS_TOTAL = ZS_DATA + ZS_ONES;
    constexpr int WAVES_N = BN / 64;
    constexpr int A_ROWS = BM / 8;
    constexpr int AX_BUF = (BKQ / 8) * 3 * BM;         // x6: 16-byte chunks per buffer
    constexpr int A_BYTES = X6 ? 2 * AX_BUF * 16 : 2 * BKQ * (BM + 1) * (int)sizeof(float);
    __shared__ __attribute__((aligned(16))) unsigned char As_raw[A_BYTES];
    float (*As)[BKQ][BM + 1] = reinterpret_cast<float (*)[BKQ][BM + 1]>(As_raw);
    u32x4* AsX = reinterpret_cast<u32x4*>(As_raw);
    __shared__ __attribute__((aligned(16))) float Zs[2][ZS_TOTAL];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = pase_uniform(tid >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int fr = lane & 31;
    const int fk = lane >> 5;

    const int tiles = pl.n_row_tiles * pl.n_col_tiles;
    const int tile = blockIdx.x % tiles;
    const int split = blockIdx.x / tiles;
    const int mt = tile % pl.n_row_tiles;
    const int ct = tile / pl.n_row_tiles;
    const int m0 = mt * BM;
    const int j0 = ct * BN;

    const int Kw = p.Cin * p.taps;
    const int Nw = Kw + ((p.dbias && !pl.bias_rowsum) ? 1 : 0);
    const int c_begin = split * pl.kt_per_split;
    const int c_end = min(pl.n_chunks, c_begin + pl.kt_per_split);
    if (c_begin >= c_end) return;   // whole block exits together (no barrier reached yet)

    // channels touched by this tile's columns, and this lane's two fixed B offsets
    const int c_lo = j0 / p.taps;
    const int j_last = min(j0 + BN, Kw) - 1;
    const int NC = j_last >= j0 ? j_last / p.taps - c_lo + 1 : 0;
    int boff[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = j0 + wn * 64 + b * 32 + fr;
        if (j < Kw) {
            const int ci = j / p.taps, kk = j - ci * p.taps;
            boff[b] = (ci - c_lo) * pl.SPANW + (ZV ? pl.zshift : 0) + (p.tapstep > 0 ? kk : p.taps - 1 - kk);
        } else {
            boff[b] = ZS_DATA;      // ones (bias column when j == Kw, discarded otherwise)
        }
    }
    const int zstep = pl.flat ? 1 : p.stride;
    bool gvec_next = false;
    for (int i = tid; i < ZS_ONES; i += NTHREADS) { Zs[0][ZS_DATA + i] = 1.f; Zs[1][ZS_DATA + i] = 1.f; }

    // ---- loader state (fixed for the whole tile) ------------------------------------------------
    // G slab [BM x 32]: thread -> rows (tid >> 3) + 32*i, time steps k4..k4+3 (float4) when gvec, else
    // rows (tid >> 5) + 8*i at time step tid & 31.  Rows past M re-read row M-1 (never stored).
    constexpr int GS = A_ROWS / 4;
    float areg[A_ROWS];
    float zreg[ZPT * ZW];
    const int k4 = (tid & 7) * 4;
    unsigned goff[GS];
    float g_al[GS];
    const bool has_ga = p.g_alpha != nullptr;
#pragma unroll
    for (int i = 0; i < GS; ++i) {
        const int m = min(m0 + (tid >> 3) + 32 * i, p.M - 1);
        goff[i] = (unsigned)((p.g_coff + m) * p.Tg);
        g_al[i] = has_ga ? p.g_alpha[m] : 1.f;
    }
    // Z spans: slot t = element e = tid + 256 t of the [NC][SPANW] slab; interior offset cl*Tz + i and the
    // channel's on-load parameters are per-tile constants
    const int ntot = p.S * p.Ncols;
    const int total = NC * pl.SPANW;
    unsigned zoff[ZPT];
    constexpr bool PRE = ZPT <= ZPT_SMALL || ZV; // on-load (scale, shift, alpha) per slot held in registers
    constexpr int NPRE = PRE ? ZPT : 1;
    float z_sc[NPRE], z_sh[NPRE], z_al[NPRE];
    const bool has_aff = p.in_scale != nullptr, has_al = p.in_alpha != nullptr;
    const bool has_xf = has_aff || has_al;
#pragma unroll
    for (int t = 0; t < ZPT; ++t) {
        // slot t = element (ZV: float4) e of the [NC][SPANW] slab; slots past the slab re-read its last element
        const int e = min((tid + NTHREADS * t) * ZW, max(total - ZW, 0));
        const int cl = (int)div_magic((unsigned)e, pl.span_magic);
        zoff[t] = (unsigned)(cl * p.Tz + (e - cl * pl.SPANW));
        if (PRE) {
            const int ci = min(c_lo + cl, p.Cin - 1);
            z_sc[t] = has_aff ? p.in_scale[ci] : 1.f;
            z_sh[t] = has_aff ? p.in_shift[ci] : 0.f;
            z_al[t] = has_al ? p.in_alpha[ci] : 1.f;
        }
    }
    unsigned zmask = 0u;
    bool zfast_next = true, gkeep_next = true;

    auto load_stage = [&](int c) __attribute__((always_inline)) {
        int s, q0;
        if (pl.flat) {
            const int nc = c * BKQ;
            s = nc / p.Ncols;
            q0 = nc - s * p.Ncols;
        } else {
            s = c / pl.chunks_per_seq;
            q0 = (c - s * pl.chunks_per_seq) * BKQ;
        }
        // a flat chunk that crosses a sequence boundary takes the per-element path
        const bool straddle = pl.flat && (q0 + pl.SPANW > p.Ncols);
        // ---- G slab.  Raw prefetch only.
        if (X6 || (pl.gvec && !straddle)) {   // (x6 launches are float4-staged by construction: host-checked)
            // Ncols % 4 == 0: a float4 is all-valid or all-out; out-of-range ones re-read the row's last
            // valid float4 and are zeroed in store_piece
            gkeep_next = q0 + k4 < p.Ncols;
            const float* gb = p.g + ((unsigned)(s * p.g_ctot * p.Tg) + (unsigned)min(q0 + k4, p.Ncols - 4));
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(gb + goff[i]);
                areg[4 * i + 0] = v.x; areg[4 * i + 1] = v.y; areg[4 * i + 2] = v.z; areg[4 * i + 3] = v.w;
            }
        } else {
            const int kq = tid & 31;
            int sg = s, qg = q0 + kq;
            bool ok = qg < p.Ncols;
            if (pl.flat) {
                const int n = c * BKQ + kq;
                ok = n < ntot;
                sg = (int)div_magic((unsigned)n, pl.ncols_magic);
                qg = n - sg * p.Ncols;
                if (qg < 0) { --sg; qg += p.Ncols; }
            }
            const float* grow = p.g + ((size_t)sg * p.g_ctot + p.g_coff) * (size_t)p.Tg + qg;
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                const int m = m0 + (tid >> 5) + 8 * i;
                areg[i] = (ok && m < p.M) ? grow[(size_t)m * p.Tg] : 0.f;   // raw prefetch
            }
        }
        // ---- Z spans: NC rows of SPANW floats.  Two paths only: (fast) the whole span is real
        // data of one sequence -> unconditional loads off a uniform base; (slow) chunk at a
        // sequence edge / crossing sequences -> per-slot padding + sequence logic.
        // (ZV: the staged row starts zshift samples in front of the span, at a 16-byte-aligned address)
        const int u0 = pl.flat ? q0 : q0 * p.stride - p.padL + (p.tapstep > 0 ? 0 : -(p.taps - 1)) - (ZV ? pl.zshift : 0);
        const bool fast = pl.flat ? (!straddle && (long)c * BKQ + pl.SPANW <= ntot) : (u0 >= 0 && u0 + pl.SPANW <= p.Tz);
        zfast_next = fast;
        if (fast) {
            const float* zb = p.z + ((size_t)s * p.z_ctot + p.z_coff + c_lo) * (size_t)p.Tz + u0;
#pragma unroll
            for (int t = 0; t < ZPT; ++t) {
                if (ZV) {
                    const float4 v = *reinterpret_cast<const float4*>(zb + zoff[t]);
                    zreg[4 * t + 0] = v.x; zreg[4 * t + 1] = v.y; zreg[4 * t + 2] = v.z; zreg[4 * t + 3] = v.w;
                } else {
                    zreg[t] = zb[zoff[t]];
                }
            }
        } else {
            zmask = 0u;
#pragma unroll
            for (int t = 0; t < ZPT; ++t) {
                const int e0 = (tid + NTHREADS * t) * ZW;
                const int cl = e0 < total ? (int)div_magic((unsigned)e0, pl.span_magic) : 0;
#pragma unroll
                for (int w = 0; w < ZW; ++w) {
                    const int e = e0 + w;
                    float v = 0.f;
                    if (e < total) {
                        const int i = e - cl * pl.SPANW;
                        int sz = s, u;
                        bool ok;
                        if (pl.flat) {
                            const int n = c * BKQ + i;
                            sz = (int)div_magic((unsigned)n, pl.ncols_magic);
                            u = n - sz * p.Ncols;
                            if (u < 0) { --sz; u += p.Ncols; }
                            ok = n < ntot;
                        } else {
                            u = u0 + i;
                            if (p.pad_mode == PASE_PAD_REFLECT) {
                                if (u < 0) u = -u;
                                if (u >= p.Tz) u = 2 * (p.Tz - 1) - u;
                            }
                            ok = u >= 0 && u < p.Tz;
                        }
                        if (ok) {
                            v = p.z[((size_t)sz * p.z_ctot + p.z_coff + c_lo + cl) * (size_t)p.Tz + u];
                            zmask |= 1u << (t * ZW + w);
                        }
                    }
                    zreg[t * ZW + w] = v;
                }
            }
        }
        gvec_next = pl.gvec && !straddle;
    };
    // The staging stores are split into NP pieces so they can be issued BETWEEN the MFMAs of the
    // current stage (a wave that has just issued four 64-cycle MFMAs has ~200 idle issue cycles):
    // piece q handles element e of an N-element list when e*NP/N == q.  part < 0 stores everything.
    constexpr int NP = 6;
    const bool do_rowsum = pl.bias_rowsum && p.dbias && ct == 0;
    // x6: the bias row sums are accumulated by the staging threads (the LDS image holds split pieces)
    float rs_v[X6 ? GS : 1];
#pragma unroll
    for (int i = 0; i < (X6 ? GS : 1); ++i) rs_v[i] = 0.f;
    auto store_piece = [&](int buf, int part) __attribute__((always_inline)) {
        // on-load transforms (g_alpha on G, affine/PReLU on Z) are applied here
        if (X6 || gvec_next) {
            const float keep = gkeep_next ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                if (part >= 0 && (i * NP) / GS != part) continue;
                const int r = (tid >> 3) + 32 * i;
                if constexpr (X6) {
                    float gv[4];
#pragma unroll
                    for (int cidx = 0; cidx < 4; ++cidx) {
                        gv[cidx] = areg[4 * i + cidx];
                        if (has_ga) gv[cidx] = gv[cidx] > 0.f ? gv[cidx] : gv[cidx] * g_al[i];
                        gv[cidx] *= keep;
                    }
                    if (do_rowsum) rs_v[i] += (gv[0] + gv[1]) + (gv[2] + gv[3]);
                    unsigned o[3][2];
                    pase_split_bf16x3_quad(gv, o);
                    // time steps k4 .. k4+3 = half (tid & 1) of k-group (tid & 7) >> 1.  Row index XOR 2 * k-group: the four
                    // k-groups of one row (4 x 6 KB apart = one bank) land in four different 16-byte slots
                    const int kgs = (tid & 7) >> 1;
                    unsigned char* dst = reinterpret_cast<unsigned char*>(&AsX[(buf * (BKQ / 8) + kgs) * 3 * BM + (r ^ (2 * kgs))]) + (tid & 1) * 8;
#pragma unroll
                    for (int pz = 0; pz < 3; ++pz)
                        *reinterpret_cast<uint2*>(dst + pz * BM * 16) = make_uint2(o[pz][0], o[pz][1]);
                } else {
#pragma unroll
                    for (int cidx = 0; cidx < 4; ++cidx) {
                        float gv = areg[4 * i + cidx];
                        if (has_ga) gv = gv > 0.f ? gv : gv * g_al[i];
                        As[buf][k4 + cidx][r] = gv * keep;
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                if (part >= 0 && (i * NP) / A_ROWS != part) continue;
                const int r = (tid >> 5) + 8 * i;
                float gv = areg[i];
                if (has_ga && m0 + r < p.M) gv = gv > 0.f ? gv : gv * p.g_alpha[m0 + r];
                if constexpr (!X6) {
                    As[buf][tid & 31][r] = gv;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < ZPT; ++t) {
            if (part >= 0 && (t * NP) / ZPT != part) continue;
            float v[ZW];
#pragma unroll
            for (int w = 0; w < ZW; ++w) v[w] = zreg[t * ZW + w];
            if (has_xf) {
                float sc, sh, al;
                if (PRE) {
                    sc = z_sc[PRE ? t : 0]; sh = z_sh[PRE ? t : 0]; al = z_al[PRE ? t : 0];
                } else {
                    const int e = min(tid + NTHREADS * t, max(total - 1, 0));
                    const int ci = min(c_lo + (int)div_magic((unsigned)e, pl.span_magic), p.Cin - 1);
                    sc = has_aff ? p.in_scale[ci] : 1.f;
                    sh = has_aff ? p.in_shift[ci] : 0.f;
                    al = has_al ? p.in_alpha[ci] : 1.f;
                }
#pragma unroll
                for (int w = 0; w < ZW; ++w) {
                    v[w] = fmaf(v[w], sc, sh);
                    v[w] = v[w] > 0.f ? v[w] : v[w] * al;
                    // padding / out-of-range samples are zeros of the ACTIVATED tensor
                    if (!zfast_next && !(zmask & (1u << (t * ZW + w)))) v[w] = 0.f;
                }
            }
            if (ZV) *reinterpret_cast<float4*>(&Zs[buf][(tid + NTHREADS * t) * 4]) = make_float4(v[0], v[ZW > 1 ? 1 : 0], v[ZW > 2 ? 2 : 0], v[ZW > 3 ? 3 : 0]);
            else Zs[buf][tid + NTHREADS * t] = v[0];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const bool row_ok0 = m0 + wm * 64 < p.M, row_ok1 = m0 + wm * 64 + 32 < p.M;
    const bool col_ok0 = j0 + wn * 64 < Nw, col_ok1 = j0 + wn * 64 + 32 < Nw;
    const bool full_tile = row_ok0 && row_ok1 && col_ok0 && col_ok1;
    float rowsum = 0.f;

    load_stage(c_begin);
    store_piece(0, -1);
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        if (c + 1 < c_end) load_stage(c + 1);
        if constexpr (X6) {
            const bool has_next = c + 1 < c_end;
            const u32x4* aB = &AsX[(cur * (BKQ / 8) + fk) * 3 * BM + wm * 64];
            const float* z0_ = &Zs[cur][boff[0] + 8 * fk * zstep];
            const float* z1_ = &Zs[cur][boff[1] + 8 * fk * zstep];
#pragma unroll
            for (int st = 0; st < BKQ / 16; ++st) {
                float xv0[8], xv1[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xv0[e] = z0_[(16 * st + e) * zstep];
                    xv1[e] = z1_[(16 * st + e) * zstep];
                }
                u32x4 fa[3][2], fb0[3], fb1[3];
                const u32x4* aL = aB + (fr ^ (2 * (2 * st + fk)));      // (row swizzle of the staging stores)
#pragma unroll
                for (int pz = 0; pz < 3; ++pz) {
                    fa[pz][0] = aL[(st * 2 * 3 + pz) * BM];
                    fa[pz][1] = aL[(st * 2 * 3 + pz) * BM + 32];
                }
                pase_split_bf16x3(xv0, fb0);
                pase_split_bf16x3(xv1, fb1);
                constexpr int PZA[6] = {1, 0, 2, 0, 1, 0}, PZB[6] = {1, 2, 0, 1, 0, 0};   // smallest terms first
#pragma unroll
                for (int pi = 0; pi < 6; ++pi) {
                    acc[0][0] = pase_mfma_bf16_32x32x16(fa[PZA[pi]][0], fb0[PZB[pi]], acc[0][0]);
                    acc[0][1] = pase_mfma_bf16_32x32x16(fa[PZA[pi]][0], fb1[PZB[pi]], acc[0][1]);
                    acc[1][0] = pase_mfma_bf16_32x32x16(fa[PZA[pi]][1], fb0[PZB[pi]], acc[1][0]);
                    acc[1][1] = pase_mfma_bf16_32x32x16(fa[PZA[pi]][1], fb1[PZB[pi]], acc[1][1]);
                }
                PASE_SCHED_BARRIER();     // keep one step's fragments live at a time
                if (has_next) {
#pragma unroll
                    for (int q = 0; q < NP / 2; ++q) store_piece(cur ^ 1, st * (NP / 2) + q);
                }
                PASE_SCHED_BARRIER();
            }
        } else
        // software-pipelined operand fetch (ping-pong registers, schedule pinned): the ds_reads of step
        // ks+1 are in flight under the four MFMAs of step ks.  MFMAs are unconditional (see conv_gemm.hip).
        {
            const float* as_ = &As[cur][fk][wm * 64 + fr];
            const float* z0_ = &Zs[cur][boff[0] + fk * zstep];
            const float* z1_ = &Zs[cur][boff[1] + fk * zstep];
            auto fetch = [&](int ks, float& a0, float& a1, float& b0, float& b1) __attribute__((always_inline)) {
                const int kc = min(ks, BKQ / 2 - 1);      // the one fetch past the end stays in bounds
                a0 = as_[kc * 2 * (BM + 1)];
                a1 = as_[kc * 2 * (BM + 1) + 32];
                b0 = z0_[kc * 2 * zstep];
                b1 = z1_[kc * 2 * zstep];
            };
            float pa0, pa1, pb0, pb1, qa0, qa1, qb0, qb1;
            fetch(0, pa0, pa1, pb0, pb1);
            const bool has_next = c + 1 < c_end;
#pragma unroll
            for (int it = 0; it < BKQ / 4; ++it) {       // 8 iterations x 2 k-steps, fully unrolled
                const int ks = it * 2;
                fetch(ks + 1, qa0, qa1, qb0, qb1);
                PASE_SCHED_BARRIER();
                acc[0][0] = pase_mfma_32x32x2(pa0, pb0, acc[0][0]);
                acc[0][1] = pase_mfma_32x32x2(pa0, pb1, acc[0][1]);
                acc[1][0] = pase_mfma_32x32x2(pa1, pb0, acc[1][0]);
                acc[1][1] = pase_mfma_32x32x2(pa1, pb1, acc[1][1]);
                PASE_SCHED_BARRIER();
                fetch(ks + 2, pa0, pa1, pb0, pb1);
                PASE_SCHED_BARRIER();
                acc[0][0] = pase_mfma_32x32x2(qa0, qb0, acc[0][0]);
                acc[0][1] = pase_mfma_32x32x2(qa0, qb1, acc[0][1]);
                acc[1][0] = pase_mfma_32x32x2(qa1, qb0, acc[1][0]);
                acc[1][1] = pase_mfma_32x32x2(qa1, qb1, acc[1][1]);
                PASE_SCHED_BARRIER();
                // next stage's staging stores ride in the issue slots behind these MFMAs (the
                // prefetch was issued >= 4 k-steps = 1000+ cycles ago)
                if (it >= BKQ / 4 - NP && has_next) {
                    store_piece(cur ^ 1, it - (BKQ / 4 - NP));
                    PASE_SCHED_BARRIER();
                }
            }
        }
        if (!X6 && do_rowsum && tid < BM) {
#pragma unroll 8
            for (int kq = 0; kq < BKQ; ++kq) rowsum += As[cur][kq][tid];
        }
        __syncthreads();
    }
    if (!X6 && do_rowsum && tid < BM && m0 + tid < p.M) atomicAdd(p.dbias + m0 + tid, rowsum);
    if (X6 && do_rowsum) {
        // staging-thread partials: 8 lanes share a row
#pragma unroll
        for (int i = 0; i < (X6 ? GS : 0); ++i) {
            float v = rs_v[i];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            const int m = m0 + (tid >> 3) + 32 * i;
            if ((tid & 7) == 0 && m < p.M && v != 0.f) atomicAdd(p.dbias + m, v);
        }
    }

    const int rbase = 4 * (lane >> 5);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + rbase;
            if (m >= p.M) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int j = j0 + wn * 64 + b * 32 + fr;
                const float v = acc[a][b][r];
                if (j < Kw) atomicAdd(p.dw + (size_t)m * p.ldw + j, v);
                else if (j < Nw) atomicAdd(p.dbias + m, v);
            }
        }
    }
}


// ---- 1x1 layers (taps == 1): dW[m, j] += sum_n G[m, n] * act(bn(Z[j, n])), n = (s, q) flattened -- a plain
// "NT" GEMM whose two operands are both contiguous along the reduction.  Dedicated kernel: every thread owns
// fixed G rows / Z channels for the whole tile (float4 along time), so the per-row PReLU slope and the
// per-channel affine are loaded once per tile, a stage costs one magic division + 8 unconditional
// global_load_dwordx4 per thread, and both slabs sit k-major in LDS (pitch BM+1 / BN+1: conflict-free
// fragment reads).  ~140 VGPRs -> 2 workgroups per CU (the generic kernel needs the 1-wave/SIMD variant here).
struct alignas(16) WF4 { float x, y, z, w; };

template <int BM, int BN>
__global__ void __launch_bounds__(NTHREADS, 2) wgrad_flat_kernel(PaseWgrad p, WgradPlan pl) {
    constexpr int WAVES_N = BN / 64;
    constexpr int GS = BM / 32;      // G float4 slots per thread (8 threads x float4 = 32 time steps per row)
    constexpr int ZSL = BN / 32;     // Z float4 slots per thread
    __shared__ float As[2][BKQ][BM + 1];
    __shared__ float Zs[2][BKQ][BN + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = pase_uniform(tid >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int fr = lane & 31;
    const int fk = lane >> 5;

    const int tiles = pl.n_row_tiles * pl.n_col_tiles;
    const int tile = blockIdx.x % tiles;
    const int split = blockIdx.x / tiles;
    const int mt = tile % pl.n_row_tiles;
    const int ct = tile / pl.n_row_tiles;
    const int m0 = mt * BM;
    const int j0 = ct * BN;
    const int Kw = p.Cin;
    const int c_begin = split * pl.kt_per_split;
    const int c_end = min(pl.n_chunks, c_begin + pl.kt_per_split);
    if (c_begin >= c_end) return;
    const int ntot = p.S * p.Ncols;

    // thread -> rows (tid >> 3) + 32*i, time steps k4 .. k4+3 of the chunk
    const int k4 = (tid & 7) * 4;
    const int r0 = tid >> 3;
    unsigned goff[GS], zoff[ZSL];
    float g_al[GS], z_sc[ZSL], z_sh[ZSL], z_al[ZSL];
    const bool has_ga = p.g_alpha != nullptr;
    const bool has_aff = p.in_scale != nullptr, has_al = p.in_alpha != nullptr;
#pragma unroll
    for (int i = 0; i < GS; ++i) {
        const int m = min(m0 + r0 + 32 * i, p.M - 1);          // rows past M re-read row M-1 (never stored)
        goff[i] = (unsigned)((p.g_coff + m) * p.Tg);
        g_al[i] = has_ga ? p.g_alpha[m] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < ZSL; ++i) {
        const int j = min(j0 + r0 + 32 * i, Kw - 1);
        zoff[i] = (unsigned)((p.z_coff + j) * p.Tz);
        z_sc[i] = has_aff ? p.in_scale[j] : 1.f;
        z_sh[i] = has_aff ? p.in_shift[j] : 0.f;
        z_al[i] = has_al ? p.in_alpha[j] : 1.f;
    }
    WF4 areg[GS], zreg[ZSL];
    bool valid_next = true;

    auto load_stage = [&](int c) __attribute__((always_inline)) {
        const int n = c * BKQ + k4;                              // 4 consecutive n share a sequence (Ncols % 4 == 0)
        valid_next = n < ntot;
        const unsigned nn = (unsigned)min(n, ntot - 4);
        int s = (int)div_magic(nn, pl.ncols_magic);
        int q = (int)nn - s * p.Ncols;
        if (q < 0) { --s; q += p.Ncols; }
        const float* gb = p.g + (unsigned)(s * p.g_ctot * p.Tg + q);
        const float* zb = p.z + (unsigned)(s * p.z_ctot * p.Tz + q);
#pragma unroll
        for (int i = 0; i < GS; ++i) areg[i] = *reinterpret_cast<const WF4*>(gb + goff[i]);
#pragma unroll
        for (int i = 0; i < ZSL; ++i) zreg[i] = *reinterpret_cast<const WF4*>(zb + zoff[i]);
    };
    auto prelu = [&](float v, float al) __attribute__((always_inline)) { return v > 0.f ? v : v * al; };
    auto store_stage = [&](int buf) __attribute__((always_inline)) {
        const float keep = valid_next ? 1.f : 0.f;               // chunk tail beyond S*Ncols contributes zero
#pragma unroll
        for (int i = 0; i < GS; ++i) {
            const int r = r0 + 32 * i;
            WF4 v = areg[i];
            if (has_ga) { v.x = prelu(v.x, g_al[i]); v.y = prelu(v.y, g_al[i]); v.z = prelu(v.z, g_al[i]); v.w = prelu(v.w, g_al[i]); }
            As[buf][k4 + 0][r] = v.x * keep;
            As[buf][k4 + 1][r] = v.y * keep;
            As[buf][k4 + 2][r] = v.z * keep;
            As[buf][k4 + 3][r] = v.w * keep;
        }
#pragma unroll
        for (int i = 0; i < ZSL; ++i) {
            const int r = r0 + 32 * i;
            WF4 v = zreg[i];
            if (has_aff || has_al) {
                v.x = prelu(fmaf(v.x, z_sc[i], z_sh[i]), z_al[i]); v.y = prelu(fmaf(v.y, z_sc[i], z_sh[i]), z_al[i]);
                v.z = prelu(fmaf(v.z, z_sc[i], z_sh[i]), z_al[i]); v.w = prelu(fmaf(v.w, z_sc[i], z_sh[i]), z_al[i]);
            }
            Zs[buf][k4 + 0][r] = v.x;
            Zs[buf][k4 + 1][r] = v.y;
            Zs[buf][k4 + 2][r] = v.z;
            Zs[buf][k4 + 3][r] = v.w;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool do_rowsum = p.dbias && ct == 0;
    float rowsum = 0.f;

    load_stage(c_begin);
    store_stage(0);
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        const bool has_next = c + 1 < c_end;
        if (has_next) load_stage(c + 1);
        const float* as_ = &As[cur][fk][wm * 64 + fr];
        const float* zs_ = &Zs[cur][fk][wn * 64 + fr];
        auto fetch = [&](int ks, float& a0, float& a1, float& b0, float& b1) __attribute__((always_inline)) {
            const int kc = min(ks, BKQ / 2 - 1);      // the one fetch past the end stays in bounds
            a0 = as_[kc * 2 * (BM + 1)];
            a1 = as_[kc * 2 * (BM + 1) + 32];
            b0 = zs_[kc * 2 * (BN + 1)];
            b1 = zs_[kc * 2 * (BN + 1) + 32];
        };
        float pa0, pa1, pb0, pb1, qa0, qa1, qb0, qb1;
        fetch(0, pa0, pa1, pb0, pb1);
#pragma unroll
        for (int it = 0; it < BKQ / 4; ++it) {       // 8 iterations x 2 k-steps, fully unrolled
            const int ks = it * 2;
            fetch(ks + 1, qa0, qa1, qb0, qb1);
            PASE_SCHED_BARRIER();
            acc[0][0] = pase_mfma_32x32x2(pa0, pb0, acc[0][0]);
            acc[0][1] = pase_mfma_32x32x2(pa0, pb1, acc[0][1]);
            acc[1][0] = pase_mfma_32x32x2(pa1, pb0, acc[1][0]);
            acc[1][1] = pase_mfma_32x32x2(pa1, pb1, acc[1][1]);
            PASE_SCHED_BARRIER();
            fetch(ks + 2, pa0, pa1, pb0, pb1);
            PASE_SCHED_BARRIER();
            acc[0][0] = pase_mfma_32x32x2(qa0, qb0, acc[0][0]);
            acc[0][1] = pase_mfma_32x32x2(qa0, qb1, acc[0][1]);
            acc[1][0] = pase_mfma_32x32x2(qa1, qb0, acc[1][0]);
            acc[1][1] = pase_mfma_32x32x2(qa1, qb1, acc[1][1]);
            PASE_SCHED_BARRIER();
        }
        if (do_rowsum && tid < BM) {
#pragma unroll 8
            for (int kq = 0; kq < BKQ; ++kq) rowsum += As[cur][kq][tid];
        }
        if (has_next) store_stage(cur ^ 1);
        __syncthreads();
    }
    if (do_rowsum && tid < BM && m0 + tid < p.M) atomicAdd(p.dbias + m0 + tid, rowsum);

    const int rbase = m0 + wm * 64 + 4 * (lane >> 5);
    const int jb = j0 + wn * 64 + fr;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
            if (m >= p.M) continue;
            const unsigned rowoff = (unsigned)(m * p.ldw);
            if (jb < Kw) atomicAdd(p.dw + (rowoff + (unsigned)jb), acc[a][0][r]);
            if (jb + 32 < Kw) atomicAdd(p.dw + (rowoff + (unsigned)(jb + 32)), acc[a][1][r]);
        }
    }
}

}  // namespace

extern "C" long pase_wgrad_x6_bytes(const PaseWgrad* d) {
    if (d->M <= 0 || d->Cin <= 0 || d->S <= 0 || d->Ncols <= 0) return 0;
    if (!(d->x6 & 1)) return 0;          // bit 0 enables the split-bf16 path; the bits above it are controls only
    PaseSincPlan sp;
    if (!(d->x6 & 1024) && pase_sinc_x6_wgrad_plan(*d, sp)) return sp.pack_bytes;
    PaseX6cWgrad o;
    return pase_x6c_wgrad_plan(*d, o) ? o.pl.pack_bytes : 0;
}

extern "C" int pase_wgrad_plan_kind(const PaseWgrad* d) {
    if (d->M <= 0 || d->Cin <= 0 || d->S <= 0 || d->Ncols <= 0) return 0;
    if (!(d->x6 & 1)) return 0;
    PaseSincPlan sp;
    if (!(d->x6 & 1024) && pase_sinc_x6_wgrad_plan(*d, sp)) return 5;
    PaseX6cWgrad o;
    return pase_x6c_wgrad_plan(*d, o) ? (o.pl.zp ? (o.pl.WM == 8 ? 7 : 4) : o.pl.tmode) : 0;
}

// 0 = accepted; the refusal code otherwise (shared by the launch and the query so that the two cannot disagree)
static int wgrad_act_bwd_refusal(const PaseWgrad& p, const PaseActBwd* g_bwd, PaseSincPlan& sp) {
    if (p.pad_mode == PASE_PAD_REFLECT && p.padL >= p.Tz) return -3;
    if (!(p.x6 & 1) || !p.gx6 || (p.x6 & 1024) || !pase_sinc_x6_wgrad_plan(p, sp)) return -11;      // only the one-channel plan
    if (!g_bwd || !pase_sinc_x6_wgrad_act_bwd_ok(p, *g_bwd)) return -13;
    return 0;
}

extern "C" int pase_wgrad_gemm_act_bwd_ok(const PaseWgrad* d, const PaseActBwd* g_bwd) {
    if (!d || d->M <= 0 || d->Cin <= 0 || d->S <= 0 || d->Ncols <= 0) return 0;
    PaseSincPlan sp;
    return wgrad_act_bwd_refusal(*d, g_bwd, sp) == 0 ? 1 : 0;
}

extern "C" int pase_wgrad_gemm_act_bwd(const PaseWgrad* d, const PaseActBwd* g_bwd, void* stream) {
    PaseWgrad p = *d;
    if (p.M <= 0 || p.Cin <= 0 || p.S <= 0 || p.Ncols <= 0) return 0;
    PaseSincPlan sp;
    const int rc = wgrad_act_bwd_refusal(p, g_bwd, sp);
    if (rc) return rc;
    return pase_sinc_x6_wgrad_launch(p, sp, (hipStream_t)stream, g_bwd);
}

extern "C" int pase_wgrad_gemm(const PaseWgrad* d, void* stream) {
    PaseWgrad p = *d;
    if (p.M <= 0 || p.Cin <= 0 || p.S <= 0 || p.Ncols <= 0) return 0;
    // argument checks shared by every kernel family (the split-bf16 paths reflect once: an out-of-range pad would contribute
    // zeros there instead of an error)
    if (p.tapstep != 1 && p.tapstep != -1) return -5;
    if (p.pad_mode == PASE_PAD_REFLECT && p.padL >= p.Tz) return -3;
    if ((p.x6 & 1) && p.gx6) {
        if ((((unsigned long long)(size_t)p.gx6) % 16) != 0) return -10;
        PaseSincPlan sp;
        if (!(p.x6 & 1024) && pase_sinc_x6_wgrad_plan(p, sp)) return pase_sinc_x6_wgrad_launch(p, sp, (hipStream_t)stream);
        PaseX6cWgrad o;
        if (!pase_x6c_wgrad_plan(p, o)) return -11;        // a pack buffer on a launch without a plan is refused, not ignored
        return pase_x6c_wgrad_launch(p, o, (hipStream_t)stream);
    }
    // from here on: the exact-fp32 matrix pipe (round 2's single-accumulator split-bf16 instantiations of the kernels below
    // were biased -- conv_x6c.hip header -- and are no longer built)
    p.x6 = 0;
    if (p.tap_major) return -4;          // express tap-major weights as per-tap launches (ldw + offset)
    if (p.tapstep != 1 && p.tapstep != -1) return -5;
    if (p.pad_mode == PASE_PAD_REFLECT && p.padL >= p.Tz) return -3;
    if ((long)p.S * p.Ncols >= 0x7fffffffL) return -8;
    hipStream_t st = (hipStream_t)stream;
    bool narrow = p.M <= 64;
    WgradPlan pl;
    pl.flat = (p.taps == 1 && p.stride == 1 && p.padL == 0 && p.tapstep == 1) ? 1 : 0;
    // dedicated 1x1 kernel: float4 along time on both operands, 32-bit element offsets
    const bool flat_fast = pl.flat && (p.Ncols % 4) == 0 && (p.Tg % 4) == 0 && (p.Tz % 4) == 0 &&
                           (((unsigned long long)(size_t)p.g) % 16) == 0 && (((unsigned long long)(size_t)p.z) % 16) == 0 &&
                           (long)p.S * p.g_ctot * (long)p.Tg < 0x7fffffffL && (long)p.S * p.z_ctot * (long)p.Tz < 0x7fffffffL &&
                           (long)p.M * p.ldw < 0x7fffffffL && (long)p.S * p.Ncols >= 4;
    if (flat_fast) narrow = false;
    // flat (1x1): one extra (unused) sample per channel row makes the LDS row pitch odd -- the 32 lanes of a
    // B fragment read 32 different rows at the same k, which with a pitch of 32 is one bank
    pl.SPANW = pl.flat ? BKQ + 1 : (BKQ - 1) * p.stride + p.taps;
    auto need = [&](int bn) {
        long max_nc = (bn - 1) / p.taps + 2;     // channels a tile of bn (ci,kk) columns can touch
        if (max_nc > p.Cin) max_nc = p.Cin;
        return max_nc * pl.SPANW;
    };
    // 16-byte span staging for the large-span case: chunk starts are multiples of 32*stride samples, so the span start
    // has a fixed residue mod 4; stage from the aligned address zshift samples earlier, rows padded to whole float4s
    constexpr int ZV_SLOTS = 5;
    const int SPANW_raw = pl.SPANW;
    const int u0res = -p.padL - (p.tapstep > 0 ? 0 : p.taps - 1);
    const int zshift = ((u0res % 4) + 4) % 4;
    const int SPANW_zv = (SPANW_raw + zshift + 3) / 4 * 4;
    const bool zv_ok = !pl.flat && ((BKQ * p.stride) % 4) == 0 && (p.Tz % 4) == 0 &&
                       (((unsigned long long)(size_t)p.z) % 16) == 0;
    auto need_zv = [&](int bn) {
        long max_nc = (bn - 1) / p.taps + 2;
        if (max_nc > p.Cin) max_nc = p.Cin;
        return max_nc * SPANW_zv;
    };
    pl.zshift = 0;
    bool use_zv = false;
    if (narrow && need(256) > ZPT_LARGE * NTHREADS && !(zv_ok && need_zv(256) <= ZV_SLOTS * 4 * NTHREADS)) narrow = false;
    if (!narrow && need(128) > ZPT_LARGE * NTHREADS && !(zv_ok && need_zv(128) <= ZV_SLOTS * 4 * NTHREADS)) return -6;
    const bool small = need(narrow ? 256 : 128) <= ZPT_SMALL * NTHREADS;
    if (!small && zv_ok && need_zv(narrow ? 256 : 128) <= ZV_SLOTS * 4 * NTHREADS) {
        use_zv = true;
        pl.zshift = zshift;
        pl.SPANW = SPANW_zv;
    }
    const int BMv = narrow ? 64 : 128, BNv = narrow ? 256 : 128;
    if ((BKQ - 1) * (pl.flat ? 1 : p.stride) + 1 > ZS_ONES) return -6;
    pl.bias_rowsum = (p.dbias && (flat_fast || ((p.Cin * p.taps) % BNv) == 0)) ? 1 : 0;
    const int Nw = p.Cin * p.taps + ((p.dbias && !pl.bias_rowsum) ? 1 : 0);
    pl.n_row_tiles = (p.M + BMv - 1) / BMv;
    pl.n_col_tiles = (Nw + BNv - 1) / BNv;
    const long kred = (long)p.S * p.Ncols;
    pl.chunks_per_seq = (p.Ncols + BKQ - 1) / BKQ;
    const int bkq = BKQ;                                 // reduction positions per stage
    pl.n_chunks = pl.flat ? (int)((kred + bkq - 1) / bkq) : p.S * pl.chunks_per_seq;
    pl.span_magic = (unsigned)((0x100000000ULL + pl.SPANW - 1) / (unsigned long long)pl.SPANW);
    pl.ncols_magic = (unsigned)((0x100000000ULL + p.Ncols - 1) / (unsigned long long)p.Ncols);
    pl.gvec = ((p.Tg % 4) == 0 && (p.Ncols % 4) == 0 && (((unsigned long long)(size_t)p.g) % 16) == 0 &&
               (long)p.S * p.g_ctot * (long)p.Tg < 0x7fffffffL) ? 1 : 0;
    const int tiles = pl.n_row_tiles * pl.n_col_tiles;
    int splitk = p.splitk;
    if (splitk <= 0) {
        // 512 workgroup slots (2 per CU).  Workgroups of one launch cost the same, so the grid runs in rounds
        // of 512; a nearly empty last round costs as much as a half-full one (measured: a lone workgroup on
        // a CU runs 1.85x faster than two co-resident ones).  Pick the split that minimises
        // rounds x (reduction share + atomic tile flush) per workgroup.
        const int min_stages = 4 * BKQ / bkq;             // at least 128 reduction positions per split
        const int max_split = (pl.n_chunks + min_stages - 1) / min_stages;
        double best = 1e30;
        splitk = 1;
        const double flush = 3.0 * (BKQ / bkq) / (double)pl.n_chunks;   // atomic tile flush ~ 3 (32-deep) stages of work
        for (int sk = 1; sk <= max_split && sk <= 2048; ++sk) {
            const long W = (long)tiles * sk;
            const long slots = 512;
            const long full = W / slots, tail = W % slots;
            const double tc = tail == 0 ? 0.0 : (tail <= 256 ? 0.55 : 1.0);
            const double est = ((double)full + tc) * (1.0 / sk + flush);   // rounds x (work + flush) per workgroup
            if (est < best - 1e-12) { best = est; splitk = sk; }
        }
    }
    if (splitk > pl.n_chunks) splitk = pl.n_chunks;
    pl.kt_per_split = (pl.n_chunks + splitk - 1) / splitk;
    splitk = (pl.n_chunks + pl.kt_per_split - 1) / pl.kt_per_split;
    const dim3 grid((unsigned)(tiles * splitk)), block(NTHREADS);
    if (flat_fast) {
        PASE_LAUNCH((wgrad_flat_kernel<128, 128>), grid, block, st, p, pl);
        PASE_CHECK_LAUNCH();
        return 0;
    }
    if (use_zv && narrow)      PASE_LAUNCH((wgrad_gemm_kernel<64, 256, 5, 1>), grid, block, st, p, pl);
    else if (use_zv)           PASE_LAUNCH((wgrad_gemm_kernel<128, 128, 5, 1>), grid, block, st, p, pl);
    else if (narrow && small)  PASE_LAUNCH((wgrad_gemm_kernel<64, 256, ZPT_SMALL>), grid, block, st, p, pl);
    else if (narrow)           PASE_LAUNCH((wgrad_gemm_kernel<64, 256, ZPT_LARGE>), grid, block, st, p, pl);
    else if (small)            PASE_LAUNCH((wgrad_gemm_kernel<128, 128, ZPT_SMALL>), grid, block, st, p, pl);
    else                       PASE_LAUNCH((wgrad_gemm_kernel<128, 128, ZPT_LARGE>), grid, block, st, p, pl);
    PASE_CHECK_LAUNCH();
    return 0;
}

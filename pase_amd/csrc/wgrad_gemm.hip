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
    int gvec, flat, SPANW, chunks_per_seq, n_chunks, kt_per_split, n_row_tiles, n_col_tiles;
    unsigned span_magic, ncols_magic;
};

__device__ __forceinline__ unsigned div_magic(unsigned e, unsigned magic) {
    // e / d with magic = ceil(2^32 / d); magic == 0 encodes d == 1 (2^32 does not fit)
    return magic ? (unsigned)(((unsigned long long)e * magic) >> 32) : e;
}

template <int BM, int BN, int ZPT>
__global__ void __launch_bounds__(NTHREADS, (ZPT <= ZPT_SMALL ? 2 : 1)) wgrad_gemm_kernel(PaseWgrad p, WgradPlan pl) {
    constexpr int ZS_DATA = ZPT * NTHREADS;
    constexpr int ZS_TOTAL = ZS_DATA + ZS_ONES;
    constexpr int WAVES_N = BN / 64;
    constexpr int A_ROWS = BM / 8;
    __shared__ float As[2][BKQ][BM + 1];
    __shared__ float Zs[2][ZS_TOTAL];

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
            boff[b] = (ci - c_lo) * pl.SPANW + (p.tapstep > 0 ? kk : p.taps - 1 - kk);
        } else {
            boff[b] = ZS_DATA;      // ones (bias column when j == Kw, discarded otherwise)
        }
    }
    const int zstep = pl.flat ? 1 : p.stride;
    bool gvec_next = false;
    for (int i = tid; i < ZS_ONES; i += NTHREADS) { Zs[0][ZS_DATA + i] = 1.f; Zs[1][ZS_DATA + i] = 1.f; }

    float areg[A_ROWS];
    float zreg[ZPT];
    unsigned zmask = 0u;
    const int ntot = p.S * p.Ncols;
    const int total = NC * pl.SPANW;
    const int zrow_skip = p.Tz - pl.SPANW;   // slot e -> element offset e + cl * (Tz - SPANW) from the span start
    int lic = 0;   // laundered zero, refreshed every stage: keeps the per-slot index math INSIDE the
                   // stage (hoisted out of the loop it would pin ~40 VGPRs for the whole kernel)
    auto zrel = [&](int t) __attribute__((always_inline)) {
        const int e = tid + NTHREADS * t + lic;
        return e + (int)div_magic((unsigned)e, pl.span_magic) * zrow_skip;
    };

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
        PASE_LAUNDER(lic);
        // a flat chunk that crosses a sequence boundary takes the per-element path
        const bool straddle = pl.flat && (q0 + BKQ > p.Ncols);
        // ---- G slab [BM x 32]: lanes along time.  Raw prefetch only.
        if (pl.gvec && !straddle) {
            const int k4 = (tid & 7) * 4;
            const bool ok = q0 + k4 < p.Ncols;       // Ncols % 4 == 0: a float4 is all-valid or all-out
            const float* grow = p.g + ((size_t)s * p.g_ctot + p.g_coff) * (size_t)p.Tg + q0 + k4;
#pragma unroll
            for (int i = 0; i < A_ROWS / 4; ++i) {
                const int m = m0 + (tid >> 3) + 32 * i;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok && m < p.M) v = *reinterpret_cast<const float4*>(grow + (size_t)m * p.Tg);
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
        // data of one sequence -> slot address = span start + e + cl*(Tz - SPANW); (slow) chunk at a
        // sequence edge / crossing sequences -> per-slot padding + sequence logic.
        zmask = 0u;
        const int u0 = pl.flat ? q0 : q0 * p.stride - p.padL + (p.tapstep > 0 ? 0 : -(p.taps - 1));
        const bool fast = pl.flat ? (!straddle && (long)c * BKQ + BKQ <= ntot) : (u0 >= 0 && u0 + pl.SPANW <= p.Tz);
        if (fast) {
            const float* zb = p.z + ((size_t)s * p.z_ctot + p.z_coff + c_lo) * (size_t)p.Tz + u0;
#pragma unroll
            for (int t = 0; t < ZPT; ++t) {
                const bool ok = (tid + NTHREADS * t) < total;
                zreg[t] = ok ? zb[zrel(t)] : 0.f;
                if (ok) zmask |= 1u << t;
            }
        } else {
#pragma unroll
            for (int t = 0; t < ZPT; ++t) {
                const int e = tid + NTHREADS * t + lic;
                float v = 0.f;
                if (e < total) {
                    const int cl = (int)div_magic((unsigned)e, pl.span_magic);
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
                        zmask |= 1u << t;
                    }
                }
                zreg[t] = v;
            }
        }
        gvec_next = pl.gvec && !straddle;
    };
    // The staging stores are split into NP pieces so they can be issued BETWEEN the MFMAs of the
    // current stage (a wave that has just issued four 64-cycle MFMAs has ~200 idle issue cycles):
    // piece q handles element e of an N-element list when e*NP/N == q.  part < 0 stores everything.
    constexpr int NP = 6;
    auto store_piece = [&](int buf, int part) __attribute__((always_inline)) {
        // on-load transforms (g_alpha on G, affine/PReLU on Z) are applied here
        if (gvec_next) {
            const int k4 = (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < A_ROWS / 4; ++i) {
                if (part >= 0 && (i * NP) / (A_ROWS / 4) != part) continue;
                const int r = (tid >> 3) + 32 * i;
                float al = 1.f;
                if (p.g_alpha && m0 + r < p.M) al = p.g_alpha[m0 + r];
#pragma unroll
                for (int cidx = 0; cidx < 4; ++cidx) {
                    float gv = areg[4 * i + cidx];
                    if (p.g_alpha) gv = gv > 0.f ? gv : gv * al;
                    As[buf][k4 + cidx][r] = gv;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                if (part >= 0 && (i * NP) / A_ROWS != part) continue;
                const int r = (tid >> 5) + 8 * i;
                float gv = areg[i];
                if (p.g_alpha && m0 + r < p.M) gv = gv > 0.f ? gv : gv * p.g_alpha[m0 + r];
                As[buf][tid & 31][r] = gv;
            }
        }
#pragma unroll
        for (int t = 0; t < ZPT; ++t) {
            if (part >= 0 && (t * NP) / ZPT != part) continue;
            float v = zreg[t];
            if ((p.in_scale || p.in_alpha) && (zmask & (1u << t))) {
                const int ci = c_lo + (int)div_magic((unsigned)(tid + NTHREADS * t), pl.span_magic);
                if (p.in_scale) v = v * p.in_scale[ci] + p.in_shift[ci];
                if (p.in_alpha) v = v > 0.f ? v : v * p.in_alpha[ci];
            }
            Zs[buf][tid + NTHREADS * t] = v;
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
    const bool do_rowsum = pl.bias_rowsum && p.dbias && ct == 0;
    float rowsum = 0.f;

    load_stage(c_begin);
    store_piece(0, -1);
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        if (c + 1 < c_end) load_stage(c + 1);
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
        if (do_rowsum && tid < BM) {
#pragma unroll 8
            for (int kq = 0; kq < BKQ; ++kq) rowsum += As[cur][kq][tid];
        }
        __syncthreads();
    }
    if (do_rowsum && tid < BM && m0 + tid < p.M) atomicAdd(p.dbias + m0 + tid, rowsum);

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

}  // namespace

extern "C" int pase_wgrad_gemm(const PaseWgrad* d, void* stream) {
    const PaseWgrad p = *d;
    if (p.M <= 0 || p.Cin <= 0 || p.S <= 0 || p.Ncols <= 0) return 0;
    if (p.tap_major) return -4;          // express tap-major weights as per-tap launches (ldw + offset)
    if (p.tapstep != 1 && p.tapstep != -1) return -5;
    if (p.pad_mode == PASE_PAD_REFLECT && p.padL >= p.Tz) return -3;
    if ((long)p.S * p.Ncols >= 0x7fffffffL) return -8;
    hipStream_t st = (hipStream_t)stream;
    bool narrow = p.M <= 64;
    WgradPlan pl;
    pl.flat = (p.taps == 1 && p.stride == 1 && p.padL == 0 && p.tapstep == 1) ? 1 : 0;
    pl.SPANW = pl.flat ? BKQ : (BKQ - 1) * p.stride + p.taps;
    auto need = [&](int bn) {
        long max_nc = (bn - 1) / p.taps + 2;     // channels a tile of bn (ci,kk) columns can touch
        if (max_nc > p.Cin) max_nc = p.Cin;
        return max_nc * pl.SPANW;
    };
    if (narrow && need(256) > ZPT_LARGE * NTHREADS) narrow = false;
    if (!narrow && need(128) > ZPT_LARGE * NTHREADS) return -6;
    const bool small = need(narrow ? 256 : 128) <= ZPT_SMALL * NTHREADS;
    const int BMv = narrow ? 64 : 128, BNv = narrow ? 256 : 128;
    if ((BKQ - 1) * (pl.flat ? 1 : p.stride) + 1 > ZS_ONES) return -6;
    pl.bias_rowsum = (p.dbias && ((p.Cin * p.taps) % BNv) == 0) ? 1 : 0;
    const int Nw = p.Cin * p.taps + ((p.dbias && !pl.bias_rowsum) ? 1 : 0);
    pl.n_row_tiles = (p.M + BMv - 1) / BMv;
    pl.n_col_tiles = (Nw + BNv - 1) / BNv;
    const long kred = (long)p.S * p.Ncols;
    pl.chunks_per_seq = (p.Ncols + BKQ - 1) / BKQ;
    pl.n_chunks = pl.flat ? (int)((kred + BKQ - 1) / BKQ) : p.S * pl.chunks_per_seq;
    pl.span_magic = (unsigned)((0x100000000ULL + pl.SPANW - 1) / (unsigned long long)pl.SPANW);
    pl.ncols_magic = (unsigned)((0x100000000ULL + p.Ncols - 1) / (unsigned long long)p.Ncols);
    pl.gvec = ((p.Tg % 4) == 0 && (p.Ncols % 4) == 0 && (((unsigned long long)(size_t)p.g) % 16) == 0) ? 1 : 0;
    const int tiles = pl.n_row_tiles * pl.n_col_tiles;
    int splitk = p.splitk;
    if (splitk <= 0) {
        splitk = (1024 + tiles - 1) / tiles;              // ~2 resident waves of workgroups (2 WG/CU)
        const int max_split = (pl.n_chunks + 3) / 4;      // at least 4 stages (256 MFMAs/wave) per split
        if (splitk > max_split) splitk = max_split;
        if (splitk < 1) splitk = 1;
    }
    if (splitk > pl.n_chunks) splitk = pl.n_chunks;
    pl.kt_per_split = (pl.n_chunks + splitk - 1) / splitk;
    splitk = (pl.n_chunks + pl.kt_per_split - 1) / pl.kt_per_split;
    const dim3 grid((unsigned)(tiles * splitk)), block(NTHREADS);
    if (narrow && small)       PASE_LAUNCH((wgrad_gemm_kernel<64, 256, ZPT_SMALL>), grid, block, st, p, pl);
    else if (narrow)           PASE_LAUNCH((wgrad_gemm_kernel<64, 256, ZPT_LARGE>), grid, block, st, p, pl);
    else if (small)            PASE_LAUNCH((wgrad_gemm_kernel<128, 128, ZPT_SMALL>), grid, block, st, p, pl);
    else                       PASE_LAUNCH((wgrad_gemm_kernel<128, 128, ZPT_LARGE>), grid, block, st, p, pl);
    PASE_CHECK_LAUNCH();
    return 0;
}

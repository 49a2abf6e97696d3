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
//     (B, D*r, F) prediction/target pair of a regression worker is never materialised,
//   * optional split-K (few output tiles, long reduction: the data-gradients of the wide heads).
//
// gfx950 mapping.  256 threads = 4 waves; block tile BM x BN; each wave owns a 64x64 sub-tile as 2x2
// v_mfma_f32_32x32x2_f32 tiles (exact fp32, 64 accumulator VGPRs).  The input is NOT expanded to an
// im2col tile: per stage the block stages, with fully coalesced row loads, the raw sliding-window
// SPANS of CB input channels (length (BN-1)*stride + TB, affine+PReLU and padding applied once per
// element) plus the matching [BM x CB*TB] weight slab into LDS (double-buffered, register
// prefetch, one barrier per stage).  An MFMA B fragment is then a strided ds_read_b32 straight out
// of the span: column j, tap kk -> Xs[cl][j*stride + kk]; each lane walks its (channel row, tap)
// offset incrementally, so neither integer division nor a table lookup sits in the MFMA loop.
// Loads are issued as raw prefetches one stage ahead and only touched (affine / PReLU, ds_write)
// after the MFMA loop of the current stage, so HBM/L2 latency hides under the matrix pipe.
// Two shapes: <128,128> (waves 2x2) and <64,256> (waves 1x4) for the 64-row layers.
#include "hip_compat.h"
#include "pase_amd.h"

namespace {

constexpr int NTHREADS = 256;
constexpr int KGMAX = 48;     // flat k (channels x taps) per stage
constexpr int XSMAX = 3072;   // staged span floats per stage
constexpr int XPT = XSMAX / NTHREADS;

struct ConvPlan {
    int CB, TB, SPAN, n_gc, n_gt, mode, tiles_per_seq, splitk, avec;
    unsigned span_magic;      // ceil(2^32 / SPAN)
    unsigned ncols_magic;     // ceil(2^32 / Ncols)
};
// column-tile modes
//   MODE_FLAT  : 1x1, stride 1, no padding: columns are the flattened (s, q) index, a row of the
//                staged slab is just BN consecutive columns (may cross any number of sequences)
//   MODE_SEG   : general taps/stride/padding with Ncols >= BN: columns are the flattened (s, q)
//                index; a tile touches at most two sequences, each staged as its own span
//   MODE_PERSEQ: Ncols < BN: one (ragged) tile row per sequence
enum { MODE_FLAT = 0, MODE_SEG = 1, MODE_PERSEQ = 2 };

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    // bijective XCD-aware remap (cdna_hip_programming.md T1): blocks that run on one XCD (bid % 8)
    // get a contiguous range of tile ids, so row-tiles sharing an input span hit the same L2.
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ unsigned div_magic(unsigned e, unsigned magic) {
    // e / d with magic = ceil(2^32 / d); magic == 0 encodes d == 1 (2^32 does not fit)
    return magic ? (unsigned)(((unsigned long long)e * magic) >> 32) : e;
}

template <int BM, int BN>
__global__ void __launch_bounds__(NTHREADS, 2) conv_gemm_kernel(PaseConvGemm p, ConvPlan pl) {
    constexpr int WAVES_N = BN / 64;
    static_assert((BM / 64) * WAVES_N == 4, "4 waves");
    constexpr int A_ROWS = BM / 8;   // rows per thread per k slot (scalar path)
    constexpr int A_VROWS = BM / 16; // rows per thread (float4 path)
    constexpr int LDA = BM + 1;
    __shared__ float As[2][KGMAX][LDA];
    __shared__ float Xs[2][XSMAX];
    __shared__ int kinfo[2][2];   // per stage: {flat k count, taps in this sub-range}
    __shared__ float red[WAVES_N][BM][2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = pase_uniform(tid >> 6);   // provably wave-uniform: tile-shape branches stay scalar
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int fr = lane & 31;
    const int fk = lane >> 5;

    // ---- tile decode -----------------------------------------------------------------------
    const int ntot = p.S * p.Ncols;
    const int n_row_tiles = (p.M + BM - 1) / BM;
    const int n_col_tiles = (pl.mode == MODE_PERSEQ) ? p.S * pl.tiles_per_seq : (ntot + BN - 1) / BN;
    const int ntiles = n_row_tiles * n_col_tiles;
    const int split = blockIdx.x / ntiles;
    const int tile = xcd_swizzle(blockIdx.x % ntiles, ntiles);
    const int mt = tile % n_row_tiles;
    const int nt = tile / n_row_tiles;
    const int m0 = mt * BM;
    // segment A = first sequence touched (s0, columns qA .. qA+lenA-1), segment B = the next one
    int s0, qA, lenA, lenB, n0 = 0;
    if (pl.mode == MODE_PERSEQ) {
        s0 = nt / pl.tiles_per_seq;
        qA = (nt - s0 * pl.tiles_per_seq) * BN;
        lenA = min(BN, p.Ncols - qA);
        lenB = 0;
    } else {
        n0 = nt * BN;
        s0 = n0 / p.Ncols;
        qA = n0 - s0 * p.Ncols;
        lenA = min(BN, p.Ncols - qA);
        lenB = min(BN - lenA, ntot - n0 - lenA);
        if (lenB < 0) lenB = 0;
    }
    const int ncols_valid = (pl.mode == MODE_FLAT) ? min(BN, ntot - n0) : lenA + lenB;
    const int xstep = (pl.mode == MODE_FLAT) ? 1 : p.stride;
    const int SA = (lenA - 1) * xstep + pl.TB;       // LDS row offset of segment B

    // ---- stage enumeration: g = gc * n_gt + gt ; this split's range -------------------------
    const int G = pl.n_gc * pl.n_gt;
    const int g_per = (G + pl.splitk - 1) / pl.splitk;
    const int g_begin = split * g_per;
    const int g_end = min(G, g_begin + g_per);
    if (g_begin >= g_end) return;   // uniform for the whole block, before any barrier

    float areg[2 * A_ROWS];
    float xreg[XPT];
    int xoff[XPT];                 // element offset of slot t relative to channel ci0 of sequence 0
    unsigned xmask = 0u;           // slot t holds a real sample (else zero padding / out of range)
    int kg_next = 0, tbe_next = 0, ci0_next = 0;

    // slot t of this thread = element e = tid + 256 t of the [CB][SPAN] slab -> (cl, i) -> (s, u)
    auto slot_setup = [&](int kk0, int TBe) __attribute__((always_inline)) {
        xmask = 0u;
        const int koffs = (p.tapstep > 0) ? kk0 : -(kk0 + TBe - 1);
#pragma unroll
        for (int t = 0; t < XPT; ++t) {
            const int e = tid + NTHREADS * t;
            const int cl = (int)div_magic((unsigned)e, pl.span_magic);
            const int i = e - cl * pl.SPAN;
            int s, u;
            bool ok;
            if (pl.mode == MODE_FLAT) {
                const unsigned n = (unsigned)(n0 + i);
                s = (int)div_magic(n, pl.ncols_magic);   // may overshoot by one for huge n*Ncols
                u = (int)n - s * p.Ncols;
                if (u < 0) { --s; u += p.Ncols; }
                ok = (int)n < ntot;
            } else {
                const bool segB = i >= SA;
                s = segB ? s0 + 1 : s0;
                u = (segB ? i - SA : qA * p.stride + i) - p.padL + koffs;
                ok = segB ? lenB > 0 : true;
                if (p.pad_mode == PASE_PAD_REFLECT) {
                    if (u < 0) u = -u;
                    if (u >= p.Tin) u = 2 * (p.Tin - 1) - u;
                }
                ok = ok && u >= 0 && u < p.Tin;
            }
            xoff[t] = ok ? (s * p.x_ctot + cl) * p.Tin + u : 0;
            if (ok) xmask |= 1u << t;
        }
    };
    const bool slots_invariant = pl.n_gt == 1;
    if (slots_invariant) slot_setup(0, p.taps);

    auto load_stage = [&](int g) __attribute__((always_inline)) {
        const int gc = g / pl.n_gt, gt = g - gc * pl.n_gt;
        const int ci0 = gc * pl.CB, kk0 = gt * pl.TB;
        const int TBe = min(pl.TB, p.taps - kk0);
        const int CBe = min(pl.CB, p.Cin - ci0);
        const int KGe = CBe * TBe;
        kg_next = KGe;
        tbe_next = TBe;
        ci0_next = ci0;
        // ---- A slab [BM x KGe], rows K-contiguous in HBM.  Raw prefetch only.
        if (pl.avec) {
            // float4 along K: lane group of 16 covers up to 64 k of one row
            const int k4 = (tid & 15) * 4;
            const float* wrow = p.w + (size_t)(m0 + (tid >> 4)) * p.ldw + (size_t)ci0 * p.taps + kk0 + k4;
            const bool kok = k4 < KGe;
#pragma unroll
            for (int i = 0; i < A_VROWS; ++i) {
                const int m = m0 + (tid >> 4) + 16 * i;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kok && m < p.M) v = *reinterpret_cast<const float4*>(wrow + (size_t)16 * i * p.ldw);
                areg[4 * i + 0] = v.x; areg[4 * i + 1] = v.y; areg[4 * i + 2] = v.z; areg[4 * i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kl = (tid & 31) + 32 * h;
                const bool kok = kl < KGe;
                int ka = 0;
                if (kok) {
                    const int cl = kl / TBe, kkl = kl - cl * TBe;
                    ka = p.tap_major ? (kk0 + kkl) * p.Cin + ci0 + cl : (ci0 + cl) * p.taps + kk0 + kkl;
                }
#pragma unroll
                for (int i = 0; i < A_ROWS; ++i) {
                    const int m = m0 + (tid >> 5) + 8 * i;
                    areg[h * A_ROWS + i] = (kok && m < p.M) ? p.w[(size_t)m * p.ldw + ka] : 0.f;
                }
            }
        }
        // ---- X spans: CBe rows of SPAN floats, consecutive threads on consecutive samples.  Only the
        // raw loads are issued here (they stay in flight under the MFMAs of the current stage); the
        // on-load affine / PReLU is applied in store_stage, after the MFMA loop.
        if (!slots_invariant) slot_setup(kk0, TBe);
        const int total = CBe * pl.SPAN;
        const float* xb = p.x + (size_t)(p.x_coff + ci0) * p.Tin;
#pragma unroll
        for (int t = 0; t < XPT; ++t) {
            const bool ok = ((xmask >> t) & 1u) && (tid + NTHREADS * t) < total;
            xreg[t] = ok ? xb[xoff[t]] : 0.f;
        }
    };
    auto store_stage = [&](int buf) __attribute__((always_inline)) {
        if (pl.avec) {
            const int k4 = (tid & 15) * 4;
            if (k4 < KGMAX) {
#pragma unroll
                for (int i = 0; i < A_VROWS; ++i) {
                    const int r = (tid >> 4) + 16 * i;
                    As[buf][k4 + 0][r] = areg[4 * i + 0];
                    As[buf][k4 + 1][r] = areg[4 * i + 1];
                    As[buf][k4 + 2][r] = areg[4 * i + 2];
                    As[buf][k4 + 3][r] = areg[4 * i + 3];
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kl = (tid & 31) + 32 * h;
                if (kl < KGMAX) {
#pragma unroll
                    for (int i = 0; i < A_ROWS; ++i) As[buf][kl][(tid >> 5) + 8 * i] = areg[h * A_ROWS + i];
                }
            }
        }
        if (p.in_scale || p.in_alpha) {
            const int total = kg_next / tbe_next * pl.SPAN;
#pragma unroll
            for (int t = 0; t < XPT; ++t) {
                const int e = tid + NTHREADS * t;
                if (((xmask >> t) & 1u) && e < total) {
                    const int ci = ci0_next + (int)div_magic((unsigned)e, pl.span_magic);
                    float v = xreg[t];
                    if (p.in_scale) v = v * p.in_scale[ci] + p.in_shift[ci];
                    if (p.in_alpha) v = v > 0.f ? v : v * p.in_alpha[ci];
                    xreg[t] = v;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < XPT; ++t) Xs[buf][tid + NTHREADS * t] = xreg[t];
        if (tid == 0) { kinfo[buf][0] = kg_next; kinfo[buf][1] = tbe_next; }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // wave-uniform "this 32-wide block has work" flags (ragged tiles: T_out = 200, M = 273, ...)
    const bool row_ok0 = m0 + wm * 64 < p.M, row_ok1 = m0 + wm * 64 + 32 < p.M;
    const bool col_ok0 = wn * 64 < ncols_valid, col_ok1 = wn * 64 + 32 < ncols_valid;
    const bool full_tile = row_ok0 && row_ok1 && col_ok0 && col_ok1;
    // LDS offset of this lane's two columns within a slab row (segment B starts at SA)
    const int j0c = wn * 64 + fr, j1c = wn * 64 + 32 + fr;
    const int xc0 = (pl.mode != MODE_FLAT && j0c >= lenA) ? SA + (j0c - lenA) * xstep : j0c * xstep;
    const int xc1 = (pl.mode != MODE_FLAT && j1c >= lenA) ? SA + (j1c - lenA) * xstep : j1c * xstep;

    load_stage(g_begin);
    store_stage(0);
    __syncthreads();
    for (int g = g_begin; g < g_end; ++g) {
        const int cur = (g - g_begin) & 1;
        if (g + 1 < g_end) load_stage(g + 1);   // global loads in flight under the MFMAs
        const int kg = kinfo[cur][0];
        const int tbe = kinfo[cur][1];
        const int nks = (kg + 1) >> 1;
        // K order inside a stage: channel rows are taken two at a time ("super-row" = 2*tbe flat k, so
        // an MFMA step never straddles super-rows even for odd tap counts).  Step j of a super-row
        // gives this lane flat position f = 2j + fk -> row f >= tbe, tap f - row*tbe.  All of it is
        // scalar loop state plus 3-4 VALU; operands for step ks+1 are fetched from LDS BEFORE the
        // MFMAs of step ks are issued (register double buffer), so the ds_read latency hides under
        // the 256 matrix-pipe cycles of the current step.
        // K order inside a stage: channel rows are taken two at a time ("super-row" = 2*tbe flat k,
        // tbe MFMA steps, so a step never straddles super-rows even for odd tap counts).  This lane's
        // flat position in step j is f = 2j + fk.  Its span offset advances by +-2 per step, plus one
        // extra jump D when f crosses from the first to the second row (step j == jc, a per-lane
        // constant) and the same D at the end of the super-row: ~7 VALU per step, no division, no table.
        const int ts = p.tapstep;
        int xo = (fk >= tbe) ? ((ts > 0) ? pl.SPAN - tbe : pl.SPAN + 2 * tbe - 1) : ((ts > 0) ? 0 : tbe - 1);
        xo = (ts > 0) ? xo + fk : xo - fk;
        const bool one_tap = tbe == 1;                       // rows of one tap: plain 2*SPAN stride
        const int xstepk = one_tap ? 2 * pl.SPAN : 2 * ts;
        const int D = one_tap ? 0 : ((ts > 0) ? pl.SPAN - tbe : pl.SPAN + tbe);
        const int jc = one_tap ? -1 : (tbe - fk + 1) / 2 - 1;
        const int xo_lim = XSMAX - 1 - max(xc0, xc1);        // a padded / look-ahead step must stay inside Xs
        const float* as_ = &As[cur][fk][wm * 64 + fr];
        const float* xs_ = &Xs[cur][0];
        int j = 0;
        auto fetch = [&](int ks, float& a0, float& a1, float& b0, float& b1) __attribute__((always_inline)) {
            const int ka = min(ks * 2, KGMAX - 2) * LDA;   // (the one fetch past the end stays in bounds)
            const int xoc = min(xo, xo_lim);
            a0 = as_[ka];
            a1 = as_[ka + 32];
            b0 = xs_[xoc + xc0];
            b1 = xs_[xoc + xc1];
            const int endj = (j == tbe - 1) ? D : 0;       // uniform
            xo += xstepk + ((j == jc) ? D : 0) + endj;
            j = (j == tbe - 1) ? 0 : j + 1;
        };
        // ping-pong operand registers (P/Q), two k-steps per iteration: no register copies, so the
        // wait before a step's MFMAs covers only that step's own ds_reads
        float pa0, pa1, pb0, pb1, qa0, qa1, qb0, qb1;
        fetch(0, pa0, pa1, pb0, pb1);
        const int nks2 = (nks + 1) & ~1;     // an odd step count is padded with a zero-weight step (A rows
                                             // beyond kg are zero-filled) so the loop body is branch-free
        // Interleave: the index arithmetic + ds_reads of the NEXT step are spread between the four MFMAs
        // of the current step (a wave that has issued a 64-cycle MFMA cannot issue the next one for
        // ~60 cycles anyway), instead of sitting in a serial block behind them.
#define PASE_STEP_SCHED()                                             \
        PASE_SGB(0x008, 1); PASE_SGB(0x002, 3); PASE_SGB(0x100, 1);   \
        PASE_SGB(0x008, 1); PASE_SGB(0x002, 3); PASE_SGB(0x100, 1);   \
        PASE_SGB(0x008, 1); PASE_SGB(0x002, 3); PASE_SGB(0x100, 1);   \
        PASE_SGB(0x008, 1); PASE_SGB(0x002, 3);
        for (int ks = 0; ks < nks2; ks += 2) {
            fetch(ks + 1, qa0, qa1, qb0, qb1);
            // All MFMAs are issued unconditionally: rows / columns beyond the tile edge multiply
            // zero-filled A rows or finite staged data and are discarded in the epilogue.
            acc[0][0] = pase_mfma_32x32x2(pa0, pb0, acc[0][0]);
            acc[0][1] = pase_mfma_32x32x2(pa0, pb1, acc[0][1]);
            acc[1][0] = pase_mfma_32x32x2(pa1, pb0, acc[1][0]);
            acc[1][1] = pase_mfma_32x32x2(pa1, pb1, acc[1][1]);
            PASE_STEP_SCHED();
            PASE_SCHED_BARRIER();
            fetch(ks + 2, pa0, pa1, pb0, pb1);
            acc[0][0] = pase_mfma_32x32x2(qa0, qb0, acc[0][0]);
            acc[0][1] = pase_mfma_32x32x2(qa0, qb1, acc[0][1]);
            acc[1][0] = pase_mfma_32x32x2(qa1, qb0, acc[1][0]);
            acc[1][1] = pase_mfma_32x32x2(qa1, qb1, acc[1][1]);
            PASE_STEP_SCHED();
            PASE_SCHED_BARRIER();
        }
#undef PASE_STEP_SCHED
        if (g + 1 < g_end) store_stage(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------
    // D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int rbase = 4 * (lane >> 5);
    int cs[2], cq[2];
    bool cok[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = wn * 64 + b * 32 + fr;
        cok[b] = j < ncols_valid;
        if (pl.mode == MODE_FLAT) {
            const int n = n0 + j;
            cs[b] = cok[b] ? n / p.Ncols : 0;
            cq[b] = cok[b] ? n % p.Ncols : 0;
        } else {
            cs[b] = j < lenA ? s0 : s0 + 1;
            cq[b] = j < lenA ? qA + j : j - lenA;
        }
    }

    if (p.epilogue == PASE_EPI_STORE && (p.post_op == PASE_POST_POW || p.post_op == PASE_POST_LOGPOW)) {
        // spectra: accumulator rows r, r+1 (same lane) are the (re, im) parts of one frequency bin
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + rbase;   // even row
                if (m >= p.M) continue;          // M is even (host-checked): a pair is valid or absent
                const int bin = m >> 1;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float re = acc[a][b][r], im = acc[a][b][r + 1];
                    float v = re * re + im * im;
                    v = (p.post_op == PASE_POST_LOGPOW) ? p.post_scale * logf(v + p.post_eps) : v * p.post_scale;
                    const int pos = cq[b] + p.poff;
                    if (cok[b] && pos >= 0 && pos < p.Tout)
                        p.y[((size_t)cs[b] * p.y_ctot + p.y_coff + bin) * (size_t)p.Tout + pos] = v;
                }
            }
        }
    } else if (p.epilogue == PASE_EPI_STORE) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + rbase;  // row in tile
                const int m = m0 + ml;
                const bool mok = m < p.M;
                const int ph = mok ? m / p.Cout_store : 0;
                const int co = mok ? m - ph * p.Cout_store : 0;
                const float bv = (mok && p.bias && split == 0) ? p.bias[co] : 0.f;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float v = acc[a][b][r] + bv;
                    if (p.post_op == PASE_POST_LOG) v = p.post_scale * logf(v == 0.f ? p.post_eps : v);
                    const int pos = cq[b] * p.ps + ph + p.poff;
                    const bool ok = mok && cok[b] && pos >= 0 && pos < p.Tout;
                    if (ok) {
                        float* dst = p.y + ((size_t)cs[b] * p.y_ctot + p.y_coff + co) * (size_t)p.Tout + pos;
                        if (pl.splitk > 1) atomicAdd(dst, v);
                        else *dst = v;
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

struct HostPlan {
    ConvPlan pl;
    int BN, narrow;
    long blocks;
    int n_col_tiles;
};

HostPlan make_plan(const PaseConvGemm& p) {
    HostPlan h;
    h.narrow = (p.tile_hint == 64) || (p.tile_hint == 0 && p.M <= 64);
    const int BM = h.narrow ? 64 : 128;
    h.BN = h.narrow ? 256 : 128;
    ConvPlan& pl = h.pl;
    const bool flat = (p.taps == 1 && p.stride == 1 && p.padL == 0 && p.tapstep == 1);
    pl.mode = flat ? MODE_FLAT : (p.Ncols >= h.BN ? MODE_SEG : MODE_PERSEQ);
    if (flat) {
        pl.TB = 1;
        pl.SPAN = h.BN;
        pl.CB = XSMAX / h.BN;
        if (pl.CB > KGMAX) pl.CB = KGMAX;
    } else {
        pl.TB = p.taps <= KGMAX ? p.taps : 32;
        // a tile may touch two sequences: each segment carries its own halo of TB samples
        pl.SPAN = (h.BN - 1) * p.stride + (pl.mode == MODE_SEG ? 2 : 1) * pl.TB;
        pl.CB = KGMAX / pl.TB;
        if (pl.CB * pl.SPAN > XSMAX) pl.CB = XSMAX / pl.SPAN;
    }
    if (pl.CB > p.Cin) pl.CB = p.Cin;
    if (pl.CB < 1) pl.CB = 0;   // span does not fit: rejected by the caller
    pl.n_gc = pl.CB ? (p.Cin + pl.CB - 1) / pl.CB : 0;
    pl.n_gt = (p.taps + pl.TB - 1) / pl.TB;
    pl.tiles_per_seq = (p.Ncols + h.BN - 1) / h.BN;
    pl.span_magic = (unsigned)((0x100000000ULL + pl.SPAN - 1) / (unsigned long long)pl.SPAN);
    pl.ncols_magic = (unsigned)((0x100000000ULL + p.Ncols - 1) / (unsigned long long)p.Ncols);
    // float4 weight loads: rows 16-B aligned, whole channels per stage, natural (ci, kk) K order
    pl.avec = (!p.tap_major && pl.n_gt == 1 && (p.ldw % 4) == 0 && ((pl.CB * pl.TB) % 4) == 0 &&
               (((unsigned long long)(size_t)p.w) % 16) == 0 && pl.CB * pl.TB <= 64 && pl.CB > 0 &&
               (p.Cin % pl.CB) == 0) ? 1 : 0;
    const long ntot = (long)p.S * p.Ncols;
    h.n_col_tiles = (pl.mode == MODE_PERSEQ) ? p.S * pl.tiles_per_seq : (int)((ntot + h.BN - 1) / h.BN);
    const long tiles = (long)((p.M + BM - 1) / BM) * h.n_col_tiles;
    int splitk = 1;
    const int G = pl.n_gc * pl.n_gt;
    if (p.splitk > 1) splitk = p.splitk;
    else if (p.splitk == 0 && !p.stat_part && p.epilogue == PASE_EPI_STORE && p.post_op == PASE_POST_NONE && tiles < 192) {
        // auto: few output tiles and a long reduction (head / deconv data-gradients) -> fill the chip
        splitk = (int)((384 + tiles - 1) / tiles);
        if (splitk > G / 6) splitk = G / 6;
        if (splitk < 1) splitk = 1;
    }
    if (splitk > G) splitk = G > 0 ? G : 1;
    if (splitk > 1) {   // every split must own at least one stage
        const int g_per = (G + splitk - 1) / splitk;
        splitk = (G + g_per - 1) / g_per;
    }
    pl.splitk = splitk;
    h.blocks = tiles * splitk;
    return h;
}

}  // namespace

extern "C" int pase_conv_gemm(const PaseConvGemm* d, void* stream) {
    const PaseConvGemm p = *d;
    if (p.M <= 0 || p.K <= 0 || p.S <= 0 || p.Ncols <= 0) return 0;
    if (p.K != p.Cin * p.taps) return -4;
    if (p.epilogue == PASE_EPI_MSE_CTX && (!p.label || !p.loss_acc || p.r_ctx < 1)) return -2;
    if (p.pad_mode == PASE_PAD_REFLECT && (p.padL >= p.Tin)) return -3;
    if (p.tapstep != 1 && p.tapstep != -1) return -5;
    const HostPlan h = make_plan(p);
    if (h.pl.CB < 1) return -6;
    if (h.pl.splitk > 1 && (p.stat_part || p.epilogue != PASE_EPI_STORE || p.post_op != PASE_POST_NONE)) return -7;
    if ((p.post_op == PASE_POST_POW || p.post_op == PASE_POST_LOGPOW) && (p.ps != 1 || p.stat_part || (p.M & 1))) return -9;
    if ((long)p.S * p.Ncols >= 0x7fffffffL) return -8;
    if ((long)p.S * p.x_ctot * (long)p.Tin >= 0x7fffffffL) return -8;   // int element offsets in the loader
    hipStream_t st = (hipStream_t)stream;
    if (h.narrow) {
        PASE_LAUNCH((conv_gemm_kernel<64, 256>), dim3((unsigned)h.blocks), dim3(NTHREADS), st, p, h.pl);
    } else {
        PASE_LAUNCH((conv_gemm_kernel<128, 128>), dim3((unsigned)h.blocks), dim3(NTHREADS), st, p, h.pl);
    }
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_conv_gemm_stat_tiles(const PaseConvGemm* d) {
    return make_plan(*d).n_col_tiles;
}

extern "C" int pase_conv_gemm_splitk(const PaseConvGemm* d) {
    return make_plan(*d).pl.splitk;
}

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
// im2col tile: per stage the block stages the raw sliding-window SPANS of CB input channels (length
// (BN-1)*stride + TB, affine+PReLU and padding applied once per element) plus the matching
// [CB*TB x BM] slab of the K-MAJOR weight pack (pase_pack_wt) into LDS (double-buffered, register
// prefetch, one barrier per stage).  An MFMA B fragment is then a strided ds_read_b32 straight out
// of the span: column j, tap kk -> Xs[cl][j*stride + kk]; each lane walks its (channel row, tap)
// offset incrementally, so neither integer division nor a table lookup sits in the MFMA loop.
//
// The per-stage loader is the part that competes with the matrix pipe for issue slots (two
// workgroups share a CU), so it is kept to a handful of instructions per thread:
//   * A slab: <= 6 unconditional global_load_dwordx4 from the K-major pack (rows = flat k, BM
//     contiguous floats per row), written with ds_write_b128; no per-element index arithmetic.
//   * X spans: every thread owns ONE channel row of the slab (row = tid / TPR) and Q samples of it;
//     all (sequence, time, padding) arithmetic is done once per tile into per-slot offsets + a mask,
//     per stage a slot is one global load off a uniform base that advances by CB channels.  The
//     thread's channel is fixed, so the on-load affine / PReLU needs one (scale, shift, alpha)
//     triple per stage instead of one per element.
//   * a ragged last channel group is shifted back to end at Cin (its already-covered rows get zero
//     weights), so the loader has no channel-validity branches.
// Two shapes: <128,128> (waves 2x2) and <64,256> (waves 1x4) for the 64-row layers.
#include <cstdlib>
#include <type_traits>

#include "hip_compat.h"
#include "pase_amd.h"
#include "conv_x6c.h"
#include "sinc_x6.h"

namespace {

constexpr int NTHREADS = 256;
constexpr int KGMAX = 48;     // flat k (channels x taps) per stage
constexpr int XSMAX = 3072;   // staged span floats per stage
constexpr int XPT = XSMAX / NTHREADS;   // 12 staged floats per thread per stage
constexpr int NPAR = 4;       // on-load (scale, shift, alpha) triples prefetched per thread
// flat 1x1 instantiation (float4 slots): 32 k per stage x BN columns = 4096 staged floats and a 32-row weight
// slab -- more columns of X per stage than the span layout needs, fewer rows of A, same LDS footprint
constexpr int KG_FLAT = 32, XS_FLAT = 4096, NS_FLAT = 4;
// split-bf16 ("x6") instantiations: a stage is up to X6 (template value: 3, or 4 for the 30-tap layers) MFMA steps of 16 k; its weight slab is the three bf16
// planes in fragment order, 16-byte chunks [step][k-group][plane][tile row], single-buffered (36 KB for 128 rows)
constexpr int X6_STEPS = 3, X6_STEPS_LONG = 4;

struct ConvPlan {
    int CB, TB, SPAN, SPANV, n_gc, n_gt, mode, tiles_per_seq, splitk;
    int tl;        // log2(threads per slab row)
    int pmajor;    // 0: slot t = sample c + t*TPR of row tid/TPR;  1 (flat 1x1): slot t = row t*RPP + tid/TPR
    int nslots;    // slots per thread (Q or PX)
    int xvec;      // flat mode, float4 slots
    int PA;        // A slab passes
    unsigned ncols_magic;     // ceil(2^32 / Ncols)
    unsigned cout_magic;      // ceil(2^32 / Cout_store)
    unsigned rctx_magic;      // ceil(2^32 / r_ctx)
    // x6: a 16-deep MFMA step = xR channel rows x xTq taps (xR * xTq = 16; lane half fk takes rows fk, fk+2, ...);
    // a stage = (CB / xR) row groups x xTS tap blocks = xSteps steps; TB is the tap count padded to xTS * xTq
    // xR == 1 (long filters on few channels, e.g. the 251-tap Sinc FIR on one): a step = 16 taps of ONE row (lane half fk
    // takes taps 8 fk .. 8 fk + 7), a stage = up to xSteps steps of one tap group (TB = 16 xSteps), xTaps = padded tap count
    int x6, xR, xTq, xTS, xSteps, xTaps;
    // x6 pixel-shuffle launches: tile rows are re-ordered (channel, phase) by pase_pack_x6 (m = co * ps + phase), so the four
    // consecutive rows a lane holds per register quad are consecutive OUTPUT samples: one 16-byte store instead of four
    // 4-byte stores 40 bytes apart (the ps = 10 layers were bound by partial-line write transactions)
    int xPerm;
    unsigned ps_magic;        // ceil(2^32 / ps)
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

#ifdef PASE_TRACE   // tools/trace_conv.py only: per-workgroup phase timestamps (100 MHz wall clock)
#define PASE_TRACE_SLOTS 16384
__device__ unsigned long long g_trace[PASE_TRACE_SLOTS * 6];
#define PASE_STAMP(i)                                                                              \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < PASE_TRACE_SLOTS) g_trace[blockIdx.x * 6 + (i)] = wall_clock64(); \
    } while (0)
#define PASE_TACC_DECL unsigned long long tacc_[4] = {0, 0, 0, 0}, tl_ = 0
#define PASE_TACC_BEGIN() tl_ = wall_clock64()
#define PASE_TACC(i) do { const unsigned long long n_ = wall_clock64(); tacc_[i] += n_ - tl_; tl_ = n_; } while (0)
#define PASE_TACC_DUMP()                                                                        \
    do {                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < PASE_TRACE_SLOTS)                                  \
            for (int i_ = 0; i_ < 4; ++i_) g_tacc[blockIdx.x * 4 + i_] = tacc_[i_];             \
    } while (0)
__device__ unsigned long long g_tacc[PASE_TRACE_SLOTS * 4];
#else
#define PASE_STAMP(i)
#define PASE_TACC_DECL
#define PASE_TACC_BEGIN()
#define PASE_TACC(i)
#define PASE_TACC_DUMP()
#endif

struct alignas(16) F4 { float x, y, z, w; };

// identity on-load parameters for launches without an affine / PReLU ({scale = alpha = 1}, {shift = 0})
// (plain global memory, not constant address space: a pointer selected between it and a kernel argument
// must stay a GLOBAL pointer -- a generic one turns the loads into flat_load and every wait into vmcnt(0))
__device__ float g_ident[2] = {1.f, 0.f};

// NS = X slots per thread (the plan rounds its slot count up to 3 / 6 / 12), XV = float4 slots (flat 1x1).
// Both are compile-time so that the per-stage loader is straight-line code: every load is issued
// unconditionally from a precomputed offset (slots / rows that do not exist re-read a valid address and
// land in slab padding), which keeps the loads independent of each other in the instruction stream.
// OCC = workgroups per CU the instantiation is sized for: 3 needs <= 53 KB of LDS (44-row weight slab, 3-slot span
// slab: the 11-tap and 30-tap layers) and <= 168 VGPRs.
// XFM (flat instantiation only; -1 = decided at run time): the row-major float4 slots of the flat 1x1 path carry one
// (scale, shift, alpha) triple PER SLOT, i.e. 12 extra loads per stage next to the 8 operand loads -- specialised away
// when the launch has no on-load transform (0: every data-gradient) or only a PReLU slope (1: the worker heads).
// X6: the contraction runs on v_mfma_f32_32x32x16_bf16 with both operands split into three bf16 pieces (see
// PaseConvGemm::wx6).  The weight slab arrives pre-split (pase_pack_x6); the activation spans are staged exactly as in
// the fp32 instantiation and each wave splits its B fragment when it reads it.
template <int BM, int BN, int NS, int XV, int OCC = 2, int XFM = -1, int X6 = 0>
__global__ void __launch_bounds__(NTHREADS, OCC) conv_gemm_kernel(PaseConvGemm p, ConvPlan pl) {
    constexpr int WAVES_N = BN / 64;
    static_assert((BM / 64) * WAVES_N == 4, "4 waves");
    constexpr int TA = BM / 4;             // threads per A slab row (one float4 each)
    constexpr int RA = NTHREADS / TA;      // slab rows per pass (8 / 16)
    constexpr int KG_T = XV ? KG_FLAT : (OCC == 3 ? 44 : KGMAX);
    constexpr int XS_T = XV ? XS_FLAT : (OCC == 3 ? NS * NTHREADS : XSMAX);
    constexpr int PA_MAX = (KG_T + RA - 1) / RA;      // 6 / 3 (4 / 2 flat)
    constexpr int LDA = BM + 4;
    constexpr int AX_CHUNKS = 3 * (X6 ? X6 : 1) * 2 * BM;          // x6: 16-byte chunks of one stage's weight slab
    constexpr int NCH = (AX_CHUNKS + NTHREADS - 1) / NTHREADS;    // ... per thread (9 / 12; 5 for the 64-row tile)
    constexpr int A_BYTES = X6 ? AX_CHUNKS * 16 : 2 * KG_T * LDA * (int)sizeof(float);
    __shared__ __attribute__((aligned(16))) unsigned char As_raw[A_BYTES];
    float (*As)[KG_T][LDA] = reinterpret_cast<float (*)[KG_T][LDA]>(As_raw);
    u32x4* AsX = reinterpret_cast<u32x4*>(As_raw);
    // 4 guard floats (zero) in front of each X buffer: the one zero-weight tap a reversed-tap single-row stage
    // with an odd tap count reads at span offset -1 must be finite
    __shared__ __attribute__((aligned(16))) float XsG[2][XS_T + 4];
    // epilogue scratch (BN partial sums / loss partials) reuses the weight slab: it is dead after the last stage
    float (*red)[BM][2] = reinterpret_cast<float (*)[BM][2]>(As_raw);
    static_assert(sizeof(float) * WAVES_N * BM * 2 <= sizeof(As_raw), "epilogue scratch");

    PASE_STAMP(0);
    const int tid = threadIdx.x;
    if (tid < 8) XsG[tid >> 2][tid & 3] = 0.f;
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

    // ---- loader state ------------------------------------------------------------------------
    // A: thread -> (slab row ar + RA*pass, 4 tile columns from acl) of the K-major pack
    const int ar = tid / TA;
    const int acl = (tid % TA) * 4;
    const unsigned a_col = (unsigned)min(m0 + acl, p.ldwt - 4);
    constexpr int PA_N = X6 ? NCH : PA_MAX;      // A pieces per stage
    F4 areg[X6 ? 1 : PA_MAX];
    u32x4 aregx[X6 ? NCH : 1];
    const int ax_nch = 3 * pl.xSteps * 2 * BM;   // x6: chunks of this launch's stages
    const u32x4* ax_st = nullptr;
    // X: thread -> slab row `xrow` (+ t*RPP when pmajor), samples xc + t*TPR (q-major)
    constexpr int NXR = XV ? 4 * NS : NS;      // staged floats per thread
    static_assert(NXR * NTHREADS <= XS_T, "slab");
    const int TPR = 1 << pl.tl;
    const int RPP = NTHREADS >> pl.tl;
    const int xrow = tid >> pl.tl;
    const int xc = tid & (TPR - 1);
    const int xs_tbase = xrow * pl.SPAN + (XV ? 4 * xc : xc);
    constexpr bool PM = XV != 0;   // row-major slots (flat 1x1, always float4) vs sample-major slots
    const int slot_stride = PM ? RPP * pl.SPAN : TPR;   // LDS distance between a thread's slots
    float xreg[NXR];
    int xoff[NS];                  // element offset of slot t relative to channel ci0 of sequence 0
    unsigned xmask = 0u;           // slot t holds a real sample (else zero padding / out of range)
    static_assert(!PM || NS <= NPAR, "row-major slots prefetch one parameter triple per slot");
    constexpr int NP = PM ? NS : 1;            // (scale, shift, alpha) triples prefetched per stage
    float par_s[NP], par_h[NP], par_a[NP];
    const bool has_xf = XFM >= 0 ? XFM > 0 : (p.in_scale != nullptr || p.in_alpha != nullptr);
    const float* sc_p = p.in_scale ? p.in_scale : &g_ident[0];
    const float* sh_p = p.in_scale ? p.in_shift : &g_ident[1];
    const float* al_p = p.in_alpha ? p.in_alpha : &g_ident[0];
    const int aff_on = p.in_scale ? 1 : 0, alpha_on = p.in_alpha ? 1 : 0;
    int kg_next = 0, tbe_next = 0, lo_next = 0;

    // (sequence, time) of span sample i for tap-group offset koffs -> element offset / validity
    auto locate = [&](int i, int koffs, int& off, bool& ok) __attribute__((always_inline)) {
        int s, u;
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
        ok = ok && i < pl.SPANV;
        off = s * p.x_ctot * p.Tin + u;
    };
    auto slot_setup = [&](int kk0, int TBe) __attribute__((always_inline)) {
        xmask = 0u;
        const int koffs = (p.tapstep > 0) ? kk0 : -(kk0 + TBe - 1);
        if (PM) {
            int off;
            bool ok;
            locate(4 * xc, koffs, off, ok);
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int row = xrow + t * RPP;
                const bool okr = ok && row < pl.CB;
                xoff[t] = okr ? off + row * p.Tin : 0;
                if (okr) xmask |= 1u << t;
            }
        } else {
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                int off;
                bool ok;
                locate(xc + (t << pl.tl), koffs, off, ok);
                ok = ok && xrow < pl.CB;
                xoff[t] = ok ? off + xrow * p.Tin : 0;
                if (ok) xmask |= 1u << t;
            }
        }
    };
    const bool slots_invariant = pl.n_gt == 1;
    if (slots_invariant) slot_setup(0, p.taps);
    constexpr unsigned all_slots = (1u << NS) - 1u;
    // wave-uniform: no slot of this wave needs zeroing (interior tile)
    bool x_allvalid = pase_wave_all(xmask == all_slots) != 0;

    int gc_n = g_begin / pl.n_gt, gt_n = g_begin - gc_n * pl.n_gt;   // (gc, gt) of the next stage to load

    // A stage's loads = stage_begin() (uniform bookkeeping: which channels / taps, base pointers) followed by
    // NPIECE independent pieces (A slab passes, then X slots; the on-load parameters ride with the last X piece).
    // load_stage() issues them back to back; the flat instantiation spreads them over the first half of the
    // current stage's MFMA loop instead (see the k-loop).
    constexpr int NPIECE = PA_N + NS;
    int k0_st = 0, ci0s_st = 0;
    const float* xb_st = p.x;
    auto stage_begin = [&]() __attribute__((always_inline)) {
        const int ci0 = gc_n * pl.CB, kk0 = gt_n * pl.TB;
        const int TBe = min(pl.TB, p.taps - kk0);
        // a ragged last channel group is shifted back so it ends at Cin; rows below `lo` (channels the
        // previous stage already covered) get zero weights in store_stage
        const int ci0s = slots_invariant ? min(ci0, p.Cin - pl.CB) : ci0;
        lo_next = (ci0 - ci0s) * pl.TB;
        kg_next = slots_invariant ? pl.CB * pl.TB : TBe;
        tbe_next = TBe;
        if (X6) ax_st = reinterpret_cast<const u32x4*>(p.wx6) +
                        (size_t)((mt * pl.n_gc + gc_n) * pl.n_gt + gt_n) * (unsigned)ax_nch;
        if (++gt_n == pl.n_gt) { gt_n = 0; ++gc_n; }
        if (!slots_invariant) {
            slot_setup(kk0, TBe);
            x_allvalid = pase_wave_all(xmask == all_slots) != 0;
        }
        k0_st = ci0s * p.taps + kk0 + ar;
        ci0s_st = ci0s;
        xb_st = p.x + (size_t)(p.x_coff + ci0s) * p.Tin;
    };
    // ---- straight-line issue: A slab rows k0 + RA*pass of the K-major pack (BM contiguous floats
    // each; rows past K re-read row K-1 and are zeroed / ignored), then the X slots.  Only raw loads
    // here (they stay in flight under the MFMAs of the current stage); the on-load affine / PReLU is
    // applied in store_stage.
    auto load_piece = [&](auto piece_tag) __attribute__((always_inline)) {
        constexpr int i = decltype(piece_tag)::value;
        if constexpr (i < PA_N) {
            if constexpr (X6) {
                aregx[i] = ax_st[min(tid + NTHREADS * i, ax_nch - 1)];
            } else {
                const unsigned row = (unsigned)min(k0_st + RA * i, p.K - 1);
                areg[i] = *reinterpret_cast<const F4*>(p.wt + (row * (unsigned)p.ldwt + a_col));
            }
        } else {
            constexpr int t = i - PA_N;
            if (XV) {
                const F4 v = *reinterpret_cast<const F4*>(xb_st + (unsigned)xoff[t]);
                xreg[4 * t + 0] = v.x; xreg[4 * t + 1] = v.y; xreg[4 * t + 2] = v.z; xreg[4 * t + 3] = v.w;
            } else {
                xreg[t] = xb_st[(unsigned)xoff[t]];
            }
            if (t == NS - 1 && XFM != 0) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const int ch = min(ci0s_st + xrow + j * RPP, p.Cin - 1);
                    if (XFM != 1) {
                        par_s[j] = sc_p[ch * aff_on];
                        par_h[j] = sh_p[ch * aff_on];
                    }
                    par_a[j] = al_p[ch * alpha_on];
                }
            }
        }
    };
    auto load_pieces = [&](auto first_tag, auto count_tag) __attribute__((always_inline)) {
        constexpr int F = decltype(first_tag)::value, N = decltype(count_tag)::value;
        pase_static_for<N>([&](auto j) __attribute__((always_inline)) {
            constexpr int i = F + decltype(j)::value;
            if constexpr (i < NPIECE) load_piece(std::integral_constant<int, i>{});
        });
    };
    auto load_stage = [&]() __attribute__((always_inline)) {
        stage_begin();
        load_pieces(std::integral_constant<int, 0>{}, std::integral_constant<int, NPIECE>{});
    };
    auto xform = [&](float v, float sc, float sh, float al) __attribute__((always_inline)) {
        v = fmaf(v, sc, sh);
        return v > 0.f ? v : v * al;
    };
    auto store_stage = [&](int buf) __attribute__((always_inline)) {
        // ---- A
        if constexpr (X6) {
            // pre-split chunks, already in LDS order (ragged channel groups / padded taps are zero in the pack)
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (tid + NTHREADS * i < ax_nch) AsX[tid + NTHREADS * i] = aregx[i];
        } else {
            const bool a_zero = lo_next > 0 || (kg_next & 3) != 0;   // uniform
#pragma unroll
            for (int ps = 0; ps < PA_MAX; ++ps) {
                F4 v = areg[ps];
                const int kl = ar + RA * ps;
                if (a_zero && (kl < lo_next || kl >= kg_next)) v = F4{0.f, 0.f, 0.f, 0.f};
                if (PA_MAX * RA == KG_T || kl < KG_T) *reinterpret_cast<F4*>(&As[buf][kl][acl]) = v;
            }
        }
        // ---- X
        float* xs = &XsG[buf][4 + xs_tbase];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            constexpr int W = XV ? 4 : 1;
            float v[W];
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] = xreg[W * t + e];
            if (XFM == 1) {
                const float al = par_a[PM ? t : 0];
#pragma unroll
                for (int e = 0; e < W; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * al;
            } else if (has_xf) {   // uniform
                const float sc = par_s[PM ? t : 0], sh = par_h[PM ? t : 0], al = par_a[PM ? t : 0];
#pragma unroll
                for (int e = 0; e < W; ++e) v[e] = xform(v[e], sc, sh, al);
            }
            if (!x_allvalid && !((xmask >> t) & 1u)) {
#pragma unroll
                for (int e = 0; e < W; ++e) v[e] = 0.f;
            }
            if (XV) *reinterpret_cast<F4*>(xs + t * slot_stride) = F4{v[0], v[W > 1 ? 1 : 0], v[W > 2 ? 2 : 0], v[W > 3 ? 3 : 0]};
            else xs[t * slot_stride] = v[0];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // LDS offset of this lane's two columns within a slab row (segment B starts at SA)
    const int j0c = wn * 64 + fr, j1c = wn * 64 + 32 + fr;
    const int xc0 = (pl.mode != MODE_FLAT && j0c >= lenA) ? SA + (j0c - lenA) * xstep : j0c * xstep;
    const int xc1 = (pl.mode != MODE_FLAT && j1c >= lenA) ? SA + (j1c - lenA) * xstep : j1c * xstep;

    load_stage();
    PASE_STAMP(1);
    store_stage(0);
    int kg = kg_next, tbe = tbe_next;
    __syncthreads();
    PASE_STAMP(2);
    PASE_TACC_DECL;
    for (int g = g_begin; g < g_end; ++g) {
        const int cur = (g - g_begin) & 1;
        PASE_TACC_BEGIN();
#ifndef PASE_FLAT_GENERIC
        const bool spread_loads = !X6 && XV && kg == KG_T;        // uniform: flat full stage issues them inside the loop
#else
        const bool spread_loads = false;
#endif
        if (g + 1 < g_end && !spread_loads) load_stage();   // global loads in flight under the MFMAs
        PASE_TACC(0);
        // K order inside a stage.  An MFMA step consumes two flat k values (fk = 0 / 1).  With two or more
        // channel rows per stage (CB is even then) the pair is (row 2p, tap j) / (row 2p+1, tap j); with a
        // single row it is taps (2s, 2s+1) (an odd tap count ends on a zero-weight tap).  Either way the
        // LDS offsets of a step are [per-lane constant] + [uniform scalar walk]: the walk is SALU only and
        // the per-step VALU work is the three address adds of the ds_reads -- index arithmetic in this
        // loop competes directly with MFMA issue (tools/mfma_probe: -15 % for a 10-instruction walk).
        if constexpr (X6) {
            // ---- split-bf16 stage.  Step st = (row group rg, tap block tb): lane half fk holds rows
            // rg*xR + fk + 2*(e / xTq), taps tb*xTq + e % xTq (e = 0..7) of its column -- the order pase_pack_x6 wrote
            // the weight fragments in.  LDS offsets = [per-lane constant] + [uniform]; the B fragment is split into
            // its three bf16 pieces here (2 x (8 ds_read_b32 + 44 VALU) per 24 MFMAs).
            const int ts = p.tapstep;
            const bool one_row = pl.xR == 1;                        // uniform
            const int lt = one_row ? 4 : (pl.xTq == 8 ? 3 : (pl.xTq == 4 ? 2 : (pl.xTq == 2 ? 1 : 0)));
            const int tap0 = ts > 0 ? 0 : pl.TB - 1;
            const int fkoff = one_row ? 8 * ts : pl.SPAN;
            const float* x0L = &XsG[cur][4 + fk * fkoff + tap0 + xc0];
            const float* x1L = &XsG[cur][4 + fk * fkoff + tap0 + xc1];
            const u32x4* aL = &AsX[fk * 3 * BM + wm * 64 + fr];
            int oe[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) oe[e] = one_row ? ts * e : 2 * (e >> lt) * pl.SPAN + ts * (e & (pl.xTq - 1));
            const int n_rg = one_row ? 1 : pl.xSteps / pl.xTS;
            const int n_tb = one_row ? (tbe >> 4) : pl.xTS;         // (a tap group's last stage may hold fewer steps)
            int st = 0;
            for (int rg = 0; rg < n_rg; ++rg) {
                for (int tb = 0; tb < n_tb; ++tb, ++st) {
                    const int sb = rg * pl.xR * pl.SPAN + ts * (tb << lt);
                    float xv0[8], xv1[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        xv0[e] = x0L[sb + oe[e]];
                        xv1[e] = x1L[sb + oe[e]];
                    }
                    u32x4 fa[3][2], fb0[3], fb1[3];
#pragma unroll
                    for (int pz = 0; pz < 3; ++pz) {
                        fa[pz][0] = aL[(st * 6 + pz) * BM];
                        fa[pz][1] = aL[(st * 6 + pz) * BM + 32];
                    }
                    pase_split_bf16x3(xv0, fb0);
                    pase_split_bf16x3(xv1, fb1);
                    // hh + hm + mh + hl + lh + mm, smallest terms first (plane 0 = hi, 1 = mid, 2 = lo)
                    constexpr int PZA[6] = {1, 0, 2, 0, 1, 0}, PZB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                    for (int pi = 0; pi < 6; ++pi) {
                        acc[0][0] = pase_mfma_bf16_32x32x16(fa[PZA[pi]][0], fb0[PZB[pi]], acc[0][0]);
                        acc[0][1] = pase_mfma_bf16_32x32x16(fa[PZA[pi]][0], fb1[PZB[pi]], acc[0][1]);
                        acc[1][0] = pase_mfma_bf16_32x32x16(fa[PZA[pi]][1], fb0[PZB[pi]], acc[1][0]);
                        acc[1][1] = pase_mfma_bf16_32x32x16(fa[PZA[pi]][1], fb1[PZB[pi]], acc[1][1]);
                    }
                }
            }
            PASE_TACC(1);
            // the weight slab is single-buffered: everyone is done reading it before the next one lands
            __syncthreads();
            if (g + 1 < g_end) {
                store_stage(cur ^ 1);
                tbe = tbe_next;
            }
            PASE_TACC(2);
            __syncthreads();
            PASE_TACC(3);
            continue;
        }
#ifndef PASE_FLAT_GENERIC      // (-DPASE_FLAT_GENERIC: A/B build that keeps the flat path on the generic stage loop below)
        if (XV && kg == KG_T) {
            // ---- flat 1x1, full 32-row stage: the K order is simply rows (2 ks + fk), so every LDS offset is an
            // immediate and the 8 x (2 k-steps) loop is fully unrolled (no address VALU, static waitcnt pattern).
            constexpr int NIT = KG_T / 4;
            // the whole next stage is issued behind the first iteration's MFMAs.  tools/ab_flat_loads.py (A/B builds
            // -DPASE_FLAT_SPREAD: the eight loads spread over the first half of the loop; -DPASE_FLAT_GENERIC: the
            // generic stage loop below): no measurable difference between the three (104-105 TFLOP/s on K = 21 525).
#ifdef PASE_FLAT_SPREAD
            constexpr int SPREAD_ITS = NIT / 2;
#else
            constexpr int SPREAD_ITS = 1;
#endif
            constexpr int PER_IT = (NPIECE + SPREAD_ITS - 1) / SPREAD_ITS;
            const float* aL = &As[cur][fk][wm * 64 + fr];
            const float* x0L = &XsG[cur][4 + fk * BN + xc0];
            const float* x1L = &XsG[cur][4 + fk * BN + xc1];
            const bool has_next = g + 1 < g_end;
            if (has_next) stage_begin();
            float pa0 = aL[0], pa1 = aL[32], pb0 = x0L[0], pb1 = x1L[0], qa0, qa1, qb0, qb1;
            pase_static_for<NIT>([&](auto it_tag) __attribute__((always_inline)) {
                constexpr int it = decltype(it_tag)::value;
                constexpr int k1 = 2 * it + 1, k2 = (2 * it + 2 < KG_T / 2) ? 2 * it + 2 : 2 * it + 1;
                qa0 = aL[k1 * 2 * LDA]; qa1 = aL[k1 * 2 * LDA + 32]; qb0 = x0L[k1 * 2 * BN]; qb1 = x1L[k1 * 2 * BN];
                PASE_SCHED_BARRIER();
                acc[0][0] = pase_mfma_32x32x2(pa0, pb0, acc[0][0]);
                acc[0][1] = pase_mfma_32x32x2(pa0, pb1, acc[0][1]);
                acc[1][0] = pase_mfma_32x32x2(pa1, pb0, acc[1][0]);
                acc[1][1] = pase_mfma_32x32x2(pa1, pb1, acc[1][1]);
                PASE_SCHED_BARRIER();
                pa0 = aL[k2 * 2 * LDA]; pa1 = aL[k2 * 2 * LDA + 32]; pb0 = x0L[k2 * 2 * BN]; pb1 = x1L[k2 * 2 * BN];
                PASE_SCHED_BARRIER();
                acc[0][0] = pase_mfma_32x32x2(qa0, qb0, acc[0][0]);
                acc[0][1] = pase_mfma_32x32x2(qa0, qb1, acc[0][1]);
                acc[1][0] = pase_mfma_32x32x2(qa1, qb0, acc[1][0]);
                acc[1][1] = pase_mfma_32x32x2(qa1, qb1, acc[1][1]);
                PASE_SCHED_BARRIER();
                if constexpr (it < SPREAD_ITS) {
                    if (has_next) {
                        load_pieces(std::integral_constant<int, it * PER_IT>{}, std::integral_constant<int, PER_IT>{});
                        PASE_SCHED_BARRIER();
                    }
                }
            });
            PASE_TACC(1);
            if (has_next) {
                store_stage(cur ^ 1);
                kg = kg_next;
                tbe = tbe_next;
            }
            PASE_TACC(2);
            __syncthreads();
            PASE_TACC(3);
            continue;
        }
#endif
        const int ts = p.tapstep;
        const bool pair_rows = pl.CB > 1;                       // uniform
        const int nks = pair_rows ? (kg >> 1) : ((tbe + 1) >> 1);
        const int stepX = pair_rows ? ts : 2 * ts;
        const int stepA = pair_rows ? LDA : 2 * LDA;
        const int period = pair_rows ? tbe : 0x7fffffff;        // steps until the walk moves to the next row pair
        const int wrapX = 2 * pl.SPAN - tbe * ts;
        const int wrapA = tbe * LDA;
        const int tap0 = ts > 0 ? 0 : tbe - 1;
        const int lx = pair_rows ? fk * pl.SPAN + tap0 : tap0 + fk * ts;
        const float* aL = &As[cur][pair_rows ? fk * tbe : fk][wm * 64 + fr];
        const float* x0L = &XsG[cur][4 + lx + xc0];
        const float* x1L = &XsG[cur][4 + lx + xc1];
        int sx = 0, sa = 0, j = 0;
        auto fetch = [&](float& a0, float& a1, float& b0, float& b1) __attribute__((always_inline)) {
            a0 = aL[sa];
            a1 = aL[sa + 32];
            b0 = x0L[sx];
            b1 = x1L[sx];
            ++j;
            const bool wrap = j == period;                      // uniform
            sx += stepX + (wrap ? wrapX : 0);
            sa += stepA + (wrap ? wrapA : 0);
            j = wrap ? 0 : j;
        };
        // ping-pong operand registers (P/Q), two k-steps per iteration: no register copies, so the
        // wait before a step's MFMAs covers only that step's own ds_reads.  The operands of step ks+1 are
        // fetched from LDS BEFORE the MFMAs of step ks are issued; the fetch past the last step is unused.
        float pa0, pa1, pb0, pb1, qa0, qa1, qb0, qb1;
        fetch(pa0, pa1, pb0, pb1);
        int nloop = nks;
        if (nks & 1) {   // odd step count: peel one step so the unrolled loop stays branch-free
            fetch(qa0, qa1, qb0, qb1);
            acc[0][0] = pase_mfma_32x32x2(pa0, pb0, acc[0][0]);
            acc[0][1] = pase_mfma_32x32x2(pa0, pb1, acc[0][1]);
            acc[1][0] = pase_mfma_32x32x2(pa1, pb0, acc[1][0]);
            acc[1][1] = pase_mfma_32x32x2(pa1, pb1, acc[1][1]);
            pa0 = qa0; pa1 = qa1; pb0 = qb0; pb1 = qb1;
            --nloop;
        }
        // Interleave: the ds_reads of the NEXT step are spread between the four MFMAs of the current step
#define PASE_STEP_SCHED()                                             \
        PASE_SGB(0x008, 1); PASE_SGB(0x100, 1);                       \
        PASE_SGB(0x008, 1); PASE_SGB(0x100, 1);                       \
        PASE_SGB(0x008, 1); PASE_SGB(0x100, 1);                       \
        PASE_SGB(0x008, 1);
        for (int ks = 0; ks < nloop; ks += 2) {
            fetch(qa0, qa1, qb0, qb1);
            // All MFMAs are issued unconditionally: rows / columns beyond the tile edge multiply
            // zero-filled A rows or finite staged data and are discarded in the epilogue.
            acc[0][0] = pase_mfma_32x32x2(pa0, pb0, acc[0][0]);
            acc[0][1] = pase_mfma_32x32x2(pa0, pb1, acc[0][1]);
            acc[1][0] = pase_mfma_32x32x2(pa1, pb0, acc[1][0]);
            acc[1][1] = pase_mfma_32x32x2(pa1, pb1, acc[1][1]);
            PASE_STEP_SCHED();
            PASE_SCHED_BARRIER();
            fetch(pa0, pa1, pb0, pb1);
            acc[0][0] = pase_mfma_32x32x2(qa0, qb0, acc[0][0]);
            acc[0][1] = pase_mfma_32x32x2(qa0, qb1, acc[0][1]);
            acc[1][0] = pase_mfma_32x32x2(qa1, qb0, acc[1][0]);
            acc[1][1] = pase_mfma_32x32x2(qa1, qb1, acc[1][1]);
            PASE_STEP_SCHED();
            PASE_SCHED_BARRIER();
        }
#undef PASE_STEP_SCHED
        PASE_TACC(1);
        if (g + 1 < g_end) {
            store_stage(cur ^ 1);
            kg = kg_next;
            tbe = tbe_next;
        }
        PASE_TACC(2);
        __syncthreads();
        PASE_TACC(3);
    }
    PASE_TACC_DUMP();

    // ---- epilogue ---------------------------------------------------------------------------
    // D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  All offsets are 32-bit element
    // offsets off a uniform base (host-checked), rows advance by compile-time constants.
    PASE_STAMP(3);
    const int rbase = m0 + wm * 64 + 4 * (lane >> 5);
    int cs[2], cq[2];
    bool cok[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int jj = wn * 64 + b * 32 + fr;
        cok[b] = jj < ncols_valid;
        if (pl.mode == MODE_FLAT) {
            const unsigned n = (unsigned)(n0 + jj);
            int s = (int)div_magic(n, pl.ncols_magic);
            int u = (int)n - s * p.Ncols;
            if (u < 0) { --s; u += p.Ncols; }
            cs[b] = cok[b] ? s : 0;
            cq[b] = cok[b] ? u : 0;
        } else {
            cs[b] = jj < lenA ? s0 : s0 + 1;
            cq[b] = jj < lenA ? qA + jj : jj - lenA;
        }
    }
    const bool rows_full = m0 + wm * 64 + 64 <= p.M;   // uniform

    if (p.epilogue == PASE_EPI_STORE &&
        (p.post_op == PASE_POST_POW || p.post_op == PASE_POST_LOGPOW || p.post_op == PASE_POST_MAG)) {
        // spectra: accumulator rows r, r+1 (same lane) are the (re, im) parts of one frequency bin
        int cbase[2];
        bool colok[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int pos = cq[b] + p.poff;
            cbase[b] = (cs[b] * p.y_ctot + p.y_coff) * p.Tout + pos;
            colok[b] = cok[b] && pos >= 0 && pos < p.Tout;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int m = rbase + a * 32 + (r & 3) + 8 * (r >> 2);   // even row
                if (m >= p.M) continue;          // M is even (host-checked): a pair is valid or absent
                const int rowoff = (m >> 1) * p.Tout;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float re = acc[a][b][r], im = acc[a][b][r + 1];
                    float v = re * re + im * im;
                    v = (p.post_op == PASE_POST_LOGPOW) ? p.post_scale * logf(v + p.post_eps)
                        : (p.post_op == PASE_POST_MAG ? p.post_scale * sqrtf(v) : v * p.post_scale);
                    if (colok[b]) p.y[(unsigned)(cbase[b] + rowoff)] = v;
                }
            }
        }
    } else if (p.epilogue == PASE_EPI_STORE) {
        const bool pshuf = p.ps != 1;
        const float* biasp = (p.bias && split == 0) ? p.bias : nullptr;
        int cbase[2], posb[2];
        bool colok[2];
        bool interior = true;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            posb[b] = cq[b] * p.ps + p.poff;
            cbase[b] = (cs[b] * p.y_ctot + p.y_coff) * p.Tout + posb[b];
            colok[b] = cok[b] && (pshuf || (posb[b] >= 0 && posb[b] < p.Tout));
            interior = interior && cok[b] && posb[b] >= 0 && posb[b] + p.ps <= p.Tout;
        }
        // FAST (wave-uniform): every (row, column, phase) of this wave's 64x64 block is stored, no
        // post-op -> no predicates at all in the unrolled body.  ATOMIC: split-K partial tile.
        auto store_rows = [&](auto fast_tag, auto atomic_tag) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_tag)::value;
            constexpr bool ATOMIC = decltype(atomic_tag)::value;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                // the block's 16 bias values first (independent loads; see mse_rows on why not inside the store loop)
                float bvs[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bvs[r] = 0.f;
                if (biasp) {   // uniform (data-gradients carry no bias: no index arithmetic for them)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
                        int co = m;
                        if (pshuf) co = (X6 && pl.xPerm) ? (int)div_magic((unsigned)m, pl.ps_magic)
                                                         : m - (int)div_magic((unsigned)m, pl.cout_magic) * p.Cout_store;
                        if (FAST || m < p.M) bvs[r] = biasp[co];
                    }
                }
                if constexpr (X6 != 0 && FAST && !ATOMIC) {
                    if (pl.xPerm) {   // uniform: (channel, phase)-ordered rows -> runs of consecutive output samples
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const int m4 = rbase + a * 32 + 8 * g4;               // rows m4 .. m4 + 3 (m4 % 4 == 0)
                            const int co0 = (int)div_magic((unsigned)m4, pl.ps_magic);
                            const int ph0 = m4 - co0 * p.ps;
                            const int n1 = min(4, p.ps - ph0);                   // samples left in channel co0
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                float v[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = acc[a][b][4 * g4 + i] + bvs[4 * g4 + i];
                                float* d0 = p.y + (unsigned)(cbase[b] + co0 * p.Tout + ph0);
                                if (n1 == 4) {
                                    pase_store_run4(d0, v);
                                } else {          // the quad straddles two channels: n1 samples, then 4 - n1 of co0 + 1
                                    float* d1 = p.y + (unsigned)(cbase[b] + (co0 + 1) * p.Tout);
                                    if (n1 == 2) {
                                        pase_store_run2(d0, v[0], v[1]);
                                        pase_store_run2(d1, v[2], v[3]);
                                    } else if (n1 == 1) {
                                        d0[0] = v[0];
                                        pase_store_run2(d1, v[1], v[2]);
                                        d1[2] = v[3];
                                    } else {
                                        pase_store_run2(d0, v[0], v[1]);
                                        d0[2] = v[2];
                                        d1[0] = v[3];
                                    }
                                }
                            }
                        }
                        continue;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
                    const bool mok = FAST || m < p.M;
                    int ph = 0, co = m;
                    if (pshuf) {   // uniform
                        if (X6 && pl.xPerm) {
                            co = (int)div_magic((unsigned)m, pl.ps_magic);
                            ph = m - co * p.ps;
                        } else {
                            ph = (int)div_magic((unsigned)m, pl.cout_magic);
                            co = m - ph * p.Cout_store;
                        }
                    }
                    const float bv = bvs[r];
                    const int rowoff = co * p.Tout + ph;
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        float v = acc[a][b][r] + bv;
                        if (!FAST && p.post_op == PASE_POST_LOG) v = p.post_scale * logf(v == 0.f ? p.post_eps : v);
                        if (!FAST && p.post_op == PASE_POST_RELU) v = fmaxf(v, 0.f);
                        if (!FAST && p.post_op == PASE_POST_SQRTPOS) v = sqrtf(fmaxf(v, 0.f));
                        const bool ok = FAST || (mok && colok[b] && (!pshuf || (unsigned)(posb[b] + ph) < (unsigned)p.Tout));
                        if (ok) {
                            float* dst = p.y + (unsigned)(cbase[b] + rowoff);
                            if (ATOMIC) atomicAdd(dst, v);
                            else *dst = v;
                            s1 += v;
                            s2 += v * v;
                        }
                    }
                    if (!ATOMIC && p.stat_part) {   // uniform branch
                        s1 = pase_half_sum_lane31(s1);
                        s2 = pase_half_sum_lane31(s2);
                        if (fr == 31) {
                            const int ml = m - m0;
                            red[wn][ml][0] = s1;
                            red[wn][ml][1] = s2;
                        }
                    }
                }
            }
        };
        const bool fast = rows_full && p.post_op == PASE_POST_NONE && pase_wave_all(interior) != 0;
        if (pl.splitk > 1) {
            if (fast) store_rows(std::true_type{}, std::true_type{});
            else store_rows(std::false_type{}, std::true_type{});
        } else {
            if (fast) store_rows(std::true_type{}, std::false_type{});
            else store_rows(std::false_type{}, std::false_type{});
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
        int lbase[2], obase[2], tb[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            tb[b] = cq[b] - half;
            lbase[b] = cs[b] * p.label_D * p.Ncols + tb[b];
            obase[b] = cs[b] * p.M * p.Ncols + cq[b];
        }
        // Two passes per 32-row block: first ALL the block's label / bias loads (32 independent loads in flight), then
        // the arithmetic and the stores.  Written as one loop the compiler has to keep every load behind the previous
        // row's grad_out store (it cannot prove the two float* do not alias), which serialises 64 L2 round trips per
        // lane: 9 us of a 56 us workgroup on the 21 525-row heads.
        auto mse_rows = [&](auto fast_tag) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float tg[16][2], bvs[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
                    const bool mok = FAST || m < p.M;
                    const int d = (int)div_magic((unsigned)m, pl.rctx_magic);
                    const int jj = m - d * p.r_ctx;
                    bvs[r] = (mok && p.bias) ? p.bias[m] : 0.f;
                    const int lrow = d * p.Ncols + jj;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        tg[r][b] = 0.f;
                        if ((FAST || (mok && cok[b])) && (unsigned)(tb[b] + jj) < (unsigned)p.Ncols)
                            tg[r][b] = p.label[(unsigned)(lbase[b] + lrow)];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
                    const bool mok = FAST || m < p.M;
                    const int orow = m * p.Ncols;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        if (FAST || (mok && cok[b])) {
                            const float pred = acc[a][b][r] + bvs[r];
                            const float diff = pred - tg[r][b];
                            lsum += diff * diff;
                            const unsigned o = (unsigned)(obase[b] + orow);
                            if (p.y) p.y[o] = pred;
                            if (p.grad_out) p.grad_out[o] = diff * p.grad_scale;
                        }
                    }
                }
            }
        };
        if (rows_full && pase_wave_all(cok[0] && cok[1]) != 0) mse_rows(std::true_type{});
        else mse_rows(std::false_type{});
        lsum = pase_wave_sum64(lsum);
        if (lane == 0) red[0][wave][0] = lsum;
        __syncthreads();
        if (tid == 0) {
            const double t = (double)red[0][0][0] + (double)red[0][1][0] + (double)red[0][2][0] + (double)red[0][3][0];
            atomicAdd(p.loss_acc, t);
        }
    }
#ifdef PASE_TRACE
    __builtin_amdgcn_s_waitcnt(0);   // stores acknowledged
    PASE_STAMP(4);
    if (threadIdx.x == 0 && blockIdx.x < PASE_TRACE_SLOTS) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_trace[blockIdx.x * 6 + 5] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

#ifdef PASE_TRACE
extern "C" int pase_debug_trace(unsigned long long* host, int nslots) {
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 6 * (size_t)nslots);
}
extern "C" int pase_debug_tacc(unsigned long long* host, int nslots) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tacc), sizeof(unsigned long long) * 4 * (size_t)nslots);
}
#endif

// ---- K-major weight pack: wt[k, m] = A[m, k], k = ci*taps + kk, A[m, (ci,kk)] = w[m*ldw + (tap_major ?
// kk*Cin + ci : ci*taps + kk)]; columns m >= M of a row are zero-filled up to ldwt.  32x32 LDS transpose.
__global__ void __launch_bounds__(256) pack_wt_kernel(const float* w, float* wt, int M, int K, int Cin, int taps,
                                                      int ldw, int tap_major, int ldwt) {
    __shared__ float tile[32][33];
    const int kt = blockIdx.x * 32, mt = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mt + ty + 8 * i, k = kt + tx;
        float v = 0.f;
        if (m < M && k < K) {
            int src = k;
            if (tap_major) {
                const int ci = k / taps, kk = k - ci * taps;
                src = kk * Cin + ci;
            }
            v = w[(size_t)m * ldw + src];
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = kt + ty + 8 * i, m = mt + tx;
        if (k < K && m < ldwt) wt[(size_t)k * ldwt + m] = tile[tx][ty + 8 * i];
    }
}

struct HostPlan {
    ConvPlan pl;
    int BN, narrow;
    long blocks;
    int n_col_tiles;
    long x6_chunks;     // 16-byte chunks of the split-bf16 pack (0: fp32 plan)
    bool x6c;           // the launch runs on conv_x6c.hip (channel-minor split-bf16 kernel, two accumulators per tile)
    PaseX6cPlan c;
    bool sinc;          // the launch runs on sinc_x6.hip (one input channel: window-image split-bf16 kernel)
    PaseSincPlan sp;
};

unsigned magic_of(int d) {
    return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned long long)d);
}

// Split-bf16 launches run on conv_x6c.hip (two accumulators per tile, round-to-nearest pieces); a launch without a plan
// there runs on the exact-fp32 matrix pipe.  (Round 2's span-major split instantiations of conv_gemm_kernel -- one accumulator,
// truncated pieces: systematically biased, see the conv_x6c.hip header -- are no longer built.)
HostPlan make_plan(const PaseConvGemm& p, bool want_x6) {
    HostPlan h;
    h.x6_chunks = 0;
    h.x6c = false;
    h.sinc = false;
    if (want_x6 && !(p.x6_ctl & 8) && pase_sinc_x6_plan(p, h.sp)) {      // (x6_ctl bit 3: A/B runs keep the layer on the fp32 pipe)
        h.sinc = true;
        h.pl = ConvPlan{};
        h.pl.CB = 16;
        h.pl.splitk = 1;
        h.narrow = 1;
        h.BN = 256;
        h.n_col_tiles = p.S * h.sp.tiles_per_seq;
        h.x6_chunks = h.sp.pack_bytes / 16;
        h.blocks = h.n_col_tiles;
        return h;
    }
    if (want_x6) {
        if (pase_x6c_plan(p, h.c)) {
            h.x6c = true;
            h.pl = ConvPlan{};
            h.pl.x6 = 1;
            h.pl.CB = 16;
            h.pl.splitk = h.c.splitk;
            h.narrow = h.c.WM == 2;
            h.BN = h.c.BN;
            h.n_col_tiles = h.c.n_col_tiles;
            h.x6_chunks = h.c.pack_bytes / 16;
            h.blocks = (long)h.c.n_row_tiles * h.c.n_col_tiles * h.c.splitk;
            return h;
        }
        want_x6 = false;
    }
    h.narrow = (p.tile_hint == 64) || (p.tile_hint == 0 && p.M <= 64);
    const int BM = h.narrow ? 64 : 128;
    h.BN = h.narrow ? 256 : 128;
    ConvPlan& pl = h.pl;
    // flat (1x1 as a plain GEMM over the flattened (s, q) columns) needs float4 slots: 4 consecutive columns
    // stay inside one sequence and are 16-B aligned.  Other 1x1 shapes run as a one-tap convolution.
    // (a single tap has no direction: the 1x1 data-gradients arrive with tapstep = -1 like every transposed conv)
    const bool flat = (p.taps == 1 && p.stride == 1 && p.padL == 0) && (p.Ncols % 4) == 0 &&
                      (p.Tin % 4) == 0 && (((unsigned long long)(size_t)p.x) % 16) == 0;
    pl.mode = flat ? MODE_FLAT : (p.Ncols >= h.BN ? MODE_SEG : MODE_PERSEQ);
    pl.xvec = flat ? 1 : 0;
    pl.pmajor = pl.xvec;
    pl.x6 = pl.xR = pl.xTq = pl.xTS = pl.xSteps = pl.xTaps = 0;
    // (the 4-step slab costs 12 KB of LDS and 12 VGPRs more: only where 3 steps cannot hold the padded taps;
    //  64-row tiles have a split-bf16 instantiation for the one-row plan only -- the Sinc FIR)
    if (flat) {
        pl.TB = 1;
        pl.SPANV = pl.SPAN = h.BN;
        pl.CB = XS_FLAT / h.BN;
        if (pl.CB > KG_FLAT) pl.CB = KG_FLAT;
        if (pl.CB > p.Cin) pl.CB = p.Cin;
        if (pl.CB > 1) pl.CB &= ~1;
        const int TPR = h.BN / 4;
        pl.tl = 0;
        while ((1 << pl.tl) < TPR) ++pl.tl;
        pl.nslots = NS_FLAT;
    } else {
        pl.TB = p.taps <= KGMAX ? p.taps : 32;
        // a tile may touch two sequences: each segment carries its own halo of TB samples
        pl.SPANV = (h.BN - 1) * p.stride + (pl.mode == MODE_SEG ? 2 : 1) * pl.TB;
        pl.CB = KGMAX / pl.TB;
        if (pl.CB > p.Cin) pl.CB = p.Cin;
        // one slab row per thread: TPR = largest power of two with 256/TPR >= CB; Q samples each.  CB is kept
        // even (the MFMA loop pairs channel rows) unless it is 1; one slack sample per row (SPANV + 1) keeps the
        // zero-weight tap of an odd single-row stage inside the slab.
        pl.nslots = XPT + 1;
        pl.tl = 8;
        for (; pl.CB >= 1; --pl.CB) {
            if (pl.CB > 1 && (pl.CB & 1)) continue;
            pl.tl = 8;
            while ((NTHREADS >> pl.tl) < pl.CB) --pl.tl;
            pl.nslots = (pl.SPANV + 1 + (1 << pl.tl) - 1) >> pl.tl;
            if (pl.nslots <= XPT) break;
        }
        if (pl.CB < 1) pl.CB = 0;   // span does not fit: rejected by the caller
        pl.nslots = pl.nslots <= 3 ? 3 : (pl.nslots <= 6 ? 6 : 12);   // kernel instantiations: 3 / 6 / 12 slots
        pl.SPAN = pl.nslots << pl.tl;
    }
    pl.n_gc = pl.CB ? (p.Cin + pl.CB - 1) / pl.CB : 0;
    pl.n_gt = (p.taps + pl.TB - 1) / pl.TB;
    const int RA = NTHREADS / (BM / 4);
    pl.PA = (pl.CB * pl.TB + RA - 1) / RA;
    pl.tiles_per_seq = (p.Ncols + h.BN - 1) / h.BN;
    pl.ncols_magic = magic_of(p.Ncols);
    pl.cout_magic = magic_of(p.Cout_store);
    pl.ps_magic = magic_of(p.ps);
    pl.xPerm = (pl.x6 && p.ps > 1 && p.epilogue == PASE_EPI_STORE && !p.stat_part && p.post_op == PASE_POST_NONE &&
                p.M == p.ps * p.Cout_store) ? 1 : 0;
    pl.rctx_magic = magic_of(p.r_ctx);
    const long ntot = (long)p.S * p.Ncols;
    h.n_col_tiles = (pl.mode == MODE_PERSEQ) ? p.S * pl.tiles_per_seq : (int)((ntot + h.BN - 1) / h.BN);
    const long tiles = (long)((p.M + BM - 1) / BM) * h.n_col_tiles;
    int splitk = 1;
    const int G = pl.n_gc * pl.n_gt;
    if (p.splitk > 1) splitk = p.splitk;
    else if (p.splitk == 0 && !p.stat_part && p.epilogue == PASE_EPI_STORE && p.post_op == PASE_POST_NONE && G >= 12) {
        // auto (data-gradients, transposed convs, plain 1x1s): the grid runs in rounds of 512 workgroup slots
        // (2 per CU) and a nearly empty last round costs as much as a half-full one (a lone workgroup on a CU
        // runs 1.85x faster than two co-resident ones).  Pick the split minimising
        // rounds x (reduction share + atomic tile flush) per workgroup; the flush (64 atomics per lane on a
        // caller-zeroed output) is worth ~4 stages.
        const double flush = 4.0 / (double)G;
        double best = 1e30;
        const int max_split = G / 6 < 1 ? 1 : G / 6;
        // workgroup slots: 2 per CU, 3 for the small-footprint instantiation (see PASE_CONV_LAUNCH)
        const long slots = (!h.narrow && !pl.xvec && !pl.x6 && pl.nslots == 3 && pl.CB * pl.TB <= 44) ? 768 : 512;
        for (int sk = 1; sk <= max_split && sk <= 64; ++sk) {
            const long W = tiles * sk;
            const long full = W / slots, tail = W % slots;
            const double tc = tail == 0 ? 0.0 : (tail <= 256 ? 0.55 : 1.0);
            const double est = ((double)full + tc) * (1.0 / sk + (sk > 1 ? flush : 0.0));
            if (est < best * 0.97) { best = est; splitk = sk; }     // a larger split must win by 3 %
        }
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

extern "C" int pase_pack_wt(const float* w, float* wt, int M, int K, int Cin, int taps, int ldw, int tap_major,
                            int ldwt, void* stream) {
    if (M <= 0 || K <= 0) return 0;
    if (ldwt < M || (ldwt & 3) || K != Cin * taps) return -4;
    PASE_LAUNCH(pack_wt_kernel, dim3((unsigned)((K + 31) / 32), (unsigned)((ldwt + 31) / 32)), dim3(256),
                (hipStream_t)stream, w, wt, M, K, Cin, taps, ldw, tap_major, ldwt);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_conv_gemm(const PaseConvGemm* d, void* stream) {
    const PaseConvGemm p = *d;
    if (p.M <= 0 || p.K <= 0 || p.S <= 0 || p.Ncols <= 0) return 0;
    if (p.K != p.Cin * p.taps) return -4;
    // (a split-bf16 launch reads only its pack wx6: the K-major fp32 pack is the fp32-pipe kernels' operand)
    if (!p.wx6 && (!p.wt || p.ldwt < p.M || p.ldwt < 4 || (p.ldwt & 3) || (((unsigned long long)(size_t)p.wt) % 16) != 0)) return -10;
    if (p.epilogue == PASE_EPI_MSE_CTX && (!p.label || !p.loss_acc || p.r_ctx < 1)) return -2;
    if (p.pad_mode == PASE_PAD_REFLECT && (p.padL >= p.Tin)) return -3;
    if (p.tapstep != 1 && p.tapstep != -1) return -5;
    const HostPlan h = make_plan(p, p.wx6 != nullptr);
    if (h.pl.CB < 1) return -6;
    if (p.wx6 && ((!h.x6c && !h.sinc) || (((unsigned long long)(size_t)p.wx6) % 16) != 0)) return -11;
    if (h.pl.splitk > 1 && (p.stat_part || p.epilogue != PASE_EPI_STORE || p.post_op != PASE_POST_NONE)) return -7;
    if ((p.post_op == PASE_POST_POW || p.post_op == PASE_POST_LOGPOW || p.post_op == PASE_POST_MAG) &&
        (p.ps != 1 || p.stat_part || (p.M & 1))) return -9;
    // 32-bit element offsets in the loader and the epilogue
    const long LIM = 0x7fffffffL;
    if ((long)p.S * p.Ncols >= LIM) return -8;
    if ((long)p.S * p.x_ctot * (long)p.Tin >= LIM) return -8;
    if ((long)(p.K + KGMAX) * p.ldwt >= LIM) return -8;
    if (p.epilogue == PASE_EPI_STORE && (long)p.S * p.y_ctot * (long)p.Tout + (long)p.ps * p.Ncols >= LIM) return -8;
    if (p.epilogue == PASE_EPI_MSE_CTX &&
        ((long)p.S * p.M * (long)p.Ncols >= LIM || (long)p.S * p.label_D * (long)p.Ncols >= LIM)) return -8;
    if (p.ps != 1 && (long)p.M * p.Cout_store >= 0xffffffffL) return -8;      // exact magic division
    if (p.epilogue == PASE_EPI_MSE_CTX && (long)p.M * p.r_ctx >= 0xffffffffL) return -8;
    hipStream_t st = (hipStream_t)stream;
    if (h.sinc) return pase_sinc_x6_launch(p, h.sp, st);
    if (h.x6c) return pase_x6c_launch(p, h.c, st);
    const dim3 grid((unsigned)h.blocks), block(NTHREADS);
#define PASE_CONV_LAUNCH(BM_, BN_)                                                                       \
    do {                                                                                                 \
        if (h.pl.xvec && !p.in_scale && !p.in_alpha)                                                     \
            PASE_LAUNCH((conv_gemm_kernel<BM_, BN_, NS_FLAT, 1, 2, 0>), grid, block, st, p, h.pl);       \
        else if (h.pl.xvec && !p.in_scale)                                                               \
            PASE_LAUNCH((conv_gemm_kernel<BM_, BN_, NS_FLAT, 1, 2, 1>), grid, block, st, p, h.pl);       \
        else if (h.pl.xvec) PASE_LAUNCH((conv_gemm_kernel<BM_, BN_, NS_FLAT, 1, 2, 2>), grid, block, st, p, h.pl);  \
        else if (h.pl.nslots == 3 && BM_ == 128 && h.pl.CB * h.pl.TB <= 44)                              \
            PASE_LAUNCH((conv_gemm_kernel<128, 128, 3, 0, 3>), grid, block, st, p, h.pl);                \
        else if (h.pl.nslots == 3) PASE_LAUNCH((conv_gemm_kernel<BM_, BN_, 3, 0>), grid, block, st, p, h.pl);  \
        else if (h.pl.nslots == 6) PASE_LAUNCH((conv_gemm_kernel<BM_, BN_, 6, 0>), grid, block, st, p, h.pl);  \
        else PASE_LAUNCH((conv_gemm_kernel<BM_, BN_, 12, 0>), grid, block, st, p, h.pl);                 \
    } while (0)
    if (h.narrow) PASE_CONV_LAUNCH(64, 256);
    else PASE_CONV_LAUNCH(128, 128);
#undef PASE_CONV_LAUNCH
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_conv_gemm_stat_tiles(const PaseConvGemm* d) {
    return make_plan(*d, d->wx6 != nullptr).n_col_tiles;
}

extern "C" int pase_conv_gemm_splitk(const PaseConvGemm* d) {
    return make_plan(*d, d->wx6 != nullptr).pl.splitk;
}

extern "C" int pase_conv_gemm_plan_kind(const PaseConvGemm* d) {
    if (d->M <= 0 || d->K <= 0 || d->S <= 0 || d->Ncols <= 0 || d->K != d->Cin * d->taps) return 0;
    const HostPlan h = make_plan(*d, d->wx6 != nullptr);
    return h.sinc ? 3 : (h.x6c ? 2 : 0);
}

// Which kernel instantiation the launch runs, for reports (bench.py's roofline.dominant matches it against the kernel names
// of a rocprofv3 trace) and tests: 0 = exact-fp32 matrix pipe (conv_gemm_kernel), 1 = sinc_x6_fwd_kernel, otherwise
// conv_x6c_kernel<NPOS, KGS, false, ZP, NARROW, SYM, DUO> encoded as NPOS * 1000 + KGS * 100 + 8 * DUO + 4 * SYM + 2 * ZP + NARROW
extern "C" int pase_conv_gemm_kernel_id(const PaseConvGemm* d) {
    if (d->M <= 0 || d->K <= 0 || d->S <= 0 || d->Ncols <= 0 || d->K != d->Cin * d->taps) return 0;
    const HostPlan h = make_plan(*d, d->wx6 != nullptr);
    if (h.sinc) return 1;
    if (!h.x6c) return 0;
    const bool narrow = h.c.WM == 2;
    const int npos = narrow ? 320 : (h.c.A == 1 ? 128 : 192), kgs = narrow ? 2 : (h.c.A == 1 ? 3 : 2);
    return npos * 1000 + kgs * 100 + (h.c.xp ? 2 : 0) + (narrow ? 1 : 0) + ((h.c.sym && h.c.xp) ? 4 : 0) + ((h.c.sym == 2 && h.c.xp) ? 8 : 0);
}

extern "C" long pase_conv_gemm_x6_bytes(const PaseConvGemm* d) {
    if (d->M <= 0 || d->K <= 0 || d->S <= 0 || d->Ncols <= 0 || d->K != d->Cin * d->taps) return 0;
    if (d->tapstep != 1 && d->tapstep != -1) return 0;
    const HostPlan h = make_plan(*d, true);
    return h.pl.CB >= 1 ? h.x6_chunks * 16 : 0;
}

extern "C" long pase_conv_gemm_xp_bytes(const PaseConvGemm* d) {
    if (d->M <= 0 || d->K <= 0 || d->S <= 0 || d->Ncols <= 0 || d->K != d->Cin * d->taps) return 0;
    if (d->tapstep != 1 && d->tapstep != -1) return 0;
    const HostPlan h = make_plan(*d, true);
    return h.x6c ? h.c.xp_plane * 3 * 16 : 0;
}

extern "C" int pase_pack_xp(const PaseConvGemm* d, void* stream) {
    const PaseConvGemm p = *d;
    if (!p.xp6 || !p.x || (((unsigned long long)(size_t)p.xp6) % 16) != 0) return -10;
    if (p.K != p.Cin * p.taps) return -4;
    const HostPlan h = make_plan(p, true);
    if (!h.x6c) return -11;
    return pase_x6c_pack_xp(p, h.c, (hipStream_t)stream);
}

extern "C" int pase_pack_x6(const PaseConvGemm* d, void* stream) {
    const PaseConvGemm p = *d;
    if (!p.wx6 || (!p.wt && !p.w) || (((unsigned long long)(size_t)p.wx6) % 16) != 0) return -10;
    if (p.K != p.Cin * p.taps || (p.wt && p.ldwt < p.M)) return -4;
    const HostPlan h = make_plan(p, true);
    if (h.sinc) return pase_sinc_x6_pack(p, h.sp, (hipStream_t)stream);
    if (!h.x6c) return -11;
    return pase_x6c_pack(p, h.c, (hipStream_t)stream);
}

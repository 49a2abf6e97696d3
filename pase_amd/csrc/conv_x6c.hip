// conv_x6c.hip -- split-bf16 implicit-GEMM convolution, channel-minor form ("x6c"): the kernel behind every
// PaseConvGemm launch that carries a split-bf16 weight pack (PaseConvGemm::wx6) and has a 16-channel k-group to work on.
//
//   Y[s, row, q] = sum_{ci,kk} A[row, (ci,kk)] * act(bn(X[s, ci, q*stride + kk*tapstep - padL]))
//
// (reference ops: nn.Conv1d of FeBlock pase/models/modules.py:1058-1077, the 1x1 convs of MLPBlock :527-556 and
// frontend.py:182,195, nn.ConvTranspose1d of GDeconv1DBlock :558-589 and every strided data-gradient as
// pixel-shuffle stores, torchqrnn's Linear over [x_t ; x_{t-1}].)
//
// Arithmetic.  Every fp32 operand is the sum of three bf16 pieces hi + mid + lo, each piece ROUNDED TO NEAREST
// (v_cvt_pk_bf16_f32; the remainders x - hi, x - hi - mid are exact), and a product is evaluated as
// hh + hm + mh + hl + lh + mm on v_mfma_f32_32x32x16_bf16.  The matrix core aligns its 16 products and the
// accumulator to the largest exponent and drops the bits below its internal width TOWARDS -INFINITY
// (tools/experiments/mfma_round_probe.hip), so the five small terms must not be added to the large running sum --
// that is a systematic error of -2e-6 * K/2048 of the sum of |a b| (tools/experiments/x6_accum_probe.hip: the round-2
// kernels).  Here hh accumulates in accH and the five small terms in accS (2^-8 of accH: their dropped bits are
// 2^-8 smaller), added once in the epilogue: mean error / rms error 0.002, relative L2 against fp64 2.7e-7 at
// K = 2048 and 8.2e-7 at K = 16 384, against 8.0e-7 / 2.3e-6 for the k-ordered fp32 fma chain of v_mfma_f32_32x32x2_f32.
//
// Mapping.  A strided convolution is first viewed as a stride-1 convolution over P = stride polyphase channels:
// channel' c' = ci * P + b, tap' a: kk = a * P + b, Xp[c'][j] = X[ci][P * j + b - padL], Y[q] = sum W'[c', a] Xp[c'][q + a]
// (taps past the real count have zero weights).  A 16-deep MFMA step is then 16 CHANNELS' AT ONE TAP: lane (col j,
// half fk) needs the eight channels' 8 fk .. 8 fk + 7 at position j + a.  The activation stage is therefore laid out
// CHANNEL-MINOR in LDS: three bf16 planes of 16-byte chunks [plane][fk][position] = 8 channels' of one position, so
// that a B fragment is ONE ds_read_b128 per plane, conflict-free (consecutive lanes = consecutive chunks), and the
// next tap is the next chunk.  The operand split (and the on-load BatchNorm affine + PReLU, padding, two-sequence
// logic) is done ONCE per staged element, by the staging threads, and shared by every tap, every row tile of the
// workgroup and every wave -- the round-2 kernel redid it in each wave for every fragment it read.
// The weight operand never touches LDS: pase_pack_x6 stores it in fragment order [32-row tile][step][plane][lane],
// so a wave's A fragment is one coalesced 1 KB global_load_dwordx4 per plane, prefetched one step ahead, and each
// of a workgroup's waves owns different rows (4 x 1 wave layout: no redundant loads).
//
// Tile.  256 threads = 4 waves; wave tile 32 rows x 128 columns (4 B tiles x 2 accumulators = 128 VGPRs); workgroup
// 128 x 128 (waves 4 x 1) or 64 x 256 (waves 2 x 2, for M <= 64).  LDS: double-buffered stages of KGS (1 or 2)
// 16-channel' groups, 2 workgroups per CU.  One barrier per stage (KGS * taps' steps of 24 MFMAs per wave); the next
// stage's global loads are issued a stage ahead into registers and converted into the other LDS buffer in slices
// between the steps of the current stage.
#include <cstdlib>
#include <type_traits>

#include "conv_x6c.h"

namespace {

constexpr int NT = 256;
constexpr int HALO_MAX = 64;      // extra positions a stage holds beyond its BN columns (halo of every sequence touched)
constexpr int KGS_MAX = 2;

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
__device__ __forceinline__ unsigned div_magic(unsigned e, unsigned magic) {
    return magic ? (unsigned)(((unsigned long long)e * magic) >> 32) : e;
}

__device__ float g_identc[2] = {1.f, 0.f};

template <int WM, int NBT>
__global__ void __launch_bounds__(NT, 2) conv_x6c_kernel(PaseConvGemm p, PaseX6cPlan pl) {
    constexpr int WN = 4 / WM;
    constexpr int BM = 32 * WM, BN = 32 * NBT * WN;
    constexpr int NPOS = BN + HALO_MAX;            // positions (16-byte chunks) per (plane, fk) row
    constexpr int NPS = (NPOS + 127) / 128;        // position slots per thread and k-group
    constexpr int KGS_T = (BN <= 128) ? KGS_MAX : 1;
    constexpr int NSLOT = NPS * KGS_T;
    constexpr int PLANE = 2 * NPOS;                // chunks per plane of one k-group: [fk][pos]
    constexpr int KGC = 3 * PLANE;                 // chunks per k-group
    constexpr int BUF = KGS_T * KGC;               // chunks per stage buffer
    __shared__ __attribute__((aligned(16))) u32x4 Xs[2 * BUF];
    float (*red)[BM][2] = reinterpret_cast<float (*)[BM][2]>(Xs);       // epilogue scratch (stage buffers are dead)
    static_assert(sizeof(float) * WN * BM * 2 <= sizeof(u32x4) * 2 * BUF, "epilogue scratch");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = pase_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fk = lane >> 5;
    const int fkL = wave >> 1;                     // loader: octet of the k-group this wave stages (uniform)
    const int posL = tid & 127;                    // loader: position within a 128-position slot

    // ---- tile decode -----------------------------------------------------------------------
    const int ntot = p.S * p.Ncols;
    const int ntiles = pl.n_row_tiles * pl.n_col_tiles;
    const int split = blockIdx.x / ntiles;
    const int tile = xcd_swizzle(blockIdx.x % ntiles, ntiles);
    const int mt = tile % pl.n_row_tiles;
    const int nt = tile / pl.n_row_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int s0 = (int)div_magic((unsigned)n0, pl.ncols_magic);
    const int qA = n0 - s0 * p.Ncols;
    const int lenA = min(BN, p.Ncols - qA);
    const int ncols_valid = min(BN, ntot - n0);
    const int H = pl.A - 1;                        // halo positions per sequence segment
    const int segL = p.Ncols + H;                  // span positions of a full middle segment
    const int nseg = (int)div_magic((unsigned)(n0 + ncols_valid - 1), pl.ncols_magic) - s0 + 1;
    const int span_len = ncols_valid + nseg * H;
    const int KGS = KGS_T > 1 ? pl.KGS : 1;
    const int nps_run = (span_len + 127) >> 7;     // position slots that hold data (a 1x1 launch has no halo: one slot less)

    // ---- stage range of this split ----------------------------------------------------------
    const int GS = (pl.G + KGS - 1) / KGS;         // stages
    const int g_per = (GS + pl.splitk - 1) / pl.splitk;
    const int g_begin = split * g_per;
    const int g_end = min(GS, g_begin + g_per);
    if (g_begin >= g_end) return;                  // uniform for the whole block, before any barrier
    const int nsteps = KGS * pl.A;                 // MFMA steps per stage

    // ---- loader state: position slot ps -> span index i = posL + 128 ps -> (sequence, time of tap 0) ----
    int slot_u0[NPS], slot_off[NPS];
    unsigned slot_valid = 0u, slot_inter = 0u;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
        const int i = posL + 128 * ps;
        bool valid = i < span_len;
        int k, r;
        if (i < lenA + H) {
            k = 0;
            r = i;
        } else {
            const int d = i - (lenA + H);
            const int k1 = (int)div_magic((unsigned)d, pl.seg_magic);
            k = 1 + k1;
            r = d - k1 * segL;
        }
        const int q = (k == 0 ? qA : 0) + r;
        const int s = s0 + k;
        valid = valid && s < p.S;
        const int u0 = valid ? pl.P * q - pl.padLp : 0;
        slot_u0[ps] = u0;
        slot_off[ps] = valid ? s * p.x_ctot * p.Tin : 0;
        if (valid) slot_valid |= 1u << ps;
        if (!valid || (u0 >= 0 && u0 + pl.P - 1 < p.Tin)) slot_inter |= 1u << ps;
    }
    // wave-uniform: every element this wave stages is an in-range sample (no padding arithmetic in the loader)
    const bool all_inter = pase_wave_all(slot_inter == ((1u << NPS) - 1u)) != 0;

    const bool has_xf = p.in_scale != nullptr || p.in_alpha != nullptr;       // uniform
    const float* sc_p = p.in_scale ? p.in_scale : &g_identc[0];
    const float* sh_p = p.in_scale ? p.in_shift : &g_identc[1];
    const float* al_p = p.in_alpha ? p.in_alpha : &g_identc[0];
    const int aff_on = p.in_scale ? 1 : 0, alpha_on = p.in_alpha ? 1 : 0;

    float xreg[NSLOT][8];
    unsigned xmask[NSLOT];         // bit e: element e of the slot is a real sample of a real channel

    // channel' -> (input channel, phase) of element e of this wave's octet in k-group kg of stage g (all uniform)
    auto chan_of = [&](int g, int kg, int e, int& ci, int& b, bool& ok) __attribute__((always_inline)) {
        const int cp = (g * KGS + kg) * 16 + fkL * 8 + e;
        ok = cp < pl.CinP;
        const int cq = (int)div_magic((unsigned)cp, pl.p_magic);
        b = cp - cq * pl.P;
        ci = min(cq, p.Cin - 1);
    };
    auto load_slot = [&](auto sl_tag, int g) __attribute__((always_inline)) {
        constexpr int sl = decltype(sl_tag)::value;
        constexpr int kg = sl / NPS, ps = sl % NPS;
        const bool v = (slot_valid >> ps) & 1u;
        unsigned mask = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int ci, b;
            bool chok;
            chan_of(g, kg, e, ci, b, chok);
            const int choff = (p.x_coff + ci) * p.Tin;
            int off;
            bool ok;
            if (all_inter) {
                off = slot_off[ps] + choff + slot_u0[ps] + (v ? b : 0);
                ok = v;
            } else {
                int u = slot_u0[ps] + b;
                if (p.pad_mode == PASE_PAD_REFLECT) {
                    if (u < 0) u = -u;
                    if (u >= p.Tin) u = 2 * (p.Tin - 1) - u;
                }
                ok = v && u >= 0 && u < p.Tin;
                off = ok ? slot_off[ps] + choff + u : choff;
            }
            xreg[sl][e] = p.x[(unsigned)off];
            if (ok && chok) mask |= 1u << e;
        }
        xmask[sl] = mask;
    };
    // registers of slot sl (stage g) -> on-load transform -> three bf16 planes -> LDS buffer bsel
    auto store_slot = [&](auto sl_tag, int g, int bsel) __attribute__((always_inline)) {
        constexpr int sl = decltype(sl_tag)::value;
        constexpr int kg = sl / NPS, ps = sl % NPS;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = xreg[sl][e];
            if (has_xf) {
                int ci, b;
                bool chok;
                chan_of(g, kg, e, ci, b, chok);
                t = fmaf(t, sc_p[ci * aff_on], sh_p[ci * aff_on]);
                t = t > 0.f ? t : t * al_p[ci * alpha_on];
            }
            v[e] = ((xmask[sl] >> e) & 1u) ? t : 0.f;      // zero padding applies AFTER the transform
        }
        u32x4 o[3];
        pase_split_bf16x3_rne(v, o);
        const int i = posL + 128 * ps;
        if (NPS * 128 == NPOS || i < NPOS) {
            u32x4* dst = &Xs[bsel * BUF + kg * KGC + fkL * NPOS + i];
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) dst[pz * PLANE] = o[pz];
        }
    };

    // ---- B fragment bases: chunk index of (column, tap 0) inside a (plane, fk) row -----------------------
    int bbase[NBT];
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
        const int c = (wn * NBT + j) * 32 + fr;
        const int s = (int)div_magic((unsigned)(n0 + c), pl.ncols_magic);
        int i = c + (s - s0) * H;
        i = min(i, NPOS - 1 - H);                  // columns past the end of the data: any staged position will do
        bbase[j] = fk * NPOS + i;
    }

    f32x16 accH[NBT], accS[NBT];
#pragma unroll
    for (int j = 0; j < NBT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accH[j][r] = 0.f;
            accS[j][r] = 0.f;
        }

    // ---- A fragments: [32-row tile][step][plane][lane] 16-byte chunks, one step ahead ---------------------
    const u32x4* ap = reinterpret_cast<const u32x4*>(p.wx6) +
                      ((size_t)(mt * WM + wm) * (unsigned)pl.steps_total + (size_t)g_begin * (unsigned)nsteps) * 192u + lane;
    auto load_a = [&](u32x4 (&a)[3]) __attribute__((always_inline)) {
        a[0] = ap[0];
        a[1] = ap[64];
        a[2] = ap[128];
        ap += 192;
    };
    auto mfma_step = [&](const u32x4 (&a)[3], const u32x4* xb) __attribute__((always_inline)) {
        // plane pairs of the five small terms, smallest first: mm, hl, lh, hm, mh -> accS; hh -> accH
        constexpr int PZA[5] = {1, 0, 2, 0, 1}, PZB[5] = {1, 2, 0, 1, 0};
#pragma unroll
        for (int jp = 0; jp < NBT; jp += 2) {
            u32x4 b0[3], b1[3];
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) {
                b0[pz] = xb[pz * PLANE + bbase[jp]];
                b1[pz] = xb[pz * PLANE + bbase[jp + 1]];
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

    // ---- prologue ---------------------------------------------------------------------------------------
    u32x4 a0[3], a1[3];
    pase_static_for<NSLOT>([&](auto sl) __attribute__((always_inline)) {
        if (decltype(sl)::value / NPS < KGS && decltype(sl)::value % NPS < nps_run) load_slot(sl, g_begin);
    });
    load_a(a0);
    pase_static_for<NSLOT>([&](auto sl) __attribute__((always_inline)) {
        if (decltype(sl)::value / NPS < KGS && decltype(sl)::value % NPS < nps_run) store_slot(sl, g_begin, 0);
    });
    if (g_begin + 1 < g_end)
        pase_static_for<NSLOT>([&](auto sl) __attribute__((always_inline)) {
            if (decltype(sl)::value / NPS < KGS && decltype(sl)::value % NPS < nps_run) load_slot(sl, g_begin + 1);
        });
    __syncthreads();

    // ---- main loop: stage g = KGS k-groups x A taps; step st = kg * A + t ---------------------------------
    int g = g_begin, st = 0, kg = 0, t = 0, bsel = 0;
    const int nslot_run = NPS * KGS;
    bool done = false;
    auto step = [&](const u32x4 (&acur)[3], u32x4 (&anxt)[3]) __attribute__((always_inline)) {
        const bool last = (g == g_end - 1) && (st == nsteps - 1);            // uniform
        if (!last) load_a(anxt);
        // a slice of the next stage: slot sl is converted at step sl (mod nsteps) of the current stage and its
        // registers are refilled with the stage after that
        if (g + 1 < g_end) {
            pase_static_for<NSLOT>([&](auto sl_tag) __attribute__((always_inline)) {
                constexpr int sl = decltype(sl_tag)::value;
                if (sl < nslot_run && sl % NPS < nps_run && (nsteps >= nslot_run ? sl : sl % nsteps) == st) {
                    store_slot(sl_tag, g + 1, bsel ^ 1);
                    if (g + 2 < g_end) load_slot(sl_tag, g + 2);
                }
            });
        }
        mfma_step(acur, &Xs[bsel * BUF + kg * KGC + t]);
        ++st;
        if (++t == pl.A) {
            t = 0;
            ++kg;
        }
        if (st == nsteps) {
            st = 0;
            kg = 0;
            ++g;
            __syncthreads();
            bsel ^= 1;
        }
        done = last;
    };
    while (true) {
        step(a0, a1);
        if (done) break;
        step(a1, a0);
        if (done) break;
    }

    // ---- accumulators: hh + (the five small terms) -------------------------------------------------------
    f32x16 (&acc)[NBT] = accH;
#pragma unroll
    for (int j = 0; j < NBT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] += accS[j][r];

    // ---- epilogue -----------------------------------------------------------------------------------------
    // D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int rbase = m0 + wm * 32 + 4 * fk;
    int cs[NBT], cq[NBT];
    bool cok[NBT];
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
        const int jj = (wn * NBT + j) * 32 + fr;
        cok[j] = jj < ncols_valid;
        const unsigned n = (unsigned)(n0 + jj);
        const int s = (int)div_magic(n, pl.ncols_magic);
        cs[j] = cok[j] ? s : 0;
        cq[j] = cok[j] ? (int)n - s * p.Ncols : 0;
    }
    const bool rows_full = m0 + wm * 32 + 32 <= p.M;        // uniform

    if (p.epilogue == PASE_EPI_STORE &&
        (p.post_op == PASE_POST_POW || p.post_op == PASE_POST_LOGPOW || p.post_op == PASE_POST_MAG)) {
        // spectra: accumulator rows r, r + 1 (same lane) are the (re, im) parts of one frequency bin
#pragma unroll
        for (int j = 0; j < NBT; ++j) {
            const int pos = cq[j] + p.poff;
            const int cbase = (cs[j] * p.y_ctot + p.y_coff) * p.Tout + pos;
            const bool colok = cok[j] && pos >= 0 && pos < p.Tout;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int m = rbase + (r & 3) + 8 * (r >> 2);
                if (m >= p.M) continue;                    // M is even (host-checked)
                const float re = acc[j][r], im = acc[j][r + 1];
                float v = re * re + im * im;
                v = (p.post_op == PASE_POST_LOGPOW) ? p.post_scale * logf(v + p.post_eps)
                    : (p.post_op == PASE_POST_MAG ? p.post_scale * sqrtf(v) : v * p.post_scale);
                if (colok) p.y[(unsigned)(cbase + (m >> 1) * p.Tout)] = v;
            }
        }
    } else if (p.epilogue == PASE_EPI_STORE) {
        const bool pshuf = p.ps != 1;
        const float* biasp = (p.bias && split == 0) ? p.bias : nullptr;
        int cbase[NBT], posb[NBT];
        bool colok[NBT];
        bool interior = true;
#pragma unroll
        for (int j = 0; j < NBT; ++j) {
            posb[j] = cq[j] * p.ps + p.poff;
            cbase[j] = (cs[j] * p.y_ctot + p.y_coff) * p.Tout + posb[j];
            colok[j] = cok[j] && (pshuf || (posb[j] >= 0 && posb[j] < p.Tout));
            interior = interior && cok[j] && posb[j] >= 0 && posb[j] + p.ps <= p.Tout;
        }
        float bvs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bvs[r] = 0.f;
        if (biasp) {   // uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rbase + (r & 3) + 8 * (r >> 2);
                int co = m;
                if (pshuf) co = pl.xPerm ? (int)div_magic((unsigned)m, pl.ps_magic)
                                         : m - (int)div_magic((unsigned)m, pl.cout_magic) * p.Cout_store;
                if (m < p.M) bvs[r] = biasp[co];
            }
        }
        auto store_rows = [&](auto fast_tag, auto atomic_tag) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_tag)::value;
            constexpr bool ATOMIC = decltype(atomic_tag)::value;
            if constexpr (FAST && !ATOMIC) {
                if (pl.xPerm) {   // uniform: (channel, phase)-ordered rows -> runs of consecutive output samples
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int m4 = rbase + 8 * g4;                          // rows m4 .. m4 + 3 (m4 % 4 == 0)
                        const int co0 = (int)div_magic((unsigned)m4, pl.ps_magic);
                        const int ph0 = m4 - co0 * p.ps;
                        const int n1 = min(4, p.ps - ph0);                      // samples left in channel co0
#pragma unroll
                        for (int j = 0; j < NBT; ++j) {
                            float v[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = acc[j][4 * g4 + i] + bvs[4 * g4 + i];
                            float* d0 = p.y + (unsigned)(cbase[j] + co0 * p.Tout + ph0);
                            if (n1 == 4) {
                                pase_store_run4(d0, v);
                            } else {          // the quad straddles two channels
                                float* d1 = p.y + (unsigned)(cbase[j] + (co0 + 1) * p.Tout);
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
                    return;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rbase + (r & 3) + 8 * (r >> 2);
                const bool mok = FAST || m < p.M;
                int ph = 0, co = m;
                if (pshuf) {   // uniform
                    if (pl.xPerm) {
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
                for (int j = 0; j < NBT; ++j) {
                    float v = acc[j][r] + bv;
                    if (!FAST && p.post_op == PASE_POST_LOG) v = p.post_scale * logf(v == 0.f ? p.post_eps : v);
                    if (!FAST && p.post_op == PASE_POST_RELU) v = fmaxf(v, 0.f);
                    if (!FAST && p.post_op == PASE_POST_SQRTPOS) v = sqrtf(fmaxf(v, 0.f));
                    const bool ok = FAST || (mok && colok[j] && (!pshuf || (unsigned)(posb[j] + ph) < (unsigned)p.Tout));
                    if (ok) {
                        float* dst = p.y + (unsigned)(cbase[j] + rowoff);
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
            for (int ml = tid; ml < BM; ml += NT) {
                const int m = m0 + ml;
                if (m < p.M) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int w = 0; w < WN; ++w) {
                        s1 += red[w][ml][0];
                        s2 += red[w][ml][1];
                    }
                    float* dst = p.stat_part + ((size_t)nt * p.M + m) * 2;
                    dst[0] = s1;
                    dst[1] = s2;
                }
            }
        }
    } else {  // PASE_EPI_MSE_CTX: rows m = d * r + j, columns (b, t); target = label[b, d, t + j - r / 2]
        float lsum = 0.f;
        const int half = p.r_ctx / 2;
        // Two passes per 16-row block: first ALL its label / bias loads (independent loads in flight), then the
        // arithmetic and the stores (the compiler cannot prove label and grad_out do not alias).
        auto mse_rows = [&](auto fast_tag) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_tag)::value;
            float bvs[16];
            int lrow[16];
            int jj16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rbase + (r & 3) + 8 * (r >> 2);
                const bool mok = FAST || m < p.M;
                const int d = (int)div_magic((unsigned)m, pl.rctx_magic);
                jj16[r] = m - d * p.r_ctx;
                bvs[r] = (mok && p.bias) ? p.bias[m] : 0.f;
                lrow[r] = d * p.Ncols + jj16[r];
            }
#pragma unroll
            for (int j = 0; j < NBT; ++j) {
                const int tb = cq[j] - half;
                const int lbase = cs[j] * p.label_D * p.Ncols + tb;
                const int obase = cs[j] * p.M * p.Ncols + cq[j];
                float tg[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rbase + (r & 3) + 8 * (r >> 2);
                    const bool mok = FAST || m < p.M;
                    tg[r] = 0.f;
                    if ((FAST || (mok && cok[j])) && (unsigned)(tb + jj16[r]) < (unsigned)p.Ncols)
                        tg[r] = p.label[(unsigned)(lbase + lrow[r])];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rbase + (r & 3) + 8 * (r >> 2);
                    const bool mok = FAST || m < p.M;
                    if (FAST || (mok && cok[j])) {
                        const float pred = acc[j][r] + bvs[r];
                        const float diff = pred - tg[r];
                        lsum += diff * diff;
                        const unsigned o = (unsigned)(obase + m * p.Ncols);
                        if (p.y) p.y[o] = pred;
                        if (p.grad_out) p.grad_out[o] = diff * p.grad_scale;
                    }
                }
            }
        };
        bool allc = true;
#pragma unroll
        for (int j = 0; j < NBT; ++j) allc = allc && cok[j];
        if (rows_full && pase_wave_all(allc) != 0) mse_rows(std::true_type{});
        else mse_rows(std::false_type{});
        lsum = pase_wave_sum64(lsum);
        if (lane == 0) red[0][wave][0] = lsum;
        __syncthreads();
        if (tid == 0) {
            const double tsum = (double)red[0][0][0] + (double)red[0][1][0] + (double)red[0][2][0] + (double)red[0][3][0];
            atomicAdd(p.loss_acc, tsum);
        }
    }
}

// weights (K-major fp32 pack wt[k * ldwt + m], k = ci * taps + kk) -> fragment-ordered bf16 planes:
// out[((rt32 * steps + st) * 3 + plane) * 64 + lane], step st = g * A + a, lane = (fk, row): element e = channel'
// 16 g + 8 fk + e at tap' a.  Zero for channels' past Cin * P, taps past the real count and rows past M.
__global__ void pack_x6c_kernel(const float* __restrict__ wt, u32x4* __restrict__ out, int M, int ldwt, int Cin, int taps,
                                int P, int A, int CinP, int rev, int steps, long total, int perm_ps, int perm_cout) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long rs = idx >> 6;
        const int st = (int)(rs % steps);
        const int rt = (int)(rs / steps);
        const int g = st / A, a = st - g * A;
        const int fk = lane >> 5, row = lane & 31;
        const int m = rt * 32 + row;
        const int msrc = perm_ps > 1 ? (m % perm_ps) * perm_cout + m / perm_ps : m;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cp = g * 16 + fk * 8 + e;
            const int ci = cp / P, b = cp - ci * P;
            const int kk = rev ? taps - 1 - a : a * P + b;
            v[e] = (cp < CinP && kk >= 0 && kk < taps && m < M) ? wt[((size_t)ci * taps + kk) * ldwt + msrc] : 0.f;
        }
        u32x4 o[3];
        pase_split_bf16x3_rne(v, o);
        u32x4* dst = out + ((size_t)rs * 3) * 64 + lane;
#pragma unroll
        for (int pz = 0; pz < 3; ++pz) dst[pz * 64] = o[pz];
    }
}

unsigned magic_of(int d) {
    return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned long long)d);
}

}  // namespace

bool pase_x6c_plan(const PaseConvGemm& p, PaseX6cPlan& pl) {
    if (p.tapstep == 1) {
        pl.P = p.stride;
        pl.rev = 0;
        pl.padLp = p.padL;
    } else if (p.tapstep == -1 && p.stride == 1) {
        pl.P = 1;
        pl.rev = 1;
        pl.padLp = p.padL + p.taps - 1;
    } else {
        return false;
    }
    if (pl.P < 1) return false;
    pl.A = (p.taps + pl.P - 1) / pl.P;
    if (pl.A > 16) return false;
    pl.CinP = p.Cin * pl.P;
    if (pl.CinP < 16) return false;
    pl.G = (pl.CinP + 15) / 16;
    if (pl.G * 16 * 4 > pl.CinP * 5) return false;          // more than 25 % zero channels'
    if (const char* e = getenv("PASE_X6C")) {
        if (e[0] == '0') return false;
    }
    // tile: 64 x 256 (waves 2 x 2) for M <= 64, else 128 x 128 (waves 4 x 1)
    pl.NBT = 4;
    pl.WM = p.M <= 64 ? 2 : 4;
    pl.BM = 32 * pl.WM;
    pl.BN = 32 * pl.NBT * (4 / pl.WM);
    // every sequence a column tile touches carries its own halo of A - 1 positions
    const int nseg_max = (pl.BN - 2) / p.Ncols + 2;
    if ((long)nseg_max * (pl.A - 1) > HALO_MAX) return false;
    // two k-groups per stage where a stage would otherwise be shorter than four steps (128-column tile only)
    pl.KGS = (pl.BN <= 128 && pl.A < 4 && pl.G >= 2) ? 2 : 1;
    const int GS = (pl.G + pl.KGS - 1) / pl.KGS;
    pl.steps_total = GS * pl.KGS * pl.A;
    const long ntot = (long)p.S * p.Ncols;
    pl.n_row_tiles = (p.M + pl.BM - 1) / pl.BM;
    pl.n_col_tiles = (int)((ntot + pl.BN - 1) / pl.BN);
    pl.pack_chunks = (long)pl.n_row_tiles * pl.WM * pl.steps_total * 192;
    pl.ncols_magic = magic_of(p.Ncols);
    pl.cout_magic = magic_of(p.Cout_store);
    pl.ps_magic = magic_of(p.ps);
    pl.rctx_magic = magic_of(p.r_ctx);
    pl.seg_magic = magic_of(p.Ncols + pl.A - 1);
    pl.p_magic = magic_of(pl.P);
    pl.xPerm = (p.ps > 1 && p.epilogue == PASE_EPI_STORE && !p.stat_part && p.post_op == PASE_POST_NONE &&
                p.M == p.ps * p.Cout_store) ? 1 : 0;
    // split-K (data-gradients of the wide heads, decoder layers on few columns): rounds of 512 workgroup slots
    const long tiles = (long)pl.n_row_tiles * pl.n_col_tiles;
    int splitk = 1;
    if (p.splitk > 1) splitk = p.splitk;
    else if (p.splitk == 0 && !p.stat_part && p.epilogue == PASE_EPI_STORE && p.post_op == PASE_POST_NONE && GS >= 8) {
        const double flush = 2.0 / (double)GS;               // the atomic tile flush is worth ~2 stages
        double best = 1e30;
        const int max_split = GS / 4 < 1 ? 1 : GS / 4;
        const long slots = 512;
        for (int sk = 1; sk <= max_split && sk <= 64; ++sk) {
            const long W = tiles * sk;
            const long full = W / slots, tail = W % slots;
            const double tc = tail == 0 ? 0.0 : (tail <= 256 ? 0.55 : 1.0);
            const double est = ((double)full + tc) * (1.0 / sk + (sk > 1 ? flush : 0.0));
            if (est < best * 0.97) {
                best = est;
                splitk = sk;
            }
        }
    }
    if (splitk > GS) splitk = GS;
    if (splitk > 1) {   // every split owns at least one stage
        const int g_per = (GS + splitk - 1) / splitk;
        splitk = (GS + g_per - 1) / g_per;
    }
    pl.splitk = splitk;
    return true;
}

int pase_x6c_pack(const PaseConvGemm& p, const PaseX6cPlan& pl, hipStream_t st) {
    const long total = pl.pack_chunks / 3;
    const long nb = (total + 255) / 256;
    PASE_LAUNCH(pack_x6c_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), st, p.wt,
                reinterpret_cast<u32x4*>(const_cast<void*>(p.wx6)), p.M, p.ldwt, p.Cin, p.taps, pl.P, pl.A, pl.CinP, pl.rev,
                pl.steps_total, total, pl.xPerm ? p.ps : 1, p.Cout_store);
    PASE_CHECK_LAUNCH();
    return 0;
}

int pase_x6c_launch(const PaseConvGemm& p, const PaseX6cPlan& pl, hipStream_t st) {
    const dim3 grid((unsigned)((long)pl.n_row_tiles * pl.n_col_tiles * pl.splitk)), block(NT);
    if (pl.WM == 2) PASE_LAUNCH((conv_x6c_kernel<2, 4>), grid, block, st, p, pl);
    else PASE_LAUNCH((conv_x6c_kernel<4, 4>), grid, block, st, p, pl);
    PASE_CHECK_LAUNCH();
    return 0;
}

// conv_x6c.h -- internal interface between conv_gemm.hip (the C-ABI entry points pase_conv_gemm / pase_pack_x6 /
// pase_conv_gemm_x6_bytes / ..._stat_tiles / ..._splitk) and conv_x6c.hip (the channel-minor split-bf16 kernel).
// Not part of the ABI.
#pragma once
#include "hip_compat.h"
#include "pase_amd.h"

struct PaseX6cPlan {
    int P;              // polyphase factor (= stride): channel' c' = ci * P + b reads x[ci][P * j + b - padLp]
    int A;              // taps of the stride-1 view: ceil(taps / P)
    int CinP;           // Cin * P channels'
    int G;              // 16-channel' k-groups
    int KGS;            // k-groups per stage (1 or 2)
    int padLp;          // left pad of the stride-1 view
    int rev;            // taps reversed in the pack (tapstep == -1)
    int WM, NBT;        // waves along M (4 / 2), 32-column B tiles per wave (8 / 4)
    int BM, BN;
    int n_row_tiles, n_col_tiles, splitk;
    int steps_total;    // stages * KGS * A MFMA steps (16 k each; k-groups past G are zero in the pack)
    int epi32;          // output (and label) below 2 GiB: the lean epilogues address them with 32-bit byte offsets
    int stagger;        // start phase spacing of the persistent workgroups (units of 512 clocks; 0 = none)
    int prio;           // s_setprio of the staging waves (bits 0-1) and of the compute waves (bits 2-3)
    int tmode;          // 0: convolution.  Weight gradients (contraction over positions): 1 rows = g (packed), columns =
                        // (channel, tap) of z (staged);  2 1x1 swapped: rows = z channels (packed), columns = g rows (staged);
                        // 3 rows = (channel, tap) read at 2-byte granularity from row-major bf16 planes of z, columns = g rows
    int t_taps, t_tapstep, t_padL, t_stride;   // tmode 3: the layer's taps (the staged operand's descriptor says taps = 1)
    int t_lseg, t_hh, t_dmin;                  // tmode 3: plane row = S segments of t_lseg = 16 QP16 + t_hh elements
    int t_vec;                                 // T-mode: the staged operand's 8-position chunks are 16-byte aligned
    int zp;                                    // tmode 1 with the staged operand PRE-SPLIT ("ZP"): the (channel, tap) columns are
                                               // COPIED out of phase-decomposed bf16 planes of z~ (pack_zph_kernel), no conversion
    int zp_n, zp_rem;                          // ... taps = stride * zp_n + zp_rem: GEMM column r of a channel is tap
                                               //     kk0 + stride * dd (taps of ONE phase adjacent: zp_tap_of in conv_x6c.hip)
    int zp_rows;                               // ... rows of a plane: Cin * stride (+ 1: the all-ones row of the bias column)
    long zp_off;                               // ... byte offset of the planes inside PaseWgrad::gx6 (behind the pack of g)
    long t_plane;                              // tmode 3: elements per plane
    int xp;             // convolution launches: the activation is pre-split (PaseConvGemm::xp6, pase_pack_xp): staging = copy
    int xp_tpad;        // ... padded positions per sequence: Ncols + A - 1
    long xp_plane;      // ... 16-byte chunks per plane: G * 2 * S * xp_tpad
    int sym;            // convolution launches on a pre-split activation: the symmetric form (256 x 128 tile, no staging waves)
    int pairs;          // strided convolution launches with an even stride: interior slots load (phase, phase + 1) pairs
    int xPerm;          // pixel-shuffle launches: tile rows ordered (channel, phase) -> 16-byte output runs
    long pack_chunks;   // 16-byte chunks of the weight pack
    int prm_n;          // channels' of the expanded on-load parameter arrays behind the chunks (3 x prm_n floats)
    long pack_bytes;    // pack_chunks * 16 + 3 * prm_n * 4, rounded up to 16
    unsigned ncols_magic, cout_magic, rctx_magic, ps_magic, seg_magic, p_magic;
};

// false: the launch has no x6c plan (the caller falls back to the span-major split-bf16 / fp32 kernels)
bool pase_x6c_plan(const PaseConvGemm& p, PaseX6cPlan& pl);
int pase_x6c_pack(const PaseConvGemm& p, const PaseX6cPlan& pl, hipStream_t st);
int pase_x6c_launch(const PaseConvGemm& p, const PaseX6cPlan& pl, hipStream_t st);
int pase_x6c_pack_xp(const PaseConvGemm& p, const PaseX6cPlan& pl, hipStream_t st);

// weight gradients on the T-mode instantiation (see conv_x6c.hip)
struct PaseX6cWgrad {
    PaseConvGemm pc;      // the staged operand and the output, in the convolution descriptor's terms
    PaseX6cPlan pl;
    int swapped;
    const float* a_src;   // the packed operand: rows x positions
    int a_rows, a_ctot, a_coff, a_T;
    const float *a_sc, *a_sh, *a_al;
};
bool pase_x6c_wgrad_plan(const PaseWgrad& w, PaseX6cWgrad& o);
int pase_x6c_wgrad_launch(const PaseWgrad& w, const PaseX6cWgrad& o, hipStream_t st);

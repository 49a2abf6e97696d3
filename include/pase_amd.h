/* pase_amd.h -- C ABI of libpase_hip.so: the hand-written gfx950 (MI355X) kernels behind the
 * PASE / PASE+ self-supervised training step.
 *
 * The reference (santi-pdp/pase) has no FFI layer: its hot path is Python calling torch ops
 * (cuDNN conv1d / conv-transpose1d / batch-norm, cuBLAS, and the third-party torchqrnn CUDA
 * ForgetMult).  This header is the boundary a maintainer would bind instead: every entry point
 * names the reference op (file:line under /root/reference) whose arithmetic it replaces.
 *
 * Conventions:
 *   - all tensors fp32, contiguous, NCT = (batch, channels, time) exactly like the reference;
 *   - the CALLER owns every buffer (in practice the PyTorch caching allocator); the library never
 *     allocates, frees or synchronises; scratch is passed in;
 *   - every call is asynchronous on the HIP stream passed as `stream` (a hipStream_t);
 *   - return value 0 = launched, >0 = hipError_t, <0 = argument error; functions are stateless and
 *     re-entrant;
 *   - activations are stored RAW (pre-BatchNorm / pre-PReLU); consumers apply
 *     `v = x*scale[c] + shift[c]; v = v > 0 ? v : alpha[c]*v` on load (any of the three arrays
 *     may be NULL = identity).
 */
#ifndef PASE_AMD_H
#define PASE_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

enum { PASE_PAD_ZERO = 0, PASE_PAD_REFLECT = 1 };
enum { PASE_EPI_STORE = 0, PASE_EPI_MSE_CTX = 1 };

/* ------------------------------------------------------------------------------------------
 * pase_conv_gemm -- implicit-GEMM 1-D convolution on v_mfma_f32_32x32x2_f32.
 *
 *   out[s, row, q] = bias[row % Cout_store] + sum_{ci,kk} w[row, k(ci,kk)] * X~[s, ci, q*stride + kk*tapstep - padL]
 *   stored at y[s, y_coff + row % Cout_store, q*ps + row / Cout_store + poff]   (if 0 <= pos < Tout)
 *
 * Replaces: nn.Conv1d inside FeBlock (pase/models/modules.py:1047-1051,1058-1077, reflect pad
 * :1061-1071), SincConv_fast's F.conv1d (:932), the 1x1 convs (frontend.py:182,195,262;
 * modules.py:543; Minions/minions.py:510), nn.ConvTranspose1d of GDeconv1DBlock (modules.py:571-575,
 * as a pixel-shuffle store with ps = stride), torchqrnn's Linear over [x_t ; x_{t-1}]
 * (tap_major = 1, taps = 2, tapstep = -1), and -- with the transposed weight pack from
 * pase_pack_dgrad -- every data-gradient of the above.  With epilogue = PASE_EPI_MSE_CTX it also
 * replaces ContextualizedLoss(nn.MSELoss(), r) (pase/losses.py:6-37) on the regression workers.
 * ------------------------------------------------------------------------------------------ */
typedef struct PaseConvGemm {
    const float* x;        /* input  (S, x_ctot, Tin); channels [x_coff, x_coff+Cin) are read      */
    const float* w;        /* A operand, row-major (M, ldw); column k = ci*taps+kk (or kk*Cin+ci)  */
    float* y;              /* output (S, y_ctot, Tout) (EPI_STORE) / prediction (S, M, Ncols) or NULL (EPI_MSE_CTX) */
    const float* bias;     /* (Cout_store) or NULL                                                */
    const float* in_scale; /* (Cin) on-load affine, NULL = identity                               */
    const float* in_shift; /* (Cin)                                                               */
    const float* in_alpha; /* (Cin) on-load PReLU slope, NULL = none                              */
    float* stat_part;      /* (n_col_tiles, M, 2) per-tile (sum, sumsq) of the stored values, or NULL */
    const float* label;    /* EPI_MSE_CTX: (S, label_D, Ncols) target                             */
    float* grad_out;       /* EPI_MSE_CTX: (S, M, Ncols) d(loss)/d(pred) = (pred-tgt)*grad_scale, or NULL */
    double* loss_acc;      /* EPI_MSE_CTX: += sum (pred-tgt)^2  (caller zeroes)                   */
    float grad_scale;
    int S, Cin, Tin, x_ctot, x_coff;
    int M, K, ldw, taps, tap_major;
    int stride, tapstep, padL, pad_mode;
    int Ncols;             /* GEMM columns per sequence                                           */
    int y_ctot, y_coff, Cout_store, ps, poff, Tout;
    int epilogue, r_ctx, label_D;
    int tile_hint;         /* 0 = auto, 64 = 64x256 block tile, 128 = 128x128                     */
} PaseConvGemm;

int pase_conv_gemm(const PaseConvGemm* desc, void* stream);
/* number of column tiles (= first dim of stat_part) the launch above will use */
int pase_conv_gemm_stat_tiles(int M, int S, int Ncols, int tile_hint);

/* sizeof() of the ABI structs (0 = PaseConvGemm), for binding self-checks */
int pase_abi_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* PASE_AMD_H */

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
/* post-ops of the PASE_EPI_STORE epilogue (on-device target DSP: spectra as DFT-basis convolutions) */
enum { PASE_POST_NONE = 0,
       PASE_POST_POW = 1,     /* rows come in (re, im) pairs: out[row/2] = (re^2 + im^2) * post_scale            */
       PASE_POST_LOGPOW = 2,  /* out[row/2] = post_scale * ln(re^2 + im^2 + post_eps)                          */
       PASE_POST_LOG = 3,     /* out[row] = post_scale * ln(v == 0 ? post_eps : v)                             */
       PASE_POST_MAG = 4,     /* pairs: out[row/2] = post_scale * sqrt(re^2 + im^2)   (SWIPE' magnitude spectra)  */
       PASE_POST_RELU = 5,    /* out[row] = max(v, 0)                                                          */
       PASE_POST_SQRTPOS = 6 };/* out[row] = sqrt(max(v, 0))          (SWIPE' loudness at the ERB frequencies)  */
enum { PASE_LOSS_NONE = 0, PASE_LOSS_L1 = 1, PASE_LOSS_MSE = 2, PASE_LOSS_BCE_LOGITS = 3 };

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
    const float* w;        /* A operand as the reference stores it, row-major (M, ldw); column k = ci*taps+kk
                              (or kk*Cin+ci when tap_major): the SOURCE of the pack below, not read by the kernel */
    const float* wt;       /* K-major pack of A the kernel reads: wt[k*ldwt + m] = A[m, k], k = ci*taps+kk
                              (pase_pack_wt; a weight that already is K-major, e.g. W (Cout, Cin) as the A of a
                              1x1 data-gradient, can be passed as is).  16-B aligned, ldwt % 4 == 0, ldwt >= M  */
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
    int M, K, ldw, ldwt, taps, tap_major;
    int stride, tapstep, padL, pad_mode;
    int Ncols;             /* GEMM columns per sequence                                           */
    int y_ctot, y_coff, Cout_store, ps, poff, Tout;
    int epilogue, r_ctx, label_D;
    int tile_hint;         /* 0 = auto, 64 = 64x256 block tile, 128 = 128x128                     */
    int post_op;           /* PASE_POST_* (EPI_STORE, ps == 1, no stat_part / bias for the pair ops)       */
    float post_scale, post_eps;
    int splitk;            /* 1: none; >1: split the reduction, partial tiles atomically added into a
                              caller-zeroed y (EPI_STORE without stat_part only); 0: library decides --
                              query pase_conv_gemm_splitk() and zero y when it returns > 1        */
    const void* wx6;       /* NULL: contraction on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32).  Else: the
                              split-bf16 pack of A written by pase_pack_x6() for THIS descriptor
                              (pase_conv_gemm_x6_bytes() > 0): every fp32 operand is the exact sum of three
                              truncated bf16 pieces hi + mid + lo and a product is evaluated as
                              hh + hm + mh + hl + lh + mm on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
                              (dropped terms <= 3 * 2^-24 |a b|: the error of an fp32 fma chain).  wt is then
                              only the source of the pack.                                        */
    const void* xp6;       /* split-bf16 launches only, NULL = the kernel splits the activation while staging it.  Else: the
                              PRE-SPLIT activation written by pase_pack_xp() for THIS descriptor (pase_conv_gemm_xp_bytes()
                              > 0): channel-minor bf16 planes with the on-load transform and the padding applied, in the
                              chunk order of the kernel's LDS image -- staging becomes a copy.  The library asks for it on
                              stride-1 launches with one or two taps and >= 1024 rows, where a column tile is re-staged by
                              every row tile (the 256 -> 21 525 heads: 169 times).  Where it is asked for and the launch
                              has >= 128 rows, the weight pack is laid out for the symmetric kernel forms (x6_ctl bit 7),
                              which have no other way to stage: pase_conv_gemm then returns -12 when xp6 is NULL  */
    int x6_ctl;            /* split-bf16 plan control (0 = the library's routing).  bit 0: take the split-bf16 kernel
                              wherever it has a plan, skipping the measured per-shape routing rules; bit 1: ask for the
                              pre-split activation on every stride-1 launch; bit 2: never; bit 3: keep the one-channel
                              (SincNet) layer off its window-image kernel; bit 4: no 64 x 256 tile for launches of at
                              most 64 rows; bit 5: the general epilogue on every tile (no lean store / MSE path); bit 6: the bias
                              added in the epilogue instead of being the accumulators' initial value; bit 7: launches on a
                              pre-split activation stay on the staging-wave form (no symmetric form); bit 16: strided launches
                              with an even stride load single samples while staging (default: pairs of phases); bits 17 / 18: the
                              eight-wave / the four-wave symmetric form wherever eligible (A/B runs); bits 8-15: start
                              the persistent workgroups n x 512 clocks out of phase (A/B runs and tests; the library
                              itself reads NO environment variables)                                             */
    int max_wg;            /* cap on the persistent grid of the split-bf16 kernel (0 = one workgroup per CU, 256):
                              data-parallel runs leave CUs to the RCCL channel kernels this way; tests use it to make
                              every workgroup walk several (split-K slice, tile) items                       */
} PaseConvGemm;

int pase_conv_gemm(const PaseConvGemm* desc, void* stream);
/* wt (K, ldwt) <- transpose of the logical A (M, K) held in w (row-major, ldw, optional tap-major columns);
 * columns M..ldwt-1 are zero-filled.  Weights change every optimizer step, so the training step re-packs
 * each weight once per use (29.7 M parameters: ~0.1 ms of HBM traffic per step). */
int pase_pack_wt(const float* w, float* wt, int M, int K, int Cin, int taps, int ldw, int tap_major, int ldwt,
                 void* stream);
/* number of column tiles (= first dim of stat_part) the launch described by desc will use */
int pase_conv_gemm_stat_tiles(const PaseConvGemm* desc);
/* the split-K factor the launch will actually use (after clamping) */
int pase_conv_gemm_splitk(const PaseConvGemm* desc);
/* which kernel family the launch described by desc runs on: 0 = exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32),
 * 2 = split-bf16 channel-minor kernel (conv_x6c.hip; needs desc->wx6), 3 = the one-input-channel window-image split-bf16
 * kernel (sinc_x6.hip: the SincNet layer).  For tests and bench reports. */
int pase_conv_gemm_plan_kind(const PaseConvGemm* desc);
/* which kernel INSTANTIATION the launch runs (reports: bench.py matches it against the kernel names of a rocprofv3 trace):
 * 0 = conv_gemm_kernel (exact-fp32 pipe), 1 = sinc_x6_fwd_kernel, otherwise conv_x6c_kernel<NPOS, KGS, false, ZP, NARROW, SYM, DUO> as
 * NPOS * 1000 + KGS * 100 + 8 * DUO + 4 * SYM + 2 * ZP + NARROW (ZP: the pre-split activation of xp6 is staged by LDS DMA; SYM:
 * the symmetric form of such launches -- all waves multiply -- on a 256 x 128 tile, or with DUO on 128 x 128 tiles, two
 * four-wave workgroups per CU) */
int pase_conv_gemm_kernel_id(const PaseConvGemm* desc);
/* bytes of the split-bf16 pack the launch described by desc (wx6 ignored) would read; 0 = this shape only runs on
 * the fp32 matrix pipe */
long pase_conv_gemm_x6_bytes(const PaseConvGemm* desc);
/* desc->wx6 (pase_conv_gemm_x6_bytes(desc) bytes, 16-B aligned) <- desc->wt (or, when wt is NULL, desc->w in the reference's
 * own layout: no K-major intermediate is needed on this pipe) split into three bf16 planes in the
 * fragment order of the launch desc describes (tile, stage and tap padding are functions of the descriptor: pack and
 * launch must see the same one).  Like pase_pack_wt it runs once per weight use. */
int pase_pack_x6(const PaseConvGemm* desc, void* stream);
/* bytes of PaseConvGemm::xp6 the launch described by desc (xp6 ignored) wants; 0 = the launch splits while staging */
long pase_conv_gemm_xp_bytes(const PaseConvGemm* desc);
/* desc->xp6 (pase_conv_gemm_xp_bytes(desc) bytes, 16-B aligned, caller-owned scratch, read by that one launch) <-
 * act(bn(desc->x)) split into three bf16 planes; once per launch, before pase_conv_gemm */
int pase_pack_xp(const PaseConvGemm* desc, void* stream);

/* ------------------------------------------------------------------------------------------
 * pase_wgrad_gemm -- weight (+bias) gradient contraction, split-K with fp32 atomics.
 *
 *   dw[m, j(ci,kk)] += sum_{s,q} g[s, g_coff+m, q] * Z~[s, z_coff+ci, q*stride + kk*tapstep - padL]
 *   dbias[m]        += sum_{s,q} g[s, g_coff+m, q]                       (if dbias != NULL)
 *
 * Replaces the conv1d / conv_transpose1d / linear weight-gradient kernels autograd dispatches for
 * `tot_loss.backward()` (WorkerScheduler/worker_scheduler.py:67).  The caller zeroes dw / dbias.
 * ------------------------------------------------------------------------------------------ */
typedef struct PaseWgrad {
    const float* g;        /* (S, g_ctot, Tg): rows [g_coff, g_coff+M), first Ncols time steps      */
    const float* z;        /* (S, z_ctot, Tz): channels [z_coff, z_coff+Cin), on-load transform     */
    float* dw;             /* (M, ldw) row-major, column j = ci*taps+kk (or kk*Cin+ci)              */
    float* dbias;          /* (M) or NULL                                                         */
    const float* in_scale; const float* in_shift; const float* in_alpha;   /* (Cin) or NULL        */
    const float* g_alpha;  /* (M) PReLU slope applied to g on load (ConvTranspose1d wgrad: g is the
                              layer's raw input), or NULL                                         */
    int S, M, Tg, g_ctot, g_coff, Ncols;
    int Cin, Tz, z_ctot, z_coff, taps, tap_major, stride, tapstep, padL, pad_mode, ldw;
    int splitk;            /* 0 = auto                                                            */
    int x6;                /* 0: fp32 matrix pipe.  bit 0: split-bf16 contraction (see PaseConvGemm::wx6) where the library
                              has a plan for the shape AND gx6 is given; else the fp32 matrix pipe.  Measurement / test
                              controls (the library reads no environment variables): bits 4-7 force an orientation
                              (pase_wgrad_plan_kind value, 0 = the library's routing); bit 8: 1x1 layers may take the split
                              kernel in either orientation; bit 9: no row-coalesced staging; bit 10: keep the
                              one-channel (SincNet) layer off its window-image kernel; bit 11: pre-split launches of >= 256 rows
                              stay on the four-compute-wave kernel (no symmetric form); bits 12-14: k-groups per stage
                              of the pre-split-planes kernel (0 = its capacity)                                     */
    void* gx6;             /* scratch for the split-bf16 operands: pase_wgrad_x6_bytes(desc) bytes, 16-B
                              aligned, caller-owned, written and read by this launch only; NULL = fp32 matrix pipe */
    int max_wg;            /* cap on the persistent grid (0 = 256), see PaseConvGemm::max_wg                      */
} PaseWgrad;
int pase_wgrad_gemm(const PaseWgrad* desc, void* stream);
/* bytes of PaseWgrad::gx6 the launch described by desc needs (0: the shape runs on the fp32 matrix pipe) */
long pase_wgrad_x6_bytes(const PaseWgrad* desc);
/* which kernel a launch with x6 = 1 and a gx6 scratch runs on: 0 fp32 matrix pipe; split-bf16 position contraction with
 * 1 rows = g (packed), columns = (channel, tap) of z staged; 2 1x1 layer, rows = z channels (packed), columns = g staged;
 * 3 rows = (channel, tap) read from row-major bf16 planes of z, columns = g staged;
 * 4 rows = g (packed), columns = (channel, tap) COPIED out of pre-split phase-decomposed bf16 planes of z~ (no conversion
 *   in the GEMM: the default for every layer with taps);
 * 5 one input channel (SincNet): sinc_x6.hip, both operands converted while staged, window image of z;
 * 7 as 4 for at least 256 rows of g: the symmetric form -- 256 x 128 workgroup tile, all eight waves multiply, the planes
 *   copied by every wave's LDS DMA (x6c_wgrad_sym_kernel) */
int pase_wgrad_plan_kind(const PaseWgrad* desc);

/* ------------------------------------------------------------------------------------------
 * BatchNorm1d (training-mode batch statistics) pieces.  Reference: nn.BatchNorm1d built by
 * build_norm_layer (pase/models/modules.py:77-79), applied in FeBlock.forward (:1072-1074) and as
 * norm_out (frontend.py:206-210,267-268; affine=False -> gamma = beta = NULL).
 * pase_bn_finalize turns the per-tile (sum, sumsq) partials written by pase_conv_gemm into the
 * on-load affine (scale = gamma*rstd, shift = beta - mean*scale), saves mean / rstd for backward
 * and updates running_mean / running_var (momentum, unbiased variance) exactly like torch.
 * ------------------------------------------------------------------------------------------ */
int pase_bn_finalize(const float* stat_part, int ntiles, int C, double count, const float* gamma,
                     const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                     float* scale, float* shift, float* mean_out, float* rstd_out, void* stream);

/* out[s, o_coff+c, f] = mean_{i<d} act(bn(y[s, c, f*d+i])): WaveFe.fuse_skip's
 * skip.view(b, f, T//d, d).mean(3) (pase/models/frontend.py:213-232), applied BEFORE the 1x1
 * dense-skip projection (mean-pool and a bias-free 1x1 conv commute exactly). */
int pase_bn_act_pool(const float* y, float* out, const float* scale, const float* shift, const float* alpha,
                     int S, int C, int T, int F, int d, int o_ctot, int o_coff, void* stream);

/* out = act(bn(y)) materialised (public encoder output after norm_out; API-compat activations) */
int pase_bn_act_apply(const float* y, float* out, const float* scale, const float* shift, const float* alpha,
                      int S, int C, int T, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward of  a = PReLU(BN(y))  (either stage optional).  The gradient w.r.t. `a` is assembled
 * on the fly from (1) the data-gradient a pase_conv_gemm dgrad launch wrote in padded
 * coordinates -- reflect padding folds mirrored edges back, the autograd of
 * F.pad(mode='reflect') at modules.py:1071 -- and (2) the pooled dense-skip gradient
 * (autograd of fuse_skip's mean, frontend.py:225-226).
 *   reduce: sums[c] = { sum dz, sum dz*xhat, sum dA*z*[z<=0] }  (doubles, caller zeroes)
 *           -> dbeta = sums[.,0], dgamma = sums[.,1], dalpha = sums[.,2]
 *   apply : dy = scale*(dz - sums0/N - xhat*sums1/N)  (has_bn == 1, batch statistics)
 *           dy = scale*dz  (has_bn == 2: frozen / eval-mode statistics)   or   dy = dz  (has_bn == 0)
 *   With has_bn != 1 and dy != NULL the reduce pass already writes dy: the apply pass is not needed.
 * ------------------------------------------------------------------------------------------ */
typedef struct PaseActBwd {
    const float* y;        /* (S, y_ctot, T) raw layer output; channels [y_coff, y_coff+C)          */
    const float* dsrc;     /* (S, dsrc_ctot, Tp) padded data-gradient, or NULL                    */
    const float* dpool;    /* (S, dpool_ctot, pool_F) pooled-branch gradient, or NULL             */
    const float* scale; const float* shift; const float* alpha;   /* forward on-load params, NULL = identity */
    const float* mean; const float* rstd;                         /* from pase_bn_finalize (has_bn) */
    double* sums;          /* (C, 3)                                                              */
    float* dy;             /* (S, y_ctot, T) output of the apply pass, same channel slice as y    */
    int S, C, T, y_ctot, y_coff;
    int dsrc_ctot, dsrc_coff, Tp, padL, pad_mode;
    int dpool_ctot, dpool_coff, pool_F, pool_d;
    float pool_inv;        /* 1 / pool_d                                                          */
    int has_bn;            /* 0 none, 1 BatchNorm with batch statistics, 2 BatchNorm with frozen statistics;
                              3 / 4: InstanceNorm / LayerNorm for pase_rownorm_act_bwd */
} PaseActBwd;
int pase_act_bwd_reduce(const PaseActBwd* desc, void* stream);
int pase_act_bwd_apply(const PaseActBwd* desc, void* stream);
/* pase_wgrad_gemm whose gradient operand is not read from desc->g (ignored, may be NULL) but evaluated while it is staged as
 * the APPLY pass of g_bwd: same arithmetic as pase_act_bwd_apply, dy never written (g_bwd->dy ignored).  g_bwd->sums must be
 * complete (pase_act_bwd_reduce enqueued earlier on the same stream); g_bwd->S / C / T = desc->S / M / Ncols; has_bn 0..2.
 * For the layer whose dy has no other consumer: the SincNet layer's 786 MB dy at bs32 -- the first layer needs no data
 * gradient (autograd of F.conv1d w.r.t. the filters only, pase/models/modules.py:932) -- so its apply pass (read y, read dA,
 * write dy) and the weight gradient's read of dy become one read of y and dA.  Only the one-input-channel plan
 * (pase_wgrad_plan_kind 5) has this form: -11 when desc would run on another kernel (x6 bit 0 and gx6 are required),
 * -13 when g_bwd does not describe desc's gradient operand or the kernel cannot evaluate it while staging: shape mismatch,
 * y NULL, g_alpha / dbias set, norms 3 / 4 (has_bn outside 0..2), has_bn 1 without sums, a pooled branch with pool_d < 16
 * (a staged run of 16 positions may touch at most two pooled frames), a padded data gradient with Tp < T + padL.
 * pase_wgrad_gemm_act_bwd_ok answers the same question without enqueueing anything (1 = the launch would be accepted,
 * 0 = -11 / -13): callers test it and otherwise materialise dy with pase_act_bwd_apply and run the plain pase_wgrad_gemm. */
int pase_wgrad_gemm_act_bwd(const PaseWgrad* desc, const PaseActBwd* g_bwd, void* stream);
int pase_wgrad_gemm_act_bwd_ok(const PaseWgrad* desc, const PaseActBwd* g_bwd);

/* Per-sample normalisations of the other norm_type values (pase/models/modules.py:77-109): nn.InstanceNorm1d
 * ('inorm', 'affinorm', and WaveFe.norm_out when norm_type != 'bnorm', frontend.py:206-210; mode 0: statistics per
 * (sequence, channel) over time) and nn.LayerNorm(C) on the transposed tensor ('lnorm'; mode 1: statistics per
 * (sequence, time step) over channels).  Forward materialises out = PReLU(gamma * xhat + beta) (gamma / beta /
 * alpha NULL = identity) and the group statistics mean_out / rstd_out ((S, C) in mode 0, (S, T) in mode 1; biased
 * variance, eps inside the square root, like torch).  Backward: the PaseActBwd descriptor with has_bn = 3 (instance)
 * or 4 (layer), scale / shift = gamma / beta, mean / rstd = the forward's group statistics; writes dy and accumulates
 * sums[c] = {dbeta, dgamma, dalpha} in one launch. */
int pase_rownorm_act_fwd(const float* y, float* out, const float* gamma, const float* beta, const float* alpha,
                         float* mean_out, float* rstd_out, int S, int C, int T, float eps, int mode, void* stream);
int pase_rownorm_act_bwd(const PaseActBwd* desc, void* stream);

/* ------------------------------------------------------------------------------------------
 * QRNN (third-party salesforce/pytorch-qrnn; call sites pase/models/modules.py:48-53,
 * frontend.py:256-259): gates (S, 3H, F) in chunk order Z | F | O.
 *   fwd: C_t = sig(F_t)*tanh(Z_t) + (1-sig(F_t))*C_{t-1}, H_t = sig(O_t)*C_t  (C_{-1} = 0)
 *        h_out (S, h_ctot, F) at channel offset h_coff, c_out (S, H, F) saved for backward.
 *   bwd: d(gates) from dH.
 * ------------------------------------------------------------------------------------------ */
int pase_qrnn_scan_fwd(const float* gates, float* h_out, float* c_out, int S, int H, int F, int h_ctot,
                       int h_coff, void* stream);
int pase_qrnn_scan_bwd(const float* gates, const float* c_saved, const float* dh, float* dgates, int S, int H,
                       int F, int dh_ctot, int dh_coff, void* stream);

/* ------------------------------------------------------------------------------------------
 * Single-output heads + losses.
 *   pase_head1_fwd: y[s,t] = bias + sum_c w[c]*act(z[s,c,t]) (final Conv1d(hidden,1,1) of
 *     DecoderMinion / MLPMinion, Minions/minions.py:431,:510) fused with L1 / MSE /
 *     BCE-with-logits against `target` (S,1,T): loss_acc += sum(loss), dy = dloss/dy * grad_scale.
 *   pase_head1_bwd: dz (S,C,T), sums = {dw[c], dalpha[c], sum_st dz[.,c,.]} x C then db at sums[3*C].
 *   pase_ctx_loss : ContextualizedLoss.__call__ (pase/losses.py:33-37) on a materialised
 *     prediction (B, M, F); r_ctx > 1 gathers the target with contextualize_r's stacking (:14-31).
 * ------------------------------------------------------------------------------------------ */
int pase_head1_fwd(const float* z, const float* in_scale, const float* in_shift, const float* in_alpha,
                   const float* w, const float* bias, const float* target, float* y, float* dy, double* loss_acc,
                   int S, int C, int T, int loss_type, float grad_scale, void* stream);
int pase_head1_bwd(const float* z, const float* in_alpha, const float* w, const float* dy, float* dz,
                   double* sums, int S, int C, int T, void* stream);
int pase_ctx_loss(const float* pred, const float* label, float* dpred, double* loss_acc, int B, int M, int F,
                  int r_ctx, int label_D, int loss_type, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * The pointwise tail of a decoder worker, forward, loss and backward in ONE pass over its input:
 *   a0 = PReLU(y, alpha0);  y1 = W1 a0 + b1  (MLPBlock, context 1, pase/models/modules.py:527-556);  a1 = PReLU(y1, alpha1);
 *   pred = w2 . a1 + b2  (DecoderMinion.W = Conv1d(hidden, 1, 1), Minions/minions.py:416-417,446);  loss(pred, target)
 *   (pase/losses.py:33-37, r = None)  and the autograd of all of it down to dy = d(loss * grad_scale') / dy.
 * y is the raw output of the layer below (a GDeconv1DBlock with norm_type None, modules.py:558-589: nothing but its PReLU
 * stands between the layers).  Replaces, for that tail, pase_conv_gemm (64-row 1x1) + pase_head1_fwd + pase_head1_bwd +
 * pase_wgrad_gemm (1x1) + pase_conv_gemm (1x1 data gradient) + pase_act_bwd_reduce over y: y1, dy1 and the gradient of
 * a0 are never written.  Exact-fp32 matrix pipe (the arithmetic of the launches it replaces).
 *   Outputs: dy (S, C, T);  pred (S, T) or NULL;  loss_acc += sum of per-element losses;  dw1 (H, C) += ;
 *   sums0 (C, 3) doubles += {sum dy (the bias gradient of the layer below), -, dalpha0}   (PaseActBwd::sums layout);
 *   sums1 (H, 3) + 1 doubles += {dw2, dalpha1, sum dy1 (= db1)} then db2 at [3 H]        (pase_head1_bwd layout).
 * Specialised for C = 128, H = 64 (cfg/workers/workers+.cfg "cchunk"): pase_mlp_head1_supported says whether a shape has the
 * form; pase_mlp_head1_step returns -11 otherwise and the caller runs the six launches.
 * ------------------------------------------------------------------------------------------ */
typedef struct PaseMlpHead1 {
    const float* y;        /* (S, C, T)                                                           */
    const float* alpha0;   /* (C) PReLU slopes of the layer below, NULL = identity                */
    const float* w1;       /* (H, C) row-major                                                    */
    const float* b1;       /* (H) or NULL                                                         */
    const float* alpha1;   /* (H) or NULL                                                         */
    const float* w2;       /* (H)                                                                 */
    const float* b2;       /* (1) or NULL                                                         */
    const float* target;   /* (S, T); NULL with PASE_LOSS_NONE                                    */
    float* pred;           /* (S, T) or NULL                                                      */
    float* dy;             /* (S, C, T)                                                           */
    double* loss_acc;      /* (1)                                                                 */
    double* sums0;         /* (C, 3)                                                              */
    double* sums1;         /* (3 H + 1)                                                           */
    float* dw1;            /* (H, C)                                                              */
    int S, C, T, H, loss_type;
    float grad_scale;      /* dpred = dloss/dpred * grad_scale (loss weight / number of elements) */
    int max_wg;            /* cap on the persistent grid (0 = 256), see PaseConvGemm::max_wg      */
} PaseMlpHead1;
int pase_mlp_head1_supported(const PaseMlpHead1* desc);
int pase_mlp_head1_step(const PaseMlpHead1* desc, void* stream);

/* Sinc band-pass bank: SincConv_fast.forward filter synthesis (pase/models/modules.py:881-915) and
 * its gradient w.r.t. low_hz_ / band_hz_.  n_ and window_ are the module's constant buffers. */
int pase_sinc_filters(const float* low_hz_, const float* band_hz_, const float* n_, const float* window_,
                      float* filt, int C, int Kw, float min_low, float min_band, float sr, void* stream);
int pase_sinc_filters_bwd(const float* low_hz_, const float* band_hz_, const float* n_, const float* window_,
                          const float* dfilt, float* dlow, float* dband, int C, int Kw, float min_low,
                          float min_band, float sr, void* stream);

/* dst[(p*O + o), (red*taps_p + j)] = src[red*s_red + o*s_out + (p + st*j)*s_k]  (0 beyond k),
 * taps_p = ceil(k/st): the A operand that turns pase_conv_gemm into the data-gradient of a strided
 * conv / the forward of nn.ConvTranspose1d (phase decomposition). */
/* g_k[c] += (float) sums[c*ld + col_k], k < 3 (NULL buffers skipped): one launch commits the fp64 per-channel sums
 * of pase_act_bwd_reduce / pase_head1_bwd into the BatchNorm / PReLU / bias gradient buffers (what autograd's
 * AccumulateGrad does for those parameters in the reference) */
int pase_commit_cols(const double* sums, int ld, int C, float* g0, int c0, float* g1, int c1, float* g2, int c2,
                     void* stream);
/* dst_k[r, c] += src_k[r, c] for up to 16 row-major blocks (rows x width, leading dimensions src_ld / dst_ld) in one
 * launch: staged weight gradients of concatenated / stacked GEMMs into their parameters' gradient buffers (what autograd's
 * slicing backward of torch.cat does in the reference: frontend.py:244-266 dense-skip sum, one Conv1d per worker) */
typedef struct PaseAddBlock { const float* src; float* dst; int rows, width, src_ld, dst_ld; } PaseAddBlock;
typedef struct PaseAddBlocks { int n; PaseAddBlock seg[16]; } PaseAddBlocks;
int pase_add_blocks(const PaseAddBlocks* desc, void* stream);
int pase_pack_dgrad(const float* src, float* dst, int R, int O, int k, int st, long s_red, long s_out, long s_k,
                    void* stream);
/* same pack written K-major for pase_conv_gemm's `wt` operand: dst ((R*ceil(k/st)), ldt),
 * dst[(red*taps_p + j)*ldt + (p*O + o)]; ldt % 4 == 0, ldt >= st*O, pad columns zero-filled */
int pase_pack_dgrad_t(const float* src, float* dst, int R, int O, int k, int st, long s_red, long s_out,
                      long s_k, int ldt, void* stream);

/* torch.optim.Adam (defaults; WorkerScheduler/trainer.py:91,111,134) over flat buffers; lr and step
 * are device scalars so a captured hipGraph stays valid.  grad_mul pre-scales g (1/world_size). */
int pase_adam_step(float* p, const float* g, float* m, float* v, long n, const float* lr, const int* step,
                   float beta1, float beta2, float eps, float grad_mul, void* stream);
int pase_step_tick(int* step, void* stream);

/* ------------------------------------------------------------------------------------------
 * On-device regression targets (pase/transforms.py LPS :439-487, FBanks :489-548, MFCC :671-722,
 * ZNorm :183-205 -- host numpy / librosa / python_speech_features code in the reference's
 * DataLoader workers).  The spectra are DFT-basis convolutions on pase_conv_gemm (PASE_POST_*);
 * these two entry points cover the rest.
 *   pase_delta_znorm: out (B, (order+1)*D, Fo) = [x, delta_1(x), .., delta_order(x)] z-normalised;
 *     delta = librosa.feature.delta = Savitzky-Golay width 9 (coef: (order+1, 9, 9) host table,
 *     pos 4 = interior, other pos = mode='interp' edge fits); columns F..Fo-1 replicate column F-1
 *     (FBanks' replicate padding, transforms.py:534-538); mean / istd (per output channel) or NULL.
 *   pase_power_to_db: librosa.power_to_db(S, ref, amin, top_db) with the clamp at the per-utterance
 *     maximum - top_db; umax_scratch: B unsigned ints.
 * ------------------------------------------------------------------------------------------ */
int pase_delta_znorm(const float* x, const float* coef, const float* mean, const float* istd, float* out, int B,
                     int D, int F, int Fo, int order, int x_ctot, int x_coff, void* stream);
int pase_power_to_db(const float* x, float* y, unsigned* umax_scratch, long per_utt, int B, float amin,
                     float ref_db, float top_db, void* stream);
/* Prosody target, energy + zero-crossing rows (pase/transforms.py:967-978: librosa.feature.rmse with
 * pad_mode='constant' and librosa.feature.zero_crossing_rate, both centred frames of `win` samples every `hop`):
 * out[b, out_coff, f] = sqrt(mean(x_zero-padded^2)), out[b, out_coff+1, f] = sign-bit changes inside the
 * edge-padded frame / win (|x| <= 1e-10 counts as +0).  x (B, T), out (B, out_ctot, F). */
int pase_zcr_rms(const float* x, float* out, int B, int T, int F, int hop, int win, int out_ctot, int out_coff,
                 void* stream);
/* Prosody target, pitch rows (pase/transforms.py:948-961): f0 (B, Fin) in Hz with 0 on unvoiced frames ->
 * out[b, out_coff] = log(f0 + 1e-10) with unvoiced stretches interpolated over the WHOLE contour (ahoproc_tools
 * interpolation(lf0, -1)), then truncated to F <= Fin frames; out[b, out_coff+1] = voiced flag; a chunk without a
 * voiced frame among the kept ones gets log(f0_min) / 0.  out (B, out_ctot, F). */
int pase_lf0_interp(const float* f0, float* out, int B, int Fin, int F, int out_ctot, int out_coff, float f0_min,
                    void* stream);
/* SWIPE' f0 tracker (what pysptk.swipe computes for the Prosody target, pase/transforms.py:948-952), the two steps
 * that are not pase_conv_gemm launches:
 *   pase_swipe_accumulate: one window size's pitch strengths num / sqrt(den2) ((B, nj, nfr) each, 0 where den2 == 0),
 *     interpolated linearly to the output frames (frame f sits at f * frames_per_out of that window's hops) and
 *     added, times mu[c], into row cand[c] of S (B, NC, F) (caller-zeroed before the first window);
 *   pase_swipe_pick: per output frame the strongest candidate (candidate c = 2^(log2_fmin + c * dlog2p) Hz), f0 = 0
 *     when its strength is below st, else the maximum of the parabola through the three strengths around it on a
 *     `polyv`-octave grid; strength (optional) receives the winning strength. */
int pase_swipe_accumulate(const float* num, const float* den2, const float* mu, const int* cand, float* S, int B, int nj,
                          int nfr, int NC, int F, float frames_per_out, void* stream);
int pase_swipe_pick(const float* S, float* f0, float* strength, int B, int NC, int F, float log2_fmin, float dlog2p,
                    float polyv, float st, void* stream);
/* Framing prologue of LPS / FBanks / MFCC (transforms.py:465-466 torch.stft centre padding, :517
 * logfbank framing + pre-emphasis, :700 librosa stft): y (B, hop, Q) with
 * y[b][r][q] = xpad[q*hop + r], xpad = x padded by padL on the left (pad_mode PASE_PAD_REFLECT or
 * PASE_PAD_ZERO; the right side is padded the same way as far as Q*hop reaches), after the optional
 * pre-emphasis x[n] - preemph*x[n-1] (0 = off).  A hop-strided frame of `win` samples then is a
 * stride-1 conv over q with Cin = hop and ceil(win/hop) taps, which pase_conv_gemm runs. */
int pase_frame_prep(const float* x, float* y, int B, int T, int hop, int Q, int padL, int pad_mode,
                    float preemph, void* stream);

/* Gammatone (pase/transforms.py:550-613 -> gammatone.gtgram.gtgram): 4th-order gammatone filterbank as four
 * cascaded biquads per channel (coef (C, 10) doubles in gammatone.filters.make_erb_filters' column order),
 * squared and summed over blocks of g samples -> blocks (B*C, ceil(T/g)); pase_gammatone_frames turns block
 * sums into out (rows, ncol) = log(sqrt(mean over nwin samples from c*hop) + eps) (g divides nwin and hop). */
int pase_gammatone_blocks(const float* x, const double* coef, float* blocks, int B, int C, int T, int g, void* stream);
int pase_gammatone_frames(const float* blocks, float* out, int rows, int T, int g, int nwin, int hop, int ncol,
                          float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * On-device batch producer: what the reference's DataLoader workers do
 * per utterance in numpy / scipy, for the whole batch on resident waveforms.  Random decisions are made by the
 * caller and passed as index arrays (device memory).
 * ------------------------------------------------------------------------------------------ */
/* SingleChunkWav.select_chunk / MIChunkWav (pase/transforms.py:309-356,388-436): out[n, :] = wav_{src[n]}[beg[n] :
 * beg[n]+T]; waveform u lives at pool[off[u] : off[u]+len[u]]; a waveform with len <= T is taken from 0 and
 * right-padded by reflection (torch F.pad mode='reflect'). */
int pase_chunk_gather(const float* pool, const long long* off, const int* len, const int* src, const int* beg,
                      float* out, int N, int T, void* stream);
/* norm_and_scale (pase/transforms.py:148-151), in place on (N, T): x / max|x| * u[n] */
int pase_peak_scale(float* x, const float* u, int N, int T, void* stream);
/* Reverb.__call__ (pase/transforms.py:1071-1103), in place on x (B, T): full = scipy.signal.convolve(x_b,
 * IR, 'full'); shifted left by the IR's peak position, trimmed to T, scaled by sqrt(sum x^2 / sum full^2).
 * IR i = irs[ir_off[i] : +ir_len[i]] (already truncated / peak-normalised by the caller, load_IR :1027-1044),
 * ir_pmax[i] = argmax|IR_i|; ir_idx[b] < 0 leaves utterance b untouched.  Scratch: full (B, T+max_ir_len-1)
 * floats, energies (2B) doubles. */
int pase_reverb(float* x, const float* irs, const long long* ir_off, const int* ir_len, const int* ir_pmax,
                const int* ir_idx, float* full, double* energies, int B, int T, int max_ir_len, void* stream);
/* The same FIR with the filter-distortion conventions: BandDrop / Downsample (pase/transforms.py:1113-1300) shift
 * by round(len/2) (ir_shift[i], computed by the caller with the reference's Python round) and take the energy
 * ratio on the shifted, trimmed signal (trimmed_energy = 1); trimmed_energy = 0 is pase_reverb. */
int pase_fir_distort(float* x, const float* irs, const long long* ir_off, const int* ir_len, const int* ir_shift,
                     const int* ir_idx, float* full, double* energies, int B, int T, int max_ir_len,
                     int trimmed_energy, void* stream);
/* SimpleAdditiveShift (overlapped speech, pase/transforms.py:1684-1766): out[b, t] = 0 for t < shift[b], else
 * wav_{src[b]}[beg[b] + t - shift[b]] (zero past the file end; src[b] < 0 leaves row b untouched);
 * pase_zero_front re-zeroes the first shift[b] samples after the optional reverberation of that crop.  The mix
 * itself is pase_add_noise with the crop buffer as the noise pool. */
int pase_overlap_gather(const float* pool, const long long* off, const int* len, const int* src, const int* beg,
                        const int* shift, float* out, int B, int T, void* stream);
int pase_zero_front(float* x, const int* shift, int B, int T, void* stream);
/* Clipping.__call__ (pase/transforms.py:1514-1535), in place: clamp utterance b to [f * min, f * max], f = factor[b]
 * (<= 0: untouched) */
int pase_clip(float* x, const float* factor, int B, int T, void* stream);
/* SimpleAdditive.__call__ (pase/transforms.py:1633-1675), in place on x (B, T): noise crop
 * npool[noff[i] + nbeg[b] : +T] (zero beyond nlen[i]), K = sqrt(Ex / (10^(snr/10) En)), x <- (x + K n) *
 * sqrt(Ex / (E(x + K n) + 1e-14)); nidx[b] < 0 or a silent crop leaves utterance b untouched. */
int pase_add_noise(float* x, const float* npool, const long long* noff, const int* nlen, const int* nidx,
                   const int* nbeg, const float* snr, int B, int T, void* stream);

/* sizeof() of the ABI structs (0 = PaseConvGemm, 1 = PaseWgrad, 2 = PaseActBwd), for binding self-checks */
int pase_abi_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* PASE_AMD_H */

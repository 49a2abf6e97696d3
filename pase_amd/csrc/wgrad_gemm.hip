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
// input at the low rate, Z = dY at the high rate) and the result lands directly in the
// (in, out, k) weight layout.
//
// GEMM view: M x Nw x Kred with Kred = S*Ncols (19 200 ... 3 072 000): the reduction is the long
// axis, so the grid is (row tiles x col tiles x split-K) and partial tiles are combined with fp32
// global atomics into a caller-zeroed dW.  Same 4-wave / 2x2x(32x32x2) register tiling and
// double-buffered LDS as conv_gemm.hip; both operands are K-contiguous in HBM so the loaders put
// consecutive lanes along K.
#include "hip_compat.h"
#include "pase_amd.h"

namespace {

constexpr int BK = 16;
constexpr int NTHREADS = 256;

template <int BM, int BN>
__global__ void __launch_bounds__(NTHREADS) wgrad_gemm_kernel(PaseWgrad p, int n_row_tiles, int n_col_tiles,
                                                              int kt_per_split) {
    constexpr int WAVES_N = BN / 64;
    constexpr int A_PER_T = BM * BK / NTHREADS;
    constexpr int B_PER_T = BN * BK / NTHREADS;
    constexpr int RSTEP = NTHREADS / BK;  // 16 rows / cols per pass
    __shared__ float As[2][BK][BM + 1];
    __shared__ float Bs[2][BK][BN + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    const int tiles = n_row_tiles * n_col_tiles;
    const int tile = blockIdx.x % tiles;
    const int split = blockIdx.x / tiles;
    const int mt = tile % n_row_tiles;
    const int ct = tile / n_row_tiles;
    const int m0 = mt * BM;
    const int j0 = ct * BN;

    const int Kw = p.Cin * p.taps;
    const int Nw = Kw + (p.dbias ? 1 : 0);
    const long kred = (long)p.S * p.Ncols;
    const int nk_total = (int)((kred + BK - 1) / BK);
    const int kt_begin = split * kt_per_split;
    const int kt_end = min(nk_total, kt_begin + kt_per_split);
    if (kt_begin >= kt_end) return;   // whole block exits together (no barrier reached yet)

    // loader coordinates: consecutive lanes along the reduction axis n = (s, q)
    const int kc = tid % BK;
    const int r0 = tid / BK;
    // per-thread fixed B columns -> (ci, kk)
    int bci[B_PER_T], bkk[B_PER_T];
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
        const int j = j0 + r0 + i * RSTEP;
        if (j < Kw) {
            if (p.tap_major) { bkk[i] = j / p.Cin; bci[i] = j - bkk[i] * p.Cin; }
            else             { bci[i] = j / p.taps; bkk[i] = j - bci[i] * p.taps; }
        } else {
            bci[i] = (j == Kw && p.dbias) ? -1 : -2;   // -1: ones column (bias), -2: out of range
            bkk[i] = 0;
        }
    }

    float areg[A_PER_T], breg[B_PER_T];
    auto load_tile = [&](int kt) {
        const long n = (long)kt * BK + kc;
        const bool nok = n < kred;
        const int s = nok ? (int)(n / p.Ncols) : 0;
        const int q = nok ? (int)(n - (long)s * p.Ncols) : 0;
        const float* grow = p.g + ((size_t)s * p.g_ctot + p.g_coff) * (size_t)p.Tg + q;
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            const int m = m0 + r0 + i * RSTEP;
            float gv = (nok && m < p.M) ? grow[(size_t)m * p.Tg] : 0.f;
            if (p.g_alpha) gv = gv > 0.f ? gv : gv * p.g_alpha[m < p.M ? m : 0];
            areg[i] = gv;
        }
        const float* zrow = p.z + ((size_t)s * p.z_ctot + p.z_coff) * (size_t)p.Tz;
        const int ubase = q * p.stride - p.padL;
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) {
            float v = 0.f;
            const int ci = bci[i];
            if (nok && ci >= 0) {
                int u = ubase + bkk[i] * p.tapstep;
                if (p.pad_mode == PASE_PAD_REFLECT) {
                    if (u < 0) u = -u;
                    if (u >= p.Tz) u = 2 * (p.Tz - 1) - u;
                }
                if (u >= 0 && u < p.Tz) {
                    v = zrow[(size_t)ci * p.Tz + u];
                    if (p.in_scale) v = v * p.in_scale[ci] + p.in_shift[ci];
                    if (p.in_alpha) v = v > 0.f ? v : v * p.in_alpha[ci];
                }
            } else if (nok && ci == -1) {
                v = 1.f;
            }
            breg[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) As[buf][kc][r0 + i * RSTEP] = areg[i];
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) Bs[buf][kc][r0 + i * RSTEP] = breg[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    load_tile(kt_begin);
    store_tile(0);
    __syncthreads();
    const int fr = lane & 31;
    const int fk = lane >> 5;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int kb = ks * 2 + fk;
            const float a0 = As[cur][kb][wm * 64 + fr];
            const float a1 = As[cur][kb][wm * 64 + 32 + fr];
            const float b0 = Bs[cur][kb][wn * 64 + fr];
            const float b1 = Bs[cur][kb][wn * 64 + 32 + fr];
            acc[0][0] = pase_mfma_32x32x2(a0, b0, acc[0][0]);
            acc[0][1] = pase_mfma_32x32x2(a0, b1, acc[0][1]);
            acc[1][0] = pase_mfma_32x32x2(a1, b0, acc[1][0]);
            acc[1][1] = pase_mfma_32x32x2(a1, b1, acc[1][1]);
        }
        if (kt + 1 < kt_end) store_tile(cur ^ 1);
        __syncthreads();
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

}  // namespace

extern "C" int pase_wgrad_gemm(const PaseWgrad* d, void* stream) {
    const PaseWgrad p = *d;
    if (p.M <= 0 || p.Cin <= 0 || p.S <= 0 || p.Ncols <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int Nw = p.Cin * p.taps + (p.dbias ? 1 : 0);
    const long kred = (long)p.S * p.Ncols;
    const int nk = (int)((kred + BK - 1) / BK);
    const bool narrow = p.M <= 64;
    const int BMv = narrow ? 64 : 128, BNv = narrow ? 256 : 128;
    const int nrt = (p.M + BMv - 1) / BMv, nct = (Nw + BNv - 1) / BNv;
    const int tiles = nrt * nct;
    int splitk = p.splitk;
    if (splitk <= 0) {
        splitk = (1536 + tiles - 1) / tiles;            // ~6 workgroups per CU in flight
        const int max_split = (nk + 7) / 8;             // at least 8 K-tiles (2 k MFMAs/wave) per split
        if (splitk > max_split) splitk = max_split;
        if (splitk < 1) splitk = 1;
    }
    const int kt_per_split = (nk + splitk - 1) / splitk;
    splitk = (nk + kt_per_split - 1) / kt_per_split;
    if (narrow)
        PASE_LAUNCH((wgrad_gemm_kernel<64, 256>), dim3((unsigned)(tiles * splitk)), dim3(NTHREADS), st, p, nrt, nct, kt_per_split);
    else
        PASE_LAUNCH((wgrad_gemm_kernel<128, 128>), dim3((unsigned)(tiles * splitk)), dim3(NTHREADS), st, p, nrt, nct, kt_per_split);
    PASE_CHECK_LAUNCH();
    return 0;
}

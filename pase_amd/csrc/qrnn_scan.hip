// qrnn_scan.hip -- QRNN gate non-linearities + ForgetMult recurrence, forward and backward.
//
// Third-party semantics (salesforce/pytorch-qrnn, un-vendored & un-pinned: requirements.txt:16;
// call sites pase/models/modules.py:48-53, pase/models/frontend.py:190-194,256-259):
//   Y = Linear([x_t ; x_{t-1}]);  Z, F, O = Y.chunk(3);  Z = tanh(Z); F = sigmoid(F)
//   C_t = F_t * Z_t + (1 - F_t) * C_{t-1}   (C_{-1} absent: C_0 = F_0 Z_0)     [ForgetMult]
//   H_t = sigmoid(O_t) * C_t
// The upstream CUDA kernel runs one thread per (batch, hidden) serially over time on a
// (T, B, H) layout.  Here the gates stay in the encoder's NCT layout (S, 3H, F) -- time is the
// contiguous axis -- and one 64-lane wave owns one (s, h) row: each lane takes 4 consecutive time
// steps, composes its affine maps c -> a*c + b locally, and the wave runs a 6-step shuffle scan
// over the (a, b) pairs (the recurrence is an associative composition).  HBM-bound:
// forward reads 3 gate rows + writes h and c; backward reads 3 gates + c + dh and writes 3 grads.
#include "hip_compat.h"
#include "pase_amd.h"

namespace {

constexpr int NT = 256;
constexpr int PER = 4;
constexpr int CHUNK = 64 * PER;

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(NT) qrnn_fwd_kernel(const float* gates, float* h_out, float* c_out, int S,
                                                      int H, int F, int h_ctot, int h_coff) {
    const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);   // (s, h); whole wave shares it
    const int lane = threadIdx.x & 63;
    if (row >= S * H) return;
    const int s = row / H, hch = row % H;
    const float* gz = gates + ((size_t)s * 3 * H + hch) * (size_t)F;
    const float* gf = gz + (size_t)H * F;
    const float* go = gf + (size_t)H * F;
    float* hrow = h_out + ((size_t)s * h_ctot + h_coff + hch) * (size_t)F;
    float* crow = c_out + ((size_t)s * H + hch) * (size_t)F;
    float carry = 0.f;
    for (int t0 = 0; t0 < F; t0 += CHUNK) {
        float a[PER], b[PER], o[PER];
        float A = 1.f, B = 0.f;   // composed map of this lane's PER steps
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int t = t0 + lane * PER + i;
            if (t < F) {
                const float z = tanhf(gz[t]);
                const float f = sigm(gf[t]);
                o[i] = sigm(go[t]);
                a[i] = 1.f - f;
                b[i] = f * z;
            } else { a[i] = 1.f; b[i] = 0.f; o[i] = 0.f; }
            B = a[i] * B + b[i];
            A = a[i] * A;
        }
        // inclusive scan over lanes of (A, B): x -> A*x + B, later lane applied after earlier
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float Ap = __shfl(A, lane - d >= 0 ? lane - d : lane);
            const float Bp = __shfl(B, lane - d >= 0 ? lane - d : lane);
            if (lane >= d) { B = A * Bp + B; A = A * Ap; }
        }
        // exclusive prefix for this lane = inclusive of lane-1 applied to carry
        float Ae = __shfl(A, lane > 0 ? lane - 1 : 0);
        float Be = __shfl(B, lane > 0 ? lane - 1 : 0);
        float c = lane > 0 ? Ae * carry + Be : carry;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int t = t0 + lane * PER + i;
            c = a[i] * c + b[i];
            if (t < F) { crow[t] = c; hrow[t] = o[i] * c; }
        }
        // carry = state after the last element of the chunk (lane 63 inclusive)
        const float At = __shfl(A, 63), Bt = __shfl(B, 63);
        carry = At * carry + Bt;
    }
}

// backward: dC_t = dH_t * O_t + (1 - F_{t+1}) dC_{t+1};  dZ = dC*F*(1-Z^2);  dF = dC*(Z - C_{t-1})*F(1-F)
//           dO = dH * C * O(1-O).
// In reversed time tau = F-1-t the adjoint is the same affine recurrence
//   D_tau = a_tau * D_{tau-1} + b_tau,  a_tau = 1 - F_{t+1},  b_tau = dH_t * O_t,  D_{-1} = 0.
__global__ void __launch_bounds__(NT) qrnn_bwd_kernel(const float* gates, const float* c_saved, const float* dh,
                                                      float* dgates, int S, int H, int F, int dh_ctot,
                                                      int dh_coff) {
    const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= S * H) return;
    const int s = row / H, hch = row % H;
    const float* gz = gates + ((size_t)s * 3 * H + hch) * (size_t)F;
    const float* gf = gz + (size_t)H * F;
    const float* go = gf + (size_t)H * F;
    const float* crow = c_saved + ((size_t)s * H + hch) * (size_t)F;
    const float* dhrow = dh + ((size_t)s * dh_ctot + dh_coff + hch) * (size_t)F;
    float* dz_ = dgates + ((size_t)s * 3 * H + hch) * (size_t)F;
    float* df_ = dz_ + (size_t)H * F;
    float* do_ = df_ + (size_t)H * F;
    float carry = 0.f;
    for (int tau0 = 0; tau0 < F; tau0 += CHUNK) {
        float a[PER], b[PER];
        float A = 1.f, B = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int tau = tau0 + lane * PER + i;
            const int t = F - 1 - tau;
            if (t >= 0) {
                a[i] = (t + 1 < F) ? 1.f - sigm(gf[t + 1]) : 0.f;
                b[i] = dhrow[t] * sigm(go[t]);
            } else { a[i] = 1.f; b[i] = 0.f; }
            B = a[i] * B + b[i];
            A = a[i] * A;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float Ap = __shfl(A, lane - d >= 0 ? lane - d : lane);
            const float Bp = __shfl(B, lane - d >= 0 ? lane - d : lane);
            if (lane >= d) { B = A * Bp + B; A = A * Ap; }
        }
        const float Ae = __shfl(A, lane > 0 ? lane - 1 : 0);
        const float Be = __shfl(B, lane > 0 ? lane - 1 : 0);
        float dC = lane > 0 ? Ae * carry + Be : carry;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int tau = tau0 + lane * PER + i;
            const int t = F - 1 - tau;
            dC = a[i] * dC + b[i];
            if (t >= 0) {
                const float z = tanhf(gz[t]);
                const float f = sigm(gf[t]);
                const float o = sigm(go[t]);
                const float c = crow[t];
                const float cprev = t > 0 ? crow[t - 1] : 0.f;
                dz_[t] = dC * f * (1.f - z * z);
                df_[t] = dC * (z - cprev) * f * (1.f - f);
                do_[t] = dhrow[t] * c * o * (1.f - o);
            }
        }
        const float At = __shfl(A, 63), Bt = __shfl(B, 63);
        carry = At * carry + Bt;
    }
}

}  // namespace

extern "C" int pase_qrnn_scan_fwd(const float* gates, float* h_out, float* c_out, int S, int H, int F, int h_ctot,
                                  int h_coff, void* stream) {
    const long rows = (long)S * H;
    if (rows <= 0) return 0;
    PASE_LAUNCH(qrnn_fwd_kernel, dim3((unsigned)((rows + NT / 64 - 1) / (NT / 64))), dim3(NT), (hipStream_t)stream,
                gates, h_out, c_out, S, H, F, h_ctot, h_coff);
    PASE_CHECK_LAUNCH();
    return 0;
}

extern "C" int pase_qrnn_scan_bwd(const float* gates, const float* c_saved, const float* dh, float* dgates, int S,
                                  int H, int F, int dh_ctot, int dh_coff, void* stream) {
    const long rows = (long)S * H;
    if (rows <= 0) return 0;
    PASE_LAUNCH(qrnn_bwd_kernel, dim3((unsigned)((rows + NT / 64 - 1) / (NT / 64))), dim3(NT), (hipStream_t)stream,
                gates, c_saved, dh, dgates, S, H, F, dh_ctot, dh_coff);
    PASE_CHECK_LAUNCH();
    return 0;
}

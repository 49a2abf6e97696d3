// How does v_mfma_f32_32x32x16_bf16 round?  (round 3: is the split-bf16 contraction biased?)
// D = C + sum_k a_k b_k with a, b exact bf16 values chosen so that the exact result falls between fp32 neighbours.
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/experiments/mfma_round_probe tools/experiments/mfma_round_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned bf(float f) { return __float_as_uint(f) >> 16; }

// every lane gets the same fragments: A[i][k] = a[k], B[k][j] = b[k]  (k = 8*(lane>>5) + e)
__global__ void probe(const float* av, const float* bv, const float* cv, float* out, int ncase) {
    const int lane = threadIdx.x, fk = lane >> 5;
    for (int c = 0; c < ncase; ++c) {
        u32x4 a, b;
        for (int i = 0; i < 4; ++i) {
            a[i] = bf(av[c * 16 + fk * 8 + 2 * i]) | (bf(av[c * 16 + fk * 8 + 2 * i + 1]) << 16);
            b[i] = bf(bv[c * 16 + fk * 8 + 2 * i]) | (bf(bv[c * 16 + fk * 8 + 2 * i + 1]) << 16);
        }
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = cv[c];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        if (lane == 0) out[c] = acc[0];
    }
}

int main() {
    const int NC = 12;
    float a[NC][16] = {}, b[NC][16] = {}, c[NC] = {};
    const char* what[NC];
    const float u = ldexpf(1.f, -23);    // ulp(1.0)
    int n = 0;
    // 0: C = 1, one product 0.75 ulp          -> RNE 1+ulp, truncate 1
    what[n] = "C=1 + 0.75ulp (one product)"; c[n] = 1.f; a[n][0] = 0.75f; b[n][0] = u; ++n;
    what[n] = "C=1 + 0.25ulp (one product)"; c[n] = 1.f; a[n][0] = 0.25f; b[n][0] = u; ++n;
    what[n] = "C=1 - 0.25ulp (one product)"; c[n] = 1.f; a[n][0] = -0.25f; b[n][0] = u; ++n;   // below 1 the ulp halves: exact
    what[n] = "C=1 - 0.125ulp"; c[n] = 1.f; a[n][0] = -0.125f; b[n][0] = u; ++n;              // 1 - 2^-26: between 1-2^-24 and 1
    what[n] = "C=-1 - 0.75ulp"; c[n] = -1.f; a[n][0] = -0.75f; b[n][0] = u; ++n;
    what[n] = "C=-1 - 0.25ulp"; c[n] = -1.f; a[n][0] = -0.25f; b[n][0] = u; ++n;
    // 16 products of 0.125 ulp each (sum = 2 ulp): added one by one with truncation -> 1; summed first -> 1 + 2ulp
    what[n] = "C=1 + 16 x 0.125ulp"; c[n] = 1.f; for (int k = 0; k < 16; ++k) { a[n][k] = 0.125f; b[n][k] = u; } ++n;
    // 16 products of 0.046875 ulp (sum 0.75 ulp)
    what[n] = "C=1 + 16 x 0.046875ulp (=0.75)"; c[n] = 1.f; for (int k = 0; k < 16; ++k) { a[n][k] = 0.046875f; b[n][k] = u; } ++n;
    // tie: 0.5 ulp -> RNE to even (1.0); 1.5 ulp -> 1+2ulp
    what[n] = "C=1 + 0.5ulp (tie)"; c[n] = 1.f; a[n][0] = 0.5f; b[n][0] = u; ++n;
    what[n] = "C=1 + 1.5ulp (tie)"; c[n] = 1.f; a[n][0] = 1.5f; b[n][0] = u; ++n;
    // cancellation inside the products: +2^20 - 2^20 + 0.75 ulp
    what[n] = "C=1 + (2^10*2^10 - 2^10*2^10) + 0.75ulp"; c[n] = 1.f; a[n][0] = 1024.f; b[n][0] = 1024.f; a[n][1] = -1024.f; b[n][1] = 1024.f; a[n][2] = 0.75f; b[n][2] = u; ++n;
    // big product + small C: product 1.0, C = 0.75 ulp
    what[n] = "C=0.75ulp + 1*1"; c[n] = 0.75f * u; a[n][0] = 1.f; b[n][0] = 1.f; ++n;
    float *da, *db, *dc, *dout, out[NC];
    hipMalloc(&da, sizeof(a)); hipMalloc(&db, sizeof(b)); hipMalloc(&dc, sizeof(c)); hipMalloc(&dout, sizeof(out));
    hipMemcpy(da, a, sizeof(a), hipMemcpyHostToDevice); hipMemcpy(db, b, sizeof(b), hipMemcpyHostToDevice);
    hipMemcpy(dc, c, sizeof(c), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dc, dout, n);
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        double exact = c[i];
        for (int k = 0; k < 16; ++k) exact += (double)a[i][k] * b[i][k];
        const float rne = (float)exact;
        printf("%-44s exact %.10e  RNE %.10e  mfma %.10e  (mfma - exact)/ulp %+.3f  %s\n", what[i], exact, rne, out[i],
               (out[i] - exact) / u, out[i] == rne ? "= RNE" : "!= RNE");
    }
    return 0;
}

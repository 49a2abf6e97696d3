// Round 6: is  v_dot2c_f32_bf16 d, pk, K  (d += pk.lo * K.lo + pk.hi * K.hi)  with K = (-1, 0) / (0, -1) an EXACT replacement of the
// two instructions  t = pk << 16 (or pk & 0xffff0000);  r = r - t  in the operand split (pase_split_bf16x3_rne: residuals x - hi,
// (x - hi) - mid)?  The residual is exactly representable, so any correctly rounded evaluation returns it -- unless the dot
// instruction flushes denormals, rounds the products, or mishandles signed zeros / non-finite neighbours.
// Also: throughput of the conversion both ways (clock64 around 4096 conversions of one wave; VALU only, no MFMA beside it).
//   build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/dot2c_probe tools/experiments/dot2c_probe.hip && /tmp/dot2c_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float dot2c(float acc, unsigned pk, unsigned k) {
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(pk), "v"(k));
    return acc;
}

// out[6 * i + ..] = {r1a, r1b, r2a, r2b} by subtraction, then the same by dot2c: two stages of residuals of the pair (x[2i], x[2i+1])
__global__ void probe(const float* x, float* sub, float* dot, unsigned* pieces, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    // reference: the shipped arithmetic
    float ra = a, rb = b;
    unsigned p0 = cvt_pk(ra, rb);
    ra -= __uint_as_float(p0 << 16);
    rb -= __uint_as_float(p0 & 0xffff0000u);
    unsigned p1 = cvt_pk(ra, rb);
    float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
    sub[4 * i] = ra; sub[4 * i + 1] = rb; sub[4 * i + 2] = sa; sub[4 * i + 3] = sb;
    pieces[3 * i] = p0; pieces[3 * i + 1] = p1; pieces[3 * i + 2] = cvt_pk(sa, sb);
    // candidate
    const unsigned KLO = 0x0000BF80u, KHI = 0xBF800000u;      // (-1, 0) and (0, -1) as packed bf16
    float da = dot2c(a, p0, KLO), db = dot2c(b, p0, KHI);
    unsigned q1 = cvt_pk(da, db);
    float ea = dot2c(da, q1, KLO), eb = dot2c(db, q1, KHI);
    dot[4 * i] = da; dot[4 * i + 1] = db; dot[4 * i + 2] = ea; dot[4 * i + 3] = eb;
}

template <bool DOT>
__global__ void speed(const float* x, unsigned* out, long long* ticks, int reps) {
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = x[threadIdx.x * 8 + e];
    unsigned acc = 0;
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        float q[8];
        for (int e = 0; e < 8; ++e) q[e] = v[e] + (float)r;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned pk = cvt_pk(q[2 * i], q[2 * i + 1]);
                acc ^= pk;
                if (DOT) {
                    q[2 * i] = dot2c(q[2 * i], pk, 0x0000BF80u);
                    q[2 * i + 1] = dot2c(q[2 * i + 1], pk, 0xBF800000u);
                } else {
                    q[2 * i] -= __uint_as_float(pk << 16);
                    q[2 * i + 1] -= __uint_as_float(pk & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc ^= cvt_pk(q[2 * i], q[2 * i + 1]);
    }
    const long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    const int n = 1 << 22;
    std::vector<float> hx(2 * n);
    srand(7);
    auto rnd = []() { return (unsigned)rand() ^ ((unsigned)rand() << 15) ^ ((unsigned)rand() << 30); };
    for (int i = 0; i < 2 * n; ++i) {
        unsigned u = rnd();
        const int cls = i % 8;
        if (cls == 0) u = (u & 0x807fffffu) | ((unsigned)(rand() % 24) << 23);              // tiny: residuals go denormal / zero
        else if (cls == 1) u = (u & 0x807fffffu) | ((unsigned)(230 + rand() % 24) << 23);   // huge
        else if (cls == 2) u = (u & 0x807fffffu) | ((unsigned)(100 + rand() % 56) << 23);   // wide range
        else u = (u & 0x807fffffu) | ((unsigned)(120 + rand() % 12) << 23);                 // O(1)
        if (i % 4099 == 0) u &= 0xffff0000u;                                                  // exactly a bf16: zero residual
        if (i % 8191 == 0) u = 0x80000000u;                                                   // -0
        memcpy(&hx[i], &u, 4);
    }
    float *dx, *ds, *dd;
    unsigned* dp;
    hipMalloc(&dx, 8 * n); hipMalloc(&ds, 16 * n); hipMalloc(&dd, 16 * n); hipMalloc(&dp, 12 * n);
    hipMemcpy(dx, hx.data(), 8 * n, hipMemcpyHostToDevice);
    probe<<<n / 256, 256>>>(dx, ds, dd, dp, n);
    std::vector<unsigned> hs(4 * n), hd(4 * n);
    hipMemcpy(hs.data(), ds, 16 * n, hipMemcpyDeviceToHost);
    hipMemcpy(hd.data(), dd, 16 * n, hipMemcpyDeviceToHost);
    long bad = 0, bad_zero_sign = 0, bad_denorm = 0, denorm_seen = 0;
    for (long i = 0; i < 4L * n; ++i) {
        const unsigned a = hs[i], b = hd[i];
        const bool den = (a & 0x7f800000u) == 0 && (a & 0x007fffffu) != 0;
        denorm_seen += den;
        if (a != b) {
            ++bad;
            if ((a | b) == 0x80000000u) ++bad_zero_sign;
            else if (den) ++bad_denorm;
            if (bad <= 8) printf("  mismatch at %ld: sub %08x dot %08x\n", i, a, b);
        }
    }
    printf("residuals compared: %ld; mismatches %ld (sign-of-zero only %ld, denormal residuals %ld of %ld denormal residuals seen)\n",
           4L * n, bad, bad_zero_sign, bad_denorm, denorm_seen);
    // throughput: one wave per SIMD x 4 per CU, 1024 blocks
    float* sx; unsigned* so; long long* st;
    hipMalloc(&sx, 64 * 8 * 4); hipMalloc(&so, 64 * 4); hipMalloc(&st, 1024 * 8);
    hipMemcpy(sx, hx.data() + 1000, 64 * 8 * 4, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which) {
        for (int it = 0; it < 2; ++it) {
            if (which) speed<true><<<1024, 64>>>(sx, so, st, 512);
            else speed<false><<<1024, 64>>>(sx, so, st, 512);
        }
        std::vector<long long> ht(1024);
        hipMemcpy(ht.data(), st, 1024 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : ht) s += (double)v;
        printf("%s: %.1f clocks per 8-element conversion (one wave per block, 1024 blocks)\n", which ? "dot2c" : "shift/and + sub", s / 1024 / 512);
    }
    return 0;
}

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
// The weight operand never touches LDS: pase_x6c_pack stores it in fragment order [32-row tile][step][plane][lane],
// so a wave's A fragment is one coalesced 1 KB buffer_load_dwordx4 per plane (descriptor + constant lane offset + a scalar
// offset that advances per step: no vector address arithmetic), prefetched two steps ahead, and each of a workgroup's
// compute waves owns different rows (no redundant loads).
//
// Tile and roles.  512 threads = 8 waves, one workgroup per CU, persistent grid (<= 256 workgroups walk the (split-K slice,
// tile) items).  Waves 0-3 only multiply: wave tile 32 rows x 128 columns (4 B tiles x 2 accumulators = 128 VGPRs), workgroup
// tile 128 x 128 (waves 4 x 1) or 64 x 256 (waves 2 x 2, launches of at most 64 rows); the B fragments of the next half step
// are read from LDS one per two MFMAs of the current one.  Waves 4-7 only stage, into two LDS buffers of KGS k-groups
// (16 channels' each) x taps': activations that are split while they are staged travel through registers, loaded two stages
// ahead by inline-asm loads with hand-counted s_waitcnt (x6c_gload / x6c_vmwait_slots / x6c_claim: the compiler's own
// bookkeeping made that pipeline synchronous); pre-split operands (ZP weight-gradient planes, XP activation planes) are copied
// by global_load_lds_dwordx4 without registers.  One barrier per stage (KGS * taps' steps of 24 MFMAs per compute wave).
// Template instantiations: <NPOS, KGS_T> = positions per stage row / k-groups a stage buffer holds; TM = contraction over
// positions (weight gradients); ZP = pre-split staged operand; NARROW = the 64 x 256 tile.
#include <cstdlib>
#include <type_traits>

#include "conv_x6c.h"

namespace {

constexpr int NT = 512;        // 4 compute waves (one per SIMD) + 4 staging waves
constexpr int HALO_MAX = 64;      // extra positions a stage holds beyond its BN columns (halo of every sequence touched)
constexpr int TMZ_KGS = 5;        // k-groups per stage buffer of the weight-gradient kernel on pre-split planes (<128, TMZ_KGS, true, true>)

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

// ZP weight gradients of strided layers: the GEMM columns of one channel are its taps ordered PHASE BY PHASE -- column r is tap
// kk0 + stride * dd, where the first `rem` phases (kk0 < rem) hold n + 1 taps and the others n (taps = stride * n + rem).
// Consecutive columns (= consecutive lanes of the staging waves) then read the SAME plane row at consecutive shifts, i.e.
// overlapping 16-byte windows 2 bytes apart, which the memory pipeline coalesces; in natural tap order consecutive lanes
// cycle through the `stride` phase rows (52 KB apart) and every lane is its own request (measured: the stride-4 / stride-10
// decoder layers 1.6x SLOWER than the fp32-staged form).  Stride 1: the identity.
__device__ __forceinline__ int zp_tap_of(int r, const PaseX6cPlan& pl) {
    const int n1 = pl.zp_n + 1, head = pl.zp_rem * n1;
    if (r < head) {
        const int kk0 = (int)div_magic((unsigned)r, pl.rctx_magic);           // r / (n + 1)
        return kk0 + pl.t_stride * (r - kk0 * n1);
    }
    const int r2 = r - head;
    const int q = (int)div_magic((unsigned)r2, pl.cout_magic);                // r2 / n
    return pl.zp_rem + q + pl.t_stride * (r2 - q * pl.zp_n);
}

#ifdef PASE_X6C_TRACE   // tools/trace_x6c.py only: per-item phase timestamps (shader clock) of workgroups 0 and 131
#define X6C_TRACE_ITEMS 64
__device__ unsigned long long g_x6c_trace[2 * X6C_TRACE_ITEMS * 20];
#define X6C_STAMP(slot)                                                                                     \
    do {                                                                                                    \
        if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 131) && trace_n < X6C_TRACE_ITEMS)               \
            g_x6c_trace[((blockIdx.x ? 1 : 0) * X6C_TRACE_ITEMS + trace_n) * 20 + (slot)] = clock64();      \
    } while (0)
#define X6C_TRACE_NEXT() ++trace_n
#ifdef PASE_X6C_TRACE_FINE      // stamps inside the epilogue (each costs an s_memtime round trip: ~500 clocks)
#define X6C_FSTAMP(slot) X6C_STAMP(slot)
#else
#define X6C_FSTAMP(slot)
#endif
// accumulate the cycles between X6C_T0() and X6C_TACC(slot) into a per-item sum
#define X6C_T0() const unsigned long long t0_ = clock64()
#define X6C_TACC(slot)                                                                                      \
    do {                                                                                                    \
        if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 131) && trace_n < X6C_TRACE_ITEMS)               \
            g_x6c_trace[((blockIdx.x ? 1 : 0) * X6C_TRACE_ITEMS + trace_n) * 20 + (slot)] += clock64() - t0_; \
    } while (0)
#else
#define X6C_STAMP(slot)
#define X6C_FSTAMP(slot)
#define X6C_TRACE_NEXT()
#define X6C_T0()
#define X6C_TACC(slot)
#endif

// uniform (scalar-unit) loads of per-channel on-load parameters: the values stay in SGPRs and no vector-memory
// instruction (and no vmcnt wait, which would drain the prefetched operands) is spent on them
#ifdef PASE_HIPEMU
__device__ __forceinline__ void sload8(const float* q, float (&o)[8]) {
    for (int i = 0; i < 8; ++i) o[i] = q[i];
}
__device__ __forceinline__ float sload1(const float* q) { return *q; }
__device__ __forceinline__ void sload32(const float* q, float (&o)[32]) {
    for (int i = 0; i < 32; ++i) o[i] = q[i];
}
__device__ __forceinline__ void sload8x2(const float* q0, const float* q1, float (&o0)[8], float (&o1)[8]) {
    sload8(q0, o0);
    sload8(q1, o1);
}
#else
typedef float pase_f8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void sload8(const float* q, float (&o)[8]) {
    pase_f8 v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(q) : "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = v[i];
}
__device__ __forceinline__ float sload1(const float* q) {
    float v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(q) : "memory");
    return v;
}
// 32 consecutive floats (the bias of a wave's 32 rows): the scalar cache answers in a few hundred clocks, while a vector load
// issued at this point queues behind the staging waves' loads for the NEXT tile (3.7 k clocks measured, tools/trace_x6c.py)
typedef float pase_f16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void sload32(const float* q, float (&o)[32]) {
    pase_f16 v0, v1;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(v0), "=&s"(v1)
                 : "s"(q)
                 : "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        o[i] = v0[i];
        o[16 + i] = v1[i];
    }
}
// two arrays, one wait
__device__ __forceinline__ void sload8x2(const float* q0, const float* q1, float (&o0)[8], float (&o1)[8]) {
    pase_f8 v0, v1;
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(v0), "=&s"(v1)
                 : "s"(q0), "s"(q1)
                 : "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        o0[i] = v0[i];
        o1[i] = v1[i];
    }
}
#endif

// Activation loads of the staging waves: a uniform base (SGPR pair) + a 32-bit per-lane byte offset, issued two stages before
// they are consumed (x6c_gload below; hidden from the compiler and waited for by hand, see there).
// Padding: validity is carried as INTEGER bits (okb, xmask), never as bools -- a bool per element becomes a 64-bit lane
// mask in an SGPR pair, 8 per slot x 4 slots x 3 register sets, i.e. > 100 spilled SGPRs and a select chain per element.
__device__ __forceinline__ int x6c_reflect(int u, int T) {              // single reflection about 0 and T - 1
    u = max(u, -u);
    return min(u, 2 * (T - 1) - u);
}
__device__ __forceinline__ unsigned x6c_in_range(int u, int T) {        // 1 when 0 <= u < T (|u|, T < 2^30)
    return (unsigned)(~u & (u - T)) >> 31;
}
__device__ __forceinline__ float x6c_keep(float v, unsigned mask, int e) {      // v when bit e of mask is set, else +0
    const int keep = (int)(mask << (31 - e)) >> 31;
    int bits;
    __builtin_memcpy(&bits, &v, 4);
    bits &= keep;
    __builtin_memcpy(&v, &bits, 4);
    return v;
}
// 16 bytes at 2-byte granularity (one global_load_dwordx4: global accesses need no alignment on gfx950)
struct __attribute__((packed, aligned(2))) X6cU16 { u32x4 v; };
__device__ __forceinline__ u32x4 x6c_load16u(const unsigned short* q) {
#ifdef PASE_HIPEMU
    u32x4 v;
    __builtin_memcpy(&v, q, 16);
    return v;
#else
    return reinterpret_cast<const X6cU16*>(q)->v;
#endif
}
// Buffer-descriptor loads (weight fragments): address = descriptor base (4 SGPRs) + per-lane byte offset (one VGPR that never
// changes) + scalar byte offset (one SGPR, advanced per step) -- no vector address arithmetic inside the loop
#ifdef PASE_HIPEMU
struct X6cRsrc { const char* base; };
__device__ __forceinline__ X6cRsrc x6c_make_rsrc(const void* base) { return X6cRsrc{reinterpret_cast<const char*>(base)}; }
__device__ __forceinline__ u32x4 x6c_buffer_load16(const X6cRsrc& r, unsigned voff, unsigned soff) {
    u32x4 v;
    __builtin_memcpy(&v, r.base + (size_t)voff + (size_t)soff, 16);
    return v;
}
#else
typedef __amdgpu_buffer_rsrc_t X6cRsrc;
__device__ __forceinline__ X6cRsrc x6c_make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xffffffffu, 0x00020000);
}
__device__ __forceinline__ u32x4 x6c_buffer_load16(X6cRsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
#endif
// One 16-byte chunk per lane straight from global memory into LDS (global_load_lds_dwordx4): destination = wave-uniform LDS
// address + 16 * lane, no staging registers, no ds_write; complete once the wave's vmcnt has drained (x6c_vm_drain) AND a
// barrier has passed before another wave reads it
#ifdef PASE_HIPEMU
__device__ __forceinline__ void x6c_load_lds16(const void* src, u32x4* lds_wave_base, int lane) {
    __builtin_memcpy(&lds_wave_base[lane], src, 16);
}
__device__ __forceinline__ void x6c_vm_drain() {}
#else
__device__ __forceinline__ void x6c_load_lds16(const void* src, u32x4* lds_wave_base, int) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void x6c_vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
// ... and one dword per lane (global_load_lds_dword): destination = wave-uniform LDS address + 4 * lane, for the active lanes
#ifdef PASE_HIPEMU
__device__ __forceinline__ void x6c_load_lds4(const void* src, float* lds_wave_base, int lane) {
    __builtin_memcpy(&lds_wave_base[lane], src, 4);
}
#else
__device__ __forceinline__ void x6c_load_lds4(const void* src, float* lds_wave_base, int) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
#endif
// The same copy HIDDEN from the compiler (symmetric forms: the waves that copy are the waves that read).  Visible, a pending LDS DMA
// makes the compiler guard every later LDS read that may alias its destination with s_waitcnt vmcnt(0) and turns every
// __syncthreads into a full drain; as inline asm it is invisible, and the wave waits for it by hand in front of the barrier that
// publishes the stage.  M0 (the LDS base of the copy) is saved and restored: a reserved register the compiler may hold a value in.
#ifdef PASE_HIPEMU
__device__ __forceinline__ void x6c_dma16_hidden(const void* src, u32x4* lds_wave_base, int lane) {
    __builtin_memcpy(&lds_wave_base[lane], src, 16);
}
#else
__device__ __forceinline__ void x6c_dma16_hidden(const void* src, u32x4* lds_wave_base, int) {
    const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) u32x4*)lds_wave_base);
    unsigned m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved)
                 : "v"(src), "s"(lds)
                 : "memory");
}
#endif
struct alignas(16) X6cF4 { float x, y, z, w; };
// Staging registers: eight fp32 values per slot as two 4-vectors (the row-coalesced weight-gradient path fills them with two
// global_load_dwordx4, every other path with eight global_load_dword).
#ifdef PASE_HIPEMU
typedef float x6c_f4 __attribute__((vector_size(16)));
#else
typedef float x6c_f4 __attribute__((ext_vector_type(4)));
#endif
#if defined(PASE_HIPEMU) || defined(PASE_X6C_AUTOWAIT)
// (emulator, and A/B builds with -DPASE_X6C_AUTOWAIT: plain loads, the compiler's own s_waitcnt bookkeeping)
template <int E, int RS>
__device__ __forceinline__ void x6c_gload(x6c_f4 (&q)[2], const float* base, unsigned voff_bytes) {
    q[E >> 2][E & 3] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + voff_bytes);
}
template <int E, int RS>
__device__ __forceinline__ void x6c_gload2(x6c_f4 (&q)[2], const float* base, unsigned voff_bytes) {
    const char* a = reinterpret_cast<const char*>(base) + voff_bytes;
    float lo, hi;
    __builtin_memcpy(&lo, a, 4);
    __builtin_memcpy(&hi, a + 4, 4);
    q[E >> 2][E & 3] = lo;
    q[E >> 2][(E & 3) + 1] = hi;
}
template <int RS>
__device__ __forceinline__ void x6c_gload_x8(x6c_f4 (&q)[2], const float* base, unsigned voff_bytes) {
    const X6cF4* s4 = reinterpret_cast<const X6cF4*>(reinterpret_cast<const char*>(base) + voff_bytes);
    const X6cF4 lo = s4[0], hi = s4[1];
    q[0] = x6c_f4{lo.x, lo.y, lo.z, lo.w};
    q[1] = x6c_f4{hi.x, hi.y, hi.z, hi.w};
}
template <int N>
__device__ __forceinline__ void x6c_vmwait() {}
template <int RS>
__device__ __forceinline__ void x6c_claim(x6c_f4 (&)[2]) {}
#else
// The staging waves' loads are HIDDEN from the compiler (cdna_hip_programming.md 5.7 form (ii)): issued as inline asm two
// stages before their conversion, waited for with hand-counted s_waitcnt vmcnt(N) (every live slot issues the same number
// of them and nothing else of a staging wave uses the vector memory path inside the stage loop), and tied to their first use
// by x6c_claim.  Left to the compiler the pipeline was synchronous: the per-slot branches, the rotated register sets and
// address temporaries allocated on top of the load destinations made it wait vmcnt(0) in front of every load group and
// every conversion -- the compute waves of the 1x1 / stride-2 / swapped launches spent 11 ... 60 % of their loop in the stage
// barrier (tools/trace_x6c.py: LPS data gradient 60 %, LPS weight gradient 47 %, blocks 6 / 7 11 %).
// RS = the register set the load belongs to, spelled into the asm text: IDENTICAL asm statements in sibling branches are
// merged by the optimiser into one load whose result is then copied -- a copy of a register whose load has not landed
// (round 5, seen in the ISA of a loop that chose the set at run time).  Distinct asm strings cannot be merged: every load
// writes its home register, and pase_amd/build.py's ISA lint checks set membership.
template <int E, int RS>
__device__ __forceinline__ void x6c_gload(x6c_f4 (&q)[2], const float* base, unsigned voff_bytes) {
    asm volatile("global_load_dword %0, %1, %2 ; staging set %3"
                 : "=v"(q[E >> 2][E & 3])
                 : "v"(voff_bytes), "s"(base), "n"(RS)
                 : "memory");
}
// elements E, E + 1 (E even) = two consecutive samples: one 8-byte load at 4-byte alignment (strided launches with an even
// stride: the phases of a channel are consecutive samples and a pair never straddles two channels)
template <int E, int RS>
__device__ __forceinline__ void x6c_gload2(x6c_f4 (&q)[2], const float* base, unsigned voff_bytes) {
    static_assert((E & 1) == 0, "pairs start at even elements");
    if constexpr ((E & 3) == 0)
        asm volatile("global_load_dwordx2 %0, %1, %2 ; staging set %3" : "=v"(q[E >> 2].lo) : "v"(voff_bytes), "s"(base), "n"(RS) : "memory");
    else
        asm volatile("global_load_dwordx2 %0, %1, %2 ; staging set %3" : "=v"(q[E >> 2].hi) : "v"(voff_bytes), "s"(base), "n"(RS) : "memory");
}
template <int RS>
__device__ __forceinline__ void x6c_gload_x8(x6c_f4 (&q)[2], const float* base, unsigned voff_bytes) {
    asm volatile("global_load_dwordx4 %0, %2, %3 ; staging set %4\n\tglobal_load_dwordx4 %1, %2, %3 offset:16"
                 : "=&v"(q[0]), "=&v"(q[1])
                 : "v"(voff_bytes), "s"(base), "n"(RS)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void x6c_vmwait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int RS>
__device__ __forceinline__ void x6c_claim(x6c_f4 (&q)[2]) {
    // (the operands are spelled into the text: pase_amd/build.py's ISA lint checks that the registers the compiler hands to the
    //  conversion are the very registers the set's loads were issued into -- a value that was copied while in flight is stale)
    asm volatile("; claim staging set %2 regs %0 %1" : "+v"(q[0]), "+v"(q[1]) : "n"(RS));
}
#endif
// wait until at most `per_slot` * nslots of this wave's loads are outstanding (nslots: uniform; per_slot = the FEWEST loads a
// live slot issues: 8, 2 on the row-coalesced weight-gradient path, 4 where interior slots load pairs -- slots that issue more
// only make the wait stricter than needed).  Rounded down to an instantiated count: stricter, never looser.
__device__ __forceinline__ void x6c_vmwait_slots(int nslots, int per_slot) {
    const int n = nslots <= 0 ? 0 : per_slot * nslots;
    if (n >= 40) x6c_vmwait<40>();
    else if (n >= 32) x6c_vmwait<32>();
    else if (n >= 24) x6c_vmwait<24>();
    else if (n >= 20) x6c_vmwait<20>();
    else if (n >= 16) x6c_vmwait<16>();
    else if (n >= 12) x6c_vmwait<12>();
    else if (n >= 10) x6c_vmwait<10>();
    else if (n >= 8) x6c_vmwait<8>();
    else if (n >= 6) x6c_vmwait<6>();
    else if (n >= 4) x6c_vmwait<4>();
    else if (n >= 2) x6c_vmwait<2>();
    else x6c_vmwait<0>();
}

// NPOS: positions (16-byte chunks) per (plane, fk) row of a k-group; KGS_T: k-groups a stage buffer holds.
//   <192, 2>: convolutions (128 columns + up to 64 halo positions)      <128, 3>: 1x1 layers (no halo)
// The grid is PERSISTENT: workgroup b works through the items b, b + gridDim.x, ... (item = (split-K slice, tile)); the
// staging waves start on the next item's first stage while the compute waves are still in the epilogue of the current one.
// TM (weight gradients, see pase_x6c_wgrad): the contraction runs over POSITIONS -- k-group = 16 consecutive positions q
// of one sequence -- and the columns are (channel, tap) pairs: column j reads x[s][ci][q * stride + kk * tapstep - padL].
// Same staging machinery (a chunk is 8 consecutive q of one column, i.e. 8 strided samples of one channel row), but the
// per-channel on-load parameters belong to the lane (its column), not to the element.
// ZP (TM only, pl.zp): the staged operand is PRE-SPLIT -- phase-decomposed bf16 planes of z~ with the on-load transform and
// the padding already applied (pack_zph_kernel): plane[pz][row = ci * stride + b][s][i] = piece pz of z~[s][ci][stride * (i +
// dmin) + b].  Column (ci, kk) with kk * tapstep - padL = d * stride + b is then row (ci, b) shifted by d, a chunk (8 consecutive
// positions of one column) is 16 contiguous bytes of a plane at 2-byte granularity, and staging is three 16-byte loads + three
// ds_write_b128 per slot: no conversion, no padding arithmetic, no per-element loads (the round-3 T-mode spent 1.9k cycles of
// staging-wave issue per 768 cycles of MFMA on exactly those).  The bias column is one more plane row holding 1.0.
// ZP on a convolution launch (!TM, pl.xp; "XP"): the ACTIVATION is pre-split -- channel-minor bf16 planes with the on-load
// transform and the (zero / reflect) padding applied, written once by pase_pack_xp in the LDS image's own chunk order
//   plane[pz][k-group g][octet fk][s][up]  = 16-byte chunk: pieces pz of channels 16 g + 8 fk .. + 7 at padded position up
// (up = q + tap, Tpad = Ncols + A - 1 per sequence; stride-1 launches only).  A wave's staging load is then 64 consecutive
// chunks = 1 KB contiguous.  Worth it where every staged element used to be converted many times over -- a column tile is
// re-staged by each of the M / 128 row tiles that need it (169 times on the 21 525-row heads) -- and the k-loop has one or
// two taps to amortise the conversion over.
// NARROW (convolutions of at most 64 rows: block 1 of the encoder): workgroup tile 64 x 256, compute waves 2 (rows) x 2 (column
// halves) -- the 128-row tile would spend half of every MFMA on zero rows.  Same wave tile (32 x 128), same loop; the stage
// holds 256 + 64 positions (<320, 2>).
// SYM (round 6; convolutions on a pre-split activation, pl.sym): the SYMMETRIC form -- no staging waves.  All eight waves
// multiply (wave w owns rows 32 w .. of a 256 x 128 workgroup tile: two multiplying waves per SIMD cover each other's waits and
// each other's epilogues' latencies, the staged columns of a k-group feed twice the MFMAs) and share out the stage's LDS DMA
// among themselves: unit u = (k-group, octet, 64-position block) belongs to wave u mod 8, at most two units = six hidden DMA
// instructions per wave and stage, issued at the top of the stage's first two steps (see x6c_wgrad_sym_kernel for why hidden
// and why three at a time).
// DUO (SYM only): the same with FOUR waves per workgroup -- 128 x 128 tile, 256 threads, TWO workgroups per CU (73 KB of LDS each).
// The two workgroups of a CU run out of step, so one's epilogue (the matrix pipe idles: 13 k of an LPS head's 38 k clocks) lies
// under the other's MFMA loop; the price is that they do not share a staged k-group (256 bytes from L2 per MFMA, as in the
// staging-wave form) and that each wave issues up to nine DMA instructions per stage.  Same row tiling as the staging-wave form:
// no weight-pack or round-count consequences.
template <int NPOS, int KGS_T, bool TM = false, bool ZP = false, bool NARROW = false, bool SYM = false, bool DUO = false>
__global__ void __launch_bounds__(NT, 2) conv_x6c_kernel(PaseConvGemm p, PaseX6cPlan pl) {
    static_assert(!(NARROW && TM), "the 64 x 256 tile is a convolution tile");
    static_assert(!SYM || (ZP && !TM && !NARROW), "the symmetric form copies a pre-split activation");
    static_assert(!DUO || SYM, "DUO is a symmetric form");
    constexpr int NTH = DUO ? 256 : NT;            // threads of the workgroup
    constexpr int SYW = DUO ? 4 : 8;               // SYM: waves that share out the stage's copy
    constexpr int SYU = DUO ? 3 : 2;               // ... units (k-group, octet, 64-position block) per wave at most
    constexpr int WM = SYM ? SYW : (NARROW ? 2 : 4), WN = NARROW ? 2 : 1, NBT = 4;
    constexpr int BM = 32 * WM, BN = 32 * NBT * WN;
    constexpr int NPS = (NPOS + 127) / 128;        // position slots per thread and k-group
    constexpr int NSLOT = NPS * KGS_T;
    constexpr int PLANE = 2 * NPOS;                // chunks per plane of one k-group: [fk][pos]
    constexpr int KGC = 3 * PLANE;                 // chunks per k-group
    constexpr int BUF = KGS_T * KGC;               // chunks per stage buffer
    constexpr int XR = 3;                          // register sets of the staging waves: loads run XR - 1 stages ahead
    constexpr int RED_CHUNKS = (int)(sizeof(float) * WN * BM * 2 / 16);
    // TM, transposed accumulation (modes 2 / 3): each compute wave turns 64 columns x 32 rows of its tile through a private
    // [64][33]-float LDS block so that the atomics run along the rows of dw
    // (pre-split weight gradients are tmode 1: row-contiguous atomics, no transposed flush)
    constexpr int TR_FLOATS = 64 * 33;
    constexpr int TR_CHUNKS = (TM && !ZP) ? (4 * TR_FLOATS * 4 + 15) / 16 : 0;
    // (Round 6 measured the PACKED operand through LDS as well -- 12 KB per k-group copied by the staging waves' DMA one stage
    //  ahead, no vector-memory instruction left in the compute waves' loop, three k-groups per stage to fit 2 x 24 KB per
    //  k-group: 4 ... 9 % SLOWER on all eight pre-split weight gradients of the PASE+ step, same box.  An LDS-DMA instruction
    //  costs the issuing wave 100+ cycles beside busy LDS / MFMA pipes; six per step and staging wave is the step's length.)
    // two stage buffers + the epilogue scratch (its own region: the next item's first stage is staged during the epilogue)
    __shared__ __attribute__((aligned(16))) u32x4 Xs[2 * BUF + RED_CHUNKS + TR_CHUNKS];
    float (*red)[BM][2] = reinterpret_cast<float (*)[BM][2]>(&Xs[2 * BUF]);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = pase_uniform(tid >> 6);
    // Roles.  Waves 0-3 (one per SIMD) only multiply: A fragments from global memory, B fragments from LDS, MFMA.  Waves
    // 4-7 only stage: activation loads, on-load transform, operand split, LDS writes, ahead of the compute waves.  The two
    // kinds of work run on different pipes (matrix core / vector ALU) of the same SIMD concurrently, which an in-order
    // wave doing both cannot arrange; each role has its own vmcnt counter, so a staging wave waiting for activations from
    // HBM never holds back a weight fragment.  One barrier per stage joins them.
    const bool stager = !SYM && wave >= 4;         // uniform
    const int wm = SYM ? wave : (NARROW ? (wave & 1) : (wave & 3)), wn = NARROW ? ((wave >> 1) & 1) : 0;
    const int fr = lane & 31, fk = lane >> 5;
    const int fkL = (wave >> 1) & 1;               // stager: octet of the k-group this wave stages (uniform)
    const int whalf = wave & 1;                    // stager: which 64 of a slot's 128 positions (uniform)

    const int ntot = p.S * p.Ncols;
    const int ntiles = pl.n_row_tiles * pl.n_col_tiles;
    const int nitems = ntiles * pl.splitk;
    const int H = pl.A - 1;                        // halo positions per sequence segment
    const int segL = p.Ncols + H;                  // span positions of a full middle segment
    const int KGS = pl.KGS;
    const int GS = (pl.G + KGS - 1) / KGS;         // stages
    const int g_per = (GS + pl.splitk - 1) / pl.splitk;
    const int nsteps = KGS * pl.A;                 // MFMA steps per stage
    const bool has_aff = p.in_scale != nullptr, has_alpha = p.in_alpha != nullptr;       // uniform
    // convolutions whose bias index is the tile row (no pixel shuffle, no spectrum post-op): see the accumulator initialisation
    const bool bias_init = !TM && p.bias != nullptr && p.ps == 1 && !(p.x6_ctl & 64) && p.post_op != PASE_POST_POW &&
                           p.post_op != PASE_POST_LOGPOW && p.post_op != PASE_POST_MAG;
    const float* xbase = p.x + (size_t)p.x_coff * p.Tin;
    const int prm_n = GS * KGS * 16;               // channels' incl. the zero groups that fill the last stage
    const float* prm = reinterpret_cast<const float*>(reinterpret_cast<const u32x4*>(p.wx6) + pl.pack_chunks);
    int bsel = 0;                                  // stage buffer of the next stage to be multiplied / first staged
#ifdef PASE_X6C_TRACE
    int trace_n = 0;
#endif
    // Start the workgroups out of phase (16 phases, `stagger` x 512 clocks apart): tiles of one launch take the same time, so
    // 256 persistent workgroups started together reach their epilogues together and the store bursts queue up behind one
    // another while the memory system idles during the main loops
    if (pl.stagger > 0) {
        const int ph = ((int)blockIdx.x >> 3) & 15;
        for (int i = 0; i < ph * pl.stagger; ++i) PASE_SLEEP(8);
    }

#ifndef PASE_HIPEMU
    {   // wave priorities (uniform): the staging waves are the younger half of the workgroup and lose every VALU issue
        // arbitration against a compute wave that always has an MFMA waiting for the pipe
        const int pr = stager ? (pl.prio & 3) : ((pl.prio >> 2) & 3);
        if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
        else if (pr == 3) __builtin_amdgcn_s_setprio(3);
    }
#endif
    if (stager) {
    // ---- loader state.  Slot (k-group kg, position slot ps): this thread stages position
    //   i = 128 ps + 64 (whalf ^ (kg & 1)) + lane        (odd k-groups swap the wave halves, so that the mostly
    // empty second position slot -- the halo -- is shared out evenly), eight channels' of octet fkL.
    // Per position: element offset of tap-0 / phase-0 (sequence and time), whether it is a real position, whether
    // all its phases are in-range samples.
    constexpr int NPAR = KGS_T > 1 ? 2 : 1;
    constexpr unsigned POS_ALL = (1u << (NPAR * NPS)) - 1u;
    // per-item state of the staging waves (set by setup_item)
    int g_begin = 0, nst = 0;
    int pos_u0[NPAR][NPS];
    unsigned pos_voff[NPAR][NPS];
    unsigned pos_sbase[NPAR][NPS];                  // element offset of the position's sequence (0 when not a real position)
    unsigned pos_xoff[NPAR][NPS];                   // XP: chunk offset s * Tpad + up of the position inside a (g, fk) row
    unsigned pos_valid = 0u, pos_inter = 0u;       // bit par * NPS + ps
    unsigned live = 0u, full = 0u;
    unsigned inter_slots = 0u;     // slots whose 64 positions (of this wave) are all in-range samples: no padding arithmetic
    // TM: per column (lane): channel row offset + tap offset (relative to the smallest tap offset), the tap offset itself,
    // on-load parameters, "all-ones column" flag (bias gradient); running column sums of the staged values (tmode 2)
    int t_koff[NPAR][NPS];
    float t_sc[NPAR][NPS], t_sh[NPAR][NPS], t_al[NPAR][NPS];
    float t_colsum[NPAR][NPS];
    unsigned t_ones = 0u;
    bool t_any_ones = false;
    int t_n0 = 0, t_mt = 0;
    // TM, pl.t_vec (g staged, modes 2 / 3): row-coalesced mapping -- slot sl of this lane is column
    //   n0 + 32 (wave - 4) + 16 (sl >> 1) + (lane & 15),   chunk c = 4 (sl & 1) + (lane >> 4) of the stage's 8 (kg = c >> 1, fk = c & 1)
    // so one load instruction covers 16 rows x 128 contiguous bytes (whole cache lines, each fetched once) instead of 64 rows x
    // 32 bytes, and a 16-lane group of the LDS write covers 16 consecutive columns of one (kg, fk) row (conflict-free)
    unsigned zp_col[NPAR];          // ZP: element offset of this lane's column inside a plane (row, shift, octet)
#pragma unroll
    for (int par = 0; par < NPAR; ++par) zp_col[par] = 0u;
    unsigned v_voff[2] = {0u, 0u};
    bool v_ok[2] = {false, false};
    float v_al[2] = {1.f, 1.f}, v_sum[2] = {0.f, 0.f};
    const int t_kmin = -p.padL - (p.tapstep < 0 ? p.taps - 1 : 0);
    const int t_kmax = -p.padL + (p.tapstep > 0 ? p.taps - 1 : 0);
    auto item_range = [&](int item, int& gb, int& ge) __attribute__((always_inline)) {
        const int split = item / ntiles;
        gb = split * g_per;
        ge = min(GS, gb + g_per);
    };
    auto setup_item = [&](int item) __attribute__((always_inline)) {
        const int split = item / ntiles;
        const int tile = xcd_swizzle(item - split * ntiles, ntiles);
        // (TM, rows = g: consecutive tiles are the COLUMN tiles of one row tile -- the workgroups of an XCD then stream the
        //  same rows of the packed operand, which is far larger than the L2, at the same time)
        const int nt = (TM && pl.tmode == 1) ? tile % pl.n_col_tiles : tile / pl.n_row_tiles;
        const int n0 = nt * BN;
        if constexpr (TM) {
            int ge;
            item_range(item, g_begin, ge);
            nst = ge - g_begin;
            t_n0 = n0;
            t_mt = pl.tmode == 1 ? tile / pl.n_col_tiles : tile - nt * pl.n_row_tiles;
            if (pl.t_vec) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = n0 + 32 * (wave - 4) + 16 * h + (lane & 15);
                    v_ok[h] = j < p.K;
                    v_voff[h] = (unsigned)((v_ok[h] ? j : 0) * p.Tin);
                    v_al[h] = p.in_alpha ? p.in_alpha[v_ok[h] ? j : 0] : 1.f;
                    v_sum[h] = 0.f;
                }
                live = 3u;
                full = 0u;
                return;
            }
            if constexpr (ZP) {
                // column j = (ci, kk): plane row ci * stride + b shifted by d, where kk * tapstep - padL = d * stride + b;
                // column K (bias gradient) = the all-ones row.  Columns past the end alias the last real one: their products
                // land in accumulator columns the epilogue never stores.
                pos_valid = 0u;
#pragma unroll
                for (int par = 0; par < NPAR; ++par) {
                    const int j = n0 + 64 * (whalf ^ par) + lane;
                    const bool ones = p.bias != nullptr && j == p.K;
                    const bool valid = j < p.K || ones;
                    const int jj = min(j, p.K - 1);
                    const int ci = (int)div_magic((unsigned)jj, pl.ncols_magic);
                    const int kk = zp_tap_of(jj - ci * p.taps, pl);
                    const int ob = kk * p.tapstep - p.padL - pl.t_dmin * p.stride;   // >= 0: dmin = floor(min offset / stride)
                    const int db = (int)div_magic((unsigned)ob, pl.ps_magic);          // d - dmin
                    const int b = ob - db * p.stride;
                    const int row = ones ? pl.zp_rows - 1 : ci * p.stride + b;
                    const int sh = ones ? 0 : db;
                    // (each plane holds its rows twice, the second copy one element later: an odd shift reads that copy one
                    //  element earlier, so every chunk address is a multiple of 4 bytes -- tools/experiments/window_load_probe:
                    //  a wave's 16-byte loads at 2-byte-aligned addresses take 256 ticks each, at 4-byte-aligned ones 64;
                    //  in the kernel: 14 % fewer cycles per item)
                    // (round 6, with the planes staged by LDS DMA: odd shifts read from the FIRST copy at 2-byte-aligned addresses are
                    //  correct and cost 5 ... 8 % per launch, +0.3 ms per step -- what writing the second copy costs: kept)
                    const int odd = sh & 1;
                    zp_col[par] = (unsigned)odd * (unsigned)(pl.t_plane / 2) + (unsigned)(row * p.S) * (unsigned)pl.t_lseg +
                                  (unsigned)(sh - odd + fkL * 8);
                    if (valid) pos_valid |= 1u << par;
                }
                live = 0u;
                full = 0u;
#pragma unroll
                for (int b = 0; b < NPAR; ++b)
                    if (!pase_wave_all(!((pos_valid >> b) & 1u))) live |= 1u << b;
                live = (unsigned)pase_uniform((int)live);
                return;
            }
            pos_valid = 0u;
            t_ones = 0u;
#pragma unroll
            for (int par = 0; par < NPAR; ++par) {
                const int j = n0 + 64 * (whalf ^ par) + lane;              // column (NPS == 1)
                const bool ones = pl.tmode == 1 && p.bias != nullptr && j == p.K;
                const bool valid = j < p.K || ones;
                const int ci = valid && !ones ? (int)div_magic((unsigned)j, pl.ncols_magic) : 0;
                const int kk = valid && !ones ? j - ci * p.taps : 0;
                t_koff[par][0] = kk * p.tapstep - p.padL;
                pos_voff[par][0] = (unsigned)(ci * p.Tin + (t_koff[par][0] - t_kmin));
                pos_u0[par][0] = ci * p.Tin;
                t_sc[par][0] = p.in_scale ? p.in_scale[ci] : 1.f;
                t_sh[par][0] = p.in_scale ? p.in_shift[ci] : 0.f;
                t_al[par][0] = p.in_alpha ? p.in_alpha[ci] : 1.f;
                t_colsum[par][0] = 0.f;
                if (valid) pos_valid |= 1u << par;
                if (ones) t_ones |= 1u << par;
            }
            live = 0u;
            full = 0u;
            t_any_ones = !pase_wave_all(t_ones == 0u);
#pragma unroll
            for (int b = 0; b < NPAR; ++b) {
                if (!pase_wave_all(!((pos_valid >> b) & 1u))) live |= 1u << b;
                if (pase_wave_all(((pos_valid & ~t_ones) >> b) & 1u)) full |= 1u << b;
            }
            live = (unsigned)pase_uniform((int)live);
            full = (unsigned)pase_uniform((int)full);
            return;
        }
        const int s0 = (int)div_magic((unsigned)n0, pl.ncols_magic);
        const int qA = n0 - s0 * p.Ncols;
        const int lenA = min(BN, p.Ncols - qA);
        const int ncols_valid = min(BN, ntot - n0);
        const int nseg = (int)div_magic((unsigned)(n0 + ncols_valid - 1), pl.ncols_magic) - s0 + 1;
        const int span_len = ncols_valid + nseg * H;
        int ge;
        item_range(item, g_begin, ge);
        nst = ge - g_begin;
        pos_valid = 0u;
        pos_inter = 0u;
#pragma unroll
    for (int par = 0; par < NPAR; ++par)
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int i = 128 * ps + 64 * (whalf ^ par) + lane;
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
            const bool inter = !valid || (u0 >= 0 && u0 + pl.P - 1 < p.Tin);
            pos_u0[par][ps] = u0;
            pos_xoff[par][ps] = (unsigned)(valid ? s * pl.xp_tpad + q : 0);
            pos_sbase[par][ps] = (unsigned)(valid ? s * p.x_ctot * p.Tin : 0);
            pos_voff[par][ps] = pos_sbase[par][ps] + (unsigned)(inter ? u0 : 0);
            if (valid) pos_valid |= 1u << (par * NPS + ps);
            if (inter) pos_inter |= 1u << (par * NPS + ps);
        }
        // wave-uniform, per slot: at least one real position / only real positions / only in-range samples
        live = 0u;
        full = 0u;
        inter_slots = 0u;
#pragma unroll
        for (int b = 0; b < NPAR * NPS; ++b) {
            if (!pase_wave_all(!((pos_valid >> b) & 1u))) live |= 1u << b;
            if (pase_wave_all((pos_valid >> b) & 1u)) full |= 1u << b;
            if (pase_wave_all((pos_inter >> b) & 1u)) inter_slots |= 1u << b;
        }
        live = (unsigned)pase_uniform((int)live);
        full = (unsigned)pase_uniform((int)full);
        inter_slots = (unsigned)pase_uniform((int)inter_slots);
    };

    const bool has_aff = p.in_scale != nullptr, has_alpha = p.in_alpha != nullptr;       // uniform
    const float* xbase = p.x + (size_t)p.x_coff * p.Tin;

    x6c_f4 xreg[XR][ZP ? 1 : NSLOT][2];
    unsigned xmask[XR][ZP ? 1 : NSLOT];         // bit e: element e of the slot is a real sample (else: zero AFTER the transform)
    u32x4 xpl[XR][ZP ? NSLOT : 1][3];           // ZP: the slot's three plane chunks as loaded
    const unsigned short* zpb = reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(p.wx6) + pl.zp_off);
    const u32x4* xpc = reinterpret_cast<const u32x4*>(p.xp6);

    // channel' -> (input channel, phase) of element e of this wave's octet in k-group kg of stage g (all uniform)
    auto chan_of = [&](int g, int kg, int e, int& ci, int& b, bool& ok) __attribute__((always_inline)) {
        const int cp = (g * KGS + kg) * 16 + fkL * 8 + e;
        ok = cp < pl.CinP;
        const int cq = (int)div_magic((unsigned)cp, pl.p_magic);
        b = cp - cq * pl.P;
        ci = min(cq, p.Cin - 1);
    };
    auto slot_live = [&](int kg, int ps) __attribute__((always_inline)) {
        return kg < KGS && ((live >> ((kg & (NPAR - 1)) * NPS + ps)) & 1u) != 0;
    };
    // TM: k-group (g, kg) -> sequence s, first position of this wave's octet; is every sample of the octet, for every tap,
    // an in-range sample of a real sequence (uniform)
    auto t_geom = [&](int g, int kg, int& sq, int& qb0, bool& inter) __attribute__((always_inline)) {
        const int kgi = g * KGS + kg;
        const int s_ = (int)div_magic((unsigned)kgi, pl.seg_magic);         // kgi / QP16
        qb0 = (kgi - s_ * pl.P) * 16 + fkL * 8;
        inter = s_ < p.S && qb0 + 7 < p.Ncols && qb0 * p.stride + t_kmin >= 0 && (qb0 + 7) * p.stride + t_kmax < p.Tin;
        sq = min(s_, p.S - 1);
        if (s_ >= p.S) qb0 = p.Ncols;                                       // past the last sequence: no real position
    };
    auto load_slot = [&](auto r_tag, auto sl_tag, int g) __attribute__((always_inline)) {
        constexpr int sl = decltype(sl_tag)::value, rs = decltype(r_tag)::value;
        constexpr int kg = sl / NPS, ps = sl % NPS, par = kg & (NPAR - 1);
        if constexpr (ZP && TM) {
            // k-group (uniform): sequence s, first position 16 q16; past the last sequence the pack of g holds zeros and any
            // finite value will do here
            const int kgi = g * KGS + kg;
            const int s_ = (int)div_magic((unsigned)kgi, pl.seg_magic);
            const int q16 = kgi - s_ * pl.P;
            const unsigned off = zp_col[par] + (unsigned)(min(s_, p.S - 1) * pl.t_lseg + q16 * 16);
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) xpl[rs][sl][pz] = x6c_load16u(zpb + (size_t)pz * (size_t)pl.t_plane + off);
            return;
        } else
        if constexpr (TM) {
            if (pl.t_vec) {
                constexpr int h = (sl >> 1) & 1;
                const int c = (sl & 1) * 4 + (lane >> 4);
                const int kgi = g * KGS + (c >> 1);
                const int s_ = (int)div_magic((unsigned)kgi, pl.seg_magic);
                const int qv = (kgi - s_ * pl.P) * 16 + (c & 1) * 8;
                const bool ok = v_ok[h] && s_ < p.S && qv < p.Ncols;            // Ncols % 8 == 0: whole chunks only
                unsigned off = ok ? (unsigned)(s_ * p.x_ctot * p.Tin + qv) + v_voff[h] : 0u;
#ifdef PASE_X6C_TRACE
                if (pl.prio & 64) off = (unsigned)(lane & 15) * 8u;      // ablation: every load hits the same two cache lines
#endif
                x6c_gload_x8<rs>(xreg[rs][sl], xbase, off * 4u);
                xmask[rs][sl] = ok ? 0xffu : 0u;
                return;
            }
            int sq, qb0;
            bool inter;
            t_geom(g, kg, sq, qb0, inter);
            const float* sb = xbase + (size_t)sq * p.x_ctot * p.Tin;
            const unsigned vbit = ((pos_valid & ~t_ones) >> par) & 1u;
            if (inter) {
                pase_static_for<8>([&](auto et) __attribute__((always_inline)) {
                    constexpr int e = decltype(et)::value;
                    x6c_gload<e, rs>(xreg[rs][sl], sb + ((qb0 + e) * p.stride + t_kmin), pos_voff[par][ps] * 4u);
                });
                xmask[rs][sl] = (0u - vbit) & 0xffu;
            } else {
                unsigned mask = 0u;
                pase_static_for<8>([&](auto et) __attribute__((always_inline)) {
                    constexpr int e = decltype(et)::value;
                    int u = (qb0 + e) * p.stride + t_koff[par][ps];
                    if (p.pad_mode == PASE_PAD_REFLECT) u = x6c_reflect(u, p.Tin);
                    const unsigned okb = (qb0 + e < p.Ncols ? vbit : 0u) & x6c_in_range(u, p.Tin);
                    x6c_gload<e, rs>(xreg[rs][sl], sb, ((unsigned)(pos_u0[par][ps] + u) & (0u - okb)) * 4u);
                    mask |= okb << e;
                });
                xmask[rs][sl] = mask;
            }
            return;
        }
        if constexpr (ZP && !TM) {
            // XP: three aligned 16-byte chunks; k-groups past the last real one (stage padding) carry zero weights: any
            // finite data will do -> clamp (uniform)
            const int gidx = min(g * KGS + kg, pl.G - 1);
            const unsigned off = (unsigned)((gidx * 2 + fkL) * p.S) * (unsigned)pl.xp_tpad + pos_xoff[par][ps];
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) xpl[rs][sl][pz] = xpc[(size_t)pz * (size_t)pl.xp_plane + off];
            return;
        } else {
        const bool all_inter = ((inter_slots >> (par * NPS + ps)) & 1u) != 0;      // uniform
        const unsigned vbit = (pos_valid >> (par * NPS + ps)) & 1u;
        const int c0l = (g * KGS + kg) * 16 + fkL * 8;                             // first channel' of the octet (uniform)
        if (all_inter && pl.P == 1 && c0l + 8 <= p.Cin) {
            // interior, stride 1, eight real channels: ONE 64-bit base per slot and a scalar increment per element (the
            // general form below spends a magic division, a clamp and a 64-bit multiply-add per element on the scalar
            // unit -- the staging waves of the K = 21 525 data gradients were 2/3 address arithmetic)
            const float* bp = xbase + (size_t)c0l * p.Tin;
            pase_static_for<8>([&](auto et) __attribute__((always_inline)) {
                constexpr int e = decltype(et)::value;
                x6c_gload<e, rs>(xreg[rs][sl], bp, pos_voff[par][ps] * 4u);
                bp += p.Tin;
            });
            xmask[rs][sl] = (0u - vbit) & 0xffu;
        } else
        if (all_inter && c0l + 8 <= pl.CinP) {
            // interior, eight real channels' of a strided launch: (channel, phase) of element 0 by one division, then the
            // base pointer walks -- + 1 sample to the next phase, + Tin - (P - 1) to the next channel's phase 0
            const int cq0 = (int)div_magic((unsigned)c0l, pl.p_magic);
            int bph = c0l - cq0 * pl.P;
            const float* bp = xbase + (size_t)cq0 * p.Tin + bph;
            const int to_next = p.Tin - (pl.P - 1);
            if (pl.pairs) {
                // even stride: the octet starts at an even phase and elements (e, e + 1) are two consecutive samples of one
                // channel -- four 8-byte loads per slot instead of eight 4-byte ones (lanes are P samples apart: every load
                // instruction of a strided launch touches the same ~P / 32 x 64 cache lines whatever its width)
                pase_static_for<4>([&](auto et) __attribute__((always_inline)) {
                    constexpr int e = 2 * decltype(et)::value;
                    x6c_gload2<e, rs>(xreg[rs][sl], bp, pos_voff[par][ps] * 4u);
                    bph += 2;
                    const bool wrap = bph == pl.P;                                // uniform
                    bp += wrap ? to_next + 1 : 2;
                    bph = wrap ? 0 : bph;
                });
            } else {
            pase_static_for<8>([&](auto et) __attribute__((always_inline)) {
                constexpr int e = decltype(et)::value;
                x6c_gload<e, rs>(xreg[rs][sl], bp, pos_voff[par][ps] * 4u);
                const bool wrap = ++bph == pl.P;                                  // uniform
                bp += wrap ? to_next : 1;
                bph = wrap ? 0 : bph;
            });
            }
            xmask[rs][sl] = (0u - vbit) & 0xffu;
        } else
        if (all_inter) {
            // interior: one load per element off a uniform base, no per-element address arithmetic
            pase_static_for<8>([&](auto et) __attribute__((always_inline)) {
                constexpr int e = decltype(et)::value;
                int ci, b;
                bool chok;
                chan_of(g, kg, e, ci, b, chok);
                x6c_gload<e, rs>(xreg[rs][sl], xbase + (size_t)ci * p.Tin + b, pos_voff[par][ps] * 4u);
            });
            xmask[rs][sl] = (0u - vbit) & 0xffu;
        } else {
            // padding arithmetic in integer lanes (no per-element lane-mask SGPR pairs: eight of them per slot, three
            // register sets deep, is what spilled): okb = 1 for a real sample, the offset is ANDed with 0 - okb
            unsigned mask = 0u;
            pase_static_for<8>([&](auto et) __attribute__((always_inline)) {
                constexpr int e = decltype(et)::value;
                int ci, b;
                bool chok;
                chan_of(g, kg, e, ci, b, chok);
                int u = pos_u0[par][ps] + b;
                if (p.pad_mode == PASE_PAD_REFLECT) u = x6c_reflect(u, p.Tin);
                const unsigned okb = vbit & x6c_in_range(u, p.Tin);
                x6c_gload<e, rs>(xreg[rs][sl], xbase + (size_t)ci * p.Tin, ((pos_sbase[par][ps] + (unsigned)u) & (0u - okb)) * 4u);
                mask |= okb << e;
            });
            xmask[rs][sl] = mask;
        }
        }
    };
    // registers of slot sl (stage g) -> on-load transform -> three bf16 planes -> LDS buffer bsel
    auto store_slot = [&](auto r_tag, auto sl_tag, int g, int bsel) __attribute__((always_inline)) {
        constexpr int sl = decltype(sl_tag)::value, rs = decltype(r_tag)::value;
        constexpr int kg = sl / NPS, ps = sl % NPS, par = kg & (NPAR - 1);
        if constexpr (ZP) {
            const int i = 128 * ps + 64 * (whalf ^ par) + lane;
            if (NPS * 128 == NPOS || i < NPOS) {
                u32x4* dst = &Xs[bsel * BUF + kg * KGC + fkL * NPOS + i];
#pragma unroll
                for (int pz = 0; pz < 3; ++pz) dst[pz * PLANE] = xpl[rs][sl][pz];
            }
            return;
        } else {
        x6c_claim<rs>(xreg[rs][sl]);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = xreg[rs][sl][e >> 2][e & 3];
        if constexpr (TM) {
            if (pl.t_vec) {
                constexpr int h = (sl >> 1) & 1;
                const int c = (sl & 1) * 4 + (lane >> 4);
                if (has_alpha) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * v_al[h];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x6c_keep(v[e], xmask[rs][sl], e);
                if (p.bias != nullptr) {                                      // uniform
#pragma unroll
                    for (int e = 0; e < 8; ++e) v_sum[h] += v[e];
                }
                u32x4 o[3];
                pase_split_bf16x3_rne(v, o);
                u32x4* dst = &Xs[bsel * BUF + (c >> 1) * KGC + (c & 1) * NPOS + 32 * (wave - 4) + 16 * h + (lane & 15)];
#pragma unroll
                for (int pz = 0; pz < 3; ++pz) dst[pz * PLANE] = o[pz];
                return;
            }
            int sq, qb0;
            bool inter;
            t_geom(g, kg, sq, qb0, inter);
            if (has_aff) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], t_sc[par][ps], t_sh[par][ps]);
            }
            if (has_alpha) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * t_al[par][ps];
            }
            const bool slot_full = ((full >> par) & 1u) != 0;                 // uniform: 64 real, non-"ones" columns
            if (!(inter && slot_full)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x6c_keep(v[e], xmask[rs][sl], e);
                if (t_any_ones) {      // uniform: this item's tile holds the bias column
                    const bool ones = (t_ones >> par) & 1u;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (ones) v[e] = (qb0 + e < p.Ncols) ? 1.f : 0.f;
                }
            }
            if (pl.tmode >= 2 && p.bias != nullptr) {                         // uniform
#pragma unroll
                for (int e = 0; e < 8; ++e) t_colsum[par][ps] += v[e];
            }
            u32x4 o[3];
            pase_split_bf16x3_rne(v, o);
            u32x4* dst = &Xs[bsel * BUF + kg * KGC + fkL * NPOS + 64 * (whalf ^ par) + lane];
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) dst[pz * PLANE] = o[pz];
            return;
        }
        const int c0 = (g * KGS + kg) * 16 + fkL * 8;                    // first channel' of the octet (uniform)
        const bool chan_full = c0 + 8 <= pl.CinP;                         // uniform
        // on-load parameters per channel' (expanded by pase_pack_x6 behind the weight chunks, padded to whole stages):
        // three scalar loads per octet, values stay in SGPRs
        if (has_aff) {      // uniform
            float sc[8], sh[8];
            sload8x2(prm + c0, prm + prm_n + c0, sc, sh);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
        }
        if (has_alpha) {
            float al[8];
            sload8(prm + 2 * prm_n + c0, al);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * al[e];
        }
        // zero padding applies AFTER the transform (skipped when every lane of the wave holds real samples)
        const bool all_inter = ((inter_slots >> (par * NPS + ps)) & 1u) != 0;     // uniform (as in load_slot)
        const bool pos_full = ((full >> (par * NPS + ps)) & 1u) != 0;              // uniform
        if (!(all_inter && pos_full && chan_full)) {
            const int nch = pl.CinP - c0;                                                 // uniform
            const unsigned m = xmask[rs][sl] & (nch >= 8 ? 0xffu : nch > 0 ? (1u << nch) - 1u : 0u);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = x6c_keep(v[e], m, e);
        }
        u32x4 o[3];
        pase_split_bf16x3_rne(v, o);
        const int i = 128 * ps + 64 * (whalf ^ par) + lane;
        if (NPS * 128 == NPOS || i < NPOS) {
            u32x4* dst = &Xs[bsel * BUF + kg * KGC + fkL * NPOS + i];
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) dst[pz * PLANE] = o[pz];
        }
        }
    };

        // ---- staging.  Register set r holds stage g_begin + r, + XR, ...: a stage's loads are issued two stages
        // before its conversion (HBM latency under load exceeds the duration of a short stage).  Stage g + 1 is converted
        // into the other LDS buffer while the compute waves multiply stage g.  The NEXT item's first stage is staged
        // before the barrier inside the compute waves' epilogue of the current one (its buffer is free by then).
        auto load_stage = [&](auto r_tag, int g) __attribute__((always_inline)) {
            pase_static_for<NSLOT>([&](auto sl) __attribute__((always_inline)) {
                if (slot_live(decltype(sl)::value / NPS, decltype(sl)::value % NPS)) load_slot(r_tag, sl, g);
            });
        };
        auto store_stage = [&](auto r_tag, int g, int bs) __attribute__((always_inline)) {
            pase_static_for<NSLOT>([&](auto sl) __attribute__((always_inline)) {
                if (slot_live(decltype(sl)::value / NPS, decltype(sl)::value % NPS)) store_slot(r_tag, sl, g, bs);
            });
        };
        int nlive = 0;                 // live slots of this wave for the current item: 8 * nlive loads per stage
        const int per_slot = (TM && pl.t_vec) ? 2 : ((!TM && pl.pairs) ? 4 : 8);      // fewest loads of a live slot (uniform)
        auto prologue = [&](int item) __attribute__((always_inline)) {
            if (wave == 4) X6C_STAMP(4);
            setup_item(item);
#if !defined(PASE_HIPEMU)
            // (weight gradients: setup_item loads per-column PReLU slopes with ordinary loads.  Left pending in the compiler's
            //  bookkeeping until their first use inside the stage loop, they made it put s_waitcnt vmcnt(0) in front of every
            //  hidden load group of the row-coalesced path -- a synchronous pipeline.  Nothing hidden is in flight here.)
            if constexpr (TM) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
#endif
            nlive = 0;
            pase_static_for<NSLOT>([&](auto sl) __attribute__((always_inline)) {
                if (slot_live(decltype(sl)::value / NPS, decltype(sl)::value % NPS)) ++nlive;
            });
            load_stage(std::integral_constant<int, 0>{}, g_begin);
            if (1 < nst) load_stage(std::integral_constant<int, 1>{}, g_begin + 1);
            x6c_vmwait_slots(1 < nst ? nlive : 0, per_slot);         // stage 0 has landed
            store_stage(std::integral_constant<int, 0>{}, g_begin, bsel);
            if (2 < nst) load_stage(std::integral_constant<int, 2>{}, g_begin + 2);
            if (wave == 4) X6C_STAMP(5);
        };
        auto next_item = [&](int item) __attribute__((always_inline)) {
            for (item += gridDim.x; item < nitems; item += gridDim.x) {
                int gb, ge;
                item_range(item, gb, ge);
                if (gb < ge) break;
            }
            return item;
        };

        const bool spectrum = p.post_op == PASE_POST_POW || p.post_op == PASE_POST_LOGPOW || p.post_op == PASE_POST_MAG;
        const bool epi_barrier = TM ? false : (p.epilogue == PASE_EPI_STORE ? (!spectrum && p.stat_part != nullptr) : true);
        if constexpr (ZP) {
            // Pre-split operands (weight gradients on phase planes, convolutions on channel-minor planes): a stage is a COPY,
            // and the stage buffers' chunk order is lane-linear -- global_load_lds_dwordx4 does it without registers, vector
            // ALU work or LDS write instructions.  (Through registers the compiler's s_waitcnt bookkeeping had degraded to
            // vmcnt(0) in front of every load group and every write: the per-slot branches and the address registers it
            // allocated on top of the load destinations -- the staging waves ran one memory latency per slot, 90 % busy, and
            // the weight gradients at 1290 clocks per step against 790 of MFMA work.)
            auto direct_stage = [&](int g, int bs, int fk) __attribute__((always_inline)) {      // fk: octet of the k-groups (0 / 1)
                pase_static_for<NSLOT>([&](auto sl_tag) __attribute__((always_inline)) {
                    constexpr int sl = decltype(sl_tag)::value;
                    constexpr int kg = sl / NPS, ps = sl % NPS, par = kg & (NPAR - 1);
                    const int i0 = 128 * ps + 64 * (whalf ^ par);                  // uniform: first position of this wave
                    if (slot_live(kg, ps) && (NPS * 128 == NPOS || i0 < NPOS)) {   // uniform
                        u32x4* dst = &Xs[bs * BUF + kg * KGC + fk * NPOS + i0];
                        if constexpr (TM) {
                            const int kgi = g * KGS + kg;
                            const int s_ = (int)div_magic((unsigned)kgi, pl.seg_magic);
                            const int q16 = kgi - s_ * pl.P;
                            const unsigned off = zp_col[par] + (unsigned)(min(s_, p.S - 1) * pl.t_lseg + q16 * 16);
#pragma unroll
                            for (int pz = 0; pz < 3; ++pz)
                                x6c_load_lds16(zpb + (size_t)pz * (size_t)pl.t_plane + off, dst + pz * PLANE, lane);
                        } else {
                            const int gidx = min(g * KGS + kg, pl.G - 1);
                            const unsigned off = (unsigned)((gidx * 2 + fk) * p.S) * (unsigned)pl.xp_tpad + pos_xoff[par][ps];
#pragma unroll
                            for (int pz = 0; pz < 3; ++pz)
                                x6c_load_lds16(xpc + (size_t)pz * (size_t)pl.xp_plane + off, dst + pz * PLANE, lane);
                        }
                    }
                });
            };

            auto prologue_dl = [&](int item) __attribute__((always_inline)) {
                if (wave == 4) X6C_STAMP(4);
                setup_item(item);
#if !defined(PASE_HIPEMU)
                if constexpr (TM) __builtin_amdgcn_s_waitcnt(0x0F70);      // (see prologue)
#endif
                direct_stage(g_begin, bsel, fkL);
                if (wave == 4) X6C_STAMP(5);
            };
            int item = next_item((int)blockIdx.x - (int)gridDim.x);
            if (item < nitems) prologue_dl(item);
            while (item < nitems) {
                x6c_vm_drain();
                __syncthreads();           // the item's first stage is visible to the compute waves
                for (int gi = 0; gi < nst; ++gi) {
                    X6C_T0();
                    if (gi + 1 < nst) direct_stage(g_begin + gi + 1, bsel ^ 1, fkL);
                    x6c_vm_drain();
                    if (wave == 4) X6C_TACC(8);
                    __syncthreads();
                    bsel ^= 1;
                }
                if (wave == 4) X6C_STAMP(6);
                X6C_TRACE_NEXT();
                item = next_item(item);
                if (item < nitems) prologue_dl(item);
                if (epi_barrier) {
                    x6c_vm_drain();
                    __syncthreads();       // the barrier of the compute waves' epilogue
                }
            }
            return;
        }
        int item = next_item((int)blockIdx.x - (int)gridDim.x);
        if (item < nitems) prologue(item);
        while (item < nitems) {
            __syncthreads();               // the item's first stage is visible to the compute waves
            for (int gb = 0; gb < nst; gb += XR) {
                pase_static_for<XR>([&](auto r) __attribute__((always_inline)) {
                    constexpr int rn = (decltype(r)::value + 1) % XR;          // register set of stage gi + 1
                    const int gi = gb + decltype(r)::value;                     // stage (relative) being multiplied
                    if (gi < nst) {
                        X6C_T0();
#ifdef PASE_X6C_TRACE
                        if (!(pl.prio & 128))      // ablation: the staging waves only keep the barriers (results are garbage)
#endif
                        if (gi + 1 < nst) {
                            // in flight: stage gi + 1 (set rn, the older one) and stage gi + 2
                            {
                                X6C_T0();
                                x6c_vmwait_slots(gi + 2 < nst ? nlive : 0, per_slot);
                                if (wave == 4) X6C_TACC(10);
                            }
                            {
                                X6C_T0();
                                store_stage(std::integral_constant<int, rn>{}, g_begin + gi + 1, bsel ^ 1);
                                if (wave == 4) X6C_TACC(11);
                            }
                        }
                        // set r (stage gi, converted one window ago) is free: stage gi + 3
#ifdef PASE_X6C_TRACE
                        if (!(pl.prio & 128))
#endif
                        if (gi + XR < nst) load_stage(r, g_begin + gi + XR);
                        if (wave == 4) X6C_TACC(8);
                        __syncthreads();
                        bsel ^= 1;
                    }
                });
            }
            if (wave == 4) X6C_STAMP(6);
            X6C_TRACE_NEXT();
            if constexpr (TM) {
                // swapped weight gradient (columns = rows of g): the bias gradient is the column sum of everything staged
                // (each octet half adds its eight positions per k-group); once per column tile (row tile 0 only)
                if (pl.t_vec && p.bias != nullptr && t_mt == 0) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (v_ok[h])
                            atomicAdd(const_cast<float*>(p.bias) + t_n0 + 32 * (wave - 4) + 16 * h + (lane & 15), v_sum[h]);
                } else if (pl.tmode >= 2 && p.bias != nullptr && t_mt == 0) {
#pragma unroll
                    for (int par = 0; par < NPAR; ++par)
                        if ((pos_valid >> par) & 1u) {
                            const int j = t_n0 + 64 * (whalf ^ par) + lane;
                            atomicAdd(const_cast<float*>(p.bias) + j, t_colsum[par][0]);
                        }
                }
            }
            item = next_item(item);
            if (item < nitems) prologue(item);
            // the barrier of the compute waves' epilogue (partial BatchNorm sums / loss partials go through LDS)
            if (epi_barrier) __syncthreads();
        }
        return;
    }

  f32x16 accH[4], accS[4];
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    // ---- tile decode -----------------------------------------------------------------------
    const int split = item / ntiles;
    const int tile = xcd_swizzle(item - split * ntiles, ntiles);
    const bool col_major = TM && pl.tmode == 1;
    const int mt = col_major ? tile / pl.n_col_tiles : tile % pl.n_row_tiles;
    const int nt = col_major ? tile % pl.n_col_tiles : tile / pl.n_row_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int s0 = (int)div_magic((unsigned)n0, pl.ncols_magic);
    const int qA = n0 - s0 * p.Ncols;
    const int lenA = min(BN, p.Ncols - qA);
    const int ncols_valid = min(BN, ntot - n0);
    const int nseg = (int)div_magic((unsigned)(n0 + ncols_valid - 1), pl.ncols_magic) - s0 + 1;
    const int span_len = ncols_valid + nseg * H;
    const int g_begin = split * g_per;
    const int g_end = min(GS, g_begin + g_per);
    if (g_begin >= g_end) continue;                // uniform for the whole block: no barrier is skipped one-sidedly
    const int nst = g_end - g_begin;

    // ================= compute waves =================
    if (wave == 0) X6C_STAMP(0);
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

    // The bias is the accumulators' initial value (rows = output channels, no pixel shuffle: index = row): its loads are
    // issued here, a barrier and two weight-fragment loads before they are needed.  In the epilogue the same 16 values cost
    // 2.9 k clocks per tile -- a load there waits behind whatever the memory pipeline holds at that point, and the compute
    // wave is alone on its SIMD: nothing hides it (tools/trace_x6c.py, `qrnn` against `qrnn_nb`).
    static_assert(NBT == 4, "accumulators are declared for four B tiles");
    {
        float binit[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) binit[r] = 0.f;
        if (!TM && bias_init && split == 0) {      // uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 + 4 * fk + (r & 3) + 8 * (r >> 2);
                binit[r] = p.bias[min(m, p.M - 1)];
            }
        }
#pragma unroll
        for (int j = 0; j < NBT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accH[j][r] = binit[r];
                accS[j][r] = 0.f;
            }
    }

    // ---- A fragments, TWO steps ahead: three uniform plane pointers + one per-lane byte offset ----------------------
    //   packs (convolutions, tmode 1 / 2): [32-row tile][step][plane][lane] 16-byte chunks, 3072 bytes per step
    //   tmode 3: row (ci, kk) of the weight gradient = phase b / shift d of channel ci in row-major bf16 planes of z~
    //     (pack_zplanes_kernel: [plane][ci * stride + b][s][t_lseg]); the k-group (s, q0) of that row starts at element
    //     s * t_lseg + q0 + (d - dmin): consecutive k-groups are 16 elements apart, t_hh more across a sequence boundary
    // (scalar byte offsets from the scratch's start, p.wx6: the plans keep it below 4 GiB)
    const X6cRsrc a_rs = x6c_make_rsrc(p.wx6);
    unsigned ab[3];
    unsigned a_loff;
    int a_adv, a_q16 = 0, a_wrap = 0x7fffffff, a_hh2 = 0;
    if (TM && pl.tmode == 3) {
        const int j = min(m0 + wm * 32 + fr, p.M - 1);
        const int ci = (int)div_magic((unsigned)j, pl.p_magic);
        const int kk = j - ci * pl.t_taps;
        const int o = kk * pl.t_tapstep - pl.t_padL;
        const int d = o >= 0 ? o / pl.t_stride : -((-o + pl.t_stride - 1) / pl.t_stride);
        const int b = o - d * pl.t_stride;
        const int kgi0 = g_begin * KGS;
        const int sq = (int)div_magic((unsigned)kgi0, pl.seg_magic);
        a_q16 = kgi0 - sq * pl.P;
        a_wrap = pl.P;
        a_hh2 = 2 * pl.t_hh;
        a_adv = 32;
        // (each plane holds the rows twice, the second copy one element later: a row whose shift d - dmin is odd reads that
        //  copy, so every fragment address is a multiple of 4 bytes -- 2-byte-aligned 16-byte loads run at half rate)
        const int sh = d - pl.t_dmin, odd = sh & 1;
        a_loff = 2u * (unsigned)(odd * (int)(pl.t_plane / 2) + ((ci * pl.t_stride + b) * p.S) * pl.t_lseg + sh - odd + fk * 8);
#pragma unroll
        for (int pz = 0; pz < 3; ++pz)
            ab[pz] = (unsigned)(2 * ((size_t)pz * pl.t_plane + (size_t)sq * pl.t_lseg + (size_t)a_q16 * 16));
    } else {
        a_adv = 3072;
        a_loff = 16u * (unsigned)lane;
#pragma unroll
        for (int pz = 0; pz < 3; ++pz)
            ab[pz] = (unsigned)(16 * (((size_t)(mt * WM + wm) * (unsigned)pl.steps_total + (size_t)g_begin * (unsigned)nsteps) * 192u +
                                      64u * pz));
    }
    const int nsteps_run = nst * nsteps;
    int a_issued = 0;
    auto load_a = [&](u32x4 (&a)[3]) __attribute__((always_inline)) {
        // unconditional (the steps past the end re-read the last fragments): a load behind a branch makes the compiler's
        // vmcnt bookkeeping fall back to vmcnt(0), which would wait for the fragments issued a moment ago
#ifndef PASE_ABL_NOA      // (ablation builds of tools/trace_x6c.py: no A-fragment loads inside the loop)
        // Buffer loads: descriptor + the lane's constant byte offset + a scalar offset.  (As plain global loads the compiler
        // added base + lane offset into a VGPR pair per step, allocated on top of the destination registers; the
        // write-after-write wait it then needs against the load that last filled them came out as s_waitcnt vmcnt(0) at the
        // loop header: every fragment in flight had to land before the next one was requested -- prefetch distance 1, not 2.)
        if constexpr (!TM || ZP) {
            // packs: the three planes of a step are 1 KB apart -- ONE scalar offset
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) a[pz] = x6c_buffer_load16(a_rs, a_loff + 1024u * pz, ab[0]);
        } else {
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) a[pz] = x6c_buffer_load16(a_rs, a_loff, ab[pz]);
        }
#endif
        ++a_issued;
        int adv = (a_issued < nsteps_run) ? a_adv : 0;
        if constexpr (TM) {      // (mode 3 only: plane rows wrap at the sequence boundaries; the convolutions carry no such state)
            const bool wrap = a_q16 + 1 == a_wrap;
            a_q16 = wrap ? 0 : a_q16 + 1;
            adv = (a_issued < nsteps_run) ? a_adv + (wrap ? a_hh2 : 0) : 0;
        }
        if constexpr (!TM || ZP) ab[0] += adv;
        else {
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) ab[pz] += adv;
        }
    };
    // The step's 12 B fragments are SOFTWARE-PIPELINED over its two halves (tiles {0, 1} and {2, 3}): the six ds_read_b128 of
    // the next half -- the second half of this step, then the first half of the next step of the stage -- are issued one per
    // two MFMAs of the current half, so a fragment has ~10 MFMAs (320+ cycles) to arrive.  Round 3 read a half's six fragments
    // right in front of its 12 MFMAs: with ONE compute wave per SIMD nothing else covers the LDS latency, and a step took
    // 1050-1100 ticks against 768 of pure MFMA issue.  tools/experiments/mfma_dep_probe.hip isolates it (ticks per 24-MFMA
    // step, one wave per SIMD): bare MFMAs 788 in ANY accumulator order (dependent accumulators cost nothing); "6 reads, then
    // 12 MFMAs" twice = 1040 (the old loop, to the tick); the next half's reads in front of this half's MFMAs 833; one read
    // per two MFMAs 800.  (Round 3's hand schedules moved MFMAs and waits around but kept reads and their first use in the
    // same half; its "prefetch the next pair" variant lost to register copies.)  Price: 24 VGPRs.  The first half of a
    // stage's first step is read behind the stage barrier (the data is not there earlier): once per stage.
    u32x4 bq[2][2][3];          // [half][tile of the pair][plane]
    auto load_first = [&](const u32x4* xb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) bq[0][i][pz] = xb[pz * PLANE + bbase[i]];
    };
    // xn: the next step's fragments base in the same stage buffer (== xb when this is the stage's last step: a harmless re-read)
    auto mfma_step = [&](const u32x4 (&a)[3], const u32x4* xb, const u32x4* xn) __attribute__((always_inline)) {
        // plane pairs of the five small terms, smallest first: mm, hl, lh, hm, mh -> accS; hh -> accH
        constexpr int PZA[6] = {1, 0, 2, 0, 1, 0}, PZB[6] = {1, 2, 0, 1, 0, 0};
        // (the three A-fragment loads of load_a() first: left to itself the scheduler sinks them behind ~19 MFMAs -- their
        //  address temporaries reuse the retiring fragment's registers -- and the "two steps ahead" prefetch becomes 160 cycles)
        PASE_SGB(0x020, 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32x4* src = h == 0 ? xb : xn;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                // one fragment of the NEXT half ...
#ifndef PASE_ABL_NOB      // (ablation builds: no B-fragment reads inside the steps)
                bq[h ^ 1][i / 3][i % 3] = src[(i % 3) * PLANE + bbase[2 * (h ^ 1) + i / 3]];
#endif
                // ... per two products of this one
                if (i < 5) {
                    accS[2 * h] = pase_mfma_bf16_32x32x16(a[PZA[i]], bq[h][0][PZB[i]], accS[2 * h]);
                    accS[2 * h + 1] = pase_mfma_bf16_32x32x16(a[PZA[i]], bq[h][1][PZB[i]], accS[2 * h + 1]);
                } else {
                    accH[2 * h] = pase_mfma_bf16_32x32x16(a[0], bq[h][0][0], accH[2 * h]);
                    accH[2 * h + 1] = pase_mfma_bf16_32x32x16(a[0], bq[h][1][0], accH[2 * h + 1]);
                }
                PASE_SGB(0x100, 1);      // pin: one DS read, then two MFMAs
                PASE_SGB(0x008, 2);
            }
        }
    };

    // ---- main loop: stage = KGS k-groups x A taps; step st = kg * A + t -------------------------------------
    u32x4 a0[3], a1[3], a2[3];
#ifdef PASE_ABL_NOA
#pragma unroll
    for (int pz = 0; pz < 3; ++pz) {
        a0[pz] = a1[pz] = a2[pz] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        asm volatile("" : "+v"(a0[pz]), "+v"(a1[pz]), "+v"(a2[pz]));
    }
#endif
    // ---- SYM: this wave's share of a stage's copy.  Unit u = (kg, octet, 64-position block), u = wave, wave + 8
    constexpr int NP64 = NPOS / 64;
    int sy_kg[SYU], sy_dst[SYU];
    unsigned sy_src[SYU];
    bool sy_on[SYU];
#pragma unroll
    for (int j = 0; j < SYU; ++j) {
        sy_kg[j] = 0;
        sy_dst[j] = 0;
        sy_src[j] = 0u;
        sy_on[j] = false;
    }
    const u32x4* const sy_xpc = reinterpret_cast<const u32x4*>(p.xp6);
    if constexpr (SYM) {
#pragma unroll
        for (int j = 0; j < SYU; ++j) {
            const int u = wave + SYW * j;                                      // uniform
            sy_on[j] = u < KGS * 2 * NP64;
            const int kg_ = u / (2 * NP64), rem = u - kg_ * (2 * NP64);
            const int fk_ = rem / NP64, p64 = rem - fk_ * NP64;
            // position i of the stage row -> (sequence, padded position): as the staging waves' setup_item
            const int i = 64 * p64 + lane;
            bool valid = i < span_len;
            int k_, r_;
            if (i < lenA + H) {
                k_ = 0;
                r_ = i;
            } else {
                const int d_ = i - (lenA + H);
                const int k1 = (int)div_magic((unsigned)d_, pl.seg_magic);
                k_ = 1 + k1;
                r_ = d_ - k1 * segL;
            }
            const int q_ = (k_ == 0 ? qA : 0) + r_;
            const int s_ = s0 + k_;
            valid = valid && s_ < p.S;
            sy_kg[j] = kg_;
            sy_dst[j] = kg_ * KGC + fk_ * NPOS + 64 * p64;
            // chunk offset inside a plane, without the k-group part ((gidx * 2 + fk) * S * xp_tpad)
            sy_src[j] = (unsigned)(fk_ * p.S) * (unsigned)pl.xp_tpad + (unsigned)(valid ? s_ * pl.xp_tpad + q_ : 0);
        }
    }
    auto sym_dma = [&](int j, int g, int bs) __attribute__((always_inline)) {
        if constexpr (SYM) {
            if (sy_on[j]) {                                                    // uniform
                const int gidx = min(g * KGS + sy_kg[j], pl.G - 1);          // (stage padding: zero weights, any finite data)
                const unsigned off = (unsigned)(gidx * 2 * p.S) * (unsigned)pl.xp_tpad + sy_src[j];
                u32x4* dst = &Xs[bs * BUF + sy_dst[j]];
#pragma unroll
                for (int pz = 0; pz < 3; ++pz) x6c_dma16_hidden(sy_xpc + (size_t)pz * (size_t)pl.xp_plane + off, dst + pz * PLANE, lane);
            }
        }
    };
    if constexpr (SYM) {      // the item's first stage (every wave has passed the barrier that ended the previous item's last stage)
#pragma unroll
        for (int j = 0; j < SYU; ++j) sym_dma(j, g_begin, bsel);
    }
    load_a(a0);
    load_a(a1);
    if constexpr (SYM) x6c_vm_drain();
    __syncthreads();
    if (wave == 0) X6C_STAMP(1);
    // step bookkeeping in increments (no multiplies, four scalar counters): chunk offset of the step's fragments inside Xs,
    // taps left in the k-group, steps left in the stage, stages left in the item
    int xoff = bsel * BUF, taps_left = pl.A, steps_left = nsteps, stages_left = nst;
    bool done = false;
    load_first(&Xs[xoff]);
    auto step = [&](const u32x4 (&acur)[3], u32x4 (&anxt)[3]) __attribute__((always_inline)) {
        if constexpr (SYM) {
            // the next stage's copy, three DMA instructions at the top of this stage's first two steps (uniform branches
            // around hidden instructions: the compiler's vmcnt bookkeeping of the fragment loads is the same on both paths)
            if (stages_left > 1) {
                const int g_nxt = g_end - stages_left + 1;
                if (steps_left == nsteps) sym_dma(0, g_nxt, bsel ^ 1);
                else if (steps_left == nsteps - 1) sym_dma(1, g_nxt, bsel ^ 1);
                else if (SYU > 2 && steps_left == nsteps - 2) sym_dma(SYU - 1, g_nxt, bsel ^ 1);
            }
        }
        load_a(anxt);
        const u32x4* xb = &Xs[xoff];
        const bool kg_end = --taps_left == 0;                                 // uniform
        xoff += kg_end ? KGC - (pl.A - 1) : 1;
        taps_left = kg_end ? pl.A : taps_left;
        const bool stage_end = --steps_left == 0;                             // uniform
        mfma_step(acur, xb, stage_end ? xb : &Xs[xoff]);
        if (stage_end) {
            {
                X6C_T0();
#if !defined(PASE_HIPEMU)
                // SYM: the next stage's copy has landed -- it is older than the fragment loads of this stage's steps from the
                // second on, and the counter retires in order: at most the newest three loads may stay in flight
                if constexpr (SYM) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
#endif
                __syncthreads();
                if (wave == 0) X6C_TACC(9);
            }
            bsel ^= 1;
            xoff = bsel * BUF;
            taps_left = pl.A;
            steps_left = nsteps;
            --stages_left;
            load_first(&Xs[xoff]);                // (past the last stage: a harmless read of the other buffer)
        }
        done = stages_left == 0;
    };
    while (true) {
        step(a0, a2);
        if (done) break;
        step(a1, a0);
        if (done) break;
        step(a2, a1);
        if (done) break;
    }

    if (wave == 0) X6C_STAMP(2);
    // ---- accumulators: hh + (the five small terms) -------------------------------------------------------
    // From here on the descriptor is read through the kernel-argument segment again (`p` is the first argument): the two
    // dozen fields only the epilogue needs then do not occupy scalar registers during the main loop (the compiler loads
    // every by-value field it sees at kernel entry; with ~85 of them live it spilled 180 SGPRs into the loop).
#if defined(PASE_HIPEMU) || !defined(__HIP_DEVICE_COMPILE__)
    const PaseConvGemm& pe = p;
    const PaseX6cPlan& ple = pl;
#else
    typedef const __attribute__((address_space(4))) char* kargs_t;      // constant address space: scalar loads
    kargs_t kargs = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kargs));
    // (copies, not references: a store through p.y could alias a reference and force a reload after every store; only the
    //  INTEGER fields are taken from the copy -- a pointer loaded from memory is a generic pointer to the compiler, i.e.
    //  flat_load / flat_store with a vmcnt(0) lgkmcnt(0) wait behind every access; the six pointers stay kernel arguments)
    PaseConvGemm pe;
    PaseX6cPlan ple;
    __builtin_memcpy(&pe, (const void*)kargs, sizeof(pe));
    __builtin_memcpy(&ple, (const void*)(kargs + ((sizeof(PaseConvGemm) + 7) & ~(size_t)7)), sizeof(ple));
#endif
    f32x16 (&acc)[NBT] = accH;
#pragma unroll
    for (int j = 0; j < NBT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] += accS[j][r];

    if constexpr (TM) {
        // weight-gradient tile: += into the caller-zeroed dw (split-K slices and other launches add into the same buffer)
        if (pl.tmode >= 2) {
            // swapped operands: tile row = input channel (x tap) -> a COLUMN of dw, tile column = output channel -> a row.
            // Through LDS (wave-private block, LDS operations of one wave execute in order): lanes 0-31 / 32-63 then add
            // 32 consecutive elements of two rows of dw per instruction instead of 64 elements 4 * ldw bytes apart.
            float* tr = reinterpret_cast<float*>(&Xs[2 * BUF + RED_CHUNKS]) + wm * TR_FLOATS;
            const int mrow = m0 + wm * 32 + fr;                       // dw column of this lane in the read pass
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                pase_wave_sync();                                     // the previous half has been read
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tr[(jj * 32 + fr) * 33 + (r & 3) + 8 * (r >> 2) + 4 * fk] = acc[half * 2 + jj][r];
                pase_wave_sync();
#pragma unroll 8
                for (int cc = 0; cc < 64; cc += 2) {
                    const int cl = cc + fk;
                    const int col = n0 + half * 64 + cl;
                    const float v = tr[cl * 33 + fr];
                    if (mrow < p.M && col < p.K) atomicAdd(p.y + (size_t)col * p.Tout + mrow, v);
                }
            }
        } else {
        const int rb = m0 + wm * 32 + 4 * fk;
#pragma unroll
        for (int j = 0; j < NBT; ++j) {
            const int col = n0 + j * 32 + fr;
            int dcol = col;                                   // column of dw
            if (ZP && pl.t_stride > 1 && col < p.K) {         // phase-ordered GEMM columns (zp_tap_of)
                const int ci = (int)div_magic((unsigned)col, pl.ncols_magic);
                dcol = ci * p.taps + zp_tap_of(col - ci * p.taps, pl);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rb + (r & 3) + 8 * (r >> 2);
                const float v = acc[j][r];
                if (m < p.M) {
                    if (col < p.K) {
                        atomicAdd(p.y + (size_t)m * p.Tout + dcol, v);
                    } else if (col == p.K && p.bias) {
                        atomicAdd(const_cast<float*>(p.bias) + m, v);
                    }
                }
            }
        }
        }
    } else {
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
        const int s = (int)div_magic(n, ple.ncols_magic);
        cs[j] = cok[j] ? s : 0;
        cq[j] = cok[j] ? (int)n - s * pe.Ncols : 0;
    }
    const bool rows_full = m0 + wm * 32 + 32 <= pe.M;        // uniform
    if (wave == 0) X6C_STAMP(12);

    if (pe.epilogue == PASE_EPI_STORE &&
        (pe.post_op == PASE_POST_POW || pe.post_op == PASE_POST_LOGPOW || pe.post_op == PASE_POST_MAG)) {
        // spectra: accumulator rows r, r + 1 (same lane) are the (re, im) parts of one frequency bin
#pragma unroll
        for (int j = 0; j < NBT; ++j) {
            const int pos = cq[j] + pe.poff;
            const int cbase = (cs[j] * pe.y_ctot + pe.y_coff) * pe.Tout + pos;
            const bool colok = cok[j] && pos >= 0 && pos < pe.Tout;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int m = rbase + (r & 3) + 8 * (r >> 2);
                if (m >= pe.M) continue;                    // M is even (host-checked)
                const float re = acc[j][r], im = acc[j][r + 1];
                float v = re * re + im * im;
                v = (pe.post_op == PASE_POST_LOGPOW) ? pe.post_scale * logf(v + pe.post_eps)
                    : (pe.post_op == PASE_POST_MAG ? pe.post_scale * sqrtf(v) : v * pe.post_scale);
                if (colok) p.y[(unsigned)(cbase + (m >> 1) * pe.Tout)] = v;
            }
        }
    } else if (pe.epilogue == PASE_EPI_STORE) {
        if (wave == 0) X6C_FSTAMP(16);
        const bool pshuf = pe.ps != 1;
        const float* biasp = (p.bias && split == 0 && !bias_init) ? p.bias : nullptr;      // (else: already in the accumulators)
        int cbase[NBT], posb[NBT];
        bool colok[NBT];
        bool interior = true;
#pragma unroll
        for (int j = 0; j < NBT; ++j) {
            posb[j] = cq[j] * pe.ps + pe.poff;
            cbase[j] = (cs[j] * pe.y_ctot + pe.y_coff) * pe.Tout + posb[j];
            colok[j] = cok[j] && (pshuf || (posb[j] >= 0 && posb[j] < pe.Tout));
            interior = interior && cok[j] && posb[j] >= 0 && posb[j] + pe.ps <= pe.Tout;
        }
        float bvs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bvs[r] = 0.f;
        auto load_bias = [&]() __attribute__((always_inline)) {
            if (biasp) {   // uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rbase + (r & 3) + 8 * (r >> 2);
                    int co = m;
                    if (pshuf) co = ple.xPerm ? (int)div_magic((unsigned)m, ple.ps_magic)
                                             : m - (int)div_magic((unsigned)m, ple.cout_magic) * pe.Cout_store;
                    if (m < pe.M) bvs[r] = biasp[co];
                }
            }
        };
        auto store_rows = [&](auto fast_tag, auto atomic_tag) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_tag)::value;
            constexpr bool ATOMIC = decltype(atomic_tag)::value;
            if constexpr (FAST && !ATOMIC) {
                if (ple.xPerm) {   // uniform: (channel, phase)-ordered rows -> runs of consecutive output samples
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int m4 = rbase + 8 * g4;                          // rows m4 .. m4 + 3 (m4 % 4 == 0)
                        const int co0 = (int)div_magic((unsigned)m4, ple.ps_magic);
                        const int ph0 = m4 - co0 * pe.ps;
                        const int n1 = min(4, pe.ps - ph0);                      // samples left in channel co0
#pragma unroll
                        for (int j = 0; j < NBT; ++j) {
                            float v[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = acc[j][4 * g4 + i] + bvs[4 * g4 + i];
                            float* d0 = p.y + (unsigned)(cbase[j] + co0 * pe.Tout + ph0);
                            if (n1 == 4) {
                                pase_store_run4(d0, v);
                            } else {          // the quad straddles two channels
                                float* d1 = p.y + (unsigned)(cbase[j] + (co0 + 1) * pe.Tout);
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
                const bool mok = FAST || m < pe.M;
                int ph = 0, co = m;
                if (pshuf) {   // uniform
                    if (ple.xPerm) {
                        co = (int)div_magic((unsigned)m, ple.ps_magic);
                        ph = m - co * pe.ps;
                    } else {
                        ph = (int)div_magic((unsigned)m, ple.cout_magic);
                        co = m - ph * pe.Cout_store;
                    }
                }
                const float bv = bvs[r];
                const int rowoff = co * pe.Tout + ph;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < NBT; ++j) {
                    float v = acc[j][r] + bv;
                    if (!FAST && pe.post_op == PASE_POST_LOG) v = pe.post_scale * logf(v == 0.f ? pe.post_eps : v);
                    if (!FAST && pe.post_op == PASE_POST_RELU) v = fmaxf(v, 0.f);
                    if (!FAST && pe.post_op == PASE_POST_SQRTPOS) v = sqrtf(fmaxf(v, 0.f));
                    const bool ok = FAST || (mok && colok[j] && (!pshuf || (unsigned)(posb[j] + ph) < (unsigned)pe.Tout));
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
        const bool fast = rows_full && pe.post_op == PASE_POST_NONE && pase_wave_all(interior) != 0;
        if (wave == 0) X6C_FSTAMP(17);
        if (fast && !pshuf && ple.splitk == 1 && ple.epi32) {
            // The common tile (whole rows, interior columns, plain store): the address arithmetic is the epilogue's cost --
            // one compute wave per SIMD, so every VALU instruction here is an idle matrix core.  Row pointers are wave-uniform
            // (scalar unit), the lane's part of the address is one 32-bit byte offset per column block (computed once): a
            // store is `global_store_dword v_off, v_val, s[row]` and nothing else.
            const unsigned to4 = (unsigned)pe.Tout * 4u;
            const int mu0 = m0 + wm * 32;                                       // uniform
            unsigned cb4[NBT];
#pragma unroll
            for (int j = 0; j < NBT; ++j) cb4[j] = (unsigned)(cbase[j] + 4 * fk * pe.Tout) * 4u;
            char* ybase = reinterpret_cast<char*>(p.y);
            if (wave == 0) X6C_FSTAMP(18);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mu = mu0 + (r & 3) + 8 * (r >> 2);                    // uniform; this lane's row = mu + 4 * fk
                char* yrow = ybase + (size_t)mu * to4;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < NBT; ++j) {
                    const float v = acc[j][r];               // (bias included: !pshuf, see bias_init)
                    *reinterpret_cast<float*>(yrow + cb4[j]) = v;
                    s1 += v;
                    s2 += v * v;
                }
                if (r == 0 && wave == 0) X6C_FSTAMP(14);
                if (r == 7 && wave == 0) X6C_FSTAMP(15);
                if (p.stat_part) {   // uniform branch
                    s1 = pase_half_sum_lane31(s1);
                    s2 = pase_half_sum_lane31(s2);
                    if (fr == 31) {
                        const int ml = mu - m0 + 4 * fk;
                        red[wn][ml][0] = s1;
                        red[wn][ml][1] = s2;
                    }
                }
            }
        } else {
        load_bias();
        if (ple.splitk > 1) {
            if (fast) store_rows(std::true_type{}, std::true_type{});
            else store_rows(std::false_type{}, std::true_type{});
        } else {
            if (fast) store_rows(std::true_type{}, std::false_type{});
            else store_rows(std::false_type{}, std::false_type{});
        }
        }
        if (wave == 0) X6C_STAMP(13);
        if (p.stat_part) {
            __syncthreads();
            // one partial (sum, sumsq) per (column tile, output row); rows are channels here
            for (int ml = tid; ml < BM; ml += NTH) {
                const int m = m0 + ml;
                if (m < pe.M) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int w = 0; w < WN; ++w) {
                        s1 += red[w][ml][0];
                        s2 += red[w][ml][1];
                    }
                    float* dst = p.stat_part + ((size_t)nt * pe.M + m) * 2;
                    dst[0] = s1;
                    dst[1] = s2;
                }
            }
        }
    } else {  // PASE_EPI_MSE_CTX: rows m = d * r + j, columns (b, t); target = label[b, d, t + j - r / 2]
        float lsum = 0.f;
        const int half = pe.r_ctx / 2;
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
                const bool mok = FAST || m < pe.M;
                const int d = (int)div_magic((unsigned)m, ple.rctx_magic);
                jj16[r] = m - d * pe.r_ctx;
                bvs[r] = (mok && p.bias && !bias_init) ? p.bias[m] : 0.f;      // (bias_init: already in the accumulators)
                lrow[r] = d * pe.Ncols + jj16[r];
            }
            // all 64 label loads of the tile first (one memory latency instead of four), then the arithmetic and stores
            float tg[NBT][16];
#pragma unroll
            for (int j = 0; j < NBT; ++j) {
                const int tb = cq[j] - half;
                const int lbase = cs[j] * pe.label_D * pe.Ncols + tb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rbase + (r & 3) + 8 * (r >> 2);
                    const bool mok = FAST || m < pe.M;
                    tg[j][r] = 0.f;
                    if ((FAST || (mok && cok[j])) && (unsigned)(tb + jj16[r]) < (unsigned)pe.Ncols)
                        tg[j][r] = p.label[(unsigned)(lbase + lrow[r])];
                }
            }
#pragma unroll
            for (int j = 0; j < NBT; ++j) {
                const int obase = cs[j] * pe.M * pe.Ncols + cq[j];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rbase + (r & 3) + 8 * (r >> 2);
                    const bool mok = FAST || m < pe.M;
                    if (FAST || (mok && cok[j])) {
                        const float pred = acc[j][r] + bvs[r];
                        const float diff = pred - tg[j][r];
                        lsum += diff * diff;
                        const unsigned o = (unsigned)(obase + m * pe.Ncols);
                        if (p.y) p.y[o] = pred;
                        if (p.grad_out) p.grad_out[o] = diff * pe.grad_scale;
                    }
                }
            }
        };
        // The common tile (whole rows, all columns valid), lean: wave-uniform row pointers + one 32-bit byte offset per column
        // block for the stores, one add per label load, the context-window range check only in column blocks that touch a
        // sequence edge (see the store epilogue: the address arithmetic was 3/4 of this epilogue's 19 k clocks per tile)
        auto mse_lean = [&](auto y_tag, auto g_tag) __attribute__((always_inline)) {
            constexpr bool HAS_Y = decltype(y_tag)::value;
            constexpr bool HAS_G = decltype(g_tag)::value;
            const unsigned nc4 = (unsigned)pe.Ncols * 4u;
            const int mu0 = m0 + wm * 32;                                       // uniform
            unsigned ooff[NBT], loff[NBT];
            bool inner[NBT];
#pragma unroll
            for (int j = 0; j < NBT; ++j) {
                ooff[j] = (unsigned)((cs[j] * pe.M + 4 * fk) * pe.Ncols + cq[j]) * 4u;
                loff[j] = (unsigned)(cs[j] * pe.label_D * pe.Ncols + cq[j] - half) * 4u;
                inner[j] = pase_wave_all(cq[j] >= half && cq[j] - half + pe.r_ctx <= pe.Ncols) != 0;
            }
            float bvs[16];
            int jj16[16];
            unsigned lrow4[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mu0 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                const int d = (int)div_magic((unsigned)m, ple.rctx_magic);
                jj16[r] = m - d * pe.r_ctx;
                lrow4[r] = (unsigned)(d * pe.Ncols + jj16[r]) * 4u;
                bvs[r] = 0.f;
            }
            if (p.bias && !bias_init) {   // uniform: through the scalar cache (see sload32)
                float bsc[32];
                sload32(p.bias + mu0, bsc);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ri = (r & 3) + 8 * (r >> 2);
                    bvs[r] = fk ? bsc[ri + 4] : bsc[ri];
                }
            }
            const char* lab = reinterpret_cast<const char*>(p.label);
            float tg[NBT][16];
#pragma unroll
            for (int j = 0; j < NBT; ++j) {
                if (inner[j]) {   // uniform
#pragma unroll
                    for (int r = 0; r < 16; ++r) tg[j][r] = *reinterpret_cast<const float*>(lab + (unsigned)(loff[j] + lrow4[r]));
                } else {
                    const int tb = cq[j] - half;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        tg[j][r] = 0.f;
                        if ((unsigned)(tb + jj16[r]) < (unsigned)pe.Ncols)
                            tg[j][r] = *reinterpret_cast<const float*>(lab + (unsigned)(loff[j] + lrow4[r]));
                    }
                }
            }
            char* ybase = reinterpret_cast<char*>(p.y);
            char* gbase = reinterpret_cast<char*>(p.grad_out);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t rowb = (size_t)(mu0 + (r & 3) + 8 * (r >> 2)) * nc4;   // uniform
#pragma unroll
                for (int j = 0; j < NBT; ++j) {
                    const float pred = acc[j][r] + bvs[r];
                    const float diff = pred - tg[j][r];
                    lsum += diff * diff;
                    if constexpr (HAS_Y) *reinterpret_cast<float*>(ybase + rowb + ooff[j]) = pred;
                    if constexpr (HAS_G) *reinterpret_cast<float*>(gbase + rowb + ooff[j]) = diff * pe.grad_scale;
                }
                if (r == 0 && wave == 0) X6C_FSTAMP(14);
                if (r == 7 && wave == 0) X6C_FSTAMP(15);
            }
        };
        bool allc = true;
#pragma unroll
        for (int j = 0; j < NBT; ++j) allc = allc && cok[j];
        if (rows_full && pase_wave_all(allc) != 0) {
            if (!ple.epi32) mse_rows(std::true_type{});
            else if (p.y && p.grad_out) mse_lean(std::true_type{}, std::true_type{});
            else if (p.grad_out) mse_lean(std::false_type{}, std::true_type{});
            else if (p.y) mse_lean(std::true_type{}, std::false_type{});
            else mse_lean(std::false_type{}, std::false_type{});
        } else mse_rows(std::false_type{});
        if (wave == 0) X6C_STAMP(13);
        lsum = pase_wave_sum64(lsum);
        if (lane == 0) red[0][wave][0] = lsum;
        __syncthreads();
        if (tid == 0) {
            double tsum = (double)red[0][0][0] + (double)red[0][1][0] + (double)red[0][2][0] + (double)red[0][3][0];
            if constexpr (SYM && !DUO) tsum += (double)red[0][4][0] + (double)red[0][5][0] + (double)red[0][6][0] + (double)red[0][7][0];
            atomicAdd(p.loss_acc, tsum);
        }
    }
    }   // !TM
    if (wave == 0) X6C_STAMP(3);
    X6C_TRACE_NEXT();
  }   // items
#if !defined(PASE_HIPEMU)
  // Close the compiler's vector-memory books before this path joins the staging waves' code in the control-flow graph (the
  // compiler lays the role branch and this exit through one block): the weight-fragment prefetches past the last step are
  // never consumed, hence never waited for, and -- reaching the staging loop as "pending loads" -- they made the compiler put
  // s_waitcnt vmcnt(1 .. 2) into the hidden load sequences of the staging waves (round 5, seen in the ISA of the 1x1 and
  // weight-gradient instantiations; pase_amd/build.py lints for it)
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
#endif
}

#ifdef PASE_X6C_TRACE
extern "C" int pase_x6c_trace_read(unsigned long long* host) {
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x6c_trace), sizeof(unsigned long long) * 2 * X6C_TRACE_ITEMS * 20);
}
extern "C" int pase_x6c_trace_reset() {
    static unsigned long long zeros[2 * X6C_TRACE_ITEMS * 20];
    hipDeviceSynchronize();
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_x6c_trace), zeros, sizeof(zeros));
}
#endif

// ================================================================================================================
// Weight gradients on pre-split planes, SYMMETRIC form (round 6): all eight waves multiply.
//   dw[m, (ci, kk)] += sum_{s, q} g~[s, m, q] * z~[s, ci, q * stride + kk * tapstep - padL]        (tmode 1 with pl.zp)
// conv_x6c_kernel<128, 5, true, true> keeps four waves for staging that have nothing to do but issue three DMA instructions per
// k-group, while its four compute waves -- one per SIMD, nothing else to cover an L2 round trip or an LDS read -- run a 24-MFMA
// step in ~1160 clocks (768 = the pipe) and every MFMA needs 256 bytes from L2 (12 KB of packed rows + 12 KB of staged columns
// per step).  Here the workgroup tile is 256 x 128: wave w owns rows 32 w .. 32 w + 31 (same 32 x 128 wave tile, same two
// accumulator sets, same step), TWO multiplying waves per SIMD cover each other's waits, the staged columns of a k-group feed
// twice the MFMAs (192 bytes per MFMA), and the stage's 72 DMA instructions are shared out over all eight waves -- three at the
// top of each of a stage's first three steps.  A stage = SIX k-groups (2 x 6 x 12 KB of LDS) = two turns of the three fragment
// register sets: the loop body is one stage, unrolled, with the buffer chosen at run time.
// The DMA is HIDDEN from the compiler (inline asm, M0 saved / restored): visible, it makes every later LDS read that may alias
// its destination wait vmcnt(0) and every __syncthreads a full drain (the first build's ISA: one wait per step, accumulators
// spilled around the two-buffer loop).  Hidden, the vector-memory counter still retires in order, so the compiler's own
// s_waitcnt vmcnt(6) in front of a step's first MFMA (fragments of steps s + 1, s + 2 may stay in flight) now leaves "the six
// newest operations" in flight, three of them this step's DMA: the fragment prefetch is effectively one step deep where DMA
// is issued -- with two waves per SIMD a step is ~1.5 k clocks, more than an L2 round trip.  In front of the stage barrier
// s_waitcnt vmcnt(6) by hand: the stage's DMA is nine fragment loads old.
constexpr int SYM_KGS = 6;
__global__ void __launch_bounds__(NT, 2) x6c_wgrad_sym_kernel(PaseConvGemm p, PaseX6cPlan pl) {
    constexpr int NPOS = 128, PLANE = 2 * NPOS, KGC = 3 * PLANE, BUF = SYM_KGS * KGC, NBT = 4;
    __shared__ __attribute__((aligned(16))) u32x4 Xs[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = pase_uniform(tid >> 6);
    const int fr = lane & 31, fk = lane >> 5;
    // DMA role of this wave: octet fkL and column half whalf of the k-groups kgsel, kgsel + 2, kgsel + 4 of a stage
    const int fkL = (wave >> 1) & 1, whalf = wave & 1, kgsel = wave >> 2;
    const int ntiles = pl.n_row_tiles * pl.n_col_tiles;
    const int nitems = ntiles * pl.splitk;
    const int GS = (pl.G + SYM_KGS - 1) / SYM_KGS;
    const int g_per = (GS + pl.splitk - 1) / pl.splitk;
    const unsigned short* zpb = reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(p.wx6) + pl.zp_off);
    const X6cRsrc a_rs = x6c_make_rsrc(p.wx6);
    const unsigned a_loff = 16u * (unsigned)lane;
    int bbase[NBT];
#pragma unroll
    for (int j = 0; j < NBT; ++j) bbase[j] = fk * NPOS + j * 32 + fr;
    int bsel = 0;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int split = item / ntiles;
        const int tile = xcd_swizzle(item - split * ntiles, ntiles);
        const int mt = tile / pl.n_col_tiles, nt = tile - mt * pl.n_col_tiles;      // consecutive tiles: the column tiles of a row tile
        const int m0 = mt * 256, n0 = nt * 128;
        const int g_begin = split * g_per;
        const int g_end = min(GS, g_begin + g_per);
        if (g_begin >= g_end) continue;                                             // uniform
        const int nst = g_end - g_begin;
        // ---- this lane's column of the planes (see conv_x6c_kernel's setup_item, ZP): column j = (ci, kk) -> plane row ci * stride
        // + b shifted by d; column K = the all-ones row (bias gradient); columns past the end alias the last real one
        unsigned zp_col;
        {
            const int j = n0 + 64 * whalf + lane;
            const bool ones = p.bias != nullptr && j == p.K;
            const int jj = min(j, p.K - 1);
            const int ci = (int)div_magic((unsigned)jj, pl.ncols_magic);
            const int kk = zp_tap_of(jj - ci * p.taps, pl);
            const int ob = kk * p.tapstep - p.padL - pl.t_dmin * p.stride;
            const int db = (int)div_magic((unsigned)ob, pl.ps_magic);
            const int b = ob - db * p.stride;
            const int row = ones ? pl.zp_rows - 1 : ci * p.stride + b;
            const int sh = ones ? 0 : db;
            const int odd = sh & 1;
            zp_col = (unsigned)odd * (unsigned)(pl.t_plane / 2) + (unsigned)(row * p.S) * (unsigned)pl.t_lseg + (unsigned)(sh - odd + fkL * 8);
        }
        // the three plane chunks of k-group kg of stage g -> stage buffer bs (this wave's octet / column half)
        auto dma_kg = [&](int bs, int g, int kg) __attribute__((always_inline)) {
            const int kgi = g * SYM_KGS + kg;
            const int s_ = (int)div_magic((unsigned)kgi, pl.seg_magic);
            const int q16 = kgi - s_ * pl.P;
            const unsigned off = zp_col + (unsigned)(min(s_, p.S - 1) * pl.t_lseg + q16 * 16);
            u32x4* dst = &Xs[bs * BUF + kg * KGC + fkL * NPOS + 64 * whalf];
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) x6c_dma16_hidden(zpb + (size_t)pz * (size_t)pl.t_plane + off, dst + pz * PLANE, lane);
        };
        f32x16 accH[NBT], accS[NBT];
#pragma unroll
        for (int j = 0; j < NBT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accH[j][r] = 0.f;
                accS[j][r] = 0.f;
            }
        // packed rows of g: [32-row tile][step][plane][lane], one scalar byte offset that advances 3072 per step
        unsigned ab = (unsigned)(16 * (((size_t)(mt * 8 + wave) * (unsigned)pl.steps_total + (size_t)g_begin * SYM_KGS) * 192u));
        const int nsteps_run = nst * SYM_KGS;
        int a_issued = 0;
        auto load_a = [&](u32x4 (&a)[3]) __attribute__((always_inline)) {
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) a[pz] = x6c_buffer_load16(a_rs, a_loff + 1024u * pz, ab);
            ++a_issued;
            ab += (a_issued < nsteps_run) ? 3072u : 0u;         // (past the end: a harmless re-read of the last fragments)
        };
        u32x4 bq[2][2][3];
        auto load_first = [&](const u32x4* xb) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pz = 0; pz < 3; ++pz) bq[0][i][pz] = xb[pz * PLANE + bbase[i]];
        };
        // one 24-MFMA step (conv_x6c_kernel's mfma_step): the six fragments of the next half read one per two MFMAs
        auto mfma_step = [&](const u32x4 (&a)[3], const u32x4* xb, const u32x4* xn) __attribute__((always_inline)) {
            constexpr int PZA[6] = {1, 0, 2, 0, 1, 0}, PZB[6] = {1, 2, 0, 1, 0, 0};
            PASE_SGB(0x020, 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4* src = h == 0 ? xb : xn;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    bq[h ^ 1][i / 3][i % 3] = src[(i % 3) * PLANE + bbase[2 * (h ^ 1) + i / 3]];
                    if (i < 5) {
                        accS[2 * h] = pase_mfma_bf16_32x32x16(a[PZA[i]], bq[h][0][PZB[i]], accS[2 * h]);
                        accS[2 * h + 1] = pase_mfma_bf16_32x32x16(a[PZA[i]], bq[h][1][PZB[i]], accS[2 * h + 1]);
                    } else {
                        accH[2 * h] = pase_mfma_bf16_32x32x16(a[0], bq[h][0][0], accH[2 * h]);
                        accH[2 * h + 1] = pase_mfma_bf16_32x32x16(a[0], bq[h][1][0], accH[2 * h + 1]);
                    }
                    PASE_SGB(0x100, 1);
                    PASE_SGB(0x008, 2);
                }
            }
        };
        u32x4 a0[3], a1[3], a2[3];
        // ---- first stage of the item (every wave has passed the barrier that ended the previous item's last stage)
        dma_kg(bsel, g_begin, kgsel);
        dma_kg(bsel, g_begin, kgsel + 2);
        dma_kg(bsel, g_begin, kgsel + 4);
        load_a(a0);
        load_a(a1);
        x6c_vm_drain();
        __syncthreads();
        for (int gi = 0; gi < nst; ++gi) {
            // the next stage arrives in the other buffer while this one is multiplied (past the item's last stage: that stage
            // once more into the idle buffer -- a branch around the copies would split the steps into basic blocks)
            const int g_next = min(g_begin + gi + 1, g_end - 1);
            const u32x4* CUR = &Xs[bsel * BUF];
            load_first(CUR);
            auto st = [&](auto k_tag, const u32x4 (&acur)[3], u32x4 (&anxt)[3]) __attribute__((always_inline)) {
                constexpr int k = decltype(k_tag)::value;
                if constexpr (k < 3) dma_kg(bsel ^ 1, g_next, kgsel + 2 * k);
                load_a(anxt);
                mfma_step(acur, CUR + k * KGC, CUR + (k < SYM_KGS - 1 ? k + 1 : k) * KGC);
            };
            st(std::integral_constant<int, 0>{}, a0, a2);
            st(std::integral_constant<int, 1>{}, a1, a0);
            st(std::integral_constant<int, 2>{}, a2, a1);
            st(std::integral_constant<int, 3>{}, a0, a2);
            st(std::integral_constant<int, 4>{}, a1, a0);
            st(std::integral_constant<int, 5>{}, a2, a1);
#if !defined(PASE_HIPEMU)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // this stage's DMA has landed (it is >= nine fragment loads old)
#endif
            __syncthreads();
            bsel ^= 1;
        }
        // ---- += into the caller-zeroed dw (row-contiguous atomics; the bias column adds into dbias)
#pragma unroll
        for (int j = 0; j < NBT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accH[j][r] += accS[j][r];
        const int rb = m0 + wave * 32 + 4 * fk;
#pragma unroll
        for (int j = 0; j < NBT; ++j) {
            const int col = n0 + j * 32 + fr;
            int dcol = col;
            if (pl.t_stride > 1 && col < p.K) {
                const int ci = (int)div_magic((unsigned)col, pl.ncols_magic);
                dcol = ci * p.taps + zp_tap_of(col - ci * p.taps, pl);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rb + (r & 3) + 8 * (r >> 2);
                const float v = accH[j][r];
                if (m < p.M) {
                    if (col < p.K) atomicAdd(p.y + (size_t)m * p.Tout + dcol, v);
                    else if (col == p.K && p.bias) atomicAdd(const_cast<float*>(p.bias) + m, v);
                }
            }
        }
#if !defined(PASE_HIPEMU)
        // the fragment prefetches past the last step are never consumed and the last stage's surplus copy is in flight: drain
        // both before the next item's prologue writes into the buffers
        __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
        __syncthreads();
    }
}

// weights (K-major fp32 pack wt[k * ldwt + m], k = ci * taps + kk) -> fragment-ordered bf16 planes:
// out[((rt32 * steps + st) * 3 + plane) * 64 + lane], step st = g * A + a, lane = (fk, row): element e = channel'
// 16 g + 8 fk + e at tap' a.  Zero for channels' past Cin * P, taps past the real count and rows past M.
// (wt == NULL: straight from the weight as the reference stores it, w[m * ldw + (tap_major ? kk * Cin + ci : ci * taps + kk)] --
//  the K-major intermediate is only needed by the fp32-pipe kernels; one launch less per split-bf16 convolution)
__global__ void pack_x6c_kernel(const float* __restrict__ wt, u32x4* __restrict__ out, int M, int ldwt, int Cin, int taps,
                                int P, int A, int CinP, int rev, int steps, long total, int perm_ps, int perm_cout,
                                const float* in_scale, const float* in_shift, const float* in_alpha, float* prm, int prm_n,
                                const float* __restrict__ w, int ldw, int tap_major) {
    // the on-load parameters expanded per channel' behind the chunks (same launch: one pack launch per GEMM launch)
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < prm_n; c += (long)gridDim.x * blockDim.x) {
        const int ci = min((int)c / P, Cin - 1);
        const bool ok = c < CinP;
        prm[c] = (ok && in_scale) ? in_scale[ci] : 1.f;
        prm[prm_n + c] = (ok && in_scale) ? in_shift[ci] : 0.f;
        prm[2 * prm_n + c] = (ok && in_alpha) ? in_alpha[ci] : 1.f;
    }
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
            const bool ok = cp < CinP && kk >= 0 && kk < taps && m < M;
            if (wt) v[e] = ok ? wt[((size_t)ci * taps + kk) * ldwt + msrc] : 0.f;
            else v[e] = ok ? w[(size_t)msrc * ldw + (tap_major ? kk * Cin + ci : ci * taps + kk)] : 0.f;
        }
        u32x4 o[3];
        pase_split_bf16x3_rne(v, o);
        u32x4* dst = out + ((size_t)rs * 3) * 64 + lane;
#pragma unroll
        for (int pz = 0; pz < 3; ++pz) dst[pz * 64] = o[pz];
    }
}

// on-load parameters per channel' behind the weight chunks: [scale | shift | alpha], prm_n floats each (identity where
// the descriptor has none, and for the zero channels' that pad the last stage)
// rows of an activation tensor -> fragment-ordered bf16 planes with the contraction over POSITIONS (weight gradients):
// out[((rt32 * steps + st) * 3 + plane) * 64 + lane], step st = k-group (sequence s = st / QP16, positions 16 (st % QP16) ..),
// lane = (fk, row): element e = position 16 (st % QP16) + 8 fk + e of row 32 rt32 + row, after its on-load transform
// v -> prelu(v * sc[row] + sh[row], al[row]) (NULL arrays = identity).  Zero past Ncols, past S and past `rows`.
__global__ void pack_rows_x6c_kernel(const float* __restrict__ src, u32x4* __restrict__ out, int rows, int S, int ctot,
                                     int coff, int T, int Ncols, int QP16, int steps, long total, const float* sc,
                                     const float* sh, const float* al) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long rs = idx >> 6;
        const int st = (int)(rs % steps);
        const int rt = (int)(rs / steps);
        const int s_ = st / QP16;
        const int fk = lane >> 5, row = lane & 31;
        const int q0 = (st - s_ * QP16) * 16 + fk * 8;
        const int m = rt * 32 + row;
        float v[8];
        const bool rok = m < rows && s_ < S;
        const float* r = src + ((size_t)(rok ? s_ : 0) * ctot + coff + (rok ? m : 0)) * T;
        const float a_sc = (rok && sc) ? sc[m] : 1.f, a_sh = (rok && sc) ? sh[m] : 0.f, a_al = (rok && al) ? al[m] : 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = (rok && q0 + e < Ncols) ? r[q0 + e] : 0.f;
            if (rok && q0 + e < Ncols) {
                t = fmaf(t, a_sc, a_sh);
                t = t > 0.f ? t : t * a_al;
            }
            v[e] = t;
        }
        u32x4 o[3];
        pase_split_bf16x3_rne(v, o);
        u32x4* dst = out + ((size_t)rs * 3) * 64 + lane;
#pragma unroll
        for (int pz = 0; pz < 3; ++pz) dst[pz * 64] = o[pz];
    }
}

// tmode 3: z~ (on-load transform applied, padding materialised) as three row-major bf16 planes,
//   out[plane][r = ci * st + b][s][i],  i in [0, lseg):  z~[s][ci][st * (i + dmin) + b]   (reflected / zero outside [0, T))
// followed by a zero tail (the k-groups a last stage runs past the last sequence read on into the next row, or into the tail).
// One thread per 8 consecutive elements of a plane (t_plane % 8 == 0: 16-byte stores).
__global__ void pack_zplanes_kernel(const float* __restrict__ src, u32x4* __restrict__ out, int Cin, int S, int ctot, int coff,
                                    int T, int st, int lseg, int dmin, int pad_mode, long t_plane, const float* sc,
                                    const float* sh, const float* al) {
    const long ngroups = t_plane / 8;
    const long body = (long)Cin * st * S * lseg;
    const long half = t_plane / 2;              // second copy: element i = element i + 1 of the first
    for (long gidx = (long)blockIdx.x * blockDim.x + threadIdx.x; gidx < ngroups; gidx += (long)gridDim.x * blockDim.x) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            long idx = gidx * 8 + e;
            if (idx >= half) idx = idx - half + 1;
            float t = 0.f;
            if (idx < body) {
                const long rs = idx / lseg;
                const int i = (int)(idx - rs * lseg);
                const int s_ = (int)(rs % S);
                const int r = (int)(rs / S);
                const int ci = r / st, b = r - ci * st;
                int u = st * (i + dmin) + b;
                if (pad_mode == PASE_PAD_REFLECT) u = x6c_reflect(u, T);
                if (u >= 0 && u < T) {
                    t = src[((size_t)s_ * ctot + coff + ci) * T + u];
                    if (sc) t = fmaf(t, sc[ci], sh[ci]);
                    if (al) t = t > 0.f ? t : t * al[ci];
                }
            }
            v[e] = t;
        }
        u32x4 o[3];
        pase_split_bf16x3_rne(v, o);
#pragma unroll
        for (int pz = 0; pz < 3; ++pz) out[(pz * t_plane) / 8 + gidx] = o[pz];
    }
}

// XP: the activation of a stride-1 convolution launch as channel-minor bf16 planes in the LDS image's chunk order,
//   out[plane][g][fk][s][up]: 16-byte chunk = pieces of act(bn(x[s][16 g + 8 fk + e][up - padLp])), e = 0 .. 7
// (zero / reflected outside [0, Tin), zero for channels past Cin).  Thread = one padded position of one (g, fk, s) row: the
// eight channel loads of a wave are 256 contiguous bytes each, its three stores 1 KB each.  grid = (position blocks, 2 G, S).
__global__ void pack_xcm_kernel(const float* __restrict__ x, u32x4* __restrict__ out, int Cin, int S, int ctot, int coff,
                                int Tin, int tpad, int padLp, int pad_mode, long plane, const float* sc, const float* sh,
                                const float* al) {
    const int up = blockIdx.x * blockDim.x + threadIdx.x;
    if (up >= tpad) return;
    const int gf = blockIdx.y, s_ = blockIdx.z;
    const int c0 = gf * 8;                                  // (16 g + 8 fk)
    int u = up - padLp;
    if (pad_mode == PASE_PAD_REFLECT) u = x6c_reflect(u, Tin);
    const bool in = u >= 0 && u < Tin;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        float t = 0.f;
        if (in && c < Cin) {
            t = x[((size_t)s_ * ctot + coff + c) * Tin + u];
            if (sc) t = fmaf(t, sc[c], sh[c]);
            if (al) t = t > 0.f ? t : t * al[c];
        }
        v[e] = t;
    }
    u32x4 o[3];
    pase_split_bf16x3_rne(v, o);
    u32x4* dst = out + ((size_t)gf * S + s_) * (size_t)tpad + up;
#pragma unroll
    for (int pz = 0; pz < 3; ++pz) dst[(size_t)pz * (size_t)plane] = o[pz];
}

// ZP (tmode 1 with pl.zp): z~ (on-load transform applied, padding materialised) as three phase-decomposed bf16 planes,
//   out[plane][row = ci * st + b][s][i],  i in [0, lseg):  piece of z~[s][ci][st * (i + dmin) + b]   (reflected / zero outside
//   [0, T)),  plus -- `ones` -- a last row of 1.0 (the bias column of the weight gradient).
// Flat work list, one thread per (sequence s, channel ci, 8-position group gq, phase b) with b fastest, then gq: the eight
// loads of a wave walk a contiguous stretch of one source row (every byte of a fetched line is used within the eight
// instructions) and each thread writes one 16-byte chunk per plane into row ci * st + b.  (A workgroup per (s, ci) row -- the
// first form -- left most of its threads idle on the short rows of the upper layers: 54 work items per 256-thread block on
// block 7, 1.8 ms per step in eight launches.)
__global__ void pack_zph_kernel(const float* __restrict__ src, u32x4* __restrict__ out, int Cin, int S, int ctot, int coff,
                                int T, int st, int lseg, int dmin, int pad_mode, long t_plane, const float* sc,
                                const float* sh, const float* al, int ones) {
    const int ngrp = lseg / 8;
    const int per_row = ngrp * st;                             // work items of one (s, ci) source row
    const long nrows = (long)S * (Cin + (ones ? 1 : 0));
    const long total = nrows * per_row;
    const size_t pstride = (size_t)(t_plane / 8);
    for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (long)gridDim.x * blockDim.x) {
        const long rowi = w / per_row;
        const int idx = (int)(w - rowi * per_row);
        const int C1 = Cin + (ones ? 1 : 0);
        const int s_ = (int)(rowi / C1), ci = (int)(rowi - (long)s_ * C1);       // source rows in memory order
        const int gq = idx / st, b = idx - gq * st;
        if (ci == Cin) {                                       // the ones row (bias column): 1.0 in hi, zero mid / lo
            if (b == 0) {
                u32x4* orow = out + ((size_t)(Cin * st) * S + s_) * (size_t)ngrp + gq;
                const u32x4 one = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, zero = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
                    orow[cp * (pstride / 2)] = one;
                    orow[pstride + cp * (pstride / 2)] = zero;
                    orow[2 * pstride + cp * (pstride / 2)] = zero;
                }
            }
            continue;
        }
        const float* row = src + ((size_t)s_ * ctot + coff + ci) * T;
        const float a_sc = sc ? sc[ci] : 1.f, a_sh = sc ? sh[ci] : 0.f, a_al = al ? al[ci] : 1.f;
        float v[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            int u = st * (gq * 8 + e + dmin) + b;
            if (pad_mode == PASE_PAD_REFLECT) u = x6c_reflect(u, T);
            float t = 0.f;
            if (u >= 0 && u < T) {
                t = fmaf(row[u], a_sc, a_sh);
                t = t > 0.f ? t : t * a_al;
            }
            v[e] = t;
        }
        u32x4 o0[3], o1[3];
        pase_split_two_windows(v, o0, o1);
        u32x4* orow = out + ((size_t)(ci * st + b) * S + s_) * (size_t)ngrp + gq;
#pragma unroll
        for (int pz = 0; pz < 3; ++pz) {
            orow[pz * pstride] = o0[pz];
            orow[pz * pstride + pstride / 2] = o1[pz];          // second copy: elements one later
        }
    }
}

// (wave priorities: s_setprio of either role changed nothing on the PASE+ step -- DESIGN.md 3.0; the plan field stays for
//  the trace build's ablations)
#ifdef PASE_X6C_TRACE
int x6c_prio() {      // trace builds only (tools/trace_x6c.py): ablation bits
    const char* e = getenv("PASE_X6C_ABLATE");
    return e ? atoi(e) : 0;
}
#else
int x6c_prio() { return 0; }
#endif

// persistent grids: one workgroup per CU of the device the launch goes to (256 on an MI355X in SPX mode; a partitioned
// device reports its own count).  The split-K models below are tuned for 256 and stay so.
int x6c_cu_count() {
#ifdef PASE_HIPEMU
    return 256;
#else
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        n = (hipGetDevice(&dev) == hipSuccess &&
             hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return n;
#endif
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
    // tile 128 x 128 (waves 4 x 1).  The operand split is paid once per staged element and shared by BM / 32 row tiles
    // x A taps: launches of at most 64 rows with fewer than four taps' would spend more issue slots splitting than
    // multiplying -- they stay on the fp32 matrix pipe
    const bool force = (p.x6_ctl & 1) != 0;      // measurement runs: skip the routing rules below
    // ... and so do launches with fewer than 128 k (eight MFMA steps per tile): they are store-bound
    if (!force && (long)pl.CinP * pl.A < 128) return false;
    // Measured on the PASE+ bs32 step (profiles/gemm_launches_r03.json): 1x1 launches with K < 768 (the 256-channel worker
    // heads: six stages per tile, the staging waves' conversion work per MFMA is 5x that of an 11-tap layer) are faster on
    // the exact-fp32 matrix pipe unless they have thousands of row tiles to amortise a column tile's staging over (the
    // 21 525-channel heads: 0.81 -> 0.75 ms) or hardly any columns at all (the 128-column classifier launches)
    // Round 4: with the activation pre-split once per launch (XP, see the kernel) the conversion no longer scales with the row
    // tiles; launches with >= 1024 rows and one or two taps take it (x6_ctl bit 1 forces it on any stride-1 launch, bit 2 forbids
    // it), which also lifts the K < 768 rule for them (the stacked 2304-row first layers of the MLP heads)
    const bool xp_ok = pl.P == 1 && (long)pl.G * 2 * p.S * (long)(p.Ncols + pl.A - 1) < (1L << 27);
    // (with the planes staged by global_load_lds the break-even moved: one or two taps from 512 rows, three taps from 1024 --
    //  PASE+ bs32, same box: the decoder's 1280-row 3-tap layer 1.26 -> 1.10 ms, the 512-row QRNN projection 0.45 -> 0.41,
    //  block 1's data gradient 0.58 -> 0.50; 6- and 11-tap layers and the 256-row launches lose 5 ... 15 % to the pack)
    const bool xp_want = xp_ok && !(p.x6_ctl & 4) &&
                         ((p.x6_ctl & 2) || (pl.A <= 2 && p.M >= 512) || (pl.A <= 3 && p.M >= 1024));
    // Round 4: launches of at most 64 rows get a 64 x 256 tile (NARROW) when they have taps' to walk and >= 512 k (block 1 of
    // the encoder: 640 channels' x 2 taps'); the 1x1 launches of 64 rows have 8 ... 16 MFMA steps per tile and are store-bound
    // on either pipe.  x6_ctl bit 4 forbids it (A/B measurements)
    const bool narrow = p.M <= 64 && pl.A >= 2 && (long)pl.CinP * pl.A >= 512 && !xp_want && !(p.x6_ctl & 16);
    if (!force && !narrow && p.M <= 64 && pl.A < 4) return false;
    if (!force && pl.A == 1 && pl.CinP < 768 && p.M < 8192 && (long)p.S * p.Ncols > 128 && !xp_want) return false;
    pl.NBT = 4;
    pl.WM = 4;
    pl.BM = 128;
    pl.BN = 128;
    // Round 6: launches on a pre-split activation with at least 256 rows and an epilogue without BatchNorm partial sums run the
    // SYMMETRIC form (256 x 128 tile, all eight waves multiply and share out the stage's LDS DMA; conv_x6c_kernel<..., SYM>).
    // Decided from the descriptor alone -- NOT from xp6 being there: the weight pack is sized and written before the caller
    // has the activation planes, and its row-tile count is the plan's.  x6_ctl bit 7 forbids it (A/B runs).
    const bool spectrum_op = p.post_op == PASE_POST_POW || p.post_op == PASE_POST_LOGPOW || p.post_op == PASE_POST_MAG;
    pl.sym = (xp_want && !narrow && p.M >= 256 && !p.stat_part && !spectrum_op && !(p.x6_ctl & 128)) ? 1 : 0;
    // ... or its four-wave variant, two workgroups per CU (DUO, pl.sym == 2): same 128-row tiling as the staging-wave form, so it
    // has no round-count condition and takes launches from 128 rows on.  x6_ctl bits 17 / 18 (A/B runs): bit 17 = the eight-wave
    // form wherever it is eligible, never DUO; bit 18 = DUO wherever it is eligible.
    const bool duo_ok = xp_want && !narrow && p.M >= 128 && !p.stat_part && !spectrum_op && !(p.x6_ctl & 128);
    if ((p.x6_ctl & 0x40000) && duo_ok) pl.sym = 2;
    // Routing between the two (measured per launch on the PASE+ bs32 step, same box, profiles/experiments/ab_r06.json): 1x1
    // launches take DUO (LPS heads 0.590 staging-wave -> 0.520 eight-wave -> 0.500 DUO; M = 1920: 0.145 -> 0.132); launches with
    // taps take the eight-wave form where its 256-row tiles fill whole rounds (QRNN projection 0.351 -> 0.331, DUO 0.357) and DUO
    // otherwise (M = 512 x 2 taps 0.411 -> 0.395, block 1's data gradient 0.520 -> 0.463)
    if (pl.sym == 1 && pl.A == 1 && duo_ok && !(p.x6_ctl & 0x20000)) pl.sym = 2;
    if (pl.sym == 1 && !force && !(p.x6_ctl & 0x20000)) {
        // ... where the 256-row tiles do not cost whole rounds of the persistent grid: a round of 256-row tiles takes about two
        // rounds of 128-row tiles (0.9 x measured), so the form is taken when 2 x its rounds <= the rounds of the 128-row tiling.
        // Measured on the PASE+ bs32 step, same box: LPS heads (85 x 50 tiles against 169 x 50) 0.568 -> 0.516 ms, the 840-row
        // heads -8 %, QRNN projection -2 %; M = 512 / 640 / 1920 (two, 2.5 and 7.5 tiles of 256 rows: 4 rounds against 3, 58 against
        // 47, 10 against 9) +14 % / +10 % / +6 % -- those keep the staging-wave form.
        const long cap = p.max_wg > 0 ? p.max_wg : x6c_cu_count();
        const long ncolt = ((long)p.S * p.Ncols + 127) / 128;
        const long r_sym = (((long)(p.M + 255) / 256) * ncolt + cap - 1) / cap;
        const long r_std = (((long)(p.M + 127) / 128) * ncolt + cap - 1) / cap;
        if (2 * r_sym > r_std) pl.sym = duo_ok ? 2 : 0;
    }
    if (pl.sym == 0 && duo_ok && !(p.x6_ctl & 0x20000)) pl.sym = 2;      // (pre-split launches of 128 ... 255 rows)
    if (pl.sym == 1) {
        pl.WM = 8;
        pl.BM = 256;
    }
    if (narrow) {          // 64 x 256 (conv_x6c_kernel<320, 2, ..., NARROW>)
        pl.WM = 2;
        pl.BM = 64;
        pl.BN = 256;
    }
    // every sequence a column tile touches carries its own halo of A - 1 positions
    const int nseg_max = (pl.BN - 2) / p.Ncols + 2;
    if ((long)nseg_max * (pl.A - 1) > HALO_MAX) return false;
    // k-groups per stage: three for 1x1 layers (no halo: 12 KB per k-group), two where a stage would otherwise be
    // shorter than four steps
    pl.KGS = pl.A == 1 ? (pl.G >= 3 ? 3 : pl.G) : ((pl.A < 4 && pl.G >= 2) ? 2 : 1);
#ifndef PASE_X6C_KGS3      // (A/B builds: always three)
    // 1x1 launches whose k-groups do not fill whole stages of three: the zero k-groups that pad the last stage are multiplied
    // like any other (K = 256: 16 k-groups = 5 stages + 1 group, 18 steps for 16) -- two per stage then, when that divides
    if (pl.A == 1 && pl.G > 3 && pl.G % 3 != 0 && pl.G % 2 == 0 && pl.G <= 64) pl.KGS = 2;
#endif
    if (narrow && pl.KGS > 2) pl.KGS = 2;
    if (pl.sym && pl.KGS * pl.A < 2) {      // (a stage of one step has no second step to issue the second DMA unit in)
        pl.sym = 0;
        pl.WM = 4;
        pl.BM = 128;
    }
    const int GS = (pl.G + pl.KGS - 1) / pl.KGS;
    pl.steps_total = GS * pl.KGS * pl.A;
    const long ntot = (long)p.S * p.Ncols;
    pl.n_row_tiles = (p.M + pl.BM - 1) / pl.BM;
    pl.n_col_tiles = (int)((ntot + pl.BN - 1) / pl.BN);
    pl.prio = x6c_prio();
    pl.stagger = (p.x6_ctl >> 8) & 255;
    {   // the lean epilogues address the output with 32-bit BYTE offsets
        const size_t out_elems = p.epilogue == PASE_EPI_STORE ? (size_t)p.S * p.y_ctot * p.Tout : (size_t)p.S * p.M * p.Ncols;
        const size_t lab_elems = p.epilogue == PASE_EPI_STORE ? 0 : (size_t)p.S * p.label_D * p.Ncols;
        pl.epi32 = (out_elems < (1u << 29) && lab_elems < (1u << 29) && !(p.x6_ctl & 32)) ? 1 : 0;
    }
    pl.tmode = 0;
    pl.pairs = (pl.P > 1 && (pl.P & 1) == 0 && !(p.x6_ctl & 0x10000)) ? 1 : 0;      // (x6_ctl bit 16: single loads, A/B runs)
    pl.xp_tpad = p.Ncols + pl.A - 1;
    pl.xp_plane = xp_want ? (long)pl.G * 2 * p.S * pl.xp_tpad : 0;
    pl.xp = (xp_want && p.xp6 != nullptr) ? 1 : 0;
    pl.pack_chunks = (long)pl.n_row_tiles * pl.WM * pl.steps_total * 192;
    pl.prm_n = GS * pl.KGS * 16;
    pl.pack_bytes = (pl.pack_chunks * 16 + 3L * pl.prm_n * 4 + 15) / 16 * 16;
    if (pl.pack_bytes >= (1L << 32)) return false;      // the weight fragments are addressed with 32-bit scalar offsets
    pl.ncols_magic = magic_of(p.Ncols);
    pl.cout_magic = magic_of(p.Cout_store);
    pl.ps_magic = magic_of(p.ps);
    pl.rctx_magic = magic_of(p.r_ctx);
    pl.seg_magic = magic_of(p.Ncols + pl.A - 1);
    pl.p_magic = magic_of(pl.P);
    pl.xPerm = (p.ps > 1 && p.epilogue == PASE_EPI_STORE && !p.stat_part && p.post_op == PASE_POST_NONE &&
                p.M == p.ps * p.Cout_store) ? 1 : 0;
    // split-K (data gradients of the wide heads, decoder layers whose tiles do not fill whole rounds): the persistent grid
    // deals (slice, tile) items round-robin to 256 workgroups, so a launch takes ceil(items / 256) item times; an item of a
    // 1/sk slice costs 1/sk of the tile plus the atomic flush (~2 stages)
    const long tiles = (long)pl.n_row_tiles * pl.n_col_tiles;
    int splitk = 1;
    if (p.splitk > 1) splitk = p.splitk;
    else if (p.splitk == 0 && !p.stat_part && p.epilogue == PASE_EPI_STORE && p.post_op == PASE_POST_NONE && GS >= 8) {
        const double flush = 2.0 / (double)GS;
        double best = 1e30;
        const int max_split = GS / 4 < 1 ? 1 : GS / 4;
        for (int sk = 1; sk <= max_split && sk <= 64; ++sk) {
            const long rounds = (tiles * sk + 255) / 256;
            const double est = (double)rounds * (1.0 / sk + (sk > 1 ? flush : 0.0));
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
                pl.steps_total, total, pl.xPerm ? p.ps : 1, p.Cout_store, p.in_scale, p.in_shift, p.in_alpha,
                reinterpret_cast<float*>(reinterpret_cast<u32x4*>(const_cast<void*>(p.wx6)) + pl.pack_chunks), pl.prm_n,
                p.w, p.ldw, p.tap_major);
    PASE_CHECK_LAUNCH();
    return 0;
}

int pase_x6c_launch(const PaseConvGemm& p, const PaseX6cPlan& pl, hipStream_t st) {
    // persistent grid: one workgroup per CU (8 waves, 74 KB of LDS), items dealt round-robin
    long nwg = (long)pl.n_row_tiles * pl.n_col_tiles * pl.splitk;
    const long cap = p.max_wg > 0 ? p.max_wg : x6c_cu_count();      // data-parallel runs leave CUs to RCCL; tests force several items per workgroup
    if (nwg > cap) nwg = cap;
    const dim3 grid((unsigned)nwg), block(NT);
    if (pl.WM == 2) {
        PASE_LAUNCH((conv_x6c_kernel<320, 2, false, false, true>), grid, block, st, p, pl);
    } else if (pl.sym == 2) {
        if (!pl.xp) return -12;
        long nwg2 = (long)pl.n_row_tiles * pl.n_col_tiles * pl.splitk;
        if (nwg2 > 2 * cap) nwg2 = 2 * cap;                                  // two workgroups per CU
        const dim3 grid2((unsigned)nwg2), block2(256);
        if (pl.A == 1) PASE_LAUNCH((conv_x6c_kernel<128, 3, false, true, false, true, true>), grid2, block2, st, p, pl);
        else PASE_LAUNCH((conv_x6c_kernel<192, 2, false, true, false, true, true>), grid2, block2, st, p, pl);
    } else if (pl.sym) {
        if (!pl.xp) return -12;      // the plan (and the weight pack's row tiles) counted on the pre-split activation
        if (pl.A == 1) PASE_LAUNCH((conv_x6c_kernel<128, 3, false, true, false, true>), grid, block, st, p, pl);
        else PASE_LAUNCH((conv_x6c_kernel<192, 2, false, true, false, true>), grid, block, st, p, pl);
    } else if (pl.xp) {
        if (pl.A == 1) PASE_LAUNCH((conv_x6c_kernel<128, 3, false, true>), grid, block, st, p, pl);
        else PASE_LAUNCH((conv_x6c_kernel<192, 2, false, true>), grid, block, st, p, pl);
    } else if (pl.A == 1) PASE_LAUNCH((conv_x6c_kernel<128, 3>), grid, block, st, p, pl);
    else PASE_LAUNCH((conv_x6c_kernel<192, 2>), grid, block, st, p, pl);
    PASE_CHECK_LAUNCH();
    return 0;
}

int pase_x6c_pack_xp(const PaseConvGemm& p, const PaseX6cPlan& pl, hipStream_t st) {
    if (!pl.xp) return -11;
    const dim3 grid((unsigned)((pl.xp_tpad + 255) / 256), (unsigned)(2 * pl.G), (unsigned)p.S);
    PASE_LAUNCH(pack_xcm_kernel, grid, dim3(256), st, p.x, reinterpret_cast<u32x4*>(const_cast<void*>(p.xp6)), p.Cin, p.S,
                p.x_ctot, p.x_coff, p.Tin, pl.xp_tpad, pl.padLp, p.pad_mode, pl.xp_plane, p.in_scale, p.in_shift, p.in_alpha);
    PASE_CHECK_LAUNCH();
    return 0;
}


// ---- weight gradients on the same kernel (TM instantiation) ------------------------------------------------------------
//   dw[m, (ci,kk)] += sum_{s,q} g~[s, m, q] * z~[s, ci, q * stride + kk * tapstep - padL]
// One operand is packed (pack_rows_x6c_kernel: rows x positions, read by the compute waves in fragment order), the other is
// staged by the T-mode loader (columns x positions).  Normally the rows are g's and the columns the (channel, tap) pairs
// of z; a 1x1 layer with more output than input channels (the 256 -> 21 525 heads) runs SWAPPED -- rows = z's channels,
// columns = g's rows, transposed accumulation into dw -- so that the big operand is split exactly once, on the fly,
// and the small one is the pack.
bool pase_x6c_wgrad_plan(const PaseWgrad& w, PaseX6cWgrad& o) {
    if (w.tap_major || (w.tapstep != 1 && w.tapstep != -1) || w.taps < 1 || w.stride < 1) return false;
    const long LIM = 0x7fffffffL;
    if ((long)w.S * w.g_ctot * (long)w.Tg >= LIM || (long)w.S * w.z_ctot * (long)w.Tz >= LIM) return false;
    if (w.Ncols < 8) return false;
    o.swapped = (w.taps == 1 && w.stride == 1 && w.padL == 0 && w.M > w.Cin) ? 1 : 0;
    // layers with taps: rows = (channel, tap) straight out of row-major planes of z~, columns = g's channels (mode 3), unless
    // g has too few channels to fill the 128-column tile or PASE_X6C_WGRAD_MODE=1 asks for the staged-Toeplitz orientation
    // Measured on the PASE+ bs32 step (profiles/gemm_launches_r03.json): mode 3 wins on stride-1 layers with >= 256 output
    // channels (block 5: 1.01 -> 0.93 ms); strided layers lose to the cost of writing the planes (the decoder's stride-10
    // ConvTranspose1d: 1.6 GB of planes for a 1.7 ms launch) or tie
    // (PaseWgrad::x6 bits 4-7 force an orientation for A/B runs and tests; 0 = the routing below)
    const int force_mode = (w.x6 >> 4) & 15;
    bool toep = w.taps > 1 && w.M >= 96 && force_mode == 3;
    // Round 4: the pre-split staged operand (mode 4, "ZP") replaces both the staged-Toeplitz orientation with its on-the-fly
    // conversion (mode 1) and the plane-rows orientation (mode 3) on every layer with taps
    // (... except stride >= 8 unless forced: a phase row then carries only 2-3 taps, 64 staging lanes read ~22 different plane
    //  rows per load instruction and the memory pipeline is request-bound -- the decoder's stride-10 layer 1.81 vs 1.48 ms;
    //  the fp32-staged mode 1 reads 64 ADJACENT samples per instruction there)
    const bool want_zp = force_mode == 4 || (force_mode == 0 && w.stride < 8);
    PaseConvGemm& c = o.pc;
    c = PaseConvGemm{};
    PaseX6cPlan& pl = o.pl;
    pl = PaseX6cPlan{};
    const int QP16 = (w.Ncols + 15) / 16;
    if (toep) {
        o.swapped = 1;
        o.a_src = w.z; o.a_rows = w.Cin * w.taps; o.a_ctot = w.z_ctot; o.a_coff = w.z_coff; o.a_T = w.Tz;
        o.a_sc = w.in_scale; o.a_sh = w.in_shift; o.a_al = w.in_alpha;
        c.x = w.g; c.x_ctot = w.g_ctot; c.x_coff = w.g_coff; c.Tin = w.Tg; c.Cin = w.M;
        c.taps = 1; c.stride = 1; c.tapstep = 1; c.padL = 0; c.pad_mode = PASE_PAD_ZERO;
        c.in_scale = nullptr; c.in_shift = nullptr; c.in_alpha = w.g_alpha;
        c.K = w.M;
        pl.tmode = 3;
        pl.t_taps = w.taps; pl.t_tapstep = w.tapstep; pl.t_padL = w.padL; pl.t_stride = w.stride;
        const int omin = (w.tapstep > 0 ? 0 : -(w.taps - 1)) - w.padL, omax = (w.tapstep > 0 ? w.taps - 1 : 0) - w.padL;
        auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
        pl.t_dmin = fdiv(omin, w.stride);
        pl.t_hh = (fdiv(omax, w.stride) - pl.t_dmin + 1) & ~1;           // even: the parity of a row's shift is the same in
        pl.t_lseg = QP16 * 16 + pl.t_hh;                                   // every sequence segment
        const long body = (long)w.Cin * w.stride * w.S * pl.t_lseg;
        if (body >= (1L << 29)) return false;
        pl.t_plane = 2 * ((body + 4L * (16 + pl.t_hh) + 64 + 15) / 16 * 16);   // two copies (see the A-fragment loads)
    } else if (!o.swapped) {
        o.a_src = w.g; o.a_rows = w.M; o.a_ctot = w.g_ctot; o.a_coff = w.g_coff; o.a_T = w.Tg;
        o.a_sc = nullptr; o.a_sh = nullptr; o.a_al = w.g_alpha;
        c.x = w.z; c.x_ctot = w.z_ctot; c.x_coff = w.z_coff; c.Tin = w.Tz; c.Cin = w.Cin;
        c.taps = w.taps; c.stride = w.stride; c.tapstep = w.tapstep; c.padL = w.padL; c.pad_mode = w.pad_mode;
        c.in_scale = w.in_scale; c.in_shift = w.in_shift; c.in_alpha = w.in_alpha;
        c.K = w.Cin * w.taps;
        pl.tmode = 1;
        if (want_zp) {
            // pre-split staged operand: plane[pz][ci * stride + b][s][i] = piece pz of z~[s][ci][stride * (i + dmin) + b],
            // i in [0, lseg), lseg = 16 QP16 + (shifts of a phase row, rounded up to 8); + one all-ones row for the bias column
            auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
            const int omin = (w.tapstep > 0 ? 0 : -(w.taps - 1)) - w.padL, omax = (w.tapstep > 0 ? w.taps - 1 : 0) - w.padL;
            pl.t_dmin = fdiv(omin, w.stride);
            pl.t_hh = (fdiv(omax, w.stride) - pl.t_dmin + 1 + 7) & ~7;
            pl.t_lseg = QP16 * 16 + pl.t_hh;
            pl.zp_rows = w.Cin * w.stride + (w.dbias ? 1 : 0);
            const long body = (long)pl.zp_rows * w.S * pl.t_lseg;
            if (body < (1L << 30) && (long)w.padL + w.taps < (1L << 20)) {
                pl.zp = 1;
                pl.t_plane = 2 * body;                              // two copies (see the stager); body is a multiple of 8 (lseg is)
                pl.ps_magic = magic_of(w.stride);
                pl.t_stride = w.stride;
                pl.zp_n = w.taps / w.stride;
                pl.zp_rem = w.taps % w.stride;
                pl.rctx_magic = magic_of(pl.zp_n + 1);
                pl.cout_magic = magic_of(pl.zp_n);
            }
        }
    } else {
        o.a_src = w.z; o.a_rows = w.Cin; o.a_ctot = w.z_ctot; o.a_coff = w.z_coff; o.a_T = w.Tz;
        o.a_sc = w.in_scale; o.a_sh = w.in_shift; o.a_al = w.in_alpha;
        c.x = w.g; c.x_ctot = w.g_ctot; c.x_coff = w.g_coff; c.Tin = w.Tg; c.Cin = w.M;
        c.taps = 1; c.stride = 1; c.tapstep = 1; c.padL = 0; c.pad_mode = PASE_PAD_ZERO;
        c.in_scale = nullptr; c.in_shift = nullptr; c.in_alpha = w.g_alpha;
        c.K = w.M;
        pl.tmode = 2;
    }
    // g staged (modes 2 / 3): 8-position chunks start on 16-byte boundaries when the rows do
    pl.t_vec = (pl.tmode >= 2 && w.Tg % 4 == 0 && w.Ncols % 8 == 0 && (long)w.S * QP16 >= 4 &&
                (reinterpret_cast<uintptr_t>(w.g) & 15) == 0 && !(w.x6 & 512) &&
                (long)w.S * w.g_ctot * w.Tg * 4 < (1L << 32)) ? 1 : 0;        // (32-bit byte offsets of the staging loads)
    // the split is paid once per staged element and shared by the row tiles of the workgroup: at most 64 rows would leave
    // half of every MFMA multiplying zeros
    // (round 6 built the 64 x 256 tile for this kernel -- compute waves 2 x 2, 256 staged columns per k-group, for layers of
    //  at most 64 output channels with taps: block 1 of the encoder, 64 x 1280 over 96 x 3200 positions -- correct, and
    //  0.78 ms against 0.64 ms on the exact-fp32 pipe, same box: 16 conversions per staging lane and k-group feed only 24
    //  MFMAs per compute wave.  Not kept.)
    if (o.a_rows <= 64) return false;
    // 1x1 layers: every staged element feeds only four row tiles and there are no taps to share the conversion between --
    // measured slower than the exact-fp32 matrix pipe on every PASE+ 1x1 weight gradient (profiles/gemm_launches_r03.json)
    // (... except the swapped orientation on aligned rows of >= 1024 output channels: the 21 525-channel heads 0.68 -> 0.59 ms,
    //  the QRNN's 1536-channel Linear 0.33 -> 0.27 ms)
    if (w.taps == 1 && !(w.x6 & 256) && !(o.swapped && pl.t_vec && w.M >= 1024)) return false;
    if ((long)c.Cin * c.Tin >= LIM) return false;
    c.S = w.S; c.Ncols = w.Ncols; c.M = o.a_rows;
    c.y = w.dw; c.Tout = w.ldw; c.bias = w.dbias; c.ldw = w.ldw;
    c.epilogue = PASE_EPI_STORE;
    const long Gk = (long)w.S * QP16;
    if (Gk * 16 >= LIM) return false;
    pl.P = QP16; pl.A = 1; pl.G = (int)Gk; pl.CinP = 0x7fffffff;
    // k-groups (16 positions) per stage.  A stage ends in a barrier and the first fragment reads of the next one, ~1.4 k clocks
    // against 0.8 k per step: the pre-split planes' kernel holds TMZ_KGS of them (LDS: 2 x 12 KB per k-group)
    const int kgs_req = (int)((w.x6 >> 12) & 7);           // (A/B runs: fewer k-groups per stage than the buffers hold)
    // round 6: pre-split launches with at least 256 rows of g run the SYMMETRIC form (x6c_wgrad_sym_kernel: 256 x 128 tile, all
    // eight waves multiply, six k-groups per stage); x6 bit 11 keeps them on conv_x6c_kernel<128, 5, true, true> (A/B runs)
    const bool sym = pl.zp && o.a_rows >= 256 && Gk >= 2 * SYM_KGS && !(w.x6 & 2048);
    const int kgs_cap = sym ? SYM_KGS : (pl.zp ? (kgs_req && kgs_req < TMZ_KGS ? kgs_req : TMZ_KGS) : 4);
    pl.KGS = Gk >= kgs_cap ? kgs_cap : (int)Gk;
    const int GS = (pl.G + pl.KGS - 1) / pl.KGS;
    pl.steps_total = GS * pl.KGS;
    pl.WM = sym ? 8 : 4; pl.NBT = 4; pl.BM = sym ? 256 : 128; pl.BN = 128;
    const int ncolw = c.K + ((pl.tmode == 1 && w.dbias) ? 1 : 0);
    pl.n_row_tiles = (o.a_rows + pl.BM - 1) / pl.BM;
    pl.n_col_tiles = (ncolw + 127) / 128;
    pl.seg_magic = magic_of(QP16);
    pl.ncols_magic = magic_of(c.taps);
    pl.p_magic = pl.tmode == 3 ? magic_of(pl.t_taps) : 0;
    const long tiles = (long)pl.n_row_tiles * pl.n_col_tiles;
    // split-K over the persistent grid of 256 workgroups (items dealt round-robin): rounds x (1 / slices + flush), where the
    // flush of a 128 x 128 tile costs about 13k cycles as row-contiguous atomics and about 50k transposed (modes 2 / 3),
    // against 768 cycles per k-group of MFMA work
    long sk = 1;
    if (w.splitk > 0) sk = w.splitk;
    else {
        const double flush = (pl.tmode >= 2 ? 50e3 : 13e3) / (768.0 * (double)pl.G);
        double best = 1e30;
        const long max_sk = GS / 4 < 1 ? 1 : GS / 4;
        for (long k = 1; k <= max_sk && k <= 256; ++k) {
            const long rounds = (tiles * k + 255) / 256;
            const double est = (double)rounds * (1.0 / (double)k + flush);
            if (est < best * 0.98) {
                best = est;
                sk = k;
            }
        }
    }
    if (sk < 1) sk = 1;
    {   // every slice owns at least one stage
        const long g_per = (GS + sk - 1) / sk;
        sk = (GS + g_per - 1) / g_per;
    }
    pl.splitk = (int)sk;
    pl.prio = x6c_prio();
    pl.pack_chunks = (long)pl.n_row_tiles * pl.WM * pl.steps_total * 192;
    pl.prm_n = 0;
    pl.pack_bytes = pl.tmode == 3 ? 3 * pl.t_plane * 2 : pl.pack_chunks * 16;
    if (pl.pack_bytes >= (1L << 32)) return false;      // (32-bit scalar offsets of the fragment loads)
    if (pl.zp) {
        pl.zp_off = pl.pack_bytes;
        pl.pack_bytes += 3 * pl.t_plane * 2;
    }
    return true;
}

int pase_x6c_wgrad_launch(const PaseWgrad& w, const PaseX6cWgrad& o, hipStream_t st) {
    PaseConvGemm c = o.pc;
    c.wx6 = w.gx6;
    const PaseX6cPlan& pl = o.pl;
    if (pl.tmode == 3) {
        const long nb = (pl.t_plane / 8 + 255) / 256;
        PASE_LAUNCH(pack_zplanes_kernel, dim3((unsigned)(nb < 32768 ? nb : 32768)), dim3(256), st, o.a_src,
                    reinterpret_cast<u32x4*>(w.gx6), w.Cin, w.S, o.a_ctot, o.a_coff, o.a_T, w.stride, pl.t_lseg, pl.t_dmin,
                    w.pad_mode, pl.t_plane, o.a_sc, o.a_sh, o.a_al);
    } else {
        const long total = pl.pack_chunks / 3;
        const long nb = (total + 255) / 256;
        PASE_LAUNCH(pack_rows_x6c_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), st, o.a_src,
                    reinterpret_cast<u32x4*>(w.gx6), o.a_rows, w.S, o.a_ctot, o.a_coff, o.a_T, w.Ncols, pl.P, pl.steps_total,
                    total, o.a_sc, o.a_sh, o.a_al);
    }
    PASE_CHECK_LAUNCH();
    if (pl.zp) {
        // one thread per (sequence, channel, 8-position group, phase); the ones row (bias column) is the last plane row
        const long zitems = (long)w.S * (w.Cin + (w.dbias ? 1 : 0)) * (pl.t_lseg / 8) * w.stride;
        const long zblocks = (zitems + 255) / 256;
        PASE_LAUNCH(pack_zph_kernel, dim3((unsigned)(zblocks < 65536 ? zblocks : 65536)), dim3(256), st, w.z,
                    reinterpret_cast<u32x4*>(reinterpret_cast<char*>(w.gx6) + pl.zp_off), w.Cin, w.S, w.z_ctot, w.z_coff, w.Tz,
                    w.stride, pl.t_lseg, pl.t_dmin, w.pad_mode, pl.t_plane, w.in_scale, w.in_shift, w.in_alpha,
                    w.dbias ? 1 : 0);
        PASE_CHECK_LAUNCH();
    }
    long nwg = (long)pl.n_row_tiles * pl.n_col_tiles * pl.splitk;
    const long cap = w.max_wg > 0 ? w.max_wg : x6c_cu_count();
    if (nwg > cap) nwg = cap;
    if (pl.zp && pl.WM == 8) PASE_LAUNCH(x6c_wgrad_sym_kernel, dim3((unsigned)nwg), dim3(NT), st, c, pl);
    else if (pl.zp) PASE_LAUNCH((conv_x6c_kernel<128, TMZ_KGS, true, true>), dim3((unsigned)nwg), dim3(NT), st, c, pl);
    else PASE_LAUNCH((conv_x6c_kernel<128, 4, true>), dim3((unsigned)nwg), dim3(NT), st, c, pl);
    PASE_CHECK_LAUNCH();
    return 0;
}

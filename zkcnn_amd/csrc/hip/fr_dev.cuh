// Device-side BLS12-381 scalar-field arithmetic for gfx950 (K0 in SURVEY.md).
// Replaces mcl's Fr (absent dependency of the reference) on the GPU: 8 x 32-bit limbs held in
// VGPRs, Montgomery form with R = 2^256 -- bit-compatible with the host Fr (4 x u64 limbs).
// gfx950 has no 64x64 multiplier; v_mad_u64_u32 (32x32+64) is the widest integer MAD, so the
// product is built from 32-bit limbs. r = 1 (mod 2^32), so -r^{-1} mod 2^32 = 0xffffffff and the
// Montgomery quotient digit is just the negated low limb.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct __align__(16) fr_t {
    uint32_t v[8];
};

#define FR_MOD_INIT {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u}
// R mod r
#define FR_ONE_INIT {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u}

__device__ __forceinline__ fr_t fr_zero() {
    fr_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = 0;
    return z;
}
__device__ __forceinline__ fr_t fr_one() {
    const uint32_t o[8] = FR_ONE_INIT;
    fr_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = o[i];
    return z;
}
__device__ __forceinline__ bool fr_is_zero(const fr_t &a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc |= a.v[i];
    return acc == 0;
}
__device__ __forceinline__ bool fr_eq(const fr_t &a, const fr_t &b) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc |= a.v[i] ^ b.v[i];
    return acc == 0;
}

// z = t - r if t >= r else t   (extra_carry: a ninth limb of t is set)
__device__ __forceinline__ void fr_cond_sub(fr_t &z, const uint32_t *t, uint32_t extra_carry) {
    const uint32_t m[8] = FR_MOD_INIT;
    uint32_t d[8];
    unsigned bo = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = __builtin_subc(t[i], m[i], bo, &bo);
    const bool use_d = extra_carry || !bo;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = use_d ? d[i] : t[i];
}

__device__ __forceinline__ fr_t fr_add(const fr_t &a, const fr_t &b) {
    uint32_t t[8];
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
    fr_t z;
    fr_cond_sub(z, t, c);                  // r < 2^255, so c is always 0 here; kept for safety
    return z;
}

__device__ __forceinline__ fr_t fr_sub(const fr_t &a, const fr_t &b) {
    const uint32_t m[8] = FR_MOD_INIT;
    uint32_t t[8];
    unsigned bo = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = __builtin_subc(a.v[i], b.v[i], bo, &bo);
    const uint32_t mask = bo ? 0xffffffffu : 0u;
    fr_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = __builtin_addc(t[i], m[i] & mask, c, &c);
    return z;
}

__device__ __forceinline__ fr_t fr_neg(const fr_t &a) {
    const uint32_t m[8] = FR_MOD_INIT;
    const uint32_t mask = fr_is_zero(a) ? 0u : 0xffffffffu;
    fr_t z;
    unsigned bo = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = __builtin_subc(m[i], a.v[i], bo, &bo) & mask;
    return z;
}

// 96-bit column accumulator step: acc (64-bit VGPR pair) += x * y, carries out of the pair counted in ovf.
// v_mad_u64_u32 is the widest integer MAD of gfx950 and returns its carry in VCC. gfx950 needs TWO wait states
// between a VALU write of VCC and a VALU read of it as carry-in (hipcc emits `s_nop 1` in its own add/addc
// chains); inline asm is opaque to the hazard recogniser, so the wait states are part of the string.
// (Pairing two MACs on two SGPR carry registers so that they share wait states measured no faster.)
#define ZK_MAC(acc, ovf, x, y)                                                                            \
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"          \
        : "+v"(acc), "+v"(ovf) : "v"(x), "v"(y) : "vcc")

// same step with the second factor in a scalar register: the modulus limbs are compile-time constants, one s_mov each instead of a
// VGPR held across the whole product (VOP3 reads one SGPR operand per instruction)
#define ZK_MAC_S(acc, ovf, x, y)                                                                          \
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"          \
        : "+v"(acc), "+v"(ovf) : "v"(x), "s"(y) : "vcc")

// Montgomery product by product scanning (finely integrated: multiplication and reduction columns interleaved)
// with ONE 96-bit accumulator: 2 instructions per 32x32 partial product, no carry ripple inside a column.
template <int N, bool INV_IS_MINUS1>
__device__ __forceinline__ void mont_mul_comba(uint32_t *z, const uint32_t *a, const uint32_t *b, const uint32_t *m, uint32_t inv) {
    uint32_t q[N], r[N + 1];
    uint64_t acc = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) ZK_MAC(acc, ovf, a[i], b[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) ZK_MAC_S(acc, ovf, q[i], m[k - i]);
        q[k] = INV_IS_MINUS1 ? 0u - (uint32_t) acc : (uint32_t) acc * inv;
        ZK_MAC_S(acc, ovf, q[k], m[0]);
        acc = (acc >> 32) | ((uint64_t) ovf << 32);
        ovf = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) ZK_MAC(acc, ovf, a[i], b[k - i]);
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) ZK_MAC_S(acc, ovf, q[i], m[k - i]);
        r[k - N] = (uint32_t) acc;
        acc = (acc >> 32) | ((uint64_t) ovf << 32);
        ovf = 0;
    }
    r[N] = (uint32_t) acc;
    uint32_t d[N];
    unsigned bo = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = __builtin_subc(r[i], m[i], bo, &bo);
    const bool use_d = r[N] || !bo;
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = use_d ? d[i] : r[i];
}

// Montgomery product (R = 2^256). r = 1 (mod 2^32): the quotient digit is the negated low word.
__device__ __forceinline__ fr_t fr_mul(const fr_t &a, const fr_t &b) {
    const uint32_t m[8] = FR_MOD_INIT;
    fr_t z;
    mont_mul_comba<8, true>(z.v, a.v, b.v, m, 0xffffffffu);
    return z;
}

__device__ __forceinline__ fr_t fr_sqr(const fr_t &a) { return fr_mul(a, a); }

// a + r * (b - a): fold of one table pair with the round challenge
__device__ __forceinline__ fr_t fr_lerp(const fr_t &a, const fr_t &b, const fr_t &r) {
    return fr_add(a, fr_mul(r, fr_sub(b, a)));
}

// 32-byte element <-> two 16-byte vector memory operations
__device__ __forceinline__ fr_t fr_load(const fr_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 lo = q[0], hi = q[1];
    fr_t z;
    z.v[0] = lo.x; z.v[1] = lo.y; z.v[2] = lo.z; z.v[3] = lo.w;
    z.v[4] = hi.x; z.v[5] = hi.y; z.v[6] = hi.z; z.v[7] = hi.w;
    return z;
}
__device__ __forceinline__ void fr_store(fr_t *p, const fr_t &a) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// wave64 cross-lane move of a whole element
__device__ __forceinline__ fr_t fr_shfl_down(const fr_t &a, int delta) {
    fr_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = (uint32_t) __shfl_down((int) a.v[i], delta, 64);
    return z;
}
__device__ __forceinline__ fr_t fr_shfl_up(const fr_t &a, int delta) {
    fr_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = (uint32_t) __shfl_up((int) a.v[i], delta, 64);
    return z;
}

// sum over the 64 lanes of a wave (result valid in lane 0)
__device__ __forceinline__ fr_t fr_wave_sum(fr_t a) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a = fr_add(a, fr_shfl_down(a, d));
    return a;
}

// Block-wide sum of `K` accumulators; result valid in thread 0. `smem` needs K * (blockDim/64) elements.
template <int K>
__device__ __forceinline__ void fr_block_sum(fr_t (&acc)[K], fr_t *smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        acc[k] = fr_wave_sum(acc[k]);
        if (lane == 0) smem[wave * K + k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nwave; ++w)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = fr_add(acc[k], smem[w * K + k]);
    }
}

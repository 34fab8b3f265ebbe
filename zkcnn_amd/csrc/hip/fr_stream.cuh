// Field primitives of the STREAMING round kernels (round_stream.cuh): what a kernel that is bound by the integer multiplier AND by HBM needs
// on top of fr_dev.cuh.
//   * fr_mul_u: Montgomery product whose second factor is WAVE-UNIFORM (the round challenge) and lives in SGPRs: 8 VGPRs less per product.
//   * the quotient digit's own partial product q[k] * m[0] is not a multiplication: r = 1 (mod 2^32), so m[0] = 1 and the column's low word
//     becomes zero with a carry iff it was not zero already: 120 multiply-accumulate steps per product instead of 128.
//   * fr_acc512: sums of products sum_i a_i b_i are accumulated as 512-bit integers and reduced ONCE (64 multiply-accumulate steps per term
//     instead of 128): the Montgomery reduction is linear, so reducing the sum gives the same field element as summing the reduced products.
//   * fr_sub_lazy: a - b + r in (0, 2r): a difference that only feeds a product needs no conditional correction (products take any operand
//     below 2^256 on one side; the 512-bit accumulator takes anything).
// Everything is exact arithmetic in the BLS12-381 scalar field: results are bit-identical to fr_mul / fr_add / fr_sub chains.
#pragma once
#include "fr_dev.cuh"

// R^2 mod r (R = 2^256): fr_mul(x, FR_R2) = x R
#define FR_R2_INIT {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u}

// acc (96 bits: VGPR pair + ovf) += x, through the multiplier's adder (inline constant 1 as the second factor)
#define ZK_ACC_ADD(acc, ovf, x)                                                                           \
    asm("v_mad_u64_u32 %0, vcc, %2, 1, %0\n\ts_nop 1\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"            \
        : "+v"(acc), "+v"(ovf) : "v"(x) : "vcc")

// z = a * b / R mod r.  B_UNIFORM: b's limbs are wave-uniform (held in SGPRs). a < 2^256, b < r (or the other way round); z canonical.
template <bool B_UNIFORM>
__device__ __forceinline__ void fr_mont_mul_m01(uint32_t *z, const uint32_t *a, const uint32_t *b) {
    const uint32_t m[8] = FR_MOD_INIT;
    uint32_t q[8], r[9];
    uint64_t acc = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            if (B_UNIFORM) ZK_MAC_S(acc, ovf, a[i], b[k - i]);
            else ZK_MAC(acc, ovf, a[i], b[k - i]);
        }
#pragma unroll
        for (int i = 0; i < k; ++i) ZK_MAC_S(acc, ovf, q[i], m[k - i]);
        const uint32_t lo = (uint32_t) acc;
        q[k] = 0u - lo;
        // + q[k] * m[0], m[0] = 1: lo + q[k] = 0 (mod 2^32), carry iff lo != 0
        acc = ((acc >> 32) | ((uint64_t) ovf << 32)) + (lo != 0 ? 1u : 0u);
        ovf = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; ++k) {
#pragma unroll
        for (int i = k - 7; i < 8; ++i) {
            if (B_UNIFORM) ZK_MAC_S(acc, ovf, a[i], b[k - i]);
            else ZK_MAC(acc, ovf, a[i], b[k - i]);
        }
#pragma unroll
        for (int i = k - 7; i < 8; ++i) ZK_MAC_S(acc, ovf, q[i], m[k - i]);
        r[k - 8] = (uint32_t) acc;
        acc = (acc >> 32) | ((uint64_t) ovf << 32);
        ovf = 0;
    }
    r[8] = (uint32_t) acc;
    uint32_t d[8];
    unsigned bo = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = __builtin_subc(r[i], m[i], bo, &bo);
    const bool use_d = r[8] || !bo;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = use_d ? d[i] : r[i];
}

// the round challenge as the kernel holds it: limbs in scalar registers
struct fr_u {
    uint32_t v[8];
};
__device__ __forceinline__ fr_u fr_uniform(const fr_t &r) {
    fr_u u;
#pragma unroll
    for (int i = 0; i < 8; ++i) u.v[i] = __builtin_amdgcn_readfirstlane(r.v[i]);
    return u;
}
__device__ __forceinline__ fr_t fr_mul_u(const fr_t &a, const fr_u &b) {
    fr_t z;
    fr_mont_mul_m01<true>(z.v, a.v, b.v);
    return z;
}
__device__ __forceinline__ fr_t fr_mul2(const fr_t &a, const fr_t &b) {
    fr_t z;
    fr_mont_mul_m01<false>(z.v, a.v, b.v);
    return z;
}

// a - b + r: in (0, 2r) for canonical a, b (no borrow test, no mask)
__device__ __forceinline__ fr_t fr_sub_lazy(const fr_t &a, const fr_t &b) {
    const uint32_t m[8] = FR_MOD_INIT;
    uint32_t t[8];
    unsigned bo = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = __builtin_subc(a.v[i], b.v[i], bo, &bo);
    fr_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = __builtin_addc(t[i], m[i], c, &c);
    return z;
}
// a + r (b - a), r wave-uniform
__device__ __forceinline__ fr_t fr_lerp_u(const fr_t &a, const fr_t &b, const fr_u &r) {
    return fr_add(a, fr_mul_u(fr_sub_lazy(b, a), r));
}

// sum of 512-bit products (up to 2^32 of them)
struct fr_acc512 {
    uint32_t w[17];
};
__device__ __forceinline__ fr_acc512 fr_acc512_zero() {
    fr_acc512 W;
#pragma unroll
    for (int i = 0; i < 17; ++i) W.w[i] = 0;
    return W;
}
// W += a * b (plain integers below 2^256)
__device__ __forceinline__ void fr_acc512_mac(fr_acc512 &W, const fr_t &a, const fr_t &b) {
    uint64_t acc = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        ZK_ACC_ADD(acc, ovf, W.w[k]);
#pragma unroll
        for (int i = (k < 8 ? 0 : k - 7); i <= (k < 8 ? k : 7); ++i) ZK_MAC(acc, ovf, a.v[i], b.v[k - i]);
        W.w[k] = (uint32_t) acc;
        acc = (acc >> 32) | ((uint64_t) ovf << 32);
        ovf = 0;
    }
    // column 15 holds no product; the rest of the carry goes into the 17th limb
    acc += W.w[15];
    W.w[15] = (uint32_t) acc;
    W.w[16] += (uint32_t) (acc >> 32);
}
// the field element sum a_i b_i / R (canonical). The 17th limb is folded first (2^512 = R^2 mod r: a 32 x 256-bit product added at the low end, and
// once more if that sum carries out of 512 bits -- it cannot a second time: the first sum is below 2^512 + 2^287), then ONE Montgomery reduction of
// the 512-bit integer (56 multiply-accumulate steps; the result is below 2^256 + r, i.e. at most three subtractions from canonical).
__device__ __forceinline__ fr_t fr_acc512_reduce(const fr_acc512 &W) {
    const uint32_t m[8] = FR_MOD_INIT, r2[8] = FR_R2_INIT;
    uint32_t T[16];
    const uint32_t top = W.w[16];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { c += (uint64_t) top * r2[i] + W.w[i]; T[i] = (uint32_t) c; c >>= 32; }      // (2^32-1)^2 + 2 (2^32-1) = 2^64 - 1
#pragma unroll
    for (int i = 8; i < 16; ++i) { c += W.w[i]; T[i] = (uint32_t) c; c >>= 32; }
    const bool again = c != 0;
    c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { c += (uint64_t) T[i] + (again ? r2[i] : 0u); T[i] = (uint32_t) c; c >>= 32; }
#pragma unroll
    for (int i = 8; i < 16; ++i) { c += T[i]; T[i] = (uint32_t) c; c >>= 32; }
    uint32_t q[8], r[9];
    uint64_t acc = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ZK_ACC_ADD(acc, ovf, T[k]);
#pragma unroll
        for (int i = 0; i < k; ++i) ZK_MAC_S(acc, ovf, q[i], m[k - i]);
        const uint32_t lo = (uint32_t) acc;
        q[k] = 0u - lo;
        acc = ((acc >> 32) | ((uint64_t) ovf << 32)) + (lo != 0 ? 1u : 0u);
        ovf = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; ++k) {
        ZK_ACC_ADD(acc, ovf, T[k]);
#pragma unroll
        for (int i = k - 7; i < 8; ++i) ZK_MAC_S(acc, ovf, q[i], m[k - i]);
        r[k - 8] = (uint32_t) acc;
        acc = (acc >> 32) | ((uint64_t) ovf << 32);
        ovf = 0;
    }
    r[8] = (uint32_t) acc;
#pragma unroll
    for (int rep = 0; rep < 3; ++rep) {
        uint32_t d[9];
        unsigned bo = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = __builtin_subc(r[i], m[i], bo, &bo);
        d[8] = __builtin_subc(r[8], 0u, bo, &bo);
#pragma unroll
        for (int i = 0; i < 9; ++i) r[i] = bo ? r[i] : d[i];
    }
    fr_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = r[i];
    return z;
}

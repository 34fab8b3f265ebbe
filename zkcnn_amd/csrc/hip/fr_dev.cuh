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

// t >= r ?  (t has 8 limbs)
__device__ __forceinline__ bool fr_ge_mod(const uint32_t *t) {
    const uint32_t m[8] = FR_MOD_INIT;
    // compute t - r and look at the borrow
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t) t[i] - m[i] - borrow;
        borrow = (d >> 32) & 1;
    }
    return borrow == 0;
}

// z = t - r if t >= r else t
__device__ __forceinline__ void fr_cond_sub(fr_t &z, const uint32_t *t, uint32_t extra_carry) {
    const uint32_t m[8] = FR_MOD_INIT;
    uint32_t d[8];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t x = (uint64_t) t[i] - m[i] - borrow;
        d[i] = (uint32_t) x;
        borrow = (x >> 32) & 1;
    }
    const bool use_d = extra_carry || !borrow;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = use_d ? d[i] : t[i];
}

__device__ __forceinline__ fr_t fr_add(const fr_t &a, const fr_t &b) {
    uint32_t t[8];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t) a.v[i] + b.v[i];
        t[i] = (uint32_t) c;
        c >>= 32;
    }
    fr_t z;
    fr_cond_sub(z, t, (uint32_t) c);       // r < 2^255, so c is always 0 here; kept for safety
    return z;
}

__device__ __forceinline__ fr_t fr_sub(const fr_t &a, const fr_t &b) {
    const uint32_t m[8] = FR_MOD_INIT;
    uint32_t t[8];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t x = (uint64_t) a.v[i] - b.v[i] - borrow;
        t[i] = (uint32_t) x;
        borrow = (x >> 32) & 1;
    }
    const uint32_t mask = borrow ? 0xffffffffu : 0u;
    fr_t z;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t) t[i] + (m[i] & mask);
        z.v[i] = (uint32_t) c;
        c >>= 32;
    }
    return z;
}

__device__ __forceinline__ fr_t fr_neg(const fr_t &a) {
    const uint32_t m[8] = FR_MOD_INIT;
    const uint32_t mask = fr_is_zero(a) ? 0u : 0xffffffffu;
    fr_t z;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t x = (uint64_t) m[i] - a.v[i] - borrow;
        z.v[i] = (uint32_t) x & mask;
        borrow = (x >> 32) & 1;
    }
    return z;
}

// Montgomery product, CIOS over 32-bit limbs: 8 x (8 MADs for a*b_i + 7 MADs for the reduction row).
__device__ __forceinline__ fr_t fr_mul(const fr_t &a, const fr_t &b) {
    const uint32_t m[8] = FR_MOD_INIT;
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c += (uint64_t) a.v[j] * b.v[i] + t[j];
            t[j] = (uint32_t) c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t) c;
        t[9] = (uint32_t) (c >> 32);

        const uint32_t q = 0u - t[0];               // q = t0 * (-r^-1) mod 2^32
        // q * r[0] + t[0] = q + t0 = 0 (mod 2^32), carry = (t0 != 0)
        c = (t[0] != 0) ? 1 : 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            c += (uint64_t) q * m[j] + t[j];
            t[j - 1] = (uint32_t) c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t) c;
        t[8] = t[9] + (uint32_t) (c >> 32);
    }
    fr_t z;
    fr_cond_sub(z, t, t[8]);
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

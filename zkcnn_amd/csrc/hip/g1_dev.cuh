// Device-side BLS12-381 base field Fp (381 bits, 12 x 32-bit limbs, Montgomery R = 2^384) and G1
// (y^2 = x^3 + 4) point arithmetic for the Hyrax commitment kernels (K13/K14 in SURVEY.md).
// Replaces the G1 arithmetic of mcl that the absent hyrax-bls12-381 library uses for its Pedersen
// commitments. Limb layout equals the host zkff::Fp (6 x u64) so points cross the C-ABI unchanged.
#pragma once
#include "fr_dev.cuh"

struct __align__(16) fp_t {
    uint32_t v[12];
};

#define FP_MOD_INIT {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, \
                     0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau}
#define FP_ONE_INIT {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, \
                     0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u}
#define FP_INV32 0xfffcfffdu

__device__ __forceinline__ fp_t fp_zero() {
    fp_t z;
#pragma unroll
    for (int i = 0; i < 12; ++i) z.v[i] = 0;
    return z;
}
__device__ __forceinline__ fp_t fp_one() {
    const uint32_t o[12] = FP_ONE_INIT;
    fp_t z;
#pragma unroll
    for (int i = 0; i < 12; ++i) z.v[i] = o[i];
    return z;
}
__device__ __forceinline__ bool fp_is_zero(const fp_t &a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) acc |= a.v[i];
    return acc == 0;
}
__device__ __forceinline__ bool fp_eq(const fp_t &a, const fp_t &b) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) acc |= a.v[i] ^ b.v[i];
    return acc == 0;
}
__device__ __forceinline__ void fp_cond_sub(fp_t &z, const uint32_t *t) {
    const uint32_t m[12] = FP_MOD_INIT;
    uint32_t d[12];
    unsigned bo = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) d[i] = __builtin_subc(t[i], m[i], bo, &bo);
#pragma unroll
    for (int i = 0; i < 12; ++i) z.v[i] = bo ? t[i] : d[i];
}
__device__ __forceinline__ fp_t fp_add(const fp_t &a, const fp_t &b) {
    uint32_t t[12];
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) t[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
    fp_t z;
    fp_cond_sub(z, t);          // p < 2^381: no carry out of the top limb
    return z;
}
__device__ __forceinline__ fp_t fp_sub(const fp_t &a, const fp_t &b) {
    const uint32_t m[12] = FP_MOD_INIT;
    uint32_t t[12];
    unsigned bo = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) t[i] = __builtin_subc(a.v[i], b.v[i], bo, &bo);
    const uint32_t mask = bo ? 0xffffffffu : 0u;
    fp_t z;
#pragma unroll
    for (int i = 0; i < 12; ++i) z.v[i] = __builtin_addc(t[i], m[i] & mask, c, &c);
    return z;
}
__device__ __forceinline__ fp_t fp_neg(const fp_t &a) { return fp_sub(fp_zero(), a); }
__device__ __forceinline__ fp_t fp_dbl(const fp_t &a) { return fp_add(a, a); }

// Montgomery product over 12 limbs: 300 MAC steps of the 96-bit column accumulator (fr_dev.cuh: mont_mul_comba).
// ONE copy of the code per translation unit, reached by a call whose 24 operand limbs and 12 result limbs travel in VGPRs
// (scalar arguments: the AMDGPU calling convention passes a by-value struct of this size on the stack, i.e. through scratch
// memory -- the round-1 kernels had a 144..688-byte private segment for exactly that reason). With inter-procedural register
// allocation the caller keeps its live points in the registers the callee does not touch, so nothing is spilled around the call,
// and a mixed addition is 11 calls + a few hundred add / sub instructions (~7 KB of code) instead of 11 inlined products (~60 KB,
// the size of the instruction cache two CUs share).
struct fp_regs { uint32_t v[12]; };
__device__ __noinline__ fp_regs fp_mul_r(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7,
                                         uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3,
                                         uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7, uint32_t b8, uint32_t b9, uint32_t b10, uint32_t b11) {
    const uint32_t m[12] = FP_MOD_INIT;
    const uint32_t a[12] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11}, b[12] = {b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11};
    fp_regs z;
    mont_mul_comba<12, false>(z.v, a, b, m, FP_INV32);
    return z;
}
__device__ __forceinline__ fp_t fp_mul(const fp_t &a, const fp_t &b) {
    const fp_regs r = fp_mul_r(a.v[0], a.v[1], a.v[2], a.v[3], a.v[4], a.v[5], a.v[6], a.v[7], a.v[8], a.v[9], a.v[10], a.v[11],
                               b.v[0], b.v[1], b.v[2], b.v[3], b.v[4], b.v[5], b.v[6], b.v[7], b.v[8], b.v[9], b.v[10], b.v[11]);
    fp_t z;
#pragma unroll
    for (int i = 0; i < 12; ++i) z.v[i] = r.v[i];
    return z;
}
__device__ __forceinline__ fp_t fp_sqr(const fp_t &a) { return fp_mul(a, a); }
// inlined copy (micro-benchmarks only)
__device__ __forceinline__ fp_t fp_mul_i(const fp_t &a, const fp_t &b) {
    const uint32_t m[12] = FP_MOD_INIT;
    fp_t z;
    mont_mul_comba<12, false>(z.v, a.v, b.v, m, FP_INV32);
    return z;
}

// Inverse (inv(0) = 0) by the binary extended Euclidean algorithm on the Montgomery representation taken as a plain integer x = aR:
// pairs (u, A), (v, C) with A x = u and C x = v (mod p), starting from (x, 1), (p, 0). Every step halves one of u, v -- after subtracting the
// other when both are odd -- and does the same to its coefficient mod p, until one of them is 1: at most ~580 steps of ~190 full-rate integer
// instructions, against the 571 products (2 MACs of the quarter-rate multiplier per limb pair each) of a^(p-2): 0.25 instead of 1.8 ms for a
// lone wave, which is what the table kernels' single inversion per thread costs (k_window_tables, k_digit_affine). The result x^-1 = a^-1 R^-1
// goes back to Montgomery form through one product with R^3.
__device__ __forceinline__ fp_t fp_inv(const fp_t &a) {
    const uint32_t m[12] = FP_MOD_INIT;
    const uint32_t r3[12] = {0xd94ca1e0u, 0xed48ac6bu, 0x03a7adf8u, 0x315f831eu, 0x615e29ddu, 0x9a53352au,
                             0x921e1761u, 0x34c04e5eu, 0x65724728u, 0x2512d435u, 0x91755d4du, 0x0aa63460u};     // 2^1152 mod p
    if (fp_is_zero(a)) return a;
    uint32_t u[12], v[12], A[12], C[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { u[i] = a.v[i]; v[i] = m[i]; A[i] = i == 0 ? 1u : 0u; C[i] = 0u; }
    for (;;) {
        uint32_t ur = u[0] ^ 1u, vr = v[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < 12; ++i) { ur |= u[i]; vr |= v[i]; }
        if (ur == 0 || vr == 0) break;
        const bool u_even = !(u[0] & 1u), v_even = !(v[0] & 1u);
        unsigned bo = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) (void) __builtin_subc(u[i], v[i], bo, &bo);
        const bool both_odd = !u_even && !v_even;
        const bool on_v = u_even ? false : (v_even ? true : bo != 0);       // the pair that is halved in this step
        // (u, A) becomes the pair that is halved; the roles of the two pairs are symmetric
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const uint32_t tu = u[i], tv = v[i], tA = A[i], tC = C[i];
            u[i] = on_v ? tv : tu; v[i] = on_v ? tu : tv;
            A[i] = on_v ? tC : tA; C[i] = on_v ? tA : tC;
        }
        const uint32_t mask = both_odd ? 0xffffffffu : 0u;
        unsigned b1 = 0, b2 = 0, c1 = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) u[i] = __builtin_subc(u[i], v[i] & mask, b1, &b1);         // u >= v here
#pragma unroll
        for (int i = 0; i < 12; ++i) A[i] = __builtin_subc(A[i], C[i] & mask, b2, &b2);
        const uint32_t fix = b2 ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < 12; ++i) A[i] = __builtin_addc(A[i], m[i] & fix, c1, &c1);
        // halve: u is even; A / 2 mod p = (A + (A odd ? p : 0)) / 2, and A + p < 2^382
        const uint32_t odd = (A[0] & 1u) ? 0xffffffffu : 0u;
        unsigned c2 = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) A[i] = __builtin_addc(A[i], m[i] & odd, c2, &c2);
#pragma unroll
        for (int i = 0; i < 11; ++i) { u[i] = (u[i] >> 1) | (u[i + 1] << 31); A[i] = (A[i] >> 1) | (A[i + 1] << 31); }
        u[11] >>= 1; A[11] >>= 1;
    }
    uint32_t ur = u[0] ^ 1u;
#pragma unroll
    for (int i = 1; i < 12; ++i) ur |= u[i];
    fp_t res, k;
#pragma unroll
    for (int i = 0; i < 12; ++i) { res.v[i] = ur == 0 ? A[i] : C[i]; k.v[i] = r3[i]; }
    return fp_mul(res, k);
}

struct g1a_t { fp_t x, y; };                 // affine; (0,0) = infinity
struct g1j_t { fp_t X, Y, Z; };              // Jacobian; Z == 0 = infinity

__device__ __forceinline__ g1j_t g1_inf() {
    g1j_t p;
    p.X = fp_zero(); p.Y = fp_one(); p.Z = fp_zero();
    return p;
}
__device__ __forceinline__ bool g1_is_inf(const g1j_t &p) { return fp_is_zero(p.Z); }
__device__ __forceinline__ bool g1a_is_inf(const g1a_t &p) { return fp_is_zero(p.x) && fp_is_zero(p.y); }

__device__ __forceinline__ g1j_t g1_dbl(const g1j_t &p) {
    if (g1_is_inf(p)) return p;
    fp_t A = fp_sqr(p.X), B = fp_sqr(p.Y), C = fp_sqr(B);
    fp_t t = fp_add(p.X, B);
    fp_t D = fp_sub(fp_sub(fp_sqr(t), A), C);
    D = fp_dbl(D);
    fp_t E = fp_add(fp_dbl(A), A), F = fp_sqr(E);
    g1j_t r;
    r.X = fp_sub(F, fp_dbl(D));
    fp_t C8 = fp_dbl(fp_dbl(fp_dbl(C)));
    r.Y = fp_sub(fp_mul(E, fp_sub(D, r.X)), C8);
    r.Z = fp_dbl(fp_mul(p.Y, p.Z));
    return r;
}

// general Jacobian addition (handles infinity, doubling and inverse points)
__device__ __forceinline__ g1j_t g1_add(const g1j_t &p, const g1j_t &q) {
    if (g1_is_inf(p)) return q;
    if (g1_is_inf(q)) return p;
    fp_t Z1Z1 = fp_sqr(p.Z), Z2Z2 = fp_sqr(q.Z);
    fp_t U1 = fp_mul(p.X, Z2Z2), U2 = fp_mul(q.X, Z1Z1);
    fp_t S1 = fp_mul(fp_mul(p.Y, q.Z), Z2Z2), S2 = fp_mul(fp_mul(q.Y, p.Z), Z1Z1);
    if (fp_eq(U1, U2)) {
        if (fp_eq(S1, S2)) return g1_dbl(p);
        return g1_inf();
    }
    fp_t H = fp_sub(U2, U1), Rr = fp_sub(S2, S1);
    fp_t HH = fp_sqr(H), HHH = fp_mul(H, HH), V = fp_mul(U1, HH);
    g1j_t r;
    r.X = fp_sub(fp_sub(fp_sqr(Rr), HHH), fp_dbl(V));
    r.Y = fp_sub(fp_mul(Rr, fp_sub(V, r.X)), fp_mul(S1, HHH));
    r.Z = fp_mul(fp_mul(p.Z, q.Z), H);
    return r;
}

// Jacobian + affine
__device__ __forceinline__ g1j_t g1_madd(const g1j_t &p, const g1a_t &q) {
    if (g1a_is_inf(q)) return p;
    if (g1_is_inf(p)) {
        g1j_t r;
        r.X = q.x; r.Y = q.y; r.Z = fp_one();
        return r;
    }
    fp_t Z1Z1 = fp_sqr(p.Z);
    fp_t U2 = fp_mul(q.x, Z1Z1), S2 = fp_mul(fp_mul(q.y, p.Z), Z1Z1);
    if (fp_eq(p.X, U2)) {
        if (fp_eq(p.Y, S2)) return g1_dbl(p);
        return g1_inf();
    }
    fp_t H = fp_sub(U2, p.X), Rr = fp_sub(S2, p.Y);
    fp_t HH = fp_sqr(H), HHH = fp_mul(H, HH), V = fp_mul(p.X, HH);
    g1j_t r;
    r.X = fp_sub(fp_sub(fp_sqr(Rr), HHH), fp_dbl(V));
    r.Y = fp_sub(fp_mul(Rr, fp_sub(V, r.X)), fp_mul(p.Y, HHH));
    r.Z = fp_mul(p.Z, H);
    return r;
}

__device__ __forceinline__ g1a_t g1_to_affine(const g1j_t &p) {
    g1a_t a;
    if (g1_is_inf(p)) { a.x = fp_zero(); a.y = fp_zero(); return a; }
    fp_t zi = fp_inv(p.Z), zi2 = fp_sqr(zi);
    a.x = fp_mul(p.X, zi2);
    a.y = fp_mul(fp_mul(p.Y, zi2), zi);
    return a;
}

// Fr in Montgomery form -> canonical integer (one Montgomery product with 1)
__device__ __forceinline__ fr_t fr_to_canonical(const fr_t &a) {
    fr_t one_raw = fr_zero();
    one_raw.v[0] = 1;
    return fr_mul(a, one_raw);
}

// signed form for the MSM: |s| <= (r-1)/2 in limbs 0..7, sign returned separately
__device__ __forceinline__ fr_t fr_signed_magnitude(const fr_t &mont, bool &neg) {
    const uint32_t half[8] = {0x80000000u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u, 0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};
    fr_t c = fr_to_canonical(mont);
    // c > (r-1)/2 ?
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t x = (uint64_t) half[i] - c.v[i] - borrow;
        borrow = (x >> 32) & 1;
    }
    neg = borrow != 0;
    if (neg) {
        const uint32_t m[8] = FR_MOD_INIT;
        uint64_t b2 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t x = (uint64_t) m[i] - c.v[i] - b2;
            c.v[i] = (uint32_t) x;
            b2 = (x >> 32) & 1;
        }
    }
    return c;
}

// Row-cooperative BLS12-381 Fp / G1 arithmetic for gfx950: ONE field element spread over a 16-lane DPP row, limb i in lane i
// (i < 12; lanes 12..15 of a row hold 0), so a wave carries four elements and a Montgomery product is 12 steps of two
// multiply-adds per lane instead of 300 in one lane.
//
// Why (round 6): the opening's multi-scalar multiplications and every point-reduction tree of the commitment (upstream Hyrax is
// absent; call sites reference src/prover.cpp:503-511, src/verifier.cpp:128,360) are chains of DEPENDENT point additions on few
// points. In the one-lane-per-element form (g1_dev.cuh) a lone wave needs ~6 500 issue cycles (2.7 us) per Fp product whatever the
// number of live lanes, so a 64-point tree costs 6 x 16 products x 2.7 us = 0.26 ms with 63 lanes idle at the end. In row form a
// product is ~1 200 issue cycles (0.5 us) and a wave works on four additions at once: the same tree, spread over the 32 rows of a
// 512-thread block, takes ~60 us. Throughput per SIMD is about a third of the one-lane form's (more instructions per product), so
// the streaming kernels (k_msm_codes, k_planes_acc) keep one element per lane and only the reductions use rows.
//
// Cross-lane traffic is DPP only (row_newbcast, row_shl / row_shr by one): no LDS, no readlane. Carries between lanes are resolved
// with a carry-lookahead on the wave's ballot masks (one scalar 64-bit addition): lanes 12..15 of every row generate and propagate
// nothing, so the four rows of a wave never interact.
#pragma once
#include "g1_dev.cuh"

#define FPC_ROW_LIMBS 0x0fff0fff0fff0fffull       // ballot bits of the limb lanes (0..11 of every row)

template <int CTRL>
__device__ __forceinline__ uint32_t fpc_dpp(uint32_t v) {         // out-of-row sources read as 0
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, CTRL, 0xf, 0xf, true);
}
#define FPC_SHL1 0x101        // lane i <- lane i + 1
#define FPC_SHR1 0x111        // lane i <- lane i - 1
#define FPC_BCAST(n) (0x150 + (n))    // row_newbcast: every lane of the row <- lane n

// this lane's limb of a 12-limb constant (0 in the padding lanes)
__device__ __forceinline__ uint32_t fpc_const_limb(const uint32_t (&k)[12]) {
    const uint32_t i = threadIdx.x & 15;
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) r = i == j ? k[j] : r;
    return r;
}
__device__ __forceinline__ uint32_t fpc_mod_limb() { const uint32_t m[12] = FP_MOD_INIT; return fpc_const_limb(m); }
__device__ __forceinline__ uint32_t fpc_one_limb() { const uint32_t o[12] = FP_ONE_INIT; return fpc_const_limb(o); }

// bit `lane` of a wave mask / bit 12 of this lane's row (what leaves limb 11)
__device__ __forceinline__ uint32_t fpc_lane_bit(uint64_t mask) { return (uint32_t) (mask >> (threadIdx.x & 63)) & 1u; }
__device__ __forceinline__ bool fpc_row_out(uint64_t mask) { return ((mask >> ((threadIdx.x & 48) + 12)) & 1u) != 0; }

// carries INTO every lane of a chain in which lane i generates (G) or propagates (P) a carry: the carry chain of the binary
// addition (G | P) + G has exactly that recurrence, and its carry into bit i is sum_i ^ P_i
__device__ __forceinline__ uint64_t fpc_lookahead(uint64_t G, uint64_t P) { return ((G | P) + G) ^ P; }

// lo + (carries rippling up the row), g = lane's own carry out of lo (then lo is not all ones)
__device__ __forceinline__ uint32_t fpc_resolve_add(uint32_t lo, bool g) {
    const uint64_t G = __ballot(g), P = __ballot(lo == 0xffffffffu);
    return lo + fpc_lane_bit(fpc_lookahead(G, P));
}
// s - p if s >= p else s (s < 2p)
__device__ __forceinline__ uint32_t fpc_cond_sub(uint32_t s, uint32_t m) {
    const uint64_t B = __ballot(s < m), E = __ballot(s == m) & FPC_ROW_LIMBS;
    const uint64_t C = fpc_lookahead(B, E);                  // borrows into every lane
    const uint32_t d = s - m - fpc_lane_bit(C);
    return fpc_row_out(C) ? s : d;                           // a borrow out of limb 11: s < p
}
__device__ __forceinline__ uint32_t fpc_add(uint32_t a, uint32_t b, uint32_t m) {
    const uint32_t s = a + b;
    return fpc_cond_sub(fpc_resolve_add(s, s < a), m);       // a + b < 2p < 2^382: nothing leaves limb 11
}
__device__ __forceinline__ uint32_t fpc_sub(uint32_t a, uint32_t b, uint32_t m) {
    const uint64_t B = __ballot(a < b), E = __ballot(a == b) & FPC_ROW_LIMBS;
    const uint64_t C = fpc_lookahead(B, E);
    uint32_t t = a - b - fpc_lane_bit(C);
    t = m ? t : 0u;                                          // (the borrow out of limb 11 lands in lane 12)
    const uint32_t s = t + (fpc_row_out(C) ? m : 0u);        // a < b: + p
    const uint32_t r = fpc_resolve_add(s, s < t);
    return m ? r : 0u;                                       // (... and so does the carry that cancels it)
}
__device__ __forceinline__ uint32_t fpc_dbl(uint32_t a, uint32_t m) { return fpc_add(a, a, m); }
// all twelve limbs zero (the element is canonical, < p); uniform over the row
__device__ __forceinline__ bool fpc_is_zero(uint32_t a) {
    const uint64_t nz = __ballot(a != 0u);
    return ((nz >> (threadIdx.x & 48)) & 0xffffu) == 0;
}

// Montgomery product, R = 2^384 (operand scanning over the row, CIOS): after step j the row holds
// (sum_{k <= j} a b_k 2^(32k) + q_k p 2^(32k)) / 2^(32(j+1)) as T_i + c_i + d_i at weight 2^(32i); a step is
//   w = a_i b_j + T_i + c_i              (<= 2^64 - 1)          c_i <- w >> 32
//   q = (w_0 + d_0) * (-p^-1)  mod 2^32                          (lane 0's, broadcast)
//   u = q p_i + (w mod 2^32) + d_i       (<= 2^64 - 1)          d_i <- u >> 32, T_i <- (u mod 2^32) of lane i + 1
// so the two carry words never leave their lane and the only moves are one broadcast of b_j, one of lane 0's low word and one shift.
// ONE copy of the product per translation unit, reached by a call (three operands in, one result out, all in VGPRs): a point addition is 16 products,
// and inlined they made k_cl_tree 65 KB of code -- the size of the instruction cache two CUs share; with the call it is ~6 KB.
__device__ __noinline__ uint32_t fpc_mul(uint32_t a, uint32_t b, uint32_t m) {
    uint32_t T = 0, c = 0, d = 0;
#define FPC_STEP(j) {                                                                  \
        const uint32_t bj = fpc_dpp<FPC_BCAST(j)>(b);                                  \
        const uint64_t w = (uint64_t) a * bj + T + c;                                  \
        const uint32_t lo1 = (uint32_t) w;                                             \
        c = (uint32_t) (w >> 32);                                                      \
        const uint32_t q = fpc_dpp<FPC_BCAST(0)>(lo1 + d) * FP_INV32;                  \
        const uint64_t u = (uint64_t) q * m + lo1 + d;                                 \
        d = (uint32_t) (u >> 32);                                                      \
        T = fpc_dpp<FPC_SHL1>((uint32_t) u);                                           \
    }
    FPC_STEP(0) FPC_STEP(1) FPC_STEP(2) FPC_STEP(3) FPC_STEP(4) FPC_STEP(5)
    FPC_STEP(6) FPC_STEP(7) FPC_STEP(8) FPC_STEP(9) FPC_STEP(10) FPC_STEP(11)
#undef FPC_STEP
    const uint64_t s = (uint64_t) T + c + d;                 // the value is < 2p: limb 11's sum has no high part
    const uint32_t lo = (uint32_t) s, up = fpc_dpp<FPC_SHR1>((uint32_t) (s >> 32));
    const uint32_t t = lo + up;
    return fpc_cond_sub(fpc_resolve_add(t, t < lo), m);
}
__device__ __forceinline__ uint32_t fpc_sqr(uint32_t a, uint32_t m) { return fpc_mul(a, a, m); }

// ---- points: Jacobian, one coordinate per register; memory layout = g1j_t (limb i of a coordinate is word i) ----
struct g1c_t { uint32_t X, Y, Z; };

__device__ __forceinline__ g1c_t g1c_load(const g1j_t *p) {
    const uint32_t i = threadIdx.x & 15;
    g1c_t r = {0u, 0u, 0u};
    if (i < 12) { r.X = p->X.v[i]; r.Y = p->Y.v[i]; r.Z = p->Z.v[i]; }
    return r;
}
__device__ __forceinline__ void g1c_store(g1j_t *p, const g1c_t &a) {
    const uint32_t i = threadIdx.x & 15;
    if (i < 12) { p->X.v[i] = a.X; p->Y.v[i] = a.Y; p->Z.v[i] = a.Z; }
}
__device__ __forceinline__ g1c_t g1c_inf() { g1c_t r = {0u, fpc_one_limb(), 0u}; return r; }
__device__ __forceinline__ g1c_t g1c_select(bool take_a, const g1c_t &a, const g1c_t &b) {
    g1c_t r = {take_a ? a.X : b.X, take_a ? a.Y : b.Y, take_a ? a.Z : b.Z};
    return r;
}
// doubling, a = 0 (dbl-2009-l): 7 products; infinity (Z = 0) stays infinity
__device__ __forceinline__ g1c_t g1c_dbl(const g1c_t &p, uint32_t m) {
    const uint32_t A = fpc_sqr(p.X, m), B = fpc_sqr(p.Y, m), C = fpc_sqr(B, m);
    uint32_t D = fpc_sqr(fpc_add(p.X, B, m), m);
    D = fpc_dbl(fpc_sub(fpc_sub(D, A, m), C, m), m);
    const uint32_t E = fpc_add(fpc_dbl(A, m), A, m);
    g1c_t r;
    r.X = fpc_sub(fpc_sqr(E, m), fpc_dbl(D, m), m);
    r.Y = fpc_sub(fpc_mul(E, fpc_sub(D, r.X, m), m), fpc_dbl(fpc_dbl(fpc_dbl(C, m), m), m), m);
    r.Z = fpc_dbl(fpc_mul(p.Y, p.Z, m), m);
    return r;
}
// general addition with every special case (infinity, P = Q, P = -Q), the cases chosen per row without divergence: 16 products
__device__ __forceinline__ g1c_t g1c_add(const g1c_t &p, const g1c_t &q, uint32_t m) {
    const bool inf1 = fpc_is_zero(p.Z), inf2 = fpc_is_zero(q.Z);
    const uint32_t Z1Z1 = fpc_sqr(p.Z, m), Z2Z2 = fpc_sqr(q.Z, m);
    const uint32_t U1 = fpc_mul(p.X, Z2Z2, m), U2 = fpc_mul(q.X, Z1Z1, m);
    const uint32_t S1 = fpc_mul(fpc_mul(p.Y, q.Z, m), Z2Z2, m), S2 = fpc_mul(fpc_mul(q.Y, p.Z, m), Z1Z1, m);
    const uint32_t H = fpc_sub(U2, U1, m), R = fpc_sub(S2, S1, m);
    const bool hz = fpc_is_zero(H), rz = fpc_is_zero(R);
    const uint32_t HH = fpc_sqr(H, m), HHH = fpc_mul(H, HH, m), V = fpc_mul(U1, HH, m);
    g1c_t r;
    r.X = fpc_sub(fpc_sub(fpc_sqr(R, m), HHH, m), fpc_dbl(V, m), m);
    r.Y = fpc_sub(fpc_mul(R, fpc_sub(V, r.X, m), m), fpc_mul(S1, HHH, m), m);
    r.Z = fpc_mul(fpc_mul(p.Z, q.Z, m), H, m);
    const bool same = !inf1 && !inf2 && hz && rz;
    if (__any(same)) {                                       // (wave-uniform branch; degenerate bases only)
        const g1c_t dd = g1c_dbl(p, m);
        r = g1c_select(same, dd, r);
    }
    r = g1c_select(!inf1 && !inf2 && hz && !rz, g1c_inf(), r);
    r = g1c_select(inf2, p, r);
    r = g1c_select(inf1, q, r);
    return r;
}

// mixed addition (Jacobian + affine point (qx, qy), not at infinity; madd-2007-bl without the doubling trick: 7 M + 4 S = 11 products), every special case
// chosen per row without divergence
__device__ __forceinline__ g1c_t g1c_madd(const g1c_t &p, uint32_t qx, uint32_t qy, uint32_t m) {
    const bool inf1 = fpc_is_zero(p.Z);
    const uint32_t Z1Z1 = fpc_sqr(p.Z, m);
    const uint32_t U2 = fpc_mul(qx, Z1Z1, m), S2 = fpc_mul(fpc_mul(qy, p.Z, m), Z1Z1, m);
    const uint32_t H = fpc_sub(U2, p.X, m), R = fpc_sub(S2, p.Y, m);
    const bool hz = fpc_is_zero(H), rz = fpc_is_zero(R);
    const uint32_t HH = fpc_sqr(H, m), HHH = fpc_mul(H, HH, m), V = fpc_mul(p.X, HH, m);
    g1c_t r;
    r.X = fpc_sub(fpc_sub(fpc_sqr(R, m), HHH, m), fpc_dbl(V, m), m);
    r.Y = fpc_sub(fpc_mul(R, fpc_sub(V, r.X, m), m), fpc_mul(p.Y, HHH, m), m);
    r.Z = fpc_mul(p.Z, H, m);
    const bool same = !inf1 && hz && rz;
    if (__any(same)) {
        const g1c_t dd = g1c_dbl(p, m);
        r = g1c_select(same, dd, r);
    }
    r = g1c_select(!inf1 && hz && !rz, g1c_inf(), r);
    const g1c_t q = {qx, qy, fpc_one_limb()};
    return g1c_select(inf1, q, r);
}

// Multi-scalar-multiplication kernels of the Hyrax commitment (K13 / K14 in SURVEY.md) for gfx950.
// They replace the Pedersen row commitments and the inner-product-argument cross terms of the absent upstream library
// hyrax-bls12-381 (call sites: reference src/prover.cpp:503-511, src/verifier.cpp:128,360).
//
// Shape of the work: MANY MSMs over ONE generator vector (4096 rows x 4096 generators for vgg11), scalars almost all tiny and
// signed (bits, 9-bit quantised weights), a few rows with wide scalars, and -- in the opening -- two MSMs per round with
// full-width scalars. Tables remove every doubling: D[d][j] = d g_j (first use of a generator set) and, once a set is used
// again, F[w][d][j] = d 2^(8w) g_j, so every non-zero scalar BYTE is one mixed addition.
//
// Pipeline of a commitment:
//   k_scalar_codes   scalar (Montgomery) -> 16-bit code: low byte of |s|, sign, "wide" flag; per-row wide flag      (HBM bound)
//   k_msm_codes      one wave per row chunk; a lane walks ITS non-zero codes only (bit mask built up front), the next table point
//                    is fetched while the current mixed addition runs; every lane leaves one Jacobian partial sum   (integer ALU)
//   k_reduce_rows16  the 64 lane sums of a row chunk: 16 lanes per row add strided, then a 4-level tree
//   rows flagged wide: their windows >= 1 ride along as virtual rows of k_msm_codes (through F's windows, or -- no byte table -- through D with the window
//   sums shifted by k_cl_whorner); the opening's full-width MSMs: msm_cl.cuh (k_bytes_acc / k_planes_acc + row-cooperative trees)
// Mixed additions are written for register pressure: 11 calls of the one shared Fp product (g1_dev.cuh: fp_mul_r), at most six Fp
// values live across a call, exceptional cases (P = +-Q, negligible for random generators) only flagged in the fast variant; the
// host re-runs a flagged batch with the SAFE variant that handles them in place. No scratch memory in any of these kernels.
#pragma once
#include "g1_dev.cuh"
#include "launch.cuh"

#define MSM_WINDOWS 32
#define MSM_PLANES 8
#define MSM_BLOCK 64                 // one wave per block: reduction trees of 6 levels, many blocks per CU
#define MSM_CODE_NEG 0x100u
#define MSM_CODE_WIDE 0x200u
#define MSM_ROW_WIDE 1u              // row flags of k_scalar_codes
#define MSM_ROW_NONBIT 2u
#define MSM_ROW_HIGH 4u              // a scalar of the row reaches beyond window MSM_LOW_WINDOWS (|s| >= 2^64)
#define MSM_LOW_WINDOWS 7u           // windows 1..7 of a fresh generator set are built in order, the rest beside the proof (hyrax.hip: ensure_tables)

// ---- exceptional-case aware in-place point operations on separate coordinate registers ----
// doubling, a = 0 (dbl-2009-l); Z == 0 stays Z == 0
__device__ __forceinline__ void g1_dbl_ip(fp_t &X, fp_t &Y, fp_t &Z) {
    Z = fp_mul(Y, Z);
    Z = fp_dbl(Z);                         // Z3 = 2 Y Z
    fp_t A = fp_sqr(X), B = fp_sqr(Y);
    fp_t C = fp_sqr(B);
    B = fp_add(X, B);
    B = fp_sqr(B);
    B = fp_sub(fp_sub(B, A), C);
    B = fp_dbl(B);                         // D = 2 ((X + B)^2 - A - C)
    A = fp_add(fp_dbl(A), A);              // E = 3 A
    X = fp_sqr(A);
    X = fp_sub(X, fp_dbl(B));              // X3 = E^2 - 2 D
    B = fp_sub(B, X);
    B = fp_mul(A, B);
    C = fp_dbl(fp_dbl(fp_dbl(C)));
    Y = fp_sub(B, C);                      // Y3 = E (D - X3) - 8 C
}

// (X, Y, Z) += (px, py) for a finite accumulator and a finite affine point. px / py are clobbered. Returns false -- and leaves the
// accumulator untouched -- when the two points have equal x (P = +-Q); *same then tells which.
// Order chosen so that at most six field elements are live across any call (7 M + 4 S, madd-2007-bl without the doubling trick).
__device__ __forceinline__ bool g1_madd_ip(fp_t &X, fp_t &Y, fp_t &Z, fp_t &px, fp_t &py, bool *same) {
    fp_t A = fp_sqr(Z);                    // Z1Z1
    px = fp_mul(px, A);                    // U2
    py = fp_mul(py, Z);
    py = fp_mul(py, A);                    // S2
    px = fp_sub(px, X);                    // H
    py = fp_sub(py, Y);                    // R
    if (fp_is_zero(px)) {
        *same = fp_is_zero(py);
        return false;
    }
    Z = fp_mul(Z, px);                     // Z3 = Z1 H
    A = fp_sqr(px);                        // HH
    px = fp_mul(px, A);                    // HHH
    A = fp_mul(X, A);                      // V
    X = fp_sqr(py);
    X = fp_sub(fp_sub(X, px), fp_dbl(A));  // X3 = R^2 - HHH - 2 V
    A = fp_sub(A, X);
    A = fp_mul(py, A);                     // R (V - X3)
    Y = fp_mul(Y, px);                     // Y1 HHH
    Y = fp_sub(A, Y);
    return true;
}

__device__ __forceinline__ void g1a_load(fp_t &x, fp_t &y, const g1a_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5];
    x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w; x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
    x.v[8] = c.x; x.v[9] = c.y; x.v[10] = c.z; x.v[11] = c.w;
    y.v[0] = d.x; y.v[1] = d.y; y.v[2] = d.z; y.v[3] = d.w; y.v[4] = e.x; y.v[5] = e.y; y.v[6] = e.z; y.v[7] = e.w;
    y.v[8] = f.x; y.v[9] = f.y; y.v[10] = f.z; y.v[11] = f.w;
}
__device__ __forceinline__ void fp_store16(fp_t *p, const fp_t &a) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
    q[2] = make_uint4(a.v[8], a.v[9], a.v[10], a.v[11]);
}
__device__ __forceinline__ fp_t fp_load16(const fp_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 a = q[0], b = q[1], c = q[2];
    fp_t z;
    z.v[0] = a.x; z.v[1] = a.y; z.v[2] = a.z; z.v[3] = a.w; z.v[4] = b.x; z.v[5] = b.y; z.v[6] = b.z; z.v[7] = b.w;
    z.v[8] = c.x; z.v[9] = c.y; z.v[10] = c.z; z.v[11] = c.w;
    return z;
}
__device__ __forceinline__ void g1j_store(g1j_t *p, const fp_t &X, const fp_t &Y, const fp_t &Z, bool empty) {
    fp_store16(&p->X, empty ? fp_zero() : X);
    fp_store16(&p->Y, empty ? fp_one() : Y);
    fp_store16(&p->Z, empty ? fp_zero() : Z);
}

// one accumulation step shared by the MSM kernels: acc += +-pt, with the first point, infinity and P = +-Q handled
template <bool SAFE>
__device__ __forceinline__ void g1_accumulate(fp_t &X, fp_t &Y, fp_t &Z, bool &empty, fp_t &px, fp_t &py, bool neg, bool &exc) {
    if (fp_is_zero(px) && fp_is_zero(py)) return;          // (0, 0) = the point at infinity (a generator may be)
    if (neg) py = fp_neg(py);
    if (empty) {
        X = px; Y = py; Z = fp_one();
        empty = false;
        return;
    }
    bool same = false;
    if (!g1_madd_ip(X, Y, Z, px, py, &same)) {
        if (SAFE) {
            if (same) g1_dbl_ip(X, Y, Z);
            else empty = true;             // P + (-P)
        } else exc = true;                 // the host repeats the batch with SAFE = true
    }
}

// ---- the same step on XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; madd-2008-s): 8 M + 2 S = 10 calls of the shared Fp product
// instead of the 11 of the Jacobian mixed addition, for one more coordinate in registers (round 5: the commitment's hot kernel, which runs at the
// multiplier's rate, i.e. 10 / 11 of the time). At most seven field elements live across a call. ----
// Y lives in the lane's 48-byte LDS slot `park` between additions: it is needed twice per addition (R = S2 - Y, Y3 = R (Q - X3) - Y PPP) and never across
// a product otherwise, so three coordinates stay in registers like the Jacobian form's -- the third wave per SIMD depends on it (180 VGPRs with Y in registers).
__device__ __forceinline__ bool g1_madd_xyzz_ip(fp_t &X, fp_t *park, fp_t &ZZ, fp_t &ZZZ, fp_t &px, fp_t &py, bool *same) {
    px = fp_mul(px, ZZ);                   // U2
    py = fp_mul(py, ZZZ);                  // S2
    px = fp_sub(px, X);                    // P
    py = fp_sub(py, fp_load16(park));      // R
    asm volatile("" ::: "memory");
    if (fp_is_zero(px)) {
        *same = fp_is_zero(py);
        return false;
    }
    fp_t A = fp_sqr(px);                   // PP
    ZZ = fp_mul(ZZ, A);
    px = fp_mul(px, A);                    // PPP
    A = fp_mul(X, A);                      // Q
    X = fp_sqr(py);
    X = fp_sub(fp_sub(X, px), fp_dbl(A));  // X3 = R^2 - PPP - 2 Q
    A = fp_sub(A, X);
    A = fp_mul(py, A);                     // R (Q - X3)
    asm volatile("" ::: "memory");         // (a second, late load of Y: merged with the first one it would stay in registers across every product above)
    py = fp_load16(park);
    py = fp_mul(py, px);                   // Y1 PPP
    fp_store16(park, fp_sub(A, py));
    ZZZ = fp_mul(ZZZ, px);                 // (last: nothing else is live by now but X and ZZ)
    return true;
}
// doubling on XYZZ (dbl-2008-s-1, a = 0): 6 M + 3 S; infinity (ZZ = 0) stays infinity
__device__ __forceinline__ void g1_dbl_xyzz_ip(fp_t &X, fp_t *park, fp_t &ZZ, fp_t &ZZZ) {
    fp_t Y = fp_load16(park);
    fp_t U = fp_dbl(Y);
    fp_t V = fp_sqr(U);
    fp_t W = fp_mul(U, V);
    fp_t S = fp_mul(X, V);
    fp_t M = fp_sqr(X);
    M = fp_add(fp_dbl(M), M);              // 3 X^2
    X = fp_sub(fp_sqr(M), fp_dbl(S));
    S = fp_sub(S, X);
    S = fp_mul(M, S);
    Y = fp_mul(W, Y);
    fp_store16(park, fp_sub(S, Y));
    ZZ = fp_mul(V, ZZ);
    ZZZ = fp_mul(W, ZZZ);
}
template <bool SAFE>
__device__ __forceinline__ void g1_accumulate_xyzz(fp_t &X, fp_t *park, fp_t &ZZ, fp_t &ZZZ, bool &empty, fp_t &px, fp_t &py, bool neg, bool &exc) {
    if (fp_is_zero(px) && fp_is_zero(py)) return;          // (0, 0) = the point at infinity
    if (neg) py = fp_neg(py);
    if (empty) {
        X = px; fp_store16(park, py); ZZ = fp_one(); ZZZ = fp_one();
        empty = false;
        return;
    }
    bool same = false;
    if (!g1_madd_xyzz_ip(X, park, ZZ, ZZZ, px, py, &same)) {
        if (SAFE) {
            if (same) g1_dbl_xyzz_ip(X, park, ZZ, ZZZ);
            else empty = true;             // P + (-P)
        } else exc = true;
    }
}
// the Jacobian point of an XYZZ accumulator: (X ZZ, Y ZZZ, ZZ)  [x = X ZZ / ZZ^2, y = Y ZZZ / ZZ^3]
__device__ __forceinline__ void g1j_store_xyzz(g1j_t *p, const fp_t &X, const fp_t *park, const fp_t &ZZ, const fp_t &ZZZ, bool empty) {
    if (empty) { g1j_store(p, X, X, ZZ, true); return; }
    const fp_t jx = fp_mul(X, ZZ), jy = fp_mul(fp_load16(park), ZZZ);
    g1j_store(p, jx, jy, ZZ, false);
}

// ------------------------------------------------------------------------------------------------
// scalar pre-passes
// ------------------------------------------------------------------------------------------------
// codes[row * cols + c]: bits 0..7 low byte of |s| (|s| <= (r-1)/2), MSM_CODE_NEG, MSM_CODE_WIDE when |s| >= 256;
// row_flags[row] = 1 if the row holds a wide scalar (the caller then adds that row's higher windows, see k_scalar_codes_wide)
__device__ __forceinline__ void k_scalar_codes(uint16_t *codes, uint32_t *row_flags, const fr_t *scalars, uint64_t ld, uint32_t cols) {
    const uint32_t row = blockIdx.y;
    bool wide = false, nonbit = false, high = false;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) {
        const fr_t raw = fr_load(scalars + (size_t) row * ld + c);
        uint32_t code = 0;
        if (!fr_is_zero(raw)) {
            bool neg;
            const fr_t s = fr_signed_magnitude(raw, neg);
            uint32_t rest = s.v[0] >> 8;
#pragma unroll
            for (int k = 1; k < 8; ++k) rest |= s.v[k];
            code = (s.v[0] & 0xffu) | (neg ? MSM_CODE_NEG : 0u) | (rest ? MSM_CODE_WIDE : 0u);
            wide |= rest != 0;
            nonbit |= code != 1u;
            high |= (s.v[2] | s.v[3] | s.v[4] | s.v[5] | s.v[6] | s.v[7]) != 0;
        }
        codes[(size_t) row * cols + c] = (uint16_t) code;
    }
    // bit 0: a wide scalar in the row; bit 1: an entry other than 0 / 1 (rows without it are rows of bits: k_bit_masks, k_msm_codes)
    // (one atomic per wave, not per thread: every entry of a weight row sets the second bit)
    const uint32_t f = (__any(wide) ? MSM_ROW_WIDE : 0u) | (__any(nonbit) ? MSM_ROW_NONBIT : 0u) | (__any(high) ? MSM_ROW_HIGH : 0u);
    if (f && (threadIdx.x & 63) == 0) atomicOr(row_flags + row, f);
}

// Rows of bits (the auxiliary witnesses of RELU / pooling layers: a quarter of vgg11's rows): eight columns become ONE table lookup.
// masks[row * 512 + s * 64 + j] = sum_k bit(codes[row][s * 512 + k * 64 + j]) << k  -- the eight columns lane j of the commitment kernel owns inside
// span s -- and T8[mask][s * 64 + j] = the sum of those generators (k_subset_table). The commitment kernel then runs such a row as a row
// of 512 "columns" with byte codes over T8: 512 mixed additions instead of ~2048. cols must be a multiple of 512. grid (2, rows), 256 threads
__device__ __forceinline__ void k_bit_masks(uint16_t *masks, const uint16_t *codes, const uint32_t *row_flags, uint32_t cols) {
    const uint32_t row = blockIdx.y;
    if (row_flags[row] & MSM_ROW_NONBIT) return;
    const uint32_t spans = cols / 512;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < spans * 64; t += gridDim.x * blockDim.x) {
        const uint32_t s = t >> 6, j = t & 63;
        const uint16_t *rc = codes + (size_t) row * cols + (size_t) s * 512 + j;
        uint32_t mask = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) mask |= (uint32_t) (rc[k * 64] & 1u) << k;
        masks[(size_t) row * (spans * 64) + t] = (uint16_t) mask;
    }
}
// T8[mask * n + t] = sum of the generators G[s * 512 + k * 64 + j] over the set bits k of mask, t = s * 64 + j, n = spans * 64; affine,
// (0, 0) for mask 0. One thread per entry: <= 7 additions and its own inversion; built once per generator set. grid (ceil(n / 64), 255)
__device__ __forceinline__ void k_subset_table(g1a_t *T8, const g1a_t *G, uint32_t n) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x, mask = blockIdx.y + 1;
    if (t >= n) return;
    const uint32_t s = t >> 6, j = t & 63;
    fp_t X = fp_zero(), Y = fp_zero(), Z = fp_zero();
    bool empty = true, exc = false;
    for (uint32_t k = 0; k < 8; ++k) {
        if (!((mask >> k) & 1u)) continue;
        fp_t px, py;
        g1a_load(px, py, G + (size_t) s * 512 + k * 64 + j);
        g1_accumulate<true>(X, Y, Z, empty, px, py, false, exc);
    }
    g1j_t acc;
    acc.X = X; acc.Y = Y; acc.Z = empty ? fp_zero() : Z;
    T8[(size_t) mask * n + t] = g1_to_affine(acc);
}

// row_list[0 .. *count) = the rows whose flag is set, ascending (single block; rows <= a few 10^4); *count may exceed `cap`: the
// list then holds the first `cap` of them and the caller takes its slow path
__device__ __forceinline__ void k_compact_flags(uint32_t *row_list, uint32_t *count, const uint32_t *flags, uint32_t rows, uint32_t cap) {
    __shared__ uint32_t s_wave[16], s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (uint32_t r0 = 0; r0 < rows; r0 += 1024) {
        const uint32_t r = r0 + threadIdx.x;
        const bool on = r < rows && (flags[r] & MSM_ROW_WIDE) != 0;
        const unsigned long long b = __ballot(on);
        const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) s_wave[wave] = (uint32_t) __popcll(b);
        __syncthreads();
        uint32_t off = s_base;
        for (uint32_t w = 0; w < wave; ++w) off += s_wave[w];
        const uint32_t pos = off + (uint32_t) __popcll(b & ((1ull << lane) - 1));
        if (on && pos < cap) row_list[pos] = r;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < 16; ++w) t += s_wave[w]; s_base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = s_base;
}

// The higher windows of the rows that hold wide scalars (biases, maxima, the picture: whole rows of 2..4-byte values), as VIRTUAL rows
// of the same hot kernel: codes[(ri * 31 + w - 1) * cols + c] = byte w of |scalars[row_list[ri]][c]| with its sign, w = 1..31.
// k_msm_codes takes virtual row v through window table 1 + v % 31; windows no scalar reaches are rows of zeros that cost nothing.
__device__ __forceinline__ void k_scalar_codes_wide(uint16_t *codes, const fr_t *scalars, uint64_t ld, const uint32_t *row_list,
                                                           const uint32_t *count, uint32_t cols) {
    const uint32_t ri = blockIdx.y;
    if (ri >= *count) return;                           // the grid covers the list's capacity
    const uint32_t row = row_list[ri];
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) {
        const fr_t raw = fr_load(scalars + (size_t) row * ld + c);
        fr_t s = raw;
        bool neg = false;
        if (!fr_is_zero(raw)) s = fr_signed_magnitude(raw, neg);
        const uint32_t sign = neg ? MSM_CODE_NEG : 0u;
#pragma unroll
        for (uint32_t w = 1; w < MSM_WINDOWS; ++w) {
            const uint32_t byte = (s.v[w >> 2] >> ((w & 3) * 8)) & 0xffu;
            codes[((size_t) ri * (MSM_WINDOWS - 1) + (w - 1)) * cols + c] = (uint16_t) (byte ? (byte | sign) : 0u);
        }
    }
}

// mag[ri * cols + c] = |s| of scalars[row][c] as a canonical integer, sign in bit 255 (|s| < 2^254); row = row_map ? row_map[ri] : ri
__device__ __forceinline__ void k_scalar_mags(fr_t *mag, const fr_t *scalars, uint64_t ld, const uint32_t *row_map, uint32_t cols) {
    const uint32_t ri = blockIdx.y, row = row_map ? row_map[ri] : ri;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) {
        const fr_t raw = fr_load(scalars + (size_t) row * ld + c);
        fr_t s = raw;
        if (!fr_is_zero(raw)) {
            bool neg;
            s = fr_signed_magnitude(raw, neg);
            if (neg) s.v[7] |= 0x80000000u;
        }
        fr_store(mag + (size_t) ri * cols + c, s);
    }
}
__device__ __forceinline__ uint32_t mag_byte(const fr_t *mag, size_t i, uint32_t w) {     // w-th byte of the magnitude (sign bit masked)
    const uint32_t limb = reinterpret_cast<const uint32_t *>(mag + i)[w >> 2];
    const uint32_t b = (limb >> ((w & 3) * 8)) & 0xffu;
    return w == 31 ? (b & 0x7fu) : b;
}
__device__ __forceinline__ bool mag_neg(const fr_t *mag, size_t i) { return (reinterpret_cast<const uint32_t *>(mag + i)[7] >> 31) != 0; }

// ------------------------------------------------------------------------------------------------
// the commitment's hot kernel: signed-byte codes through a digit table. grid (chunks, rows), one wave per block; lane l owns columns
// base + 64 i + l, i < cpt <= 64. out[(row * chunks + chunk) * 64 + lane] = the lane's partial sum.
// Rows < n_real use D[d][j] = d g_j. Rows >= n_real are virtual rows (k_scalar_codes_wide): virtual row v uses window 1 + v % 31 of
// the full byte table F[w][d][j] = d 2^(8w) g_j, of which D is window 0 (vstride = 256 m), or -- a fresh generator set, no byte table -- D itself
// (vstride = 0; the caller multiplies the window sums by 2^(8w): k_cl_whorner); only the first 31 * *n_wide of them exist.
// ------------------------------------------------------------------------------------------------
template <bool SAFE>
__device__ __forceinline__ void k_msm_codes(g1j_t *out, uint32_t *exc_flag, const uint16_t *codes, const g1a_t *D, uint32_t m, uint32_t cols,
                                                         uint32_t cpt, uint32_t n_real, const uint32_t *n_wide, uint32_t vstride,
                                                         const uint32_t *row_flags, const uint16_t *masks, const g1a_t *T8) {
    // the virtual rows come FIRST in the grid (their few long chains then run alongside the heavy rows instead of forming a tail);
    // `row` is the logical row: real rows 0 .. n_real - 1, virtual rows behind them
    const uint32_t n_virtual = gridDim.y - n_real;
    const uint32_t row = blockIdx.y < n_virtual ? n_real + blockIdx.y : blockIdx.y - n_virtual, lane = threadIdx.x;
    const uint32_t base = blockIdx.x * (MSM_BLOCK * cpt) + lane;
    const uint16_t *rc = codes + (size_t) row * cols;
    g1j_t *dst = out + ((size_t) row * gridDim.x + blockIdx.x) * MSM_BLOCK + lane;
    // a row of bits (k_bit_masks; offered when the row's blocks split it into whole spans): 8 "columns" per lane, one per 512-column span, whose byte
    // code is the mask of the lane's eight bits in that span, through the subset-sum table T8 laid out like a digit table of cols / 8 columns
    uint32_t base_c = base;
    if (T8 && row < n_real && !(row_flags[row] & MSM_ROW_NONBIT)) {
        m = cols >> 3;
        rc = masks + (size_t) row * m;
        D = T8;
        cpt = m / MSM_BLOCK / gridDim.x;                 // the spans are split over the row's blocks like the columns are
        cols = m;
        base_c = blockIdx.x * (MSM_BLOCK * cpt) + lane;
    }
    if (row >= n_real) {
        const uint32_t v = row - n_real;
        if (v >= *n_wide * (MSM_WINDOWS - 1)) {          // beyond the list: an empty partial sum, nothing to read
            g1j_store(dst, fp_zero(), fp_zero(), fp_zero(), true);
            return;
        }
        D += (size_t) (1 + v % (MSM_WINDOWS - 1)) * vstride;       // vstride = 256 m: the window's own table; 0: every window through D (k_cl_whorner shifts the sums)
    }
    // which of this lane's columns carry a non-zero byte: cpt independent 2-byte loads, then the loop below visits set bits only,
    // so a lane of a half-empty bit row is done after its ~cpt/2 additions instead of idling through cpt iterations
    unsigned long long mask = 0;
    for (uint32_t i = 0; i < cpt; ++i) {
        const uint32_t c = base_c + i * MSM_BLOCK;
        if (c < cols && (rc[c] & 0xffu)) mask |= 1ull << i;
    }
    __shared__ fp_t park[MSM_BLOCK];
    fp_t X = fp_zero(), ZZ = fp_zero(), ZZZ = fp_zero();      // XYZZ accumulator (Y in the lane's LDS slot): 10 products per addition instead of 11
    bool empty = true, exc = false;
    // the code of the NEXT column is fetched one addition ahead (one register); the table point itself is loaded where it is used: its
    // latency (an L2 / Infinity-Cache hit: the digit table of 4096 generators is 100 MB) is a few percent of the ~14 us addition and the
    // other waves of the SIMD run meanwhile -- holding a second point in registers would cost the third wave per SIMD
    uint32_t c_next = 0, code_next = 0;
    bool have = mask != 0;
    if (have) {
        c_next = base_c + (uint32_t) (__ffsll((long long) mask) - 1) * MSM_BLOCK;
        mask &= mask - 1;
        code_next = rc[c_next];
    }
    while (__any(have)) {
        const bool cur = have;
        const uint32_t c = c_next, code = code_next;
        have = mask != 0;
        if (have) {
            c_next = base_c + (uint32_t) (__ffsll((long long) mask) - 1) * MSM_BLOCK;
            mask &= mask - 1;
            code_next = rc[c_next];
        }
        if (cur) {
            fp_t px, py;
            g1a_load(px, py, D + (size_t) (code & 0xffu) * m + c);
            g1_accumulate_xyzz<SAFE>(X, park + lane, ZZ, ZZZ, empty, px, py, (code & MSM_CODE_NEG) != 0, exc);
        }
    }
    g1j_store_xyzz(dst, X, park + lane, ZZ, ZZZ, empty);
    if (!SAFE && exc) *exc_flag = 1;
}

// general Jacobian addition with every special case (infinity, doubling, inverse), operands in memory order
__device__ __forceinline__ g1j_t g1_add_any(const g1j_t &p, const g1j_t &q) {
    if (g1_is_inf(p)) return q;
    if (g1_is_inf(q)) return p;
    g1j_t r;
    fp_t Z1Z1 = fp_sqr(p.Z), Z2Z2 = fp_sqr(q.Z);
    fp_t U1 = fp_mul(p.X, Z2Z2), U2 = fp_mul(q.X, Z1Z1);
    fp_t S1 = fp_mul(p.Y, q.Z), S2 = fp_mul(q.Y, p.Z);
    S1 = fp_mul(S1, Z2Z2);
    S2 = fp_mul(S2, Z1Z1);
    U2 = fp_sub(U2, U1);                   // H
    S2 = fp_sub(S2, S1);                   // R
    if (fp_is_zero(U2)) {
        if (!fp_is_zero(S2)) return g1_inf();
        r = p;
        g1_dbl_ip(r.X, r.Y, r.Z);
        return r;
    }
    r.Z = fp_mul(p.Z, q.Z);
    r.Z = fp_mul(r.Z, U2);
    Z1Z1 = fp_sqr(U2);                     // HH
    U2 = fp_mul(U2, Z1Z1);                 // HHH
    U1 = fp_mul(U1, Z1Z1);                 // V
    r.X = fp_sqr(S2);
    r.X = fp_sub(fp_sub(r.X, U2), fp_dbl(U1));
    U1 = fp_sub(U1, r.X);
    U1 = fp_mul(S2, U1);
    S1 = fp_mul(S1, U2);
    r.Y = fp_sub(U1, S1);
    return r;
}

// out[row] = sum of the n partial points in[row * n ..]: 16 lanes per row (4 rows per wave), lane l adds partials l, l + 16, ...
// one after the other, then a 4-level tree through LDS. One launch instead of log(n) dependent ones: the chain is n / 16 - 1 + 4
// additions deep (7 for the 64 lane sums of a commitment row) and all 64 lanes of a wave work until the tree starts.
// (n_real / n_wide, optional: the rows behind n_real are the virtual rows of a commitment -- 31 per wide row of the device-side list; those beyond the list
//  hold nothing and are skipped: 3 968 row slots for ~24 wide rows of a vgg11 commitment)
__device__ __forceinline__ void k_reduce_rows16(g1j_t *out, const g1j_t *in, uint32_t n, uint32_t rows, uint32_t n_real, const uint32_t *n_wide) {
    __shared__ g1j_t sm[MSM_BLOCK];
    const uint32_t l = threadIdx.x & 15, row = blockIdx.x * 4 + (threadIdx.x >> 4);
    if (n_wide && blockIdx.x * 4 >= n_real + (MSM_WINDOWS - 1) * *n_wide) return;      // (the whole block: uniform)
    g1j_t acc = g1_inf();
    if (row < rows) {
        const g1j_t *src = in + (size_t) row * n;
        if (l < n) acc = src[l];
        for (uint32_t k = l + 16; k < n; k += 16) acc = g1_add_any(acc, src[k]);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 8; s >= 1; s >>= 1) {
        if (l < s) sm[threadIdx.x] = g1_add_any(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (l == 0 && row < rows) out[row] = sm[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// Round 2-5's bit-plane kernel pair, kept for MSMs of a handful of terms (hyrax.hip: msm_windows, cols * windows <= 64); everything larger runs through
// msm_cl.cuh. Window tables T[w][j] = 2^(8w) g_j only. Digit d of window w contributes
// d T[w][j] = sum_k bit_k(d) 2^k T[w][j]: for each of the 8 bit planes one block sums the selected table points, and the row
// result is sum_k 2^k S_k (k_msm_finish). One block = (row, bit plane, column chunk x window group).
// out[(row * 8 + plane) * nparts + part]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void k_msm_planes(g1j_t *out, const fr_t *mag, uint64_t ld, const uint32_t *idx_base, const g1a_t *T,
                                                          uint32_t m, uint32_t cols, uint32_t cpt, uint32_t wsplit, uint32_t w_lo) {
    __shared__ g1j_t sm[MSM_BLOCK];
    const uint32_t plane = blockIdx.y % MSM_PLANES, row = blockIdx.y / MSM_PLANES, lane = threadIdx.x;      // (gridDim.z belongs to the lanes of a fused launch: launch.cuh)
    const uint32_t wg = blockIdx.x % wsplit, chunk = blockIdx.x / wsplit;
    const uint32_t wpg = MSM_WINDOWS / wsplit, w0 = max(wg * wpg, w_lo), w1 = (wg + 1) * wpg;
    const uint32_t *idx = idx_base ? idx_base + (size_t) row * ld : nullptr;
    fp_t X = fp_zero(), Y = fp_zero(), Z = fp_zero();
    bool empty = true, exc = false;
    for (uint32_t i = 0; i < cpt; ++i) {
        const uint32_t c = chunk * (MSM_BLOCK * cpt) + i * MSM_BLOCK + lane;
        if (c >= cols) break;
        const size_t si = (size_t) row * cols + c;
        const uint32_t j = idx ? idx[c] : c;
        const bool neg = mag_neg(mag, si);
        for (uint32_t w = w0; w < w1; ++w) {
            if (!((mag_byte(mag, si, w) >> plane) & 1u)) continue;
            fp_t px, py;
            g1a_load(px, py, T + (size_t) w * m + j);
            g1_accumulate<true>(X, Y, Z, empty, px, py, neg, exc);
        }
    }
    const size_t o = ((size_t) row * MSM_PLANES + plane) * gridDim.x + blockIdx.x;
    if (!__syncthreads_or(!empty)) {
        if (lane == 0) g1j_store(out + o, X, Y, Z, true);
        return;
    }
    g1j_store(&sm[lane], X, Y, Z, empty);
    __syncthreads();
    for (uint32_t s = MSM_BLOCK / 2; s >= 1; s >>= 1) {
        if (lane < s) sm[lane] = g1_add_any(sm[lane], sm[lane + s]);
        __syncthreads();
    }
    if (lane == 0) out[o] = sm[0];
}

// One block per row, one wave per bit plane: wave k sums the partial points of plane k (lane-strided, then a tree through LDS) and
// pre-multiplies by 2^k (k doublings, the waves run concurrently); a final 3-level tree adds the 8 weighted plane sums.
__device__ __forceinline__ void k_msm_finish(g1j_t *outJ, const g1j_t *partials, uint32_t nparts) {
    __shared__ g1j_t sm[MSM_PLANES][32];
    const uint32_t row = blockIdx.x, lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    g1j_t acc = g1_inf();
    const g1j_t *src = partials + ((size_t) row * MSM_PLANES + k) * nparts;
    for (uint32_t p = lane; p < nparts; p += 64) acc = g1_add_any(acc, src[p]);
    const uint32_t live = nparts < 64 ? nparts : 64;          // lanes beyond this hold the point at infinity
    if (lane >= 32) sm[k][lane - 32] = acc;
    __syncthreads();
    if (lane < 32) {
        if (lane + 32 < live) acc = g1_add_any(acc, sm[k][lane]);
    }
    __syncthreads();
    if (lane < 32) sm[k][lane] = acc;
    __syncthreads();
    for (uint32_t s = 16; s >= 1; s >>= 1) {
        if (lane < s && lane + s < live) sm[k][lane] = g1_add_any(sm[k][lane], sm[k][lane + s]);
        __syncthreads();
    }
    if (lane == 0) {
        g1j_t R = sm[k][0];
        for (uint32_t d = 0; d < k; ++d) g1_dbl_ip(R.X, R.Y, R.Z);
        sm[k][0] = R;
    }
    __syncthreads();
    for (uint32_t s = MSM_PLANES / 2; s >= 1; s >>= 1) {
        if (lane == 0 && k < s) sm[k][0] = g1_add_any(sm[k][0], sm[k + s][0]);
        __syncthreads();
    }
    if (threadIdx.x == 0) outJ[row] = sm[0][0];
}

// rows[list[i]] += extra[i], i < min(n, *count)   (list == NULL: rows[i] += extra[i])
__device__ __forceinline__ void k_add_rows(g1j_t *rows, const g1j_t *extra, const uint32_t *list, uint32_t n, const uint32_t *count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (!count || i < *count)) {
        const uint32_t r = list ? list[i] : i;
        rows[r] = g1_add_any(rows[r], extra[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// tables
// ------------------------------------------------------------------------------------------------
// T[w0 - 1][j] is in place (T[0][j] = g_j); fills T[w][j] = 2^(8w) g_j for w0 <= w < w1, converted to affine with
// one field inversion per generator (Montgomery's trick over the Jacobian points of that thread). J, pre: (w1 - w0) * m entries.
__device__ __forceinline__ void k_window_tables(g1a_t *T, g1j_t *J, fp_t *pre, uint32_t m, uint32_t w0, uint32_t w1) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const g1a_t g = T[(size_t) (w0 - 1) * m + j];
    if (g1a_is_inf(g)) {
        for (uint32_t w = w0; w < w1; ++w) T[(size_t) w * m + j] = g;
        return;
    }
    fp_t X = g.x, Y = g.y, Z = fp_one();
    fp_t run = fp_one();
    for (uint32_t w = w0; w < w1; ++w) {
        for (int d = 0; d < 8; ++d) g1_dbl_ip(X, Y, Z);
        g1j_store(J + (size_t) (w - w0) * m + j, X, Y, Z, false);
        pre[(size_t) (w - w0) * m + j] = run;
        run = fp_mul(run, Z);
    }
    fp_t inv = fp_inv(run);
    for (uint32_t w = w1 - 1; w >= w0; --w) {
        const g1j_t Q = J[(size_t) (w - w0) * m + j];
        const fp_t zi = fp_mul(inv, pre[(size_t) (w - w0) * m + j]);
        inv = fp_mul(inv, Q.Z);
        const fp_t zi2 = fp_sqr(zi);
        g1a_t a;
        a.x = fp_mul(Q.X, zi2);
        a.y = fp_mul(fp_mul(Q.Y, zi2), zi);
        T[(size_t) w * m + j] = a;
    }
}

// out[i] = in[i]^-1 (kernel-level entry point of the parity tests)
__device__ __forceinline__ void k_fp_inv(fp_t *out, const fp_t *in, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fp_inv(in[i]);
}

// digit table by levels: D[1] = g, D[d] = 2 D[d/2] (+ g if d is odd); every entry of a level is independent
__device__ __forceinline__ void k_digit_level(g1j_t *J, const g1a_t *G, uint32_t m, uint32_t level) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cnt = 1u << level;
    if (tid >= cnt * m) return;
    const uint32_t j = tid % m, d = cnt + tid / m;
    fp_t gx, gy;
    g1a_load(gx, gy, G + j);
    const bool ginf = fp_is_zero(gx) && fp_is_zero(gy);
    fp_t X, Y, Z;
    bool empty = ginf, exc = false;
    if (level == 0) {
        X = gx; Y = gy; Z = fp_one();
    } else {
        const g1j_t P = J[(size_t) (d >> 1) * m + j];
        X = P.X; Y = P.Y; Z = P.Z;
        empty = fp_is_zero(Z);
        if (!empty) g1_dbl_ip(X, Y, Z);
        empty = empty || fp_is_zero(Z);
        if ((d & 1) && !ginf) g1_accumulate<true>(X, Y, Z, empty, gx, gy, false, exc);
    }
    g1j_store(J + (size_t) d * m + j, X, Y, Z, empty);
}
// Jacobian -> affine for `per` consecutive digits of one generator with one inversion (Montgomery's trick; per divides 256). The inversion is a chain of
// ~480 products, a digit costs 7: per = 16 (592 products, 1 024 waves for 4 096 generators) for a proof on its own; per = 64 does the same table in 0.39 of
// the issue slots with a chain of 928: several proofs in flight (hyrax.hip: digit_affine_per)
__device__ __forceinline__ void k_digit_affine(g1a_t *D, const g1j_t *J, fp_t *pre, uint32_t m, uint32_t per) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (256 / per) * m) return;
    const uint32_t j = tid % m, d0 = (tid / m) * per;
    fp_t run = fp_one();
    for (uint32_t d = d0; d < d0 + per; ++d) {
        if (d == 0) continue;
        const fp_t z = J[(size_t) d * m + j].Z;
        pre[(size_t) d * m + j] = run;
        if (!fp_is_zero(z)) run = fp_mul(run, z);
    }
    fp_t inv = fp_inv(run);
    for (uint32_t d = d0 + per; d-- > d0;) {
        if (d == 0) continue;
        const g1j_t Q = J[(size_t) d * m + j];
        g1a_t a;
        if (fp_is_zero(Q.Z)) { a.x = fp_zero(); a.y = fp_zero(); }
        else {
            const fp_t zi = fp_mul(inv, pre[(size_t) d * m + j]);
            inv = fp_mul(inv, Q.Z);
            const fp_t zi2 = fp_sqr(zi);
            a.x = fp_mul(Q.X, zi2);
            a.y = fp_mul(fp_mul(Q.Y, zi2), zi);
        }
        D[(size_t) d * m + j] = a;
    }
}

// ---- batched Jacobian -> affine: ONE field inversion for all rows (Montgomery's trick), and that single inversion -- a
// chain of ~570 dependent products, 1 ms on one GPU lane -- is done by the host between two small kernels (27 us on a CPU core).
// per thread: running products inside its segment of AFF_SEG points (4 for up to 4096 points: short chains, the scan takes 1024 segments); seg[t] = product of the segment's Z (infinity counts as 1)
__device__ __forceinline__ void k_aff_prefix(fp_t *pre, fp_t *seg, const g1j_t *in, uint32_t n, uint32_t AFF_SEG) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t * AFF_SEG >= n) return;
    fp_t run = fp_one();
    for (uint32_t i = t * AFF_SEG; i < min(n, (t + 1) * AFF_SEG); ++i) {
        pre[i] = run;
        const fp_t z = in[i].Z;
        if (!fp_is_zero(z)) run = fp_mul(run, z);
    }
    seg[t] = run;
}
// single block of N threads: exclusive prefix and suffix products over the nseg <= N segment products (in place), total product to *total.
// N = 1024 (96 KB of LDS: a CU to itself) is the short chain of a proof on its own; with other proofs' kernels refilling every wave slot that frees up, such a
// block waited 6 ms for a CU to drain (profiles/r06_final_kernel_stats_s64.md: 6.4 ms per launch against 0.11 ms alone) -- N = 256 (24 KB) fits into the gaps.
template <int N>
__device__ __forceinline__ void aff_scan_n(fp_t *seg_pre, fp_t *seg_suf, fp_t *total, const fp_t *seg, uint32_t nseg) {
    __shared__ fp_t a[N], b[N];
    const uint32_t t = threadIdx.x;
    a[t] = t < nseg ? seg[t] : fp_one();                      // inclusive prefix
    b[t] = t < nseg ? seg[nseg - 1 - t] : fp_one();           // inclusive prefix of the reversed sequence = suffix
    __syncthreads();
    for (uint32_t d = 1; d < N; d <<= 1) {
        fp_t x, y;
        const bool on = t >= d;
        if (on) { x = fp_mul(a[t], a[t - d]); y = fp_mul(b[t], b[t - d]); }
        __syncthreads();
        if (on) { a[t] = x; b[t] = y; }
        __syncthreads();
    }
    if (t < nseg) {
        seg_pre[t] = t ? a[t - 1] : fp_one();
        seg_suf[t] = (nseg - 1 - t) ? b[nseg - 2 - t] : fp_one();
    }
    if (t == 0) *total = a[nseg - 1];
}
__device__ __forceinline__ void k_aff_scan(fp_t *seg_pre, fp_t *seg_suf, fp_t *total, const fp_t *seg, uint32_t nseg) { aff_scan_n<1024>(seg_pre, seg_suf, total, seg, nseg); }
__device__ __forceinline__ void k_aff_scan256(fp_t *seg_pre, fp_t *seg_suf, fp_t *total, const fp_t *seg, uint32_t nseg) { aff_scan_n<256>(seg_pre, seg_suf, total, seg, nseg); }
__device__ __forceinline__ void k_aff_finish(g1a_t *out, const g1j_t *in, const fp_t *pre, const fp_t *seg_pre, const fp_t *seg_suf,
                                                          const fp_t *total_inv, uint32_t n, uint32_t AFF_SEG) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t * AFF_SEG >= n) return;
    fp_t inv = fp_mul(fp_mul(*total_inv, seg_pre[t]), seg_suf[t]);        // 1 / (product of this segment's Z)
    const uint32_t lo = t * AFF_SEG, hi = min(n, (t + 1) * AFF_SEG);
    for (uint32_t i = hi; i-- > lo;) {
        const g1j_t Q = in[i];
        g1a_t r;
        if (fp_is_zero(Q.Z)) { r.x = fp_zero(); r.y = fp_zero(); }
        else {
            const fp_t zi = fp_mul(inv, pre[i]);
            inv = fp_mul(inv, Q.Z);
            const fp_t zi2 = fp_sqr(zi);
            r.x = fp_mul(Q.X, zi2);
            r.y = fp_mul(fp_mul(Q.Y, zi2), zi);
        }
        out[i] = r;
    }
}

// ---- fixed-base multiples k_i B for ONE base point (the verifier's fresh generators: reference src/verifier.cpp:121-126, gens[i] = G * k_i with k_i from its
// CSPRNG). T[w * 255 + d - 1] = d 256^w B (affine); k: canonical integers, 8 words each. One thread per scalar: 32 table lookups and mixed additions
// (exceptional cases handled in place: the SAFE form), Jacobian result.
__device__ __forceinline__ void k_fixed_base(g1j_t *out, const uint32_t *k, const g1a_t *T, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fp_t X = fp_zero(), Y = fp_zero(), Z = fp_zero();
    bool empty = true, exc = false;
    for (uint32_t w = 0; w < 32; ++w) {
        const uint32_t d = (k[(size_t) i * 8 + (w >> 2)] >> ((w & 3) * 8)) & 0xffu;
        if (!d) continue;
        fp_t px, py;
        g1a_load(px, py, T + (size_t) w * 255 + d - 1);
        g1_accumulate<true>(X, Y, Z, empty, px, py, false, exc);
    }
    g1j_store(out + i, X, Y, Z, empty);
}

__device__ __forceinline__ void k_to_affine(g1a_t *out, const g1j_t *in, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = g1_to_affine(in[i]);
}

// ------------------------------------------------------------------------------------------------
// inner-product argument (scalar side)
// ------------------------------------------------------------------------------------------------
// scalars of the two cross terms of one round, expressed over the ORIGINAL generators:
//   g^(k)_i = sum_{j = i mod len} coef[j] g_j, so  L = <a_lo, g_hi> = sum_{j: (j mod len) >= h} a[(j mod len) - h] coef[j] g_j
__device__ __forceinline__ void k_ipa_scalars(fr_t *sL, uint32_t *idxL, fr_t *sR, uint32_t *idxR, const fr_t *a, const fr_t *coef, uint32_t m,
                              uint32_t len) {      // sL/sR and idxL/idxR are the two rows of one (2 x m/2) batch
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t h = len >> 1, i = j & (len - 1), pos = (j / len) * h + (i & (h - 1));
    const fr_t cj = fr_load(coef + j);
    if (i >= h) {
        fr_store(sL + pos, fr_mul(fr_load(a + i - h), cj));
        idxL[pos] = j;
    } else {
        fr_store(sR + pos, fr_mul(fr_load(a + i + h), cj));
        idxR[pos] = j;
    }
}

// y[0] = <a_lo, b_hi>, y[1] = <a_hi, b_lo>; single block
__device__ __forceinline__ void k_ipa_dots(fr_t *y, const fr_t *a, const fr_t *b, uint32_t h) {
    __shared__ fr_t smem[2 * 256 / 64];
    fr_t acc[2] = {fr_zero(), fr_zero()};
    for (uint32_t i = threadIdx.x; i < h; i += 256) {
        acc[0] = fr_add(acc[0], fr_mul(fr_load(a + i), fr_load(b + i + h)));
        acc[1] = fr_add(acc[1], fr_mul(fr_load(a + i + h), fr_load(b + i)));
    }
    fr_block_sum<2>(acc, smem);
    if (threadIdx.x == 0) { fr_store(y, acc[0]); fr_store(y + 1, acc[1]); }
}

// a' = a_lo + c a_hi, b' = c b_lo + b_hi (in place), coef[j] *= c for generators in the low half
__device__ __forceinline__ void k_ipa_fold(fr_t *a, fr_t *b, fr_t *coef, fr_t c, uint32_t m, uint32_t len) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t h = len >> 1;
    if (j < m && (j & (len - 1)) < h) fr_store(coef + j, fr_mul(fr_load(coef + j), c));
    if (j < h) {
        fr_store(a + j, fr_add(fr_load(a + j), fr_mul(c, fr_load(a + j + h))));
        fr_store(b + j, fr_add(fr_mul(c, fr_load(b + j)), fr_load(b + j + h)));
    }
}

__device__ __forceinline__ void k_fill(fr_t *dst, fr_t v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fr_store(dst + i, v);
}

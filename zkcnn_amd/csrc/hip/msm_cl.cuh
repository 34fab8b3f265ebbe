// Multi-scalar multiplication WITHOUT per-generator digit / byte tables (round 6): what a proof over a FRESH generator set runs
// (reference semantics: the verifier draws new generators for every proof, reference src/verifier.cpp:119-128, and hands them to
// commitInput, src/prover.cpp:503-511), and what the opening's full-width MSMs run whenever no byte table exists.
//
//   k_planes_acc   digit planes (non-adjacent form: digits -1 / 0 / +1) over the window tables T[w][j] = 2^(8w) g_j: every lane of a wave accumulates an EQUAL share of the
//                  wave's selected (generator, window) pairs (the wave compacts them into an LDS list first), mixed additions on an
//                  XYZZ accumulator, one Jacobian partial sum per lane -- no reduction tree inside the streaming kernel
//   k_cl_tree      sums runs of partial points: 512 threads = 32 rows of row-cooperative arithmetic (fpc_dev.cuh), every row adds
//                  its strided share, then a 5-level tree through LDS; ~10 us per level instead of the ~43 us of a one-lane addition
//   k_cl_horner    row result = sum_k 2^k (plane sum k): plane k doubles k times in its own row, then a 3-level tree
//
// Round 5's path for the same job (k_msm_planes + k_msm_finish) ended every 64-lane block in a 6-level one-lane tree (two thirds of
// the kernel's issue cycles, 13 % of the multiplier's rate in a batch) and summed the blocks' results in one 512-thread block per
// row (0.98 ms per call, a chain of 11 one-lane additions and 7 doublings).
#pragma once
#include "fpc_dev.cuh"
#include "msm_kernels.cuh"

#define CL_ROWS 32                  // rows of a k_cl_tree block (512 threads)
#define ACC_MAX_PAIRS 128u          // (columns per lane) x (windows per lane) of k_planes_acc / k_bytes_acc: the wave's list holds 64 x this many entries (32 KB)

// ---- k_planes_acc: grid (chunks * wsplit, rows * MSM_PLANES), one wave per block.
// Block (x, y): row = y / 8, plane = y % 8, window group wg = x % wsplit (windows [w_lo + wg * wpg, min(w_lo + (wg + 1) * wpg, w_hi)),
// wpg = ceil((w_hi - w_lo) / wsplit)), column chunk = x / wsplit (columns chunk * 64 * cpt + i * 64 + lane, i < cpt). cpt * wpg <= ACC_MAX_PAIRS.
// mag: canonical magnitudes, sign in bit 255 (k_scalar_mags), dense rows of `cols`; idx (optional): generator of every column, rows `ld` apart.
// out[((row * 8 + plane) * gridDim.x + x) * 64 + lane] = the lane's partial sum (Jacobian; infinity if it took nothing).
// The magnitude at `p` in NON-ADJACENT FORM (digits -1 / 0 / +1, no two neighbours non-zero: a third of the positions instead of half of the bits, so a
// third fewer additions): with h = x >> 1 and t = x + h, the positive digits are t & (h ^ t) and the negative ones h & (h ^ t). Returns the windows of
// `wmask` whose byte has a digit at bit `plane`; *negw = those whose digit is -1. (|x| < 2^254, so t < 2^255: bit 255 stays free for the sign flag.)
// Bytes below window w_lo are somebody else's (a commitment's wide rows: window 0 went through the digit table as a plain byte) and are cleared BEFORE the
// recoding -- the non-adjacent form of x is not the form of its low byte followed by the form of the rest (0xff = 0x100 - 1). A magnitude below 2^k may have its
// top digit at bit k: callers that bound the windows leave room for it.
__device__ __forceinline__ uint32_t acc_select(const fr_t *p, uint32_t plane, uint32_t wmask, uint32_t w_lo, uint32_t *negw) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 lo = q[0], hi = q[1];
    uint32_t x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w & 0x7fffffffu};
    if (w_lo) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t first = 4u * (uint32_t) k;              // the limb holds windows first .. first + 3
            x[k] = w_lo >= first + 4 ? 0u : (w_lo > first ? x[k] & (0xffffffffu << (8u * (w_lo - first))) : x[k]);
        }
    }
    uint32_t s = 0, n = 0;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t h = (x[k] >> 1) | (k < 7 ? x[k + 1] << 31 : 0u);
        const uint32_t t = __builtin_addc(x[k], h, c, &c);
        const uint32_t d = h ^ t;
        const uint32_t b = d >> plane, e = (h & d) >> plane;      // bits 0, 8, 16, 24 = the plane bit of the limb's four bytes
        s |= ((b & 1u) | ((b >> 7) & 2u) | ((b >> 14) & 4u) | ((b >> 21) & 8u)) << (4 * k);
        n |= ((e & 1u) | ((e >> 7) & 2u) | ((e >> 14) & 4u) | ((e >> 21) & 8u)) << (4 * k);
    }
    *negw = n;
    return s & wmask;
}
__device__ __forceinline__ void k_planes_acc(g1j_t *out, const fr_t *mag, uint64_t ld, const uint32_t *idx_base, const g1a_t *T, uint32_t m, uint32_t cols,
                                             uint32_t cpt, uint32_t wsplit, uint32_t w_lo, uint32_t w_hi) {
    __shared__ uint32_t list[64 * ACC_MAX_PAIRS];
    __shared__ fp_t park[MSM_BLOCK];
    const uint32_t plane = blockIdx.y % MSM_PLANES, row = blockIdx.y / MSM_PLANES, lane = threadIdx.x;
    const uint32_t wg = blockIdx.x % wsplit, chunk = blockIdx.x / wsplit;
    const uint32_t wpg = (w_hi - w_lo + wsplit - 1) / wsplit, w0 = w_lo + wg * wpg, w1 = min(w0 + wpg, w_hi);
    const uint32_t *idx = idx_base ? idx_base + (size_t) row * ld : nullptr;
    const uint32_t wmask = w0 >= w1 ? 0u : (w1 >= 32 ? 0xffffffffu : ((1u << w1) - 1u)) & ~((1u << w0) - 1u);
    const fr_t *mrow = mag + (size_t) row * cols;
    uint32_t count = 0, negw;
    for (uint32_t i = 0; i < cpt; ++i) {
        const uint32_t c = chunk * (MSM_BLOCK * cpt) + i * MSM_BLOCK + lane;
        if (c < cols) count += (uint32_t) __popc(acc_select(mrow + c, plane, wmask, w_lo, &negw));
    }
    // exclusive prefix of the lanes' counts
    uint32_t incl = count;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t) __shfl_up((int) incl, d, 64);
        if ((int) lane >= d) incl += t;
    }
    const uint32_t total = (uint32_t) __shfl((int) incl, 63, 64);
    uint32_t pos = incl - count;
    for (uint32_t i = 0; i < cpt; ++i) {
        const uint32_t c = chunk * (MSM_BLOCK * cpt) + i * MSM_BLOCK + lane;
        if (c >= cols) continue;
        uint32_t s = acc_select(mrow + c, plane, wmask, w_lo, &negw);
        if (!s) continue;
        const uint32_t tag = idx ? idx[c] : c;
        if (mag_neg(mag, (size_t) row * cols + c)) negw = ~negw;
        while (s) {
            const uint32_t w = (uint32_t) __ffs((int) s) - 1u;
            s &= s - 1u;
            list[pos++] = tag | (w << 20) | (((negw >> w) & 1u) << 25);
        }
    }
    __syncthreads();
    fp_t X = fp_zero(), ZZ = fp_zero(), ZZZ = fp_zero();
    bool empty = true, exc = false;
    for (uint32_t k = lane; k < total; k += MSM_BLOCK) {
        const uint32_t e = list[k];
        fp_t px, py;
        g1a_load(px, py, T + (size_t) ((e >> 20) & 31u) * m + (e & 0xfffffu));
        g1_accumulate_xyzz<true>(X, park + lane, ZZ, ZZZ, empty, px, py, ((e >> 25) & 1u) != 0, exc);
    }
    g1j_store_xyzz(out + ((size_t) (row * MSM_PLANES + plane) * gridDim.x + blockIdx.x) * MSM_BLOCK + lane, X, park + lane, ZZ, ZZZ, empty);
}

// ---- k_bytes_acc: the same shape over the BYTE table F[w][d][j] = d 2^(8w) g_j of a generator set that is used again: every non-zero scalar byte is
// one mixed addition and there are no planes. grid (chunks * wsplit, rows), one wave per block; the wave's list holds (column, window) of every
// non-zero byte, every lane takes an equal share. out[(row * gridDim.x + x) * 64 + lane]. The fast variant flags P = +-Q (the host repeats the batch
// with SAFE = true, like k_msm_codes).
// wstride = 256 m: the byte table. wstride = 0: the DIGIT table D[d][j] = d g_j stands in for every window (a fresh generator set whose commitment has
// built it) -- with wsplit = w_hi - w_lo every block sums ONE window, V_w = sum_j byte_w(s_j) g_j, and `by_window` lays the partial sums of a (row, window)
// side by side, out[((row * wsplit + wg) * chunks + chunk) * 64 + lane], for the trees; k_cl_whorner32 weighs the window sums.
__device__ __forceinline__ uint32_t acc_nonzero_bytes(const fr_t *p, uint32_t wmask) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 lo = q[0], hi = q[1];
    const uint32_t L[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w & 0x7fffffffu};      // (bit 255 is the sign)
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t b = L[k] | (L[k] >> 4);
        b |= b >> 2;
        b |= b >> 1;                                  // bits 0, 8, 16, 24: the byte is not zero
        s |= ((b & 1u) | ((b >> 7) & 2u) | ((b >> 14) & 4u) | ((b >> 21) & 8u)) << (4 * k);
    }
    return s & wmask;
}
template <bool SAFE>
__device__ __forceinline__ void k_bytes_acc(g1j_t *out, uint32_t *exc_flag, const fr_t *mag, uint64_t ld, const uint32_t *idx_base, const g1a_t *F, uint32_t m,
                                            uint32_t cols, uint32_t cpt, uint32_t wsplit, uint32_t w_lo, uint32_t w_hi, uint32_t wstride, uint32_t by_window) {
    __shared__ uint32_t list[64 * ACC_MAX_PAIRS];
    __shared__ fp_t park[MSM_BLOCK];
    const uint32_t row = blockIdx.y, lane = threadIdx.x;
    const uint32_t wg = blockIdx.x % wsplit, chunk = blockIdx.x / wsplit;
    const uint32_t wpg = (w_hi - w_lo + wsplit - 1) / wsplit, w0 = w_lo + wg * wpg, w1 = min(w0 + wpg, w_hi);
    const uint32_t *idx = idx_base ? idx_base + (size_t) row * ld : nullptr;
    const uint32_t wmask = w0 >= w1 ? 0u : (w1 >= 32 ? 0xffffffffu : ((1u << w1) - 1u)) & ~((1u << w0) - 1u);
    const fr_t *mrow = mag + (size_t) row * cols;
    const uint32_t c0 = chunk * (MSM_BLOCK * cpt);
    uint32_t count = 0;
    for (uint32_t i = 0; i < cpt; ++i) {
        const uint32_t c = c0 + i * MSM_BLOCK + lane;
        if (c < cols) count += (uint32_t) __popc(acc_nonzero_bytes(mrow + c, wmask));
    }
    uint32_t incl = count;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t) __shfl_up((int) incl, d, 64);
        if ((int) lane >= d) incl += t;
    }
    const uint32_t total = (uint32_t) __shfl((int) incl, 63, 64);
    uint32_t pos = incl - count;
    for (uint32_t i = 0; i < cpt; ++i) {
        const uint32_t c = c0 + i * MSM_BLOCK + lane;
        if (c >= cols) continue;
        uint32_t s = acc_nonzero_bytes(mrow + c, wmask);
        while (s) {
            const uint32_t w = (uint32_t) __ffs((int) s) - 1u;
            s &= s - 1u;
            list[pos++] = ((i * MSM_BLOCK + lane) << 5) | w;             // (column inside the chunk)
        }
    }
    __syncthreads();
    fp_t X = fp_zero(), ZZ = fp_zero(), ZZZ = fp_zero();
    bool empty = true, exc = false;
    for (uint32_t k = lane; k < total; k += MSM_BLOCK) {
        const uint32_t e = list[k], w = e & 31u, c = c0 + (e >> 5);
        const size_t si = (size_t) row * cols + c;
        fp_t px, py;
        g1a_load(px, py, F + (size_t) w * wstride + (size_t) mag_byte(mag, si, w) * m + (idx ? idx[c] : c));
        g1_accumulate_xyzz<SAFE>(X, park + lane, ZZ, ZZZ, empty, px, py, mag_neg(mag, si), exc);
    }
    if (!SAFE && exc) *exc_flag = 1;
    const size_t slot = by_window ? ((size_t) row * wsplit + wg) * (gridDim.x / wsplit) + chunk : (size_t) row * gridDim.x + blockIdx.x;
    g1j_store_xyzz(out + slot * MSM_BLOCK + lane, X, park + lane, ZZ, ZZZ, empty);
}

// sum over the 32 rows of a 512-thread block (valid in row 0): 5 levels through LDS, whole waves drop out as the tree narrows
// (`live`: rows at or beyond it hold the point at infinity -- levels that would only add those are skipped)
__device__ __forceinline__ g1c_t cl_tree32(g1c_t acc, uint32_t (*sm)[3][16], uint32_t m, uint32_t live = CL_ROWS) {
    const uint32_t row = threadIdx.x >> 4, limb = threadIdx.x & 15, wave_row0 = (threadIdx.x >> 6) << 2;
    for (uint32_t s = CL_ROWS / 2; s >= 1; s >>= 1) {
        if (s >= live) continue;
        if (row >= s && row < 2 * s) { sm[row][0][limb] = acc.X; sm[row][1][limb] = acc.Y; sm[row][2][limb] = acc.Z; }
        __syncthreads();
        if (wave_row0 < s) {                          // (rows of such a wave at or beyond s compute on stale slots; their sums are never used)
            const uint32_t r2 = (row + s) & (CL_ROWS - 1);
            const g1c_t q = {sm[r2][0][limb], sm[r2][1][limb], sm[r2][2][limb]};
            acc = g1c_add(acc, q, m);
        }
    }
    return acc;
}

// ---- k_cl_tree: segment y = blockIdx.y holds n_seg points at in[y * n_seg ..]; block x sums its run [x * n_in, min((x + 1) * n_in, n_seg))
// into out[y * gridDim.x + x]. 512 threads.
__device__ __forceinline__ void k_cl_tree(g1j_t *out, const g1j_t *in, uint32_t n_seg, uint32_t n_in) {
    __shared__ uint32_t sm[CL_ROWS][3][16];
    const uint32_t row = threadIdx.x >> 4, wave_row0 = (threadIdx.x >> 6) << 2;
    const uint32_t m = fpc_mod_limb();
    const uint32_t first = blockIdx.x * n_in, n = min(n_in, n_seg - min(n_seg, first));
    const g1j_t *src = in + (size_t) blockIdx.y * n_seg + first;
    g1c_t acc = row < n ? g1c_load(src + row) : g1c_inf();
    for (uint32_t k0 = CL_ROWS; k0 + wave_row0 < n; k0 += CL_ROWS) {      // (rows of the wave past the end add infinity)
        const g1c_t q = k0 + row < n ? g1c_load(src + k0 + row) : g1c_inf();
        acc = g1c_add(acc, q, m);
    }
    acc = cl_tree32(acc, sm, m, n);
    if (row == 0) g1c_store(out + (size_t) blockIdx.y * gridDim.x + blockIdx.x, acc);
}

// ---- k_cl_horner: out[r] = sum_k 2^k in[r * 8 + k], r = blockIdx.x; 128 threads = one row per plane
__device__ __forceinline__ void k_cl_horner(g1j_t *out, const g1j_t *in) {
    __shared__ uint32_t sm[MSM_PLANES][3][16];
    const uint32_t k = threadIdx.x >> 4, limb = threadIdx.x & 15, wave_k0 = (threadIdx.x >> 6) << 2;
    const uint32_t m = fpc_mod_limb();
    g1c_t acc = g1c_load(in + (size_t) blockIdx.x * MSM_PLANES + k);
    for (uint32_t d = 0; d < wave_k0 + 3; ++d) {
        const g1c_t dd = g1c_dbl(acc, m);
        acc = g1c_select(d < k, dd, acc);
    }
    for (uint32_t s = MSM_PLANES / 2; s >= 1; s >>= 1) {
        if (k >= s && k < 2 * s) { sm[k][0][limb] = acc.X; sm[k][1][limb] = acc.Y; sm[k][2][limb] = acc.Z; }
        __syncthreads();
        if (wave_k0 < s) {
            const uint32_t k2 = (k + s) & (MSM_PLANES - 1);
            const g1c_t q = {sm[k2][0][limb], sm[k2][1][limb], sm[k2][2][limb]};
            acc = g1c_add(acc, q, m);
        }
    }
    if (k == 0) g1c_store(out + blockIdx.x, acc);
}

// ---- k_cl_whorner: the commitment's rows with wide scalars when no byte table exists. Window w >= 1 of wide row row_list[ri] was summed
// through the DIGIT table like a row of its own (virtual row ri * 31 + w - 1 behind the `rows` real ones: V_w = sum_j byte_w(s_j) g_j), so
// rows[row] = V_0 + sum_w 2^(8w) V_w: CL row w doubles its point 8 w times (points at infinity -- windows no scalar reaches -- cost nothing,
// and a wave stops as soon as none of its four windows has doublings left), then the 32 rows are summed. 512 threads, block ri < *count.
__device__ __forceinline__ void k_cl_whorner(g1j_t *rows_pts, uint32_t rows, const uint32_t *row_list, const uint32_t *count) {
    __shared__ uint32_t sm[CL_ROWS][3][16];
    const uint32_t ri = blockIdx.x;
    if (ri >= *count) return;
    const uint32_t w = threadIdx.x >> 4, row = row_list[ri];
    const uint32_t m = fpc_mod_limb();
    g1c_t acc = g1c_load(w == 0 ? rows_pts + row : rows_pts + rows + (size_t) ri * (MSM_WINDOWS - 1) + (w - 1));
    const bool live = !fpc_is_zero(acc.Z);
    for (uint32_t d = 0; __any(live && d < 8 * w); ++d) {
        const g1c_t dd = g1c_dbl(acc, m);
        acc = g1c_select(d < 8 * w, dd, acc);
    }
    acc = cl_tree32(acc, sm, m);
    if (w == 0) g1c_store(rows_pts + row, acc);
}

// ---- k_cl_whorner32: out[row] = sum_w 2^(8w) V[row][w] for the 32 window sums of an MSM that went through the digit table (k_bytes_acc, by_window):
// CL row w doubles its point 8 w times, the 32 rows are summed. The longest chain is 248 doublings (~0.7 ms): the price of NOT building the window tables
// 2^(8w) g_j of a generator set that one proof uses (248 doublings for each of its generators) -- paid where throughput counts (several proofs in flight,
// other work fills the GPU); a proof on its own takes the window tables, built beside its sumcheck.
__device__ __forceinline__ void k_cl_whorner32(g1j_t *out, const g1j_t *V) {
    __shared__ uint32_t sm[CL_ROWS][3][16];
    const uint32_t w = threadIdx.x >> 4;
    const uint32_t m = fpc_mod_limb();
    g1c_t acc = g1c_load(V + (size_t) blockIdx.x * MSM_WINDOWS + w);
    const bool live = !fpc_is_zero(acc.Z);
    for (uint32_t d = 0; __any(live && d < 8 * w); ++d) {
        const g1c_t dd = g1c_dbl(acc, m);
        acc = g1c_select(d < 8 * w, dd, acc);
    }
    acc = cl_tree32(acc, sm, m);
    if (w == 0) g1c_store(out + blockIdx.x, acc);
}

// ---- k_cl_blind_rows: rows[i] += k_i H for ONE affine point H and a scalar per row (the blinding terms of a zero-knowledge commitment over a generator
// set without a byte table): double-and-add over the non-adjacent form of |k_i| (k_scalar_mags: canonical magnitude, sign in bit 255), one CL row per
// output row -- 254 doublings + ~85 mixed additions = ~2 700 products of 0.39 us, all rows side by side (4 096 rows = one wave per SIMD: ~1.1 ms; the generic
// plane kernels spent ~16 ms on the same column of scalars, a block with ONE busy lane per (row, plane)). 256 threads = 16 rows per block.
__device__ __forceinline__ void k_cl_blind_rows(g1j_t *rows_pts, const fr_t *mag, const g1a_t *H, uint32_t rows) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, limb = threadIdx.x & 15;
    const uint32_t m = fpc_mod_limb();
    const bool on = i < rows;
    // the row's scalar: every lane of the row reads all eight limbs (a broadcast load), digits of the non-adjacent form as two 256-bit masks
    uint32_t x[8], pos[8], neg[8];
    {
        const uint4 *q = reinterpret_cast<const uint4 *>(mag + (on ? i : 0));
        const uint4 lo = q[0], hi = q[1];
        const uint32_t raw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = on ? raw[k] : 0u;
    }
    const bool kneg = (x[7] >> 31) != 0;
    x[7] &= 0x7fffffffu;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t h = (x[k] >> 1) | (k < 7 ? x[k + 1] << 31 : 0u);
        const uint32_t t = __builtin_addc(x[k], h, c, &c);
        const uint32_t d = h ^ t;
        pos[k] = t & d;
        neg[k] = h & d;
    }
    const uint32_t hx = limb < 12 ? H->x.v[limb] : 0u, hy = limb < 12 ? H->y.v[limb] : 0u;
    const uint32_t hyn = fpc_sub(0u, hy, m);                 // -H
    if (fpc_is_zero(hx) && fpc_is_zero(hy)) return;          // H at infinity (a degenerate generator set): nothing to add -- uniform over the grid
    g1c_t acc = g1c_inf();
    for (int b = 254; b >= 0; --b) {
        acc = g1c_dbl(acc, m);
        uint32_t p = 0, n = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if ((b >> 5) == k) { p = (pos[k] >> (b & 31)) & 1u; n = (neg[k] >> (b & 31)) & 1u; }
        if (__any(p | n)) {
            const g1c_t s = g1c_madd(acc, hx, (n != 0) != kneg ? hyn : hy, m);
            acc = g1c_select((p | n) != 0, s, acc);
        }
    }
    if (__any(on)) {
        const g1c_t cur = on ? g1c_load(rows_pts + i) : g1c_inf();
        const g1c_t r = g1c_add(cur, acc, m);
        if (on) g1c_store(rows_pts + i, r);
    }
}

// ---- kernel-level entry points of the parity tests (zk_k_fpc_ops / zk_k_cl_add): one element / one point addition per row ----
// out[i] = a[i] b[i], out[n + i] = a[i] + b[i], out[2 n + i] = a[i] - b[i]
__device__ __forceinline__ void k_fpc_ops(fp_t *out, const fp_t *a, const fp_t *b, uint32_t n) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l = threadIdx.x & 15;
    const uint32_t m = fpc_mod_limb();
    const bool on = i < n && l < 12;
    const uint32_t x = on ? a[i].v[l] : 0u, y = on ? b[i].v[l] : 0u;
    const uint32_t p = fpc_mul(x, y, m), s = fpc_add(x, y, m), d = fpc_sub(x, y, m);
    if (on) { out[i].v[l] = p; out[(size_t) n + i].v[l] = s; out[2 * (size_t) n + i].v[l] = d; }
}
// out[i] = p[i] + q[i] (Jacobian, every special case), out[n + i] = 2 p[i]
__device__ __forceinline__ void k_cl_add(g1j_t *out, const g1j_t *p, const g1j_t *q, uint32_t n) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t m = fpc_mod_limb();
    const g1c_t a = i < n ? g1c_load(p + i) : g1c_inf(), b = i < n ? g1c_load(q + i) : g1c_inf();
    const g1c_t r = g1c_add(a, b, m), d = g1c_dbl(a, m);
    if (i < n) { g1c_store(out + i, r); g1c_store(out + (size_t) n + i, d); }
}

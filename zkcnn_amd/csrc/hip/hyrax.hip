// libzkcnn_hip.so, part 2: Hyrax commitment of the layer-0 witness on the GPU (K13/K14).
// Replaces hyrax_bls12_381::polyProver of the absent upstream library (call sites: reference
// src/prover.cpp:503-511, src/verifier.cpp:128,360). Protocol: zkcnn_amd/csrc/hyrax-bls12-381/polyCommit.hpp.
//
// MSM structure (many MSMs over ONE generator vector, scalars mostly tiny and signed):
//   * per generator set, window tables T[w][j] = 2^(8w) g_j (affine) are built once, so a 255-bit
//     scalar becomes 32 independent 8-bit digits and no doubling chain is left on the critical path;
//   * scalars are folded to |s| <= (r-1)/2 with a sign, so quantised weights / bits only touch w = 0;
//   * digit d contributes d * T[w][j] = sum_k bit_k(d) 2^k T[w][j]: for each of the 8 bit planes one
//     block sums the selected table points (thread-sequential, then an LDS tree), and the row result
//     is sum_k 2^k S_k (7 doublings). No buckets, no sorting, no atomics on 144-byte points.
#include <algorithm>
#include <cstring>
#include "ctx.hpp"
#include "g1_dev.cuh"
#include "../ff/g1.hpp"

#define MSM_WINDOWS 32
#define MSM_PLANES 8
// threads per MSM block: one wave. The kernels are chains of dependent Fp products, so what matters is how many
// independent chains are resident (2 waves/SIMD at ~200 VGPRs) and how short each is: one wave per block keeps the
// reduction tree at 6 levels and lets 8 blocks share a CU.
#define MSM_BLOCK 64

struct msm_state {
    g1a_t *tables = nullptr;           // [MSM_WINDOWS][m]
    uint64_t m = 0;
    std::vector<uint64_t> gens_host;   // generators the tables were built for
    g1j_t *partials = nullptr; size_t partials_cap = 0;
    g1j_t *rowsJ = nullptr; g1a_t *rowsA = nullptr; size_t rows_cap = 0;
    // inner-product argument state
    int rb = 0, cb = 0;
    uint32_t len = 0;                  // current length of a / b
    fr_t *a = nullptr, *b = nullptr, *coef = nullptr, *Lrow = nullptr;
    fr_t *sL = nullptr, *sR = nullptr; uint32_t *idxL = nullptr, *idxR = nullptr;
    fr_t *d_y = nullptr;               // 2 elements
    void *tbl_scratch = nullptr; size_t tbl_scratch_cap = 0;
    // commitment fast path: digit table D[d][j] = d * g_j (d = 1..255, affine), per-row "has high bytes" flags
    g1a_t *digit = nullptr; uint64_t digit_m = 0; bool digit_ready = false;
    uint32_t *hi_flags = nullptr, *row_list = nullptr; size_t flags_cap = 0;
    g1j_t *tmpJ = nullptr; size_t tmp_cap = 0;
    void *aff_scratch = nullptr; size_t aff_cap = 0;
    // generators seen again: full byte table F[w][d][j] = d * 2^(8w) * g_j (d = 1..255), so that EVERY non-zero scalar byte is
    // one mixed addition and no bit planes / doublings are left (3.2 GB for 4096 generators; built on the second use of a set)
    g1a_t *full = nullptr; uint64_t full_m = 0; bool full_ready = false, full_failed = false;
    uint32_t gens_hits = 0;
    g1j_t *parts2 = nullptr; size_t parts2_cap = 0;
    bool host_rows_valid = false;      // few-row MSMs end with a short sum on the host (like the single inversion of fetch_points)
    zkff::G1 host_rows[8];
};
#define MSM_FULL_MAX_M 16384u

void zk_msm_destroy(zk_ctx *ctx) {
    if (!ctx->msm) return;
    msm_state *s = ctx->msm;
    void *bufs[] = {s->tables, s->partials, s->rowsJ, s->rowsA, s->a, s->b, s->coef, s->Lrow, s->sL, s->idxL, s->d_y, s->tbl_scratch,
                    s->digit, s->hi_flags, s->row_list, s->tmpJ, s->aff_scratch, s->full, s->parts2};
    for (void *p : bufs) if (p) hipFree(p);
    delete s;
    ctx->msm = nullptr;
}

static int32_t regrow(zk_ctx *ctx, void **p, size_t *cap, size_t bytes) {
    if (*cap >= bytes) return ZK_OK;
    if (*p) { ZK_HIP(hipStreamSynchronize(ctx->stream)); ZK_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
    ZK_HIP(hipMalloc(p, bytes));
    *cap = bytes;
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// T[0][j] = g_j is already in place; fills T[w][j] = 2^(8w) g_j for w >= 1, converted to affine with
// one field inversion per generator (Montgomery's trick over the 31 Jacobian points of that thread).
__global__ void k_window_tables(g1a_t *T, g1j_t *J, fp_t *pre, uint32_t m) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const g1a_t g = T[j];
    if (g1a_is_inf(g)) {
        for (int w = 1; w < MSM_WINDOWS; ++w) T[(size_t) w * m + j] = g;
        return;
    }
    g1j_t P;
    P.X = g.x; P.Y = g.y; P.Z = fp_one();
    fp_t run = fp_one();
    for (int w = 1; w < MSM_WINDOWS; ++w) {
        for (int d = 0; d < 8; ++d) P = g1_dbl(P);
        J[(size_t) (w - 1) * m + j] = P;
        pre[(size_t) (w - 1) * m + j] = run;
        run = fp_mul(run, P.Z);
    }
    fp_t inv = fp_inv(run);
    for (int w = MSM_WINDOWS - 1; w >= 1; --w) {
        const g1j_t Q = J[(size_t) (w - 1) * m + j];
        const fp_t zi = fp_mul(inv, pre[(size_t) (w - 1) * m + j]);
        inv = fp_mul(inv, Q.Z);
        const fp_t zi2 = fp_sqr(zi);
        g1a_t a;
        a.x = fp_mul(Q.X, zi2);
        a.y = fp_mul(fp_mul(Q.Y, zi2), zi);
        T[(size_t) w * m + j] = a;
    }
}

// One block = (row, bit plane, column chunk x window group). Each thread walks `cpt` columns
// (stride MSM_BLOCK: coalesced scalar and table reads), adds the selected table points, then the block
// tree-reduces through LDS. out[(row * 8 + plane) * nparts + part]
__global__ void __launch_bounds__(MSM_BLOCK) k_msm_planes(g1j_t *out, const fr_t *scalars, uint64_t ld, const uint32_t *idx_base,
                                                     const g1a_t *T, uint32_t m, uint32_t cols, uint32_t cpt, uint32_t wsplit,
                                                     const uint32_t *row_map, uint32_t w_lo) {
    __shared__ g1j_t sm[MSM_BLOCK];
    const uint32_t plane = blockIdx.y, row = row_map ? row_map[blockIdx.z] : blockIdx.z;
    const uint32_t wg = blockIdx.x % wsplit, chunk = blockIdx.x / wsplit;
    const uint32_t wpg = MSM_WINDOWS / wsplit, w0 = max(wg * wpg, w_lo);
    const uint32_t *idx = idx_base ? idx_base + (size_t) row * ld : nullptr;     // index rows are laid out like the scalar rows
    g1j_t acc = g1_inf();
    for (uint32_t i = 0; i < cpt; ++i) {
        const uint32_t c = chunk * (MSM_BLOCK * cpt) + i * MSM_BLOCK + threadIdx.x;
        if (c >= cols) break;
        bool neg;
        const fr_t s = fr_signed_magnitude(fr_load(scalars + (size_t) row * ld + c), neg);
        const uint32_t j = idx ? idx[c] : c;
        for (uint32_t w = w0; w < (wg + 1) * wpg; ++w) {
            const uint32_t byte = (s.v[w >> 2] >> ((w & 3) * 8)) & 0xffu;
            if ((byte >> plane) & 1u) {
                g1a_t pt = T[(size_t) w * m + j];
                if (neg) pt.y = fp_neg(pt.y);
                acc = g1_madd(acc, pt);
            }
        }
    }
    // blocks that selected no point at all (high windows of small scalars) skip the reduction tree
    if (!__syncthreads_or(!g1_is_inf(acc))) {
        if (threadIdx.x == 0) out[((size_t) blockIdx.z * MSM_PLANES + plane) * gridDim.x + blockIdx.x] = acc;
        return;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = MSM_BLOCK / 2; s >= 1; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[((size_t) blockIdx.z * MSM_PLANES + plane) * gridDim.x + blockIdx.x] = sm[0];
}

// ---- commitment fast path -------------------------------------------------------------------------
// digit table by levels: D[1] = g, D[d] = 2 D[d/2] (+ g if d is odd); every entry of a level is independent
__global__ void k_digit_level(g1j_t *J, const g1a_t *G, uint32_t m, uint32_t level) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cnt = 1u << level;
    if (tid >= cnt * m) return;
    const uint32_t j = tid % m, d = cnt + tid / m;
    const g1a_t g = G[j];
    g1j_t P;
    if (level == 0) {
        if (g1a_is_inf(g)) P = g1_inf();
        else { P.X = g.x; P.Y = g.y; P.Z = fp_one(); }
    } else {
        P = g1_dbl(J[(size_t) (d >> 1) * m + j]);
        if (d & 1) P = g1_madd(P, g);
    }
    J[(size_t) d * m + j] = P;
}
// Jacobian -> affine for 16 consecutive digits of one generator with one inversion (Montgomery's trick)
__global__ void k_digit_affine(g1a_t *D, const g1j_t *J, fp_t *pre, uint32_t m) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= 16 * m) return;
    const uint32_t j = tid % m, d0 = (tid / m) * 16;
    fp_t run = fp_one();
    for (uint32_t d = d0; d < d0 + 16; ++d) {
        if (d == 0) continue;
        const fp_t z = J[(size_t) d * m + j].Z;
        pre[(size_t) d * m + j] = run;
        if (!fp_is_zero(z)) run = fp_mul(run, z);
    }
    fp_t inv = fp_inv(run);
    for (uint32_t d = d0 + 16; d-- > d0;) {
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

// Row commitment when (almost) every scalar is a signed byte: ONE mixed addition per non-zero scalar, taken
// from the digit table. Scalars with higher bytes only contribute their low byte here and flag the row; the
// caller adds their remaining windows with k_msm_planes(w_lo = 1). out[row * gridDim.x + chunk]
__global__ void __launch_bounds__(MSM_BLOCK) k_msm_digit(g1j_t *out, uint32_t *hi_flags, const fr_t *scalars, uint64_t ld,
                                                    const g1a_t *D, uint32_t m, uint32_t cols, uint32_t cpt) {
    __shared__ g1j_t sm[MSM_BLOCK];
    const uint32_t row = blockIdx.y, chunk = blockIdx.x;
    g1j_t acc = g1_inf();
    bool hi = false;
    for (uint32_t i = 0; i < cpt; ++i) {
        const uint32_t c = chunk * (MSM_BLOCK * cpt) + i * MSM_BLOCK + threadIdx.x;
        if (c >= cols) break;
        const fr_t raw = fr_load(scalars + (size_t) row * ld + c);
        if (fr_is_zero(raw)) continue;
        bool neg;
        const fr_t s = fr_signed_magnitude(raw, neg);
        uint32_t rest = s.v[0] >> 8;
#pragma unroll
        for (int k = 1; k < 8; ++k) rest |= s.v[k];
        hi |= rest != 0;
        const uint32_t d = s.v[0] & 0xffu;
        if (d) {
            g1a_t pt = D[(size_t) d * m + c];
            if (neg) pt.y = fp_neg(pt.y);
            acc = g1_madd(acc, pt);
        }
    }
    if (hi) hi_flags[row] = 1;
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = MSM_BLOCK / 2; s >= 1; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t) row * gridDim.x + chunk] = sm[0];
}
// rows[r] = sum of its `nparts` chunk sums
__global__ void k_sum_parts(g1j_t *rows, const g1j_t *parts, uint32_t nparts, uint32_t n) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    g1j_t acc = parts[(size_t) r * nparts];
    for (uint32_t p = 1; p < nparts; ++p) acc = g1_add(acc, parts[(size_t) r * nparts + p]);
    rows[r] = acc;
}
// rows[list[i]] += extra[i]
__global__ void k_add_rows(g1j_t *rows, const g1j_t *extra, const uint32_t *list, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[list[i]] = g1_add(rows[list[i]], extra[i]);
}

// ---- byte-table path: every non-zero byte of every scalar is ONE mixed addition from F[w][d][j] ----
// grid (chunks * wsplit, rows). A thread walks `cpt` columns and the windows of its group; out[row * gridDim.x + part].
__global__ void __launch_bounds__(MSM_BLOCK) k_msm_bytes(g1j_t *out, const fr_t *scalars, uint64_t ld, const uint32_t *idx_base,
                                                         const g1a_t *F, uint32_t m, uint32_t cols, uint32_t cpt, uint32_t wsplit) {
    __shared__ g1j_t sm[MSM_BLOCK];
    const uint32_t row = blockIdx.y;
    const uint32_t wg = blockIdx.x % wsplit, chunk = blockIdx.x / wsplit;
    const uint32_t wpg = MSM_WINDOWS / wsplit, w0 = wg * wpg, w1 = w0 + wpg;
    const uint32_t *idx = idx_base ? idx_base + (size_t) row * ld : nullptr;
    g1j_t acc = g1_inf();
    for (uint32_t i = 0; i < cpt; ++i) {
        const uint32_t c = chunk * (MSM_BLOCK * cpt) + i * MSM_BLOCK + threadIdx.x;
        if (c >= cols) break;
        const fr_t raw = fr_load(scalars + (size_t) row * ld + c);
        if (fr_is_zero(raw)) continue;
        bool neg;
        const fr_t s = fr_signed_magnitude(raw, neg);
        const uint32_t j = idx ? idx[c] : c;
        int top = 7;                                  // highest non-zero limb: small scalars leave after their last byte
        while (top > 0 && s.v[top] == 0) --top;
        const uint32_t wend = min(w1, (uint32_t) (4 * top + 4));
        for (uint32_t w = w0; w < wend; ++w) {
            const uint32_t byte = (s.v[w >> 2] >> ((w & 3) * 8)) & 0xffu;
            if (byte) {
                g1a_t pt = F[((size_t) w * 256 + byte) * m + j];
                if (neg) pt.y = fp_neg(pt.y);
                acc = g1_madd_i(acc, pt);
            }
        }
    }
    if (!__syncthreads_or(!g1_is_inf(acc))) {
        if (threadIdx.x == 0) out[(size_t) row * gridDim.x + blockIdx.x] = acc;
        return;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = MSM_BLOCK / 2; s >= 1; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] = g1_add_i(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t) row * gridDim.x + blockIdx.x] = sm[0];
}
// out[row * gridDim.x + b] = sum of in[row * nin + 64 b .. 64 b + 63]  (one tree level of width 64 per launch)
__global__ void __launch_bounds__(MSM_BLOCK) k_tree_reduce(g1j_t *out, const g1j_t *in, uint32_t nin) {
    __shared__ g1j_t sm[MSM_BLOCK];
    const uint32_t row = blockIdx.y, p = blockIdx.x * MSM_BLOCK + threadIdx.x;
    sm[threadIdx.x] = p < nin ? in[(size_t) row * nin + p] : g1_inf();
    __syncthreads();
    for (uint32_t s = MSM_BLOCK / 2; s >= 1; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] = g1_add_i(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t) row * gridDim.x + blockIdx.x] = sm[0];
}

// One block per row, one wave per bit plane: wave k sums the partial points of plane k (lane-strided, then a
// tree through LDS) and pre-multiplies by 2^k (k doublings, the waves run concurrently); a final 3-level tree
// adds the 8 weighted plane sums. Sequential depth: ceil(nparts/64) + 6 + 7 + 3 point operations.
__global__ void __launch_bounds__(512) k_msm_finish(g1j_t *outJ, const g1j_t *partials, uint32_t nparts) {
    __shared__ g1j_t sm[MSM_PLANES][32];
    const uint32_t row = blockIdx.x, lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    g1j_t acc = g1_inf();
    const g1j_t *src = partials + ((size_t) row * MSM_PLANES + k) * nparts;
    for (uint32_t p = lane; p < nparts; p += 64) acc = g1_add(acc, src[p]);
    const uint32_t live = nparts < 64 ? nparts : 64;          // lanes beyond this hold the point at infinity
    if (lane >= 32) sm[k][lane - 32] = acc;
    __syncthreads();
    if (lane < 32) {
        if (lane + 32 < live) acc = g1_add(acc, sm[k][lane]);
    }
    __syncthreads();
    if (lane < 32) sm[k][lane] = acc;
    __syncthreads();
    for (uint32_t s = 16; s >= 1; s >>= 1) {
        if (lane < s && lane + s < live) sm[k][lane] = g1_add(sm[k][lane], sm[k][lane + s]);
        __syncthreads();
    }
    if (lane == 0) {
        g1j_t R = sm[k][0];
        for (uint32_t d = 0; d < k; ++d) R = g1_dbl(R);
        sm[k][0] = R;
    }
    __syncthreads();
    for (uint32_t s = MSM_PLANES / 2; s >= 1; s >>= 1) {
        if (lane == 0 && k < s) sm[k][0] = g1_add(sm[k][0], sm[k + s][0]);
        __syncthreads();
    }
    if (threadIdx.x == 0) outJ[row] = sm[0][0];
}

// ---- batched Jacobian -> affine: ONE field inversion for all rows (Montgomery's trick), and that single inversion -- a
// chain of ~570 dependent products, 1 ms on one GPU lane -- is done by the host between two small kernels (27 us on a CPU core).
#define AFF_SEG 16
// per thread: running products inside its segment of AFF_SEG points; seg[t] = product of the segment's Z (infinity counts as 1)
__global__ void k_aff_prefix(fp_t *pre, fp_t *seg, const g1j_t *in, uint32_t n) {
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
// single block: exclusive prefix and suffix products over the nseg segment products (in place), total product to *total
__global__ void __launch_bounds__(1024) k_aff_scan(fp_t *seg_pre, fp_t *seg_suf, fp_t *total, const fp_t *seg, uint32_t nseg) {
    __shared__ fp_t a[1024], b[1024];
    const uint32_t t = threadIdx.x;
    a[t] = t < nseg ? seg[t] : fp_one();                      // inclusive prefix
    b[t] = t < nseg ? seg[nseg - 1 - t] : fp_one();           // inclusive prefix of the reversed sequence = suffix
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
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
__global__ void k_aff_finish(g1a_t *out, const g1j_t *in, const fp_t *pre, const fp_t *seg_pre, const fp_t *seg_suf, const fp_t *total_inv,
                             uint32_t n) {
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

__global__ void k_to_affine(g1a_t *out, const g1j_t *in, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = g1_to_affine(in[i]);
}

// scalars of the two cross terms of one inner-product round, expressed over the ORIGINAL generators:
//   g^(k)_i = sum_{j = i mod len} coef[j] g_j, so  L = <a_lo, g_hi> = sum_{j: (j mod len) >= h} a[(j mod len) - h] coef[j] g_j
__global__ void k_ipa_scalars(fr_t *sL, uint32_t *idxL, fr_t *sR, uint32_t *idxR, const fr_t *a, const fr_t *coef, uint32_t m,
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
__global__ void __launch_bounds__(256) k_ipa_dots(fr_t *y, const fr_t *a, const fr_t *b, uint32_t h) {
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
__global__ void k_ipa_fold(fr_t *a, fr_t *b, fr_t *coef, fr_t c, uint32_t m, uint32_t len) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t h = len >> 1;
    if (j < m && (j & (len - 1)) < h) fr_store(coef + j, fr_mul(fr_load(coef + j), c));
    if (j < h) {
        fr_store(a + j, fr_add(fr_load(a + j), fr_mul(c, fr_load(a + j + h))));
        fr_store(b + j, fr_add(fr_mul(c, fr_load(b + j)), fr_load(b + j + h)));
    }
}

__global__ void k_fill(fr_t *dst, fr_t v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fr_store(dst + i, v);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int32_t ensure_state(zk_ctx *ctx) {
    if (!ctx->msm) ctx->msm = new msm_state();
    return ZK_OK;
}

// window tables for `m` affine generators (host pointer, C-ABI layout); cached while the generators stay the same
static int32_t ensure_tables(zk_ctx *ctx, const uint64_t *gens, uint64_t m) {
    msm_state *s = ctx->msm;
    if (s->m == m && s->gens_host.size() == m * 12 && std::memcmp(s->gens_host.data(), gens, m * 96) == 0) {
        ++s->gens_hits;
        return ZK_OK;
    }
    s->gens_hits = 0;
    s->full_ready = false;
    if (s->m != m) {
        if (s->tables) { ZK_HIP(hipStreamSynchronize(ctx->stream)); ZK_HIP(hipFree(s->tables)); s->tables = nullptr; }
        ZK_HIP(hipMalloc((void **) &s->tables, (size_t) MSM_WINDOWS * m * sizeof(g1a_t)));
        s->m = m;
    }
    s->gens_host.assign(gens, gens + m * 12);
    s->digit_ready = false;
    ZK_HIP(hipMemcpyAsync(s->tables, gens, m * sizeof(g1a_t), hipMemcpyHostToDevice, ctx->stream));
    const size_t need = (size_t) (MSM_WINDOWS - 1) * m * (sizeof(g1j_t) + sizeof(fp_t));
    int32_t rc = regrow(ctx, &s->tbl_scratch, &s->tbl_scratch_cap, need);
    if (rc) return rc;
    g1j_t *J = (g1j_t *) s->tbl_scratch;
    fp_t *pre = (fp_t *) (J + (size_t) (MSM_WINDOWS - 1) * m);
    ZK_LAUNCH(PC_MSM_TABLES, 0.0, k_window_tables, dim3((uint32_t) ((m + 63) / 64)), dim3(64), s->tables, J, pre, (uint32_t) m);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

static int32_t ensure_rows(zk_ctx *ctx, uint32_t rows) {
    msm_state *s = ctx->msm;
    if (s->rows_cap >= rows) return ZK_OK;
    if (s->rowsJ) { ZK_HIP(hipStreamSynchronize(ctx->stream)); ZK_HIP(hipFree(s->rowsJ)); ZK_HIP(hipFree(s->rowsA)); s->rowsJ = nullptr; }
    ZK_HIP(hipMalloc((void **) &s->rowsJ, (size_t) rows * sizeof(g1j_t)));
    ZK_HIP(hipMalloc((void **) &s->rowsA, (size_t) rows * sizeof(g1a_t)));
    s->rows_cap = rows;
    return ZK_OK;
}

// full byte table for a generator set that is being used again (ZKCNN_MSM_FULL=0 keeps the bit-plane path)
static int32_t ensure_full_table(zk_ctx *ctx) {
    msm_state *s = ctx->msm;
    static const bool enabled = !(getenv("ZKCNN_MSM_FULL") && atoi(getenv("ZKCNN_MSM_FULL")) == 0);
    if (!enabled || s->full_ready || s->full_failed || s->gens_hits < 1 || s->m > MSM_FULL_MAX_M) return ZK_OK;
    const uint32_t m = (uint32_t) s->m;
    if (s->full_m != s->m) {
        if (s->full) { ZK_HIP(hipStreamSynchronize(ctx->stream)); ZK_HIP(hipFree(s->full)); s->full = nullptr; s->full_m = 0; }
        if (hipMalloc((void **) &s->full, (size_t) MSM_WINDOWS * 256 * m * sizeof(g1a_t)) != hipSuccess) {
            (void) hipGetLastError();
            s->full = nullptr;
            s->full_failed = true;             // not enough memory for the table: stay on the bit-plane path
            return ZK_OK;
        }
        s->full_m = s->m;
    }
    int32_t rc = regrow(ctx, &s->tbl_scratch, &s->tbl_scratch_cap, (size_t) 256 * m * (sizeof(g1j_t) + sizeof(fp_t)));
    if (rc) return rc;
    g1j_t *J = (g1j_t *) s->tbl_scratch;
    fp_t *pre = (fp_t *) (J + (size_t) 256 * m);
    for (uint32_t w = 0; w < MSM_WINDOWS; ++w) {
        for (uint32_t level = 0; level < 8; ++level) {
            const uint32_t work = (1u << level) * m;
            ZK_LAUNCH(PC_MSM_TABLES, 0.0, k_digit_level, dim3((work + 63) / 64), dim3(64), J, s->tables + (size_t) w * m, m, level);
        }
        ZK_LAUNCH(PC_MSM_TABLES, 0.0, k_digit_affine, dim3((16 * m + 63) / 64), dim3(64), s->full + (size_t) w * 256 * m, J, pre, m);
    }
    ZK_HIP(hipGetLastError());
    s->full_ready = true;
    return ZK_OK;
}

// MSMs through the byte table. Many rows: one block per row chunk walks all windows (small scalars stop early). Few rows
// (the two cross terms of an inner-product round): one (column, window) term per thread, then 64-wide trees; the last
// <= 64 partial points per row are summed on the host, where the result is needed anyway.
static int32_t run_msm_bytes(zk_ctx *ctx, const fr_t *scalars, uint64_t ld, const uint32_t *idx, uint32_t rows, uint32_t cols) {
    msm_state *s = ctx->msm;
    int32_t rc;
    if ((rc = ensure_rows(ctx, rows))) return rc;
    const uint32_t per = (cols + MSM_BLOCK - 1) / MSM_BLOCK;
    if (rows >= 64) {
        const uint32_t cpt = std::max<uint32_t>(1, std::min<uint32_t>(64, per));
        const uint32_t chunks = (cols + MSM_BLOCK * cpt - 1) / (MSM_BLOCK * cpt);
        g1j_t *dst = s->rowsJ;
        if (chunks > 1) {
            if ((rc = regrow(ctx, (void **) &s->partials, &s->partials_cap, (size_t) rows * chunks * sizeof(g1j_t)))) return rc;
            dst = s->partials;
        }
        for (uint32_t r0 = 0; r0 < rows; r0 += 32768) {
            const uint32_t nr = std::min<uint32_t>(32768, rows - r0);
            ZK_LAUNCH(PC_MSM_PLANES, 32.0 * (double) nr * (double) cols, k_msm_bytes, dim3(chunks, nr), dim3(MSM_BLOCK), dst + (size_t) r0 * chunks,
                      scalars + (size_t) r0 * ld, ld, idx ? idx + (size_t) r0 * ld : nullptr, s->full, (uint32_t) s->m, cols, cpt, 1u);
        }
        if (chunks > 1) ZK_LAUNCH(PC_MSM_FINISH, 0.0, k_sum_parts, dim3((rows + 63) / 64), dim3(64), s->rowsJ, s->partials, chunks, rows);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    // few rows: spread (column, window) terms over the whole GPU
    uint32_t cpt = 1;
    while ((uint64_t) ((cols + MSM_BLOCK * cpt - 1) / (MSM_BLOCK * cpt)) * MSM_WINDOWS > 4096) cpt *= 2;
    const uint32_t chunks = (cols + MSM_BLOCK * cpt - 1) / (MSM_BLOCK * cpt);
    uint32_t n = chunks * MSM_WINDOWS;
    if ((rc = regrow(ctx, (void **) &s->partials, &s->partials_cap, (size_t) rows * n * sizeof(g1j_t)))) return rc;
    if ((rc = regrow(ctx, (void **) &s->parts2, &s->parts2_cap, (size_t) rows * ((n + 63) / 64) * sizeof(g1j_t)))) return rc;
    ZK_LAUNCH(PC_MSM_PLANES, 32.0 * (double) rows * (double) cols, k_msm_bytes, dim3(n, rows), dim3(MSM_BLOCK), s->partials, scalars, ld, idx,
              s->full, (uint32_t) s->m, cols, cpt, (uint32_t) MSM_WINDOWS);
    g1j_t *cur = s->partials, *nxt = s->parts2;
    while (n > 64 || (rows > 8 && n > 1)) {
        const uint32_t n2 = (n + 63) / 64;
        ZK_LAUNCH(PC_MSM_FINISH, 0.0, k_tree_reduce, dim3(n2, rows), dim3(MSM_BLOCK), nxt, cur, n);
        std::swap(cur, nxt);
        n = n2;
    }
    ZK_HIP(hipGetLastError());
    if (rows > 8) {
        ZK_HIP(hipMemcpyAsync(s->rowsJ, cur, (size_t) rows * sizeof(g1j_t), hipMemcpyDeviceToDevice, ctx->stream));
        return ZK_OK;
    }
    std::vector<zkff::G1> part((size_t) rows * n);
    ZK_HIP(hipMemcpyAsync(part.data(), cur, part.size() * sizeof(g1j_t), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    for (uint32_t r = 0; r < rows; ++r) {
        zkff::G1 acc = part[(size_t) r * n];
        for (uint32_t k = 1; k < n; ++k) zkff::G1::add(acc, acc, part[(size_t) r * n + k]);
        s->host_rows[r] = acc;
    }
    s->host_rows_valid = true;
    return ZK_OK;
}

// rows independent MSMs over the cached generator tables; results (Jacobian) in s->rowsJ[0..rows)
static int32_t run_msm(zk_ctx *ctx, const fr_t *scalars, uint64_t ld, const uint32_t *idx, uint32_t rows, uint32_t cols,
                       const uint32_t *row_map = nullptr, uint32_t w_lo = 0, g1j_t *outJ = nullptr, bool sparse_windows = false) {
    msm_state *s = ctx->msm;
    s->host_rows_valid = false;
    if (!row_map && !outJ && w_lo == 0) {
        int32_t rc0 = ensure_full_table(ctx);
        if (rc0) return rc0;
        if (s->full_ready) return run_msm_bytes(ctx, scalars, ld, idx, rows, cols);
    }
    uint32_t wsplit, cpt;
    const uint32_t per = (cols + MSM_BLOCK - 1) / MSM_BLOCK;
    if (rows >= 64) { wsplit = 1; cpt = std::min<uint32_t>(64, per); }
    else if (sparse_windows) { wsplit = 1; cpt = std::min<uint32_t>(8, per); }   // few non-zero windows: one block walks them all
    else { wsplit = MSM_WINDOWS; cpt = std::min<uint32_t>(8, per); }
    cpt = std::max<uint32_t>(cpt, 1);
    const uint32_t chunks = (cols + MSM_BLOCK * cpt - 1) / (MSM_BLOCK * cpt), nparts = chunks * wsplit;
    int32_t rc;
    if ((rc = regrow(ctx, (void **) &s->partials, &s->partials_cap, (size_t) rows * MSM_PLANES * nparts * sizeof(g1j_t)))) return rc;
    if (s->rows_cap < rows) {
        if (s->rowsJ) { ZK_HIP(hipStreamSynchronize(ctx->stream)); ZK_HIP(hipFree(s->rowsJ)); ZK_HIP(hipFree(s->rowsA)); }
        ZK_HIP(hipMalloc((void **) &s->rowsJ, (size_t) rows * sizeof(g1j_t)));
        ZK_HIP(hipMalloc((void **) &s->rowsA, (size_t) rows * sizeof(g1a_t)));
        s->rows_cap = rows;
    }
    // gridDim.z is limited to 65535 rows per launch
    for (uint32_t r0 = 0; r0 < rows; r0 += 32768) {
        const uint32_t nr = std::min<uint32_t>(32768, rows - r0);
        ZK_LAUNCH(PC_MSM_PLANES, 32.0 * (double) nr * (double) cols, k_msm_planes, dim3(nparts, MSM_PLANES, nr), dim3(MSM_BLOCK), s->partials + (size_t) r0 * MSM_PLANES * nparts, scalars + (row_map ? 0 : (size_t) r0 * ld), ld, idx, s->tables, (uint32_t) s->m, cols, cpt, wsplit, row_map ? row_map + r0 : nullptr, w_lo);
    }
    ZK_LAUNCH(PC_MSM_FINISH, 0.0, k_msm_finish, dim3(rows), dim3(512), outJ ? outJ : s->rowsJ, s->partials, nparts);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// digit table D[d][j] = d g_j for the cached generators
static int32_t ensure_digit_table(zk_ctx *ctx) {
    msm_state *s = ctx->msm;
    if (s->digit_ready && s->digit_m == s->m) return ZK_OK;
    const uint32_t m = (uint32_t) s->m;
    if (s->digit_m != s->m) {
        if (s->digit) { ZK_HIP(hipStreamSynchronize(ctx->stream)); ZK_HIP(hipFree(s->digit)); s->digit = nullptr; }
        ZK_HIP(hipMalloc((void **) &s->digit, (size_t) 256 * m * sizeof(g1a_t)));
        s->digit_m = s->m;
    }
    int32_t rc = regrow(ctx, &s->tbl_scratch, &s->tbl_scratch_cap, (size_t) 256 * m * (sizeof(g1j_t) + sizeof(fp_t)));
    if (rc) return rc;
    g1j_t *J = (g1j_t *) s->tbl_scratch;
    fp_t *pre = (fp_t *) (J + (size_t) 256 * m);
    for (uint32_t level = 0; level < 8; ++level) {
        const uint32_t work = (1u << level) * m;
        ZK_LAUNCH(PC_MSM_TABLES, 0.0, k_digit_level, dim3((work + 63) / 64), dim3(64), J, s->tables, m, level);
    }
    ZK_LAUNCH(PC_MSM_TABLES, 0.0, k_digit_affine, dim3((16 * m + 63) / 64), dim3(64), s->digit, J, pre, m);
    ZK_HIP(hipGetLastError());
    s->digit_ready = true;
    return ZK_OK;
}

// commitments of `rows` rows of `cols` scalars (cols == number of cached generators): digit-table pass for the low
// byte of every scalar, then the bit-plane path for the remaining windows of the (few) rows that have any
static int32_t commit_rows(zk_ctx *ctx, const fr_t *scalars, uint64_t ld, uint32_t rows, uint32_t cols) {
    msm_state *s = ctx->msm;
    int32_t rc;
    s->host_rows_valid = false;
    if ((rc = ensure_full_table(ctx))) return rc;
    if (s->full_ready && rows >= 64) return run_msm_bytes(ctx, scalars, ld, nullptr, rows, cols);
    if ((rc = ensure_digit_table(ctx))) return rc;
    const uint32_t cpt = std::max<uint32_t>(1, std::min<uint32_t>(64, (cols + MSM_BLOCK - 1) / MSM_BLOCK));
    const uint32_t chunks = (cols + MSM_BLOCK * cpt - 1) / (MSM_BLOCK * cpt);
    if (s->rows_cap < rows) {
        if (s->rowsJ) { ZK_HIP(hipStreamSynchronize(ctx->stream)); ZK_HIP(hipFree(s->rowsJ)); ZK_HIP(hipFree(s->rowsA)); }
        ZK_HIP(hipMalloc((void **) &s->rowsJ, (size_t) rows * sizeof(g1j_t)));
        ZK_HIP(hipMalloc((void **) &s->rowsA, (size_t) rows * sizeof(g1a_t)));
        s->rows_cap = rows;
    }
    if (s->flags_cap < rows) {
        if (s->hi_flags) { ZK_HIP(hipStreamSynchronize(ctx->stream)); ZK_HIP(hipFree(s->hi_flags)); ZK_HIP(hipFree(s->row_list)); }
        ZK_HIP(hipMalloc((void **) &s->hi_flags, (size_t) rows * 4));
        ZK_HIP(hipMalloc((void **) &s->row_list, (size_t) rows * 4));
        s->flags_cap = rows;
    }
    ZK_HIP(hipMemsetAsync(s->hi_flags, 0, (size_t) rows * 4, ctx->stream));
    g1j_t *dst = s->rowsJ;
    if (chunks > 1) {
        if ((rc = regrow(ctx, (void **) &s->partials, &s->partials_cap, (size_t) rows * chunks * sizeof(g1j_t)))) return rc;
        dst = s->partials;
    }
    for (uint32_t r0 = 0; r0 < rows; r0 += 32768) {
        const uint32_t nr = std::min<uint32_t>(32768, rows - r0);
        ZK_LAUNCH(PC_MSM_PLANES, 32.0 * (double) nr * (double) cols, k_msm_digit, dim3(chunks, nr), dim3(MSM_BLOCK),
                  dst + (size_t) r0 * chunks, s->hi_flags + r0, scalars + (size_t) r0 * ld, ld, s->digit, (uint32_t) s->m, cols, cpt);
    }
    if (chunks > 1) ZK_LAUNCH(PC_MSM_FINISH, 0.0, k_sum_parts, dim3((rows + 63) / 64), dim3(64), s->rowsJ, s->partials, chunks, rows);
    ZK_HIP(hipGetLastError());
    std::vector<uint32_t> flags(rows), list;
    ZK_HIP(hipMemcpyAsync(flags.data(), s->hi_flags, (size_t) rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    for (uint32_t r = 0; r < rows; ++r) if (flags[r]) list.push_back(r);
    if (list.empty()) return ZK_OK;
    const uint32_t nl = (uint32_t) list.size();
    ZK_HIP(hipMemcpyAsync(s->row_list, list.data(), (size_t) nl * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = regrow(ctx, (void **) &s->tmpJ, &s->tmp_cap, (size_t) nl * sizeof(g1j_t)))) return rc;
    if ((rc = run_msm(ctx, scalars, ld, nullptr, nl, cols, s->row_list, 1, s->tmpJ, true))) return rc;
    ZK_LAUNCH(PC_MSM_FINISH, 0.0, k_add_rows, dim3((nl + 63) / 64), dim3(64), s->rowsJ, s->tmpJ, s->row_list, nl);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// few points: Jacobian -> affine on the host (one inversion each); many: on the device
static int32_t fetch_points(zk_ctx *ctx, uint32_t rows, uint64_t *out) {
    msm_state *s = ctx->msm;
    const uint32_t nseg = (rows + AFF_SEG - 1) / AFF_SEG;
    if (rows > 8 && nseg <= 1024) {
        int32_t rc = regrow(ctx, &s->aff_scratch, &s->aff_cap, ((size_t) rows + 3 * (size_t) nseg + 2) * sizeof(fp_t));
        if (rc) return rc;
        fp_t *pre = (fp_t *) s->aff_scratch, *seg = pre + rows, *seg_pre = seg + nseg, *seg_suf = seg_pre + nseg, *tot = seg_suf + nseg;
        ZK_LAUNCH(PC_MSM_FINISH, 0.0, k_aff_prefix, dim3((nseg + 63) / 64), dim3(64), pre, seg, s->rowsJ, rows);
        ZK_LAUNCH(PC_MSM_FINISH, 0.0, k_aff_scan, dim3(1), dim3(1024), seg_pre, seg_suf, tot, seg, nseg);
        ZK_HIP(hipGetLastError());
        zkff::Fp total, inv;
        ZK_HIP(hipMemcpyAsync(&total, tot, sizeof(fp_t), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        zkff::Fp::invert(inv, total);                      // the one sequential inversion: O(1) host work, like add_term
        ZK_HIP(hipMemcpyAsync(tot + 1, &inv, sizeof(fp_t), hipMemcpyHostToDevice, ctx->stream));
        ZK_LAUNCH(PC_MSM_FINISH, 0.0, k_aff_finish, dim3((nseg + 63) / 64), dim3(64), s->rowsA, s->rowsJ, pre, seg_pre, seg_suf, tot + 1, rows);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipMemcpyAsync(out, s->rowsA, (size_t) rows * sizeof(g1a_t), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        return ZK_OK;
    }
    if (rows > 8) {
        ZK_LAUNCH(PC_MSM_FINISH, 0.0, k_to_affine, dim3((rows + 63) / 64), dim3(64), s->rowsA, s->rowsJ, rows);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipMemcpyAsync(out, s->rowsA, (size_t) rows * sizeof(g1a_t), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        return ZK_OK;
    }
    zkff::G1 pj[8];
    if (s->host_rows_valid) {
        for (uint32_t i = 0; i < rows; ++i) pj[i] = s->host_rows[i];
    } else {
        ZK_HIP(hipMemcpyAsync(pj, s->rowsJ, (size_t) rows * sizeof(g1j_t), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    }
    for (uint32_t i = 0; i < rows; ++i) {
        zkff::G1Affine a = pj[i].toAffine();
        std::memcpy(out + 12 * i, &a, 96);
    }
    return ZK_OK;
}

static_assert(sizeof(zkff::G1) == sizeof(g1j_t) && sizeof(zkff::G1Affine) == sizeof(g1a_t), "host / device point layouts must agree");

#define CHECK_READY() do { if (!ctx || !ctx->circuit_ready) return ZK_ERR_STATE; ZK_HIP(hipSetDevice(ctx->device)); } while (0)
static inline const HFr &H(const uint64_t *p) { return *reinterpret_cast<const HFr *>(p); }

extern "C" int32_t zk_commit_input(zk_ctx *ctx, const uint64_t *gens, uint64_t n_gens, uint64_t *out_comm, uint64_t n_rows) {
    CHECK_READY();
    const dev_layer &L0 = ctx->L[0];
    const int n = L0.d.bit_length, rb = n >> 1, cb = n - rb;
    if (n_gens != (1ull << cb) || n_rows != (1ull << rb) || !gens || !out_comm) { ctx->err = "commit: wrong generator / row count"; return ZK_ERR_ARG; }
    int32_t rc;
    if ((rc = ensure_state(ctx)) || (rc = ensure_tables(ctx, gens, n_gens))) return rc;
    ctx->msm->rb = rb;
    ctx->msm->cb = cb;
    if ((rc = commit_rows(ctx, L0.val, n_gens, (uint32_t) n_rows, (uint32_t) n_gens))) return rc;
    return fetch_points(ctx, (uint32_t) n_rows, out_comm);
}

extern "C" int32_t zk_hyrax_open_init(zk_ctx *ctx, const uint64_t *x, uint32_t n) {
    CHECK_READY();
    msm_state *s = ctx->msm;
    if (!s || !s->tables) { ctx->err = "open before commit"; return ZK_ERR_STATE; }
    const dev_layer &L0 = ctx->L[0];
    if ((int) n != L0.d.bit_length) return ZK_ERR_ARG;
    const uint32_t rows = 1u << s->rb, m = 1u << s->cb;
    if (!s->a) {
        ZK_HIP(hipMalloc((void **) &s->a, (size_t) m * 32)); ZK_HIP(hipMalloc((void **) &s->b, (size_t) m * 32));
        ZK_HIP(hipMalloc((void **) &s->coef, (size_t) m * 32)); ZK_HIP(hipMalloc((void **) &s->Lrow, (size_t) rows * 32));
        ZK_HIP(hipMalloc((void **) &s->sL, (size_t) m * 32));          // rows: sL | sR
        ZK_HIP(hipMalloc((void **) &s->idxL, (size_t) m * 4));         // rows: idxL | idxR
        s->sR = s->sL + m / 2;
        s->idxR = s->idxL + m / 2;
        ZK_HIP(hipMalloc((void **) &s->d_y, 64));
    }
    const HFr *xs = reinterpret_cast<const HFr *>(x);
    int32_t rc;
    if ((rc = zk_eq_table1_dev(ctx, s->Lrow, s->rb, xs + s->cb, HFr::one()))) return rc;
    if ((rc = zk_eq_table1_dev(ctx, s->b, s->cb, xs, HFr::one()))) return rc;
    if ((rc = zk_col_combine_dev(ctx, s->a, L0.val, s->Lrow, m, rows))) return rc;        // w = L^T Z
    ZK_LAUNCH(PC_IPA, 0.0, k_fill, dim3((m + 255) / 256), dim3(256), s->coef, to_dev(HFr::one()), m);
    ZK_HIP(hipGetLastError());
    s->len = m;
    return ZK_OK;
}

extern "C" int32_t zk_hyrax_open_round(zk_ctx *ctx, uint64_t Lp[12], uint64_t Rp[12], uint64_t yL[4], uint64_t yR[4]) {
    CHECK_READY();
    msm_state *s = ctx->msm;
    if (!s || s->len < 2) return ZK_ERR_STATE;
    const uint32_t m = 1u << s->cb, h = s->len >> 1;
    ZK_LAUNCH(PC_IPA, 0.0, k_ipa_scalars, dim3((m + 255) / 256), dim3(256), s->sL, s->idxL, s->sR, s->idxR, s->a, s->coef, m, s->len);
    ZK_LAUNCH(PC_IPA, 0.0, k_ipa_dots, dim3(1), dim3(256), s->d_y, s->a, s->b, h);
    ZK_HIP(hipGetLastError());
    // two MSMs over m/2 generators each = the two rows of one batch (row stride m/2 for scalars and indices)
    int32_t rc;
    uint64_t pts[24];
    if ((rc = run_msm(ctx, s->sL, m / 2, s->idxL, 2, m / 2))) return rc;
    if ((rc = fetch_points(ctx, 2, pts))) return rc;
    ZK_HIP(hipMemcpyAsync(ctx->h_result, s->d_y, 64, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(Lp, pts, 96);
    std::memcpy(Rp, pts + 12, 96);
    std::memcpy(yL, &ctx->h_result[0], 32);
    std::memcpy(yR, &ctx->h_result[1], 32);
    return ZK_OK;
}

extern "C" int32_t zk_hyrax_open_fold(zk_ctx *ctx, const uint64_t c[4]) {
    CHECK_READY();
    msm_state *s = ctx->msm;
    if (!s || s->len < 2) return ZK_ERR_STATE;
    const uint32_t m = 1u << s->cb;
    ZK_LAUNCH(PC_IPA, 0.0, k_ipa_fold, dim3((m + 255) / 256), dim3(256), s->a, s->b, s->coef, to_dev(H(c)), m, s->len);
    ZK_HIP(hipGetLastError());
    s->len >>= 1;
    return ZK_OK;
}

extern "C" int32_t zk_hyrax_open_final(zk_ctx *ctx, uint64_t *a, uint32_t cap, uint32_t *n) {
    CHECK_READY();
    msm_state *s = ctx->msm;
    if (!s || !s->len || !a || !n) return ZK_ERR_STATE;
    if (s->len > cap) { ctx->err = "open_final: buffer too small"; return ZK_ERR_ARG; }
    ZK_HIP(hipMemcpyAsync(a, s->a, (size_t) s->len * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    *n = s->len;
    return ZK_OK;
}

// stand-alone MSM over arbitrary bases (kernel-level parity tests)
extern "C" int32_t zk_k_msm(zk_ctx *ctx, uint64_t out[12], const uint64_t *scalars, const uint64_t *bases, uint64_t n) {
    if (!ctx || !n || n > (1u << 20)) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = ensure_state(ctx)) || (rc = ensure_tables(ctx, bases, n))) return rc;
    if ((rc = zk_scratch(ctx, n * 32))) return rc;
    ZK_HIP(hipMemcpyAsync(ctx->scratch.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = run_msm(ctx, (const fr_t *) ctx->scratch.p, n, nullptr, 1, (uint32_t) n))) return rc;
    return fetch_points(ctx, 1, out);
}

// rows x cols row commitments over `cols` arbitrary bases: the commitInput data path on caller-supplied data
extern "C" int32_t zk_k_commit_rows(zk_ctx *ctx, uint64_t *out, const uint64_t *scalars, const uint64_t *bases, uint64_t rows,
                                    uint64_t cols) {
    if (!ctx || !rows || !cols || rows * cols > (1ull << 26)) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = ensure_state(ctx)) || (rc = ensure_tables(ctx, bases, cols))) return rc;
    if ((rc = zk_scratch(ctx, rows * cols * 32))) return rc;
    ZK_HIP(hipMemcpyAsync(ctx->scratch.p, scalars, rows * cols * 32, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = commit_rows(ctx, (const fr_t *) ctx->scratch.p, cols, (uint32_t) rows, (uint32_t) cols))) return rc;
    return fetch_points(ctx, (uint32_t) rows, out);
}

// Phase initialisation in few launches (round 3).
//
// A sumcheck phase of the reference starts with a handful of independent table builders -- initBetaTable (src/utils.cpp:32-51, 147-180), the
// V-table gathers (src/prover.cpp:205-212, 291-296), clearing the M tables -- followed by the gate loops (src/prover.cpp:224-233, 297-305).
// Round 2 ran each of them as its own launch (eq halves, eq expansion, gather, two zero fills, two scatters, two fix-ups, the phase-2
// constant term and its reduction: 9-11 launches and hipMemsetAsync calls per phase, ~620 per vgg11 proof, most of them a few microseconds
// of work behind a launch). Here a phase is
//     k_prep         every table builder of the phase as a JOB of one launch (jobs own disjoint block ranges; they are independent of one
//                    another and only depend on earlier phases)
//     k_gate_multi   both gate scatters of the phase and the phase-2 constant term (with its grid-wide reduction and hand-over to the host)
//     k_gate_fixup2  block carries of both scatters -- skipped when every list fits one block (then the scatter stores directly)
// i.e. 2-3 launches. Same field elements as before: sums are exact, so grouping does not change a byte of the transcript.
#pragma once
#include "kernels.cuh"
#include "conv_kernels.cuh"

// ---- job descriptors (kernel argument block: < 4 KB) ----
struct prep_eq_job {                 // out[i] = sum_p init[p] * eq(r[pt[p]][0..n), i)  for i < limit; entries >= tail_start scaled by tail_scale
    fr_t *out;
    uint64_t limit, tail_start;
    fr_t tail_scale;
    fr_t init[2];
    int32_t n, npoints, pt[2];
    int32_t c;                       // log2 of the entries one block builds (<= min(n, 12)); quarter tables of <= 64 entries in LDS
    uint32_t blk0, nblk;
};
struct prep_gather_job { fr_t *dst; const fr_t *src; const uint32_t *idx; uint64_t n_valid, n_total; uint32_t blk0, nblk; };   // dst[i] = i < n_valid ? src[idx ? idx[i] : i] : 0
struct prep_zero_job { fr_t *p; uint64_t n; uint32_t blk0, nblk; };
struct prep_small_job { fr_t *full; const liu_table *tabs; uint32_t blk0, nblk; };      // one block per small eq table (conv_kernels.cuh); tabs may be mapped host memory
#define PREP_MAX_EQ 2
#define PREP_MAX_GATHER 2
#define PREP_MAX_ZERO 4
struct prep_args {
    fr_vec r[2];                     // the points of the eq jobs (two jobs of one point each, or one job of two)
    prep_eq_job eq[PREP_MAX_EQ];
    prep_gather_job g[PREP_MAX_GATHER];
    prep_zero_job z[PREP_MAX_ZERO];
    prep_small_job sm;
    int32_t n_eq, n_g, n_z, pad_;
};

// eq table of one block's chunk. The chunk's 2^c entries share their high bits: eq(r, base + i) = s * eq(r[0..c), i) with the scalar
// s = init * prod_{j >= c} (bit_j(base) ? r_j : 1 - r_j). eq(r[0..c), .) is the outer product of two quarter tables built by doubling in LDS
// (wave 2p + side: point p, side 0 / 1), and the scalar is a tree product over the upper half of the same waves' lanes -- the SAME multiply
// instruction serves both (lanes < 32: doubling step, lanes >= 32: tree level), so the dependent chain is max(c / 2, log2(n - c)) + 2 products.
__device__ __forceinline__ void prep_eq_block(const prep_eq_job &J, const fr_vec *R, uint32_t lb) {
    __shared__ fr_t q[2][2][64];
    __shared__ fr_t s_scal[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = wave >> 1, side = wave & 1;
    const int c = J.c, qa = (c + 1) >> 1, qb = c - qa, hi_n = J.n - c;
    const uint64_t base = (uint64_t) lb << c;
    const bool on = p < J.npoints;
    const fr_vec &rv = R[J.pt[on ? p : 0]];
    const int nq = side ? qb : qa, qbase = side ? qa : 0;
    // tree operand of lanes 32..63 of the side-0 wave: factor j = lane - 32 of the high bits (one beyond them)
    fr_t tree = fr_one();
    if (on && side == 0 && lane >= 32) {
        const int j = lane - 32;
        if (j < hi_n) {
            const fr_t rj = rv.v[c + j];
            tree = ((base >> (c + j)) & 1) ? rj : fr_sub(fr_one(), rj);
        }
    }
    if (lane == 0) q[p][side][0] = fr_one();
    __syncthreads();
    const int tree_levels = hi_n <= 1 ? 0 : hi_n <= 2 ? 1 : hi_n <= 4 ? 2 : hi_n <= 8 ? 3 : hi_n <= 16 ? 4 : 5;
    const int steps = max(qa, tree_levels);
    for (int i = 0; i < steps; ++i) {
        const uint32_t half = 1u << i;
        const bool dbl = on && lane < 32 && i < nq && (uint32_t) lane < half;
        const bool tr = on && side == 0 && lane >= 32 && i < tree_levels;
        fr_t partner;                                  // tree level i: lane l (relative) takes l ^ (1 << i)
#pragma unroll
        for (int w = 0; w < 8; ++w) partner.v[w] = (uint32_t) __shfl_xor((int) tree.v[w], (int) half, 64);
        fr_t A = fr_zero(), B = fr_zero(), cur = fr_zero();
        if (dbl) { cur = q[p][side][lane]; A = cur; B = rv.v[qbase + i]; }
        if (tr) { A = tree; B = partner; }
        const fr_t x = fr_mul(A, B);
        if (dbl) { q[p][side][lane | half] = x; q[p][side][lane] = fr_sub(cur, x); }
        if (tr) tree = x;
        __syncthreads();
    }
    // scalar of the chunk into the side-1 quarter (one more product, <= 64 lanes): then an entry costs one product per point
    if (on && side == 0 && lane == 32) s_scal[p] = fr_mul(tree, J.init[p]);
    __syncthreads();
    if (on && side == 1 && lane < (1 << qb)) q[p][1][lane] = fr_mul(q[p][1][lane], s_scal[p]);
    __syncthreads();
    const uint32_t E = 1u << c, ma = (1u << qa) - 1;
    for (uint32_t i = tid; i < E; i += ZK_BLOCK) {
        const uint64_t gi = base + i;
        if (gi >= J.limit) break;
        fr_t acc = fr_mul(q[0][0][i & ma], q[0][1][i >> qa]);
        if (J.npoints > 1) acc = fr_add(acc, fr_mul(q[1][0][i & ma], q[1][1][i >> qa]));
        if (gi >= J.tail_start) acc = fr_mul(acc, J.tail_scale);
        fr_store(J.out + gi, acc);
    }
}

// small eq tables of the structured convolutions (k_eq_small_multi with 256 threads): eq(r[0..n), .) * init, n <= 12
__device__ __forceinline__ void prep_small_block(const prep_small_job &J, uint32_t lb) {
    __shared__ fr_t h[2][64];
    const liu_table &a = J.tabs[lb];
    const int n = a.n;
    if (n < 0) return;
    const int fh = n >> 1, sh = n - fh;
    const int side = threadIdx.x >> 6, t = threadIdx.x & 63;
    if (threadIdx.x < 128 && t == 0) h[side][0] = side ? fr_one() : fr_load(&a.init);
    __syncthreads();
    for (int i = 0; i < 6; ++i) {
        if (threadIdx.x < 128) {
            const int steps = side ? sh : fh, base = side ? fh : 0, half = 1 << i;
            if (i < steps && t < half) {
                const fr_t cur = h[side][t];
                const fr_t x = fr_mul(cur, fr_load(&a.r.v[base + i]));
                h[side][t | half] = x;
                h[side][t] = fr_sub(cur, x);
            }
        }
        __syncthreads();
    }
    fr_t *T = J.full + (size_t) lb * CONV_TAB_STRIDE;
    const uint32_t N = 1u << n, mask = (1u << fh) - 1;
    for (uint32_t j = threadIdx.x; j < N; j += ZK_BLOCK) fr_store(T + j, fr_mul(h[0][j & mask], h[1][j >> fh]));
}

struct k_prep_f {
    prep_args a;
    __device__ __forceinline__ void operator()() const {
    ZK_LATENCY_PRIO_F();
    const uint32_t blk = blockIdx.x;
    for (int k = 0; k < a.n_eq; ++k)
        if (blk >= a.eq[k].blk0 && blk < a.eq[k].blk0 + a.eq[k].nblk) { prep_eq_block(a.eq[k], a.r, blk - a.eq[k].blk0); return; }
    for (int k = 0; k < a.n_g; ++k) {
        const prep_gather_job &J = a.g[k];
        if (blk < J.blk0 || blk >= J.blk0 + J.nblk) continue;
        for (uint64_t i = (blk - J.blk0) * (uint64_t) ZK_BLOCK + threadIdx.x; i < J.n_total; i += (uint64_t) J.nblk * ZK_BLOCK) {
            fr_t x = fr_zero();
            if (i < J.n_valid) x = fr_load(J.src + (J.idx ? J.idx[i] : i));
            fr_store(J.dst + i, x);
        }
        return;
    }
    for (int k = 0; k < a.n_z; ++k) {
        const prep_zero_job &J = a.z[k];
        if (blk < J.blk0 || blk >= J.blk0 + J.nblk) continue;
        const fr_t z = fr_zero();
        for (uint64_t i = (blk - J.blk0) * (uint64_t) ZK_BLOCK + threadIdx.x; i < J.n; i += (uint64_t) J.nblk * ZK_BLOCK) fr_store(J.p + i, z);
        return;
    }
    if (a.sm.nblk && blk >= a.sm.blk0 && blk < a.sm.blk0 + a.sm.nblk) prep_small_block(a.sm, blk - a.sm.blk0);
    }
};

// ------------------------------------------------------------------------------------------------
// gate scatters of a phase in one launch (k_gate_reduce of kernels.cuh with the group size as a run-time bound, one block range per list),
// plus the phase-2 constant term (k_gate_sum2 + its reduction: the last block to arrive adds the block sums and hands them to the host).
// reference src/prover.cpp:224-233 (phase 1), 286-288 and 297-305 (phase 2)
// ------------------------------------------------------------------------------------------------
struct gate_list {
    const gate_rec *recs;
    uint64_t n;                      // groups (threads)
    fr_t *out;
    uint32_t G, blk0, nblk, carry0;  // records per thread; block range; first carry slot of the list
    int32_t post_scale, direct;      // direct: the list fits one block -- every segment is stored, nothing goes to the carry array
    fr_t post;
};
struct gate_multi_args {
    gate_list L[2];
    int32_t nlists, phase;
    const fr_t *beta_g, *beta_u, *val0, *val_prev, *two_mul;
    fr_t Vu0, Vu1;
    uint32_t *carry_key;
    fr_t *carry_val;
    // phase-2 constant term over the uni gates (two sums, split by where u lives)
    const gate_rec *uni2;
    uint64_t n_uni2;
    uint32_t sum_blk0, sum_nblk;
    fr_t *partials;
    uint32_t *counter;
    host_slot *aux;
    unsigned long long aux_seq;
    uint32_t nblocks, pad_;          // blocks of the whole launch
};

__device__ __forceinline__ fr_t gate_term2(const gate_rec &rc, const gate_multi_args &a, int post_scale) {
    if (GATE_DUMMY(rc.meta)) return fr_zero();
    fr_t t = fr_load(a.beta_g + rc.g);
    if (a.phase == 1) {
        if (GATE_HAS_VAL(rc.meta)) t = fr_mul(t, fr_load((GATE_IN_PREV(rc.meta) ? a.val_prev : a.val0) + rc.aux));
    } else {
        t = fr_mul(t, fr_load(a.beta_u + rc.aux));
        if (!post_scale) t = fr_mul(t, GATE_IN_PREV(rc.meta) ? a.Vu1 : a.Vu0);
    }
    const uint32_t sc = GATE_SC(rc.meta);
    if (sc) t = fr_mul(t, fr_load(a.two_mul + sc));
    return t;
}

struct k_gate_multi_f {
    gate_multi_args a;
    __device__ __forceinline__ void operator()() const {
    ZK_LATENCY_PRIO_F();
    const uint32_t blk = blockIdx.x;
    if (blk >= a.nblocks) return;               // (rows of a fused grid beyond this lane's own)
    if (a.sum_nblk && blk >= a.sum_blk0 && blk < a.sum_blk0 + a.sum_nblk) {
        __shared__ fr_t smem[2 * ZK_BLOCK / 64];
        const uint32_t lb = blk - a.sum_blk0;
        fr_t acc[2] = {fr_zero(), fr_zero()};
        for (uint64_t i = lb * (uint64_t) ZK_BLOCK + threadIdx.x; i < a.n_uni2; i += (uint64_t) a.sum_nblk * ZK_BLOCK) {
            const gate_rec rc = a.uni2[i];
            fr_t t = fr_mul(fr_load(a.beta_g + rc.g), fr_load(a.beta_u + rc.aux));
            const uint32_t sc = GATE_SC(rc.meta);
            if (sc) t = fr_mul(t, fr_load(a.two_mul + sc));
            if (GATE_IN_PREV(rc.meta)) acc[1] = fr_add(acc[1], t);
            else acc[0] = fr_add(acc[0], t);
        }
        fr_block_sum<2>(acc, smem);
        __syncthreads();
        grid_finish<2>(acc, a.partials, a.counter, a.aux, a.aux_seq, smem, false, a.sum_nblk, lb);
        return;
    }
    const int li = (a.nlists > 1 && blk >= a.L[1].blk0) ? 1 : 0;
    const gate_list &L = a.L[li];
    const uint32_t lb = blk - L.blk0;
    const uint64_t idx = lb * (uint64_t) ZK_BLOCK + threadIdx.x;
    const bool live = idx < L.n;
    uint32_t key = GATE_NOKEY;
    fr_t val = fr_zero();
    if (live) {
        const uint32_t G = L.G;
        const gate_rec *rp = L.recs + (idx >> 6) * (64ull * G) + (idx & 63);
        key = rp[0].key;
        fr_wide acc;
#pragma unroll
        for (int i = 0; i < 9; ++i) acc.v[i] = 0;
#pragma unroll 4
        for (uint32_t k = 0; k < G; ++k) {
            const gate_rec rc = rp[k * 64];
            frw_add(acc, gate_term2(rc, a, L.post_scale));
        }
        val = frw_reduce<5>(acc);
    }
    gate_segment_store(key, val, live, idx, L.n, L.out, a.carry_key + L.carry0, a.carry_val + L.carry0, L.post_scale != 0, L.post, lb, L.direct != 0);
    }
};

// carries of up to two lists: slots [0, n0) belong to out0, [n0, n0 + n1) to out1
struct k_gate_fixup2_f {
    fr_t *out0, *out1; const uint32_t *carry_key; const fr_t *carry_val; uint64_t n0, n1;
    __device__ __forceinline__ void operator()() const {
    ZK_LATENCY_PRIO_F();
    for (uint64_t s = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; s < n0 + n1; s += (uint64_t) gridDim.x * blockDim.x) {
        const uint64_t lo = s < n0 ? 0 : n0, hi = s < n0 ? n0 : n0 + n1;
        fr_t *out = s < n0 ? out0 : out1;
        const uint32_t key = carry_key[s];
        if (key == GATE_NOKEY) continue;
        bool head = true;
        if (s >= lo + 1) {
            uint32_t pk = carry_key[s - 1];
            if (pk == GATE_NOKEY && s >= lo + 2) pk = carry_key[s - 2];
            head = pk != key;
        }
        if (!head) continue;
        fr_t total = fr_load(carry_val + s);
        for (uint64_t t = s + 1; t < hi; ++t) {
            const uint32_t k2 = carry_key[t];
            if (k2 == GATE_NOKEY) continue;
            if (k2 != key) break;
            total = fr_add(total, fr_load(carry_val + t));
        }
        fr_store(out + key, total);
    }
    }
};

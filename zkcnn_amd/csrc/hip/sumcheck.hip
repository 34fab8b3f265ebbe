// libzkcnn_hip.so: the GKR sumcheck state machine on the GPU. Each extern "C" function is the binding of one reference prover method
// (see include/zkcnn_hip.h); the O(1)-per-round scalar bookkeeping (add_term, round counters, proof-size counter) stays in host code here
// exactly as in the reference, everything that is O(table) or O(gates) is a kernel. (Context: context.hip; residency: upload.hip; witness:
// witness.hip; verifier predicates: verifier.hip; commitment: hyrax.hip.)
#include <chrono>
#include <sched.h>
#include <time.h>
#include <algorithm>
#include <cmath>
#include <atomic>
#include <cstring>
#include <emmintrin.h>
#include "ctx.hpp"
#include "kernels.cuh"
#include "fs_tail.cuh"
#include "conv_kernels.cuh"
#include "prep_kernels.cuh"

// ------------------------------------------------------------------------------------------------
// launch policy: every size threshold of the state machine in one place, with the measurement behind it (all on vgg11-shaped circuits on one MI355X;
// the FFT-convolution circuits were checked not to lose by them, not tuned for)
// ------------------------------------------------------------------------------------------------
namespace policy {
// tables of up to 2^FINE_LOG entries take the 4-lanes-per-quad latency kernel (k_round_fine_f: a dependent chain of 2 products instead of 7); larger
// ones the product kernel (k_round_quad2_f). Round 1: switch-over sizes 2^14 .. 2^18 measured, 2^16 best by ~1%.
constexpr int FINE_LOG = 16;
// k_round_quad2_f holds 4 waves per SIMD (102 VGPRs, 32 KB of LDS staging per block): 1 024 blocks of 4 waves are exactly one resident set of the 1 024 SIMDs.
// A wave takes tiles of 64 pairs per table (round_stream.cuh), i.e. a block 128 quads (256 pairs in a phase's first round) per step.
constexpr uint32_t QUAD2_BLOCKS = 1024;
constexpr uint64_t QUAD2_QUADS_PER_BLOCK = 128;
constexpr uint32_t QUAD2_LANE_MIN_BLOCKS = 128;
// tables of up to 2^FULL_TABLE_LOG entries are always complete (their readers -- k_round_fine_f, k_mid, k_tail -- know no live-prefix bound); larger ones
// are built and folded up to their live prefix only. Also the largest table a resident segment kernel (k_mid) takes.
constexpr int FULL_TABLE_LOG = 18;
// a resident segment (k_mid) takes rounds of at most this many quads over both pairs: 65 536 quads = 1 024 chunks of 64 over its <= 256 workgroups
constexpr uint64_t MID_MAX_QUADS = 65536;
// DOT_PROD phase 1: once the folded tables hold at most 2^DOT_FILL_LOG entries the zeros behind X's live prefix are written out (the quadratic rounds
// behind the collapse read whole tables); and Y's fold behind the prefix becomes a streaming launch of its own from 2^DOT_STREAM_LOG pairs on (4 TB/s
// against ~2 inside the round kernel: DESIGN.md 4i)
constexpr int DOT_FILL_LOG = 20, DOT_STREAM_LOG = 17;
// Round 5: where that streaming fold would run, Y's dead region is left UNFOLDED while X's live prefix stays aligned (even): a live pair then never mixes with a
// dead entry, and X = 0 behind the prefix makes every product there vanish whatever Y holds. Before the first round that would mix them (an odd prefix) or that
// needs whole tables (DOT_FILL_LOG), the skipped folds are caught up in ONE pass over the unfolded values (k_dead_rows_f: entry j = sum_i eq(r_1..r_s; i) Y[(j << s) + i],
// 32 B per entry once instead of 48 B per entry and round). At most DEAD_MAX_FOLDS folds are skipped (the eq table of the catch-up has 2^s entries)
constexpr int DEAD_MAX_FOLDS = 20;
// a LANE of a lock-step batch hands a phase's tables to the host once they hold at most 2^LANE_TAIL_LOG entries (zk_set_host_tail's default for lanes): the
// phase's last five rounds are ~150 multiplications on the lane's host thread instead of five fused launches -- 44% of a vgg11 proof's rounds by count,
// 3e-5 of its products. 4 x 8 lanes on one box: 135 proofs/s with every round a launch, 139-143 / 144 / 144-147 / 144-147 with 2^3 / 2^4 / 2^5 / 2^6.
constexpr int LANE_TAIL_LOG = 5;
// ... or (round 5, experiment: off) the lanes' rounds on at most TAIL_QUADS quads as resident workgroups of ONE fused launch per phase, every lane trading
// polynomials and challenges through its own mailboxes (k_tail_live_f; quad_round_once). Measured in profiles/r05_lane_resident.md.
constexpr bool LANE_RESIDENT_TAIL = false;
// a LONE proof's resident tail kernel (k_tail<true>) ends when the tables hold 2^SOLO_EXPORT_LOG entries and exports them; the host finishes the phase
// (-1: the kernel runs every phase to its end, as in rounds 3-5 -- still what ZKCNN_MODE_GPU_TAIL asks for). A round of the last few entries costs the host
// ~1 us and the mailboxes ~10: vgg11 lone proof 28.1 / 26.7 / 26.3 / 27.7 ms at -1 / 6 / 5 / 4 (profiles/r05_solo_export.md).
constexpr int SOLO_EXPORT_LOG = 5;
}
// the hand-over size of the hybrid tail in force for this context (-1: none)
static inline int host_tail_log(const zk_ctx *ctx) {
    if (ctx->host_tail_log >= 0) return ctx->host_tail_log;
    return ctx->host_tail_log == -1 && ctx->batch ? policy::LANE_TAIL_LOG : -1;
}

// ------------------------------------------------------------------------------------------------
// building blocks
// ------------------------------------------------------------------------------------------------
// ---- one launch for all table builders of a phase (prep_kernels.cuh) ----
struct prep_plan {
    prep_args A;
    uint32_t blocks = 0;
    int npts = 0;
    bool overflow = false;
    bool latency = true;             // a proof that is alone on its GPU: many small blocks (short chains); several in flight: fewer, larger ones
    prep_plan() { std::memset(&A, 0, sizeof(A)); }
    explicit prep_plan(const zk_ctx *ctx) : latency(ctx->live_now) { std::memset(&A, 0, sizeof(A)); }
    bool empty() const { return blocks == 0; }
    int point(const HFr *r, int n) {
        if (npts >= 2) { overflow = true; return 0; }
        for (int i = 0; i < n; ++i) A.r[npts].v[i] = to_dev(r[i]);
        return npts++;
    }
    void zero(fr_t *p, uint64_t n) {
        if (!n) return;
        if (A.n_z >= PREP_MAX_ZERO) { overflow = true; return; }
        prep_zero_job &J = A.z[A.n_z++];
        J.p = p; J.n = n; J.blk0 = blocks; J.nblk = grid_for(n, 1024);
        blocks += J.nblk;
    }
    // dst[i] = i < n_valid ? src[idx ? idx[i] : i] : 0, i < n_total
    void gather(fr_t *dst, const fr_t *src, const uint32_t *idx, uint64_t n_valid, uint64_t n_total) {
        if (!n_total) return;
        if (A.n_g >= PREP_MAX_GATHER) { overflow = true; return; }
        prep_gather_job &J = A.g[A.n_g++];
        J.dst = dst; J.src = src; J.idx = idx; J.n_valid = n_valid; J.n_total = n_total; J.blk0 = blocks; J.nblk = grid_for(n_total, 1024);
        blocks += J.nblk;
    }
    // out[i] = a * eq(r0, i) + b * eq(r1, i), i < min(2^n, limit); entries >= tail_start additionally scaled (the relu_rou factor on the
    // constraint rows, reference src/prover.cpp:221-222). `limit`: a layer whose size is not a power of two -- nothing refers to the rest
    void eq(fr_t *out, int n, const HFr *r0, const HFr &a, const HFr *r1, const HFr &b, uint64_t tail_start, const HFr &tail_scale, uint64_t limit = ~0ull) {
        if (n < 0) return;
        if (n > ZK_MAX_VARS) { overflow = true; return; }
        const uint64_t len = std::min<uint64_t>(1ull << n, std::max<uint64_t>(limit, 1));
        const bool has_a = !a.isZero() && r0, has_b = !b.isZero() && r1;
        if (!has_a && !has_b) { zero(out, len); return; }
        if (A.n_eq >= PREP_MAX_EQ) { overflow = true; return; }
        prep_eq_job &J = A.eq[A.n_eq++];
        J.out = out; J.n = n; J.limit = len; J.tail_start = tail_start; J.tail_scale = to_dev(tail_scale);
        J.npoints = 0;
        if (has_b) { J.pt[J.npoints] = point(r1, n); J.init[J.npoints++] = to_dev(b); }
        if (has_a) { J.pt[J.npoints] = point(r0, n); J.init[J.npoints++] = to_dev(a); }
        // entries per block: small tables are latency bound (one entry per thread, short quarter tables), large ones amortise the block's prelude
        // Alone on the GPU one entry per thread and short quarter tables win; with several proofs in flight every block's prelude (the doubling steps, the
        // tree product) is SIMD time the others wait for, so blocks are four times larger.
        J.c = latency ? std::min(n, n <= 16 ? 8 : n <= 20 ? 10 : 12) : std::min(n, n <= 18 ? 10 : 12);
        J.blk0 = blocks;
        J.nblk = (uint32_t) ((len + (1ull << J.c) - 1) >> J.c);
        blocks += J.nblk;
    }
    void eq1(fr_t *out, int n, const HFr *r, const HFr &init, uint64_t limit = ~0ull) { eq(out, n, r, init, nullptr, HFr(0LL), ~0ull, HFr::one(), limit); }
    void small_tables(fr_t *full, const liu_table *tabs, uint32_t count) {
        A.sm.full = full; A.sm.tabs = tabs; A.sm.blk0 = blocks; A.sm.nblk = count;
        blocks += count;
    }
    int32_t launch(zk_ctx *ctx) {
        if (overflow) { ctx->err = "phase preparation: too many jobs for one launch"; return ZK_ERR_STATE; }
        if (!blocks) return ZK_OK;
        zk_launch_f(ctx, PC_EQ, 0.0, dim3(blocks), k_prep_f{A});
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
};

static int32_t eq_table(zk_ctx *ctx, fr_t *out, int n, const HFr *r0, const HFr &a, const HFr *r1, const HFr &b,
                        uint64_t tail_start, const HFr &tail_scale, uint64_t limit = ~0ull) {
    if (n < 0) return ZK_OK;
    if (n > ZK_MAX_VARS) { ctx->err = "eq table too large"; return ZK_ERR_ARG; }
    prep_plan P(ctx);
    P.eq(out, n, r0, a, r1, b, tail_start, tail_scale, limit);
    return P.launch(ctx);
}
static int32_t eq_table1(zk_ctx *ctx, fr_t *out, int n, const HFr *r, const HFr &init, uint64_t limit = ~0ull) {
    return eq_table(ctx, out, n, r, init, nullptr, HFr(0LL), ~0ull, HFr::one(), limit);
}

int32_t zk_eq_table1_dev(zk_ctx *ctx, fr_t *out, int n, const HFr *r, const HFr &init) { return eq_table1(ctx, out, n, r, init); }
int32_t zk_eq_table_dev(zk_ctx *ctx, fr_t *out, int n, const HFr *r0, const HFr &a, const HFr *r1, const HFr &b, uint64_t tail_start, const HFr &tail_scale, uint64_t limit) {
    return eq_table(ctx, out, n, r0, a, r1, b, tail_start, tail_scale, limit);
}

// out[j] = sum_i L[i] * Z[i * cols + j]  (Hyrax opening vector w = L^T Z)
int32_t zk_col_combine_dev(zk_ctx *ctx, fr_t *out, const fr_t *Z, const fr_t *L, uint32_t cols, uint32_t rows) {
    uint32_t chunks = std::max<uint32_t>(1, std::min<uint32_t>(rows, (1u << 18) / std::max<uint32_t>(cols, 1)));
    chunks = std::min<uint32_t>(chunks, 65535);
    const uint32_t per = (rows + chunks - 1) / chunks;
    chunks = (rows + per - 1) / per;
    int32_t rc = zk_scratch(ctx, (size_t) chunks * cols * 32);
    if (rc) return rc;
    fr_t *part = chunks == 1 ? out : (fr_t *) ctx->scratch.p;
    dim3 grid((cols + ZK_BLOCK - 1) / ZK_BLOCK, chunks);
    zk_launch_d<k_col_combine, ZK_BLOCK>(ctx, PC_MATVEC, 0.0, grid, part, Z, L, cols, rows, per);
    if (chunks > 1)
        zk_launch_f<k_sum_rows_f, 1024>(ctx, PC_MATVEC, 0.0, dim3((cols + 63) / 64), k_sum_rows_f{out, (const fr_t *) part, cols, chunks});
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// copies `count` elements of d_result to the pinned host mirror and waits for the stream
static int32_t fetch_result(zk_ctx *ctx, int count) {
    if (ctx->batch) { int32_t rc = zk_batch_sync_point(ctx); if (rc) return rc; }
    ZK_HIP(hipMemcpyAsync(ctx->h_result, ctx->d_result, (size_t) count * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// algorithmic bytes of one scatter: per uni gate 12 B record + 32 B gather, per bin gate 16 B record
// + 2 x 32 B gathers, + 32 B per output entry
// Waits until the fused round kernel has published sequence number `seq` in the mapped host slot. Spinning on the
// slot keeps a stream synchronisation (and a D2H copy) off the critical path of every round; if the value does not
// show up quickly the stream is synchronised instead (long kernels, or host memory that is not fine-grained).
static int32_t wait_slot(zk_ctx *ctx, unsigned long long seq);
int32_t zk_wait_slot(zk_ctx *ctx, unsigned long long seq) { return wait_slot(ctx, seq); }
static int32_t wait_slot(zk_ctx *ctx, unsigned long long seq) {
    // a lane of a batch: its launch may still be deferred -- the driver runs the other lanes up to here and flushes (ctx.hpp)
    if (ctx->batch) { int32_t rc = zk_batch_sync_point(ctx); if (rc) return rc; }
    volatile unsigned long long *p = &ctx->h_slot->seq;
    for (uint64_t spins = 0; *p != seq; ++spins) {
        if (spins > (1ull << 21)) {
            ZK_HIP(hipStreamSynchronize(ctx->stream));
            if (*p != seq) { ctx->err = "round result was not published"; return ZK_ERR_STATE; }
            break;
        }
        // a small round answers within ~20 us of spinning; behind a long kernel (a 2^24-entry round, a gate scatter) the thread stops
        // burning its core: 8 ranks x 8 proving threads share one host, and a yielding waiter costs a late round ~1 us, not the others a core
        if (spins < 4096) __builtin_ia32_pause();
        else if (spins < 65536) sched_yield();
        else { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return ZK_OK;
}

// ---- both gate scatters of a phase (and the phase-2 constant term) in one launch; a fix-up launch only if a list spans blocks ----
struct gate_job {
    fr_t *out;
    const gate_rec *recs;
    uint64_t n, n_real, n_uni;      // records in the padded list (a multiple of G), gates among them, uni gates among those
    uint32_t G;
    int uniform_u;                   // phase 2: every gate's u operand in layer 0 (0) / the previous layer (1), else -1
    uint64_t out_len;
};
// algorithmic bytes of one scatter: per uni gate 12 B record + 32 B gather, per bin gate 16 B record + 2 x 32 B gathers,
// + 32 B per output entry
static int32_t gate_multi(zk_ctx *ctx, int phase, const dev_layer &prev, const gate_job *jobs, int njobs, const dev_layer *sum_layer) {
    gate_multi_args A;
    std::memset(&A, 0, sizeof(A));
    double gate_bytes = 0;
    uint32_t blocks = 0, carry = 0;
    bool need_fix = false;
    for (int k = 0; k < njobs; ++k) {
        const gate_job &j = jobs[k];
        if (!j.n) continue;
        gate_list &L = A.L[A.nlists++];
        L.recs = j.recs; L.n = j.n / j.G; L.out = j.out; L.G = j.G;
        L.blk0 = blocks; L.nblk = (uint32_t) ((L.n + ZK_BLOCK - 1) / ZK_BLOCK);
        L.carry0 = carry;
        L.direct = L.nblk == 1 ? 1 : 0;
        L.post_scale = (phase == 2 && j.uniform_u >= 0) ? 1 : 0;
        L.post = to_dev(j.uniform_u == 1 ? ctx->V_u1 : ctx->V_u0);
        blocks += L.nblk;
        if (!L.direct) { carry += 2 * L.nblk; need_fix = true; }
        gate_bytes += 44.0 * (double) j.n_uni + 80.0 * (double) (j.n_real - j.n_uni) + 32.0 * (double) j.out_len;
    }
    if (carry > ctx->carry_slots) { ctx->err = "carry buffer too small"; return ZK_ERR_STATE; }
    if (sum_layer && sum_layer->n_uni2) {
        // the two sums go to a mapped host slot; add_term = V_u0 s0 + V_u1 s1 is formed when the first round needs it (resolve_add_term)
        A.uni2 = sum_layer->uni2; A.n_uni2 = sum_layer->n_uni2;
        A.sum_blk0 = blocks; A.sum_nblk = std::min<uint32_t>(grid_for(A.n_uni2, 1024), ctx->partial_blocks);
        blocks += A.sum_nblk;
        A.partials = ctx->partials; A.counter = ctx->d_counter + 1;
        A.aux = (host_slot *) ctx->d_aux; A.aux_seq = ++ctx->aux_seq;
        ctx->add_pending = true;
    }
    if (!blocks) return ZK_OK;
    A.phase = phase;
    A.beta_g = ctx->beta_g[ctx->beta_g_cur]; A.beta_u = ctx->beta_u;
    A.val0 = ctx->L[0].val; A.val_prev = prev.val; A.two_mul = ctx->two_mul;
    A.Vu0 = to_dev(ctx->V_u0); A.Vu1 = to_dev(ctx->V_u1);
    A.carry_key = ctx->carry_key; A.carry_val = ctx->carry_val;
    A.nblocks = blocks;
    zk_launch_f(ctx, PC_GATE, gate_bytes, dim3(blocks), k_gate_multi_f{A});
    if (need_fix) {
        const uint64_t n0 = (A.nlists > 0 && !A.L[0].direct) ? 2ull * A.L[0].nblk : 0, n1 = (A.nlists > 1 && !A.L[1].direct) ? 2ull * A.L[1].nblk : 0;
        // (carry slots are handed out in list order: a direct list takes none)
        fr_t *o0 = n0 ? A.L[0].out : (n1 ? A.L[1].out : nullptr), *o1 = n0 && n1 ? A.L[1].out : nullptr;
        zk_launch_f(ctx, PC_GATE_FIX, 0.0, dim3(grid_for(n0 + n1)), k_gate_fixup2_f{o0, o1, ctx->carry_key, ctx->carry_val, n0 ? n0 : n1, n0 ? n1 : 0});
    }
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// a table of more than 2^18 entries is first read by the round kernel that stops at the live prefix (rounded up to whole quads): its builders
// stop there too (one guard quad behind it); smaller tables are read whole
#define ZK_FULL_TABLE_LOG policy::FULL_TABLE_LOG
static inline uint64_t table_end(uint64_t len, uint64_t live) {
    return len <= (1ull << ZK_FULL_TABLE_LOG) ? len : std::min<uint64_t>(len, ((live + 3) & ~3ull) + 4);
}

// V-table of one side: layer-0 subset through ori ids, or the previous layer as it is
static inline const fr_t *vin(const table_pair &t) { return t.Vsrc ? t.Vsrc : t.V[t.cur]; }

static void load_v_table(zk_ctx *ctx, prep_plan &P, table_pair &t, int b, int bl, uint32_t size, const uint32_t *ori, const dev_layer &prev) {
    if (bl < 0) return;
    const uint64_t len = 1ull << bl;
    fr_t *dst = t.V[0];
    if (b == 1 && prev.val_len >= len) {            // the previous layer as it is: read in place until the first fold
        t.Vsrc = prev.val;
        return;
    }
    if (b == 0) P.gather(dst, ctx->L[0].val, ori, (uint64_t) size, table_end(len, size));
    else P.gather(dst, prev.val, nullptr, std::min<uint64_t>(len, prev.val_len), len);       // (a table longer than the layer: zero extended)
}

static void reset_pairs(zk_ctx *ctx, int bl0, int bl1) {
    const int bl[2] = {bl0, bl1};
    if (ctx->dot_quad) {               // (a DOT_PROD phase that was abandoned behind its hand-over: pair 1 gets its own M buffers back)
        ctx->tp[1].M[0] = ctx->dot_saved_M[0];
        ctx->tp[1].M[1] = ctx->dot_saved_M[1];
        ctx->dot_quad = false;
    }
    for (int b = 0; b < 2; ++b) {
        ctx->tp[b].cur = 0;
        ctx->tp[b].tail_valid = false;
        ctx->tp[b].Vsrc = nullptr;
        ctx->tp[b].len = bl[b] >= 0 ? 1ull << bl[b] : 0;
        ctx->tp[b].live = ctx->tp[b].len;
        ctx->tp[b].absorbed = false;
        ctx->tp[b].final_v.clear();
        ctx->tp[b].abs_m.clear();
        ctx->tp[b].abs_m_valid = false;
    }
}
// the multipliers of absorbed pairs follow add_term: x (1 - r) every round
static inline void scale_absorbed(zk_ctx *ctx, const HFr &r) {
    for (int b = 0; b < 2; ++b)
        if (ctx->tp[b].absorbed && ctx->tp[b].abs_m_valid) ctx->tp[b].abs_m = ctx->tp[b].abs_m * (HFr::one() - r);
}

static fr_t *powers_of_root(zk_ctx *ctx, int n, bool inverse);
fr_t *zk_powers_of_root(zk_ctx *ctx, int n, bool inverse) { return powers_of_root(ctx, n, inverse); }
static fr_t *powers_of_root(zk_ctx *ctx, int n, bool inverse) {
    auto &tab = ctx->root_pw[inverse ? 1 : 0];
    if ((int) tab.size() <= n) tab.resize(n + 1, nullptr);
    if (tab[n]) return tab[n];
    // n-1 successive square roots of -1 (reference src/utils.cpp:224-232), then the power table
    HFr w = HFr::one();
    if (n > 0) {
        w = -HFr::one();
        for (int k = 1; k < n; ++k) HFr::squareRoot(w, w);
    }
    if (inverse) HFr::inv(w, w);
    std::vector<HFr> pw((size_t) 1 << n);
    pw[0] = HFr::one();
    for (size_t i = 1; i < pw.size(); ++i) pw[i] = pw[i - 1] * w;
    fr_t *d = nullptr;
    if (zk_dev_alloc(ctx, (void **) &d, pw.size() * 32)) return nullptr;
    if (hipMemcpy(d, pw.data(), pw.size() * 32, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    tab[n] = d;
    return d;
}

static int32_t phi_table(zk_ctx *ctx, fr_t *out, const HFr *rx, const HFr &scale, int n, bool inverse);
int32_t zk_phi_table_dev(zk_ctx *ctx, fr_t *out, const HFr *rx, const HFr &scale, int n, bool inverse) { return phi_table(ctx, out, rx, scale, n, inverse); }
static int32_t phi_table(zk_ctx *ctx, fr_t *out, const HFr *rx, const HFr &scale, int n, bool inverse) {
    fr_t *pw = powers_of_root(ctx, n, inverse);
    if (!pw) { ctx->err = "root table allocation failed"; return ZK_ERR_NOMEM; }
    const int vars = inverse ? n - 1 : n;
    const uint32_t cnt = inverse ? 1u << n : 1u << (n - 1);
    fr_vec R;
    for (int j = 0; j < vars; ++j) R.v[j] = to_dev(rx[j]);
    zk_launch_d<k_phi, ZK_BLOCK>(ctx, PC_PHI, 0.0, dim3((cnt + ZK_BLOCK - 1) / ZK_BLOCK), out, (const fr_t *) pw, R, to_dev(scale), n, vars, cnt);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// out[u] = sum_g val[g * stride + u] * beta[g]
static int32_t strided_matvec(zk_ctx *ctx, fr_t *out, const fr_t *val, const fr_t *beta, uint32_t len, uint32_t stride,
                              uint32_t cnt) {
    uint32_t chunks = std::max<uint32_t>(1, std::min<uint32_t>(cnt, (1u << 16) / std::max<uint32_t>(len, 1)));
    chunks = std::min<uint32_t>(chunks, 65535);
    const uint32_t per = (cnt + chunks - 1) / chunks;
    chunks = (cnt + per - 1) / per;
    int32_t rc = zk_scratch(ctx, (size_t) chunks * len * 32);
    if (rc) return rc;
    fr_t *part = chunks == 1 ? out : (fr_t *) ctx->scratch.p;
    dim3 grid((len + ZK_BLOCK - 1) / ZK_BLOCK, chunks);
    zk_launch_d<k_strided_matvec, ZK_BLOCK>(ctx, PC_MATVEC, 0.0, grid, part, val, beta, len, stride, cnt, per);
    if (chunks > 1)
        zk_launch_f<k_sum_rows_f, 1024>(ctx, PC_MATVEC, 0.0, dim3((len + 63) / 64), k_sum_rows_f{out, (const fr_t *) part, len, chunks});
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// folds table `b`'s V (and M if with_m) with r; len must be >= 2
static int32_t fold_pair(zk_ctx *ctx, table_pair &t, const HFr &r, bool with_m) {
    const fr_t rr = to_dev(r);
    zk_launch_f(ctx, PC_FOLD, 0.0, dim3(grid_for(t.len / 2)), k_fold_f{vin(t), t.V[t.cur ^ 1], t.len, rr});
    t.Vsrc = nullptr;
    if (with_m)
        zk_launch_f(ctx, PC_FOLD, 0.0, dim3(grid_for(t.len / 2)), k_fold_f{(const fr_t *) t.M[t.cur], t.M[t.cur ^ 1], t.len, rr});
    ZK_HIP(hipGetLastError());
    t.cur ^= 1;
    t.len >>= 1;
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// state machine
// ------------------------------------------------------------------------------------------------
#define CHECK_READY_ROUND() ZK_CHECK_READY_ROUND()
#define CHECK_READY() ZK_CHECK_READY()

extern "C" int32_t zk_prover_init(zk_ctx *ctx) {
    CHECK_READY();
    ctx->proof_size = 0;
    const size_t n = ctx->L.size();
    ctx->r_u.assign(n + 1, std::vector<HFr>());
    ctx->r_v.assign(n + 1, std::vector<HFr>());
    return ZK_OK;
}

extern "C" int32_t zk_vres(zk_ctx *ctx, const uint64_t *r, uint32_t output_size, uint32_t r_size, uint64_t out[4]) {
    CHECK_READY();
    const dev_layer &top = ctx->L.back();
    if ((1ull << r_size) < output_size || (1ull << r_size) > top.val_len) return ZK_ERR_ARG;
    table_pair &t = ctx->tp[0];
    t.cur = 0;
    t.len = 1ull << r_size;
    t.Vsrc = top.val;
    for (uint32_t i = 0; i < r_size; ++i) {
        int32_t rc = fold_pair(ctx, t, H(r + 4 * i), false);
        if (rc) return rc;
    }
    // the value left goes to the host through the hand-over slot (no copy, no stream synchronisation)
    eval_args E;
    std::memset(&E, 0, sizeof(E));
    E.p[0] = vin(t); E.n[0] = 1;
    E.slot = (host_slot *) ctx->d_slot;
    E.seq = ++ctx->slot_seq;
    zk_launch_f<k_eval_pairs_f, 64>(ctx, PC_FOLD, 0.0, dim3(1), k_eval_pairs_f{E});
    ZK_HIP(hipGetLastError());
    t.Vsrc = nullptr;
    int32_t rc = wait_slot(ctx, E.seq);
    if (rc) return rc;
    put(out, ctx->h_slot->v[0]);
    t.len = 0;
    ctx->proof_size += 32;
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_init_all(zk_ctx *ctx, const uint64_t *r_0, uint32_t n) {
    CHECK_READY();
    ctx->sumcheck_id = (int) ctx->L.size();
    if ((int) n != ctx->L.back().d.bit_length) return ZK_ERR_ARG;
    auto &dst = ctx->r_u[ctx->sumcheck_id];
    dst.resize(n);
    for (uint32_t i = 0; i < n; ++i) dst[i] = H(r_0 + 4 * i);
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_init(zk_ctx *ctx, const uint64_t alpha[4], const uint64_t beta[4]) {
    CHECK_READY();
    if (ctx->sumcheck_id < 2) return ZK_ERR_STATE;
    ctx->alpha = H(alpha);
    ctx->beta = H(beta);
    ctx->r_0 = ctx->r_u[ctx->sumcheck_id].data();
    ctx->r_1 = ctx->r_v[ctx->sumcheck_id].data();
    --ctx->sumcheck_id;
    return ZK_OK;
}

// ---- factored gate sums of a structured convolution layer (conv_kernels.cuh) ----
static void conv_tab(liu_table &T, const HFr *r, int from, int n, const HFr &init) {
    T.n = n; T.fh = n >> 1; T.sh = n - (n >> 1); T.pad_ = 0;
    for (int j = 0; j < n; ++j) T.r.v[j] = to_dev(r[from + j]);
    T.init = to_dev(init);
}
// phase 1, descriptors: the small eq tables S, A, P of the layer's one or two claim points (built by the phase's k_prep launch).
// Returns K = number of points with a non-zero weight (0: every sum is zero, the caller zero-fills M)
static int conv_phase1_tables(zk_ctx *ctx, const dev_layer &cur, const HFr &a0, const HFr &a1) {
    const conv_desc &c = cur.conv;
    liu_table *T = (liu_table *) ctx->h_conv_tabs;
    for (int t = 0; t < CT_COUNT; ++t) T[t].n = -1;
    const int bpos = c.bx_o + c.by_o, bl = cur.d.bit_length;
    int K = 0;
    const HFr *pts[2] = {ctx->r_0, ctx->r_1};
    const HFr inits[2] = {a0, a1};
    for (int k = 0; k < 2; ++k) {
        if (inits[k].isZero() || !pts[k]) continue;
        conv_tab(T[K ? CT_S1 : CT_S0], pts[k], 0, bpos, HFr::one());
        conv_tab(T[K ? CT_A1 : CT_A0], pts[k], bpos, c.bc_o, HFr::one());
        conv_tab(T[K ? CT_P1 : CT_P0], pts[k], bpos + c.bc_o, bl - bpos - c.bc_o, inits[k]);
        ++K;
    }
    return K;
}
// phase 1, sums: M over the previous layer's table
static int32_t conv_phase1(zk_ctx *ctx, const dev_layer &cur, fr_t *M, uint64_t len) {
    const conv_desc &c = cur.conv;
    const int K = ctx->conv_K;
    const uint32_t wlen = c.CI * c.m * c.m, per = 8, chunks = (c.CO + per - 1) / per;
    fr_t *part = chunks == 1 ? ctx->conv_wa : ctx->conv_part;
    zk_launch_f(ctx, PC_GATE, 0.0, dim3((wlen + ZK_BLOCK - 1) / ZK_BLOCK, chunks), k_conv_wa_f{part, (const fr_t *) ctx->L[0].val + c.wstart, (const fr_t *) ctx->conv_small, wlen, c.CO, per, K});
    if (chunks > 1)
        zk_launch_f<k_sum_rows_f, 1024>(ctx, PC_GATE, 0.0, dim3((K * wlen + 63) / 64), k_sum_rows_f{ctx->conv_wa, (const fr_t *) part, K * wlen, chunks});
    zk_launch_f(ctx, PC_GATE, 0.0, dim3(grid_for(len)), k_conv_m1_f{M, (const fr_t *) ctx->conv_wa, (const fr_t *) ctx->conv_small, c, K, len});
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
// phase 2, descriptors (second set: D, C, Pu over the phase-1 point; S, A, P of the same layer stay in conv_small)
static void conv_phase2_tables(zk_ctx *ctx, const dev_layer &cur, const HFr *ru) {
    const conv_desc &c = cur.conv;
    liu_table *T = (liu_table *) ctx->h_conv_tabs + CT_COUNT;
    for (int t = 0; t < CT_COUNT; ++t) T[t].n = -1;
    const int bpos = c.bx_i + c.by_i;
    conv_tab(T[CT_D], ru, 0, bpos, HFr::one());
    conv_tab(T[CT_C], ru, bpos, c.bc_i, HFr::one());
    conv_tab(T[CT_PU], ru, bpos + c.bc_i, cur.d.max_bl_u - bpos - c.bc_i, HFr::one());
}
// phase 2, sums: M over the layer-0 subset table of the weights
static int32_t conv_phase2(zk_ctx *ctx, const dev_layer &cur, fr_t *M, uint64_t len) {
    const conv_desc &c = cur.conv;
    const int K = ctx->conv_K;
    const uint32_t mm = c.m * c.m;
    zk_launch_f(ctx, PC_GATE, 0.0, dim3(mm, K), k_conv_e_f{ctx->conv_e, (const fr_t *) ctx->conv_small, c, K});
    zk_launch_f(ctx, PC_GATE, 0.0, dim3((c.CO * mm + c.CI + ZK_BLOCK - 1) / ZK_BLOCK), k_conv_ae_f{ctx->conv_ae, (const fr_t *) ctx->conv_e, (const fr_t *) ctx->conv_small, c, K, to_dev(ctx->V_u1)});
    zk_launch_f(ctx, PC_GATE, 0.0, dim3(grid_for(len)), k_conv_m2_f{M, (const uint32_t *) cur.ori_v, (const fr_t *) ctx->conv_ae, c, cur.d.size_v[0], len});
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_init_phase1(zk_ctx *ctx, const uint64_t relu_rou[4]) {
    CHECK_READY();
    const int id = ctx->sumcheck_id;
    if (id < 1 || id >= (int) ctx->L.size()) return ZK_ERR_STATE;
    const dev_layer &cur = ctx->L[id], &prev = ctx->L[id - 1];
    const zk_layer_desc &d = cur.d;
    int32_t rc;
    reset_pairs(ctx, d.bit_length_u[0], d.bit_length_u[1]);
    ctx->r_u[id].assign(std::max<int>(d.max_bl_u, 0), HFr(0LL));
    ctx->phase_rounds = std::max<int>(d.max_bl_u, 0);
    ctx->tail_active = false;
    ctx->host_tail_active = false;
    ctx->last_poly_valid = false;
    ctx->relu_rou = H(relu_rou);
    ctx->add_term.clear();
    ctx->add_pending = false;
    ctx->round = 0;
    ctx->conv_K = 0;
    ctx->phase_kind = 1; ctx->phase_bg_cur = ctx->beta_g_cur; ctx->phase_r.clear(); ctx->phase_no_live = false;
    const HFr scale = H(d.scale);
    prep_plan P(ctx);

    if (d.ty == ZK_FFT || d.ty == ZK_IFFT) {
        const bool fwd = d.ty == ZK_FFT;
        const int fft_bl = d.fft_bit_length, fft_blh = fft_bl - 1;
        const int cnt_bl = fwd ? d.bit_length - fft_bl : d.bit_length - fft_blh;
        const uint32_t cnt_len = d.size >> (fwd ? fft_bl : fft_blh);
        fr_t *bg = ctx->beta_g[ctx->beta_g_cur];
        if (fwd) P.eq(bg, cnt_bl, ctx->r_0 + fft_bl, ctx->alpha, ctx->r_1, ctx->beta, ~0ull, HFr::one());
        else {
            P.eq1(bg, cnt_bl, ctx->r_0 + fft_blh, ctx->alpha);
            ctx->bg_r.assign(ctx->r_0 + fft_blh, ctx->r_0 + fft_blh + cnt_bl);     // (the DOT_PROD layer below splits this table: dotprod_init_phase1)
            ctx->bg_alpha = ctx->alpha;
            ctx->bg_layer = id;
        }
        if ((rc = P.launch(ctx))) return rc;
        table_pair &t = ctx->tp[1];
        const uint32_t len = (uint32_t) t.len;
        if ((rc = strided_matvec(ctx, t.V[0], prev.val, bg, len, 1u << d.max_bl_u, cnt_len))) return rc;
        if ((rc = phi_table(ctx, t.M[0], ctx->r_0, scale, fft_bl, !fwd))) return rc;
        return ZK_OK;
    }

    // where both tables of a pair end: M has no gate behind p1_live (the factored convolution: behind the tensor it reads); V is the layer-0
    // subset (size_u[0] entries) or the previous layer (non-zero up to val_live: a RELU / pooling layer's constraint rows are zero for a valid witness)
    const uint64_t live[2] = {std::min<uint64_t>(ctx->tp[0].len, d.size_u[0]),
                              std::min<uint64_t>(ctx->tp[1].len, std::max<uint64_t>(cur.p1_live[1], prev.val_live))};
    for (int b = 0; b < 2; ++b) load_v_table(ctx, P, ctx->tp[b], b, d.bit_length_u[b], d.size_u[b], cur.ori_u, prev);

    const bool padding = d.ty == ZK_PADDING;
    if (padding) {
        // beta_g[g] = beta_g_fft[g >> (n-1)] * eq(r_0[0..n-1), g & mask): the table left by the FFT layer above
        // is expanded (reference src/prover.cpp:214-219; hidden cross-layer state)
        if (d.zero_start_id < d.size) { ctx->err = "PADDING layer with constraint rows is not supported"; return ZK_ERR_ARG; }
        P.eq1(ctx->beta_gs, d.fft_bit_length - 1, ctx->r_0, HFr::one());
    } else {
        const bool tail = d.zero_start_id < d.size;
        // (gates only refer to outputs below the layer's size; the FFT-convolution layer types hand their table on to the next layer: whole table)
        const bool plain = d.ty == ZK_RELU || d.ty == ZK_MAX_POOL || d.ty == ZK_AVG_POOL || d.ty == ZK_NCONV || d.ty == ZK_FCONN;
        P.eq(ctx->beta_g[ctx->beta_g_cur], d.bit_length, ctx->r_0, ctx->alpha * scale, ctx->r_1, ctx->beta * scale,
             tail ? d.zero_start_id : ~0ull, tail ? ctx->relu_rou : HFr::one(), plain ? (uint64_t) d.size : ~0ull);
    }
    gate_job jobs[2];
    int njobs = 0;
    uint64_t conv_end = 0;
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        const uint64_t end = table_end(t.len, live[b]);
        if (b == 1 && cur.conv_ok) {
            ctx->conv_K = conv_phase1_tables(ctx, cur, ctx->alpha * scale, ctx->beta * scale);
            conv_end = end;
            if (ctx->conv_K) P.small_tables(ctx->conv_small, (const liu_table *) ctx->conv_tabs, CT_COUNT);
            else P.zero(t.M[0], end);             // both claims weighted with zero: every sum is zero
            continue;
        }
        // zeros where no gate stores: behind the covered prefix of keys, up to where the first reader stops
        if (cur.p1_cov[b] < end) P.zero(t.M[0] + cur.p1_cov[b], end - cur.p1_cov[b]);
        gate_job j = {t.M[0], cur.p1[b], cur.n_p1[b], cur.n_p1_real[b], cur.n_p1_uni[b], cur.p1_G[b], -1, t.len};
        jobs[njobs++] = j;
    }
    if ((rc = P.launch(ctx))) return rc;
    if (padding) {
        const int fft_blh = d.fft_bit_length - 1;
        fr_t *src = ctx->beta_g[ctx->beta_g_cur], *dst = ctx->beta_g[ctx->beta_g_cur ^ 1];
        zk_launch_f(ctx, PC_EQ, 0.0, dim3(grid_for(cur.val_len)), k_outer_expand_f{dst, (const fr_t *) src, (const fr_t *) ctx->beta_gs, fft_blh, cur.val_len});
        ZK_HIP(hipGetLastError());
        ctx->beta_g_cur ^= 1;
    }
    if (ctx->conv_K && (rc = conv_phase1(ctx, cur, ctx->tp[1].M[0], conv_end))) return rc;
    if ((rc = gate_multi(ctx, 1, prev, jobs, njobs, nullptr))) return rc;
    ctx->tp[0].live = live[0];
    ctx->tp[1].live = live[1];
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_dotprod_init_phase1(zk_ctx *ctx) {
    CHECK_READY();
    const int id = ctx->sumcheck_id;
    if (id < 1 || id >= (int) ctx->L.size()) return ZK_ERR_STATE;
    const dev_layer &cur = ctx->L[id], &prev = ctx->L[id - 1];
    const zk_layer_desc &d = cur.d;
    if (d.ty != ZK_DOT_PROD) return ZK_ERR_STATE;
    int32_t rc;
    const int fft_bl = d.fft_bit_length;
    reset_pairs(ctx, d.bit_length_u[1], d.bit_length_u[1]);       // V0 and V1 both span the whole FFT layer
    ctx->r_u[id].assign(d.max_bl_u, HFr(0LL));
    ctx->phase_rounds = d.bit_length_u[1];  // (the rounds behind the periodic table's collapse go through quad_round: it counts them)
    ctx->small_final_valid = false;
    ctx->add_term.clear();
    ctx->add_pending = false;
    ctx->tail_active = false;
    ctx->host_tail_active = false;
    ctx->last_poly_valid = false;
    ctx->round = 0;
    ctx->phase_kind = 0; ctx->phase_r.clear(); ctx->phase_no_live = false;
    ctx->small_len = 1u << fft_bl;
    ctx->small_cur = 0;
    ctx->dot_defer = false;
    {   // (test hook: a smaller fill threshold lets small circuits reach the deferred fold; tests/test_timed_path_gpu.py)
        const char *hooks = getenv("ZKCNN_TEST_HOOKS"), *v = getenv("ZKCNN_TEST_DOT_FILL_LOG");
        ctx->dot_fill_log = (hooks && atoi(hooks) && v) ? std::max(2, std::min(atoi(v), policy::DOT_FILL_LOG)) : policy::DOT_FILL_LOG;
    }
    const uint64_t N = ctx->tp[1].len;
    prep_plan P(ctx);
    P.eq1(ctx->small[0], fft_bl, ctx->r_0, HFr::one());
    load_v_table(ctx, P, ctx->tp[1], 1, d.bit_length_u[1], d.size_u[1], nullptr, prev);
    // the factored table (kernels.cuh: k_dot_s): the beta_g table in place is the one the IFFT layer above left, alpha * eq(bg_r, .)
    int k_lo = 0;
    while ((1u << k_lo) < cur.dot_CO) ++k_lo;
    const bool factored = cur.dot_ok && ctx->bg_layer == id + 1 && (int) ctx->bg_r.size() >= k_lo && (ctx->bg_r.size() - k_lo) <= 12 &&
                          ((uint64_t) cur.dot_pp - 1) >> (ctx->bg_r.size() - k_lo) == 0;
    // k_dot_v0 writes the rows up to the last one that has a gate, the factored kernels the pictures' rows. The rest of V0 -- the rows of the
    // weight vectors: (channel_out / (pictures + channel_out)) of the table -- is zero and is neither written nor read: the round kernel gets
    // the live prefix (k_round_cubic: x_live) and folds only Y behind it
    const uint64_t covered = std::min<uint64_t>(N, (uint64_t) (factored ? cur.dot_pp * cur.dot_CI : cur.d1_live_rows) << fft_bl);
    ctx->tp[0].live = covered;
    if ((rc = P.launch(ctx))) return rc;
    if (factored) {
        const uint32_t pp = cur.dot_pp, CO = cur.dot_CO, CI = cur.dot_CI, len = 1u << fft_bl;
        // chunks of channel_out so that a launch has ~2^18 threads even where the transforms are short
        const uint32_t want = (uint32_t) std::max<uint64_t>(1, (1ull << 18) / ((uint64_t) CI * len));
        const uint32_t per = std::max<uint32_t>(8, (CO + want - 1) / want), chunks = (CO + per - 1) / per;
        if (!ctx->dot_tabs && (rc = zk_dev_alloc(ctx, (void **) &ctx->dot_tabs, 2 * 4096 * sizeof(fr_t)))) return rc;
        if (!ctx->dot_part) {                 // the largest any DOT_PROD layer of this circuit asks for: one allocation per session
            uint64_t cap = 1;
            for (const dev_layer &D : ctx->L)
                if (D.dot_ok) {
                    const uint64_t l = 1ull << D.d.fft_bit_length, w = std::max<uint64_t>(1, (1ull << 18) / ((uint64_t) D.dot_CI * l));
                    const uint64_t pr = std::max<uint64_t>(8, (D.dot_CO + w - 1) / w);
                    cap = std::max(cap, ((D.dot_CO + pr - 1) / pr) * D.dot_CI * l);
                }
            if ((rc = zk_dev_alloc(ctx, (void **) &ctx->dot_part, cap * sizeof(fr_t)))) return rc;
            ctx->dot_part_cap = cap;
        }
        if ((uint64_t) chunks * CI * len > ctx->dot_part_cap) { ctx->err = "DOT_PROD partial sums larger than planned"; return ZK_ERR_STATE; }
        prep_plan P2(ctx);
        fr_t *lo = ctx->dot_tabs, *hi = ctx->dot_tabs + 4096;
        P2.eq1(lo, k_lo, ctx->bg_r.data(), HFr::one());
        P2.eq1(hi, (int) ctx->bg_r.size() - k_lo, ctx->bg_r.data() + k_lo, ctx->bg_alpha);
        if ((rc = P2.launch(ctx))) return rc;
        const uint32_t tiles = (len + ZK_BLOCK - 1) / ZK_BLOCK;
        zk_launch_d<k_dot_s, ZK_BLOCK>(ctx, PC_DOT, 0.0, dim3(tiles, CI * chunks), ctx->dot_part, (const fr_t *) prev.val, (const fr_t *) lo, pp, CO, CI, per, fft_bl);
        zk_launch_d<k_dot_v0s, ZK_BLOCK>(ctx, PC_DOT, 0.0, dim3(tiles, CI), ctx->tp[0].V[0], (const fr_t *) ctx->dot_part, (const fr_t *) hi, pp, CI, chunks, fft_bl);
        ZK_HIP(hipGetLastError());
    } else if (cur.d1_live_rows) {
        dim3 grid(((1u << fft_bl) + ZK_BLOCK - 1) / ZK_BLOCK, cur.d1_live_rows);
        zk_launch_d<k_dot_v0, ZK_BLOCK>(ctx, PC_DOT, 0.0, grid, ctx->tp[0].V[0], (const fr_t *) prev.val, (const fr_t *) ctx->beta_g[ctx->beta_g_cur], (const gate_rec *) cur.d1, (const uint32_t *) cur.d1_rowptr, fft_bl);
        ZK_HIP(hipGetLastError());
    }
    return ZK_OK;
}

static int32_t quad_round(zk_ctx *ctx, const HFr &r, bool with_add_term, uint64_t out_abc[12]);

extern "C" int32_t zk_sumcheck_dotprod_update1(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t out_abcd[16]) {
    CHECK_READY_ROUND();
    const int id = ctx->sumcheck_id;
    const HFr r = H(prev_r);
    const bool first = ctx->round == 0;
    if (!first) ctx->r_u[id].at(ctx->round - 1) = r;
    // Behind the collapse of the periodic table the sum is X Y m with a scalar m the host knows: a quadratic sumcheck of the pair (V = Y, M = X),
    // its polynomial times m, cubic coefficient zero (reference src/prover.cpp:103-144 keeps multiplying by the folded 1-entry table). Those rounds
    // -- all rounds on at most (pictures + channel_out) channel_in entries -- take the quadratic path: resident kernels for a lone proof, the
    // latency kernel otherwise. (Not with the device-side Fiat-Shamir chain: it would hash the unscaled three coefficients.)
    if (!ctx->dot_quad && !first && ctx->small_len == 1 && ctx->small_final_valid && !ctx->fs_state && ctx->tp[1].len >= 4 && ctx->tp[0].len == ctx->tp[1].len &&
        ctx->tp[0].live >= ctx->tp[0].len) {
        table_pair &x = ctx->tp[0], &y = ctx->tp[1];
        ctx->dot_saved_M[0] = y.M[0];
        ctx->dot_saved_M[1] = y.M[1];
        y.M[y.cur] = x.V[x.cur];
        y.M[y.cur ^ 1] = x.V[x.cur ^ 1];
        y.live = y.len;
        y.tail_valid = false;
        x.len = 0;
        ctx->dot_quad = true;
        ctx->last_poly_valid = false;
    }
    if (ctx->dot_quad) {
        uint64_t abc[12];
        int32_t rc = quad_round(ctx, r, false, abc);
        if (rc) return rc;
        put(out_abcd, HFr(0LL));
        for (int k = 0; k < 3; ++k) put(out_abcd + 4 * (k + 1), H(abc + 4 * k) * ctx->small_final);
        return ZK_OK;
    }
    ++ctx->round;
    // the periodic table is folded INSIDE the round kernel (entry i = lerp of the unfolded pair, taken on the fly; block 0 stores the folded table
    // for the next round): one launch per cubic round instead of two
    const fr_t *ms_raw = nullptr;
    fr_t *ms_out = nullptr;
    if (!first && ctx->small_len >= 2) {
        ms_raw = ctx->small[ctx->small_cur];
        ms_out = ctx->small[ctx->small_cur ^ 1];
        ctx->small_cur ^= 1;
        ctx->small_len >>= 1;
    }
    table_pair &t0 = ctx->tp[0], &t1 = ctx->tp[1];
    const uint64_t n = t1.len;
    const uint64_t npairs = first ? n / 2 : n / 4;
    if (npairs == 0) return ZK_ERR_STATE;
    const unsigned long long seq = ++ctx->slot_seq;
    // X's live prefix (entries, pre-fold); once the folded tables are small the zeros behind it are written out (the quadratic rounds read whole tables)
    uint64_t x_live = std::min<uint64_t>(t0.live, n);
    int fill = (!first && n / 2 <= (1ull << ctx->dot_fill_log)) ? 1 : 0;
    // Behind X's live prefix only Y is folded: for large tables that is a plain streaming fold (one output per thread, 64 contiguous bytes in, 32
    // out) in a launch of its own ahead of the round kernel, which then works on the live pairs alone
    uint64_t pl = std::min<uint64_t>(npairs, (x_live + 3) / 4);
    // the first fold of the phase decides: a large dead region behind an even prefix is left unfolded (policy: DEAD_*); Y must still be the layer's values in place
    // (the round kernel reads QUADS: a live prefix that is not a multiple of 4 has a last quad that reaches behind it, where a deferred Y reads as 0 --
    // round-5 advisor finding: the prefix must be a multiple of 4 to start deferring and to go on with it, not merely even)
    static const bool no_defer = getenv("ZKCNN_NO_DOT_DEFER") != nullptr;
    if (!first && ctx->round == 2 && !ctx->dot_defer && !ctx->fs_state && t1.Vsrc && npairs - pl >= (1ull << policy::DOT_STREAM_LOG) && x_live > 0 && !(x_live & 3) && !fill &&
        !no_defer) {
        ctx->dot_defer = true;
        ctx->dot_defer_Y = t1.Vsrc;
        ctx->dot_defer_to = std::min<uint64_t>(n, ctx->L[id - 1].val_live);      // (entries of the unfolded table that can be non-zero)
        ++ctx->dot_defer_count;
    } else if (ctx->dot_defer && ((x_live & 3) || fill || ctx->round - 2 >= policy::DEAD_MAX_FOLDS || x_live >= n)) {
        // catch up: `sdone` folds were skipped behind the prefix; the current tables get their entries [x_live, n) -- Y the folded values, X its zeros
        const int sdone = ctx->round - 2, cs = std::min(sdone, 12);
        const uint64_t real_rows = std::min<uint64_t>(n, (ctx->dot_defer_to + (1ull << sdone) - 1) >> sdone);
        const uint64_t rows = real_rows > x_live ? real_rows - x_live : 0, chunks = 1ull << (sdone - cs);
        int32_t rc;
        const uint64_t need = (1ull << policy::DEAD_MAX_FOLDS) + (1ull << 16);
        if (!ctx->dead_tabs && (rc = zk_dev_alloc(ctx, (void **) &ctx->dead_tabs, need * sizeof(fr_t)))) return rc;
        fr_t *E = ctx->dead_tabs, *part = ctx->dead_tabs + (1ull << policy::DEAD_MAX_FOLDS);
        fr_t *ycur = t1.V[t1.cur], *xcur = t0.V[t0.cur];
        prep_plan P(ctx);
        P.eq1(E, sdone, ctx->r_u[id].data(), HFr::one());
        if (x_live + rows < n) P.zero(ycur + x_live + rows, n - x_live - rows);
        if (x_live < n) P.zero(xcur + x_live, n - x_live);
        if ((rc = P.launch(ctx))) return rc;
        // (rows in blocks whose partial sums fit the 2^16-entry scratch: with more than 12 skipped folds a row is summed in chunks -- only the test hook gets there)
        const uint64_t blk = chunks > 1 ? std::max<uint64_t>(1, (1ull << 16) / chunks) : rows;
        for (uint64_t r0 = 0; r0 < rows; r0 += blk) {
            const uint64_t nr = std::min<uint64_t>(blk, rows - r0);
            k_dead_rows_f f;
            f.Y = ctx->dot_defer_Y; f.E = E; f.out = chunks > 1 ? part : ycur + x_live + r0;
            f.row0 = x_live + r0; f.rows = nr; f.s = sdone; f.cs = cs;
            const uint64_t waves = sdone >= 6 ? nr * chunks : (nr + (64u >> sdone) - 1) / (64u >> sdone);
            f.nblk = (uint32_t) std::min<uint64_t>((waves + ZK_BLOCK / 64 - 1) / (ZK_BLOCK / 64), 4096);
            zk_launch_f(ctx, PC_FOLD, 32.0 * (double) (nr << sdone), dim3(f.nblk), f);
            if (chunks > 1) zk_launch_f<k_sum_rows_f, 1024>(ctx, PC_FOLD, 0.0, dim3((uint32_t) ((nr + 63) / 64)), k_sum_rows_f{ycur + x_live + r0, (const fr_t *) part, (uint32_t) nr, (uint32_t) chunks});
            ZK_HIP(hipGetLastError());
        }
        ctx->dot_defer = false;
        t0.live = n;             // whole tables from here on
        x_live = n;
        pl = npairs;
    }
    if (ctx->dot_defer) fill |= 4;
    else if (!first && npairs - pl >= (1ull << policy::DOT_STREAM_LOG)) {
        const uint64_t n_in = n - 4 * pl;
        zk_launch_f(ctx, PC_FOLD, 48.0 * (double) n_in, dim3(grid_for(n_in / 2, 8192)), k_fold_f{vin(t1) + 4 * pl, t1.V[t1.cur ^ 1] + 2 * pl, n_in, to_dev(r)});
        ZK_ORDER();
        if (fill) ZK_HIP(hipMemsetAsync(t0.V[t0.cur ^ 1] + 2 * pl, 0, (n / 2 - 2 * pl) * sizeof(fr_t), ctx->stream));
        fill |= 2;
    }
    const uint32_t g = std::min<uint32_t>(grid_for(((fill & 2) || (fill & 4 && !(fill & 1))) || first ? std::max<uint64_t>(first ? std::min<uint64_t>(npairs, (x_live + 1) / 2) : pl, 1) : npairs, 1024), ctx->partial_blocks);
    zk_launch_d<k_round_cubic, ZK_BLOCK>(ctx, PC_ROUND_CUBIC, (first ? 32.0 : 48.0) * (double) ((fill & 4) ? 2 * x_live : n + x_live), dim3(g), (const fr_t *) t0.V[t0.cur], vin(t1),
              t0.V[t0.cur ^ 1], t1.V[t1.cur ^ 1], (const fr_t *) ctx->small[ctx->small_cur], ctx->small_len, n, to_dev(r), first ? 1 : 0, ctx->partials,
              ctx->d_counter, (host_slot *) ctx->d_slot, seq, ms_raw, ms_out, x_live, fill);
    ZK_HIP(hipGetLastError());
    if (!first) {
        t0.cur ^= 1; t1.cur ^= 1;
        t0.len >>= 1; t1.len >>= 1;
        t1.Vsrc = nullptr;
        t0.live = (fill & 1) ? t0.len : std::min<uint64_t>(t0.len, 2 * ((x_live + 3) / 4));
    }
    int32_t rc = wait_slot(ctx, seq);
    if (rc) return rc;
    for (int k = 0; k < 4; ++k) ctx->h_result[k] = ctx->h_slot->v[k];
    if (ms_raw && ctx->small_len == 1) { ctx->small_final = ctx->h_slot->v[4]; ctx->small_final_valid = true; }
    for (int k = 0; k < 4; ++k) put(out_abcd + 4 * k, ctx->h_result[k]);
    ctx->proof_size += 32 * (3 + (ctx->h_result[0].isZero() ? 0 : 1));
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_dotprod_finalize1(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t claim_1[4]) {
    CHECK_READY();
    const int id = ctx->sumcheck_id;
    const HFr r = H(prev_r);
    if (ctx->round < 1) return ZK_ERR_STATE;
    ctx->r_u[id].at(ctx->round - 1) = r;
    table_pair &t1 = ctx->tp[1];
    int32_t rc;
    if (ctx->dot_quad) {               // the phase is over: pair 1 gets its own M buffers back
        t1.M[0] = ctx->dot_saved_M[0];
        t1.M[1] = ctx->dot_saved_M[1];
        ctx->dot_quad = false;
        if (t1.len == 2 && t1.tail_valid && ctx->small_len == 1) {          // Y's last pair came along with the last round: O(1) on the host
            ctx->h_result[0] = t1.tail_v[0] + r * (t1.tail_v[1] - t1.tail_v[0]);
            ctx->h_result[1] = ctx->small_final;
            t1.len = 0;
            put(claim_1, ctx->h_result[0]);
            ctx->V_u1 = ctx->h_result[0] * ctx->h_result[1];
            ctx->vu1_scale = ctx->h_result[1];
            ctx->proof_size += 32;
            return ZK_OK;
        }
    }
    eval_args E;
    std::memset(&E, 0, sizeof(E));
    E.p[0] = vin(t1); E.n[0] = (uint32_t) std::min<uint64_t>(t1.len, 2);
    E.p[1] = ctx->small[ctx->small_cur]; E.n[1] = std::min<uint32_t>(ctx->small_len, 2);
    if (t1.len > 2 || ctx->small_len > 2) return ZK_ERR_STATE;
    E.r = to_dev(r);
    E.slot = (host_slot *) ctx->d_slot;
    E.seq = ++ctx->slot_seq;
    zk_launch_f<k_eval_pairs_f, 64>(ctx, PC_FOLD, 0.0, dim3(1), k_eval_pairs_f{E});
    ZK_HIP(hipGetLastError());
    if ((rc = wait_slot(ctx, E.seq))) return rc;
    ctx->h_result[0] = ctx->h_slot->v[0];
    ctx->h_result[1] = ctx->h_slot->v[1];
    t1.len = 0;
    ctx->small_len = 1;
    put(claim_1, ctx->h_result[0]);
    ctx->V_u1 = ctx->h_result[0] * ctx->h_result[1];
    ctx->vu1_scale = ctx->h_result[1];
    ctx->proof_size += 32;
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_init_phase2(zk_ctx *ctx) {
    CHECK_READY();
    const int id = ctx->sumcheck_id;
    if (id < 1 || id >= (int) ctx->L.size()) return ZK_ERR_STATE;
    const dev_layer &cur = ctx->L[id], &prev = ctx->L[id - 1];
    const zk_layer_desc &d = cur.d;
    int32_t rc;
    reset_pairs(ctx, d.bit_length_v[0], d.bit_length_v[1]);
    ctx->r_v[id].assign(std::max<int>(d.max_bl_v, 0), HFr(0LL));
    ctx->phase_rounds = std::max<int>(d.max_bl_v, 0);
    ctx->tail_active = false;
    ctx->host_tail_active = false;
    ctx->last_poly_valid = false;
    ctx->add_term.clear();
    ctx->add_pending = false;
    ctx->round = 0;
    ctx->phase_kind = 2; ctx->phase_bg_cur = ctx->beta_g_cur; ctx->phase_r.clear(); ctx->phase_no_live = false;
    const HFr *ru = ctx->r_u[id].data();
    prep_plan P(ctx);

    if (d.ty == ZK_DOT_PROD) {
        const int fft_bl = d.fft_bit_length, cnt_bl = d.max_bl_v;
        table_pair &t = ctx->tp[1];
        const uint32_t rows = d.size_v[1];
        P.eq1(ctx->beta_u, cnt_bl, ru + fft_bl, HFr::one());
        P.eq1(ctx->beta_gs, fft_bl, ru, HFr::one());
        if (rows < t.len) P.zero(t.V[0] + rows, t.len - rows);              // (k_row_dot writes one entry per row)
        if (cur.p2_cov[1] < t.len) P.zero(t.M[0] + cur.p2_cov[1], t.len - cur.p2_cov[1]);
        if ((rc = P.launch(ctx))) return rc;
        zk_launch_d<k_row_dot, ZK_BLOCK>(ctx, PC_DOT, 0.0, dim3((rows + 3) / 4), t.V[0], (const fr_t *) prev.val, (const fr_t *) ctx->beta_gs, rows, fft_bl);
        ZK_HIP(hipGetLastError());
        gate_job j = {t.M[0], cur.p2[1], cur.n_p2[1], cur.n_p2_real[1], 0, cur.p2_G[1], cur.p2_uniform[1], t.len};
        return gate_multi(ctx, 2, prev, &j, 1, nullptr);
    }

    const uint64_t live[2] = {std::min<uint64_t>(ctx->tp[0].len, d.size_v[0]),
                              std::min<uint64_t>(ctx->tp[1].len, std::max<uint64_t>(cur.p2_live[1], prev.val_live))};
    // (the u operands of the layer's gates: the layer-0 subset and / or the previous layer, both referred to below their sizes)
    P.eq1(ctx->beta_u, d.max_bl_u, ru, HFr::one(), std::max<uint64_t>(d.size_u[0], d.bit_length_u[1] >= 0 ? prev.d.size : 0));
    for (int b = 0; b < 2; ++b) load_v_table(ctx, P, ctx->tp[b], b, d.bit_length_v[b], d.size_v[b], cur.ori_v, prev);
    gate_job jobs[2];
    int njobs = 0;
    uint64_t conv_end = 0;
    bool conv = false;
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        const uint64_t end = table_end(t.len, live[b]);
        if (b == 0 && cur.conv_ok && ctx->conv_K > 0) {
            conv_phase2_tables(ctx, cur, ru);
            P.small_tables(ctx->conv_small, (const liu_table *) ctx->conv_tabs + CT_COUNT, CT_COUNT);
            conv_end = end;
            conv = true;
            continue;
        }
        if (cur.p2_cov[b] < end) P.zero(t.M[0] + cur.p2_cov[b], end - cur.p2_cov[b]);
        gate_job j = {t.M[0], cur.p2[b], cur.n_p2[b], cur.n_p2_real[b], 0, cur.p2_G[b], cur.p2_uniform[b], t.len};
        jobs[njobs++] = j;
    }
    if ((rc = P.launch(ctx))) return rc;
    if (conv && (rc = conv_phase2(ctx, cur, ctx->tp[0].M[0], conv_end))) return rc;
    // both scatters and the constant term sum_{uni gates} beta_g beta_u two_mul (reference src/prover.cpp:297-300) in one launch
    if ((rc = gate_multi(ctx, 2, prev, jobs, njobs, &cur))) return rc;
    ctx->tp[0].live = live[0];
    ctx->tp[1].live = live[1];
    return ZK_OK;
}

// add_term of a phase 2 whose two sums are still on their way (zk_sumcheck_init_phase2): waits for the mapped slot like wait_slot does
static int32_t resolve_add_term(zk_ctx *ctx) {
    if (!ctx->add_pending) return ZK_OK;
    if (ctx->batch) { int32_t rc = zk_batch_sync_point(ctx); if (rc) return rc; }
    volatile unsigned long long *p = &ctx->h_aux->seq;
    for (uint64_t spins = 0; *p != ctx->aux_seq; ++spins) {
        if (spins > (1ull << 21)) {
            ZK_HIP(hipStreamSynchronize(ctx->stream));
            if (*p != ctx->aux_seq) { ctx->err = "add_term sums were not published"; return ZK_ERR_STATE; }
            break;
        }
        if (spins >= 4096) sched_yield();
        else __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    ctx->add_term = ctx->V_u0 * ctx->h_aux->v[0] + ctx->V_u1 * ctx->h_aux->v[1];
    ctx->add_pending = false;
    return ZK_OK;
}

// the resident / device-side / hybrid round drivers (everything that is not "one launch per round")
#include "rounds_resident.hpp"

static int32_t quad_round_once(zk_ctx *ctx, const HFr &r, bool with_add_term, uint64_t out_abc[12]);

// A phase whose resident kernel was lost (it gave up after its time-out, or never got the CUs it waits for) is run AGAIN from its initialisation with
// a launch per round: the challenges answered so far are replayed (same field elements, their polynomials are dropped), then the round that was
// asked for is computed. Everything the initialisation needs is still in the context (the claim points, alpha / beta, V_u0 / V_u1, the weights of
// the layer-0 combine); the tables of the phase are rebuilt from the layer values, which no round touches.
static int32_t replay_phase(zk_ctx *ctx, const HFr &r, bool with_add_term, uint64_t out_abc[12]) {
    const int kind = ctx->phase_kind, id = ctx->sumcheck_id;
    if (kind < 1 || kind > 3) return ZK_ERR_STATE;
    const std::vector<HFr> done = ctx->phase_r;                 // challenges of the calls so far (the current call's r is the last of them; none: the phase's first call)
    if (done.size() != (size_t) ctx->round || (!done.empty() && std::memcmp(&done.back(), &r, 32) != 0)) return ZK_ERR_STATE;
    const uint64_t proof_size = ctx->proof_size;
    const std::vector<HFr> ru = ctx->r_u[id], rv = ctx->r_v[id];
    const std::string why = ctx->err;
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipMemsetAsync(ctx->d_counter, 0, 64, ctx->stream);
    (void) hipMemsetAsync(ctx->d_bcast, 0, sizeof(mid_bcast), ctx->stream);
    std::memset(ctx->h_live_in, 0, sizeof(live_in));
    ctx->live_active = false;
    ctx->live_mid = false;
    ctx->beta_g_cur = ctx->phase_bg_cur;
    int32_t rc;
    if (kind == 1) { const HFr rr = ctx->relu_rou; rc = zk_sumcheck_init_phase1(ctx, reinterpret_cast<const uint64_t *>(&rr)); }
    else if (kind == 2) rc = zk_sumcheck_init_phase2(ctx);
    else rc = zk_sumcheck_liu_init(ctx, ctx->phase_sig_u.data(), ctx->phase_sig_v.data(), (uint32_t) (ctx->phase_sig_u.size() / 4));
    if (rc) return rc;
    ctx->phase_no_live = true;
    ctx->r_u[id] = ru;
    ctx->r_v[id] = rv;
    uint64_t dummy[12];
    // calls 0 .. n-1 were answered before (call j came with challenge c_j, call 0 with none); `done` = c_1 .. c_n, c_n being the call in progress
    for (size_t j = 0; j < done.size() && rc == ZK_OK; ++j) {
        const HFr prev = j ? done[j - 1] : HFr(0LL);
        rc = quad_round_once(ctx, prev, with_add_term, dummy);
    }
    if (rc == ZK_OK) rc = quad_round_once(ctx, r, with_add_term, out_abc);
    ctx->phase_r = done;
    ctx->proof_size = proof_size + 32 * 3;
    ++ctx->live_fallbacks;
    if (rc == ZK_OK) fprintf(stderr, "[zkcnn] a resident round kernel was lost (%s); the phase was run again with a launch per round\n", why.c_str());
    return rc;
}

static int32_t quad_round(zk_ctx *ctx, const HFr &r, bool with_add_term, uint64_t out_abc[12]) {
    if (ctx->round) ctx->phase_r.push_back(r);
    const uint64_t size_before = ctx->proof_size;        // (the failed call may or may not have booked its message: the replay books it once)
    int32_t rc = quad_round_once(ctx, r, with_add_term, out_abc);
    if (rc == ZK_ERR_STATE && ctx->live_lost) {
        ctx->live_lost = false;
        if (ctx->phase_kind == 0) {
            // the quadratic rounds behind a DOT_PROD phase's collapse: their tables are the cubic rounds' output, which replay_phase cannot rebuild
            ctx->err = "a resident round kernel was lost in a DOT_PROD phase: no fallback for this phase kind (prove again)";
            return ZK_ERR_STATE;
        }
        ctx->proof_size = size_before;
        rc = replay_phase(ctx, r, with_add_term, out_abc);
    }
    return rc;
}

static int32_t quad_round_once(zk_ctx *ctx, const HFr &r, bool with_add_term, uint64_t out_abc[12]) {
    static const bool seg_timing = getenv("ZKCNN_TIMING") != nullptr;
    // the tail paths start from add_term: it must be there before they are considered (the plain round below picks it up behind its launch)
    // A lane of a batch with a driver (zk_batch_set_yield) and no explicit tail setting: the rounds on at most TAIL_QUADS quads run in a resident
    // workgroup of ONE fused launch per phase (fs_tail.cuh: k_tail_live_f) instead of a fused launch per round plus the hybrid tail on the host.
    // ZKCNN_LANE_RESIDENT=0/1 overrides policy::LANE_RESIDENT_TAIL.
    static const bool lane_resident = [] { const char *v = getenv("ZKCNN_LANE_RESIDENT"); return v ? atoi(v) != 0 : policy::LANE_RESIDENT_TAIL; }();
    const bool lane_tail = lane_resident && ctx->batch && ctx->batch->yield_fn && ctx->host_tail_log == -1 && ctx->live_rounds && !ctx->fs_state && !ctx->phase_no_live;
    const int ht_log = lane_tail ? -1 : host_tail_log(ctx);
    if (ctx->add_pending && (ht_log >= 0 || ctx->fs_state)) { int32_t rc0 = resolve_add_term(ctx); if (rc0) return rc0; }
    if (ht_log >= 0 && !ctx->host_tail_active && !ctx->tail_active && ctx->phase_rounds > ctx->round &&
        std::max(ctx->tp[0].len, ctx->tp[1].len) <= (1ull << ht_log) && ctx->tp[0].len + ctx->tp[1].len > 0) {
        int32_t rc = host_tail_begin(ctx);
        if (rc) return rc;
    }
    if (ctx->host_tail_active) {
        host_tail_round(ctx, r, with_add_term, out_abc);
        return ZK_OK;
    }
    // quads of this round over both pairs (a first round works on pairs, not quads)
    const uint64_t round_quads = ctx->round == 0 ? (ctx->tp[0].len + ctx->tp[1].len) / 2 : (ctx->tp[0].len + ctx->tp[1].len) / 4;
    const bool in_phase = !ctx->live_active && !ctx->tail_active && ctx->phase_rounds > ctx->round && ctx->tp[0].len + ctx->tp[1].len > 0;
    // (a context with a hybrid tail -- an explicit zk_set_host_tail, the zero-knowledge mode -- still takes the resident SEGMENTS of a phase's middle: they end
    //  above the tail's hand-over size with the tables in device memory; only the single-workgroup kernel that runs a phase to its end is the tail's alternative)
    const bool mid_ok = ctx->live_rounds && ctx->live_now && !ctx->batch && !ctx->phase_no_live && in_phase &&
                        (ctx->counted_active || zk_contexts_on_device(ctx->device) == 1);     // (outside a proof's bracket: only for the device's one context)
    const bool resident_ok = mid_ok && ht_log < 0;
    // The middle of a phase (more quads than the single-workgroup kernel takes, tables of at most 2^18 entries): a SEGMENT of rounds in one
    // resident multi-workgroup kernel (k_mid), as long as no table collapses or reaches its last pair. Returns the segment's length (0: none).
    auto plan_segment = [&](bool allowed) -> int {
        if (!allowed || round_quads <= TAIL_QUADS || round_quads > policy::MID_MAX_QUADS || std::max(ctx->tp[0].len, ctx->tp[1].len) > (1ull << ZK_FULL_TABLE_LOG)) return 0;
        uint64_t L0 = ctx->tp[0].len, L1 = ctx->tp[1].len;
        int rounds = 0;
        bool f = ctx->round == 0;          // (a first round works on pairs and folds nothing)
        while (ctx->round + rounds < ctx->phase_rounds && (L0 + L1) / (f ? 2 : 4) > TAIL_QUADS && (!L0 || L0 >= (f ? 4u : 8u)) && (!L1 || L1 >= (f ? 4u : 8u))) {
            ++rounds;
            if (!f) { L0 >>= 1; L1 >>= 1; }
            f = false;
        }
        return rounds >= 2 ? rounds : 0;
    };
    if (ctx->fs_state) {
        // non-interactive mode: the device derives the challenges itself -- a segment (k_mid<false>), or all remaining rounds of the phase once
        // the tables are small (k_tail<false>: it takes over at TAIL_QUADS quads); the host answers the following calls from the record
        if (in_phase && *ctx->fs_pending == 0) {
            int32_t rc = ZK_OK;
            if (const int rounds = plan_segment(resident_ok)) rc = run_device_mid(ctx, r, with_add_term, rounds, (uint32_t) std::min<uint64_t>((round_quads + 63) / 64, MID_MAX_BLOCKS));
            else if (round_quads <= TAIL_QUADS) rc = run_device_rounds(ctx, r, with_add_term);
            if (rc) return rc;
        }
    } else if (mid_ok) {
        // interactive protocol: the same two kernels, trading polynomials and challenges with the verifier through mailboxes
        int32_t rc = ZK_OK;
        if (const int rounds = plan_segment(mid_ok)) {
            rc = resolve_add_term(ctx);
            if (!rc) rc = mid_start(ctx, r, with_add_term, rounds, (uint32_t) std::min<uint64_t>((round_quads + 63) / 64, MID_MAX_BLOCKS));
        } else if (resident_ok && round_quads <= TAIL_QUADS) {
            // (a context of its own with the default tail setting: the resident kernel hands the tables to the host at 2^SOLO_EXPORT_LOG entries -- the last rounds
            //  of a phase cost the host ~1 us each, a mailbox round trip ~10; ZKCNN_SOLO_EXPORT=<log, -1: never> overrides the policy)
            static const int solo_export = [] { const char *v = getenv("ZKCNN_SOLO_EXPORT"); return v ? atoi(v) : policy::SOLO_EXPORT_LOG; }();
            rc = resolve_add_term(ctx);
            if (!rc) rc = live_start(ctx, r, with_add_term, ctx->host_tail_log == -1 ? solo_export : -1);
        }
        if (rc) return rc;
    } else if (lane_tail && in_phase && round_quads <= TAIL_QUADS) {
        int32_t rc = resolve_add_term(ctx);
        if (!rc) rc = live_start(ctx, r, with_add_term);
        if (rc) return rc;
    }
    if (ctx->live_active) return live_round(ctx, r, out_abc);
    if (ctx->tail_active) {
        const tail_out *o = (const tail_out *) ctx->h_tail;
        const int k = ctx->tail_cursor;
        if (k > 0 && std::memcmp(&r, &o->chal[k - 1], 32) != 0) {
            ctx->err = "Fiat-Shamir chain diverged: the verifier's challenge is not the one derived on the device";
            return ZK_ERR_STATE;
        }
        std::memcpy(out_abc, &o->poly[k][0], 96);
        ++ctx->round;
        ctx->proof_size += 32 * 3;
        if (++ctx->tail_cursor == ctx->tail_count) ctx->tail_active = false;
        return ZK_OK;
    }
    const double ts_enter = seg_timing ? now_s() : 0;
    double ts_wait0 = 0, ts_wait1 = 0;
    const bool first = ctx->round == 0;
    ++ctx->round;
    const bool add_deferred = ctx->add_pending;
    if (with_add_term && !add_deferred) ctx->add_term = ctx->add_term * (HFr::one() - r);
    if (with_add_term) scale_absorbed(ctx, r);
    bool collapsed[2] = {false, false};
    round2_args A;
    std::memset(&A, 0, sizeof(A));
    double alg_bytes = 0;
    // From the second round of a phase on, the claim this round's polynomial must meet is known: p(0) + p(1) = p_prev(r). The kernels then
    // skip the sum of v1 m1 (one product per pair of 7) and b follows from the claim -- the same field element the direct sum gives.
    const bool skip_p1 = !first && ctx->last_poly_valid;
    HFr claim(0LL);
    if (skip_p1) claim = (ctx->last_poly[0] * r + ctx->last_poly[1]) * r + ctx->last_poly[2];
    // small tables are latency bound: spread each quad over 4 lanes (k_round_quad2, `fine`)
    // (experiment switch: ZKCNN_TEST_HOOKS=1 ZKCNN_TEST_FINE_LOG=<14..18> moves the switch-over between the latency kernel and the streaming kernel)
    static const int fine_log = [] {
        const char *hooks = getenv("ZKCNN_TEST_HOOKS"), *v = getenv("ZKCNN_TEST_FINE_LOG");
        return (hooks && atoi(hooks) && v) ? std::max(12, std::min(atoi(v), ZK_FULL_TABLE_LOG)) : policy::FINE_LOG;
    }();
    // (one block per 64 quads and 3 partial sums per block: both pairs together must stay within the partials buffer)
    const uint64_t longest = std::max(ctx->tp[0].len, ctx->tp[1].len);
    const bool fine = longest <= (1ull << fine_log) && 2 * (longest / 4 / (ZK_BLOCK / 4) + 1) <= ctx->partial_blocks;
    A.fine = fine ? 1 : 0;
    uint64_t fine_items = 0;
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        A.Vin[b] = vin(t); A.Min[b] = t.M[t.cur];
        A.Vout[b] = t.V[t.cur ^ 1]; A.Mout[b] = t.M[t.cur ^ 1];
        A.n[b] = t.len;
        collapsed[b] = (first && t.len == 1) || (!first && t.len == 2);     // the reference's `total == 1` case
        const uint64_t npairs = first ? t.len / 2 : t.len / 4;
        // the large-table kernel skips the pairs behind the live prefix; once this table's output is small enough for the other kernels
        // (which read whole tables) the zeros are stored
        const uint64_t live = std::min(t.live, t.len);
        A.nl[b] = live;
        A.fill[b] = (t.len / 2 <= (1ull << ZK_FULL_TABLE_LOG)) ? 1 : 0;
        const uint64_t work = fine ? npairs : std::max<uint64_t>(first ? (live + 1) / 2 : (live + 3) / 4, first ? 1 : (A.fill[b] ? npairs / 8 : 1));
        // the streaming kernel: a block's four waves take 4 x 64 lane pairs per step = 128 quads (a first round: 128 "quads" = 256 pairs, two steps)
        // A lane of a batch shares the launch with the other lanes' blocks: 2 resident sets over all lanes (at least 128 blocks each), so that a wave takes
        // many tiles and the prologue -- first tile's latency, the 512-bit reductions, block totals -- is paid once per 2^14+ quads, not per 2^7
        const uint32_t lanes = ctx->batch ? (uint32_t) std::max<size_t>(ctx->batch->lanes.size(), 1) : 1;
        const uint32_t block_cap = lanes > 1 ? std::max<uint32_t>(policy::QUAD2_LANE_MIN_BLOCKS, 2 * policy::QUAD2_BLOCKS / lanes) : policy::QUAD2_BLOCKS;
        const uint32_t stream_blocks = (uint32_t) std::min<uint64_t>((work + policy::QUAD2_QUADS_PER_BLOCK - 1) / policy::QUAD2_QUADS_PER_BLOCK, block_cap);
        A.blocks[b] = collapsed[b] ? 1 : std::min<uint32_t>(fine ? grid_for(work, 1024) : std::max<uint32_t>(stream_blocks, 1), ctx->partial_blocks / 2);
        fine_items += collapsed[b] ? 1 : npairs;
        alg_bytes += (first ? 64.0 : 96.0) * (double) (fine ? t.len : live);      // entries the launch has to read: the live prefix
    }
    if (A.blocks[0] + A.blocks[1] == 0) {
        for (int k = 0; k < 3; ++k) ctx->h_result[k].clear();
    } else {
        A.r = to_dev(r);
        A.first = first ? 1 : 0;
        A.skip_p1 = skip_p1 ? 1 : 0;
        A.partials = ctx->partials;
        A.counter = ctx->d_counter;
        A.slot = (host_slot *) ctx->d_slot;
        A.seq = ++ctx->slot_seq;
        // (a lane of a batch defers the launch: one launch runs this round of every lane -- launch.cuh; wait_slot below hands over to the driver)
        if (fine) {
            const uint32_t blocks = (uint32_t) ((fine_items + ZK_BLOCK / 4 - 1) / (ZK_BLOCK / 4));
            zk_launch_f(ctx, PC_ROUND_FINE, alg_bytes, dim3(blocks), k_round_fine_f{A});      // (its own profiler class: the latency kernel of the small rounds)
        } else if (first) zk_launch_f(ctx, PC_ROUND_QUAD, alg_bytes, dim3(A.blocks[0] + A.blocks[1]), k_round_quad2_f<RQ_FIRST>{A});
        else if (skip_p1) zk_launch_f(ctx, PC_ROUND_QUAD, alg_bytes, dim3(A.blocks[0] + A.blocks[1]), k_round_quad2_f<RQ_FOLD>{A});
        else zk_launch_f(ctx, PC_ROUND_QUAD, alg_bytes, dim3(A.blocks[0] + A.blocks[1]), k_round_quad2_f<RQ_FOLD_P1>{A});
        ZK_HIP(hipGetLastError());
        for (int b = 0; b < 2; ++b) {
            table_pair &t = ctx->tp[b];
            if (t.len && !first) {
                t.cur ^= 1; t.len >>= 1; t.Vsrc = nullptr;
                t.live = (fine || A.fill[b]) ? std::min(t.len, 2 * ((A.nl[b] + 3) / 4)) : std::min(t.len, 2 * ((A.nl[b] + 3) / 4));
            }
        }
        if (seg_timing) ts_wait0 = now_s();
        int32_t rc = wait_slot(ctx, A.seq);
        if (rc) return rc;
        if (seg_timing) ts_wait1 = now_s();
        for (int k = 0; k < 8; ++k) ctx->h_result[k] = ctx->h_slot->v[k];
        for (int b = 0; b < 2; ++b) {          // a V table that is down to its last pair came along (k_round_quad_fine)
            table_pair &t = ctx->tp[b];
            if (fine && !collapsed[b] && t.len == 2) {
                t.tail_v[0] = ctx->h_slot->v[8 + 2 * b];
                t.tail_v[1] = ctx->h_slot->v[9 + 2 * b];
                t.tail_valid = true;
            }
        }
    }
    if (add_deferred) {            // (its kernels ran before this round's: the slot is there by now)
        int32_t rc1 = resolve_add_term(ctx);
        if (rc1) return rc1;
        if (with_add_term) ctx->add_term = ctx->add_term * (HFr::one() - r);
    }
    HFr a = ctx->h_result[0], c = ctx->h_result[1], p1 = ctx->h_result[2];
    HFr bcoef = p1 - a - c;
    for (int b = 0; b < 2; ++b)
        if (collapsed[b]) {
            table_pair &t = ctx->tp[b];
            t.final_v = ctx->h_result[4 + 2 * b];
            ctx->add_term = ctx->add_term + t.final_v * ctx->h_result[5 + 2 * b];
            t.abs_m = ctx->h_result[5 + 2 * b];
            t.abs_m_valid = true;
            t.absorbed = true;
            t.len = 0;
        }
    if (with_add_term) {        // + add_term * (1 - x)
        bcoef = bcoef - ctx->add_term;
        c = c + ctx->add_term;
    }
    if (skip_p1) bcoef = claim - c - c - a;           // p(0) + p(1) = c + (a + b + c) = claim
    ctx->last_poly[0] = a; ctx->last_poly[1] = bcoef; ctx->last_poly[2] = c;
    ctx->last_poly_valid = true;
    put(out_abc, a);
    put(out_abc + 4, bcoef);
    put(out_abc + 8, c);
    ctx->proof_size += 32 * 3;
    if (seg_timing && ts_wait0 > 0) {
        const double ts_exit = now_s();
        if (!first && ctx->t_last_exit > 0) { ctx->t_seg[0] += ts_enter - ctx->t_last_exit; ++ctx->n_seg; }
        ctx->t_seg[1] += ts_wait0 - ts_enter;
        ctx->t_seg[2] += ts_wait1 - ts_wait0;
        ctx->t_seg[3] += ts_exit - ts_wait1;
        ctx->t_last_exit = ts_exit;
    }
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_update1(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t out_abc[12]) {
    CHECK_READY_ROUND();
    if (ctx->round) ctx->r_u[ctx->sumcheck_id].at(ctx->round - 1) = H(prev_r);
    return quad_round(ctx, H(prev_r), true, out_abc);
}
extern "C" int32_t zk_sumcheck_update2(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t out_abc[12]) {
    CHECK_READY_ROUND();
    if (ctx->round) ctx->r_v[ctx->sumcheck_id].at(ctx->round - 1) = H(prev_r);
    return quad_round(ctx, H(prev_r), true, out_abc);
}
extern "C" int32_t zk_sumcheck_liu_update(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t out_abc[12]) {
    CHECK_READY_ROUND();
    return quad_round(ctx, H(prev_r), false, out_abc);
}

// final evaluations of the two V tables (reference src/prover.cpp:459-485)
static int32_t final_claims(zk_ctx *ctx, const HFr &r, const int8_t bl[2], HFr out[2]) {
    bool pending[2] = {false, false};
    eval_args E;
    std::memset(&E, 0, sizeof(E));
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        out[b].clear();
        if (t.len > 2) return ZK_ERR_STATE;
        if (t.len == 2 && t.tail_valid) {
            out[b] = t.tail_v[0] + r * (t.tail_v[1] - t.tail_v[0]);       // O(1): the pair arrived with the last round
        } else if (t.len >= 1) {
            E.p[b] = vin(t);
            E.n[b] = (uint32_t) t.len;
            pending[b] = true;
        } else if (t.absorbed && bl[b] >= 0) out[b] = t.final_v;
    }
    if (pending[0] || pending[1]) {
        E.r = to_dev(r);
        E.slot = (host_slot *) ctx->d_slot;
        E.seq = ++ctx->slot_seq;
        zk_launch_f<k_eval_pairs_f, 64>(ctx, PC_FOLD, 0.0, dim3(1), k_eval_pairs_f{E});
        ZK_HIP(hipGetLastError());
        int32_t rc = wait_slot(ctx, E.seq);
        if (rc) return rc;
        for (int b = 0; b < 2; ++b) if (pending[b]) out[b] = ctx->h_slot->v[b];
    }
    for (int b = 0; b < 2; ++b) ctx->tp[b].len = 0;
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_finalize1(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t claim_0[4], uint64_t claim_1[4]) {
    CHECK_READY();
    const int id = ctx->sumcheck_id;
    if (ctx->round < 1) return ZK_ERR_STATE;
    ctx->r_u[id].at(ctx->round - 1) = H(prev_r);
    HFr c[2];
    int32_t rc = final_claims(ctx, H(prev_r), ctx->L[id].d.bit_length_u, c);
    if (rc) return rc;
    ctx->V_u0 = c[0];
    ctx->V_u1 = c[1];
    ctx->vu1_scale = HFr::one();
    put(claim_0, c[0]);
    put(claim_1, c[1]);
    ctx->proof_size += 32 * 2;
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_finalize2(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t claim_0[4], uint64_t claim_1[4]) {
    CHECK_READY();
    const int id = ctx->sumcheck_id;
    if (ctx->round < 1) return ZK_ERR_STATE;
    ctx->r_v[id].at(ctx->round - 1) = H(prev_r);
    HFr c[2];
    int32_t rc = final_claims(ctx, H(prev_r), ctx->L[id].d.bit_length_v, c);
    if (rc) return rc;
    put(claim_0, c[0]);
    put(claim_1, c[1]);
    ctx->proof_size += 32 * 2;
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_liu_init(zk_ctx *ctx, const uint64_t *s_u, const uint64_t *s_v, uint32_t n) {
    CHECK_READY();
    if (n + 1 != ctx->L.size()) return ZK_ERR_ARG;
    ctx->sumcheck_id = 0;
    const dev_layer &L0 = ctx->L[0];
    reset_pairs(ctx, -1, L0.d.bit_length);
    ctx->r_u[0].assign(L0.d.bit_length, HFr(0LL));
    ctx->phase_rounds = L0.d.bit_length;
    ctx->tail_active = false;
    ctx->host_tail_active = false;
    ctx->last_poly_valid = false;
    ctx->add_term.clear();
    ctx->add_pending = false;
    ctx->round = 0;
    ctx->phase_kind = 3; ctx->phase_r.clear(); ctx->phase_no_live = false;
    if (s_u != ctx->phase_sig_u.data()) {           // (not when the phase is being replayed from these very vectors)
        ctx->phase_sig_u.assign(s_u, s_u + 4 * (size_t) n);
        ctx->phase_sig_v.assign(s_v, s_v + 4 * (size_t) n);
    }
    table_pair &t = ctx->tp[1];
    t.Vsrc = L0.val;
    t.live = std::min<uint64_t>(t.len, L0.d.size);          // layer 0 is zero padded behind its size, and no layer refers to entries there
    // descriptors of every (layer, side) table: point, sigma, bit split (mapped host memory, read by the kernel in place); then two launches
    liu_table *T = (liu_table *) ctx->h_liu_tabs;
    for (uint32_t tb = 0; tb < ctx->liu_ntabs; ++tb) {
        const int i = ctx->liu_tab_layer[tb], side = ctx->liu_tab_side[tb];
        const dev_layer &Li = ctx->L[i];
        const int bl = side ? Li.d.bit_length_v[0] : Li.d.bit_length_u[0];
        const std::vector<HFr> &pt = side ? ctx->r_v[i] : ctx->r_u[i];
        if ((int) pt.size() < bl) return ZK_ERR_STATE;
        for (int j = 0; j < bl; ++j) T[tb].r.v[j] = to_dev(pt[j]);
        T[tb].init = to_dev(H((side ? s_v : s_u) + 4 * (i - 1)));
        T[tb].n = bl;
        T[tb].fh = bl >> 1;
        T[tb].sh = bl - (bl >> 1);
        T[tb].pad_ = 0;
    }
    if (ctx->liu_ntabs)
        zk_launch_f<k_eq_halves_multi_f, 1024>(ctx, PC_EQ, 0.0, dim3(2, ctx->liu_ntabs), k_eq_halves_multi_f{ctx->liu_halves, (const liu_table *) ctx->liu_tabs});
    zk_launch_f(ctx, PC_LIU, 0.0, dim3(grid_for(t.len)), k_liu_gather_f{t.M[0], (const uint32_t *) ctx->liu_ptr, (const liu_entry *) ctx->liu_ent, (const fr_t *) ctx->liu_halves, t.len});
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_liu_finalize(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t claim_1[4]) {
    CHECK_READY();
    if (ctx->round < 1) return ZK_ERR_STATE;
    ctx->r_u[0].at(ctx->round - 1) = H(prev_r);
    HFr c[2];
    const int8_t bl[2] = {-1, ctx->L[0].d.bit_length};
    int32_t rc = final_claims(ctx, H(prev_r), bl, c);
    if (rc) return rc;
    put(claim_1, c[1]);
    ctx->proof_size += 32;
    return ZK_OK;
}

// ---- zero-knowledge mode (host/zk_mask.hpp (2)): the masks of the evaluation claims vanish on the cube, so no kernel and no table changes; the host adds
// Z's term to a phase's LAST round polynomial and needs, for that, what multiplies each operand in that round ----
// After a phase's last update call: A_b(t) of both pairs as coefficients (t^0, t^1, t^2). A live pair: its multiplier table's last two entries (the hybrid
// tail holds them on the host; otherwise two entries are read back); a pair absorbed earlier: its multiplier x (1 - t); a DOT_PROD phase 1: the periodic
// table's value(s) times the other operand's pair.
extern "C" int32_t zk_sumcheck_tail_pairs(zk_ctx *ctx, uint64_t out[24]) {
    CHECK_READY();
    if (!out || ctx->round < 1) return ZK_ERR_STATE;
    HFr A[6];
    for (HFr &x : A) x.clear();
    auto read_pair = [&](const fr_t *src, uint64_t n, HFr v[2]) -> int32_t {
        v[0].clear(); v[1].clear();
        if (!src || !n) return ZK_OK;
        ZK_HIP(hipMemcpyAsync(v, src, std::min<uint64_t>(n, 2) * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        return ZK_OK;
    };
    const bool dot1 = ctx->phase_kind == 0;            // (zk_sumcheck_dotprod_init_phase1)
    if (dot1 && !ctx->dot_quad) {
        // cubic rounds to the end: pair 0 = X (the other operand's sums), pair 1 = Y (the claimed operand), `small` = the periodic table
        table_pair &x = ctx->tp[0];
        if (x.len != 2 || ctx->tp[1].len != 2) { ctx->err = "zk_sumcheck_tail_pairs: the DOT_PROD phase is not at its last round"; return ZK_ERR_STATE; }
        HFr xv[2], sv[2];
        int32_t rc = read_pair(x.V[x.cur], 2, xv);
        if (!rc) rc = read_pair(ctx->small[ctx->small_cur], std::max<uint32_t>(ctx->small_len, 1), sv);
        if (rc) return rc;
        if (ctx->small_len < 2) sv[1] = sv[0];
        const HFr x1 = xv[1] - xv[0], s1 = sv[1] - sv[0];
        A[3] = sv[0] * xv[0];
        A[4] = sv[0] * x1 + s1 * xv[0];
        A[5] = s1 * x1;
    } else {
        for (int b = 0; b < 2; ++b) {
            table_pair &t = ctx->tp[b];
            if (t.len == 2) {
                HFr m[2];
                if (ctx->host_tail_active && ctx->ht_M[b].size() >= 2) { m[0] = ctx->ht_M[b][0]; m[1] = ctx->ht_M[b][1]; }
                else {
                    if (ctx->live_active || ctx->tail_active) { ctx->err = "zk_sumcheck_tail_pairs: the phase's last pairs are inside a resident kernel (set a host tail)"; return ZK_ERR_STATE; }
                    int32_t rc = read_pair(t.M[t.cur], 2, m);
                    if (rc) return rc;
                }
                if (ctx->dot_quad) { m[0] = m[0] * ctx->small_final; m[1] = m[1] * ctx->small_final; }       // (the quadratic rounds behind a DOT_PROD phase's collapse)
                A[3 * b] = m[0];
                A[3 * b + 1] = m[1] - m[0];
            } else if (t.absorbed) {
                if (!t.abs_m_valid) { ctx->err = "zk_sumcheck_tail_pairs: an operand was absorbed inside a resident kernel (set a host tail)"; return ZK_ERR_STATE; }
                A[3 * b] = t.abs_m;
                A[3 * b + 1] = HFr(0LL) - t.abs_m;
            } else if (t.len != 0) { ctx->err = "zk_sumcheck_tail_pairs: the phase is not at its last round"; return ZK_ERR_STATE; }
        }
    }
    for (int k = 0; k < 6; ++k) put(out + 4 * k, A[k]);
    return ZK_OK;
}
// after zk_sumcheck_finalize1 / zk_sumcheck_dotprod_finalize1: the claims left masked, phase 2 is about c + d
extern "C" int32_t zk_sumcheck_claims_adjust(zk_ctx *ctx, const uint64_t d0[4], const uint64_t d1[4]) {
    CHECK_READY();
    ctx->V_u0 = ctx->V_u0 + H(d0);
    ctx->V_u1 = ctx->V_u1 + H(d1) * ctx->vu1_scale;
    return ZK_OK;
}

#include "kernel_api.hpp"

// libzkcnn_hip.so, part 2: Hyrax commitment of the layer-0 witness on the GPU (K13/K14).
// Replaces hyrax_bls12_381::polyProver of the absent upstream library (call sites: reference
// src/prover.cpp:503-511, src/verifier.cpp:128,360). Protocol: zkcnn_amd/csrc/hyrax-bls12-381/polyCommit.hpp.
//
// MSM structure (many MSMs over ONE generator vector, scalars mostly tiny and signed):
//   * per generator set, window tables T[w][j] = 2^(8w) g_j (affine) are built once, so a 255-bit
//     scalar becomes 32 independent 8-bit digits and no doubling chain is left on the critical path;
//   * scalars are folded to |s| <= (r-1)/2 with a sign, so quantised weights / bits only touch w = 0;
//   * digit d contributes d * T[w][j] = sum_k bit_k(d) 2^k T[w][j]: for each of the 8 bit planes the
//     selected table points are summed (k_planes_acc: equal shares per lane, no tree in the streaming
//     kernel), the lanes' partial sums go through row-cooperative trees (k_cl_tree, fpc_dev.cuh) and the
//     row result is sum_k 2^k S_k (k_cl_horner). No buckets, no sorting, no atomics on 144-byte points.
//   * a generator set that is used AGAIN gets a byte table F[w][d][j] = d 2^(8w) g_j (every non-zero
//     scalar byte = one mixed addition); a FRESH set (the reference's semantics: new random generators
//     for every proof, reference src/verifier.cpp:119-128) gets the digit table of window 0 only, for
//     the commitment's 8-bit rows; rows with wider scalars sum every window through that same table
//     and combine the window sums by doublings (k_cl_whorner).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include "ctx.hpp"
#include "msm_kernels.cuh"
#include "msm_cl.cuh"
#include "../ff/g1.hpp"

struct msm_state {
    g1a_t *tables = nullptr;           // [MSM_WINDOWS][m]
    uint64_t m = 0;
    struct gen_tables *gt = nullptr;   // the (shared, ref-counted) tables of the generator set in use; the pointers / flags below are copies
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
    // window tables of a FRESH generator set are built beside the proof (ensure_tables): a stream and scratch of their own
    hipStream_t aux = nullptr; hipEvent_t aux_ev = nullptr;
    void *win_scratch = nullptr; size_t win_scratch_cap = 0;
    // commitment fast path: digit table D[d][j] = d * g_j (d = 1..255, affine), per-row "has wide scalars" flags (+ the exception word)
    g1a_t *digit = nullptr; uint64_t digit_m = 0; bool digit_ready = false;
    uint32_t *hi_flags = nullptr, *row_list = nullptr; size_t flags_cap = 0;
    g1j_t *tmpJ = nullptr; size_t tmp_cap = 0;
    void *aff_scratch = nullptr; size_t aff_cap = 0;
    // generators seen again: full byte table F[w][d][j] = d * 2^(8w) * g_j (d = 1..255), so that EVERY non-zero scalar byte is
    // one mixed addition and no bit planes / doublings are left (3.2 GB for 4096 generators; built on the second use of a set)
    g1a_t *full = nullptr; uint64_t full_m = 0; bool full_ready = false;
    bool no_full = false;              // this state never takes a byte table (the verifier's second set: points that change with every proof)
    g1j_t *parts2 = nullptr; size_t parts2_cap = 0;
    // scalar pre-pass outputs: 16-bit codes of a whole matrix, canonical signed magnitudes of the rows that need every window
    // rows of bits: masks of 8 columns each (k_bit_masks) and the subset-sum table they index (k_subset_table; built with the full byte table)
    uint16_t *masks = nullptr; size_t masks_cap = 0;
    g1a_t *t8 = nullptr; uint64_t t8_m = 0; bool t8_ready = false;
    uint16_t *codes = nullptr; size_t codes_cap = 0;          // 16-bit codes of a matrix, followed by those of the virtual rows (higher windows of wide rows)
    fr_t *mag = nullptr; size_t mag_cap = 0;
    uint32_t *exc = nullptr;           // device word set by a fast-variant kernel that met P = +-Q
    bool safe = false;                 // run the SAFE kernel variants (after a flagged batch)
    void *fb_buf = nullptr; size_t fb_cap = 0;                // zk_fixed_base_mul: results + scalars
    unsigned char *h_stage = nullptr;  // pinned: the <= 8 points + the exception word a few-row MSM hands to the host (a pageable target costs a staged copy: ~70 us)
};
#define MSM_STAGE_BYTES (8 * sizeof(g1j_t) + 64)
#define MSM_FULL_MAX_M 16384u
#define MSM_WIDE_CAP 128u          // rows with wide scalars whose higher windows ride along as virtual rows of the commitment's launches
#define ZK_RETRY_SAFE 0x5afe       // internal status: repeat the batch with the SAFE kernels

// ---- one set of tables per generator set and GPU, shared by the contexts that use the set (round 3: eight sessions of one model used to
// build eight 3.2 GB byte tables of the same public generators) ----
struct gen_tables {
    int device = 0;
    uint64_t m = 0;
    std::vector<uint64_t> gens;        // the affine generators (C-ABI layout): the key
    int refs = 0;
    uint32_t uses = 0;                 // times a context selected the set: the byte table is built when a set is used AGAIN
    std::atomic<const void *> owner{nullptr};   // the context that is building a table of the set (set_lock); no thread owns anything: lanes of a batch share one
    g1a_t *tables = nullptr;           // window tables T[w][j] = 2^(8w) g_j
    hipEvent_t win_ev = nullptr;       // set: the windows above MSM_LOW_WINDOWS are (being) written on another stream -- whoever reads them makes its stream wait (wait_windows)
    bool windows_built = false;        // T[w >= 1] exist (or are being written: win_ev). A set that starts its life among several proofs in flight builds them when somebody asks (wait_windows)
    g1a_t *digit = nullptr; bool digit_ready = false;
    g1a_t *full = nullptr; bool full_ready = false, full_failed = false;
    g1a_t *t8 = nullptr; bool t8_ready = false;
};
static std::mutex g_gen_mtx;
// Window and digit tables of generator sets that have died, kept for the next set of the same size: in the reference's semantics EVERY proof brings a new set
// (reference src/verifier.cpp:119-128), and hipFree / hipMalloc of its 113 MB per proof are device-wide synchronisations -- 48 proofs in flight stalled one
// another's streams on them (round 6: profiles/r06_final_kernel_stats_s48.md). A buffer enters the pool only after its last user's stream has drained
// (gen_release). The byte table of a reusable set (3.2 GB) is not pooled.
struct pooled_table { int device; size_t bytes; void *p; };
static std::mutex g_pool_mtx;
static std::vector<pooled_table> g_table_pool;
static hipError_t table_take(int device, size_t bytes, void **p) {
    {
        std::lock_guard<std::mutex> g(g_pool_mtx);
        for (size_t i = 0; i < g_table_pool.size(); ++i)
            if (g_table_pool[i].device == device && g_table_pool[i].bytes == bytes) {
                *p = g_table_pool[i].p;
                g_table_pool.erase(g_table_pool.begin() + i);
                return hipSuccess;
            }
    }
    return hipMalloc(p, bytes);
}
static void table_give(int device, size_t bytes, void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pool_mtx);
        if (g_table_pool.size() < 256) { g_table_pool.push_back({device, bytes, p}); return; }
    }
    (void) hipFree(p);
}
// The lock of a generator set, taken by a context. A lane of a lock-step batch may be PARKED while it holds the lock (the tables of a fresh set are
// built with deferred launches, so that the lanes' builds fuse), and the other lanes run on the same thread: whoever finds the lock taken hands
// the thread on (zk_batch_sync_point) instead of blocking in it.
struct set_lock {
    gen_tables *e;
    int32_t rc = ZK_OK;
    // (round-4 advisor finding: this used to spin on std::mutex::try_lock, which a lane may call on the very thread whose parked lane holds the mutex --
    // undefined behaviour. The lock is an owner word now: compare-and-swap by CONTEXT, released by whoever holds it, from any thread.)
    set_lock(zk_ctx *ctx, gen_tables *entry) : e(entry) {
        if (e->owner.load(std::memory_order_relaxed) == (const void *) ctx) { e = nullptr; return; }      // held by this context further up the stack: theirs to release
        for (;;) {
            const void *none = nullptr;
            if (e->owner.compare_exchange_weak(none, (const void *) ctx, std::memory_order_acquire, std::memory_order_relaxed)) return;
            if (ctx->batch) { if ((rc = zk_batch_sync_point(ctx))) { e = nullptr; return; } }
            else std::this_thread::yield();
        }
    }
    ~set_lock() { if (e) e->owner.store(nullptr, std::memory_order_release); }
    set_lock(const set_lock &) = delete;
    set_lock &operator=(const set_lock &) = delete;
};
static std::vector<gen_tables *> g_gen_sets;
static std::atomic<uint64_t> g_gen_builds{0}, g_gen_full_builds{0};
extern "C" void zk_generator_table_stats(uint64_t *window_table_builds, uint64_t *byte_table_builds) {
    if (window_table_builds) *window_table_builds = g_gen_builds.load();
    if (byte_table_builds) *byte_table_builds = g_gen_full_builds.load();
}
// copies of the entry's pointers / flags in the context's state (what the launch sites read)
static void gen_adopt(msm_state *s) {
    gen_tables *e = s->gt;
    s->tables = e->tables;
    s->digit = e->digit; s->digit_ready = e->digit_ready; s->digit_m = e->m;
    s->full = e->full; s->full_ready = e->full_ready && !s->no_full; s->full_m = e->m;
    s->t8 = e->t8; s->t8_ready = e->t8_ready && s->full_ready; s->t8_m = e->m;
}
static void gen_release(zk_ctx *ctx, msm_state *s) {
    gen_tables *e = s->gt;
    s->gt = nullptr;
    s->tables = nullptr; s->digit = nullptr; s->full = nullptr; s->t8 = nullptr;
    s->digit_ready = s->full_ready = s->t8_ready = false;
    if (!e) return;
    if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
    std::lock_guard<std::mutex> g(g_gen_mtx);
    if (--e->refs > 0) return;
    if (e->win_ev) { (void) hipEventSynchronize(e->win_ev); (void) hipEventDestroy(e->win_ev); }
    table_give(e->device, (size_t) MSM_WINDOWS * e->m * sizeof(g1a_t), e->tables);
    table_give(e->device, (size_t) 256 * e->m * sizeof(g1a_t), e->digit);
    for (void *p : {(void *) e->full, (void *) e->t8}) if (p) hipFree(p);
    g_gen_sets.erase(std::remove(g_gen_sets.begin(), g_gen_sets.end(), e), g_gen_sets.end());
    delete e;
}

static void msm_destroy_one(zk_ctx *ctx);
void zk_msm_destroy(zk_ctx *ctx) {
    msm_destroy_one(ctx);
    std::swap(ctx->msm, ctx->vmsm);
    msm_destroy_one(ctx);
}
static void msm_destroy_one(zk_ctx *ctx) {
    if (!ctx->msm) return;
    msm_state *s = ctx->msm;
    if (s->aux) (void) hipStreamSynchronize(s->aux);
    gen_release(ctx, s);
    if (s->aux_ev) (void) hipEventDestroy(s->aux_ev);
    if (s->aux) (void) hipStreamDestroy(s->aux);
    if (s->win_scratch) (void) hipFree(s->win_scratch);
    if (s->h_stage) (void) hipHostFree(s->h_stage);
    void *bufs[] = {s->partials, s->rowsJ, s->rowsA, s->a, s->b, s->coef, s->Lrow, s->sL, s->idxL, s->d_y, s->tbl_scratch,
                    s->hi_flags, s->row_list, s->tmpJ, s->aff_scratch, s->parts2, s->codes, s->mag, s->exc, s->masks, s->fb_buf};
    for (void *p : bufs) if (p) hipFree(p);
    delete s;
    ctx->msm = nullptr;
}

static int32_t regrow(zk_ctx *ctx, void **p, size_t *cap, size_t bytes) {
    if (*cap >= bytes) return ZK_OK;
    ZK_ORDER();                        // (a lane of a batch: launches it has deferred may still read the buffer that is let go)
    if (*p) { ZK_HIP(zk_stream_sync(ctx)); ZK_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
    ZK_HIP(hipMalloc(p, bytes));
    *cap = bytes;
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int32_t ensure_state(zk_ctx *ctx) {
    if (!ctx->msm) ctx->msm = new msm_state();
    msm_state *s = ctx->msm;
    if (!s->h_stage) ZK_HIP(hipHostMalloc((void **) &s->h_stage, MSM_STAGE_BYTES));
    if (!s->exc) {
        ZK_HIP(hipMalloc((void **) &s->exc, 64));
        ZK_STREAM(hipMemsetAsync(s->exc, 0, 64, ctx->stream));
    }
    return ZK_OK;
}

// The windows T[w][j] = 2^(8w) g_j, w >= 1, of the set `e` (window 0 = the generators is in place): 248 doublings per generator, one thread each -- 7.5 ms for
// 4096 generators on 64 of the 1024 SIMDs. They are read by the OPENING's bit-plane MSMs (a proof on its own), by the byte table's construction and by
// the odd fallback path; the commitment goes through the digit table of window 0. `beside`: on the context's second stream, next to its sumcheck phases --
// readers order themselves behind the set's event (wait_windows). Otherwise in order on the context's stream (a lane's launches are deferred and fused with
// the other lanes' builds; the class table of a profiled run stays whole). The caller holds the set's lock.
static int32_t build_windows(zk_ctx *ctx, gen_tables *e, bool beside) {
    msm_state *s = ctx->msm;
    const uint64_t m = e->m;
    const size_t need = (size_t) (MSM_WINDOWS - 1) * m * (sizeof(g1j_t) + sizeof(fp_t));
    if (beside) {
        if (!s->aux) { ZK_HIP(hipStreamCreateWithFlags(&s->aux, hipStreamNonBlocking)); ZK_HIP(hipEventCreateWithFlags(&s->aux_ev, hipEventDisableTiming)); }
        if (s->win_scratch_cap < need) {
            ZK_HIP(hipStreamSynchronize(s->aux));
            if (s->win_scratch) { ZK_HIP(hipFree(s->win_scratch)); s->win_scratch = nullptr; s->win_scratch_cap = 0; }
            ZK_HIP(hipMalloc(&s->win_scratch, need));
            s->win_scratch_cap = need;
        }
        g1j_t *J = (g1j_t *) s->win_scratch;
        fp_t *pre = (fp_t *) (J + (size_t) (MSM_WINDOWS - 1) * m);
        ZK_HIP(hipEventCreateWithFlags(&e->win_ev, hipEventDisableTiming));
        ZK_HIP(hipEventRecord(s->aux_ev, ctx->stream));
        ZK_HIP(hipStreamWaitEvent(s->aux, s->aux_ev, 0));
        call_f<k_window_tables, g1a_t *, g1j_t *, fp_t *, uint32_t, uint32_t, uint32_t> f;
        f.args = make_pack<g1a_t *, g1j_t *, fp_t *, uint32_t, uint32_t, uint32_t>(e->tables, J, pre, (uint32_t) m, 1u, (uint32_t) MSM_WINDOWS);
        hipLaunchKernelGGL((k_run<decltype(f), 64>), dim3((uint32_t) ((m + 63) / 64)), dim3(64), 0, s->aux, f);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipEventRecord(e->win_ev, s->aux));
        e->windows_built = true;
        return ZK_OK;
    }
    int32_t rc = regrow(ctx, &s->tbl_scratch, &s->tbl_scratch_cap, need);
    if (rc) return rc;
    g1j_t *J = (g1j_t *) s->tbl_scratch;
    fp_t *pre = (fp_t *) (J + (size_t) (MSM_WINDOWS - 1) * m);
    zk_launch_d<k_window_tables, 64>(ctx, PC_MSM_TABLES, 0.0, dim3((uint32_t) ((m + 63) / 64)), e->tables, J, pre, (uint32_t) m, 1u, (uint32_t) MSM_WINDOWS);
    ZK_HIP(hipGetLastError());
    ZK_HIP(zk_stream_sync(ctx));      // other contexts read the tables from their own streams
    e->windows_built = true;
    return ZK_OK;
}

// With several proofs in flight the opening of a FRESH generator set sums its windows through the digit table (msm_windows: k_bytes_acc by window +
// k_cl_whorner32) and the set never gets window tables (experiment switch ZKCNN_DIGIT_OPENING=0: round 6's first design, window tables for every set).
static bool digit_opening() {
    const char *e = getenv("ZKCNN_DIGIT_OPENING");      // (read per call: scripts/exp/ab_inprocess.py alternates it between steps of one process)
    return !e || atoi(e) != 0;
}

// tables for `m` affine generators (host pointer, C-ABI layout): looked up in / added to the registry of generator sets
static int32_t ensure_tables(zk_ctx *ctx, const uint64_t *gens, uint64_t m) {
    msm_state *s = ctx->msm;
    ZK_ORDER();                        // nothing of this lane may be deferred while a table is built under the set's lock (the other lanes run on this thread)
    if (s->gt && s->m == m && s->gens_host.size() == m * 12 && std::memcmp(s->gens_host.data(), gens, m * 96) == 0) {
        set_lock g(ctx, s->gt);
        if (g.rc) return g.rc;
        ++s->gt->uses;
        return ZK_OK;
    }
    ZK_HIP(zk_stream_sync(ctx));          // (nothing of this context may still read the set that is let go)
    gen_release(ctx, s);
    gen_tables *e = nullptr;
    {
        std::lock_guard<std::mutex> g(g_gen_mtx);
        for (gen_tables *c : g_gen_sets)
            if (c->device == ctx->device && c->m == m && std::memcmp(c->gens.data(), gens, m * 96) == 0) { e = c; break; }
        if (!e) {
            e = new gen_tables();
            e->device = ctx->device;
            e->m = m;
            e->gens.assign(gens, gens + m * 12);
            g_gen_sets.push_back(e);
        }
        ++e->refs;
    }
    s->gt = e;
    s->gens_host.assign(gens, gens + m * 12);
    s->m = m;
    set_lock g(ctx, e);
    if (g.rc) return g.rc;
    ++e->uses;
    if (!e->tables) {
        ZK_HIP(table_take(e->device, (size_t) MSM_WINDOWS * m * sizeof(g1a_t), (void **) &e->tables));
        // (a failure below must not leave the entry looking built: every context that looks the set up adopts e->tables)
        auto build = [&]() -> int32_t {
            ZK_STREAM(hipMemcpyAsync(e->tables, gens, m * sizeof(g1a_t), hipMemcpyHostToDevice, ctx->stream));
            // a proof on its own: every window above 0 beside its sumcheck (the first reader is the opening); a profiled run: in order; a proof among others:
            // not at all until somebody asks (wait_windows)
            const bool profiled = (ctx->prof_mask >> PC_MSM_TABLES) & 1u;
            int32_t rc = ZK_OK;
            if (!ctx->batch && !profiled && (ctx->live_now || !digit_opening())) rc = build_windows(ctx, e, true);
            else if (!digit_opening()) return build_windows(ctx, e, false);
            if (rc) return rc;
            ZK_HIP(zk_stream_sync(ctx));      // (window 0: `gens` is the caller's buffer; other contexts read it from their own streams)
            return ZK_OK;
        };
        const int32_t rc = build();
        if (rc) {
            (void) zk_stream_sync(ctx);
            if (s->aux) (void) hipStreamSynchronize(s->aux);
            if (e->win_ev) { (void) hipEventDestroy(e->win_ev); e->win_ev = nullptr; }
            table_give(e->device, (size_t) MSM_WINDOWS * e->m * sizeof(g1a_t), e->tables); e->tables = nullptr;
            e->windows_built = false;
            return rc;
        }
        ++g_gen_builds;
    }
    gen_adopt(s);
    return ZK_OK;
}

// before this context's stream reads a window w >= 1 of its generator set: builds them if nobody has, waits for whoever is writing them
static int32_t wait_windows(zk_ctx *ctx) {
    gen_tables *e = ctx->msm->gt;
    if (!e) return ZK_OK;
    if (!e->windows_built) {
        ZK_ORDER();
        set_lock g(ctx, e);
        if (g.rc) return g.rc;
        if (!e->windows_built) { int32_t rc = build_windows(ctx, e, false); if (rc) return rc; }
    }
    if (e->win_ev) ZK_STREAM(hipStreamWaitEvent(ctx->stream, e->win_ev, 0));
    return ZK_OK;
}

static int32_t ensure_rows(zk_ctx *ctx, uint32_t rows) {
    msm_state *s = ctx->msm;
    if (s->rows_cap >= rows) return ZK_OK;
    ZK_ORDER();
    if (s->rowsJ) { ZK_HIP(zk_stream_sync(ctx)); ZK_HIP(hipFree(s->rowsJ)); ZK_HIP(hipFree(s->rowsA)); s->rowsJ = nullptr; }
    ZK_HIP(hipMalloc((void **) &s->rowsJ, (size_t) rows * sizeof(g1j_t)));
    ZK_HIP(hipMalloc((void **) &s->rowsA, (size_t) rows * sizeof(g1a_t)));
    s->rows_cap = rows;
    return ZK_OK;
}

// launch shapes for a GPU that other proofs share: fewer, smaller workgroups where a proof on its own takes many or large ones (experiment switch ZKCNN_LOAD_SHAPES=0)
static bool load_shapes() {
    const char *e = getenv("ZKCNN_LOAD_SHAPES");
    return !e || atoi(e) != 0;
}

// digits per inversion of the digit table's conversion to affine (k_digit_affine; experiment switch ZKCNN_DIGIT_AFFINE_PER). A proof on its own: 16 (the short
// chain). Several proofs in flight: 64 -- 0.39 of the issue slots for a chain of 928 instead of 592 products. While the batches' streams still waited for their host
// threads most of the time that saving did not show (steps of one process alternating 16 / 64 / 128: 93.8 / 89.3-91.4 / 90.5 proofs/s); since the verifier's
// generator loop runs on the GPU the streams are busy and it does: 133.2 / 135.4 / 134.9 proofs/s (profiles/r06_retune.txt).
static uint32_t digit_affine_per(const zk_ctx *ctx) {
    const char *e = getenv("ZKCNN_DIGIT_AFFINE_PER");
    const uint32_t x = e ? (uint32_t) atoi(e) : (ctx->live_now ? 16u : 64u);
    return x == 16 || x == 32 || x == 64 || x == 128 || x == 256 ? x : 16u;
}

// d * base_j for d = 1..255 of `m` affine points (levels of doubling / adding, then a batched conversion to affine)
static int32_t build_digit_table(zk_ctx *ctx, g1a_t *dst, const g1a_t *base, uint32_t m) {
    msm_state *s = ctx->msm;
    int32_t rc = regrow(ctx, &s->tbl_scratch, &s->tbl_scratch_cap, (size_t) 256 * m * (sizeof(g1j_t) + sizeof(fp_t)));
    if (rc) return rc;
    g1j_t *J = (g1j_t *) s->tbl_scratch;
    fp_t *pre = (fp_t *) (J + (size_t) 256 * m);
    for (uint32_t level = 0; level < 8; ++level) {
        const uint32_t work = (1u << level) * m;
        zk_launch_d<k_digit_level, 64>(ctx, PC_MSM_TABLES, 0.0, dim3((work + 63) / 64), J, base, m, level);
    }
    const uint32_t per = digit_affine_per(ctx);
    zk_launch_d<k_digit_affine, 64>(ctx, PC_MSM_TABLES, 0.0, dim3(((256 / per) * m + 63) / 64), dst, (const g1j_t *) J, pre, m, per);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// full byte table for a generator set that is being used again -- by this context or by any other of the process: the table belongs to the
// set's registry entry, whoever needs it first builds it
static int32_t ensure_full_table(zk_ctx *ctx) {
    msm_state *s = ctx->msm;
    gen_tables *e = s->gt;
    if (!e || s->full_ready || s->no_full || s->m > MSM_FULL_MAX_M) return ZK_OK;
    ZK_ORDER();
    set_lock g(ctx, e);
    if (g.rc) return g.rc;
    if (!e->full_ready && !e->full_failed && e->uses >= 2) {
        const uint32_t m = (uint32_t) e->m;
        if (hipMalloc((void **) &e->full, (size_t) MSM_WINDOWS * 256 * m * sizeof(g1a_t)) != hipSuccess) {
            (void) hipGetLastError();
            e->full = nullptr;
            e->full_failed = true;             // not enough memory for the table: stay on the bit-plane path
        } else {
            { int32_t rc = wait_windows(ctx); if (rc) return rc; }
            for (uint32_t w = 0; w < MSM_WINDOWS; ++w) {
                int32_t rc = build_digit_table(ctx, e->full + (size_t) w * 256 * m, e->tables + (size_t) w * m, m);
                if (rc) return rc;
            }
            // subset sums of 8 generators (rows of bits in commit_rows): 256 x m / 8 points
            if (m % 512 == 0 && m <= 64 * MSM_BLOCK) {
                const uint32_t n8 = m / 8;
                if (hipMalloc((void **) &e->t8, (size_t) 256 * n8 * sizeof(g1a_t)) != hipSuccess) { (void) hipGetLastError(); e->t8 = nullptr; }
                else {
                    ZK_STREAM(hipMemsetAsync(e->t8, 0, (size_t) n8 * sizeof(g1a_t), ctx->stream));            // mask 0: the point at infinity, never looked up
                    zk_launch_d<k_subset_table, 64>(ctx, PC_MSM_TABLES, 0.0, dim3((n8 + 63) / 64, 255), e->t8, (const g1a_t *) (e->full + (size_t) 1 * m), n8);
                    ZK_HIP(hipGetLastError());
                    e->t8_ready = true;
                }
            }
            ZK_HIP(zk_stream_sync(ctx));
            e->full_ready = true;
            ++g_gen_full_builds;
        }
    }
    gen_adopt(s);
    return ZK_OK;
}

// digit table D[d][j] = d g_j for the cached generators
static int32_t ensure_digit_table(zk_ctx *ctx) {
    msm_state *s = ctx->msm;
    gen_tables *e = s->gt;
    if (s->digit_ready) return ZK_OK;
    ZK_ORDER();
    set_lock g(ctx, e);
    if (g.rc) return g.rc;
    if (!e->digit_ready) {
        const uint32_t m = (uint32_t) e->m;
        ZK_HIP(table_take(e->device, (size_t) 256 * m * sizeof(g1a_t), (void **) &e->digit));
        int32_t rc = build_digit_table(ctx, e->digit, e->tables, m);
        if (!rc && zk_stream_sync(ctx) != hipSuccess) { ctx->err = "digit table: stream synchronisation failed"; rc = ZK_ERR_HIP; }
        if (rc) { table_give(e->device, (size_t) 256 * m * sizeof(g1a_t), e->digit); e->digit = nullptr; return rc; }     // (not built: nobody may adopt it, and the next attempt allocates again)
        e->digit_ready = true;
    }
    gen_adopt(s);
    return ZK_OK;
}

// sums the `n` partial points of every row (row-major in `src`) into dst[row]
static int32_t reduce_rows(zk_ctx *ctx, const g1j_t *src, uint32_t n, uint32_t rows, g1j_t *dst, uint32_t n_real = 0, const uint32_t *n_wide = nullptr) {
    if (n == 1) {
        ZK_STREAM(hipMemcpyAsync(dst, src, (size_t) rows * sizeof(g1j_t), hipMemcpyDeviceToDevice, ctx->stream));
        return ZK_OK;
    }
    zk_launch_d<k_reduce_rows16, MSM_BLOCK>(ctx, PC_MSM_FINISH, 0.0, dim3((rows + 3) / 4), dst, src, n, rows, n_real, n_wide);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// canonical signed magnitudes of the listed rows (all rows if row_map is null) into s->mag, dense rows of `cols` entries
static int32_t scalar_mags(zk_ctx *ctx, const fr_t *scalars, uint64_t ld, const uint32_t *row_map, uint32_t rows, uint32_t cols) {
    msm_state *s = ctx->msm;
    int32_t rc = regrow(ctx, (void **) &s->mag, &s->mag_cap, (size_t) rows * cols * sizeof(fr_t));
    if (rc) return rc;
    for (uint32_t r0 = 0; r0 < rows; r0 += 32768) {
        const uint32_t nr = std::min<uint32_t>(32768, rows - r0);
        zk_launch_d<k_scalar_mags, 256>(ctx, PC_MSM_PLANES, 0.0, dim3(std::min<uint32_t>((cols + 255) / 256, 64), nr), s->mag + (size_t) r0 * cols,
                  row_map ? scalars : scalars + (size_t) r0 * ld, ld, row_map ? row_map + r0 : nullptr, cols);
    }
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// pairs per lane of the streaming MSM kernels when several proofs share the GPU (experiment switch ZKCNN_ACC_PAIRS; see msm_windows)
static uint32_t busy_pairs() {
    static const uint32_t v = [] {
        const char *e = getenv("ZKCNN_ACC_PAIRS");
        const uint32_t x = e ? (uint32_t) atoi(e) : 32u;
        return std::min<uint32_t>(std::max<uint32_t>(x, 1u), ACC_MAX_PAIRS);
    }();
    return v;
}

// columns per lane of the opening's digit-table route (experiment switch ZKCNN_DIGIT_PAIRS)
static uint32_t digit_pairs() {
    const char *e = getenv("ZKCNN_DIGIT_PAIRS");
    return e ? (uint32_t) std::max(1, atoi(e)) : 8u;
}

// rows independent MSMs over the cached generator tables, every window >= w_lo of every scalar; scalars are read from s->mag
// (scalar_mags ran before). idx (optional): generator index of every column, rows `ld` apart. Results (Jacobian) in `outJ` or s->rowsJ.
static int32_t msm_windows(zk_ctx *ctx, const uint32_t *idx, uint64_t ld, uint32_t rows, uint32_t cols, uint32_t w_lo, g1j_t *outJ, bool low_windows_only = false) {
    msm_state *s = ctx->msm;
    int32_t rc;
    const uint32_t nwin = MSM_WINDOWS - w_lo;
    g1j_t *dst = outJ ? outJ : s->rowsJ;
    // pairs (column, window) per lane <= ACC_MAX_PAIRS (the wave's list). A proof that is ALONE on the GPU spreads wide -- short chains of one-lane mixed additions
    // (27 us each), the rest is row-cooperative trees; with several proofs in flight the SIMD time counts, and the streaming kernel's lanes are three times as
    // efficient as the trees' rows, so chains are as long as the list allows.
    auto shape = [&](uint32_t nw, uint32_t per_lane, uint64_t min_waves, uint32_t planes, uint32_t &cpt, uint32_t &wsplit) {
        uint32_t wpg = nw;
        wsplit = 1;
        cpt = std::max<uint32_t>(1, std::min<uint32_t>(per_lane / std::min(per_lane, wpg), (cols + MSM_BLOCK - 1) / MSM_BLOCK));
        while (wpg * cpt > per_lane && wpg > 1) { wsplit *= 2; wpg = (nw + wsplit - 1) / wsplit; }
        auto waves = [&]() { return (uint64_t) rows * planes * ((cols + MSM_BLOCK * cpt - 1) / (MSM_BLOCK * cpt)) * wsplit; };
        while (waves() < min_waves && (cpt > 1 || wpg > 2)) {
            if (cpt > 1) cpt = (cpt + 1) / 2;
            else { wsplit *= 2; wpg = (nw + wsplit - 1) / wsplit; }
        }
    };
    // sums of the n partial points of each of `segs` segments (dense in `cur`) -> out[seg]; cur / nxt ping-pong
    auto trees = [&](g1j_t *cur, g1j_t *nxt, uint32_t n, uint32_t segs, g1j_t *out) {
        for (;;) {
            const uint32_t n_in = std::min<uint32_t>(n, 64), blocks = (n + n_in - 1) / n_in;
            g1j_t *o = blocks == 1 ? out : nxt;
            zk_launch_d<k_cl_tree, 512>(ctx, PC_MSM_FINISH, 0.0, dim3(blocks, segs), o, (const g1j_t *) cur, n, n_in);
            if (blocks == 1) return;
            std::swap(cur, nxt);
            n = blocks;
        }
    };
    if (s->full_ready) {
        // byte table: every non-zero scalar byte is one mixed addition (k_bytes_acc), the lanes' partial sums go through row-cooperative trees
        uint32_t cpt, wsplit;
        shape(nwin, ctx->live_now ? 4u : busy_pairs(), ctx->live_now ? 512 : 1, 1, cpt, wsplit);
        const uint32_t gx = ((cols + MSM_BLOCK * cpt - 1) / (MSM_BLOCK * cpt)) * wsplit, nparts = gx * MSM_BLOCK, n2 = (nparts + 63) / 64;
        if ((rc = regrow(ctx, (void **) &s->partials, &s->partials_cap, (size_t) rows * nparts * sizeof(g1j_t)))) return rc;
        if ((rc = regrow(ctx, (void **) &s->parts2, &s->parts2_cap, (size_t) rows * n2 * sizeof(g1j_t)))) return rc;
        for (uint32_t r0 = 0; r0 < rows; r0 += 32768) {
            const uint32_t nr = std::min<uint32_t>(32768, rows - r0);
            const double bytes = 32.0 * (double) nr * (double) cols;
            g1j_t *part = s->partials + (size_t) r0 * nparts;
            if (s->safe)
                zk_launch_d<k_bytes_acc<true>, MSM_BLOCK>(ctx, PC_MSM_PLANES, bytes, dim3(gx, nr), part, s->exc, (const fr_t *) (s->mag + (size_t) r0 * cols),
                          ld, idx ? idx + (size_t) r0 * ld : (const uint32_t *) nullptr, (const g1a_t *) s->full, (uint32_t) s->m, cols, cpt, wsplit, w_lo, (uint32_t) MSM_WINDOWS, 256u * (uint32_t) s->m, 0u);
            else
                zk_launch_d<k_bytes_acc<false>, MSM_BLOCK>(ctx, PC_MSM_PLANES, bytes, dim3(gx, nr), part, s->exc, (const fr_t *) (s->mag + (size_t) r0 * cols),
                          ld, idx ? idx + (size_t) r0 * ld : (const uint32_t *) nullptr, (const g1a_t *) s->full, (uint32_t) s->m, cols, cpt, wsplit, w_lo, (uint32_t) MSM_WINDOWS, 256u * (uint32_t) s->m, 0u);
            trees(part, s->parts2 + (size_t) r0 * n2, nparts, nr, dst + r0);
        }
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    if (s->digit_ready && s->digit_m == s->m && !ctx->live_now && w_lo == 0 && !low_windows_only && digit_opening()) {
        // no byte table (a fresh generator set: the reference's semantics) and other proofs in flight: V_w = sum_j byte_w(s_j) g_j of every window through the
        // DIGIT table the commitment built -- one mixed addition per non-zero byte, a third of the bit planes' additions and no window tables 2^(8w) g_j at all
        // (their 248 doublings per generator were 6.9 of a batch proof's ~80 GPU-ms) -- then row = sum_w 2^(8w) V_w by doublings (k_cl_whorner32: a 0.7 ms chain
        // per round of the opening, which the other proofs' work hides; a proof on its own takes the planes below)
        const uint32_t per_lane = std::min<uint32_t>(digit_pairs(), ACC_MAX_PAIRS);
        const uint32_t cpt = std::max<uint32_t>(1, std::min<uint32_t>(per_lane, (cols + MSM_BLOCK - 1) / MSM_BLOCK)), chunks = (cols + MSM_BLOCK * cpt - 1) / (MSM_BLOCK * cpt);
        const uint32_t nparts = chunks * MSM_BLOCK, n2 = (nparts + 63) / 64, segs_per_row = MSM_WINDOWS;
        if ((rc = regrow(ctx, (void **) &s->partials, &s->partials_cap, (size_t) rows * segs_per_row * nparts * sizeof(g1j_t)))) return rc;
        if ((rc = regrow(ctx, (void **) &s->parts2, &s->parts2_cap, (size_t) rows * segs_per_row * (n2 + 1) * sizeof(g1j_t)))) return rc;
        for (uint32_t r0 = 0; r0 < rows; r0 += 2047) {       // (window, row) share the trees' gridDim.y, which is limited to 65535
            const uint32_t nr = std::min<uint32_t>(2047, rows - r0);
            const double bytes = 32.0 * (double) nr * (double) cols;
            g1j_t *part = s->partials + (size_t) r0 * segs_per_row * nparts, *part2 = s->parts2 + (size_t) r0 * segs_per_row * (n2 + 1);
            if (s->safe)
                zk_launch_d<k_bytes_acc<true>, MSM_BLOCK>(ctx, PC_MSM_PLANES, bytes, dim3(chunks * MSM_WINDOWS, nr), part, s->exc, (const fr_t *) (s->mag + (size_t) r0 * cols),
                          ld, idx ? idx + (size_t) r0 * ld : (const uint32_t *) nullptr, (const g1a_t *) s->digit, (uint32_t) s->m, cols, cpt, (uint32_t) MSM_WINDOWS, 0u, (uint32_t) MSM_WINDOWS, 0u, 1u);
            else
                zk_launch_d<k_bytes_acc<false>, MSM_BLOCK>(ctx, PC_MSM_PLANES, bytes, dim3(chunks * MSM_WINDOWS, nr), part, s->exc, (const fr_t *) (s->mag + (size_t) r0 * cols),
                          ld, idx ? idx + (size_t) r0 * ld : (const uint32_t *) nullptr, (const g1a_t *) s->digit, (uint32_t) s->m, cols, cpt, (uint32_t) MSM_WINDOWS, 0u, (uint32_t) MSM_WINDOWS, 0u, 1u);
            g1j_t *V = part2 + (size_t) segs_per_row * nr * n2;            // the window sums: [row][window], behind the trees' own intermediate results
            trees(part, part2, nparts, segs_per_row * nr, V);
            zk_launch_d<k_cl_whorner32, 512>(ctx, PC_MSM_FINISH, 0.0, dim3(nr), dst + r0, (const g1j_t *) V);
        }
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    // no byte table (a fresh generator set: the reference's semantics): bit planes over the window tables (msm_cl.cuh). Every lane of k_planes_acc adds an
    // equal share of its wave's selected (generator, window) pairs; the lanes' partial sums are added by row-cooperative trees of 64 and the eight plane
    // sums of a row by k_cl_horner. 2 x 2048 generators with full-width scalars (a round of the opening): 0.52 ms, round 5's k_msm_planes + k_msm_finish 1.67 ms.
    if ((rc = wait_windows(ctx))) return rc;
    // (scalars below 2^64 reach windows 0..7; the top digit of their non-adjacent form may sit at bit 64: window 8)
    const uint32_t w_hi = low_windows_only ? MSM_LOW_WINDOWS + 2u : (uint32_t) MSM_WINDOWS, nw = w_hi - w_lo;
    if ((uint64_t) cols * nw <= 64) {
        // a handful of pairs per row (the blinding term of a zero-knowledge commitment: one column): one block per (row, plane), a one-lane tree inside it
        const uint32_t cpt = 1, chunks = (cols + MSM_BLOCK - 1) / MSM_BLOCK;
        if ((rc = regrow(ctx, (void **) &s->partials, &s->partials_cap, (size_t) rows * MSM_PLANES * chunks * sizeof(g1j_t)))) return rc;
        for (uint32_t r0 = 0; r0 < rows; r0 += 8191) {
            const uint32_t nr = std::min<uint32_t>(8191, rows - r0);
            zk_launch_d<k_msm_planes, MSM_BLOCK>(ctx, PC_MSM_PLANES, 32.0 * (double) nr * (double) cols, dim3(chunks, MSM_PLANES * nr),
                      s->partials + (size_t) r0 * MSM_PLANES * chunks, (const fr_t *) (s->mag + (size_t) r0 * cols), ld, idx ? idx + (size_t) r0 * ld : (const uint32_t *) nullptr, (const g1a_t *) s->tables,
                      (uint32_t) s->m, cols, cpt, 1u, w_lo);
        }
        zk_launch_d<k_msm_finish, 512>(ctx, PC_MSM_FINISH, 0.0, dim3(rows), dst, (const g1j_t *) s->partials, chunks);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    uint32_t cpt, wsplit;
    shape(nw, ctx->live_now ? 16u : busy_pairs(), ctx->live_now ? 1024 : 1, MSM_PLANES, cpt, wsplit);
    const uint32_t gx = ((cols + MSM_BLOCK * cpt - 1) / (MSM_BLOCK * cpt)) * wsplit, nparts = gx * MSM_BLOCK, n2 = (nparts + 63) / 64;
    if ((rc = regrow(ctx, (void **) &s->partials, &s->partials_cap, (size_t) rows * MSM_PLANES * nparts * sizeof(g1j_t)))) return rc;
    if ((rc = regrow(ctx, (void **) &s->parts2, &s->parts2_cap, (size_t) rows * MSM_PLANES * (n2 + 1) * sizeof(g1j_t)))) return rc;
    for (uint32_t r0 = 0; r0 < rows; r0 += 8191) {       // (plane, row) share gridDim.y, which is limited to 65535
        const uint32_t nr = std::min<uint32_t>(8191, rows - r0);
        g1j_t *part = s->partials + (size_t) r0 * MSM_PLANES * nparts, *part2 = s->parts2 + (size_t) r0 * MSM_PLANES * (n2 + 1);
        zk_launch_d<k_planes_acc, MSM_BLOCK>(ctx, PC_MSM_PLANES, 32.0 * (double) nr * (double) cols, dim3(gx, MSM_PLANES * nr),
                  part, (const fr_t *) (s->mag + (size_t) r0 * cols), ld, idx ? idx + (size_t) r0 * ld : (const uint32_t *) nullptr, (const g1a_t *) s->tables,
                  (uint32_t) s->m, cols, cpt, wsplit, w_lo, w_hi);
        g1j_t *planes = part2 + (size_t) MSM_PLANES * nr * n2;           // the plane sums: [row][plane], behind the trees' own intermediate results
        trees(part, part2, nparts, MSM_PLANES * nr, planes);
        zk_launch_d<k_cl_horner, 128>(ctx, PC_MSM_FINISH, 0.0, dim3(nr), dst + r0, (const g1j_t *) planes);
    }
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// rows independent MSMs with full-width scalars (Montgomery form, rows `ld` apart); results in s->rowsJ
static int32_t run_msm(zk_ctx *ctx, const fr_t *scalars, uint64_t ld, const uint32_t *idx, uint32_t rows, uint32_t cols) {
    msm_state *s = ctx->msm;
    int32_t rc;
    if ((rc = ensure_full_table(ctx)) || (rc = ensure_rows(ctx, rows)) || (rc = scalar_mags(ctx, scalars, ld, nullptr, rows, cols))) return rc;
    return msm_windows(ctx, idx, ld, rows, cols, 0, nullptr);
}

// commitments of `rows` rows of `cols` scalars (cols == number of cached generators): every scalar's low byte through the digit
// table (k_msm_codes), then the remaining windows of the (few) rows that hold wider scalars
static int32_t commit_rows(zk_ctx *ctx, const fr_t *scalars, uint64_t ld, uint32_t rows, uint32_t cols) {
    msm_state *s = ctx->msm;
    int32_t rc;
    if ((rc = ensure_full_table(ctx))) return rc;
    const g1a_t *D = s->full;                               // F[0][d][j] = d g_j
    if (!s->full_ready) {
        if ((rc = ensure_digit_table(ctx))) return rc;
        D = s->digit;
    }
    // Rows that hold wide scalars (biases, maxima, the picture: a few whole rows) get their windows 1..31 as 31 virtual rows each of
    // the SAME launches -- found on the device (flags -> compacted list), so nothing waits for the host; MSM_WIDE_CAP bounds the list,
    // more wide rows than that (or no byte table yet) take the separate pass below.
    // With the byte table a virtual row goes through ITS window's table; without it (a fresh generator set) every window goes through the digit table of
    // window 0 and the window sums are combined by doublings afterwards (k_cl_whorner) -- no window table is read by a commitment.
    const uint32_t wide_cap = std::min<uint32_t>(MSM_WIDE_CAP, (65535u - std::min<uint32_t>(rows, 65535u)) / (MSM_WINDOWS - 1));
    const uint32_t vstride = s->full_ready ? 256u * (uint32_t) s->m : 0u;
    const uint32_t nv = wide_cap * (MSM_WINDOWS - 1), rows_all = rows + nv;
    if ((rc = ensure_rows(ctx, rows_all))) return rc;
    if (s->flags_cap < rows + 1) {
        ZK_ORDER();
        if (s->hi_flags) { ZK_HIP(zk_stream_sync(ctx)); ZK_HIP(hipFree(s->hi_flags)); ZK_HIP(hipFree(s->row_list)); }
        ZK_HIP(hipMalloc((void **) &s->hi_flags, ((size_t) rows + 1) * 4));                  // [rows] = number of flagged rows
        ZK_HIP(hipMalloc((void **) &s->row_list, ((size_t) rows + 1) * 4));
        s->flags_cap = rows + 1;
    }
    ZK_STREAM(hipMemsetAsync(s->hi_flags, 0, ((size_t) rows + 1) * 4, ctx->stream));
    if ((rc = regrow(ctx, (void **) &s->codes, &s->codes_cap, (size_t) rows_all * cols * 2))) return rc;
    // columns per lane: 64 would give one wave per row, but the ~2 250 full rows of a vgg11 commitment do not divide evenly over 1 024 SIMDs
    // (some get three such waves, some two: the kernel lasts as long as three); with half rows the spread is 5 against 4.4 on average
    const uint32_t cpt_max = 32;
    const uint32_t cpt = std::max<uint32_t>(1, std::min<uint32_t>(cpt_max, (cols + MSM_BLOCK - 1) / MSM_BLOCK));
    const uint32_t chunks = (cols + MSM_BLOCK * cpt - 1) / (MSM_BLOCK * cpt), n = chunks * MSM_BLOCK;
    if ((rc = regrow(ctx, (void **) &s->partials, &s->partials_cap, (size_t) rows_all * n * sizeof(g1j_t)))) return rc;
    if ((rc = regrow(ctx, (void **) &s->tmpJ, &s->tmp_cap, (size_t) std::max<uint32_t>(wide_cap, 1) * sizeof(g1j_t)))) return rc;
    uint32_t *n_wide = s->hi_flags + rows;
    // (a thread per scalar for a proof on its own; with other proofs in flight eight scalars per thread: an eighth of the workgroups to place -- the fused launch of
    //  eight lanes took 11.5 ms against 8 x 0.2 ms alone)
    const dim3 cgrid(!ctx->live_now && load_shapes() ? std::max<uint32_t>(1, (cols + 2047) / 2048) : std::min<uint32_t>((cols + 255) / 256, 64), 1);
    for (uint32_t r0 = 0; r0 < rows; r0 += 32768)
        zk_launch_d<k_scalar_codes, 256>(ctx, PC_MSM_PLANES, 0.0, dim3(cgrid.x, std::min<uint32_t>(32768, rows - r0)), s->codes + (size_t) r0 * cols,
                  s->hi_flags + r0, scalars + (size_t) r0 * ld, ld, cols);
    if (wide_cap) {
        zk_launch_d<k_compact_flags, 1024>(ctx, PC_MSM_PLANES, 0.0, dim3(1), s->row_list, n_wide, s->hi_flags, rows, wide_cap);
        zk_launch_d<k_scalar_codes_wide, 256>(ctx, PC_MSM_PLANES, 0.0, dim3(cgrid.x, wide_cap), s->codes + (size_t) rows * cols, scalars, ld, s->row_list, n_wide, cols);
    }
    // rows of bits go through the subset-sum table: 8 columns per lookup (only when one block owns a whole row and the table is there)
    const bool bits = s->full_ready && s->t8_ready && s->t8_m == s->m && cols == s->m && rows_all <= 65535 && (cols / 8 / MSM_BLOCK) % chunks == 0 && cols % (MSM_BLOCK * cpt) == 0;
    if (bits) {
        if ((rc = regrow(ctx, (void **) &s->masks, &s->masks_cap, (size_t) rows * (cols / 8) * 2))) return rc;
        zk_launch_d<k_bit_masks, 256>(ctx, PC_MSM_PLANES, 0.0, dim3(2, rows), s->masks, (const uint16_t *) s->codes, (const uint32_t *) s->hi_flags, cols);
    }
    const uint32_t *bflags = bits ? s->hi_flags : nullptr;
    const uint16_t *bmasks = bits ? s->masks : nullptr;
    const g1a_t *bT8 = bits ? s->t8 : nullptr;
    for (uint32_t r0 = 0; r0 < rows_all; r0 += 65535) {           // one launch whenever rows + virtual rows <= 65535 (always, for the circuits here)
        const uint32_t nr = std::min<uint32_t>(65535, rows_all - r0), n_real = std::min<uint32_t>(nr, r0 < rows ? rows - r0 : 0);
        const double bytes = 32.0 * (double) std::min(nr, n_real) * (double) cols;
        if (s->safe)
            zk_launch_d<k_msm_codes<true>, MSM_BLOCK>(ctx, PC_MSM_PLANES, bytes, dim3(chunks, nr), s->partials + (size_t) r0 * n, s->exc,
                      s->codes + (size_t) r0 * cols, D, (uint32_t) s->m, cols, cpt, n_real, n_wide, vstride, bflags, bmasks, bT8);
        else
            zk_launch_d<k_msm_codes<false>, MSM_BLOCK>(ctx, PC_MSM_PLANES, bytes, dim3(chunks, nr), s->partials + (size_t) r0 * n, s->exc,
                      s->codes + (size_t) r0 * cols, D, (uint32_t) s->m, cols, cpt, n_real, n_wide, vstride, bflags, bmasks, bT8);
    }
    ZK_HIP(hipGetLastError());
    if ((rc = reduce_rows(ctx, s->partials, n, rows_all, s->rowsJ, (rows + 3u) & ~3u, wide_cap ? n_wide : nullptr))) return rc;         // rowsJ[rows + v] = sum of virtual row v
    if (wide_cap && vstride) {
        if ((rc = reduce_rows(ctx, s->rowsJ + rows, MSM_WINDOWS - 1, wide_cap, s->tmpJ))) return rc;
        zk_launch_d<k_add_rows, 64>(ctx, PC_MSM_FINISH, 0.0, dim3((wide_cap + 63) / 64), s->rowsJ, s->tmpJ, s->row_list, wide_cap, n_wide);
        ZK_HIP(hipGetLastError());
    } else if (wide_cap) {
        zk_launch_d<k_cl_whorner, 512>(ctx, PC_MSM_FINISH, 0.0, dim3(wide_cap), s->rowsJ, rows, (const uint32_t *) s->row_list, (const uint32_t *) n_wide);
        ZK_HIP(hipGetLastError());
    }
    std::vector<uint32_t> flags((size_t) rows + 2), list;
    ZK_STREAM(hipMemcpyAsync(flags.data(), s->hi_flags, ((size_t) rows + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_STREAM(hipMemcpyAsync(&flags[rows + 1], s->exc, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(zk_stream_sync(ctx));
    if (flags[rows + 1] && !s->safe) return ZK_RETRY_SAFE;
    if (wide_cap && flags[rows] <= wide_cap) return ZK_OK;       // every wide row went through the virtual rows
    bool high = false;
    for (uint32_t r = 0; r < rows; ++r) if (flags[r] & MSM_ROW_WIDE) { list.push_back(r); high |= (flags[r] & MSM_ROW_HIGH) != 0; }
    if (list.empty()) return ZK_OK;
    // separate pass: all wide rows when there was no byte table, the ones beyond the list otherwise (list rows are ascending on both sides)
    if (wide_cap) list.erase(list.begin(), list.begin() + wide_cap);
    const uint32_t nl = (uint32_t) list.size();
    ZK_STREAM(hipMemcpyAsync(s->row_list, list.data(), (size_t) nl * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = regrow(ctx, (void **) &s->tmpJ, &s->tmp_cap, (size_t) nl * sizeof(g1j_t)))) return rc;
    if ((rc = scalar_mags(ctx, scalars, ld, s->row_list, nl, cols))) return rc;
    if ((rc = msm_windows(ctx, nullptr, cols, nl, cols, 1, s->tmpJ, !high))) return rc;
    zk_launch_d<k_add_rows, 64>(ctx, PC_MSM_FINISH, 0.0, dim3((nl + 63) / 64), s->rowsJ, s->tmpJ, s->row_list, nl, (const uint32_t *) nullptr);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// few points: Jacobian -> affine on the host (one inversion each); many: on the device
static int32_t fetch_points(zk_ctx *ctx, uint32_t rows, uint64_t *out) {
    msm_state *s = ctx->msm;
    // short per-thread chains; the single-block scan takes up to 1024 segments -- 256 when other proofs share the GPU (a small block finds a CU: k_aff_scan256)
    const bool small_scan = !ctx->live_now && load_shapes() && rows <= 256 * 64;
    const uint32_t AFF_SEG = std::max<uint32_t>(4, (rows + (small_scan ? 255 : 1023)) / (small_scan ? 256 : 1024));
    const uint32_t nseg = (rows + AFF_SEG - 1) / AFF_SEG;
    uint32_t exc = 0;                    // set by a fast-variant kernel that met P = +-Q: the caller repeats the batch with the SAFE kernels
    if (rows > 8 && nseg <= 1024) {
        int32_t rc = regrow(ctx, &s->aff_scratch, &s->aff_cap, ((size_t) rows + 3 * (size_t) nseg + 2) * sizeof(fp_t));
        if (rc) return rc;
        fp_t *pre = (fp_t *) s->aff_scratch, *seg = pre + rows, *seg_pre = seg + nseg, *seg_suf = seg_pre + nseg, *tot = seg_suf + nseg;
        zk_launch_d<k_aff_prefix, 64>(ctx, PC_MSM_FINISH, 0.0, dim3((nseg + 63) / 64), pre, seg, s->rowsJ, rows, AFF_SEG);
        if (small_scan) zk_launch_d<k_aff_scan256, 256>(ctx, PC_MSM_FINISH, 0.0, dim3(1), seg_pre, seg_suf, tot, seg, nseg);
        else zk_launch_d<k_aff_scan, 1024>(ctx, PC_MSM_FINISH, 0.0, dim3(1), seg_pre, seg_suf, tot, seg, nseg);
        ZK_HIP(hipGetLastError());
        zkff::Fp total, inv;
        ZK_STREAM(hipMemcpyAsync(&total, tot, sizeof(fp_t), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(zk_stream_sync(ctx));
        zkff::Fp::invert(inv, total);                      // the one sequential inversion: O(1) host work, like add_term
        ZK_STREAM(hipMemcpyAsync(tot + 1, &inv, sizeof(fp_t), hipMemcpyHostToDevice, ctx->stream));
        zk_launch_d<k_aff_finish, 64>(ctx, PC_MSM_FINISH, 0.0, dim3((nseg + 63) / 64), s->rowsA, s->rowsJ, pre, seg_pre, seg_suf, tot + 1, rows, AFF_SEG);
        ZK_HIP(hipGetLastError());
        ZK_STREAM(hipMemcpyAsync(out, s->rowsA, (size_t) rows * sizeof(g1a_t), hipMemcpyDeviceToHost, ctx->stream));
        ZK_STREAM(hipMemcpyAsync(&exc, s->exc, 4, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(zk_stream_sync(ctx));
        return exc && !s->safe ? ZK_RETRY_SAFE : ZK_OK;
    }
    if (rows > 8) {
        zk_launch_d<k_to_affine, 64>(ctx, PC_MSM_FINISH, 0.0, dim3((rows + 63) / 64), s->rowsA, s->rowsJ, rows);
        ZK_HIP(hipGetLastError());
        ZK_STREAM(hipMemcpyAsync(out, s->rowsA, (size_t) rows * sizeof(g1a_t), hipMemcpyDeviceToHost, ctx->stream));
        ZK_STREAM(hipMemcpyAsync(&exc, s->exc, 4, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(zk_stream_sync(ctx));
        return exc && !s->safe ? ZK_RETRY_SAFE : ZK_OK;
    }
    std::vector<zkff::G1> pj(rows);
    ZK_STREAM(hipMemcpyAsync(s->h_stage, s->rowsJ, (size_t) rows * sizeof(g1j_t), hipMemcpyDeviceToHost, ctx->stream));
    ZK_STREAM(hipMemcpyAsync(s->h_stage + 8 * sizeof(g1j_t), s->exc, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(zk_stream_sync(ctx));
    std::memcpy(&exc, s->h_stage + 8 * sizeof(g1j_t), 4);
    if (exc && !s->safe) return ZK_RETRY_SAFE;
    std::memcpy((void *) pj.data(), s->h_stage, (size_t) rows * sizeof(g1j_t));
    std::vector<zkff::G1Affine> aff;
    zkff::batchToAffine(pj, aff);                  // one field inversion for all of them
    std::memcpy(out, aff.data(), (size_t) rows * 96);
    return ZK_OK;
}

// runs `body` with the fast kernels; if one of them flagged an exceptional addition, once more with the SAFE variants
template <class Body>
static int32_t with_safe_retry(zk_ctx *ctx, Body body) {
    msm_state *s = ctx->msm;
    s->safe = false;
    ZK_STREAM(hipMemsetAsync(s->exc, 0, 4, ctx->stream));
    int32_t rc = body();
    if (rc != ZK_RETRY_SAFE) return rc;
    s->safe = true;
    ZK_STREAM(hipMemsetAsync(s->exc, 0, 4, ctx->stream));
    rc = body();
    s->safe = false;
    return rc;
}

static_assert(sizeof(zkff::G1) == sizeof(g1j_t) && sizeof(zkff::G1Affine) == sizeof(g1a_t), "host / device point layouts must agree");

#define CHECK_READY() ZK_CHECK_READY()

// rowsJ[i] += blinds[i] * H for i < rows, H = generator number h_index of the cached set (zero-knowledge mode). The blinds are host
// scalars; they go through the generic window kernels as a one-column matrix whose only column is H.
static int32_t add_blinds(zk_ctx *ctx, const uint64_t *blinds, uint32_t rows, uint32_t h_index) {
    msm_state *s = ctx->msm;
    if (h_index >= s->m) { ctx->err = "blinding generator is not part of the cached generator set"; return ZK_ERR_ARG; }
    int32_t rc;
    if ((rc = zk_scratch(ctx, (size_t) rows * 36))) return rc;
    fr_t *d_bl = (fr_t *) ctx->scratch.p;
    uint32_t *d_idx = (uint32_t *) (d_bl + rows);
    std::vector<uint32_t> idx(rows, h_index);
    ZK_STREAM(hipMemcpyAsync(d_bl, blinds, (size_t) rows * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_STREAM(hipMemcpyAsync(d_idx, idx.data(), (size_t) rows * 4, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(zk_stream_sync(ctx));          // idx is a local
    if ((rc = scalar_mags(ctx, d_bl, 1, nullptr, rows, 1))) return rc;
    if (!s->full_ready) {
        // no byte table (a fresh generator set): one row-cooperative double-and-add chain per row over H itself (msm_cl.cuh: k_cl_blind_rows)
        zk_launch_d<k_cl_blind_rows, 256>(ctx, PC_MSM_PLANES, 32.0 * (double) rows, dim3((rows + 15) / 16), s->rowsJ, (const fr_t *) s->mag, (const g1a_t *) (s->tables + h_index), rows);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    if ((rc = regrow(ctx, (void **) &s->tmpJ, &s->tmp_cap, (size_t) rows * sizeof(g1j_t)))) return rc;
    if ((rc = msm_windows(ctx, d_idx, 1, rows, 1, 0, s->tmpJ))) return rc;
    zk_launch_d<k_add_rows, 64>(ctx, PC_MSM_FINISH, 0.0, dim3((rows + 63) / 64), s->rowsJ, s->tmpJ, (const uint32_t *) nullptr, rows, (const uint32_t *) nullptr);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// A generator set that will NOT come again (the reference's semantics: the verifier draws new generators for every proof, reference src/verifier.cpp:119-128)
// never gets a byte table, however often the proof itself uses it (the zero-knowledge mode commits several vectors over one set).
extern "C" int32_t zk_set_generator_reuse(zk_ctx *ctx, int32_t reusable) {
    if (!ctx) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    int32_t rc = ensure_state(ctx);
    if (rc) return rc;
    ctx->msm->no_full = !reusable;
    if (ctx->msm->no_full) ctx->msm->full_ready = ctx->msm->t8_ready = false;
    else if (ctx->msm->gt) gen_adopt(ctx->msm);
    return ZK_OK;
}

// ---- multiples of one base point through its byte-window table (the generators the reference's verifier draws for every proof) ----
// One table per GPU and base point, uploaded when first seen (783 KB); scalars and results go through buffers of the context's MSM state.
struct fixed_base_table { int device; std::vector<uint64_t> key; g1a_t *d_table; };
static std::mutex g_fb_mtx;
static std::vector<fixed_base_table> g_fb_tables;
extern "C" int32_t zk_fixed_base_mul(zk_ctx *ctx, const uint64_t *table_affine, uint64_t table_points, const uint64_t *scalars, uint64_t n, uint64_t *out_jacobian) {
    ZK_CHECK_CTX();
    if (!table_affine || table_points != 32 * 255 || !scalars || !out_jacobian || !n || n > (1u << 24)) { ctx->err = "zk_fixed_base_mul: arguments"; return ZK_ERR_ARG; }
    int32_t rc = ensure_state(ctx);
    if (rc) return rc;
    msm_state *s = ctx->msm;
    g1a_t *T = nullptr;
    {
        std::lock_guard<std::mutex> g(g_fb_mtx);
        for (const fixed_base_table &t : g_fb_tables)
            if (t.device == ctx->device && std::memcmp(t.key.data(), table_affine, 12 * sizeof(uint64_t)) == 0) { T = t.d_table; break; }      // (entry 0 = the base point)
        if (!T) {
            ZK_HIP(hipMalloc((void **) &T, table_points * sizeof(g1a_t)));
            if (hipMemcpy(T, table_affine, table_points * sizeof(g1a_t), hipMemcpyHostToDevice) != hipSuccess) { (void) hipFree(T); ctx->err = "zk_fixed_base_mul: table upload"; return ZK_ERR_HIP; }
            g_fb_tables.push_back({ctx->device, std::vector<uint64_t>(table_affine, table_affine + 12), T});
        }
    }
    if ((rc = regrow(ctx, &s->fb_buf, &s->fb_cap, (size_t) n * (32 + sizeof(g1j_t))))) return rc;
    g1j_t *d_out = (g1j_t *) s->fb_buf;
    uint32_t *d_k = (uint32_t *) (d_out + n);
    ZK_STREAM(hipMemcpyAsync(d_k, scalars, (size_t) n * 32, hipMemcpyHostToDevice, ctx->stream));
    zk_launch_d<k_fixed_base, 64>(ctx, PC_MSM_TABLES, 0.0, dim3((uint32_t) ((n + 63) / 64)), d_out, (const uint32_t *) d_k, (const g1a_t *) T, (uint32_t) n);
    ZK_HIP(hipGetLastError());
    ZK_STREAM(hipMemcpyAsync(out_jacobian, d_out, (size_t) n * sizeof(g1j_t), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(zk_stream_sync(ctx));
    return ZK_OK;
}

static int32_t commit_input_impl(zk_ctx *ctx, const uint64_t *gens, uint64_t n_gens, const uint64_t *blinds, uint64_t *out_comm, uint64_t n_rows) {
    CHECK_READY();
    const dev_layer &L0 = ctx->L[0];
    const int n = L0.d.bit_length, rb = n >> 1, cb = n - rb;
    const uint64_t cols = 1ull << cb;
    if (n_gens != cols + (blinds ? 1 : 0) || n_rows != (1ull << rb) || !gens || !out_comm) { ctx->err = "commit: wrong generator / row count"; return ZK_ERR_ARG; }
    int32_t rc;
    if ((rc = ensure_state(ctx)) || (rc = ensure_tables(ctx, gens, n_gens))) return rc;
    ctx->msm->rb = rb;
    ctx->msm->cb = cb;
    return with_safe_retry(ctx, [&]() -> int32_t {
        int32_t r = commit_rows(ctx, L0.val, cols, (uint32_t) n_rows, (uint32_t) cols);
        if (!r && blinds) r = add_blinds(ctx, blinds, (uint32_t) n_rows, (uint32_t) cols);
        return r ? r : fetch_points(ctx, (uint32_t) n_rows, out_comm);
    });
}

extern "C" int32_t zk_commit_input(zk_ctx *ctx, const uint64_t *gens, uint64_t n_gens, uint64_t *out_comm, uint64_t n_rows) {
    return commit_input_impl(ctx, gens, n_gens, nullptr, out_comm, n_rows);
}
extern "C" int32_t zk_commit_input_blinded(zk_ctx *ctx, const uint64_t *gens, uint64_t n_gens, const uint64_t *blinds, uint64_t *out_comm, uint64_t n_rows) {
    if (!blinds) return ZK_ERR_ARG;
    return commit_input_impl(ctx, gens, n_gens, blinds, out_comm, n_rows);
}

extern "C" int32_t zk_commit_vector(zk_ctx *ctx, const uint64_t *scalars, uint64_t n_rows, uint64_t cols, const uint64_t *blinds, uint64_t *out) {
    CHECK_READY();
    msm_state *s = ctx->msm;
    if (!s || !s->tables || !scalars || !out || !n_rows || n_rows > 65535 || cols + (blinds ? 1 : 0) > s->m || !cols) {
        ctx->err = "commit_vector: no cached generators for this shape";
        return ZK_ERR_ARG;
    }
    int32_t rc;
    { set_lock g(ctx, s->gt); if (g.rc) return g.rc; ++s->gt->uses; }      // the cached set is in use again: worth its byte table
    fr_t *d_v = nullptr;                              // its own buffer: commit_rows and add_blinds use the context's scratch
    ZK_HIP(hipMalloc((void **) &d_v, (size_t) n_rows * cols * 32));
    hipError_t e = hipMemcpyAsync(d_v, scalars, (size_t) n_rows * cols * 32, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { hipFree(d_v); ctx->err = hipGetErrorString(e); return ZK_ERR_HIP; }
    rc = with_safe_retry(ctx, [&]() -> int32_t {
        int32_t r = commit_rows(ctx, d_v, cols, (uint32_t) n_rows, (uint32_t) cols);
        if (!r && blinds) r = add_blinds(ctx, blinds, (uint32_t) n_rows, (uint32_t) cols);
        return r ? r : fetch_points(ctx, (uint32_t) n_rows, out);
    });
    (void) zk_stream_sync(ctx);
    hipFree(d_v);
    return rc;
}

extern "C" int32_t zk_hyrax_combine_rows(zk_ctx *ctx, const uint64_t *x, uint32_t n, uint64_t *out_w) {
    CHECK_READY();
    msm_state *s = ctx->msm;
    if (!s || !s->tables) { ctx->err = "open before commit"; return ZK_ERR_STATE; }
    const dev_layer &L0 = ctx->L[0];
    if ((int) n != L0.d.bit_length || !out_w) return ZK_ERR_ARG;
    const uint32_t rows = 1u << s->rb, m = 1u << s->cb;
    int32_t rc;
    if ((rc = regrow(ctx, (void **) &s->mag, &s->mag_cap, ((size_t) rows + m) * sizeof(fr_t)))) return rc;
    fr_t *Lrow = s->mag, *w = s->mag + rows;
    const HFr *xs = reinterpret_cast<const HFr *>(x);
    if ((rc = zk_eq_table1_dev(ctx, Lrow, s->rb, xs + s->cb, HFr::one()))) return rc;
    if ((rc = zk_col_combine_dev(ctx, w, L0.val, Lrow, m, rows))) return rc;
    ZK_STREAM(hipMemcpyAsync(out_w, w, (size_t) m * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(zk_stream_sync(ctx));
    return ZK_OK;
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
    zk_launch_d<k_fill, 256>(ctx, PC_IPA, 0.0, dim3((m + 255) / 256), s->coef, to_dev(HFr::one()), m);
    ZK_HIP(hipGetLastError());
    s->len = m;
    return ZK_OK;
}

extern "C" int32_t zk_hyrax_open_round(zk_ctx *ctx, uint64_t Lp[12], uint64_t Rp[12], uint64_t yL[4], uint64_t yR[4]) {
    CHECK_READY();
    msm_state *s = ctx->msm;
    if (!s || s->len < 2) return ZK_ERR_STATE;
    const uint32_t m = 1u << s->cb, h = s->len >> 1;
    zk_launch_d<k_ipa_scalars, 256>(ctx, PC_IPA, 0.0, dim3((m + 255) / 256), s->sL, s->idxL, s->sR, s->idxR, s->a, s->coef, m, s->len);
    zk_launch_d<k_ipa_dots, 256>(ctx, PC_IPA, 0.0, dim3(1), s->d_y, s->a, s->b, h);
    ZK_HIP(hipGetLastError());
    // two MSMs over m/2 generators each = the two rows of one batch (row stride m/2 for scalars and indices)
    int32_t rc;
    uint64_t pts[24];
    rc = with_safe_retry(ctx, [&]() -> int32_t {
        int32_t r = run_msm(ctx, s->sL, m / 2, s->idxL, 2, m / 2);
        if (r) return r;
        ZK_STREAM(hipMemcpyAsync(ctx->h_result, s->d_y, 64, hipMemcpyDeviceToHost, ctx->stream));      // (rides on the points' synchronisation: one host turn-around per round)
        return fetch_points(ctx, 2, pts);
    });
    if (rc) return rc;
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
    zk_launch_d<k_ipa_fold, 256>(ctx, PC_IPA, 0.0, dim3((m + 255) / 256), s->a, s->b, s->coef, to_dev(H(c)), m, s->len);
    ZK_HIP(hipGetLastError());
    s->len >>= 1;
    return ZK_OK;
}

extern "C" int32_t zk_hyrax_open_final(zk_ctx *ctx, uint64_t *a, uint32_t cap, uint32_t *n) {
    CHECK_READY();
    msm_state *s = ctx->msm;
    if (!s || !s->len || !a || !n) return ZK_ERR_STATE;
    if (s->len > cap) { ctx->err = "open_final: buffer too small"; return ZK_ERR_ARG; }
    ZK_STREAM(hipMemcpyAsync(a, s->a, (size_t) s->len * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(zk_stream_sync(ctx));
    *n = s->len;
    return ZK_OK;
}

extern "C" int32_t zk_verifier_msm(zk_ctx *ctx, uint64_t out[12], const uint64_t *scalars, const uint64_t *bases, uint64_t n, int32_t bases_are_generators) {
    if (!ctx || !n || n > (1u << 20) || !scalars || !bases || !out) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    int32_t rc;
    if (bases_are_generators) {
        msm_state *s = ctx->msm;
        if (!s || !s->tables || n > s->m || std::memcmp(s->gens_host.data(), bases, n * 96) != 0) { ctx->err = "verifier MSM: these are not the cached generators"; return ZK_ERR_STATE; }
        { set_lock g(ctx, s->gt); if (g.rc) return g.rc; ++s->gt->uses; }
        if ((rc = zk_scratch(ctx, n * 32))) return rc;
        ZK_STREAM(hipMemcpyAsync(ctx->scratch.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
        return with_safe_retry(ctx, [&]() -> int32_t {
            int32_t r = run_msm(ctx, (const fr_t *) ctx->scratch.p, n, nullptr, 1, (uint32_t) n);
            return r ? r : fetch_points(ctx, 1, out);
        });
    }
    std::swap(ctx->msm, ctx->vmsm);                     // the verifier's own tables: the prover's cached generator set is left alone
    rc = ensure_state(ctx);
    if (!rc) ctx->msm->no_full = true;                  // never a 3 GB byte table for points that change with every proof
    if (!rc) rc = ensure_tables(ctx, bases, n);
    if (!rc) rc = zk_scratch(ctx, n * 32);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(ctx->scratch.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { ctx->err = hipGetErrorString(e); rc = ZK_ERR_HIP; }
    }
    if (!rc)
        rc = with_safe_retry(ctx, [&]() -> int32_t {
            int32_t r = run_msm(ctx, (const fr_t *) ctx->scratch.p, n, nullptr, 1, (uint32_t) n);
            return r ? r : fetch_points(ctx, 1, out);
        });
    std::swap(ctx->msm, ctx->vmsm);
    return rc;
}

// stand-alone MSM over arbitrary bases (kernel-level parity tests)
extern "C" int32_t zk_k_msm(zk_ctx *ctx, uint64_t out[12], const uint64_t *scalars, const uint64_t *bases, uint64_t n) {
    if (!ctx || !n || n > (1u << 20)) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = ensure_state(ctx)) || (rc = ensure_tables(ctx, bases, n))) return rc;
    if ((rc = zk_scratch(ctx, n * 32))) return rc;
    ZK_STREAM(hipMemcpyAsync(ctx->scratch.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    return with_safe_retry(ctx, [&]() -> int32_t {
        int32_t r = run_msm(ctx, (const fr_t *) ctx->scratch.p, n, nullptr, 1, (uint32_t) n);
        return r ? r : fetch_points(ctx, 1, out);
    });
}

extern "C" int32_t zk_k_fp_inv(zk_ctx *ctx, uint64_t *out, const uint64_t *in, uint64_t n) {
    if (!ctx || !out || !in || !n || n > (1u << 20)) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = zk_scratch(ctx, 2 * n * sizeof(fp_t)))) return rc;
    fp_t *d_in = (fp_t *) ctx->scratch.p, *d_out = d_in + n;
    ZK_STREAM(hipMemcpyAsync(d_in, in, n * sizeof(fp_t), hipMemcpyHostToDevice, ctx->stream));
    zk_launch_d<k_fp_inv, 64>(ctx, PC_MSM_FINISH, 0.0, dim3((uint32_t) ((n + 63) / 64)), d_out, (const fp_t *) d_in, (uint32_t) n);
    ZK_HIP(hipGetLastError());
    ZK_STREAM(hipMemcpyAsync(out, d_out, n * sizeof(fp_t), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(zk_stream_sync(ctx));
    return ZK_OK;
}

// row-cooperative arithmetic (fpc_dev.cuh) on caller data: products / sums / differences of n pairs of Fp elements
extern "C" int32_t zk_k_fpc_ops(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) {
    if (!ctx || !out || !a || !b || !n || n > (1u << 20)) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = zk_scratch(ctx, 5 * n * sizeof(fp_t)))) return rc;
    fp_t *d_a = (fp_t *) ctx->scratch.p, *d_b = d_a + n, *d_o = d_b + n;
    ZK_STREAM(hipMemcpyAsync(d_a, a, n * sizeof(fp_t), hipMemcpyHostToDevice, ctx->stream));
    ZK_STREAM(hipMemcpyAsync(d_b, b, n * sizeof(fp_t), hipMemcpyHostToDevice, ctx->stream));
    zk_launch_d<k_fpc_ops, 256>(ctx, PC_MSM_FINISH, 0.0, dim3((uint32_t) ((n * 16 + 255) / 256)), d_o, (const fp_t *) d_a, (const fp_t *) d_b, (uint32_t) n);
    ZK_HIP(hipGetLastError());
    ZK_STREAM(hipMemcpyAsync(out, d_o, 3 * n * sizeof(fp_t), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(zk_stream_sync(ctx));
    return ZK_OK;
}
// ... sums p[i] + q[i] (out[0 .. n)) and doubles 2 p[i] (out[n .. 2n)) of n pairs of Jacobian points
extern "C" int32_t zk_k_cl_add(zk_ctx *ctx, uint64_t *out, const uint64_t *p, const uint64_t *q, uint64_t n) {
    if (!ctx || !out || !p || !q || !n || n > (1u << 18)) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = zk_scratch(ctx, 4 * n * sizeof(g1j_t)))) return rc;
    g1j_t *d_p = (g1j_t *) ctx->scratch.p, *d_q = d_p + n, *d_o = d_q + n;
    ZK_STREAM(hipMemcpyAsync(d_p, p, n * sizeof(g1j_t), hipMemcpyHostToDevice, ctx->stream));
    ZK_STREAM(hipMemcpyAsync(d_q, q, n * sizeof(g1j_t), hipMemcpyHostToDevice, ctx->stream));
    zk_launch_d<k_cl_add, 256>(ctx, PC_MSM_FINISH, 0.0, dim3((uint32_t) ((n * 16 + 255) / 256)), d_o, (const g1j_t *) d_p, (const g1j_t *) d_q, (uint32_t) n);
    ZK_HIP(hipGetLastError());
    ZK_STREAM(hipMemcpyAsync(out, d_o, 2 * n * sizeof(g1j_t), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(zk_stream_sync(ctx));
    return ZK_OK;
}
// ... and the reduction tree: out[s] = sum of the n_seg points of segment s (k_cl_tree, runs of `n_in` per block, as many levels as it takes)
extern "C" int32_t zk_k_cl_tree(zk_ctx *ctx, uint64_t *out, const uint64_t *pts, uint64_t n_seg, uint64_t segs, uint32_t n_in) {
    if (!ctx || !out || !pts || !n_seg || !segs || segs > 65535 || n_seg * segs > (1u << 20) || !n_in) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    int32_t rc;
    const size_t total = n_seg * segs;
    if ((rc = zk_scratch(ctx, 3 * total * sizeof(g1j_t)))) return rc;
    g1j_t *cur = (g1j_t *) ctx->scratch.p, *nxt = cur + total, *res = nxt + total;
    ZK_STREAM(hipMemcpyAsync(cur, pts, total * sizeof(g1j_t), hipMemcpyHostToDevice, ctx->stream));
    for (uint32_t n = (uint32_t) n_seg;;) {
        const uint32_t step = std::min<uint32_t>(n, n_in), blocks = (n + step - 1) / step;
        zk_launch_d<k_cl_tree, 512>(ctx, PC_MSM_FINISH, 0.0, dim3(blocks, (uint32_t) segs), blocks == 1 ? res : nxt, (const g1j_t *) cur, n, step);
        if (blocks == 1) break;
        std::swap(cur, nxt);
        n = blocks;
    }
    ZK_HIP(hipGetLastError());
    ZK_STREAM(hipMemcpyAsync(out, res, segs * sizeof(g1j_t), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(zk_stream_sync(ctx));
    return ZK_OK;
}

// rows x cols row commitments over `cols` arbitrary bases: the commitInput data path on caller-supplied data
extern "C" int32_t zk_k_commit_rows(zk_ctx *ctx, uint64_t *out, const uint64_t *scalars, const uint64_t *bases, uint64_t rows,
                                    uint64_t cols) {
    if (!ctx || !rows || !cols || rows * cols > (1ull << 26)) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = ensure_state(ctx)) || (rc = ensure_tables(ctx, bases, cols))) return rc;
    if ((rc = zk_scratch(ctx, rows * cols * 32))) return rc;
    ZK_STREAM(hipMemcpyAsync(ctx->scratch.p, scalars, rows * cols * 32, hipMemcpyHostToDevice, ctx->stream));
    return with_safe_retry(ctx, [&]() -> int32_t {
        int32_t r = commit_rows(ctx, (const fr_t *) ctx->scratch.p, cols, (uint32_t) rows, (uint32_t) cols);
        return r ? r : fetch_points(ctx, (uint32_t) rows, out);
    });
}

// libzkcnn_hip.so: the prover context -- allocation, work buffers, hand-over slots and mailboxes, the built-in profiler, the run-time
// options of a context (Fiat-Shamir chain, hybrid tail, resident rounds) and the per-proof bracket behind the resident-kernel policy.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <condition_variable>
#include "ctx.hpp"
#include "tail_types.hpp"

static std::string g_create_err;
static std::atomic<int> g_live_contexts[64];
int zk_contexts_on_device(int device) { return g_live_contexts[device & 63].load(); }
void zk_set_create_error(const std::string &msg) { g_create_err = msg; }

// ------------------------------------------------------------------------------------------------
// memory helpers
// ------------------------------------------------------------------------------------------------
int32_t zk_dev_alloc(zk_ctx *ctx, void **p, size_t bytes) {
    if (bytes == 0) bytes = 32;
    ZK_HIP(hipMalloc(p, bytes));
    // ZKCNN_ALLOC_TRACE=1: every tracked allocation of 64 MB or more on stderr (where a session's HBM goes: DESIGN.md section 6, round 5)
    static const bool trace = getenv("ZKCNN_ALLOC_TRACE") != nullptr;
    if (trace && bytes >= ((size_t) 64 << 20)) fprintf(stderr, "[zkcnn alloc] %s %.3f GB (allocation %zu of this context)\n", ctx->alloc_sink ? "shared" : "session", bytes / 1e9, ctx->owned.size());
    (ctx->alloc_sink ? *ctx->alloc_sink : ctx->owned).push_back(*p);       // (the static part of a circuit belongs to its registry entry)
    if (ctx->alloc_sink) ctx->sink_bytes += bytes;
    return ZK_OK;
}
int32_t zk_scratch(zk_ctx *ctx, size_t bytes) {
    if (ctx->scratch.bytes >= bytes) return ZK_OK;
    if (ctx->scratch.p) {
        ZK_HIP(hipStreamSynchronize(ctx->stream));
        ZK_HIP(hipFree(ctx->scratch.p));
        ctx->scratch.p = nullptr;
        ctx->scratch.bytes = 0;
    }
    size_t want = bytes + bytes / 4;
    ZK_HIP(hipMalloc(&ctx->scratch.p, want));
    ctx->scratch.bytes = want;
    return ZK_OK;
}

extern "C" int32_t zk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int32_t zk_ctx_create(int32_t device, zk_ctx **out) {
    if (!out) return ZK_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_err = std::string("no HIP device: ") + hipGetErrorString(e);
        return ZK_ERR_HIP;
    }
    if (device < 0 || device >= n) { g_create_err = "device index out of range"; return ZK_ERR_ARG; }
    zk_ctx *ctx = new zk_ctx();
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_err = hipGetErrorString(e);
        delete ctx;
        return ZK_ERR_HIP;
    }
    // fixed-size work buffers
    ctx->partial_blocks = 4096;
    if (zk_dev_alloc(ctx, (void **) &ctx->partials, (size_t) ctx->partial_blocks * 4 * 32) ||
        zk_dev_alloc(ctx, (void **) &ctx->d_result, 32 * 32) ||
        zk_dev_alloc(ctx, &ctx->d_bcast, sizeof(mid_bcast)) || hipMemsetAsync(ctx->d_bcast, 0, sizeof(mid_bcast), ctx->stream) != hipSuccess ||
        zk_dev_alloc(ctx, (void **) &ctx->d_counter, 64) ||
        hipHostMalloc((void **) &ctx->h_slot, sizeof(*ctx->h_slot), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&ctx->d_slot, ctx->h_slot, 0) != hipSuccess ||
        hipHostMalloc((void **) &ctx->h_aux, sizeof(*ctx->h_aux), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&ctx->d_aux, ctx->h_aux, 0) != hipSuccess ||
        hipMemsetAsync(ctx->d_counter, 0, 64, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipHostMalloc((void **) &ctx->h_result, 32 * 32) != hipSuccess ||
        hipHostMalloc((void **) &ctx->h_tail, std::max(sizeof(tail_out), sizeof(export_out)), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&ctx->d_tail, ctx->h_tail, 0) != hipSuccess ||
        hipHostMalloc((void **) &ctx->h_live_in, sizeof(live_in), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&ctx->d_live_in, ctx->h_live_in, 0) != hipSuccess) {
        g_create_err = ctx->err.empty() ? "allocation failed" : ctx->err;
        zk_ctx_destroy(ctx);
        return ZK_ERR_NOMEM;
    }
    std::memset((void *) ctx->h_slot, 0, sizeof(*ctx->h_slot));
    std::memset((void *) ctx->h_aux, 0, sizeof(*ctx->h_aux));
    std::memset(ctx->h_tail, 0, std::max(sizeof(tail_out), sizeof(export_out)));
    std::memset(ctx->h_live_in, 0, sizeof(live_in));
    ++g_live_contexts[device & 63];
    ctx->counted_context = true;
    *out = ctx;
    return ZK_OK;
}

static void first_gate_leave(zk_ctx *ctx, bool completed);
extern "C" void zk_ctx_destroy(zk_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->batch) (void) zk_batch_detach(ctx->batch, ctx);
    if (ctx->counted_context) { --g_live_contexts[ctx->device & 63]; ctx->counted_context = false; }
    if (ctx->live_active) (void) zk_live_abort(ctx);
    first_gate_leave(ctx, false);       // (a context destroyed inside its first proof -- an error path that skipped zk_proof_end -- must not leave the others waiting)
    (void) zk_proof_end(ctx);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->n_seg) fprintf(stderr, "[zkcnn timing] quadratic round call: %.2f us between calls (verifier + wrappers), %.2f plan + launch, %.2f waiting for the result, %.2f after (averages over %llu rounds)\n",
                            1e6 * ctx->t_seg[0] / ctx->n_seg, 1e6 * ctx->t_seg[1] / ctx->n_seg, 1e6 * ctx->t_seg[2] / ctx->n_seg, 1e6 * ctx->t_seg[3] / ctx->n_seg, (unsigned long long) ctx->n_seg);
    if (ctx->live_timed_rounds)
        fprintf(stderr, "[zkcnn timing] resident round kernel: %.2f us per round from posting the challenge to reading the polynomial, %.2f us on the host between "
                        "two round calls (%llu rounds after a kernel's first); kernel clock: %.2f us per round in total, %.2f of them polling for the challenge (%llu rounds, %llu kernels)\n",
                1e6 * ctx->live_t_gpu / ctx->live_timed_rounds, 1e6 * ctx->live_t_host / ctx->live_timed_rounds, (unsigned long long) ctx->live_timed_rounds,
                0.01 * ctx->live_ticks_total / std::max<uint64_t>(ctx->live_rounds_total, 1), 0.01 * ctx->live_ticks_wait / std::max<uint64_t>(ctx->live_rounds_total, 1),
                (unsigned long long) ctx->live_rounds_total, (unsigned long long) ctx->live_phases_total);
    zk_msm_destroy(ctx);
    for (prof_pending &p : ctx->prof_q) { hipEventDestroy(p.e0); hipEventDestroy(p.e1); }
    for (hipEvent_t e : ctx->prof_pool) hipEventDestroy(e);
    for (void *p : ctx->owned) hipFree(p);
    zk_circuit_release(ctx);
    if (ctx->scratch.p) hipFree(ctx->scratch.p);
    if (ctx->w_val0.p) hipFree(ctx->w_val0.p);
    for (dev_buf &b : ctx->w_stage) if (b.p) hipFree(b.p);
    if (ctx->h_result) hipHostFree(ctx->h_result);
    if (ctx->h_slot) hipHostFree((void *) ctx->h_slot);
    if (ctx->h_aux) hipHostFree((void *) ctx->h_aux);
    if (ctx->h_tail) hipHostFree(ctx->h_tail);
    if (ctx->h_live_in) hipHostFree(ctx->h_live_in);
    if (ctx->h_liu_tabs) hipHostFree(ctx->h_liu_tabs);
    if (ctx->h_wp_ranges) hipHostFree(ctx->h_wp_ranges);
    if (ctx->h_conv_tabs) hipHostFree(ctx->h_conv_tabs);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int32_t zk_profile_enable(zk_ctx *ctx, uint32_t class_mask) {
    if (!ctx) return ZK_ERR_ARG;
    ctx->prof_mask = class_mask;
    return ZK_OK;
}

extern "C" int32_t zk_profile_report(zk_ctx *ctx, char *buf, uint64_t cap, int32_t reset) {
    if (!ctx || !buf || !cap) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    // ZKCNN_PROF_DUMP=<file>: every profiled launch as a line "class bytes ms" (scripts/exp/launch_sizes.py: time against size within a class)
    static const char *dump_path = getenv("ZKCNN_PROF_DUMP");
    FILE *dump = dump_path && !ctx->prof_q.empty() ? fopen(dump_path, "a") : nullptr;
    for (prof_pending &p : ctx->prof_q) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            ctx->prof_ms[p.cls] += ms;
            ctx->prof_bytes[p.cls] += p.bytes;
            ++ctx->prof_cnt[p.cls];
            if (dump) fprintf(dump, "%s %.0f %.6f\n", prof_names[p.cls], p.bytes, ms);
        }
        ctx->prof_pool.push_back(p.e0);
        ctx->prof_pool.push_back(p.e1);
    }
    ctx->prof_q.clear();
    if (dump) fclose(dump);
    std::string js = "{";
    for (int c = 0; c < PC_COUNT; ++c) {
        char line[256];
        std::snprintf(line, sizeof(line), "%s\"%s\": {\"ms\": %.6f, \"launches\": %llu, \"bytes\": %.0f}", c ? ", " : "", prof_names[c],
                      ctx->prof_ms[c], (unsigned long long) ctx->prof_cnt[c], ctx->prof_bytes[c]);
        js += line;
    }
    js += "}";
    std::snprintf(buf, cap, "%s", js.c_str());
    if (reset)
        for (int c = 0; c < PC_COUNT; ++c) { ctx->prof_ms[c] = 0; ctx->prof_bytes[c] = 0; ctx->prof_cnt[c] = 0; }
    return ZK_OK;
}

extern "C" int32_t zk_fs_attach(zk_ctx *ctx, const uint32_t *state, const uint64_t *pending) {
    if (!ctx || ((state == nullptr) != (pending == nullptr))) return ZK_ERR_ARG;
    if (ctx->batch) state = nullptr, pending = nullptr;       // a lane of a batch takes its challenges from its driver: same transcript (the chain is a function of the messages)
    ctx->fs_state = state;
    ctx->fs_pending = pending;
    ctx->tail_active = false;
    ctx->host_tail_active = false;
    ctx->last_poly_valid = false;
    return ZK_OK;
}
extern "C" int32_t zk_set_host_tail(zk_ctx *ctx, int32_t log_entries) {
    if (!ctx || log_entries > 8) return ZK_ERR_ARG;
    ctx->host_tail_log = log_entries < -1 ? -2 : log_entries;          // -1: the default (off; 2^5 for a lane of a batch), -2: off
    return ZK_OK;
}
static std::atomic<int> g_active_proofs[64];
// The FIRST proof of a process runs alone: HIP loads a kernel's code lazily at its first launch, and dozens of host threads racing through the first
// launch of the same kernels is the one thing the failing runs of rounds 3 and 4 had in common (one GPU memory fault in ~25 bench runs, one SIGSEGV
// inside hipLaunchKernel under rocprofv3 -- both while 8 / 32 sessions ran their first proofs side by side; 6e6 launches with 3.3 KB argument blocks
// from 32 threads did NOT reproduce it: scripts/exp/kernarg_stress.hip). Whoever begins the first proof holds this lock until it ends; proofs that
// begin meanwhile wait. Two gates, because the two launch forms load different kernels: one for the first proof of a context on its own (k_run<F>), one for
// the first proof of a lock-step batch (k_run_b<F> / k_run_p<F>: a solo warm-up never loads those). A gate has no owning THREAD (round-4 advisor finding: a
// std::mutex held across C-API calls is undefined behaviour when released from another thread, and is never released when a caller skips zk_proof_end): it is
// a state word under a short-lived mutex plus a condition variable; the holder is a context or a batch (its lanes share a thread, so they all pass while their
// batch holds the gate); zk_proof_end and zk_ctx_destroy release it, a failed first proof hands the role to the next one.
struct first_gate {
    std::mutex m;
    std::condition_variable cv;
    int state = 0;              // 0: nobody has begun, 1: the first proof is running, 2: done
    const void *owner = nullptr;
    int holders = 0;            // contexts of the owner inside their proofs
};
static first_gate g_first_gate[2];          // [0] solo contexts, [1] batches
static std::atomic<int> g_first_done[2];
static void first_gate_enter(zk_ctx *ctx) {
    const int which = ctx->batch ? 1 : 0;
    if (g_first_done[which].load(std::memory_order_acquire) || ctx->first_gate_held) return;
    first_gate &g = g_first_gate[which];
    const void *key = ctx->batch ? (const void *) ctx->batch : (const void *) ctx;
    std::unique_lock<std::mutex> lk(g.m);
    while (g.state == 1 && g.owner != key) g.cv.wait(lk);
    if (g.state == 2) return;
    g.state = 1;
    g.owner = key;
    ++g.holders;
    ctx->first_gate_held = which + 1;
}
static void first_gate_leave(zk_ctx *ctx, bool completed) {
    if (!ctx->first_gate_held) return;
    const int which = ctx->first_gate_held - 1;
    ctx->first_gate_held = 0;
    first_gate &g = g_first_gate[which];
    {
        std::lock_guard<std::mutex> lk(g.m);
        if (--g.holders > 0) return;
        g.state = completed ? 2 : 0;          // (a first proof that did not complete has loaded who knows what: the next one is the first again)
        g.owner = nullptr;
        if (completed) g_first_done[which].store(1, std::memory_order_release);
    }
    g.cv.notify_all();
}
extern "C" int32_t zk_proof_begin(zk_ctx *ctx) {
    if (!ctx) return ZK_ERR_ARG;
    first_gate_enter(ctx);
    if (!ctx->counted_active) { ++g_active_proofs[ctx->device & 63]; ctx->counted_active = true; }
    // resident kernels only for a proof that is alone on its GPU when it starts: their workgroups wait for one another (k_mid) and for the host,
    // which is only safe -- and only profitable -- while nothing else competes for the CUs and the hardware queue for long
    ctx->live_now = !ctx->batch && g_active_proofs[ctx->device & 63].load() <= 1;       // (a lane of a batch never is: its rounds are fused launches)
    return ZK_OK;
}
extern "C" int32_t zk_proof_end(zk_ctx *ctx) {
    if (!ctx) return ZK_ERR_ARG;
    first_gate_leave(ctx, ctx->counted_active);
    if (ctx->counted_active) { --g_active_proofs[ctx->device & 63]; ctx->counted_active = false; }
    ctx->live_now = !ctx->batch;
    return ZK_OK;
}
extern "C" int32_t zk_set_live_rounds(zk_ctx *ctx, int32_t on) {
    if (!ctx) return ZK_ERR_ARG;
    if (ctx->live_active) { int32_t rc = zk_live_abort(ctx); if (rc) return rc; }
    ctx->live_rounds = on != 0;
    return ZK_OK;
}
extern "C" int32_t zk_host_tail_stats(zk_ctx *ctx, uint64_t *rounds) {
    if (!ctx || !rounds) return ZK_ERR_ARG;
    *rounds = ctx->host_tail_rounds_total;
    return ZK_OK;
}
extern "C" int32_t zk_fs_stats(zk_ctx *ctx, uint64_t *rounds, uint64_t *phases) {
    if (!ctx || !rounds || !phases) return ZK_ERR_ARG;
    *rounds = ctx->tail_rounds_total + ctx->live_rounds_total;
    *phases = ctx->tail_phases_total + ctx->live_phases_total;
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// lock-step batches (ctx.hpp: zk_batch; the kernels' batched forms and the flush of the sumcheck unit: batch.cuh)
// ------------------------------------------------------------------------------------------------
static std::string g_batch_err;
extern "C" int32_t zk_batch_create(int32_t device, zk_batch **out) {
    if (!out) return ZK_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { g_batch_err = "no such HIP device"; return ZK_ERR_ARG; }
    zk_batch *b = new zk_batch();
    b->device = device;
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_batch_err = hipGetErrorString(e);
        delete b;
        return ZK_ERR_HIP;
    }
    b->pending.reserve(16 * ZK_BATCH_MAX_LANES);
    const size_t ring_bytes = (size_t) ZK_BATCH_RING_SLOTS * ZK_BATCH_MAX_LANES * ZK_BATCH_ARG_BYTES;
    bool ok = hipHostMalloc((void **) &b->h_ring, ring_bytes, hipHostMallocMapped | hipHostMallocNonCoherent) == hipSuccess &&
              hipHostGetDevicePointer((void **) &b->d_ring, b->h_ring, 0) == hipSuccess;
    for (int q = 0; ok && q < 4; ++q) ok = hipEventCreateWithFlags(&b->ring_ev[q], hipEventDisableTiming) == hipSuccess;
    if (!ok) { g_batch_err = "zk_batch_create: no pinned memory for the argument ring"; zk_batch_destroy(b); return ZK_ERR_NOMEM; }
    *out = b;
    return ZK_OK;
}
extern "C" int32_t zk_batch_attach(zk_batch *b, zk_ctx *ctx, int32_t *lane) {
    if (!b || !ctx) return ZK_ERR_ARG;
    if (ctx->batch || ctx->device != b->device || b->lanes.size() >= ZK_BATCH_MAX_LANES) { b->err = "zk_batch_attach: the context is a lane already, lives on another device, or the batch is full"; return ZK_ERR_ARG; }
    if (!ctx->circuit_ready || !ctx->circuit) { b->err = "zk_batch_attach: the context holds no circuit"; return ZK_ERR_STATE; }
    if (!b->lanes.empty() && b->lanes[0]->circuit != ctx->circuit) { b->err = "zk_batch_attach: not the resident circuit of lane 0"; return ZK_ERR_STATE; }
    if (hipSetDevice(ctx->device) != hipSuccess) return ZK_ERR_HIP;
    if (ctx->live_active) { int32_t rc = zk_live_abort(ctx); if (rc) return rc; }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return ZK_ERR_HIP;
    ctx->fs_state = nullptr;
    ctx->fs_pending = nullptr;
    // A lane launches on the batch's stream; its own stream is not kept around idle: 56 lanes' streams next to 7 batch streams are 56 HIP streams (hardware-queue
    // slots, profiler buffers -- rocprofv3 --kernel-trace crashed processes with >= 40 session streams in round 4) that nothing uses. It is created again at detach.
    if (hipStreamDestroy(ctx->stream) != hipSuccess) return ZK_ERR_HIP;
    ctx->own_stream = nullptr;
    ctx->stream = b->stream;
    ctx->batch = b;
    ctx->live_now = false;
    ctx->lane = (int) b->lanes.size();
    b->lanes.push_back(ctx);
    if (lane) *lane = ctx->lane;
    return ZK_OK;
}
extern "C" int32_t zk_batch_detach(zk_batch *b, zk_ctx *ctx) {
    if (!b || !ctx || ctx->batch != b) return ZK_ERR_ARG;
    (void) hipSetDevice(b->device);
    (void) zk_batch_flush(b);
    (void) hipStreamSynchronize(b->stream);
    ctx->stream = nullptr;
    // (round-5 advisor finding: a detached context without a stream of its own would run everything on the legacy NULL stream, which orders differently
    //  against the non-blocking streams -- it is marked unusable instead, and the caller is told)
    const bool no_stream = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess;
    if (no_stream) {
        (void) hipGetLastError();
        ctx->stream = nullptr;
        ctx->circuit_ready = false;
        ctx->err = b->err = "zk_batch_detach: no stream for the detached context (the context is unusable)";
    }
    ctx->batch = nullptr;
    ctx->n_pending = 0;
    ctx->live_now = true;
    for (size_t i = 0; i < b->lanes.size(); ++i)
        if (b->lanes[i] == ctx) { b->lanes.erase(b->lanes.begin() + i); break; }
    for (size_t i = 0; i < b->lanes.size(); ++i) b->lanes[i]->lane = (int) i;
    ctx->lane = -1;
    return no_stream ? ZK_ERR_HIP : ZK_OK;
}
extern "C" void zk_batch_destroy(zk_batch *b) {
    if (!b) return;
    if (getenv("ZKCNN_BATCH_TRACE")) {
        fprintf(stderr, "[zkcnn batch] %llu fused launches for %llu lane launches, %llu flushes (%llu empty)\n", (unsigned long long) b->n_launches,
                (unsigned long long) b->n_lane_launches, (unsigned long long) b->n_flushes, (unsigned long long) b->n_empty_flushes);
        for (const auto &kv : b->by_kernel) {
            std::string nm = kv.first;
            const size_t at = nm.find("F = ");
            if (at != std::string::npos) nm = nm.substr(at + 4);
            if (nm.size() > 90) nm.resize(90);
            fprintf(stderr, "[zkcnn batch]   %8llu launches %8llu lanes  %s\n", (unsigned long long) kv.second.first, (unsigned long long) kv.second.second, nm.c_str());
        }
    }
    while (!b->lanes.empty()) (void) zk_batch_detach(b, b->lanes.back());
    (void) hipSetDevice(b->device);
    if (b->stream) { (void) hipStreamSynchronize(b->stream); (void) hipStreamDestroy(b->stream); }
    for (int q = 0; q < 4; ++q) if (b->ring_ev[q]) (void) hipEventDestroy(b->ring_ev[q]);
    if (b->h_ring) (void) hipHostFree(b->h_ring);
    delete b;
}
extern "C" int32_t zk_batch_set_yield(zk_batch *b, zk_yield_fn fn, void *user) {
    if (!b) return ZK_ERR_ARG;
    b->yield_fn = fn;
    b->yield_user = user;
    return ZK_OK;
}
// one slot of the pinned argument ring (launch.cuh: argument blocks too large for the kernel argument segment)
int32_t zk_batch_ring_slot(zk_batch *b, size_t bytes, void **host, void **dev) {
    const size_t slot_bytes = (size_t) ZK_BATCH_MAX_LANES * ZK_BATCH_ARG_BYTES;
    if (bytes > slot_bytes) { b->err = "argument ring: block too large"; return ZK_ERR_ARG; }
    const uint32_t slot = b->ring_next, quarter = ZK_BATCH_RING_SLOTS / 4;
    // entering a quarter of the ring: whatever read it one lap ago must be done; leaving one: mark the stream
    if (slot % quarter == 0) {
        const uint32_t q = slot / quarter, prev = (q + 3) % 4;
        if (b->ring_ev_set[q] && hipEventSynchronize(b->ring_ev[q]) != hipSuccess) { b->err = "argument ring: event wait failed"; return ZK_ERR_HIP; }
        if (b->ring_count) {                 // (the quarter just left: its last reader is on the stream by now)
            if (hipEventRecord(b->ring_ev[prev], b->stream) != hipSuccess) { b->err = "argument ring: event record failed"; return ZK_ERR_HIP; }
            b->ring_ev_set[prev] = true;
        }
    }
    *host = b->h_ring + (size_t) slot * slot_bytes;
    *dev = b->d_ring + (size_t) slot * slot_bytes;
    b->ring_next = (slot + 1) % ZK_BATCH_RING_SLOTS;
    ++b->ring_count;
    return ZK_OK;
}

// every deferred launch, generation by generation: the g-th deferred launch of every lane together, one launch per kernel (lanes in lock
// step defer the same kernel at the same position; anything else still runs, in as many launches as there are kernels)
extern "C" int32_t zk_batch_flush(zk_batch *b) {
    if (!b) return ZK_ERR_ARG;
    ++b->n_flushes;
    if (b->pending.empty()) { ++b->n_empty_flushes; return ZK_OK; }
    if (hipSetDevice(b->device) != hipSuccess) { b->err = "hipSetDevice failed"; return ZK_ERR_HIP; }
    std::vector<batch_item> items;
    items.swap(b->pending);
    b->pending.reserve(16 * ZK_BATCH_MAX_LANES);
    uint32_t gens = 0;
    for (const batch_item &it : items) gens = std::max(gens, it.gen + 1);
    for (zk_ctx *c : b->lanes) c->n_pending = 0;
    std::vector<char> done(items.size(), 0);
    int32_t rc = ZK_OK;
    for (uint32_t g = 0; g < gens && rc == ZK_OK; ++g)
        for (size_t i = 0; i < items.size() && rc == ZK_OK; ++i) {
            if (done[i] || items[i].gen != g) continue;
            const batch_item *group[ZK_BATCH_MAX_LANES];
            uint32_t n = 0, gx = 1, gy = 1;
            double bytes = 0;
            for (size_t j = i; j < items.size() && n < ZK_BATCH_MAX_LANES; ++j)
                if (!done[j] && items[j].gen == g && items[j].launch == items[i].launch &&
                    (!items[i].exact || (items[j].gx == items[i].gx && items[j].gy == items[i].gy))) {
                    group[n++] = &items[j];
                    done[j] = 1;
                    gx = std::max(gx, items[j].gx);
                    gy = std::max(gy, items[j].gy);
                    bytes += items[j].bytes;
                }
            zk_ctx *ctx = items[i].ctx;          // (the profiler books a fused launch on its first lane)
            prof_begin(ctx, items[i].prof_class, bytes);
            rc = items[i].launch(b, group, n, gx, gy);
            prof_end(ctx, items[i].prof_class);
            ++b->n_launches;
            b->n_lane_launches += n;
            std::pair<uint64_t, uint64_t> &st = b->by_kernel[items[i].name];
            ++st.first;
            st.second += n;
        }
    if (rc != ZK_OK && b->err.empty()) b->err = "a fused launch failed";
    return rc;
}
int32_t zk_batch_sync_point(zk_ctx *ctx) {
    zk_batch *b = ctx->batch;
    if (!b) return ZK_OK;
    if (b->yield_fn) b->yield_fn(b->yield_user, ctx->lane);
    if (ctx->n_pending) {                    // nobody has flushed for this lane (no driver): it does so itself
        int32_t rc = zk_batch_flush(b);
        if (rc) ctx->err = b->err;
        return rc;
    }
    return ZK_OK;
}
extern "C" int32_t zk_batch_stats(const zk_batch *b, uint64_t out[4]) {
    if (!b || !out) return ZK_ERR_ARG;
    out[0] = b->n_launches; out[1] = b->n_lane_launches; out[2] = b->n_flushes; out[3] = b->lanes.size();
    return ZK_OK;
}
extern "C" const char *zk_batch_last_error(const zk_batch *b) { return b ? b->err.c_str() : g_batch_err.c_str(); }

extern "C" const char *zk_last_error(const zk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
extern "C" uint64_t zk_proof_bytes(const zk_ctx *ctx) { return ctx->proof_size; }


// Internal state of one prover context (one GPU). Shared by sumcheck.hip and hyrax.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../../include/zkcnn_hip.h"
#include "../ff/fr.hpp"
#include "types.cuh"

typedef zkff::Fr HFr;      // host-side field element (same 32-byte layout as fr_t)

struct dev_buf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct dev_layer {
    zk_layer_desc d;               // copy of the descriptor (host pointers nulled)
    fr_t *val = nullptr;           // 2^bit_length entries, zero padded
    uint64_t val_len = 0;
    uint64_t val_live = 0;         // 1 + index of the last non-zero value (measured when the values arrive or are recomputed)
    // phase-1 lists: gates whose u operand lives in table b (0: layer-0 subset, 1: previous layer), sorted by u
    gate_rec *p1[2] = {nullptr, nullptr};
    uint64_t n_p1[2] = {0, 0};      // records incl. padding (runs of equal keys padded to multiples of GATE_GROUP)
    uint64_t n_p1_real[2] = {0, 0}, n_p2_real[2] = {0, 0};
    uint32_t p1_G[2] = {4, 4}, p2_G[2] = {4, 4};       // records per thread of the scatter kernel (runs are padded to multiples of it)
    uint64_t n_p1_uni[2] = {0, 0};  // how many of them are uni gates (for the algorithmic byte count)
    // phase-2 lists: bin gates whose v operand lives in table b, sorted by v
    gate_rec *p2[2] = {nullptr, nullptr};
    uint64_t n_p2[2] = {0, 0};
    uint32_t p1_live[2] = {0, 0}, p2_live[2] = {0, 0};  // 1 + the largest key with a gate: M is zero from there on
    uint32_t p1_cov[2] = {0, 0}, p2_cov[2] = {0, 0};   // keys [0, cov) all have a gate: the scatter writes them, only the rest is zeroed
    int p2_uniform[2] = {-1, -1};   // every gate of the list has its u operand in the same layer: 0 = layer 0, 1 = previous, -1 = mixed
    gate_rec *uni2 = nullptr;      // uni gates for the phase-2 constant term (aux = u, bit 10 = u in previous layer)
    uint64_t n_uni2 = 0;
    uint32_t *ori_u = nullptr, *ori_v = nullptr;
    // DOT_PROD: gates sorted by u (aux = v) with CSR row pointers
    gate_rec *d1 = nullptr;
    uint32_t *d1_rowptr = nullptr;
    uint32_t d1_rows = 0;
    uint32_t d1_live_rows = 0;     // 1 + the last row that has a gate: the phase-1 table is zero behind it (never built, never read: k_round_cubic's x_live)
    // ... whose gates are exactly {(g, u, v) = (p CO + co, p CI + ci, (pp + co) CI + ci)} with CO a power of two and pp >= 2 pictures: the phase-1
    // table is then beta_hi[p] * S[ci, t], S summed ONCE over co instead of once per picture (sumcheck.hip: zk_sumcheck_dotprod_init_phase1)
    bool dot_ok = false;
    uint32_t dot_pp = 0, dot_CO = 0, dot_CI = 0;
    // resident witness program (zk_witness_program_upload): gate lists grouped by output with layer-0 operands as raw layer-0 indices;
    // DOT_PROD: CSR by output vector
    void *ev_uni = nullptr, *ev_bin = nullptr;
    uint64_t n_ev_uni = 0, n_ev_bin = 0;
    gate_rec *ev_dot = nullptr;
    uint32_t *ev_dot_ptr = nullptr;
    bool ev_conv = false;          // the bin gates of this (structured) convolution are evaluated by k_conv_eval, not from a list
    long long *ev_w64 = nullptr;   // ... and its weights as 64-bit integers (k_conv_eval_i64), inside zk_ctx::wp_w64
    // direct convolution whose gate list matches the structured pattern (zk_upload_circuit_hinted): its two big gate sums are factored
    bool conv_ok = false;
    conv_desc conv;
};

// one (V, M) bookkeeping-table pair of a sumcheck, ping-pong buffered
struct table_pair {
    fr_t *V[2] = {nullptr, nullptr};
    fr_t *M[2] = {nullptr, nullptr};
    int cur = 0;
    const fr_t *Vsrc = nullptr;    // until the first fold the V input is read straight from a layer's values (no copy into V[cur])
    uint64_t len = 0;              // current (pre-fold) length; 0 = absent or already absorbed
    uint64_t live = 0;             // V and M are zero from this entry on (<= len); the large-table round kernel skips those pairs
    bool absorbed = false;         // collapsed to a constant whose product went into add_term
    bool tail_valid = false;       // the two entries left in V are known on the host (sent along with the last round)
    HFr tail_v[2];
    HFr final_v;                   // value of V when it collapsed
    HFr abs_m;                     // ... and of its multiplier, times (1 - r) of every round since (zero-knowledge mode: zk_sumcheck_tail_pairs)
    bool abs_m_valid = false;
};

// buffer sizes (in field elements) a circuit asks of each of its sessions; computed once per circuit
struct circuit_sizes {
    uint64_t tp_cap[2] = {1, 1};   // longest bookkeeping table of each pair
    uint64_t v0_cap[2] = {1, 1};   // longest V table that is built (not read in place)
    uint64_t bg = 1, bu = 1, gs = 1, max_list = 1;
    uint64_t sub = 1;              // longest layer-0 subset table (the verifier's layer-0 check builds eq tables over them)
    uint64_t conv_wa = 0, conv_part = 0, conv_ae = 0;
};

struct msm_state;                  // hyrax.hip
struct zk_batch;                   // below: a lock-step batch of contexts on one stream

// kernel classes of the built-in profiler (HIP events on the context's stream)
enum prof_class { PC_EQ = 0, PC_GATHER, PC_GATE, PC_GATE_FIX, PC_GATE_SUM, PC_SUM, PC_ROUND_QUAD, PC_ROUND_CUBIC, PC_FOLD, PC_MATVEC,
                  PC_PHI, PC_DOT, PC_LIU, PC_MSM_PLANES, PC_MSM_FINISH, PC_MSM_TABLES, PC_IPA, PC_MISC, PC_TAIL, PC_ROUND_FINE, PC_COUNT };
static const char *const prof_names[PC_COUNT] = {"eq_table", "gather", "gate_reduce", "gate_fixup", "gate_sum", "sum_partials", "round_quad",
                                                 "round_cubic", "fold", "matvec", "phi", "dot_prod", "liu_scatter", "msm_planes",
                                                 "msm_finish", "msm_tables", "ipa", "misc", "round_tail", "round_fine"};
struct prof_pending { hipEvent_t e0, e1; int cls; double bytes; };

struct zk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // lock-step batch (zk_batch_attach): the context is lane `lane` of `batch` and launches on the batch's stream; own_stream is parked meanwhile
    zk_batch *batch = nullptr;
    int lane = -1;
    uint32_t n_pending = 0;        // launches this lane has deferred since the batch's last flush
    hipStream_t own_stream = nullptr;

    std::vector<dev_layer> L;
    fr_t *two_mul = nullptr;
    int n_two_mul = 0;
    std::vector<void *> owned;     // every device allocation of this context, freed in zk_ctx_destroy
    uint64_t sink_bytes = 0;
    std::vector<void *> *alloc_sink = nullptr;   // while the static part of a circuit is built: its registry entry's list instead
    void *circuit = nullptr;       // shared_circuit (sumcheck.hip): gate lists, subset maps, layer-0 CSR -- shared by the sessions of one GPU
    circuit_sizes sz;

    // work buffers
    table_pair tp[2];
    fr_t *beta_g[2] = {nullptr, nullptr};   // ping-pong (the PADDING layer expands the FFT layer's table)
    int beta_g_cur = 0;
    uint64_t beta_g_cap = 0;
    fr_t *beta_u = nullptr; uint64_t beta_u_cap = 0;
    fr_t *beta_gs = nullptr; uint64_t beta_gs_cap = 0;
    fr_t *small[2] = {nullptr, nullptr};    // periodic table of the cubic rounds (ping-pong)
    uint32_t small_len = 0; int small_cur = 0;
    fr_t *partials = nullptr; uint32_t partial_blocks = 0;
    fr_t *d_result = nullptr;      // 32 elements
    HFr *h_result = nullptr;       // pinned, 32 elements
    // per-round hand-over: mapped pinned host slot written by the last block of a fused round kernel
    struct host_slot_h { HFr v[12]; volatile unsigned long long seq; } *h_slot = nullptr;
    void *d_slot = nullptr;        // device address of h_slot
    // second slot: the two sums behind a phase's add_term (k_gate_multi's constant-term job); read lazily, behind the phase's first round
    host_slot_h *h_aux = nullptr; void *d_aux = nullptr;
    unsigned long long aux_seq = 0;
    bool add_pending = false;
    uint32_t *d_counter = nullptr; // arrival counter of the grid-wide reduction
    unsigned long long slot_seq = 0;
    uint32_t *carry_key = nullptr; fr_t *carry_val = nullptr; uint64_t carry_slots = 0;
    dev_buf scratch;               // growable (mat-vec partial sums, ...)
    std::vector<fr_t *> root_pw[2];  // [inverse][n] powers of the 2^n-th root of unity

    // prover state machine (names follow reference src/prover.hpp:55-74)
    std::vector<std::vector<HFr>> r_u, r_v;
    const HFr *r_0 = nullptr, *r_1 = nullptr;
    bool lane_tail = false;        // this context's resident tail runs inside a batch's fused launch (counted in zk_batch::tails_running)
    HFr alpha, beta, relu_rou, add_term, V_u0, V_u1;
    HFr vu1_scale;                 // V_u1 = claim_1 x this (a DOT_PROD phase 1: the periodic table's value; 1 otherwise): zk_sumcheck_claims_adjust
    HFr small_final;               // collapsed periodic table (DOT_PROD): the scalar m of the rounds behind the transform's variables
    bool small_final_valid = false;
    // DOT_PROD phase 1 behind the collapse: sum X Y m is a QUADRATIC sumcheck of the pair (V = Y, M = X) times the scalar m -- its rounds go through
    // quad_round (resident kernels, latency kernel) with pair 1's M pointers borrowed from pair 0's V buffers until the phase is finalized
    bool dot_quad = false;
    fr_t *dot_saved_M[2] = {nullptr, nullptr};
    // the beta_g table an IFFT layer leaves for the DOT_PROD layer below it is alpha * eq(bg_r, .): kept so that the DOT_PROD layer can split it
    std::vector<HFr> bg_r;
    HFr bg_alpha;
    int bg_layer = -1;             // the IFFT layer that built it
    fr_t *dot_tabs = nullptr;      // beta_lo (4096 entries), beta_hi (4096 entries)
    // Y's dead region of a DOT_PROD phase, folded several rounds at once instead of every round (sumcheck.hip: zk_sumcheck_dotprod_update1)
    bool dot_defer = false;
    int dot_fill_log = 20;         // policy::DOT_FILL_LOG (or a test's override)
    const fr_t *dot_defer_Y = nullptr;     // the unfolded table (the previous layer's values, read in place)
    uint64_t dot_defer_to = 0;             // entries of it that can be non-zero
    fr_t *dead_tabs = nullptr;     // eq table of the catch-up (2^DEAD_MAX_FOLDS entries) + partial sums
    uint64_t dot_defer_count = 0;  // phases that took this path
    fr_t *dot_part = nullptr;      // channel_out chunks of S
    uint64_t dot_part_cap = 0;
    uint32_t dot_layers = 0;       // DOT_PROD layers with the factored table (zk_factored_dot_layers)
    uint64_t proof_size = 0;
    int sumcheck_id = 0, round = 0;
    HFr last_poly[3];              // the quadratic the previous round returned (a, b, c): p(0) + p(1) of this round must equal it at the challenge
    bool last_poly_valid = false;
    bool circuit_ready = false;

    // layer-0 combine (zk_sumcheck_liu_init): CSR of (table, index) pairs by layer-0 index, half tables, per-call table descriptors
    uint32_t *liu_ptr = nullptr; void *liu_ent = nullptr; fr_t *liu_halves = nullptr;
    void *liu_tabs = nullptr, *h_liu_tabs = nullptr;      // device / pinned host array of liu_table
    uint32_t liu_ntabs = 0;
    std::vector<int> liu_tab_layer, liu_tab_side;           // table id -> (layer, 0 = u / 1 = v)

    // structured convolution layers (conv_kernels.cuh): small eq tables, their descriptors (device + pinned host, one set per phase), work buffers
    fr_t *conv_small = nullptr;
    void *conv_tabs = nullptr, *h_conv_tabs = nullptr;
    fr_t *conv_wa = nullptr, *conv_part = nullptr, *conv_e = nullptr, *conv_ae = nullptr;
    int conv_K = 0;
    uint32_t conv_layers = 0;

    msm_state *msm = nullptr;
    msm_state *vmsm = nullptr;     // verifier-owned second table set (zk_verifier_msm over arbitrary points)

    // device-side Fiat-Shamir rounds (fs_tail.cuh): the host's chain state (8 words) and its count of not yet hashed message bytes, attached by the
    // non-interactive driver; the record of the phase the tail kernel has run ahead
    const uint32_t *fs_state = nullptr;
    const uint64_t *fs_pending = nullptr;
    void *h_tail = nullptr, *d_tail = nullptr;     // tail_out, pinned + mapped
    bool tail_active = false;
    // persistent rounds of the interactive protocol (k_tail<true>): challenge mailbox (pinned + mapped), the running kernel's round window
    void *h_live_in = nullptr, *d_live_in = nullptr;
    bool live_rounds = true;       // zk_set_live_rounds
    bool live_active = false;
    bool live_mid = false;         // the running kernel is k_mid (a segment of mid-size rounds): the host tracks tables and add_term itself
    bool live_with_add = false, live_first = false;
    void *d_bcast = nullptr;       // mid_bcast: k_mid's challenge line in device memory
    bool live_now = true;          // decided per proof (zk_proof_begin): does this proof have a hardware queue to itself?
    // what it takes to run the current phase AGAIN with a launch per round, should its resident kernel be lost (a time-out, a stall behind somebody else's
    // kernels): which phase it is, the arguments of its initialisation, the challenges it has answered so far
    int phase_kind = 0;            // 0 none / not replayable, 1 phase 1 (generic), 2 phase 2, 3 layer-0 combine
    int phase_bg_cur = 0;          // beta_g_cur when the phase began (a PADDING layer flips it)
    std::vector<HFr> phase_r;      // challenge of every round call of the phase but the first
    std::vector<uint64_t> phase_sig_u, phase_sig_v;    // layer-0 combine: its weights
    bool phase_no_live = false;    // this phase runs on launches (it is being replayed, or has been)
    bool live_lost = false;        // the resident kernel of the round in progress is gone: replay the phase
    uint64_t live_fallbacks = 0;   // phases that were finished that way
    bool counted_active = false;
    bool counted_context = false;  // counted among the live contexts of its device
    int first_gate_held = 0;        // 1 + index of the first-proof gate this context holds (context.hip: first_gate_enter): others of its kind wait for it
    int live_count = 0, live_cursor = 0;
    uint32_t live_seq32 = 16;
    uint64_t live_rounds_total = 0, live_phases_total = 0;
    double live_t_gpu = 0, live_t_host = 0, live_t_exit = 0;          // ZKCNN_TIMING: waiting for the kernel / between two round calls
    uint64_t live_ticks_wait = 0, live_ticks_total = 0, live_timed_rounds = 0;
    int tail_count = 0, tail_cursor = 0, phase_rounds = 0;
    unsigned long long tail_seq = 0;
    uint64_t tail_rounds_total = 0, tail_phases_total = 0;
    // hybrid tail (zk_set_host_tail): once the live tables of a phase have at most 2^host_tail_log entries they travel to the host (a few KB)
    // and the remaining rounds -- a few hundred multiplications each -- run there; -1 = the default (off; on for a lane of a batch: sumcheck.hip,
    // policy::LANE_TAIL_LOG), -2 = off (every round is a kernel)
    int host_tail_log = -1;
    bool host_tail_active = false;
    std::vector<HFr> ht_V[2], ht_M[2];
    uint64_t host_tail_rounds_total = 0;
    // ZKCNN_TIMING: host view of a round call: time between calls (verifier + wrappers), before / in / after the wait for the result
    double t_seg[4] = {0, 0, 0, 0}; uint64_t n_seg = 0; double t_last_exit = 0;

    // verifier-side tables (zk_verifier_*): kept apart from the prover's, whose beta_g carries state from layer to layer
    fr_t *v_bg = nullptr, *v_bu = nullptr, *v_bv = nullptr, *v_gs = nullptr;

    // witness generation (zk_witness_input / zk_witness_gates): device copy of layer 0 while it is being built + staging
    dev_buf w_val0, w_stage[5];
    uint64_t w_val0_len = 0;

    // resident witness program
    bool wp_ready = false;
    std::vector<zk_witness_step> wp_steps;
    void *wp_ops = nullptr;
    uint32_t *wp_windows = nullptr;
    uint64_t wp_n_ops = 0, wp_n_windows = 0;
    fr_t *wp_tmp = nullptr;                 // 2 x (largest generic layer) partial sums
    uint32_t *wp_carry_key = nullptr; fr_t *wp_carry_val = nullptr;
    unsigned long long *wp_ranges = nullptr;    // device: 2 per range step, then one word of flags
    unsigned long long *h_wp_ranges = nullptr;  // pinned copy
    uint32_t wp_n_ranges = 0;
    fr_t *wp_conv_part = nullptr;               // channel chunks of k_conv_eval
    long long *wp_in64 = nullptr, *wp_w64 = nullptr;   // integer copies of a convolution's input tensor (per layer, per picture) and of all weights
    unsigned long long *wp_cb_in = nullptr, *wp_cb_w = nullptr;    // per layer: {largest magnitude, wide flag} of the two (witness_kernels.cuh)
    bool wp_w64_valid = false;                  // false after layer 0 was written from outside: the weights are converted again
    int wp_force_field = 0;                     // test hook: convolutions in field arithmetic whatever the bounds say
    void *wp_segments = nullptr;                // one wit_segment per layer (k_last_nonzero)

    // profiler: when a class bit is set in prof_mask every launch of that class is bracketed by events
    uint32_t prof_mask = 0;
    std::vector<prof_pending> prof_q;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[PC_COUNT] = {0};
    double prof_bytes[PC_COUNT] = {0};
    uint64_t prof_cnt[PC_COUNT] = {0};
};

// ---- one resident circuit per GPU, shared by its sessions (upload.hip; reference src/prover.hpp:47-48: one layeredCircuit per prover) ----
// Everything a circuit's upload produces that does not depend on the witness -- sorted and padded gate lists, subset maps, the layer-0 CSR,
// the checked convolution patterns, buffer sizes -- lives in a ref-counted registry entry keyed by a digest of the upload's input. The
// first context that uploads a circuit builds the entry (counting sort of 1.2e8 gates + 1.9 GB of lists for vgg11); every later context of
// the same process and device that uploads the SAME circuit attaches to it and only allocates its own values, tables and scratch.
struct shared_circuit {
    int device = 0;
    uint64_t key[2] = {0, 0};
    int refs = 0;
    bool ready = false, failed = false;
    uint64_t bytes = 0;             // device memory of the static part
    std::mutex mtx;                 // held while the entry is being built
    std::vector<void *> owned;      // device allocations of the static part
    std::vector<dev_layer> L;       // (val == nullptr)
    fr_t *two_mul = nullptr;
    int n_two_mul = 0;
    uint32_t *liu_ptr = nullptr; void *liu_ent = nullptr; uint32_t liu_ntabs = 0;
    std::vector<int> liu_tab_layer, liu_tab_side;
    uint32_t conv_layers = 0;
    circuit_sizes sz;
    // the static part of the resident witness program (witness.hip: zk_witness_program_upload): gate lists grouped by output live in L[i].ev_*,
    // the operations, windows and steps here; built by the first context that uploads the program, adopted by every other (and by clones)
    bool wp_ready = false;
    void *wp_ops = nullptr; uint32_t *wp_windows = nullptr;
    uint64_t wp_n_ops = 0, wp_n_windows = 0;
    std::vector<zk_witness_step> wp_steps;
    uint32_t wp_n_ranges = 0;
    uint64_t wp_max_out = 1, wp_max_blocks = 1, wp_max_conv_part = 0, wp_max_conv_in = 0, wp_conv_w_total = 0;
    std::vector<uint64_t> wp_conv_w_off;
};


static inline void prof_begin(zk_ctx *ctx, int cls, double bytes) {
    if (!((ctx->prof_mask >> cls) & 1u)) return;
    prof_pending p;
    for (hipEvent_t *e : {&p.e0, &p.e1}) {
        if (ctx->prof_pool.empty()) { if (hipEventCreate(e) != hipSuccess) return; }
        else { *e = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); }
    }
    p.cls = cls;
    p.bytes = bytes;
    (void) hipEventRecord(p.e0, ctx->stream);
    ctx->prof_q.push_back(p);
}
static inline void prof_end(zk_ctx *ctx, int cls) {
    if (!((ctx->prof_mask >> cls) & 1u) || ctx->prof_q.empty()) return;
    (void) hipEventRecord(ctx->prof_q.back().e1, ctx->stream);
}
#define ZK_LAUNCH(cls, bytes, kern, grid, block, ...)                                    \
    do {                                                                                  \
        ZK_ORDER();                                                                       \
        prof_begin(ctx, cls, bytes);                                                      \
        hipLaunchKernelGGL(kern, grid, block, 0, ctx->stream, __VA_ARGS__);               \
        prof_end(ctx, cls);                                                               \
    } while (0)

// ---- lock-step batches (include/zkcnn_hip.h: zk_batch_*) ---------------------------------------------------------------------------
// K contexts that prove K pictures on ONE resident circuit share one stream and walk the protocol in lock step. A launch whose kernel is
// written as a functor (launch.cuh: zk_launch_f) is not issued by a lane: its argument block is DEFERRED into the batch, in the lane's own
// order (`gen` = how many launches the lane has deferred since the last flush). When every lane is parked -- each has reached a point where
// it needs a result from the GPU and has handed the thread to the batch's driver -- zk_batch_flush issues, generation by generation, ONE
// launch per kernel for all lanes that deferred it (blockIdx.z = lane; the argument blocks travel by value in the kernel argument segment,
// or through a pinned ring when eight of them exceed its 4 KB). Anything else a lane puts on the stream (a kernel without a functor form, a
// copy, a fill) first flushes what the lane has deferred, so each lane's own order is kept; lanes are independent of one another, so any
// interleaving between lanes is valid.
#define ZK_BATCH_ARG_BYTES 2560
struct batch_item;
typedef int32_t (*batch_launch_fn)(zk_batch *b, const batch_item *const *items, uint32_t n, uint32_t gx, uint32_t gy);
struct batch_item {
    batch_launch_fn launch;        // launch.cuh: launch_lanes<F, BLOCK> -- also the key fused launches are grouped by
    zk_ctx *ctx;
    uint32_t gen;                  // position in the lane's own sequence of deferred launches
    uint32_t gx, gy;               // the grid this lane's launch would have had
    bool exact;                    // the body relies on gridDim being its own launch's: fused only with lanes that asked for the same grid
    const char *name;              // the functor's type, as the compiler spells it (fusion report)
    int prof_class;
    double bytes;                  // its algorithmic bytes (profiler)
    alignas(16) unsigned char arg[ZK_BATCH_ARG_BYTES];
};
#define ZK_BATCH_RING_SLOTS 64
struct zk_batch {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<zk_ctx *> lanes;
    zk_yield_fn yield_fn = nullptr;
    void *yield_user = nullptr;
    std::vector<batch_item> pending;
    std::string err;
    // argument blocks too large for the kernel argument segment: a ring of slots in pinned host memory the kernels read in place (written
    // before the launch, read-only during it; an event per quarter of the ring guards reuse)
    unsigned char *h_ring = nullptr, *d_ring = nullptr;
    uint32_t ring_next = 0;
    uint64_t ring_count = 0;
    hipEvent_t ring_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ring_ev_set[4] = {false, false, false, false};
    // statistics: launches issued by flushes, lane launches they stood for, flushes, flushes that found nothing to do
    uint64_t n_launches = 0, n_lane_launches = 0, n_flushes = 0, n_empty_flushes = 0;
    std::map<const char *, std::pair<uint64_t, uint64_t>> by_kernel;       // fusion report (ZKCNN_BATCH_TRACE=1: printed when the batch is destroyed)
    // lanes whose resident tail (k_tail_live_f, one fused launch) is running: a lane that has left its tail waits for the others before it defers
    // anything else, so that the next launch is a fused one again (rounds_resident.hpp: live_round)
    int tails_running = 0;
    uint64_t tails_open_at = 0;            // value of n_flushes from which the lanes that have left their tails go on (set by the last one out)
    uint64_t n_lane_tails = 0, n_lane_tail_rounds = 0;
};
// Every point where a lane needs its deferred launches on the stream (it is about to wait for a result, or to put something else on the
// stream): hand the thread to the batch's driver, which runs the other lanes up to their own such points and flushes; without a driver (no
// yield function) the lane flushes for itself.
int32_t zk_batch_sync_point(zk_ctx *ctx);
// hipStreamSynchronize for a wait that is part of a proof: a lane first lets the other lanes issue their work up to the same point, so that
// the GPU runs all of it while the thread waits once (never call it with a lock held: the other lanes run on this thread)
static inline hipError_t zk_stream_sync(zk_ctx *ctx) {
    if (ctx->batch && zk_batch_sync_point(ctx) != ZK_OK) return hipErrorUnknown;
    return hipStreamSynchronize(ctx->stream);
}
// a copy / fill on the context's stream
#define ZK_STREAM(call) do { ZK_ORDER(); ZK_HIP(call); } while (0)
// before anything that is not deferred goes onto the stream of a context: keep the lane's order
#define ZK_ORDER() do { if (ctx->batch && ctx->n_pending) { int32_t rc_o_ = zk_batch_sync_point(ctx); if (rc_o_) return rc_o_; } } while (0)

// (for a `void` context -- the timing lambdas of the micro-benchmarks, which never run on a lane: no ordering check, no status)
#define ZK_LAUNCH_RAW(cls, bytes, kern, grid, block, ...)                                \
    do {                                                                                  \
        prof_begin(ctx, cls, bytes);                                                      \
        hipLaunchKernelGGL(kern, grid, block, 0, ctx->stream, __VA_ARGS__);               \
        prof_end(ctx, cls);                                                               \
    } while (0)

#define ZK_HIP(call)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return ZK_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)

static inline fr_t to_dev(const HFr &x) {
    fr_t z;
    std::memcpy(&z, &x, 32);
    return z;
}
static inline uint32_t grid_for(uint64_t work, uint32_t cap = 2048) {
    uint64_t b = (work + ZK_BLOCK - 1) / ZK_BLOCK;
    if (b < 1) b = 1;
    return (uint32_t) (b > cap ? cap : b);
}

// ---- helpers shared by the translation units of libzkcnn_hip.so: context.hip (context, profiler, switches), upload.hip (residency: gate
// sort, pattern checks, the registry of resident circuits), sumcheck.hip (the prover's state machine and its kernels), witness.hip (witness
// generation on the GPU), verifier.hip (the verifier's wiring predicates), hyrax.hip (commitment) ----
#define ZK_CHECK_READY_ROUND() do { if (!ctx || !ctx->circuit_ready) return ZK_ERR_STATE; ZK_HIP(hipSetDevice(ctx->device)); } while (0)
// every entry point but the four round calls first sends a resident round kernel home (its phase was abandoned)
#define ZK_CHECK_READY() do { ZK_CHECK_READY_ROUND(); if (ctx->live_active) { int32_t rc_ = zk_live_abort(ctx); if (rc_) return rc_; } ZK_ORDER(); } while (0)
#define ZK_CHECK_CTX() do { if (!ctx) return ZK_ERR_ARG; ZK_HIP(hipSetDevice(ctx->device)); if (ctx->live_active) { int32_t rc_ = zk_live_abort(ctx); if (rc_) return rc_; } ZK_ORDER(); } while (0)
static inline const HFr &H(const uint64_t *p) { return *reinterpret_cast<const HFr *>(p); }
static inline void put(uint64_t *dst, const HFr &x) { std::memcpy(dst, &x, 32); }

int32_t zk_dev_alloc(zk_ctx *ctx, void **p, size_t bytes);          // tracked allocation
int32_t zk_scratch(zk_ctx *ctx, size_t bytes);                       // grow ctx->scratch
void zk_msm_destroy(zk_ctx *ctx);
// device-side helpers implemented in sumcheck.hip and re-used by hyrax.hip
int32_t zk_eq_table1_dev(zk_ctx *ctx, fr_t *out, int n, const HFr *r, const HFr &init);
int32_t zk_col_combine_dev(zk_ctx *ctx, fr_t *out, const fr_t *Z, const fr_t *L, uint32_t cols, uint32_t rows);
// a resident round kernel whose phase nobody finishes (a verifier that rejected mid-phase) must leave before anything else uses the stream
int32_t zk_live_abort(zk_ctx *ctx);
int zk_contexts_on_device(int device);      // context.hip: live contexts of this process on a device
template <class T>
static inline int32_t zk_upload(zk_ctx *ctx, T **dst, const std::vector<T> &src) {
    *dst = nullptr;
    if (src.empty()) return ZK_OK;
    int32_t rc = zk_dev_alloc(ctx, (void **) dst, src.size() * sizeof(T));
    if (rc) return rc;
    ZK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return ZK_OK;
}
void zk_circuit_release(zk_ctx *ctx);                                                 // upload.hip
int32_t zk_witness_program_adopt(zk_ctx *ctx);                                        // witness.hip: the per-session side of the circuit's witness program
void zk_set_create_error(const std::string &msg);                                     // context.hip
// sumcheck.hip
int32_t zk_eq_table_dev(zk_ctx *ctx, fr_t *out, int n, const HFr *r0, const HFr &a, const HFr *r1, const HFr &b, uint64_t tail_start, const HFr &tail_scale, uint64_t limit);
int32_t zk_phi_table_dev(zk_ctx *ctx, fr_t *out, const HFr *rx, const HFr &scale, int n, bool inverse);
fr_t *zk_powers_of_root(zk_ctx *ctx, int n, bool inverse);
int32_t zk_wait_slot(zk_ctx *ctx, unsigned long long seq);

// libzkcnn_hip.so, part 1: context, residency and the GKR sumcheck state machine on the GPU.
// Each extern "C" function is the binding of one reference prover method (see include/zkcnn_hip.h);
// the O(1)-per-round scalar bookkeeping (add_term, round counters, proof-size counter) stays in
// host code here exactly as in the reference, everything that is O(table) or O(gates) is a kernel.
#include <chrono>
#include <sched.h>
#include <time.h>
#include <algorithm>
#include <cmath>
#include <atomic>
#include <mutex>
#include <cstring>
#include <emmintrin.h>
#include "ctx.hpp"
#include "kernels.cuh"
#include "fs_tail.cuh"
#include "witness_kernels.cuh"
#include "conv_kernels.cuh"
#include "prep_kernels.cuh"

static std::string g_create_err;
static void circuit_release(zk_ctx *ctx);

// ------------------------------------------------------------------------------------------------
// memory helpers
// ------------------------------------------------------------------------------------------------
int32_t zk_dev_alloc(zk_ctx *ctx, void **p, size_t bytes) {
    if (bytes == 0) bytes = 32;
    ZK_HIP(hipMalloc(p, bytes));
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
template <class T>
static int32_t upload(zk_ctx *ctx, T **dst, const std::vector<T> &src) {
    *dst = nullptr;
    if (src.empty()) return ZK_OK;
    int32_t rc = zk_dev_alloc(ctx, (void **) dst, src.size() * sizeof(T));
    if (rc) return rc;
    ZK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
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
    *out = ctx;
    return ZK_OK;
}

extern "C" void zk_ctx_destroy(zk_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->live_active) (void) zk_live_abort(ctx);
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
    circuit_release(ctx);
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
    for (prof_pending &p : ctx->prof_q) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            ctx->prof_ms[p.cls] += ms;
            ctx->prof_bytes[p.cls] += p.bytes;
            ++ctx->prof_cnt[p.cls];
        }
        ctx->prof_pool.push_back(p.e0);
        ctx->prof_pool.push_back(p.e1);
    }
    ctx->prof_q.clear();
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
    ctx->fs_state = state;
    ctx->fs_pending = pending;
    ctx->tail_active = false;
    ctx->host_tail_active = false;
    ctx->last_poly_valid = false;
    return ZK_OK;
}
extern "C" int32_t zk_set_host_tail(zk_ctx *ctx, int32_t log_entries) {
    if (!ctx || log_entries > 8) return ZK_ERR_ARG;
    ctx->host_tail_log = log_entries < 0 ? -1 : log_entries;
    return ZK_OK;
}
static std::atomic<int> g_active_proofs[64];
extern "C" int32_t zk_proof_begin(zk_ctx *ctx) {
    if (!ctx) return ZK_ERR_ARG;
    if (!ctx->counted_active) { ++g_active_proofs[ctx->device & 63]; ctx->counted_active = true; }
    // resident kernels only for a proof that is alone on its GPU when it starts: their workgroups wait for one another (k_mid) and for the host,
    // which is only safe -- and only profitable -- while nothing else competes for the CUs and the hardware queue for long
    ctx->live_now = g_active_proofs[ctx->device & 63].load() <= 1;
    return ZK_OK;
}
extern "C" int32_t zk_proof_end(zk_ctx *ctx) {
    if (!ctx) return ZK_ERR_ARG;
    if (ctx->counted_active) { --g_active_proofs[ctx->device & 63]; ctx->counted_active = false; }
    ctx->live_now = true;
    return ZK_OK;
}
extern "C" int32_t zk_set_live_rounds(zk_ctx *ctx, int32_t on) {
    if (!ctx) return ZK_ERR_ARG;
    if (ctx->live_active) { int32_t rc = zk_live_abort(ctx); if (rc) return rc; }
    ctx->live_rounds = on != 0;
    return ZK_OK;
}
extern "C" int32_t zk_fs_stats(zk_ctx *ctx, uint64_t *rounds, uint64_t *phases) {
    if (!ctx || !rounds || !phases) return ZK_ERR_ARG;
    *rounds = ctx->tail_rounds_total + ctx->live_rounds_total;
    *phases = ctx->tail_phases_total + ctx->live_phases_total;
    return ZK_OK;
}

extern "C" const char *zk_last_error(const zk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
extern "C" uint64_t zk_proof_bytes(const zk_ctx *ctx) { return ctx->proof_size; }

// ------------------------------------------------------------------------------------------------
// residency
// ------------------------------------------------------------------------------------------------
// number of leading keys 0, 1, 2, ... that occur in a list sorted by key
static uint32_t covered_prefix(const std::vector<gate_rec> &recs) {
    uint32_t k = 0;
    for (const gate_rec &r : recs) {
        if (r.key == k) ++k;
        else if (r.key > k) break;
    }
    return k;
}
// records after padding every run of equal keys (list sorted by key) to a multiple of G
static uint64_t padded_size(const std::vector<gate_rec> &recs, uint32_t G) {
    uint64_t total = 0;
    size_t i = 0;
    while (i < recs.size()) {
        size_t j = i;
        while (j < recs.size() && recs[j].key == recs[i].key) ++j;
        total += ((j - i + G - 1) / G) * G;
        i = j;
    }
    return total;
}
// records per thread for a list: the largest of 32 / 16 / 8 / 4 whose padding costs less than 1/8 of the list
static uint32_t choose_group(const std::vector<gate_rec> &recs) {
    for (uint32_t G = 32; G > GATE_GROUP; G >>= 1)
        if (padded_size(recs, G) <= recs.size() + recs.size() / 8) return G;
    return GATE_GROUP;
}
// pads every run of equal keys (list sorted by key) to a multiple of G records with padding records of the same key
static void pad_runs(std::vector<gate_rec> &recs, uint32_t G) {
    std::vector<gate_rec> out;
    out.reserve(recs.size() + recs.size() / 8 + 64 * G);
    size_t i = 0;
    while (i < recs.size()) {
        size_t j = i;
        while (j < recs.size() && recs[j].key == recs[i].key) ++j;
        out.insert(out.end(), recs.begin() + i, recs.begin() + j);
        for (size_t k = j - i; k % G; ++k) {
            gate_rec d = {0, recs[i].key, 0, 1u << 11};
            out.push_back(d);
        }
        i = j;
    }
    // whole waves: the list is padded to a multiple of 64 groups with records of the last key, and inside every chunk of
    // 64 x G records the k-th record of lane l is stored at slot k * 64 + l, so that each of the kernel's G loads is one
    // contiguous 1 KB access of the wave
    const size_t chunk = 64 * (size_t) G;
    while (!out.empty() && out.size() % chunk) {
        gate_rec d = {0, out.back().key, 0, 1u << 11};
        out.push_back(d);
    }
    recs.resize(out.size());
    for (size_t base = 0; base < out.size(); base += chunk)
        for (size_t l = 0; l < 64; ++l)
            for (size_t k = 0; k < G; ++k) recs[base + k * 64 + l] = out[base + l * G + k];
}
static void counting_sort(std::vector<gate_rec> &recs, uint32_t nkeys) {
    std::vector<uint32_t> cnt((size_t) nkeys + 1, 0);
    for (const gate_rec &r : recs) ++cnt[r.key + 1];
    for (uint32_t k = 0; k < nkeys; ++k) cnt[k + 1] += cnt[k];
    std::vector<gate_rec> out(recs.size());
    for (const gate_rec &r : recs) out[cnt[r.key]++] = r;
    recs.swap(out);
}

static uint32_t ceil_log2(uint32_t x) {
    uint32_t b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}
static int log2_exact(uint32_t x) {          // -1 unless x is a power of two
    if (!x || (x & (x - 1))) return -1;
    int b = 0;
    while ((1u << b) < x) ++b;
    return b;
}
// does layer S (with its predecessor P) consist of exactly the gates a direct convolution with these parameters emits?
static bool conv_hint_matches(const zk_conv_hint &h, const zk_layer_desc &S, const zk_layer_desc &P, conv_desc &c) {
    if (S.ty != ZK_NCONV) return false;
    c.pp = h.pic_parallel; c.CO = h.channel_out; c.CI = h.channel_in; c.nxi = h.nx_in; c.nyi = h.ny_in; c.nxo = h.nx_out; c.nyo = h.ny_out;
    c.m = h.m; c.pad = h.padding; c.ls = h.log_stride; c.wstart = h.weight_start;
    c.bx_i = log2_exact(c.nxi); c.by_i = log2_exact(c.nyi); c.bc_i = log2_exact(c.CI);
    c.bx_o = log2_exact(c.nxo); c.by_o = log2_exact(c.nyo); c.bc_o = log2_exact(c.CO);
    if (c.bx_i < 0 || c.by_i < 0 || c.bc_i < 0 || c.bx_o < 0 || c.by_o < 0 || c.bc_o < 0 || !c.pp || !c.m || c.m > 16 || c.ls > 4 || c.pad > 16) return false;
    if (c.bx_i + c.by_i > 12 || c.bx_o + c.by_o > 12 || c.bc_i > 12 || c.bc_o > 12 || c.pp > 4096) return false;
    const uint64_t n_out = (uint64_t) c.pp * c.CO * c.nxo * c.nyo, n_in = (uint64_t) c.pp * c.CI * c.nxi * c.nyi;
    // (the previous layer may be longer than the tensor the convolution reads: a RELU / pooling layer keeps its constraint rows behind its outputs)
    if (n_out != S.size || n_in > P.size || S.size_u[1] != P.size || S.size_v[0] != (uint64_t) c.CO * c.CI * c.m * c.m) return false;
    if (S.bit_length != c.bx_o + c.by_o + c.bc_o + (int) ceil_log2(c.pp) || S.bit_length_u[1] != P.bit_length) return false;
    if (S.max_bl_u - (c.bx_i + c.by_i + c.bc_i) > 12 || S.bit_length - (c.bx_o + c.by_o + c.bc_o) > 12) return false;
    if (((c.nxi + 2 * c.pad - c.m) >> c.ls) + 1 != c.nxo || ((c.nyi + 2 * c.pad - c.m) >> c.ls) + 1 != c.nyo) return false;
    // every bin gate, in emission order (p, co, ci, window origin, offset inside the window); operands: u in the previous layer, v in layer 0
    const int64_t lo = -(int64_t) c.pad, Rx = (int64_t) c.nxi + c.pad, Ry = (int64_t) c.nyi + c.pad, st = 1ll << c.ls;
    uint64_t k = 0;
    for (uint32_t p = 0; p < c.pp; ++p)
        for (uint32_t co = 0; co < c.CO; ++co)
            for (uint32_t ci = 0; ci < c.CI; ++ci)
                for (int64_t x = lo; x + c.m <= Rx; x += st)
                    for (int64_t y = lo; y + c.m <= Ry; y += st) {
                        const uint64_t g = (((uint64_t) p * c.CO + co) * c.nxo + ((x - lo) >> c.ls)) * c.nyo + ((y - lo) >> c.ls);
                        for (int64_t tx = x; tx < x + c.m; ++tx)
                            for (int64_t ty = y; ty < y + c.m; ++ty) {
                                if (tx < 0 || tx >= c.nxi || ty < 0 || ty >= c.nyi) continue;
                                if (k >= S.n_bin) return false;
                                const zk_bin_gate &gt = S.bin_gates[k++];
                                const uint64_t u = (((uint64_t) p * c.CI + ci) * c.nxi + tx) * c.nyi + ty;
                                const uint64_t v = (uint64_t) c.wstart + (((uint64_t) co * c.CI + ci) * c.m + (tx - x)) * c.m + (ty - y);
                                if (gt.g != g || gt.u != u || gt.sc != 0 || gt.l != 2 || gt.v >= S.size_v[0] || S.ori_id_v[gt.v] != v) return false;
                            }
                    }
    if (k != S.n_bin) return false;
    // the only other gates may be uni gates whose operand lives in layer 0 (the biases): they stay on the generic lists
    for (uint64_t j = 0; j < S.n_uni; ++j)
        if (S.uni_gates[j].lu != 0) return false;
    return true;
}

// No hint for an NCONV layer (e.g. the reference's unmodified circuit generator drives this library): look for the parameters. The number
// of distinct bias operands gives channel_out, the smallest weight index the first weight, the subset size channel_out * channel_in * m^2;
// kernel size, padding, stride and the number of pictures are tried (square pictures), and every candidate has to reproduce the gate list
// in conv_hint_matches -- a wrong guess fails on its first gates.
static bool conv_infer(const zk_layer_desc &S, const zk_layer_desc &P, int layer, conv_desc &c) {
    if (S.ty != ZK_NCONV || !S.n_bin || !S.n_uni || !S.size_v[0] || S.n_uni > S.size) return false;
    uint32_t lo_u = 0xffffffffu, hi_u = 0;
    for (uint64_t j = 0; j < S.n_uni; ++j) {
        if (S.uni_gates[j].lu != 0 || S.uni_gates[j].u >= S.size_u[0]) return false;
        const uint32_t raw = S.ori_id_u[S.uni_gates[j].u];
        lo_u = std::min(lo_u, raw);
        hi_u = std::max(hi_u, raw);
    }
    const uint32_t CO = hi_u - lo_u + 1;                    // the biases are consecutive layer-0 entries
    if (!CO || S.size % CO || S.size_v[0] % CO) return false;
    uint32_t wstart = 0xffffffffu;
    for (uint32_t v = 0; v < S.size_v[0]; ++v) wstart = std::min(wstart, S.ori_id_v[v]);
    const uint64_t per_co = S.size / CO;                    // pictures * output positions
    for (uint32_t m = 1; m <= 7; m += 2) {
        if ((S.size_v[0] / CO) % (m * m)) continue;
        const uint32_t CI = S.size_v[0] / CO / (m * m);
        for (uint32_t pp = 1; pp <= 64; ++pp) {
            if (per_co % pp) continue;
            const uint64_t pos = per_co / pp;
            const uint32_t nxo = (uint32_t) std::llround(std::sqrt((double) pos));
            if ((uint64_t) nxo * nxo != pos) continue;
            for (uint32_t ls = 0; ls <= 2; ++ls)
                for (uint32_t pad = 0; pad < m; ++pad) {
                    const int64_t nxi = ((int64_t) (nxo - 1) << ls) + m - 2 * (int64_t) pad;
                    if (nxi < 1 || (uint64_t) pp * CI * nxi * nxi > P.size) continue;
                    zk_conv_hint h = {layer, pp, CO, CI, (uint32_t) nxi, (uint32_t) nxi, nxo, nxo, m, pad, ls, wstart};
                    if (conv_hint_matches(h, S, P, c)) return true;
                }
        }
    }
    return false;
}

extern "C" int32_t zk_structured_layers(const zk_ctx *ctx) { return ctx ? (int32_t) ctx->conv_layers : 0; }

extern "C" int32_t zk_upload_circuit(zk_ctx *ctx, const zk_layer_desc *layers, int32_t n_layers, const uint64_t *two_mul,
                                     int32_t n_two_mul) {
    return zk_upload_circuit_hinted(ctx, layers, n_layers, two_mul, n_two_mul, nullptr, 0);
}

// ---- one resident circuit per GPU, shared by its sessions (reference src/prover.hpp:47-48: one layeredCircuit per prover) ----
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
};
static std::mutex g_circ_mtx;
static std::vector<shared_circuit *> g_circuits;
static std::atomic<uint64_t> g_circ_builds{0}, g_circ_attaches{0}, g_circ_bytes{0};

extern "C" void zk_sharing_stats(uint64_t *circuit_builds, uint64_t *circuit_attaches) {
    if (circuit_builds) *circuit_builds = g_circ_builds.load();
    if (circuit_attaches) *circuit_attaches = g_circ_attaches.load();
}
extern "C" uint64_t zk_shared_circuit_bytes(void) { return g_circ_bytes.load(); }

// 128-bit digest of a byte range: four 64-bit multiply-rotate lanes (NOT cryptographic: the registry is process-local, a collision would
// attach a context to another circuit's lists and its proofs would be rejected -- a safe failure)
static void fast_digest(uint64_t h[4], const void *data, size_t n) {
    const uint8_t *p = (const uint8_t *) data;
    const uint64_t k0 = 0x9e3779b97f4a7c15ull, k1 = 0xc2b2ae3d27d4eb4full, k2 = 0x165667b19e3779f9ull, k3 = 0x27d4eb2f165667c5ull;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        h[0] = ((h[0] ^ w[0]) * k0); h[0] = (h[0] << 29) | (h[0] >> 35);
        h[1] = ((h[1] ^ w[1]) * k1); h[1] = (h[1] << 31) | (h[1] >> 33);
        h[2] = ((h[2] ^ w[2]) * k2); h[2] = (h[2] << 27) | (h[2] >> 37);
        h[3] = ((h[3] ^ w[3]) * k3); h[3] = (h[3] << 33) | (h[3] >> 31);
    }
    uint64_t tail[4] = {0, 0, 0, (uint64_t) n};
    std::memcpy(tail, p + i, n - i);
    for (int j = 0; j < 4; ++j) { h[j] = (h[j] ^ tail[j]) * k1; h[j] ^= h[j] >> 32; }
}
static void circuit_key(uint64_t key[2], const zk_layer_desc *layers, int32_t n_layers, const uint64_t *two_mul, int32_t n_two_mul,
                        const zk_conv_hint *hints, uint32_t n_hints) {
    uint64_t h[4] = {0x243f6a8885a308d3ull, 0x13198a2e03707344ull, 0xa4093822299f31d0ull, 0x082efa98ec4e6c89ull};
    for (int i = 0; i < n_layers; ++i) {
        const zk_layer_desc &S = layers[i];
        const int64_t rec[20] = {S.ty, (int64_t) S.size, S.bit_length, S.fft_bit_length, (int64_t) S.zero_start_id, (int64_t) S.n_uni, (int64_t) S.n_bin,
                                 (int64_t) S.size_u[0], (int64_t) S.size_u[1], (int64_t) S.size_v[0], (int64_t) S.size_v[1], S.bit_length_u[0], S.bit_length_u[1],
                                 S.bit_length_v[0], S.bit_length_v[1], S.max_bl_u, S.max_bl_v, S.need_phase2, 0, 0};
        fast_digest(h, rec, sizeof(rec));
        fast_digest(h, S.scale, 32);
        if (S.n_uni) fast_digest(h, S.uni_gates, (size_t) S.n_uni * sizeof(zk_uni_gate));
        if (S.n_bin) fast_digest(h, S.bin_gates, (size_t) S.n_bin * sizeof(zk_bin_gate));
        if (S.size_u[0] && S.ori_id_u) fast_digest(h, S.ori_id_u, (size_t) S.size_u[0] * 4);
        if (S.size_v[0] && S.ori_id_v) fast_digest(h, S.ori_id_v, (size_t) S.size_v[0] * 4);
    }
    fast_digest(h, two_mul, (size_t) n_two_mul * 32);
    if (n_hints) fast_digest(h, hints, (size_t) n_hints * sizeof(zk_conv_hint));
    key[0] = h[0] ^ (h[2] * 0x9e3779b97f4a7c15ull);
    key[1] = h[1] ^ (h[3] * 0xc2b2ae3d27d4eb4full);
}

static int32_t build_static(zk_ctx *ctx, const zk_layer_desc *layers, int32_t n_layers, const uint64_t *two_mul, int32_t n_two_mul,
                            const zk_conv_hint *hints, uint32_t n_hints);
static int32_t alloc_session(zk_ctx *ctx);

static void circuit_release(zk_ctx *ctx) {
    shared_circuit *e = (shared_circuit *) ctx->circuit;
    if (!e) return;
    ctx->circuit = nullptr;
    std::lock_guard<std::mutex> g(g_circ_mtx);
    if (--e->refs > 0) return;
    for (void *p : e->owned) hipFree(p);
    g_circ_bytes -= e->bytes;
    g_circuits.erase(std::remove(g_circuits.begin(), g_circuits.end(), e), g_circuits.end());
    delete e;
}

extern "C" int32_t zk_upload_circuit_hinted(zk_ctx *ctx, const zk_layer_desc *layers, int32_t n_layers, const uint64_t *two_mul,
                                            int32_t n_two_mul, const zk_conv_hint *hints, uint32_t n_hints) {
    if (!ctx || !layers || n_layers < 2 || !two_mul || n_two_mul > 512 || (n_hints && !hints)) return ZK_ERR_ARG;
    ZK_HIP(hipSetDevice(ctx->device));
    if (ctx->circuit_ready) { ctx->err = "circuit already uploaded; create a new context"; return ZK_ERR_STATE; }
    for (int i = 0; i < n_layers; ++i)
        if (layers[i].bit_length < 0 || layers[i].bit_length > ZK_MAX_VARS) { ctx->err = "layer bit length out of range"; return ZK_ERR_ARG; }
    uint64_t key[2];
    circuit_key(key, layers, n_layers, two_mul, n_two_mul, hints, n_hints);
    shared_circuit *e = nullptr;
    {
        std::lock_guard<std::mutex> g(g_circ_mtx);
        for (shared_circuit *c : g_circuits)
            if (c->device == ctx->device && c->key[0] == key[0] && c->key[1] == key[1] && !c->failed) { e = c; break; }
        if (!e) {
            e = new shared_circuit();
            e->device = ctx->device;
            e->key[0] = key[0]; e->key[1] = key[1];
            g_circuits.push_back(e);
        }
        ++e->refs;
    }
    ctx->circuit = e;
    int32_t rc = ZK_OK;
    {
        std::lock_guard<std::mutex> g(e->mtx);         // (a second context uploading the same circuit at the same time waits here, then attaches)
        if (!e->ready && !e->failed) {
            ctx->alloc_sink = &e->owned;
            ctx->sink_bytes = 0;
            rc = build_static(ctx, layers, n_layers, two_mul, n_two_mul, hints, n_hints);
            ctx->alloc_sink = nullptr;
            if (rc == ZK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "upload failed"; rc = ZK_ERR_HIP; }
            if (rc == ZK_OK) {
                e->L = ctx->L;
                e->two_mul = ctx->two_mul; e->n_two_mul = ctx->n_two_mul;
                e->liu_ptr = ctx->liu_ptr; e->liu_ent = ctx->liu_ent; e->liu_ntabs = ctx->liu_ntabs;
                e->liu_tab_layer = ctx->liu_tab_layer; e->liu_tab_side = ctx->liu_tab_side;
                e->conv_layers = ctx->conv_layers;
                e->sz = ctx->sz;
                e->ready = true;
                e->bytes = ctx->sink_bytes;
                g_circ_bytes += e->bytes;
                ++g_circ_builds;
            } else e->failed = true;
        } else if (e->ready) {
            ctx->L = e->L;
            ctx->two_mul = e->two_mul; ctx->n_two_mul = e->n_two_mul;
            ctx->liu_ptr = e->liu_ptr; ctx->liu_ent = e->liu_ent; ctx->liu_ntabs = e->liu_ntabs;
            ctx->liu_tab_layer = e->liu_tab_layer; ctx->liu_tab_side = e->liu_tab_side;
            ctx->conv_layers = e->conv_layers;
            ctx->sz = e->sz;
            ++g_circ_attaches;
        } else { ctx->err = "the circuit's first upload failed"; rc = ZK_ERR_STATE; }
    }
    if (rc == ZK_OK) rc = alloc_session(ctx);
    if (rc != ZK_OK) { circuit_release(ctx); return rc; }
    ctx->circuit_ready = true;
    return ZK_OK;
}

// the witness-independent part of a circuit (runs once per circuit and device)
static int32_t build_static(zk_ctx *ctx, const zk_layer_desc *layers, int32_t n_layers, const uint64_t *two_mul, int32_t n_two_mul,
                            const zk_conv_hint *hints, uint32_t n_hints) {
    int32_t rc;
    ctx->n_two_mul = n_two_mul;
    if ((rc = zk_dev_alloc(ctx, (void **) &ctx->two_mul, (size_t) n_two_mul * 32))) return rc;
    ZK_HIP(hipMemcpy(ctx->two_mul, two_mul, (size_t) n_two_mul * 32, hipMemcpyHostToDevice));

    ctx->L.assign(n_layers, dev_layer());
    // structured convolution layers: a hint that reproduces the layer's gate list switches the factored sums on (and the layer's phase-1 list
    // of those gates is then not built at all)
    for (uint32_t k = 0; k < n_hints; ++k) {
        const int i = hints[k].layer;
        if (i < 2 || i >= n_layers) continue;
        dev_layer &D = ctx->L[i];
        conv_desc c;
        if (D.conv_ok || !conv_hint_matches(hints[k], layers[i], layers[i - 1], c)) continue;
        D.conv = c;
        D.conv_ok = true;
        ++ctx->conv_layers;
    }
    for (int i = 2; i < n_layers; ++i) {                    // layers nobody described
        dev_layer &D = ctx->L[i];
        conv_desc c;
        if (D.conv_ok || layers[i].ty != ZK_NCONV || !conv_infer(layers[i], layers[i - 1], i, c)) continue;
        D.conv = c;
        D.conv_ok = true;
        ++ctx->conv_layers;
    }
    circuit_sizes &Z = ctx->sz;
    Z = circuit_sizes();
    uint64_t max_list = 1;
    for (int i = 0; i < n_layers; ++i) {
        dev_layer &D = ctx->L[i];
        const zk_layer_desc &S = layers[i];
        D.d = S;
        D.d.uni_gates = nullptr; D.d.bin_gates = nullptr; D.d.ori_id_u = nullptr; D.d.ori_id_v = nullptr;
        D.val_len = 1ull << S.bit_length;
        D.val_live = D.val_len;          // until the values arrive
        if (i == 0) { Z.tp_cap[1] = std::max<uint64_t>(Z.tp_cap[1], D.val_len); continue; }      // the layer-0 combine runs on pair 1
        if (i == n_layers - 1) Z.tp_cap[0] = std::max<uint64_t>(Z.tp_cap[0], D.val_len);          // Vres folds the output layer on pair 0
        Z.bg = std::max<uint64_t>(Z.bg, D.val_len);
        // exact sizes of the bookkeeping buffers (round 3: a session's buffers used to be eight times the LARGEST table of the circuit)
        const uint64_t prev_len = 1ull << layers[i - 1].bit_length;
        const bool dotp = S.ty == ZK_DOT_PROD, xform = S.ty == ZK_FFT || S.ty == ZK_IFFT;
        for (int b = 0; b < 2; ++b) {
            const int blu = dotp ? S.bit_length_u[1] : S.bit_length_u[b], blv = S.bit_length_v[b];
            for (int bl : {blu, blv}) {
                if (bl < 0) continue;
                const uint64_t len = 1ull << bl;
                Z.tp_cap[b] = std::max(Z.tp_cap[b], len);
                // V[0] of pair 1 is only written when the previous layer cannot be read in place, by the transform layers and by DOT_PROD
                if (b == 0 || len > prev_len || xform || dotp) Z.v0_cap[b] = std::max(Z.v0_cap[b], len);
            }
        }
        Z.bu = std::max<uint64_t>(Z.bu, 1ull << std::max<int>(std::max<int>(S.max_bl_u, S.max_bl_v), 0));
        for (int bl : {(int) S.bit_length_u[0], (int) S.bit_length_v[0]}) if (bl >= 0) Z.sub = std::max<uint64_t>(Z.sub, 1ull << bl);
        if (S.fft_bit_length >= 0) Z.gs = std::max<uint64_t>(Z.gs, 1ull << S.fft_bit_length);

        if (S.size_u[0]) {
            std::vector<uint32_t> t(S.ori_id_u, S.ori_id_u + S.size_u[0]);
            if ((rc = upload(ctx, &D.ori_u, t))) return rc;
        }
        if (S.size_v[0]) {
            std::vector<uint32_t> t(S.ori_id_v, S.ori_id_v + S.size_v[0]);
            if ((rc = upload(ctx, &D.ori_v, t))) return rc;
        }
        const bool dot = S.ty == ZK_DOT_PROD, xf = S.ty == ZK_FFT || S.ty == ZK_IFFT;
        if (xf) continue;          // no gates: the transform layers are proved through the DFT-matrix MLE

        // phase-2 list of the bin gates, keyed by v (also used by DOT_PROD)
        std::vector<gate_rec> q[2];
        for (uint64_t k = 0; k < S.n_bin; ++k) {
            const zk_bin_gate &g = S.bin_gates[k];
            const uint32_t v_prev = g.l & 1, u_prev = g.l != 0;
            gate_rec r = {g.g, g.v, g.u, (uint32_t) g.sc | (u_prev << 10)};
            q[v_prev].push_back(r);
        }
        for (int b = 0; b < 2; ++b) {
            if (q[b].empty()) continue;
            if (S.bit_length_v[b] < 0) { ctx->err = "bin gate refers to an absent v table"; return ZK_ERR_ARG; }
            counting_sort(q[b], 1u << S.bit_length_v[b]);
            D.p2_cov[b] = covered_prefix(q[b]);
            D.p2_live[b] = q[b].empty() ? 0 : q[b].back().key + 1;
            {
                const uint32_t f0 = GATE_IN_PREV(q[b][0].meta);
                bool same = true;
                for (const gate_rec &r : q[b]) if (GATE_IN_PREV(r.meta) != f0) { same = false; break; }
                D.p2_uniform[b] = same ? (int) f0 : -1;
            }
            D.n_p2_real[b] = q[b].size();
            D.p2_G[b] = choose_group(q[b]);
            pad_runs(q[b], D.p2_G[b]);
            D.n_p2[b] = q[b].size();
            max_list = std::max<uint64_t>(max_list, q[b].size());
            if ((rc = upload(ctx, &D.p2[b], q[b]))) return rc;
            std::vector<gate_rec>().swap(q[b]);
        }
        if (dot) {
            std::vector<gate_rec> d;
            d.reserve(S.n_bin);
            uint32_t rows = S.size_u[1] >> S.fft_bit_length;
            for (uint64_t k = 0; k < S.n_bin; ++k) {
                const zk_bin_gate &g = S.bin_gates[k];
                if (g.u >= rows) { ctx->err = "DOT_PROD gate out of range"; return ZK_ERR_ARG; }
                gate_rec r = {g.g, g.u, g.v, 0};
                d.push_back(r);
            }
            counting_sort(d, rows);
            std::vector<uint32_t> ptr((size_t) rows + 1, 0);
            for (const gate_rec &r : d) ++ptr[r.key + 1];
            for (uint32_t k = 0; k < rows; ++k) ptr[k + 1] += ptr[k];
            D.d1_rows = rows;
            if ((rc = upload(ctx, &D.d1, d)) || (rc = upload(ctx, &D.d1_rowptr, ptr))) return rc;
            continue;
        }
        // phase-1 lists keyed by u; uni and bin gates of one table merged into one sorted list
        std::vector<gate_rec> p[2], un;
        un.reserve(S.n_uni);
        for (uint64_t k = 0; k < S.n_uni; ++k) {
            const zk_uni_gate &g = S.uni_gates[k];
            const uint32_t in_prev = g.lu != 0;
            gate_rec r = {g.g, g.u, 0, (uint32_t) g.sc};
            p[in_prev].push_back(r);
            ++D.n_p1_uni[in_prev];
            gate_rec r2 = {g.g, 0, g.u, (uint32_t) g.sc | (in_prev << 10)};
            un.push_back(r2);
        }
        for (uint64_t k = 0; k < S.n_bin && !D.conv_ok; ++k) {          // (a structured convolution: every bin gate is covered by the factored sum)
            const zk_bin_gate &g = S.bin_gates[k];
            const uint32_t v_prev = g.l & 1, u_prev = g.l != 0;
            const uint32_t vres = v_prev ? g.v : S.ori_id_v[g.v];     // resolve the layer-0 subset index once
            gate_rec r = {g.g, g.u, vres, (uint32_t) g.sc | (1u << 9) | (v_prev << 10)};
            p[u_prev].push_back(r);
        }
        if (D.conv_ok) D.p1_live[1] = (uint32_t) std::min<uint64_t>((uint64_t) D.conv.pp * D.conv.CI * D.conv.nxi * D.conv.nyi, 0xffffffffu);
        for (int b = 0; b < 2; ++b) {
            if (p[b].empty()) continue;
            if (S.bit_length_u[b] < 0) { ctx->err = "gate refers to an absent u table"; return ZK_ERR_ARG; }
            counting_sort(p[b], 1u << S.bit_length_u[b]);
            D.p1_cov[b] = covered_prefix(p[b]);
            D.p1_live[b] = p[b].empty() ? 0 : p[b].back().key + 1;
            D.n_p1_real[b] = p[b].size();
            D.p1_G[b] = choose_group(p[b]);
            pad_runs(p[b], D.p1_G[b]);
            D.n_p1[b] = p[b].size();
            max_list = std::max<uint64_t>(max_list, p[b].size());
            if ((rc = upload(ctx, &D.p1[b], p[b]))) return rc;
            std::vector<gate_rec>().swap(p[b]);
        }
        D.n_uni2 = un.size();
        if ((rc = upload(ctx, &D.uni2, un))) return rc;
    }
    // layer-0 combine: every (layer, side) whose operands reach into layer 0 is a table; CSR of its (index h -> layer-0 index ori[h]) pairs by x
    {
        const uint64_t n0 = 1ull << layers[0].bit_length;
        std::vector<uint32_t> cnt(n0 + 1, 0);
        uint64_t total = 0;
        for (int i = 1; i < n_layers; ++i) {
            const zk_layer_desc &S = layers[i];
            for (int side = 0; side < 2; ++side) {
                const int bl = side ? S.bit_length_v[0] : S.bit_length_u[0];
                const uint32_t sz = side ? S.size_v[0] : S.size_u[0];
                const uint32_t *ori = side ? S.ori_id_v : S.ori_id_u;
                if (bl < 0 || !sz) continue;
                if (bl > 24) { ctx->err = "layer-0 subset table too large for the combined gather"; return ZK_ERR_ARG; }
                ctx->liu_tab_layer.push_back(i);
                ctx->liu_tab_side.push_back(side);
                for (uint32_t h = 0; h < sz; ++h) {
                    if (ori[h] >= n0) { ctx->err = "ori_id out of range"; return ZK_ERR_ARG; }
                    ++cnt[ori[h] + 1];
                }
                total += sz;
            }
        }
        if (total >= 0xffffffffull || ctx->liu_tab_layer.size() >= (1u << 24)) { ctx->err = "too many layer-0 references"; return ZK_ERR_ARG; }
        for (uint64_t x = 0; x < n0; ++x) cnt[x + 1] += cnt[x];
        std::vector<liu_entry> ent(total);
        {
            std::vector<uint32_t> pos(cnt.begin(), cnt.end() - 1);
            for (size_t tb = 0; tb < ctx->liu_tab_layer.size(); ++tb) {
                const zk_layer_desc &S = layers[ctx->liu_tab_layer[tb]];
                const int side = ctx->liu_tab_side[tb];
                const uint32_t sz = side ? S.size_v[0] : S.size_u[0];
                const uint32_t *ori = side ? S.ori_id_v : S.ori_id_u;
                const int bl = side ? S.bit_length_v[0] : S.bit_length_u[0];
                for (uint32_t h = 0; h < sz; ++h) {
                    liu_entry e = {h, (uint32_t) tb | ((uint32_t) (bl >> 1) << 24)};      // table number, and the split of its index into half-table indices
                    ent[pos[ori[h]]++] = e;
                }
            }
        }
        ctx->liu_ntabs = (uint32_t) ctx->liu_tab_layer.size();
        liu_entry *d_ent = nullptr;
        if ((rc = upload(ctx, &ctx->liu_ptr, cnt)) || (rc = upload(ctx, &d_ent, ent))) return rc;
        ctx->liu_ent = d_ent;
    }
    for (const dev_layer &D : ctx->L) {
        if (!D.conv_ok) continue;
        const conv_desc &c = D.conv;
        const uint64_t len = (uint64_t) c.CI * c.m * c.m, chunks = (c.CO + 7) / 8;
        Z.conv_wa = std::max(Z.conv_wa, 2 * len);
        Z.conv_part = std::max(Z.conv_part, chunks * 2 * len);
        Z.conv_ae = std::max<uint64_t>(Z.conv_ae, (uint64_t) c.CO * c.m * c.m + c.CI);
    }
    Z.max_list = max_list;
    if (getenv("ZKCNN_DUMP_TABLES"))
        for (int i = 1; i < n_layers; ++i) {
            const dev_layer &D = ctx->L[i];
            fprintf(stderr, "[tables] layer %d ty %d size %u | u0 bl %d live %u | u1 bl %d live %u | v0 bl %d live %u | v1 bl %d live %u\n", i, D.d.ty, D.d.size,
                    D.d.bit_length_u[0], D.p1_live[0], D.d.bit_length_u[1], D.p1_live[1], D.d.bit_length_v[0], D.p2_live[0], D.d.bit_length_v[1], D.p2_live[1]);
        }
    return ZK_OK;
}

// what every session of a circuit has for itself: layer values, bookkeeping tables, scratch
static int32_t alloc_session(zk_ctx *ctx) {
    int32_t rc;
    const circuit_sizes &Z = ctx->sz;
    for (dev_layer &D : ctx->L) {
        D.val = nullptr;
        D.val_live = D.val_len;          // until the values arrive
        if ((rc = zk_dev_alloc(ctx, (void **) &D.val, D.val_len * 32))) return rc;
        ZK_HIP(hipMemsetAsync(D.val, 0, D.val_len * 32, ctx->stream));
        D.ev_uni = D.ev_bin = nullptr; D.ev_dot = nullptr; D.ev_dot_ptr = nullptr; D.n_ev_uni = D.n_ev_bin = 0; D.ev_conv = false;
    }
    {
        const uint32_t nt = std::max<uint32_t>(ctx->liu_ntabs, 1);
        if ((rc = zk_dev_alloc(ctx, (void **) &ctx->liu_halves, (size_t) nt * 2 * LIU_HALF_STRIDE * 32))) return rc;
        ZK_HIP(hipHostMalloc(&ctx->h_liu_tabs, (size_t) nt * sizeof(liu_table), hipHostMallocMapped));
        ZK_HIP(hipHostGetDevicePointer(&ctx->liu_tabs, ctx->h_liu_tabs, 0));        // the kernels read the descriptors in place
    }
    if (ctx->conv_layers) {
        if ((rc = zk_dev_alloc(ctx, (void **) &ctx->conv_small, (size_t) CT_COUNT * CONV_TAB_STRIDE * 32)) ||
            (rc = zk_dev_alloc(ctx, (void **) &ctx->conv_wa, Z.conv_wa * 32)) || (rc = zk_dev_alloc(ctx, (void **) &ctx->conv_part, Z.conv_part * 32)) ||
            (rc = zk_dev_alloc(ctx, (void **) &ctx->conv_e, 2 * 16 * 16 * 32)) || (rc = zk_dev_alloc(ctx, (void **) &ctx->conv_ae, Z.conv_ae * 32)))
            return rc;
        ZK_HIP(hipHostMalloc(&ctx->h_conv_tabs, 2 * CT_COUNT * sizeof(liu_table), hipHostMallocMapped));
        ZK_HIP(hipHostGetDevicePointer(&ctx->conv_tabs, ctx->h_conv_tabs, 0));
    }
    // bookkeeping tables: [0] takes a pair's tables as they are built, [1] only ever what a fold leaves (half, plus the guard quad)
    for (int b = 0; b < 2; ++b) {
        const uint64_t cap = std::max<uint64_t>(Z.tp_cap[b], 4), half = cap / 2 + 8;
        // (V[0] of a pair whose V table is read in place still takes every SECOND fold: a quarter of the table)
        if ((rc = zk_dev_alloc(ctx, (void **) &ctx->tp[b].V[0], std::max<uint64_t>(Z.v0_cap[b], cap / 4 + 8) * 32)) ||
            (rc = zk_dev_alloc(ctx, (void **) &ctx->tp[b].M[0], cap * 32)) ||
            (rc = zk_dev_alloc(ctx, (void **) &ctx->tp[b].V[1], half * 32)) || (rc = zk_dev_alloc(ctx, (void **) &ctx->tp[b].M[1], half * 32)))
            return rc;
    }
    ctx->beta_g_cap = std::max<uint64_t>(Z.bg, 1);
    for (int k = 0; k < 2; ++k) if ((rc = zk_dev_alloc(ctx, (void **) &ctx->beta_g[k], ctx->beta_g_cap * 32))) return rc;
    ctx->beta_u_cap = std::max<uint64_t>(Z.bu, 1);
    if ((rc = zk_dev_alloc(ctx, (void **) &ctx->beta_u, ctx->beta_u_cap * 32))) return rc;
    ctx->beta_gs_cap = std::max<uint64_t>(Z.gs, 1);
    if ((rc = zk_dev_alloc(ctx, (void **) &ctx->beta_gs, ctx->beta_gs_cap * 32))) return rc;
    for (int k = 0; k < 2; ++k) if ((rc = zk_dev_alloc(ctx, (void **) &ctx->small[k], ctx->beta_gs_cap * 32))) return rc;
    ctx->carry_slots = 4 * ((Z.max_list + ZK_BLOCK - 1) / ZK_BLOCK) + 4;          // two lists per launch (k_gate_multi), two slots per block
    if ((rc = zk_dev_alloc(ctx, (void **) &ctx->carry_key, ctx->carry_slots * 4))) return rc;
    if ((rc = zk_dev_alloc(ctx, (void **) &ctx->carry_val, ctx->carry_slots * 32))) return rc;
    if ((rc = zk_scratch(ctx, (size_t) 1 << 24))) return rc;
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

extern "C" int32_t zk_upload_layer_values(zk_ctx *ctx, int32_t layer, const uint64_t *values, uint64_t n) {
    if (!ctx || !ctx->circuit_ready || layer < 0 || layer >= (int) ctx->L.size()) return ZK_ERR_ARG;
    dev_layer &D = ctx->L[layer];
    if (n > D.val_len) { ctx->err = "more values than the layer holds"; return ZK_ERR_ARG; }
    ZK_HIP(hipSetDevice(ctx->device));
    if (n) ZK_HIP(hipMemcpyAsync(D.val, values, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if (n < D.val_len) ZK_HIP(hipMemsetAsync(D.val + n, 0, (D.val_len - n) * 32, ctx->stream));
    uint64_t last = n;
    while (last && !(values[4 * last - 4] | values[4 * last - 3] | values[4 * last - 2] | values[4 * last - 1])) --last;
    D.val_live = last;
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

extern "C" int32_t zk_poke_layer_value(zk_ctx *ctx, int32_t layer, uint64_t index, const uint64_t value[4]) {
    if (!ctx || !ctx->circuit_ready || layer < 0 || layer >= (int) ctx->L.size() || !value) return ZK_ERR_ARG;
    dev_layer &D = ctx->L[layer];
    if (index >= D.d.size) { ctx->err = "poke: index behind the layer"; return ZK_ERR_ARG; }
    ZK_HIP(hipSetDevice(ctx->device));
    ZK_HIP(hipMemcpyAsync(D.val + index, value, 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    if ((value[0] | value[1] | value[2] | value[3]) && index + 1 > D.val_live) D.val_live = index + 1;      // (an upper bound stays an upper bound)
    return ZK_OK;
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
    prep_plan() { std::memset(&A, 0, sizeof(A)); }
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
        J.c = std::min(n, n <= 16 ? 8 : n <= 20 ? 10 : 12);
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
        ZK_LAUNCH(PC_EQ, 0.0, k_prep, dim3(blocks), dim3(ZK_BLOCK), A);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
};

static int32_t eq_table(zk_ctx *ctx, fr_t *out, int n, const HFr *r0, const HFr &a, const HFr *r1, const HFr &b,
                        uint64_t tail_start, const HFr &tail_scale, uint64_t limit = ~0ull) {
    if (n < 0) return ZK_OK;
    if (n > ZK_MAX_VARS) { ctx->err = "eq table too large"; return ZK_ERR_ARG; }
    prep_plan P;
    P.eq(out, n, r0, a, r1, b, tail_start, tail_scale, limit);
    return P.launch(ctx);
}
static int32_t eq_table1(zk_ctx *ctx, fr_t *out, int n, const HFr *r, const HFr &init, uint64_t limit = ~0ull) {
    return eq_table(ctx, out, n, r, init, nullptr, HFr(0LL), ~0ull, HFr::one(), limit);
}

int32_t zk_eq_table1_dev(zk_ctx *ctx, fr_t *out, int n, const HFr *r, const HFr &init) { return eq_table1(ctx, out, n, r, init); }

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
    ZK_LAUNCH(PC_MATVEC, 0.0, k_col_combine, grid, dim3(ZK_BLOCK), part, Z, L, cols, rows, per);
    if (chunks > 1)
        ZK_LAUNCH(PC_MATVEC, 0.0, k_sum_rows, dim3((cols + 63) / 64), dim3(1024), out, part, cols, chunks);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// copies `count` elements of d_result to the pinned host mirror and waits for the stream
static int32_t fetch_result(zk_ctx *ctx, int count) {
    ZK_HIP(hipMemcpyAsync(ctx->h_result, ctx->d_result, (size_t) count * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// algorithmic bytes of one scatter: per uni gate 12 B record + 32 B gather, per bin gate 16 B record
// + 2 x 32 B gathers, + 32 B per output entry
// Waits until the fused round kernel has published sequence number `seq` in the mapped host slot. Spinning on the
// slot keeps a stream synchronisation (and a D2H copy) off the critical path of every round; if the value does not
// show up quickly the stream is synchronised instead (long kernels, or host memory that is not fine-grained).
static int32_t wait_slot(zk_ctx *ctx, unsigned long long seq) {
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
    ZK_LAUNCH(PC_GATE, gate_bytes, k_gate_multi, dim3(blocks), dim3(ZK_BLOCK), A);
    if (need_fix) {
        const uint64_t n0 = (A.nlists > 0 && !A.L[0].direct) ? 2ull * A.L[0].nblk : 0, n1 = (A.nlists > 1 && !A.L[1].direct) ? 2ull * A.L[1].nblk : 0;
        // (carry slots are handed out in list order: a direct list takes none)
        fr_t *o0 = n0 ? A.L[0].out : (n1 ? A.L[1].out : nullptr), *o1 = n0 && n1 ? A.L[1].out : nullptr;
        ZK_LAUNCH(PC_GATE_FIX, 0.0, k_gate_fixup2, dim3(grid_for(n0 + n1)), dim3(ZK_BLOCK), o0, o1, ctx->carry_key, ctx->carry_val, n0 ? n0 : n1, n0 ? n1 : 0);
    }
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// a table of more than 2^18 entries is first read by the round kernel that stops at the live prefix (rounded up to whole quads): its builders
// stop there too (one guard quad behind it); smaller tables are read whole
#define ZK_FULL_TABLE_LOG 18          // tables of up to 2^18 entries are always complete: their readers (k_round_quad_fine, k_mid, k_tail) know no bound
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
    for (int b = 0; b < 2; ++b) {
        ctx->tp[b].cur = 0;
        ctx->tp[b].tail_valid = false;
        ctx->tp[b].Vsrc = nullptr;
        ctx->tp[b].len = bl[b] >= 0 ? 1ull << bl[b] : 0;
        ctx->tp[b].live = ctx->tp[b].len;
        ctx->tp[b].absorbed = false;
        ctx->tp[b].final_v.clear();
    }
}

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

static int32_t phi_table(zk_ctx *ctx, fr_t *out, const HFr *rx, const HFr &scale, int n, bool inverse) {
    fr_t *pw = powers_of_root(ctx, n, inverse);
    if (!pw) { ctx->err = "root table allocation failed"; return ZK_ERR_NOMEM; }
    const int vars = inverse ? n - 1 : n;
    const uint32_t cnt = inverse ? 1u << n : 1u << (n - 1);
    fr_vec R;
    for (int j = 0; j < vars; ++j) R.v[j] = to_dev(rx[j]);
    ZK_LAUNCH(PC_PHI, 0.0, k_phi, dim3((cnt + ZK_BLOCK - 1) / ZK_BLOCK), dim3(ZK_BLOCK), out, pw, R, to_dev(scale), n, vars, cnt);
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
    ZK_LAUNCH(PC_MATVEC, 0.0, k_strided_matvec, grid, dim3(ZK_BLOCK), part, val, beta, len, stride, cnt, per);
    if (chunks > 1)
        ZK_LAUNCH(PC_MATVEC, 0.0, k_sum_rows, dim3((len + 63) / 64), dim3(1024), out, part, len, chunks);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// folds table `b`'s V (and M if with_m) with r; len must be >= 2
static int32_t fold_pair(zk_ctx *ctx, table_pair &t, const HFr &r, bool with_m) {
    const fr_t rr = to_dev(r);
    ZK_LAUNCH(PC_FOLD, 0.0, k_fold, dim3(grid_for(t.len / 2)), dim3(ZK_BLOCK), vin(t), t.V[t.cur ^ 1], t.len, rr);
    t.Vsrc = nullptr;
    if (with_m)
        ZK_LAUNCH(PC_FOLD, 0.0, k_fold, dim3(grid_for(t.len / 2)), dim3(ZK_BLOCK), t.M[t.cur], t.M[t.cur ^ 1], t.len, rr);
    ZK_HIP(hipGetLastError());
    t.cur ^= 1;
    t.len >>= 1;
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// state machine
// ------------------------------------------------------------------------------------------------
// every entry point but the three round calls first sends a resident round kernel home (its phase was abandoned)
#define CHECK_READY_ROUND() do { if (!ctx || !ctx->circuit_ready) return ZK_ERR_STATE; ZK_HIP(hipSetDevice(ctx->device)); } while (0)
#define CHECK_READY() do { CHECK_READY_ROUND(); if (ctx->live_active) { int32_t rc_ = zk_live_abort(ctx); if (rc_) return rc_; } } while (0)
static inline const HFr &H(const uint64_t *p) { return *reinterpret_cast<const HFr *>(p); }
static inline void put(uint64_t *dst, const HFr &x) { std::memcpy(dst, &x, 32); }

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
    ZK_HIP(hipMemcpyAsync(ctx->d_result, vin(t), 32, hipMemcpyDeviceToDevice, ctx->stream));
    t.Vsrc = nullptr;
    int32_t rc = fetch_result(ctx, 1);
    if (rc) return rc;
    put(out, ctx->h_result[0]);
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
    ZK_LAUNCH(PC_GATE, 0.0, k_conv_wa, dim3((wlen + ZK_BLOCK - 1) / ZK_BLOCK, chunks), dim3(ZK_BLOCK), part, (const fr_t *) ctx->L[0].val + c.wstart,
              (const fr_t *) ctx->conv_small, wlen, c.CO, per, K);
    if (chunks > 1)
        ZK_LAUNCH(PC_GATE, 0.0, k_sum_rows, dim3((K * wlen + 63) / 64), dim3(1024), ctx->conv_wa, (const fr_t *) part, K * wlen, chunks);
    ZK_LAUNCH(PC_GATE, 0.0, k_conv_m1, dim3(grid_for(len)), dim3(ZK_BLOCK), M, (const fr_t *) ctx->conv_wa, (const fr_t *) ctx->conv_small, c, K, len);
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
    ZK_LAUNCH(PC_GATE, 0.0, k_conv_e, dim3(mm, K), dim3(ZK_BLOCK), ctx->conv_e, (const fr_t *) ctx->conv_small, c);
    ZK_LAUNCH(PC_GATE, 0.0, k_conv_ae, dim3((c.CO * mm + c.CI + ZK_BLOCK - 1) / ZK_BLOCK), dim3(ZK_BLOCK), ctx->conv_ae, (const fr_t *) ctx->conv_e,
              (const fr_t *) ctx->conv_small, c, K, to_dev(ctx->V_u1));
    ZK_LAUNCH(PC_GATE, 0.0, k_conv_m2, dim3(grid_for(len)), dim3(ZK_BLOCK), M, (const uint32_t *) cur.ori_v, (const fr_t *) ctx->conv_ae, c, cur.d.size_v[0], len);
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
    const HFr scale = H(d.scale);
    prep_plan P;

    if (d.ty == ZK_FFT || d.ty == ZK_IFFT) {
        const bool fwd = d.ty == ZK_FFT;
        const int fft_bl = d.fft_bit_length, fft_blh = fft_bl - 1;
        const int cnt_bl = fwd ? d.bit_length - fft_bl : d.bit_length - fft_blh;
        const uint32_t cnt_len = d.size >> (fwd ? fft_bl : fft_blh);
        fr_t *bg = ctx->beta_g[ctx->beta_g_cur];
        if (fwd) P.eq(bg, cnt_bl, ctx->r_0 + fft_bl, ctx->alpha, ctx->r_1, ctx->beta, ~0ull, HFr::one());
        else P.eq1(bg, cnt_bl, ctx->r_0 + fft_blh, ctx->alpha);
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
        ZK_LAUNCH(PC_EQ, 0.0, k_outer_expand, dim3(grid_for(cur.val_len)), dim3(ZK_BLOCK), dst, src, ctx->beta_gs, fft_blh, cur.val_len);
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
    ctx->phase_rounds = 0;                  // cubic rounds are always driven from the host
    ctx->tail_active = false;
    ctx->host_tail_active = false;
    ctx->last_poly_valid = false;
    ctx->round = 0;
    ctx->small_len = 1u << fft_bl;
    ctx->small_cur = 0;
    const uint64_t N = ctx->tp[1].len;
    prep_plan P;
    P.eq1(ctx->small[0], fft_bl, ctx->r_0, HFr::one());
    load_v_table(ctx, P, ctx->tp[1], 1, d.bit_length_u[1], d.size_u[1], nullptr, prev);
    // (k_dot_v0 writes the rows that have gates; the rest of V0 is zero)
    const uint64_t covered = (uint64_t) cur.d1_rows << fft_bl;
    if (covered < N) P.zero(ctx->tp[0].V[0] + covered, N - covered);
    if ((rc = P.launch(ctx))) return rc;
    if (cur.d1_rows) {
        dim3 grid(((1u << fft_bl) + ZK_BLOCK - 1) / ZK_BLOCK, cur.d1_rows);
        ZK_LAUNCH(PC_DOT, 0.0, k_dot_v0, grid, dim3(ZK_BLOCK), ctx->tp[0].V[0], prev.val, ctx->beta_g[ctx->beta_g_cur], cur.d1, cur.d1_rowptr, fft_bl);
        ZK_HIP(hipGetLastError());
    }
    return ZK_OK;
}

extern "C" int32_t zk_sumcheck_dotprod_update1(zk_ctx *ctx, const uint64_t prev_r[4], uint64_t out_abcd[16]) {
    CHECK_READY();
    const int id = ctx->sumcheck_id;
    const HFr r = H(prev_r);
    const bool first = ctx->round == 0;
    if (!first) ctx->r_u[id].at(ctx->round - 1) = r;
    ++ctx->round;
    if (!first && ctx->small_len >= 2) {
        ZK_LAUNCH(PC_FOLD, 0.0, k_fold, dim3(grid_for(ctx->small_len / 2)), dim3(ZK_BLOCK), ctx->small[ctx->small_cur], ctx->small[ctx->small_cur ^ 1], (uint64_t) ctx->small_len, to_dev(r));
        ctx->small_cur ^= 1;
        ctx->small_len >>= 1;
    }
    table_pair &t0 = ctx->tp[0], &t1 = ctx->tp[1];
    const uint64_t n = t1.len;
    const uint64_t npairs = first ? n / 2 : n / 4;
    if (npairs == 0) return ZK_ERR_STATE;
    const uint32_t g = std::min<uint32_t>(grid_for(npairs, 1024), ctx->partial_blocks);
    const unsigned long long seq = ++ctx->slot_seq;
    ZK_LAUNCH(PC_ROUND_CUBIC, (first ? 64.0 : 96.0) * (double) n, k_round_cubic, dim3(g), dim3(ZK_BLOCK), t0.V[t0.cur], vin(t1),
              t0.V[t0.cur ^ 1], t1.V[t1.cur ^ 1], ctx->small[ctx->small_cur], ctx->small_len, n, to_dev(r), first ? 1 : 0, ctx->partials,
              ctx->d_counter, (host_slot *) ctx->d_slot, seq);
    ZK_HIP(hipGetLastError());
    if (!first) {
        t0.cur ^= 1; t1.cur ^= 1;
        t0.len >>= 1; t1.len >>= 1;
        t1.Vsrc = nullptr;
    }
    int32_t rc = wait_slot(ctx, seq);
    if (rc) return rc;
    for (int k = 0; k < 4; ++k) ctx->h_result[k] = ctx->h_slot->v[k];
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
    eval_args E;
    std::memset(&E, 0, sizeof(E));
    E.p[0] = vin(t1); E.n[0] = (uint32_t) std::min<uint64_t>(t1.len, 2);
    E.p[1] = ctx->small[ctx->small_cur]; E.n[1] = std::min<uint32_t>(ctx->small_len, 2);
    if (t1.len > 2 || ctx->small_len > 2) return ZK_ERR_STATE;
    E.r = to_dev(r);
    E.slot = (host_slot *) ctx->d_slot;
    E.seq = ++ctx->slot_seq;
    ZK_LAUNCH(PC_FOLD, 0.0, k_eval_pairs, dim3(1), dim3(64), E);
    ZK_HIP(hipGetLastError());
    if ((rc = wait_slot(ctx, E.seq))) return rc;
    ctx->h_result[0] = ctx->h_slot->v[0];
    ctx->h_result[1] = ctx->h_slot->v[1];
    t1.len = 0;
    ctx->small_len = 1;
    put(claim_1, ctx->h_result[0]);
    ctx->V_u1 = ctx->h_result[0] * ctx->h_result[1];
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
    const HFr *ru = ctx->r_u[id].data();
    prep_plan P;

    if (d.ty == ZK_DOT_PROD) {
        const int fft_bl = d.fft_bit_length, cnt_bl = d.max_bl_v;
        table_pair &t = ctx->tp[1];
        const uint32_t rows = d.size_v[1];
        P.eq1(ctx->beta_u, cnt_bl, ru + fft_bl, HFr::one());
        P.eq1(ctx->beta_gs, fft_bl, ru, HFr::one());
        if (rows < t.len) P.zero(t.V[0] + rows, t.len - rows);              // (k_row_dot writes one entry per row)
        if (cur.p2_cov[1] < t.len) P.zero(t.M[0] + cur.p2_cov[1], t.len - cur.p2_cov[1]);
        if ((rc = P.launch(ctx))) return rc;
        ZK_LAUNCH(PC_DOT, 0.0, k_row_dot, dim3((rows + 3) / 4), dim3(ZK_BLOCK), t.V[0], prev.val, ctx->beta_gs, rows, fft_bl);
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

// One quadratic round over the live table pairs. reference src/prover.cpp:368-426.
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// Non-interactive mode: all remaining rounds of the phase in one single-workgroup kernel (fs_tail.cuh). Called at the start of a round
// whose live tables are small; fills the context's tail record and leaves every table pair either collapsed or down to its last pair.
// (Round 2 also had CHAINED launches for tables of up to 2^16 entries -- one launch per round, the next challenge derived in its last block;
// transcripts identical, 0.5 ms per vgg11 proof slower than host-driven rounds: removed in round 3.)
static int32_t run_device_rounds(zk_ctx *ctx, const HFr &r, bool with_add_term) {
    const bool first = ctx->round == 0;
    const unsigned long long seq = ++ctx->tail_seq;
    tail_args A;
    std::memset(&A, 0, sizeof(A));
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        A.Vin[b] = vin(t); A.Min[b] = t.M[t.cur];
        A.Vbuf[b][0] = t.V[0]; A.Vbuf[b][1] = t.V[1];
        A.Mbuf[b][0] = t.M[0]; A.Mbuf[b][1] = t.M[1];
        A.out_idx[b] = t.cur ^ 1;
        A.n[b] = t.len;
    }
    A.first = first ? 1 : 0;
    A.rounds = ctx->phase_rounds - ctx->round;
    A.with_add_term = with_add_term ? 1 : 0;
    A.prev_r = to_dev(r);
    A.add_term = to_dev(ctx->add_term);
    std::memcpy(A.fs_state, ctx->fs_state, 32);
    A.out = (tail_out *) ctx->d_tail;
    A.seq = seq;
    ZK_LAUNCH(PC_TAIL, 0.0, k_tail<false>, dim3(1), dim3(TAIL_THREADS), A);
    ZK_HIP(hipGetLastError());
    volatile unsigned long long *p = &((tail_out *) ctx->h_tail)->seq;
    for (uint64_t spins = 0; *p != seq; ++spins) {
        if (spins > (1ull << 24)) {
            ZK_HIP(hipStreamSynchronize(ctx->stream));
            if (*p != seq) { ctx->err = "device rounds were not published"; return ZK_ERR_STATE; }
            break;
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const tail_out *o = (const tail_out *) ctx->h_tail;
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        t.Vsrc = nullptr;
        if (o->pair_state[b] == 1) {
            t.len = 2;
            std::memcpy(&t.tail_v[0], &o->tail_v[b][0], 32);
            std::memcpy(&t.tail_v[1], &o->tail_v[b][1], 32);
            t.tail_valid = true;
        } else {
            t.len = 0;
            t.absorbed = true;
            std::memcpy(&t.final_v, &o->final_v[b], 32);
        }
    }
    std::memcpy(&ctx->add_term, &o->add_term, 32);
    ctx->tail_active = true;
    ctx->tail_count = A.rounds;
    ctx->tail_cursor = 0;
    ctx->last_poly_valid = false;          // (the rounds answered from the record do not maintain the running claim)
    ctx->tail_rounds_total += (uint64_t) ctx->tail_count;
    ++ctx->tail_phases_total;
    return ZK_OK;
}

// ---- persistent rounds of the INTERACTIVE protocol (k_tail<true>, fs_tail.cuh): the kernel is resident for the rest of the phase, a round is
// "challenge into the mailbox, polynomial out of the mailbox" ----
static inline void live_post(zk_ctx *ctx, const HFr &r, uint32_t seq) {
    uint32_t w[8];
    std::memcpy(w, &r, 32);
    live_in *m = (live_in *) ctx->h_live_in;
    // one 16-byte store per chunk (the kernel reads a chunk with one 16-byte load and checks the number in each)
    _mm_store_si128((__m128i *) m->c[0], _mm_set_epi32((int) seq, (int) w[2], (int) w[1], (int) w[0]));
    _mm_store_si128((__m128i *) m->c[1], _mm_set_epi32((int) seq, (int) w[5], (int) w[4], (int) w[3]));
    _mm_store_si128((__m128i *) m->c[2], _mm_set_epi32((int) seq, 0, (int) w[7], (int) w[6]));
    _mm_sfence();
}
int32_t zk_live_abort(zk_ctx *ctx) {
    if (!ctx->live_active) return ZK_OK;
    live_post(ctx, HFr(0LL), TAIL_ABORT);
    ctx->live_active = false;
    ctx->live_mid = false;
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_HIP(hipMemsetAsync(ctx->d_counter, 0, 64, ctx->stream));          // (a segment kernel that was sent home may have left arrivals behind ...
    ZK_HIP(hipMemsetAsync(ctx->d_bcast, 0, sizeof(mid_bcast), ctx->stream));   //  ... and the abort word in its broadcast line)
    std::memset(ctx->h_live_in, 0, sizeof(live_in));
    for (int b = 0; b < 2; ++b) ctx->tp[b].len = 0;
    return ZK_OK;
}
// waits for the polynomial of round k of the running kernel; false if the kernel has left (status) or nothing arrives
static int32_t live_wait(zk_ctx *ctx, int k, uint64_t out_abc[12]) {
    const tail_out *o = (const tail_out *) ctx->h_tail;
    const uint32_t want = ctx->live_seq32 + (uint32_t) k;
    uint32_t w[24];
    for (uint64_t spins = 0;; ++spins) {
        bool ok = true;
        for (int j = 0; j < 8; ++j) {
            const __m128i c = _mm_load_si128((const __m128i *) o->live.c[j]);
            alignas(16) uint32_t t[4];
            _mm_store_si128((__m128i *) t, c);
            if (t[3] != want) { ok = false; break; }
            w[3 * j] = t[0]; w[3 * j + 1] = t[1]; w[3 * j + 2] = t[2];
        }
        if (ok) break;
        if (spins > (1ull << 16) && (spins & 1023) == 0) {
            if (*(volatile const uint32_t *) &o->status != 0 || hipStreamQuery(ctx->stream) == hipSuccess) {
                // (a kernel that has left: one last look, its final message may have landed after the check above)
                bool late = true;
                for (int j = 0; j < 8; ++j) if (((volatile const uint32_t *) o->live.c[j])[3] != want) late = false;
                if (late) continue;
                ctx->live_active = false;
                char msg[256];
                const live_in *mi = (const live_in *) ctx->h_live_in;
                std::snprintf(msg, sizeof(msg), "the resident round kernel left before the phase was over (%s, round %d of %d, status %#x, waiting for %#x, mailbox out %#x in %#x, stream %s)",
                              ctx->live_mid ? "k_mid" : "k_tail", k, ctx->live_count, (unsigned) o->status, (unsigned) want, (unsigned) o->live.c[0][3], (unsigned) mi->c[0][3],
                              hipStreamQuery(ctx->stream) == hipSuccess ? "idle" : "busy");
                ctx->err = msg;
                ctx->live_mid = false;
                return ZK_ERR_STATE;
            }
            if (spins > (1ull << 22)) sched_yield();
        } else __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    std::memcpy(out_abc, w, 96);
    return ZK_OK;
}
static int32_t live_start(zk_ctx *ctx, const HFr &r, bool with_add_term) {
    tail_args A;
    std::memset(&A, 0, sizeof(A));
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        A.Vin[b] = vin(t); A.Min[b] = t.M[t.cur];
        A.Vbuf[b][0] = t.V[0]; A.Vbuf[b][1] = t.V[1];
        A.Mbuf[b][0] = t.M[0]; A.Mbuf[b][1] = t.M[1];
        A.out_idx[b] = t.cur ^ 1;
        A.n[b] = t.len;
    }
    A.first = ctx->round == 0 ? 1 : 0;
    A.rounds = ctx->phase_rounds - ctx->round;
    A.with_add_term = with_add_term ? 1 : 0;
    A.prev_r = to_dev(r);
    A.add_term = to_dev(ctx->add_term);
    A.out = (tail_out *) ctx->d_tail;
    A.in = (const live_in *) ctx->d_live_in;
    // sequence numbers of this kernel's rounds: never 0 (a cleared mailbox), never the abort word, never one of the previous kernel's
    ctx->live_seq32 += 64;
    if (ctx->live_seq32 > 0xf0000000u) ctx->live_seq32 = 64;
    A.seq32 = ctx->live_seq32;
    ((tail_out *) ctx->h_tail)->status = 0;
    ZK_LAUNCH(PC_TAIL, 0.0, k_tail<true>, dim3(1), dim3(TAIL_THREADS), A);
    ZK_HIP(hipGetLastError());
    ctx->live_active = true;
    ctx->live_count = A.rounds;
    ctx->live_cursor = 0;
    ctx->last_poly_valid = false;          // (these rounds do not maintain the running claim)
    ctx->live_rounds_total += (uint64_t) A.rounds;
    ++ctx->live_phases_total;
    return ZK_OK;
}
// a segment of mid-size rounds (k_mid): `rounds` consecutive rounds in which every present pair keeps at least two quads
static int32_t mid_start(zk_ctx *ctx, const HFr &r, bool with_add_term, int rounds, uint32_t blocks) {
    mid_args A;
    std::memset(&A, 0, sizeof(A));
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        A.Vin[b] = vin(t); A.Min[b] = t.M[t.cur];
        A.Vbuf[b][0] = t.V[0]; A.Vbuf[b][1] = t.V[1];
        A.Mbuf[b][0] = t.M[0]; A.Mbuf[b][1] = t.M[1];
        A.out_idx[b] = t.cur ^ 1;
        A.n[b] = t.len;
    }
    A.rounds = rounds;
    A.with_add_term = with_add_term ? 1 : 0;
    A.first = ctx->round == 0 ? 1 : 0;
    A.prev_r = to_dev(r);
    A.add_term = to_dev(ctx->add_term);
    A.partials = ctx->partials;
    A.arrive = ctx->d_counter + 2;
    A.bc = (mid_bcast *) ctx->d_bcast;
    A.out = (tail_out *) ctx->d_tail;
    A.in = (const live_in *) ctx->d_live_in;
    ctx->live_seq32 += 64;
    if (ctx->live_seq32 > 0xf0000000u) ctx->live_seq32 = 64;
    A.seq32 = ctx->live_seq32;
    ((tail_out *) ctx->h_tail)->status = 0;
    ZK_LAUNCH(PC_TAIL, 0.0, k_mid<true>, dim3(blocks), dim3(ZK_BLOCK), A);
    ZK_HIP(hipGetLastError());
    if (with_add_term) ctx->add_term = ctx->add_term * (HFr::one() - r);      // (the kernel's blocks apply the same factors to their copy)
    ctx->live_active = true;
    ctx->live_mid = true;
    ctx->live_first = A.first != 0;
    ctx->live_with_add = with_add_term;
    ctx->live_count = rounds;
    ctx->live_cursor = 0;
    ctx->last_poly_valid = false;
    ctx->live_rounds_total += (uint64_t) rounds;
    ++ctx->live_phases_total;
    return ZK_OK;
}
// the same segment in the non-interactive mode (k_mid<false>): the kernel derives the challenges itself, the host waits once and answers
// the verifier's following calls from the record (like run_device_rounds)
static int32_t run_device_mid(zk_ctx *ctx, const HFr &r, bool with_add_term, int rounds, uint32_t blocks) {
    mid_args A;
    std::memset(&A, 0, sizeof(A));
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        A.Vin[b] = vin(t); A.Min[b] = t.M[t.cur];
        A.Vbuf[b][0] = t.V[0]; A.Vbuf[b][1] = t.V[1];
        A.Mbuf[b][0] = t.M[0]; A.Mbuf[b][1] = t.M[1];
        A.out_idx[b] = t.cur ^ 1;
        A.n[b] = t.len;
    }
    const bool first = ctx->round == 0;
    A.rounds = rounds;
    A.with_add_term = with_add_term ? 1 : 0;
    A.first = first ? 1 : 0;
    A.prev_r = to_dev(r);
    A.add_term = to_dev(ctx->add_term);
    A.partials = ctx->partials;
    A.arrive = ctx->d_counter + 2;
    A.bc = (mid_bcast *) ctx->d_bcast;
    A.out = (tail_out *) ctx->d_tail;
    ctx->live_seq32 += 64;
    if (ctx->live_seq32 > 0xf0000000u) ctx->live_seq32 = 64;
    A.seq32 = ctx->live_seq32;
    std::memcpy(A.fs_state, ctx->fs_state, 32);
    const unsigned long long seq = ++ctx->tail_seq;
    A.seq = seq;
    ZK_LAUNCH(PC_TAIL, 0.0, k_mid<false>, dim3(blocks), dim3(ZK_BLOCK), A);
    ZK_HIP(hipGetLastError());
    volatile unsigned long long *p = &((tail_out *) ctx->h_tail)->seq;
    for (uint64_t spins = 0; *p != seq; ++spins) {
        if (spins > (1ull << 24)) {
            ZK_HIP(hipStreamSynchronize(ctx->stream));
            if (*p != seq) { ctx->err = "device rounds were not published"; return ZK_ERR_STATE; }
            break;
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const tail_out *o = (const tail_out *) ctx->h_tail;
    // what the kernel's blocks did to their copies: add_term (1 - r) per round, one fold per round but the phase's first
    if (with_add_term) {
        ctx->add_term = ctx->add_term * (HFr::one() - r);
        for (int k = 1; k < rounds; ++k) { HFr c; std::memcpy(&c, &o->chal[k - 1], 32); ctx->add_term = ctx->add_term * (HFr::one() - c); }
    }
    const int folds = rounds - (first ? 1 : 0);
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len || folds <= 0) continue;
        t.Vsrc = nullptr;
        if (folds & 1) t.cur ^= 1;
        t.len >>= folds;
        t.live = t.len;
    }
    ctx->tail_active = true;
    ctx->tail_count = rounds;
    ctx->tail_cursor = 0;
    ctx->last_poly_valid = false;
    ctx->tail_rounds_total += (uint64_t) rounds;
    ++ctx->tail_phases_total;
    return ZK_OK;
}
// one round of the resident kernel: r is the verifier's challenge for the previous polynomial (round 0 of the kernel got it as a launch argument)
static int32_t live_round(zk_ctx *ctx, const HFr &r, uint64_t out_abc[12]) {
    static const bool timing = getenv("ZKCNN_TIMING") != nullptr;
    const int k = ctx->live_cursor;
    const double t0 = timing ? now_s() : 0;
    if (timing && k > 0) ctx->live_t_host += t0 - ctx->live_t_exit;
    if (k > 0) {
        live_post(ctx, r, ctx->live_seq32 + (uint32_t) k);
        if (ctx->live_mid && ctx->live_with_add) ctx->add_term = ctx->add_term * (HFr::one() - r);
    }
    int32_t rc = live_wait(ctx, k, out_abc);
    if (rc) return rc;
    if (timing) {
        ctx->live_t_exit = now_s();
        if (k > 0) { ctx->live_t_gpu += ctx->live_t_exit - t0; ++ctx->live_timed_rounds; }
    }
    ++ctx->round;
    ctx->proof_size += 32 * 3;
    if (++ctx->live_cursor < ctx->live_count) return ZK_OK;
    if (ctx->live_mid) {
        // the segment is over: every table was folded live_count times (nothing collapsed, nothing is down to its last pair)
        for (int b = 0; b < 2; ++b) {
            table_pair &t = ctx->tp[b];
            if (!t.len) continue;
            const int folds = ctx->live_count - (ctx->live_first ? 1 : 0);
            if (folds > 0) {
                t.Vsrc = nullptr;
                if (folds & 1) t.cur ^= 1;
                t.len >>= folds;
                t.live = t.len;
            }
        }
        ctx->live_active = false;
        ctx->live_mid = false;
        return ZK_OK;
    }
    // the phase is over: the kernel has posted the bookkeeping scalar and what is left of the tables before its last polynomial, and leaves
    const tail_out *o = (const tail_out *) ctx->h_tail;
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        t.Vsrc = nullptr;
        if (o->pair_state[b] == 1) {
            t.len = 2;
            std::memcpy(&t.tail_v[0], &o->tail_v[b][0], 32);
            std::memcpy(&t.tail_v[1], &o->tail_v[b][1], 32);
            t.tail_valid = true;
        } else {
            t.len = 0;
            t.absorbed = true;
            std::memcpy(&t.final_v, &o->final_v[b], 32);
        }
    }
    std::memcpy(&ctx->add_term, &o->add_term, 32);
    ctx->live_ticks_wait += o->ticks_wait;
    ctx->live_ticks_total += o->ticks_total;
    ctx->live_active = false;
    return ZK_OK;
}

// Hybrid tail: the (small) live tables come to the host ...
static int32_t host_tail_begin(zk_ctx *ctx) {
    export_args A;
    std::memset(&A, 0, sizeof(A));
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        A.V[b] = t.len ? vin(t) : nullptr;
        A.M[b] = t.len ? t.M[t.cur] : nullptr;
        A.n[b] = (uint32_t) t.len;
    }
    A.out = (export_out *) ctx->d_tail;
    A.seq = ++ctx->tail_seq;
    ZK_LAUNCH(PC_FOLD, 0.0, k_export_tables, dim3(1), dim3(512), A);
    ZK_HIP(hipGetLastError());
    const export_out *o = (const export_out *) ctx->h_tail;
    volatile const unsigned long long *p = &o->seq;
    for (uint64_t spins = 0; *p != A.seq; ++spins) {
        if (spins > (1ull << 24)) {
            ZK_HIP(hipStreamSynchronize(ctx->stream));
            if (*p != A.seq) { ctx->err = "tables were not published"; return ZK_ERR_STATE; }
            break;
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int b = 0; b < 2; ++b) {
        const uint64_t n = ctx->tp[b].len;
        ctx->ht_V[b].resize(n);
        ctx->ht_M[b].resize(n);
        if (n) {
            std::memcpy(ctx->ht_V[b].data(), o->V[b], n * 32);
            std::memcpy(ctx->ht_M[b].data(), o->M[b], n * 32);
        }
    }
    ctx->host_tail_active = true;
    return ZK_OK;
}
// ... and one round there: the arithmetic of reference src/prover.cpp:368-426 on plain folded tables (what the round kernels compute)
static void host_tail_round(zk_ctx *ctx, const HFr &r, bool with_add_term, uint64_t out_abc[12]) {
    const bool first = ctx->round == 0;
    ++ctx->round;
    if (with_add_term) ctx->add_term = ctx->add_term * (HFr::one() - r);
    HFr a(0LL), c(0LL), p1(0LL);
    for (int b = 0; b < 2; ++b) {
        table_pair &t = ctx->tp[b];
        if (!t.len) continue;
        std::vector<HFr> &V = ctx->ht_V[b], &M = ctx->ht_M[b];
        if (!first) {
            for (uint64_t j = 0; j < t.len / 2; ++j) {
                V[j] = V[2 * j] + r * (V[2 * j + 1] - V[2 * j]);
                M[j] = M[2 * j] + r * (M[2 * j + 1] - M[2 * j]);
            }
            t.len >>= 1;
            t.Vsrc = nullptr;
        }
        if (t.len == 1) {                                   // the reference's `total == 1` case (prover.cpp:400-404)
            t.final_v = V[0];
            ctx->add_term = ctx->add_term + V[0] * M[0];
            t.absorbed = true;
            t.len = 0;
            continue;
        }
        for (uint64_t j = 0; j < t.len / 2; ++j) {
            const HFr &v0 = V[2 * j], &v1 = V[2 * j + 1], &m0 = M[2 * j], &m1 = M[2 * j + 1];
            a = a + (v1 - v0) * (m1 - m0);
            c = c + v0 * m0;
            p1 = p1 + v1 * m1;
        }
        if (t.len == 2) { t.tail_v[0] = V[0]; t.tail_v[1] = V[1]; t.tail_valid = true; }
    }
    HFr bcoef = p1 - a - c;
    if (with_add_term) { bcoef = bcoef - ctx->add_term; c = c + ctx->add_term; }
    put(out_abc, a);
    put(out_abc + 4, bcoef);
    put(out_abc + 8, c);
    ctx->proof_size += 32 * 3;
    ++ctx->host_tail_rounds_total;
}

static int32_t quad_round(zk_ctx *ctx, const HFr &r, bool with_add_term, uint64_t out_abc[12]) {
    static const bool seg_timing = getenv("ZKCNN_TIMING") != nullptr;
    // the tail paths start from add_term: it must be there before they are considered (the plain round below picks it up behind its launch)
    if (ctx->add_pending && (ctx->host_tail_log >= 0 || ctx->fs_state)) { int32_t rc0 = resolve_add_term(ctx); if (rc0) return rc0; }
    if (ctx->host_tail_log >= 0 && !ctx->host_tail_active && !ctx->tail_active && ctx->phase_rounds > ctx->round &&
        std::max(ctx->tp[0].len, ctx->tp[1].len) <= (1ull << ctx->host_tail_log) && ctx->tp[0].len + ctx->tp[1].len > 0) {
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
    const bool resident_ok = ctx->live_rounds && ctx->live_now && ctx->host_tail_log < 0 && in_phase;
    // The middle of a phase (more quads than the single-workgroup kernel takes, tables of at most 2^18 entries): a SEGMENT of rounds in one
    // resident multi-workgroup kernel (k_mid), as long as no table collapses or reaches its last pair. Returns the segment's length (0: none).
    auto plan_segment = [&]() -> int {
        if (!resident_ok || round_quads <= TAIL_QUADS || round_quads > 65536 || std::max(ctx->tp[0].len, ctx->tp[1].len) > (1ull << ZK_FULL_TABLE_LOG)) return 0;
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
            if (const int rounds = plan_segment()) rc = run_device_mid(ctx, r, with_add_term, rounds, (uint32_t) ((round_quads + 63) / 64));
            else if (round_quads <= TAIL_QUADS) rc = run_device_rounds(ctx, r, with_add_term);
            if (rc) return rc;
        }
    } else if (resident_ok) {
        // interactive protocol: the same two kernels, trading polynomials and challenges with the verifier through mailboxes
        int32_t rc = ZK_OK;
        if (const int rounds = plan_segment()) {
            rc = resolve_add_term(ctx);
            if (!rc) rc = mid_start(ctx, r, with_add_term, rounds, (uint32_t) ((round_quads + 63) / 64));
        } else if (round_quads <= TAIL_QUADS) {
            rc = resolve_add_term(ctx);
            if (!rc) rc = live_start(ctx, r, with_add_term);
        }
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
    const int fine_log = 16;
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
        // (k_round_quad2 holds 3 waves per SIMD: 768 blocks of 4 waves are exactly one resident set of the 1 024 SIMDs -- no partial second wave of blocks)
        const uint32_t quad_cap = 768;
        A.blocks[b] = collapsed[b] ? 1 : std::min<uint32_t>(grid_for(work, fine ? 1024 : quad_cap), ctx->partial_blocks / 2);
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
        if (fine) {
            const uint32_t blocks = (uint32_t) ((fine_items + ZK_BLOCK / 4 - 1) / (ZK_BLOCK / 4));
            ZK_LAUNCH(PC_ROUND_QUAD, alg_bytes, k_round_quad_fine, dim3(blocks), dim3(ZK_BLOCK), A);
        } else ZK_LAUNCH(PC_ROUND_QUAD, alg_bytes, k_round_quad2, dim3(A.blocks[0] + A.blocks[1]), dim3(ZK_BLOCK), A);
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
        ZK_LAUNCH(PC_FOLD, 0.0, k_eval_pairs, dim3(1), dim3(64), E);
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
        ZK_LAUNCH(PC_EQ, 0.0, k_eq_halves_multi, dim3(2, ctx->liu_ntabs), dim3(1024), ctx->liu_halves, (const liu_table *) ctx->liu_tabs);
    ZK_LAUNCH(PC_LIU, 0.0, k_liu_gather, dim3(grid_for(t.len)), dim3(ZK_BLOCK), t.M[0], ctx->liu_ptr, (const liu_entry *) ctx->liu_ent, ctx->liu_halves, t.len);
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

// ------------------------------------------------------------------------------------------------
// kernel-level entry points (host arrays in / out)
// ------------------------------------------------------------------------------------------------
#define CHECK_CTX() do { if (!ctx) return ZK_ERR_ARG; ZK_HIP(hipSetDevice(ctx->device)); if (ctx->live_active) { int32_t rc_ = zk_live_abort(ctx); if (rc_) return rc_; } } while (0)

template <int OP>
static int32_t binop(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) {
    CHECK_CTX();
    int32_t rc = zk_scratch(ctx, 3 * n * 32);
    if (rc) return rc;
    fr_t *da = (fr_t *) ctx->scratch.p, *db = da + n, *dz = db + n;
    ZK_HIP(hipMemcpyAsync(da, a, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(db, b, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_LAUNCH(PC_MISC, 0.0, k_fr_binop<OP>, dim3(grid_for(n)), dim3(ZK_BLOCK), dz, da, db, n);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(out, dz, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
extern "C" int32_t zk_k_fr_mul(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) { return binop<0>(ctx, out, a, b, n); }
extern "C" int32_t zk_k_fr_add(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) { return binop<1>(ctx, out, a, b, n); }
extern "C" int32_t zk_k_fr_sub(zk_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) { return binop<2>(ctx, out, a, b, n); }

extern "C" int32_t zk_k_eq_table(zk_ctx *ctx, uint64_t *out, int32_t n, const uint64_t *r0, const uint64_t *r1,
                                 const uint64_t alpha[4], const uint64_t beta[4]) {
    CHECK_CTX();
    if (n < 0 || n > ZK_MAX_VARS) return ZK_ERR_ARG;
    const uint64_t len = 1ull << n;
    int32_t rc = zk_scratch(ctx, len * 32);
    if (rc) return rc;
    rc = eq_table(ctx, (fr_t *) ctx->scratch.p, n, reinterpret_cast<const HFr *>(r0), H(alpha), reinterpret_cast<const HFr *>(r1),
                  H(beta), ~0ull, HFr::one());
    if (rc) return rc;
    ZK_HIP(hipMemcpyAsync(out, ctx->scratch.p, len * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

extern "C" int32_t zk_k_phi_table(zk_ctx *ctx, uint64_t *out, const uint64_t *rx, const uint64_t scale[4], int32_t n,
                                  int32_t inverse) {
    CHECK_CTX();
    if (n < 1 || n > 20) return ZK_ERR_ARG;
    const uint64_t cnt = inverse ? 1ull << n : 1ull << (n - 1);
    int32_t rc = zk_scratch(ctx, cnt * 32);
    if (rc) return rc;
    if ((rc = phi_table(ctx, (fr_t *) ctx->scratch.p, reinterpret_cast<const HFr *>(rx), H(scale), n, inverse != 0))) return rc;
    ZK_HIP(hipMemcpyAsync(out, ctx->scratch.p, cnt * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

extern "C" int32_t zk_k_round_quadratic(zk_ctx *ctx, uint64_t *V, uint64_t *M, uint64_t n, const uint64_t r[4], int32_t first,
                                        uint64_t out_abc[12], uint64_t *n_out) {
    CHECK_CTX();
    if (n < 2 || (n & (n - 1)) || (!first && n < 4)) return ZK_ERR_ARG;
    int32_t rc = zk_scratch(ctx, 4 * n * 32);
    if (rc) return rc;
    fr_t *dV = (fr_t *) ctx->scratch.p, *dM = dV + n, *dV2 = dM + n, *dM2 = dV2 + n;
    ZK_HIP(hipMemcpyAsync(dV, V, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(dM, M, n * 32, hipMemcpyHostToDevice, ctx->stream));
    // the PRODUCT kernel (k_round_quad2 as quad_round launches it for tables above the fine-grained limit), one table pair
    const uint64_t npairs = first ? n / 2 : n / 4;
    round2_args A;
    std::memset(&A, 0, sizeof(A));
    A.Vin[0] = dV; A.Min[0] = dM; A.Vout[0] = dV2; A.Mout[0] = dM2;
    A.n[0] = n;
    A.nl[0] = n;
    A.blocks[0] = std::min<uint32_t>(grid_for(npairs, 1024), ctx->partial_blocks / 2);
    A.r = to_dev(H(r));
    A.first = first ? 1 : 0;
    A.partials = ctx->partials;
    A.counter = ctx->d_counter;
    A.slot = (host_slot *) ctx->d_slot;
    A.seq = ++ctx->slot_seq;
    ZK_LAUNCH(PC_ROUND_QUAD, 0.0, k_round_quad2, dim3(A.blocks[0]), dim3(ZK_BLOCK), A);
    ZK_HIP(hipGetLastError());
    if ((rc = wait_slot(ctx, A.seq))) return rc;
    for (int k = 0; k < 3; ++k) ctx->h_result[k] = ctx->h_slot->v[k];
    const uint64_t nn = first ? n : n / 2;
    if (!first) {
        ZK_HIP(hipMemcpy(V, dV2, nn * 32, hipMemcpyDeviceToHost));
        ZK_HIP(hipMemcpy(M, dM2, nn * 32, hipMemcpyDeviceToHost));
    }
    HFr a = ctx->h_result[0], c = ctx->h_result[1], p1 = ctx->h_result[2];
    put(out_abc, a);
    put(out_abc + 4, p1 - a - c);
    put(out_abc + 8, c);
    *n_out = nn;
    return ZK_OK;
}

// ---- witness kernels of the FFT convolution (host arrays in / out) ----
extern "C" int32_t zk_witness_ntt(zk_ctx *ctx, uint64_t *dst, const uint64_t *src, int32_t logn, int32_t inverse, uint64_t count) {
    CHECK_CTX();
    if (logn < 1 || logn > 12 || !count) return ZK_ERR_ARG;        // 2^12 x 32 B = 128 KiB is what fits in LDS
    const uint32_t len = 1u << logn, in_len = inverse ? len : len / 2, out_len = inverse ? len / 2 : len;
    fr_t *pw = powers_of_root(ctx, logn, inverse != 0);
    if (!pw) { ctx->err = "root table allocation failed"; return ZK_ERR_NOMEM; }
    const size_t in_bytes = (size_t) count * in_len * 32, out_bytes = (size_t) count * out_len * 32;
    int32_t rc = zk_scratch(ctx, in_bytes + out_bytes);
    if (rc) return rc;
    fr_t *d_in = (fr_t *) ctx->scratch.p, *d_out = d_in + (size_t) count * in_len;
    ZK_HIP(hipMemcpyAsync(d_in, src, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    HFr ilen;
    HFr::inv(ilen, HFr((unsigned long long) len));
    const size_t lds = (size_t) len * 32;
    static bool attr_set = false;
    if (!attr_set) {
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_batch), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    const uint32_t threads = std::min<uint32_t>(1024, std::max<uint32_t>(64, len / 2));
    for (uint64_t b0 = 0; b0 < count; b0 += 1u << 30) {
        const uint32_t nb = (uint32_t) std::min<uint64_t>(1u << 30, count - b0);
        prof_begin(ctx, PC_MISC, 64.0 * len * nb);
        hipLaunchKernelGGL(k_ntt_batch, dim3(nb), dim3(threads), lds, ctx->stream, d_out + b0 * out_len, d_in + b0 * in_len, pw, logn, in_len,
                           out_len, to_dev(ilen), inverse ? 1 : 0);
        prof_end(ctx, PC_MISC);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(dst, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

extern "C" int32_t zk_witness_dotprod(zk_ctx *ctx, uint64_t *out, uint64_t n_out, const uint64_t *F, uint64_t n_in,
                                      const zk_bin_gate *gates, uint64_t n_gates, int32_t fft_bl) {
    CHECK_CTX();
    if (fft_bl < 1 || fft_bl > 20 || !n_out || !n_in) return ZK_ERR_ARG;
    std::vector<gate_rec> recs(n_gates);
    for (uint64_t k = 0; k < n_gates; ++k) {
        if (gates[k].g >= n_out || gates[k].u >= n_in || gates[k].v >= n_in) { ctx->err = "dot-prod gate out of range"; return ZK_ERR_ARG; }
        gate_rec r = {gates[k].g, gates[k].u, gates[k].v, 0};      // key = u, aux = v for this kernel
        recs[k] = r;
    }
    // CSR by output vector g (stable: order inside a row does not matter, the sum is exact)
    std::vector<uint32_t> ptr(n_out + 1, 0);
    for (const gate_rec &r : recs) ++ptr[r.g + 1];
    for (uint64_t g = 0; g < n_out; ++g) ptr[g + 1] += ptr[g];
    std::vector<gate_rec> sorted(n_gates);
    {
        std::vector<uint32_t> pos(ptr.begin(), ptr.end() - 1);
        for (const gate_rec &r : recs) sorted[pos[r.g]++] = r;
    }
    const size_t len = (size_t) 1 << fft_bl;
    const size_t bytes = (n_in + n_out) * len * 32 + n_gates * sizeof(gate_rec) + (n_out + 1) * 4 + 64;
    int32_t rc = zk_scratch(ctx, bytes);
    if (rc) return rc;
    fr_t *dF = (fr_t *) ctx->scratch.p, *dO = dF + n_in * len;
    gate_rec *dR = (gate_rec *) (dO + n_out * len);
    uint32_t *dP = (uint32_t *) (dR + n_gates);
    ZK_HIP(hipMemcpyAsync(dF, F, n_in * len * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(dR, sorted.data(), n_gates * sizeof(gate_rec), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(dP, ptr.data(), (n_out + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    for (uint64_t g0 = 0; g0 < n_out; g0 += 32768) {
        const uint32_t ng = (uint32_t) std::min<uint64_t>(32768, n_out - g0);
        dim3 grid((uint32_t) ((len + ZK_BLOCK - 1) / ZK_BLOCK), ng);
        ZK_LAUNCH(PC_DOT, 0.0, k_dot_witness, grid, dim3(ZK_BLOCK), dO + g0 * len, dF, dR, dP + g0, fft_bl);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(out, dO, n_out * len * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------
// verifier side: wiring predicates over the resident gate lists (reference src/verifier.cpp:36-116, 304-325)
// ------------------------------------------------------------------------------------------------
static int32_t verifier_buffers(zk_ctx *ctx) {
    if (ctx->v_bg) return ZK_OK;
    int32_t rc;
    const uint64_t cap_uv = std::max(std::max(ctx->beta_u_cap, ctx->beta_g_cap), ctx->sz.sub);
    const uint64_t cap_g = std::max<uint64_t>(ctx->beta_g_cap, ctx->L[0].val_len);      // the layer-0 check needs eq over all of layer 0
    if ((rc = zk_dev_alloc(ctx, (void **) &ctx->v_bg, cap_g * 32)) || (rc = zk_dev_alloc(ctx, (void **) &ctx->v_bu, cap_uv * 32)) ||
        (rc = zk_dev_alloc(ctx, (void **) &ctx->v_bv, cap_uv * 32)) || (rc = zk_dev_alloc(ctx, (void **) &ctx->v_gs, ctx->beta_gs_cap * 32)))
        return rc;
    return ZK_OK;
}

extern "C" int32_t zk_verifier_predicates(zk_ctx *ctx, int32_t layer, const uint64_t *r_0, const uint64_t *r_1, const uint64_t alpha[4],
                                          const uint64_t beta[4], const uint64_t relu_rou[4], const uint64_t *r_u, const uint64_t *r_v,
                                          const uint64_t *r_u2, const uint64_t *r_v2, uint64_t uni[8], uint64_t bin[12]) {
    CHECK_READY();
    if (layer < 1 || layer >= (int) ctx->L.size()) return ZK_ERR_ARG;
    int32_t rc;
    if ((rc = verifier_buffers(ctx))) return rc;
    const dev_layer &cur = ctx->L[layer];
    const zk_layer_desc &d = cur.d;
    const HFr *R0 = reinterpret_cast<const HFr *>(r_0), *R1 = reinterpret_cast<const HFr *>(r_1), *RU = reinterpret_cast<const HFr *>(r_u),
              *RV = reinterpret_cast<const HFr *>(r_v), *RU2 = reinterpret_cast<const HFr *>(r_u2), *RV2 = reinterpret_cast<const HFr *>(r_v2);
    const HFr al = H(alpha), be = H(beta), scale = H(d.scale);
    const int bl = d.bit_length, fft_bl = d.fft_bit_length, fft_blh = fft_bl - 1;
    fr_t *res = ctx->d_result + 16;                       // [0,1] uni, [2,3] bin list 0, [4,5] bin list 1, [6] table dot
    ZK_HIP(hipMemsetAsync(res, 0, 8 * 32, ctx->stream));
    const bool xf = d.ty == ZK_FFT || d.ty == ZK_IFFT, dot = d.ty == ZK_DOT_PROD;
    if (xf) {
        if ((rc = phi_table(ctx, ctx->v_gs, R0, scale, fft_bl, d.ty == ZK_IFFT))) return rc;
        if ((rc = eq_table1(ctx, ctx->v_bu, d.max_bl_u, RU, HFr::one()))) return rc;
        const uint64_t n = 1ull << d.max_bl_u;
        const uint32_t g = std::min<uint32_t>(grid_for(n, 1024), ctx->partial_blocks);
        ZK_LAUNCH(PC_GATE_SUM, 0.0, k_dot_indexed, dim3(g), dim3(ZK_BLOCK), ctx->partials, ctx->v_gs, (const uint32_t *) nullptr, ctx->v_bu, n);
        ZK_LAUNCH(PC_SUM, 0.0, k_sum_partials<1>, dim3(1), dim3(ZK_BLOCK), res + 6, ctx->partials, g, 0);
    } else if (d.ty == ZK_PADDING) {
        if (!RU2) return ZK_ERR_ARG;
        if ((rc = eq_table(ctx, ctx->v_bv, bl - fft_blh, RU2 + fft_bl, al, RV2, be, ~0ull, HFr::one()))) return rc;
        if ((rc = eq_table1(ctx, ctx->v_gs, fft_blh, R0, HFr::one()))) return rc;
        ZK_LAUNCH(PC_EQ, 0.0, k_outer_expand, dim3(grid_for(cur.val_len)), dim3(ZK_BLOCK), ctx->v_bg, ctx->v_bv, ctx->v_gs, fft_blh, cur.val_len);
        if ((rc = eq_table1(ctx, ctx->v_bu, d.max_bl_u, RU, HFr::one()))) return rc;
    } else if (dot) {
        if (!RU2) return ZK_ERR_ARG;
        const int cnt_bl = bl - fft_bl, cnt_bl2 = d.max_bl_u - fft_bl;
        if ((rc = eq_table1(ctx, ctx->v_bg, cnt_bl, RU2 + fft_bl - 1, al))) return rc;
        HFr same = HFr::one();                            // eq(r_0, r_u) on the frequency bits
        for (int j = 0; j < fft_bl; ++j) same = same * (R0[j] * RU[j] + (HFr::one() - R0[j]) * (HFr::one() - RU[j]));
        if ((rc = eq_table1(ctx, ctx->v_bu, cnt_bl2, RU + fft_bl, same))) return rc;
    } else {
        const bool tail = d.zero_start_id < d.size;
        if ((rc = eq_table(ctx, ctx->v_bg, bl, R0, al * scale, R1, be * scale, tail ? d.zero_start_id : ~0ull, tail ? H(relu_rou) : HFr::one())))
            return rc;
        if ((rc = eq_table1(ctx, ctx->v_bu, d.max_bl_u, RU, HFr::one()))) return rc;
    }
    if (!xf && cur.n_uni2) {
        gate_args A;
        std::memset(&A, 0, sizeof(A));
        A.recs = cur.uni2; A.n = cur.n_uni2;
        A.beta_g = ctx->v_bg; A.beta_u = ctx->v_bu; A.two_mul = ctx->two_mul; A.phase = 2;
        const uint32_t g = std::min<uint32_t>(grid_for(cur.n_uni2, 1024), ctx->partial_blocks);
        ZK_LAUNCH(PC_GATE_SUM, 0.0, k_gate_sum2, dim3(g), dim3(ZK_BLOCK), ctx->partials, A);
        ZK_LAUNCH(PC_SUM, 0.0, k_sum_partials<2>, dim3(1), dim3(ZK_BLOCK), res, ctx->partials, g, 0);
    }
    HFr bv0 = HFr::one();
    if (d.need_phase2) {
        if ((rc = eq_table1(ctx, ctx->v_bv, d.max_bl_v, RV, HFr::one()))) return rc;
        for (int j = 0; j < d.max_bl_v; ++j) bv0 = bv0 * (HFr::one() - RV[j]);          // beta_v[0]
        for (int b = 0; b < 2; ++b) {
            if (!cur.n_p2[b]) continue;
            const uint32_t g = std::min<uint32_t>(grid_for(cur.n_p2[b], 1024), ctx->partial_blocks);
            ZK_LAUNCH(PC_GATE_SUM, 0.0, k_pred_bin, dim3(g), dim3(ZK_BLOCK), ctx->partials, cur.p2[b], cur.n_p2[b], ctx->v_bg, ctx->v_bu,
                      ctx->v_bv, ctx->two_mul, dot ? 0 : 1);
            ZK_LAUNCH(PC_SUM, 0.0, k_sum_partials<2>, dim3(1), dim3(ZK_BLOCK), res + 2 + 2 * b, ctx->partials, g, 0);
        }
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(ctx->h_result + 16, res, 8 * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    const HFr *h = ctx->h_result + 16;
    HFr u0 = h[0], u1 = xf ? h[6] : h[1];
    if (d.need_phase2) { u0 = u0 * bv0; u1 = u1 * bv0; }
    put(uni, u0);
    put(uni + 4, u1);
    put(bin, h[2]);                 // l = 0: u and v in layer 0
    put(bin + 4, h[5]);             // l = 1: both in the previous layer
    put(bin + 8, h[3]);             // l = 2: u in the previous layer, v in layer 0
    if (!h[4].isZero()) { ctx->err = "bin gate with u in layer 0 and v in the previous layer"; return ZK_ERR_STATE; }
    return ZK_OK;
}

// g(r_u0) of the layer-0 check: sum over layers of sum_j eq(r_u0, ori_id[j]) * sig * eq(r_{u,v}[layer], j)  (reference src/verifier.cpp:304-325)
extern "C" int32_t zk_verifier_input_predicate(zk_ctx *ctx, const uint64_t *r_u0, const uint64_t *const *r_u, const uint64_t *const *r_v,
                                               const uint64_t *sig_u, const uint64_t *sig_v, uint32_t n, uint64_t out[4]) {
    CHECK_READY();
    if (n + 1 != ctx->L.size()) return ZK_ERR_ARG;
    int32_t rc;
    if ((rc = verifier_buffers(ctx))) return rc;
    fr_t *res = ctx->d_result + 16;
    ZK_HIP(hipMemsetAsync(res, 0, 32, ctx->stream));
    if ((rc = eq_table1(ctx, ctx->v_bg, ctx->L[0].d.bit_length, reinterpret_cast<const HFr *>(r_u0), HFr::one()))) return rc;
    for (size_t i = 1; i < ctx->L.size(); ++i) {
        const dev_layer &Li = ctx->L[i];
        for (int side = 0; side < 2; ++side) {
            const int bl = side ? Li.d.bit_length_v[0] : Li.d.bit_length_u[0];
            const uint32_t cnt = side ? Li.d.size_v[0] : Li.d.size_u[0];
            const uint64_t *pt = side ? r_v[i] : r_u[i];
            if (bl < 0 || !cnt) continue;
            if (!pt) return ZK_ERR_ARG;
            if ((rc = eq_table1(ctx, ctx->v_bu, bl, reinterpret_cast<const HFr *>(pt), H((side ? sig_v : sig_u) + 4 * (i - 1))))) return rc;
            const uint32_t g = std::min<uint32_t>(grid_for(cnt, 1024), ctx->partial_blocks);
            ZK_LAUNCH(PC_LIU, 0.0, k_dot_indexed, dim3(g), dim3(ZK_BLOCK), ctx->partials, ctx->v_bg, side ? Li.ori_v : Li.ori_u, ctx->v_bu, (uint64_t) cnt);
            ZK_LAUNCH(PC_SUM, 0.0, k_sum_partials<1>, dim3(1), dim3(ZK_BLOCK), res, ctx->partials, g, 1);
        }
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(ctx->h_result + 16, res, 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    put(out, ctx->h_result[16]);
    return ZK_OK;
}

// ---- witness of a generic layer: every gate evaluated on the GPU (reference src/neuralNetwork.cpp:918-935) ----
static int32_t grow_buf(zk_ctx *ctx, dev_buf &b, size_t bytes, size_t keep = 0) {
    if (b.bytes >= bytes) return ZK_OK;
    size_t want = std::max(bytes, b.bytes + b.bytes / 2);
    void *np = nullptr;
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    if (hipMalloc(&np, want) != hipSuccess) {
        (void) hipGetLastError();
        want = bytes;
        ZK_HIP(hipMalloc(&np, want));
    }
    if (keep && b.p) {
        ZK_HIP(hipMemcpyAsync(np, b.p, keep, hipMemcpyDeviceToDevice, ctx->stream));
        ZK_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (b.p) ZK_HIP(hipFree(b.p));
    b.p = np;
    b.bytes = want;
    return ZK_OK;
}

extern "C" int32_t zk_witness_release(zk_ctx *ctx) {
    CHECK_CTX();
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->w_val0.p) { ZK_HIP(hipFree(ctx->w_val0.p)); ctx->w_val0 = dev_buf(); }
    for (dev_buf &b : ctx->w_stage) if (b.p) { ZK_HIP(hipFree(b.p)); b = dev_buf(); }
    ctx->w_val0_len = 0;
    return ZK_OK;
}

extern "C" int32_t zk_witness_input(zk_ctx *ctx, uint64_t offset, const uint64_t *values, uint64_t n) {
    CHECK_CTX();
    if (offset == 0) ctx->w_val0_len = 0;                            // a new layer 0 starts
    if (offset > ctx->w_val0_len) return ZK_ERR_ARG;                 // the copy has no holes
    int32_t rc = grow_buf(ctx, ctx->w_val0, (offset + n) * 32, ctx->w_val0_len * 32);
    if (rc) return rc;
    if (n) ZK_HIP(hipMemcpyAsync((fr_t *) ctx->w_val0.p + offset, values, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ctx->w_val0_len = std::max<uint64_t>(ctx->w_val0_len, offset + n);
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// true when equal g are adjacent; range-checks the list on the way
template <class G, class Check>
static int grouped_by_output(const G *gates, uint64_t n, uint64_t n_out, std::vector<uint8_t> &seen, Check in_range) {
    std::fill(seen.begin(), seen.end(), 0);
    bool grouped = true;
    uint32_t prev = 0xffffffffu;
    for (uint64_t k = 0; k < n; ++k) {
        const G &gt = gates[k];
        if (gt.g >= n_out || !in_range(gt)) return -1;
        if (gt.g != prev) {
            if (seen[gt.g]) grouped = false;
            seen[gt.g] = 1;
            prev = gt.g;
        }
    }
    return grouped ? 1 : 0;
}
template <class G>
static void regroup(std::vector<G> &dst, const G *gates, uint64_t n, uint64_t n_out) {
    std::vector<uint64_t> pos(n_out + 1, 0);
    for (uint64_t k = 0; k < n; ++k) ++pos[gates[k].g + 1];
    for (uint64_t g = 0; g < n_out; ++g) pos[g + 1] += pos[g];
    dst.resize(n);
    for (uint64_t k = 0; k < n; ++k) dst[pos[gates[k].g]++] = gates[k];
}

extern "C" int32_t zk_witness_gates(zk_ctx *ctx, uint64_t *out, uint64_t n_out, const zk_uni_gate *uni, uint64_t n_uni,
                                    const zk_bin_gate *bin, uint64_t n_bin, const uint64_t *prev, uint64_t n_prev,
                                    const uint64_t *two_mul, uint32_t n_two_mul, const uint64_t scale[4]) {
    CHECK_CTX();
    if (!n_out || n_out > 0xfffffffeull || !n_two_mul) return ZK_ERR_ARG;
    static_assert(sizeof(uni_gate_dev) == sizeof(zk_uni_gate) && sizeof(bin_gate_dev) == sizeof(zk_bin_gate), "gate layout");
    const uint64_t n0 = ctx->w_val0_len;
    if (!prev) n_prev = n0;                                          // layer 1: the previous layer IS layer 0
    std::vector<uint8_t> seen(n_out);
    std::vector<zk_uni_gate> uni_sorted;
    std::vector<zk_bin_gate> bin_sorted;
    int g1 = grouped_by_output(uni, n_uni, n_out, seen, [&](const zk_uni_gate &gt) {
        return gt.sc < n_two_mul && gt.u < (gt.lu ? n_prev : n0); });
    int g2 = g1 < 0 ? -1 : grouped_by_output(bin, n_bin, n_out, seen, [&](const zk_bin_gate &gt) {
        return gt.sc < n_two_mul && gt.l <= 2 && gt.u < (gt.l == 0 ? n0 : n_prev) && gt.v < ((gt.l & 1) ? n_prev : n0); });
    if (g1 < 0 || g2 < 0) { ctx->err = "witness gate operand out of range"; return ZK_ERR_ARG; }
    if (!g1) { regroup(uni_sorted, uni, n_uni, n_out); uni = uni_sorted.data(); }
    if (!g2) { regroup(bin_sorted, bin, n_bin, n_out); bin = bin_sorted.data(); }

    const uint64_t blocks_u = (n_uni + ZK_BLOCK - 1) / ZK_BLOCK, blocks_b = (n_bin + ZK_BLOCK - 1) / ZK_BLOCK;
    const uint64_t slots = 2 * std::max<uint64_t>(std::max(blocks_u, blocks_b), 1);
    int32_t rc;
    dev_buf &bG = ctx->w_stage[0], &bP = ctx->w_stage[1], &bO = ctx->w_stage[2], &bC = ctx->w_stage[3], &bT = ctx->w_stage[4];
    if ((rc = grow_buf(ctx, bG, std::max<size_t>(n_uni * sizeof(zk_uni_gate), n_bin * sizeof(zk_bin_gate)) + 16)) ||
        (rc = grow_buf(ctx, bP, (prev ? std::max<uint64_t>(n_prev, 1) : 1) * 32)) || (rc = grow_buf(ctx, bO, 3 * n_out * 32)) ||
        (rc = grow_buf(ctx, bC, slots * (32 + 4))) || (rc = grow_buf(ctx, bT, (size_t) n_two_mul * 32)))
        return rc;
    fr_t *dA = (fr_t *) bO.p, *dB = dA + n_out, *dO = dB + n_out;
    fr_t *carry_val = (fr_t *) bC.p;
    uint32_t *carry_key = (uint32_t *) (carry_val + slots);
    const fr_t *v0 = (const fr_t *) ctx->w_val0.p, *vp = prev ? (const fr_t *) bP.p : v0, *tm = (const fr_t *) bT.p;
    if (prev && n_prev) ZK_HIP(hipMemcpyAsync(bP.p, prev, n_prev * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(bT.p, two_mul, (size_t) n_two_mul * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemsetAsync(dA, 0, 2 * n_out * 32, ctx->stream));
    if (n_uni) {
        ZK_HIP(hipMemcpyAsync(bG.p, uni, n_uni * sizeof(zk_uni_gate), hipMemcpyHostToDevice, ctx->stream));
        ZK_LAUNCH(PC_GATE, 44.0 * (double) n_uni, k_eval_uni, dim3((uint32_t) blocks_u), dim3(ZK_BLOCK), dA, carry_key, carry_val,
                  (const uni_gate_dev *) bG.p, n_uni, v0, vp, tm);
        ZK_LAUNCH(PC_GATE_FIX, 0.0, k_gate_fixup, dim3(grid_for(2 * blocks_u)), dim3(ZK_BLOCK), dA, carry_key, carry_val, 2 * blocks_u);
    }
    if (n_bin) {
        ZK_HIP(hipMemcpyAsync(bG.p, bin, n_bin * sizeof(zk_bin_gate), hipMemcpyHostToDevice, ctx->stream));
        ZK_LAUNCH(PC_GATE, 80.0 * (double) n_bin, k_eval_bin, dim3((uint32_t) blocks_b), dim3(ZK_BLOCK), dB, carry_key, carry_val,
                  (const bin_gate_dev *) bG.p, n_bin, v0, vp, tm);
        ZK_LAUNCH(PC_GATE_FIX, 0.0, k_gate_fixup, dim3(grid_for(2 * blocks_b)), dim3(ZK_BLOCK), dB, carry_key, carry_val, 2 * blocks_b);
    }
    const HFr sc = H(scale);
    ZK_LAUNCH(PC_MISC, 0.0, k_eval_combine, dim3(grid_for(n_out)), dim3(ZK_BLOCK), dO, dA, dB, to_dev(sc), sc == HFr::one() ? 0 : 1, n_out);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(out, dO, n_out * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// channel_in chunks of k_conv_eval: about 2^20 threads per launch, so that layers with few outputs (2 x 2 pictures, 512 channels) still fill the GPU
static uint32_t conv_eval_chunks(const conv_desc &c, uint64_t n_out) {
    const uint64_t want = std::max<uint64_t>(1, ((1ull << 20) + n_out - 1) / n_out);
    const uint32_t per = (uint32_t) std::max<uint64_t>(1, (c.CI + want - 1) / want);
    return (c.CI + per - 1) / per;
}

// ------------------------------------------------------------------------------------------------
// resident witness program: the next picture without the host round trip of the layer values
// ------------------------------------------------------------------------------------------------
extern "C" int32_t zk_witness_program_upload(zk_ctx *ctx, const zk_witness_op *ops, uint64_t n_ops, const uint32_t *windows, uint64_t n_windows,
                                             const zk_witness_step *steps, uint32_t n_steps, const zk_layer_desc *layers, int32_t n_layers) {
    CHECK_READY();
    static_assert(sizeof(zk_witness_op) == sizeof(wit_op) && sizeof(zk_witness_op) == 12 && sizeof(zk_witness_step) == 48, "program records");
    if (ctx->wp_ready) { ctx->err = "witness program already uploaded"; return ZK_ERR_STATE; }
    if (n_layers != (int32_t) ctx->L.size() || !steps || !n_steps || (n_ops && !ops) || (n_windows && !windows)) return ZK_ERR_ARG;
    const uint64_t n0 = ctx->L[0].d.size;
    int32_t rc;
    // ---- the steps: every index an operation touches must exist ----
    uint32_t n_ranges = 0;
    std::vector<uint8_t> evaluated(n_layers, 0);
    evaluated[0] = 1;
    for (uint32_t k = 0; k < n_steps; ++k) {
        const zk_witness_step &st = steps[k];
        if (st.layer < 0 || st.layer >= n_layers) { ctx->err = "witness step: layer out of range"; return ZK_ERR_ARG; }
        if (st.what == 1) {
            if (st.layer < 1 || !evaluated[st.layer - 1]) { ctx->err = "witness step: layer evaluated before its inputs"; return ZK_ERR_ARG; }
            evaluated[st.layer] = 1;
        } else if (st.what == 2) {
            if (!evaluated[st.layer]) { ctx->err = "witness step: range of a layer not yet evaluated"; return ZK_ERR_ARG; }
            ++n_ranges;
        } else if (st.what == 0) {
            if (st.op_begin > st.op_end || st.op_end > n_ops || !evaluated[st.layer]) { ctx->err = "witness step: bad operation span"; return ZK_ERR_ARG; }
            const uint64_t src_n = ctx->L[st.layer].d.size;
            const bool sum = st.op_begin < st.op_end && ops[st.op_begin].op == 3;
            uint64_t n_win = 0;
            if (sum) {
                if (st.win < 1 || st.win_begin > n_windows) { ctx->err = "witness step: bad window table"; return ZK_ERR_ARG; }
                n_win = (n_windows - st.win_begin) / (uint64_t) st.win;
            }
            uint64_t used_win = 0;
            for (uint64_t j = st.op_begin; j < st.op_end; ++j) {
                const zk_witness_op &op = ops[j];
                if (op.op > 3 || op.dst >= n0 || op.shift > 63 || (op.op == 3) != sum || ((op.op == 2) != (ops[st.op_begin].op == 2)) ||
                    (sum ? op.src >= n_win : op.src >= src_n)) { ctx->err = "witness operation out of range"; return ZK_ERR_ARG; }
                if (sum) used_win = std::max<uint64_t>(used_win, (uint64_t) op.src + 1);
            }
            for (uint64_t w = st.win_begin; w < st.win_begin + used_win * (uint64_t) (sum ? st.win : 0); ++w)
                if (windows[w] >= src_n) { ctx->err = "witness window entry out of range"; return ZK_ERR_ARG; }
        } else {
            ctx->err = "witness step: unknown kind";
            return ZK_ERR_ARG;
        }
    }
    for (int i = 1; i < n_layers; ++i)
        if (!evaluated[i]) { ctx->err = "witness program does not evaluate every layer"; return ZK_ERR_ARG; }
    // ---- gate lists grouped by output, operands in layer 0 as raw layer-0 indices ----
    uint64_t max_out = 1, max_blocks = 1, max_conv_part = 0;
    for (int i = 1; i < n_layers; ++i) {
        const zk_layer_desc &S = layers[i];
        dev_layer &D = ctx->L[i];
        if (S.size != D.d.size || S.ty != D.d.ty) { ctx->err = "witness program: layer descriptors differ from the uploaded circuit"; return ZK_ERR_ARG; }
        const uint64_t n_out = S.size, n_prev = ctx->L[i - 1].d.size;
        if (S.ty == ZK_FFT || S.ty == ZK_IFFT) {
            if (S.fft_bit_length < 1 || S.fft_bit_length > 12) { ctx->err = "witness program: transform longer than 2^12"; return ZK_ERR_ARG; }
            continue;
        }
        if (S.ty == ZK_DOT_PROD) {
            const int fb = S.fft_bit_length;
            if (fb < 1 || fb > 20) return ZK_ERR_ARG;
            const uint64_t vec_out = n_out >> fb, vec_in = n_prev >> fb;
            std::vector<uint32_t> ptr(vec_out + 1, 0);
            for (uint64_t k = 0; k < S.n_bin; ++k) {
                const zk_bin_gate &gt = S.bin_gates[k];
                if (gt.g >= vec_out || gt.u >= vec_in || gt.v >= vec_in) { ctx->err = "dot-prod gate out of range"; return ZK_ERR_ARG; }
                ++ptr[gt.g + 1];
            }
            for (uint64_t g = 0; g < vec_out; ++g) ptr[g + 1] += ptr[g];
            std::vector<gate_rec> sorted(S.n_bin);
            std::vector<uint32_t> pos(ptr.begin(), ptr.end() - 1);
            for (uint64_t k = 0; k < S.n_bin; ++k) {
                const zk_bin_gate &gt = S.bin_gates[k];
                gate_rec r = {gt.g, gt.u, gt.v, 0};
                sorted[pos[gt.g]++] = r;
            }
            if ((rc = upload(ctx, &D.ev_dot, sorted)) || (rc = upload(ctx, &D.ev_dot_ptr, ptr))) return rc;
            continue;
        }
        std::vector<zk_uni_gate> uni(S.uni_gates, S.uni_gates + S.n_uni);
        // a convolution whose pattern was checked at upload is evaluated from its two tensors (k_conv_eval): its bin gates are not kept
        static const bool conv_eval_on = !(getenv("ZKCNN_CONV_EVAL") && atoi(getenv("ZKCNN_CONV_EVAL")) == 0);
        const bool conv_eval = D.conv_ok && conv_eval_on;
        std::vector<zk_bin_gate> bin;
        if (!conv_eval) bin.assign(S.bin_gates, S.bin_gates + S.n_bin);
        else max_conv_part = std::max<uint64_t>(max_conv_part, (uint64_t) conv_eval_chunks(D.conv, n_out) * n_out);
        bool ok = true;
        for (zk_uni_gate &gt : uni) {
            if (gt.lu == 0) { if (gt.u >= S.size_u[0]) { ok = false; break; } gt.u = S.ori_id_u[gt.u]; }
        }
        for (zk_bin_gate &gt : bin) {
            if (!ok || gt.l > 2) { ok = false; break; }
            if (gt.l == 0) { if (gt.u >= S.size_u[0]) { ok = false; break; } gt.u = S.ori_id_u[gt.u]; }
            if (!(gt.l & 1)) { if (gt.v >= S.size_v[0]) { ok = false; break; } gt.v = S.ori_id_v[gt.v]; }
        }
        if (!ok) { ctx->err = "witness program: gate operand outside its subset"; return ZK_ERR_ARG; }
        std::vector<uint8_t> seen(n_out);
        int g1 = grouped_by_output(uni.data(), uni.size(), n_out, seen, [&](const zk_uni_gate &gt) {
            return (int) gt.sc < ctx->n_two_mul && gt.u < (gt.lu ? n_prev : n0); });
        int g2 = g1 < 0 ? -1 : grouped_by_output(bin.data(), bin.size(), n_out, seen, [&](const zk_bin_gate &gt) {
            return (int) gt.sc < ctx->n_two_mul && gt.u < (gt.l == 0 ? n0 : n_prev) && gt.v < ((gt.l & 1) ? n_prev : n0); });
        if (g1 < 0 || g2 < 0) { ctx->err = "witness gate operand out of range"; return ZK_ERR_ARG; }
        if (!g1) { std::vector<zk_uni_gate> t; regroup(t, uni.data(), uni.size(), n_out); uni.swap(t); }
        if (!g2) { std::vector<zk_bin_gate> t; regroup(t, bin.data(), bin.size(), n_out); bin.swap(t); }
        zk_uni_gate *du = nullptr;
        zk_bin_gate *db = nullptr;
        if ((rc = upload(ctx, &du, uni)) || (rc = upload(ctx, &db, bin))) return rc;
        D.ev_uni = du; D.n_ev_uni = uni.size();
        D.ev_bin = db; D.n_ev_bin = bin.size();
        D.ev_conv = conv_eval;
        max_out = std::max(max_out, n_out);
        max_blocks = std::max<uint64_t>(max_blocks, (std::max(uni.size(), bin.size()) + ZK_BLOCK - 1) / ZK_BLOCK);
    }
    std::vector<zk_witness_op> vops(ops, ops + n_ops);
    std::vector<uint32_t> vwin(windows, windows + n_windows);
    zk_witness_op *d_ops = nullptr;
    if ((rc = upload(ctx, &d_ops, vops)) || (rc = upload(ctx, &ctx->wp_windows, vwin))) return rc;
    ctx->wp_ops = d_ops;
    ctx->wp_n_ops = n_ops;
    ctx->wp_n_windows = n_windows;
    if (max_conv_part && (rc = zk_dev_alloc(ctx, (void **) &ctx->wp_conv_part, max_conv_part * 32))) return rc;
    if ((rc = zk_dev_alloc(ctx, (void **) &ctx->wp_tmp, 2 * max_out * 32)) ||
        (rc = zk_dev_alloc(ctx, (void **) &ctx->wp_carry_val, 2 * max_blocks * 32)) ||
        (rc = zk_dev_alloc(ctx, (void **) &ctx->wp_carry_key, 2 * max_blocks * 4)) ||
        (rc = zk_dev_alloc(ctx, (void **) &ctx->wp_ranges, (2 * (size_t) n_ranges + 1 + n_layers) * 8)))
        return rc;
    ZK_HIP(hipHostMalloc((void **) &ctx->h_wp_ranges, (2 * (size_t) n_ranges + 1 + n_layers) * 8));
    {
        std::vector<wit_segment> seg(n_layers);
        for (int i = 0; i < n_layers; ++i) { seg[i].p = ctx->L[i].val; seg[i].n = ctx->L[i].d.size; }
        wit_segment *d_seg = nullptr;
        if ((rc = upload(ctx, &d_seg, seg))) return rc;
        ctx->wp_segments = d_seg;
    }
    ctx->wp_n_ranges = n_ranges;
    ctx->wp_steps.assign(steps, steps + n_steps);
    ctx->wp_ready = true;
    return ZK_OK;
}

// FFT / IFFT layer between two resident value tables (the host-array variant is zk_witness_ntt)
static int32_t resident_ntt(zk_ctx *ctx, const dev_layer &D, const dev_layer &P) {
    const int logn = D.d.fft_bit_length;
    const bool inverse = D.d.ty == ZK_IFFT;
    const uint32_t len = 1u << logn, in_len = inverse ? len : len / 2, out_len = inverse ? len / 2 : len;
    const uint64_t count = D.d.size / out_len;
    if (!count || (uint64_t) count * in_len > P.val_len) { ctx->err = "transform layer larger than its input"; return ZK_ERR_ARG; }
    fr_t *pw = powers_of_root(ctx, logn, inverse);
    if (!pw) { ctx->err = "root table allocation failed"; return ZK_ERR_NOMEM; }
    HFr ilen;
    HFr::inv(ilen, HFr((unsigned long long) len));
    static bool attr_set = false;
    if (!attr_set) {
        ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_batch), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    const uint32_t threads = std::min<uint32_t>(1024, std::max<uint32_t>(64, len / 2));
    prof_begin(ctx, PC_MISC, 64.0 * len * count);
    hipLaunchKernelGGL(k_ntt_batch, dim3((uint32_t) count), dim3(threads), (size_t) len * 32, ctx->stream, D.val, P.val, pw, logn, in_len, out_len,
                       to_dev(ilen), inverse ? 1 : 0);
    prof_end(ctx, PC_MISC);
    return ZK_OK;
}

extern "C" int32_t zk_witness_rerun(zk_ctx *ctx, const uint64_t *picture, uint64_t n_picture, uint64_t *ranges, uint32_t n_ranges,
                                    uint64_t *last_layer, uint64_t n_last) {
    CHECK_READY();
    if (!ctx->wp_ready) { ctx->err = "no witness program on this context"; return ZK_ERR_STATE; }
    dev_layer &L0 = ctx->L[0];
    if (!picture || !n_picture || n_picture > L0.d.size || n_ranges != ctx->wp_n_ranges || (n_ranges && !ranges) ||
        (last_layer && n_last > ctx->L.back().d.size))
        return ZK_ERR_ARG;
    int32_t rc;
    ZK_HIP(hipMemcpyAsync(L0.val, picture, n_picture * 32, hipMemcpyHostToDevice, ctx->stream));
    const size_t n_lay = ctx->L.size(), wp_words = 2 * (size_t) n_ranges + 1 + n_lay;
    ZK_HIP(hipMemsetAsync(ctx->wp_ranges, 0, wp_words * 8, ctx->stream));
    uint32_t *flags = (uint32_t *) (ctx->wp_ranges + 2 * (size_t) n_ranges);
    uint32_t range_k = 0;
    for (const zk_witness_step &st : ctx->wp_steps) {
        dev_layer &D = ctx->L[st.layer];
        if (st.what == 0) {
            const uint64_t n = st.op_end - st.op_begin;
            if (!n) continue;
            ZK_LAUNCH(PC_MISC, 0.0, k_witness_aux, dim3(grid_for(n)), dim3(ZK_BLOCK), L0.val, (const fr_t *) D.val, (const wit_op *) ctx->wp_ops + st.op_begin, n,
                      ctx->wp_windows ? ctx->wp_windows + st.win_begin : nullptr, (uint32_t) st.win, flags);
        } else if (st.what == 2) {
            ZK_LAUNCH(PC_MISC, 0.0, k_witness_range, dim3(grid_for(D.d.size, 512)), dim3(ZK_BLOCK), ctx->wp_ranges + 2 * (size_t) range_k, (const fr_t *) D.val,
                      (uint64_t) D.d.size, flags);
            ++range_k;
        } else {
            const dev_layer &P = ctx->L[st.layer - 1];
            if (D.d.ty == ZK_FFT || D.d.ty == ZK_IFFT) {
                if ((rc = resident_ntt(ctx, D, P))) return rc;
            } else if (D.d.ty == ZK_DOT_PROD) {
                const int fb = D.d.fft_bit_length;
                const uint64_t len = 1ull << fb, n_out = D.d.size >> fb;
                for (uint64_t g0 = 0; g0 < n_out; g0 += 32768) {
                    const uint32_t ng = (uint32_t) std::min<uint64_t>(32768, n_out - g0);
                    dim3 grid((uint32_t) ((len + ZK_BLOCK - 1) / ZK_BLOCK), ng);
                    ZK_LAUNCH(PC_DOT, 0.0, k_dot_witness, grid, dim3(ZK_BLOCK), D.val + g0 * len, (const fr_t *) P.val, (const gate_rec *) D.ev_dot, D.ev_dot_ptr + g0, fb);
                }
            } else {
                const uint64_t n_out = D.d.size;
                fr_t *dA = ctx->wp_tmp, *dB = dA + n_out;
                ZK_HIP(hipMemsetAsync(dA, 0, 2 * n_out * 32, ctx->stream));
                if (D.n_ev_uni) {
                    const uint64_t blocks = (D.n_ev_uni + ZK_BLOCK - 1) / ZK_BLOCK;
                    ZK_LAUNCH(PC_GATE, 44.0 * (double) D.n_ev_uni, k_eval_uni, dim3((uint32_t) blocks), dim3(ZK_BLOCK), dA, ctx->wp_carry_key, ctx->wp_carry_val,
                              (const uni_gate_dev *) D.ev_uni, D.n_ev_uni, (const fr_t *) L0.val, (const fr_t *) P.val, (const fr_t *) ctx->two_mul);
                    ZK_LAUNCH(PC_GATE_FIX, 0.0, k_gate_fixup, dim3(grid_for(2 * blocks)), dim3(ZK_BLOCK), dA, ctx->wp_carry_key, ctx->wp_carry_val, 2 * blocks);
                }
                if (D.ev_conv) {
                    const conv_desc &c = D.conv;
                    const uint32_t chunks = conv_eval_chunks(c, n_out), per = (c.CI + chunks - 1) / chunks;
                    fr_t *part = chunks == 1 ? dB : ctx->wp_conv_part;
                    ZK_LAUNCH(PC_GATE, 80.0 * (double) n_out * c.CI * c.m * c.m, k_conv_eval, dim3((uint32_t) ((n_out + ZK_BLOCK - 1) / ZK_BLOCK), chunks), dim3(ZK_BLOCK), part,
                              (const fr_t *) P.val, (const fr_t *) L0.val + c.wstart, c, per);
                    if (chunks > 1)
                        ZK_LAUNCH(PC_GATE, 0.0, k_sum_rows, dim3((uint32_t) ((n_out + 63) / 64)), dim3(1024), dB, (const fr_t *) part, (uint32_t) n_out, chunks);
                }
                if (D.n_ev_bin) {
                    const uint64_t blocks = (D.n_ev_bin + ZK_BLOCK - 1) / ZK_BLOCK;
                    ZK_LAUNCH(PC_GATE, 80.0 * (double) D.n_ev_bin, k_eval_bin, dim3((uint32_t) blocks), dim3(ZK_BLOCK), dB, ctx->wp_carry_key, ctx->wp_carry_val,
                              (const bin_gate_dev *) D.ev_bin, D.n_ev_bin, (const fr_t *) L0.val, (const fr_t *) P.val, (const fr_t *) ctx->two_mul);
                    ZK_LAUNCH(PC_GATE_FIX, 0.0, k_gate_fixup, dim3(grid_for(2 * blocks)), dim3(ZK_BLOCK), dB, ctx->wp_carry_key, ctx->wp_carry_val, 2 * blocks);
                }
                const HFr sc = H(D.d.scale);
                ZK_LAUNCH(PC_MISC, 0.0, k_eval_combine, dim3(grid_for(n_out)), dim3(ZK_BLOCK), D.val, (const fr_t *) dA, (const fr_t *) dB, to_dev(sc),
                          sc == HFr::one() ? 0 : 1, n_out);
            }
        }
    }
    // where every layer's values end (the round kernels skip the zero tails of the tables they fold)
    ZK_LAUNCH(PC_MISC, 0.0, k_last_nonzero, dim3(64, (uint32_t) n_lay), dim3(ZK_BLOCK), ctx->wp_ranges + 2 * (size_t) n_ranges + 1, (const wit_segment *) ctx->wp_segments);
    ZK_HIP(hipGetLastError());
    for (dev_layer &D : ctx->L) D.val_live = D.val_len;          // until the copy below is back
    ZK_HIP(hipMemcpyAsync(ctx->h_wp_ranges, ctx->wp_ranges, wp_words * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (last_layer && n_last) ZK_HIP(hipMemcpyAsync(last_layer, ctx->L.back().val, n_last * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    for (uint32_t k = 0; k < 2 * n_ranges; ++k) ranges[k] = ctx->h_wp_ranges[k];
    for (size_t i = 0; i < n_lay; ++i) ctx->L[i].val_live = ctx->h_wp_ranges[2 * (size_t) n_ranges + 1 + i];
    if (ctx->h_wp_ranges[2 * (size_t) n_ranges] & WIT_FLAG_WIDE) { ctx->err = "a layer value does not fit 63 bits"; return ZK_ERR_STATE; }
    return ZK_OK;
}

// ---- micro-benchmarks (device resident, HIP events on the context's stream) ----
template <class Launch>
static int32_t time_launches(zk_ctx *ctx, uint32_t iters, double *sec, Launch launch) {
    hipEvent_t e0, e1;
    ZK_HIP(hipEventCreate(&e0));
    ZK_HIP(hipEventCreate(&e1));
    launch();                                   // warm-up
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_HIP(hipEventRecord(e0, ctx->stream));
    for (uint32_t i = 0; i < iters; ++i) launch();
    ZK_HIP(hipEventRecord(e1, ctx->stream));
    ZK_HIP(hipEventSynchronize(e1));
    ZK_HIP(hipGetLastError());
    float ms = 0;
    ZK_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *sec = (double) ms * 1e-3 / iters;
    return ZK_OK;
}

extern "C" int32_t zk_bench_fr_mul(zk_ctx *ctx, uint64_t n_threads, uint32_t muls_per_thread, uint32_t iters, double *sec) {
    CHECK_CTX();
    int32_t rc = zk_scratch(ctx, 2 * n_threads * 32);
    if (rc) return rc;
    int lg = 0;
    while ((2ull << lg) <= n_threads) ++lg;
    std::vector<HFr> r(lg, HFr(0x1234567LL));
    for (int i = 0; i < lg; ++i) r[i] = r[i] * HFr((long long) (i * 7919 + 3)) + HFr(0x9e3779b9LL);
    ZK_HIP(hipMemsetAsync(ctx->scratch.p, 0, 2 * n_threads * 32, ctx->stream));
    if ((rc = eq_table1(ctx, (fr_t *) ctx->scratch.p, lg, r.data(), HFr(77LL)))) return rc;
    const uint32_t blocks = (uint32_t) ((n_threads + ZK_BLOCK - 1) / ZK_BLOCK);
    return time_launches(ctx, iters, sec, [&] {
        ZK_LAUNCH(PC_MISC, 0.0, k_bench_fr_mul, dim3(blocks), dim3(ZK_BLOCK), (fr_t *) ctx->scratch.p, muls_per_thread, n_threads);
    });
}

extern "C" int32_t zk_bench_copy(zk_ctx *ctx, uint64_t bytes, uint32_t iters, double *sec) {
    CHECK_CTX();
    int32_t rc = zk_scratch(ctx, 2 * bytes);
    if (rc) return rc;
    uint4 *src = (uint4 *) ctx->scratch.p, *dst = src + bytes / 16;
    ZK_HIP(hipMemsetAsync(src, 1, bytes, ctx->stream));
    return time_launches(ctx, iters, sec, [&] {
        ZK_LAUNCH(PC_MISC, 0.0, k_bench_copy, dim3(4096), dim3(ZK_BLOCK), dst, src, bytes / 16);
    });
}

// The dominant kernel of the FFT-conv configs in isolation: one non-first quadratic round (k_round_quad2, the product kernel) on two
// 2^log_n-entry tables of pseudo-random elements. Algorithmic bytes = 96 * 2^log_n.
extern "C" int32_t zk_bench_round_quadratic(zk_ctx *ctx, uint32_t log_n, uint32_t iters, double *sec_per_launch,
                                            double *algorithmic_bytes) {
    CHECK_CTX();
    if (log_n < 2 || log_n > 28) return ZK_ERR_ARG;
    const uint64_t n = 1ull << log_n;
    int32_t rc = zk_scratch(ctx, 3 * n * 32);
    if (rc) return rc;
    fr_t *dV = (fr_t *) ctx->scratch.p, *dM = dV + n, *dO = dM + n;
    // fill with field elements: eq tables of a fixed point are as good as random for timing purposes
    std::vector<HFr> r(log_n);
    zkff::Xoshiro g;
    g.seed(0x5EED0002ULL);
    for (auto &x : r) {
        uint64_t t[4] = {g.next(), g.next(), g.next(), g.next() & 0x3fffffffffffffffULL};
        x = HFr(zkff::MontField<zkff::FrParams>::fromCanonical(t));
    }
    if ((rc = eq_table1(ctx, dV, (int) log_n, r.data(), HFr(7LL)))) return rc;
    if ((rc = eq_table1(ctx, dM, (int) log_n, r.data(), HFr(11LL)))) return rc;
    // the product kernel exactly as quad_round launches it for one large table pair (fold + round sums + grid finish + host slot)
    round2_args A;
    std::memset(&A, 0, sizeof(A));
    A.Vin[0] = dV; A.Min[0] = dM; A.Vout[0] = dO; A.Mout[0] = dO + n / 2;
    A.n[0] = n;
    A.nl[0] = n;
    A.blocks[0] = std::min<uint32_t>(grid_for(n / 4, 1024), ctx->partial_blocks / 2);
    A.r = to_dev(r[0]);
    A.skip_p1 = 1;                        // as every round but the first of a phase runs: b comes from the running claim
    A.partials = ctx->partials;
    A.counter = ctx->d_counter;
    A.slot = (host_slot *) ctx->d_slot;
    *algorithmic_bytes = 96.0 * (double) n;
    rc = time_launches(ctx, iters, sec_per_launch, [&] {
        A.seq = ++ctx->slot_seq;
        ZK_LAUNCH(PC_ROUND_QUAD, 96.0 * (double) n, k_round_quad2, dim3(A.blocks[0]), dim3(ZK_BLOCK), A);
    });
    if (rc) return rc;
    return wait_slot(ctx, ctx->slot_seq);
}

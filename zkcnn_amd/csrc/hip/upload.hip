// libzkcnn_hip.so: residency. The circuit comes over the C-ABI as plain gate lists (include/zkcnn_hip.h: zk_upload_circuit*); here they are
// counting-sorted by destination, padded to whole groups, checked against the direct-convolution pattern, and kept in a ref-counted
// registry entry per (device, circuit) that the sessions of a GPU share. No kernel is launched from this file.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include "ctx.hpp"

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
extern "C" int32_t zk_factored_dot_layers(const zk_ctx *ctx) { return ctx ? (int32_t) ctx->dot_layers : 0; }
extern "C" uint64_t zk_dot_deferred_phases(const zk_ctx *ctx) { return ctx ? ctx->dot_defer_count : 0; }

extern "C" int32_t zk_upload_circuit(zk_ctx *ctx, const zk_layer_desc *layers, int32_t n_layers, const uint64_t *two_mul,
                                     int32_t n_two_mul) {
    return zk_upload_circuit_hinted(ctx, layers, n_layers, two_mul, n_two_mul, nullptr, 0);
}

// ---- one resident circuit per GPU, shared by its sessions (reference src/prover.hpp:47-48: one layeredCircuit per prover) ----
// Everything a circuit's upload produces that does not depend on the witness -- sorted and padded gate lists, subset maps, the layer-0 CSR,
// the checked convolution patterns, buffer sizes -- lives in a ref-counted registry entry keyed by a digest of the upload's input. The
// first context that uploads a circuit builds the entry (counting sort of 1.2e8 gates + 1.9 GB of lists for vgg11); every later context of
// the same process and device that uploads the SAME circuit attaches to it and only allocates its own values, tables and scratch.
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

// what a context takes over from a resident circuit it attaches to (its own values, tables and scratch come from alloc_session)
static void adopt_circuit(zk_ctx *ctx, const shared_circuit *e) {
    ctx->L = e->L;
    ctx->two_mul = e->two_mul; ctx->n_two_mul = e->n_two_mul;
    ctx->liu_ptr = e->liu_ptr; ctx->liu_ent = e->liu_ent; ctx->liu_ntabs = e->liu_ntabs;
    ctx->liu_tab_layer = e->liu_tab_layer; ctx->liu_tab_side = e->liu_tab_side;
    ctx->conv_layers = e->conv_layers;
    ctx->sz = e->sz;
}

void zk_circuit_release(zk_ctx *ctx) {
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
            adopt_circuit(ctx, e);
            ++g_circ_attaches;
        } else { ctx->err = "the circuit's first upload failed"; rc = ZK_ERR_STATE; }
    }
    if (rc == ZK_OK) rc = alloc_session(ctx);
    if (rc != ZK_OK) { zk_circuit_release(ctx); return rc; }
    ctx->dot_layers = 0;
    for (const dev_layer &D : ctx->L) ctx->dot_layers += D.dot_ok ? 1 : 0;
    ctx->circuit_ready = true;
    return ZK_OK;
}

// A second context on the resident circuit of `src`, holding a COPY of its witness: what a caller wants who proves many pictures of one model
// (reference src/main_demo_vgg.cpp builds circuit and witness per picture) -- no host copy of the circuit, no gate sort, no digest over 1.2e8
// gates: the new context attaches to the registry entry, allocates its own values / tables / scratch, copies the layer values in HBM and
// adopts the witness program if the circuit has one (zk_witness_rerun then gives it a picture of its own).
extern "C" int32_t zk_ctx_clone(zk_ctx *src, zk_ctx **out) {
    if (!src || !out) return ZK_ERR_ARG;
    *out = nullptr;
    shared_circuit *e = (shared_circuit *) src->circuit;
    if (!src->circuit_ready || !e) { src->err = "zk_ctx_clone: the context holds no circuit"; return ZK_ERR_STATE; }
    zk_ctx *ctx = nullptr;
    int32_t rc = zk_ctx_create(src->device, &ctx);
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> g(g_circ_mtx);
        ++e->refs;
    }
    ctx->circuit = e;
    {
        std::lock_guard<std::mutex> g(e->mtx);      // (another session of the circuit may be uploading its witness program: it writes e->L[i].ev_* under this lock)
        adopt_circuit(ctx, e);
    }
    ++g_circ_attaches;
    rc = alloc_session(ctx);
    if (rc == ZK_OK && (hipSetDevice(src->device) != hipSuccess || hipStreamSynchronize(src->stream) != hipSuccess)) rc = ZK_ERR_HIP;     // (the values copied below are complete)
    for (size_t i = 0; rc == ZK_OK && i < ctx->L.size(); ++i) {
        if (hipMemcpyAsync(ctx->L[i].val, src->L[i].val, ctx->L[i].val_len * 32, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = ZK_ERR_HIP;
        ctx->L[i].val_live = src->L[i].val_live;
    }
    if (rc == ZK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ZK_ERR_HIP;
    if (rc == ZK_OK) {
        ctx->dot_layers = 0;
        for (const dev_layer &D : ctx->L) ctx->dot_layers += D.dot_ok ? 1 : 0;
        ctx->circuit_ready = true;
        std::lock_guard<std::mutex> g(e->mtx);
        if (e->wp_ready) rc = zk_witness_program_adopt(ctx);
    }
    if (rc != ZK_OK) {
        zk_set_create_error(ctx->err.empty() ? "zk_ctx_clone failed" : ctx->err);
        src->err = ctx->err.empty() ? "zk_ctx_clone failed" : ctx->err;
        zk_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
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
        // exact sizes of the bookkeeping buffers (round 3: a session's buffers used to be eight times the LARGEST table of the circuit)
        const uint64_t prev_len = 1ull << layers[i - 1].bit_length;
        const bool dotp = S.ty == ZK_DOT_PROD, xform = S.ty == ZK_FFT || S.ty == ZK_IFFT;
        // beta_g holds what the layer's phase-1 initialisation writes (sumcheck.hip: zk_sumcheck_init_phase1; verifier.hip: zk_verifier_predicates, the same
        // shapes): a transform layer's table runs over its "which vector" bits only, a DOT_PROD layer reads the IFFT layer's (the verifier splits one of
        // 2^(bl - fft_bl)), every other layer -- PADDING's expanded table included -- one entry per output. Round 6: the two buffers were sized by the
        // FFT layer's 2^26 outputs (2 x 2.1 GB per session of vgg11 with pic_cnt = 8) of which 2^-fft_bl were ever written.
        uint64_t bg_need = D.val_len;
        if (S.ty == ZK_FFT || dotp) bg_need = 1ull << std::max<int>(S.bit_length - S.fft_bit_length, 0);
        else if (S.ty == ZK_IFFT) bg_need = 1ull << std::max<int>(S.bit_length - (S.fft_bit_length - 1), 0);
        Z.bg = std::max<uint64_t>(Z.bg, bg_need);
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
        // beta_u / the verifier's beta_v: eq tables over the u (v) variables of a phase 2. A DOT_PROD layer's phase 2 runs over the vector bits only
        // (max_bl_v; the verifier's over max_bl_u - fft_bl): its max_bl_u -- every bit of the FFT layer's index -- sized this buffer at 2.1 GB before round 6
        Z.bu = std::max<uint64_t>(Z.bu, 1ull << std::max<int>(dotp ? std::max<int>(S.max_bl_v, S.max_bl_u - S.fft_bit_length) : std::max<int>(S.max_bl_u, S.max_bl_v), 0));
        for (int bl : {(int) S.bit_length_u[0], (int) S.bit_length_v[0]}) if (bl >= 0) Z.sub = std::max<uint64_t>(Z.sub, 1ull << bl);
        if (S.fft_bit_length >= 0) Z.gs = std::max<uint64_t>(Z.gs, 1ull << S.fft_bit_length);

        if (S.size_u[0]) {
            std::vector<uint32_t> t(S.ori_id_u, S.ori_id_u + S.size_u[0]);
            if ((rc = zk_upload(ctx, &D.ori_u, t))) return rc;
        }
        if (S.size_v[0]) {
            std::vector<uint32_t> t(S.ori_id_v, S.ori_id_v + S.size_v[0]);
            if ((rc = zk_upload(ctx, &D.ori_v, t))) return rc;
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
            if ((rc = zk_upload(ctx, &D.p2[b], q[b]))) return rc;
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
            D.d1_live_rows = 0;
            for (const gate_rec &r : d) D.d1_live_rows = std::max<uint32_t>(D.d1_live_rows, r.key + 1);
            // the generator's pattern (host/neuralNetwork.cpp emitDotProd; reference src/neuralNetwork.cpp dotProdLayer): checked gate by gate, every
            // triple exactly once -- a list that deviates anywhere keeps the generic table builder
            {
                const uint64_t G = S.n_bin, out_vecs = S.size >> S.fft_bit_length;
                uint64_t CI = 0, vmin = ~0ull;
                for (uint64_t k = 0; k < G; ++k) {
                    if (S.bin_gates[k].g == 0) ++CI;
                    vmin = std::min<uint64_t>(vmin, S.bin_gates[k].v);
                }
                const uint64_t pp = CI ? vmin / CI : 0, CO = pp ? out_vecs / pp : 0;
                bool ok = CI && pp >= 2 && CO >= 2 && (CO & (CO - 1)) == 0 && CO <= 4096 && pp * CO == out_vecs && vmin == pp * CI && pp * CO * CI == G &&
                          (pp + CO) * CI <= rows && G < (1ull << 32);
                if (ok) {
                    std::vector<uint8_t> seen((G + 7) / 8, 0);
                    for (uint64_t k = 0; k < G && ok; ++k) {
                        const zk_bin_gate &g = S.bin_gates[k];
                        const uint64_t p = g.g / CO, co = g.g % CO;
                        if (p >= pp || g.u < p * CI || g.u >= (p + 1) * CI) { ok = false; break; }
                        const uint64_t ci = g.u - p * CI, id = (p * CO + co) * CI + ci;
                        if (g.v != (pp + co) * CI + ci || (seen[id >> 3] >> (id & 7)) & 1) { ok = false; break; }
                        seen[id >> 3] |= (uint8_t) (1u << (id & 7));
                    }
                }
                D.dot_ok = ok;
                if (ok) { D.dot_pp = (uint32_t) pp; D.dot_CO = (uint32_t) CO; D.dot_CI = (uint32_t) CI; }
            }
            if ((rc = zk_upload(ctx, &D.d1, d)) || (rc = zk_upload(ctx, &D.d1_rowptr, ptr))) return rc;
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
            if ((rc = zk_upload(ctx, &D.p1[b], p[b]))) return rc;
            std::vector<gate_rec>().swap(p[b]);
        }
        D.n_uni2 = un.size();
        if ((rc = zk_upload(ctx, &D.uni2, un))) return rc;
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
                if (bl > 26) { ctx->err = "layer-0 subset table too large for the combined gather"; return ZK_ERR_ARG; }
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
        if ((rc = zk_upload(ctx, &ctx->liu_ptr, cnt)) || (rc = zk_upload(ctx, &d_ent, ent))) return rc;
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
    ZK_CHECK_CTX();                     // (a resident round kernel of an abandoned phase leaves first; a lane's deferred launches go ahead of the copy)
    if (n) ZK_HIP(hipMemcpyAsync(D.val, values, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if (n < D.val_len) ZK_HIP(hipMemsetAsync(D.val + n, 0, (D.val_len - n) * 32, ctx->stream));
    uint64_t last = n;
    while (last && !(values[4 * last - 4] | values[4 * last - 3] | values[4 * last - 2] | values[4 * last - 1])) --last;
    D.val_live = last;
    if (layer == 0) ctx->wp_w64_valid = false;
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

extern "C" int32_t zk_poke_layer_value(zk_ctx *ctx, int32_t layer, uint64_t index, const uint64_t value[4]) {
    if (!ctx || !ctx->circuit_ready || layer < 0 || layer >= (int) ctx->L.size() || !value) return ZK_ERR_ARG;
    dev_layer &D = ctx->L[layer];
    if (index >= D.d.size) { ctx->err = "poke: index behind the layer"; return ZK_ERR_ARG; }
    ZK_CHECK_CTX();
    ZK_HIP(hipMemcpyAsync(D.val + index, value, 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    if ((value[0] | value[1] | value[2] | value[3]) && index + 1 > D.val_live) D.val_live = index + 1;      // (an upper bound stays an upper bound)
    if (layer == 0) ctx->wp_w64_valid = false;
    return ZK_OK;
}


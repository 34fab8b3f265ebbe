// C entry points of libzkcnn_host.so (include/zkcnn_api.h): circuit + witness on the host, the
// prover on the GPU, the reference verifier driving it.
#include <mutex>
#include <stdexcept>
#include <cstddef>
#define ZKFIBER_IMPLEMENTATION
#include "session.hpp"
#include "verifier_alias.hpp"

// witness kernels of the FFT convolution on the GPU (include/zkcnn_hip.h: zk_witness_ntt / zk_witness_dotprod)
class hipWitnessAccel : public witnessAccel {
public:
    // Runs on the session's OWN context and stream (the circuit is uploaded to it afterwards): one context, one HIP stream and one set
    // of scratch buffers per session; the witness buffers are released when the witness is complete
    explicit hipWitnessAccel(zk_ctx *session_ctx) : ctx(session_ctx) {}
    ~hipWitnessAccel() override { if (ctx) zk_witness_release(ctx); }
    bool ntt(F *dst, const F *src, int logn, bool inverse, size_t count) override {
        if (logn > 12) return false;                 // longer transforms do not fit in LDS; the host loop takes them
        int rc = zk_witness_ntt(ctx, reinterpret_cast<uint64_t *>(dst), reinterpret_cast<const uint64_t *>(src), logn, inverse, count);
        if (rc != ZK_OK) throw std::runtime_error(string("zk_witness_ntt: ") + zk_last_error(ctx));
        return true;
    }
    bool dotProd(F *out, size_t n_out, const F *in, size_t n_in, const binGate *gates, size_t n_gates, int fft_bl) override {
        int rc = zk_witness_dotprod(ctx, reinterpret_cast<uint64_t *>(out), n_out, reinterpret_cast<const uint64_t *>(in), n_in,
                                    reinterpret_cast<const zk_bin_gate *>(gates), n_gates, fft_bl);
        if (rc != ZK_OK) throw std::runtime_error(string("zk_witness_dotprod: ") + zk_last_error(ctx));
        return true;
    }
    bool input(size_t offset, const F *values, size_t n) override {
        int rc = zk_witness_input(ctx, offset, reinterpret_cast<const uint64_t *>(values), n);
        if (rc != ZK_OK) throw std::runtime_error(string("zk_witness_input: ") + zk_last_error(ctx));
        return true;
    }
    bool gates(F *out, size_t n_out, const uniGate *uni, size_t n_uni, const binGate *bin, size_t n_bin, const F *prev, size_t n_prev,
               const F *two_mul, size_t n_two_mul, const F &scale) override {
        static_assert(sizeof(uniGate) == sizeof(zk_uni_gate) && sizeof(binGate) == sizeof(zk_bin_gate), "gate records cross the C-ABI as they are");
        int rc = zk_witness_gates(ctx, reinterpret_cast<uint64_t *>(out), n_out, reinterpret_cast<const zk_uni_gate *>(uni), n_uni,
                                  reinterpret_cast<const zk_bin_gate *>(bin), n_bin, reinterpret_cast<const uint64_t *>(prev), n_prev,
                                  reinterpret_cast<const uint64_t *>(two_mul), (uint32_t) n_two_mul, reinterpret_cast<const uint64_t *>(&scale));
        if (rc != ZK_OK) throw std::runtime_error(string("zk_witness_gates: ") + zk_last_error(ctx));
        return true;
    }
private:
    zk_ctx *ctx;
};

// wiring predicates of the verifier on the prover's GPU context (include/zkcnn_hip.h: zk_verifier_*)
class hipVerifierAccel : public verifierAccel {
public:
    explicit hipVerifierAccel(prover *pr) : p(pr) {}
    void predicates(u8 layer, const F *r_0, const F *r_1, const F &alpha, const F &beta, const F &relu_rou, const F *r_u, const F *r_v,
                    const F *r_u2, const F *r_v2, F uni[2], F bin[3]) override {
        auto q = [](const F *x) { return reinterpret_cast<const uint64_t *>(x); };
        int rc = zk_verifier_predicates(p->context(), layer, q(r_0), q(r_1), q(&alpha), q(&beta), q(&relu_rou), q(r_u), q(r_v), q(r_u2), q(r_v2),
                                        reinterpret_cast<uint64_t *>(uni), reinterpret_cast<uint64_t *>(bin));
        if (rc != ZK_OK) throw std::runtime_error(string("zk_verifier_predicates: ") + zk_last_error(p->context()));
    }
    // the two MSMs of the opening check on the prover's GPU: over the proof's generators (their tables are resident) or over the
    // commitment rows (a second, verifier-owned table set so that the prover's byte table survives)
    bool msm(G1 &out, const Fr *k, const G1Affine *bases, size_t n, bool bases_are_generators) override {
        uint64_t pt[12];
        int rc = zk_verifier_msm(p->context(), pt, reinterpret_cast<const uint64_t *>(k), reinterpret_cast<const uint64_t *>(bases), n, bases_are_generators ? 1 : 0);
        if (rc == ZK_ERR_STATE) return false;            // generators not cached on this context: the host takes it
        if (rc != ZK_OK) throw std::runtime_error(string("zk_verifier_msm: ") + zk_last_error(p->context()));
        G1Affine a;
        std::memcpy(&a, pt, 96);
        out = G1::fromAffine(a);
        return true;
    }
    F inputPredicate(const vector<F> &r_u0, const vector<vector<F>> &r_u, const vector<vector<F>> &r_v, const vector<F> &sig_u,
                     const vector<F> &sig_v) override {
        const size_t n = sig_u.size();
        vector<const uint64_t *> pu(n + 1, nullptr), pv(n + 1, nullptr);
        for (size_t i = 1; i <= n; ++i) {
            pu[i] = reinterpret_cast<const uint64_t *>(r_u[i].data());
            pv[i] = reinterpret_cast<const uint64_t *>(r_v[i].data());
        }
        F out;
        int rc = zk_verifier_input_predicate(p->context(), reinterpret_cast<const uint64_t *>(r_u0.data()), pu.data(), pv.data(),
                                             reinterpret_cast<const uint64_t *>(sig_u.data()), reinterpret_cast<const uint64_t *>(sig_v.data()),
                                             (uint32_t) n, reinterpret_cast<uint64_t *>(&out));
        if (rc != ZK_OK) throw std::runtime_error(string("zk_verifier_input_predicate: ") + zk_last_error(p->context()));
        return out;
    }
private:
    prover *p;
};

static_assert(sizeof(witnessOp) == sizeof(zk_witness_op) && sizeof(witnessStep) == sizeof(zk_witness_step) && offsetof(witnessStep, win) == offsetof(zk_witness_step, win) &&
              offsetof(witnessOp, shift) == offsetof(zk_witness_op, shift), "the recorded witness program crosses the C-ABI as it is");

struct gpuSession : public sessionT<prover> {
    explicit gpuSession(int device) : sessionT<prover>(device), dev(device), va(&p) { vaccel = &va; }
    int dev;
    hipVerifierAccel va;

    // the next picture on the resident circuit: 0 = done, 1 = other input scale (untouched), 2 = other activation scale (values stale)
    int newImage(const vector<double> &pixels) {
        if (!nn || nn->program().steps.empty() || nn->program().picture_values == 0 || !ever_had_witness)
            throw std::runtime_error("this session has no witness program (verifier-only session?)");
        vector<F> picture;
        if (!nn->quantisePicture(pixels, picture)) return 1;
        const witnessProgram &pg = nn->program();
        if (picture.size() != pg.picture_values) throw std::runtime_error("picture size differs from the circuit's");
        ensureProgram();
        size_t n_ranges = 0;
        for (const witnessStep &st : pg.steps) n_ranges += st.what == witnessStep::RANGE;
        vector<u64> raw;
        vector<F> last;
        has_witness = false;
        p.rerunWitness(picture, raw, n_ranges, last);
        vector<std::pair<u64, u64>> ranges(n_ranges);
        for (size_t k = 0; k < n_ranges; ++k) ranges[k] = std::make_pair(raw[2 * k], raw[2 * k + 1]);
        if (!nn->rangesFitScales(ranges)) {
            // refused: put the picture the session proved before back (a second replay; its ranges are the circuit's by construction)
            if (good_picture.size() == picture.size()) {
                p.rerunWitness(good_picture, raw, n_ranges, last);
                for (size_t k = 0; k < n_ranges; ++k) ranges[k] = std::make_pair(raw[2 * k], raw[2 * k + 1]);
                has_witness = nn->rangesFitScales(ranges);
            }
            return 2;
        }
        inferred = nn->inferenceFrom(last);      // (this session's: the generator object is shared with its clones and stays untouched)
        good_picture.swap(picture);
        has_witness = true;
        return 0;
    }
    // the witness program on the GPU (recorded by the circuit generator): uploaded with the first new picture, or before the session is cloned
    void ensureProgram() {
        if (p.hasWitnessProgram() || !nn || nn->program().steps.empty()) return;
        const witnessProgram &pg = nn->program();
        p.uploadWitnessProgram(reinterpret_cast<const zk_witness_op *>(pg.ops.data()), pg.ops.size(), pg.windows.data(), pg.windows.size(),
                               reinterpret_cast<const zk_witness_step *>(pg.steps.data()), pg.steps.size());
    }
    std::mutex clone_mtx;              // zkcnn_session_clone: the parent's one-time preparations
    bool ever_had_witness = false;     // set once a build with picture + weights succeeded (a verifier-only session never gets there)
    vector<int> inferred;              // classes inferred for the picture in HBM (after a new_image)
    vector<F> good_picture;            // quantised picture of the witness in HBM (what a refused new_image restores)
};

// Test hooks (a message corrupted on purpose, a resident witness value overwritten) are part of the library because the parity and
// soundness tests drive the PRODUCT through them -- but a deployment must opt in: ZKCNN_TEST_HOOKS=1 (tests/conftest.py sets it).
static bool test_hooks_enabled() {
    static const bool on = getenv("ZKCNN_TEST_HOOKS") && atoi(getenv("ZKCNN_TEST_HOOKS")) != 0;
    return on;
}

// K sessions of one model on one GPU as the lanes of a lock-step batch (include/zkcnn_hip.h: zk_batch_*): their contexts share one stream,
// their verifier loops one host thread (session.hpp: batchSessionT), and every sumcheck round of the K proofs is ONE kernel launch
struct gpuBatch : public batchSessionT<gpuSession> {
    zk_batch *b = nullptr;
    ~gpuBatch() { if (b) zk_batch_destroy(b); }             // (the lanes' contexts go back to their own streams; the sessions stay their owner's)
    static void yieldLane(void *, int32_t) { zkfiber::fiber::yield(); }
};

extern "C" {

void *zkcnn_batch_create(void *const *sessions, int32_t n) {
    if (!sessions || n < 1 || n > ZK_BATCH_MAX_LANES) return nullptr;
    try {
        std::unique_ptr<gpuBatch> g(new gpuBatch());
        for (int32_t k = 0; k < n; ++k) {
            gpuSession *s = (gpuSession *) sessions[k];
            if (!s || !s->p.context()) return nullptr;
            if (!g->b && zk_batch_create(s->dev, &g->b) != ZK_OK) {
                fprintf(stderr, "zkcnn_batch_create: %s\n", zk_batch_last_error(nullptr));
                return nullptr;
            }
            if (zk_batch_attach(g->b, s->p.context(), nullptr) != ZK_OK) {
                fprintf(stderr, "zkcnn_batch_create: lane %d: %s\n", (int) k, zk_batch_last_error(g->b));
                return nullptr;
            }
            g->lanes.push_back(s);
        }
        zk_batch_set_yield(g->b, &gpuBatch::yieldLane, nullptr);
        return g.release();
    } catch (const std::exception &e) {
        fprintf(stderr, "zkcnn_batch_create: %s\n", e.what());
        return nullptr;
    }
}

// Several batches at once: every batch's STREAM is created before any lane is attached. HIP spreads streams over GPU_MAX_HW_QUEUES hardware queues as they
// are created, and streams that share a hardware queue run their kernels one after the other: made one by one (a batch stream, then its lanes' own streams
// destroyed, then the next batch stream ...), six batches landed 2 + 2 + 1 + 1 on four queues and eight batches 1 + 3 + 2 + 2 -- the step time is the most
// loaded queue's (round 6: profiles/r06_variance.md; scripts/exp/stream_pipes.hip shows the effect in isolation). Created back to back they spread evenly.
int32_t zkcnn_batch_create_group(void *const *sessions, const int32_t *counts, int32_t n_batches, void **out) {
    if (!sessions || !counts || !out || n_batches < 1) return -1;
    std::vector<std::unique_ptr<gpuBatch>> gs;
    try {
        int32_t at = 0;
        for (int32_t j = 0; j < n_batches; ++j) {
            if (counts[j] < 1 || counts[j] > ZK_BATCH_MAX_LANES) return -1;
            gpuSession *s0 = (gpuSession *) sessions[at];
            if (!s0 || !s0->p.context()) return -1;
            gs.emplace_back(new gpuBatch());
            if (zk_batch_create(s0->dev, &gs.back()->b) != ZK_OK) {
                fprintf(stderr, "zkcnn_batch_create_group: %s\n", zk_batch_last_error(nullptr));
                return -2;
            }
            at += counts[j];
        }
        at = 0;
        for (int32_t j = 0; j < n_batches; ++j) {
            for (int32_t k = 0; k < counts[j]; ++k, ++at) {
                gpuSession *s = (gpuSession *) sessions[at];
                if (!s || !s->p.context()) return -1;
                if (zk_batch_attach(gs[j]->b, s->p.context(), nullptr) != ZK_OK) {
                    fprintf(stderr, "zkcnn_batch_create_group: batch %d lane %d: %s\n", (int) j, (int) k, zk_batch_last_error(gs[j]->b));
                    return -2;
                }
                gs[j]->lanes.push_back(s);
            }
            zk_batch_set_yield(gs[j]->b, &gpuBatch::yieldLane, nullptr);
        }
        for (int32_t j = 0; j < n_batches; ++j) out[j] = gs[j].release();
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "zkcnn_batch_create_group: %s\n", e.what());
        return -2;
    }
}

// The zero-knowledge mode adds the masks' share of a phase's LAST round polynomial on the host, from the phase's last table pairs, so it needs a host tail
// and fixes the hand-over point itself (tables of <= 32 entries finish their phase on the host: prover::zkModeOn). ZKCNN_MODE_HOST_TAIL next to it is
// harmless (a host tail either way; the zero-knowledge mode's size applies); ZKCNN_MODE_GPU_TAIL -- "every round of every phase is a kernel" -- cannot
// be honoured and used to be ignored silently (round-5 advisor finding): it is refused.
static bool zk_tail_conflict(uint32_t mode) { return (mode & ZKCNN_MODE_ZK) && (mode & ZKCNN_MODE_GPU_TAIL); }
static const char *const ZK_TAIL_MSG = "ZKCNN_MODE_ZK needs its host tail (the masks of a phase's last round are added on the host): not together with ZKCNN_MODE_GPU_TAIL";

int32_t zkcnn_batch_prove(void *batch, const uint64_t *seeds, uint32_t mode, uint8_t *const *transcripts, const uint64_t *caps, zkcnn_result *out, double *wall_s) {
    if (!batch || !out) return -1;
    gpuBatch *g = (gpuBatch *) batch;
    if ((mode & ZKCNN_MODE_TAMPER) && !test_hooks_enabled()) return -4;
    if (zk_tail_conflict(mode)) {
        std::memset(out, 0, sizeof(*out));
        std::snprintf(out[0].message, sizeof(out[0].message), "%s", ZK_TAIL_MSG);
        return -4;
    }
    const double t0 = gpuSession::now();
    int flush_rc = ZK_OK;
    zk_batch *b = g->b;
    int rc = g->prove(seeds, mode, transcripts, caps, out, [b, &flush_rc]() { int r = zk_batch_flush(b); if (r != ZK_OK && flush_rc == ZK_OK) flush_rc = r; });
    for (size_t k = 0; k < g->lanes.size(); ++k) out[k].upload_s = g->lanes[k]->p.uploadTime();
    if (wall_s) *wall_s = gpuSession::now() - t0;
    if (flush_rc != ZK_OK) {
        std::snprintf(out[0].message, sizeof(out[0].message), "zk_batch_flush: %s", zk_batch_last_error(b));
        return -2;
    }
    return rc;
}

int32_t zkcnn_batch_stats(void *batch, uint64_t out[5]) {
    if (!batch || !out) return -1;
    gpuBatch *g = (gpuBatch *) batch;
    if (zk_batch_stats(g->b, out) != ZK_OK) return -2;
    out[4] = g->rounds_of_flushes;
    return 0;
}

void zkcnn_batch_destroy(void *batch) { delete (gpuBatch *) batch; }

void *zkcnn_session_create(const zkcnn_model_desc *desc, int32_t device) {
    if (!desc) return nullptr;
    try {
        std::unique_ptr<gpuSession> s(new gpuSession(device));      // owns the session until it is complete: nothing leaks when a step throws
        {
            s->p.ensureContext();
            hipWitnessAccel accel(s->p.context());
            s->accel = &accel;
            bool ok = s->build(desc);
            s->accel = nullptr;
            if (!ok) return nullptr;
        }
        s->p.init();                 // residency: circuit + witness to HBM, outside any timed region
        if (s->nn && !s->p.val.empty() && s->p.val[0].size() >= s->nn->program().picture_values)
            s->good_picture.assign(s->p.val[0].begin(), s->p.val[0].begin() + (size_t) s->nn->program().picture_values);
        s->ever_had_witness = true;
        s->p.releaseHostValues();    // the host copy of every layer's values (0.7 GB for vgg11) has no reader left: proofs and new pictures work on the HBM copy
        return s.release();
    } catch (const std::exception &e) {
        fprintf(stderr, "zkcnn_session_create: %s\n", e.what());
        return nullptr;
    }
}

void *zkcnn_session_create_calibrated(const zkcnn_model_desc *desc, const int32_t *scales, uint64_t n_scales, int32_t device) {
    if (!desc || !scales || !n_scales) return nullptr;
    try {
        std::unique_ptr<gpuSession> s(new gpuSession(device));
        const vector<int> given(scales, scales + n_scales);
        {
            s->p.ensureContext();
            hipWitnessAccel accel(s->p.context());
            s->accel = &accel;
            bool ok = s->build(desc, &given);
            s->accel = nullptr;
            if (!ok) return nullptr;
        }
        s->p.init();
        if (s->nn && !s->p.val.empty() && s->p.val[0].size() >= s->nn->program().picture_values)
            s->good_picture.assign(s->p.val[0].begin(), s->p.val[0].begin() + (size_t) s->nn->program().picture_values);
        s->ever_had_witness = true;
        s->p.releaseHostValues();
        return s.release();
    } catch (const std::exception &e) {
        fprintf(stderr, "zkcnn_session_create_calibrated: %s\n", e.what());
        return nullptr;
    }
}


void *zkcnn_session_clone(void *session) {
    if (!session) return nullptr;
    gpuSession *src = (gpuSession *) session;
    try {
        if (!src->ever_had_witness || !src->has_witness) return nullptr;            // (a verifier-only session, or one whose last picture was refused)
        {   // several threads may clone one session side by side (bench.py does): what a clone needs of its parent is made ONCE, under the parent's lock --
            // the witness program on the GPU (the clone adopts it with the circuit: it has no gate lists to build it from) and the wiring digest
            // (cached in the circuit; the structure copy carries it). Everything else a clone reads of its parent is read-only.
            std::lock_guard<std::mutex> g(src->clone_mtx);
            src->ensureProgram();
            (void) src->p.C.wiringDigest();
        }
        std::unique_ptr<gpuSession> s(new gpuSession(src->dev));
        s->nn = src->nn;
        s->gens = src->gens;
        s->model_name = src->model_name;
        s->pic_cnt = src->pic_cnt; s->pic_x = src->pic_x; s->pic_y = src->pic_y; s->pic_channel = src->pic_channel;
        s->data_seed = src->data_seed;
        s->p.cloneFrom(src->p);
        s->good_picture = src->good_picture;
        s->ever_had_witness = true;
        s->has_witness = true;
        return s.release();
    } catch (const std::exception &e) {
        fprintf(stderr, "zkcnn_session_clone: %s\n", e.what());
        return nullptr;
    }
}

int32_t zkcnn_session_prove(void *session, uint64_t seed, uint32_t mode, uint8_t *transcript, uint64_t cap, zkcnn_result *out) {
    if (!session || !out) return -1;
    gpuSession *s = (gpuSession *) session;
    if ((mode & ZKCNN_MODE_TAMPER) && !test_hooks_enabled()) {
        std::memset(out, 0, sizeof(*out));
        std::snprintf(out->message, sizeof(out->message), "ZKCNN_MODE_TAMPER is a test hook: set ZKCNN_TEST_HOOKS=1");
        return -4;
    }
    if (zk_tail_conflict(mode)) {
        std::memset(out, 0, sizeof(*out));
        std::snprintf(out->message, sizeof(out->message), "%s", ZK_TAIL_MSG);
        return -4;
    }
    try {
        int rc = s->prove(seed, mode, transcript, cap, out);
        out->upload_s = s->p.uploadTime();
        return rc;
    } catch (const std::exception &e) {
        std::memset(out, 0, sizeof(*out));
        std::snprintf(out->message, sizeof(out->message), "%s", e.what());
        return -2;
    }
}

int64_t zkcnn_session_statement(void *session, int32_t *scales, uint64_t cap) {
    if (!session) return -1;
    const vector<int> &v = ((gpuSession *) session)->statementScales();
    for (size_t i = 0; i < v.size() && i < cap; ++i) scales[i] = v[i];
    return (int64_t) v.size();
}

void *zkcnn_verifier_create(const zkcnn_model_desc *desc, const int32_t *scales, uint64_t n_scales, int32_t device) {
    if (!desc || (!scales && n_scales)) return nullptr;
    try {
        std::unique_ptr<gpuSession> s(new gpuSession(device));
        if (!s->buildStatement(desc, scales, n_scales)) return nullptr;
        s->p.init();                 // the circuit goes to the GPU for the wiring predicates; there are no values to upload
        return s.release();
    } catch (const std::exception &e) {
        fprintf(stderr, "zkcnn_verifier_create: %s\n", e.what());
        return nullptr;
    }
}

int32_t zkcnn_session_verify(void *session, uint64_t seed, uint32_t mode, const uint8_t *proof, uint64_t len, zkcnn_result *out) {
    if (!session || !out || !proof) return -1;
    try {
        return ((gpuSession *) session)->verifyProof(proof, len, seed, mode, out);
    } catch (const std::exception &e) {
        std::memset(out, 0, sizeof(*out));
        std::snprintf(out->message, sizeof(out->message), "%s", e.what());
        return -2;
    }
}

int32_t zkcnn_session_new_image(void *session, uint64_t picture_seed, const double *pixels, uint64_t n_pixels, double *ms) {
    if (!session) return -1;
    gpuSession *s = (gpuSession *) session;
    try {
        const double t0 = gpuSession::now();
        if (!s->nn) return -1;
        vector<double> px = pixels ? vector<double>(pixels, pixels + n_pixels) : s->nn->syntheticPicture(picture_seed);
        int rc = s->newImage(px);
        if (ms) *ms = 1e3 * (gpuSession::now() - t0);
        return rc;
    } catch (const std::exception &e) {
        fprintf(stderr, "zkcnn_session_new_image: %s\n", e.what());
        return -2;
    }
}

int64_t zkcnn_session_synthetic_picture(void *session, uint64_t picture_seed, double *pixels, uint64_t cap) {
    if (!session || !((gpuSession *) session)->nn || (!pixels && cap)) return -1;
    vector<double> px = ((gpuSession *) session)->nn->syntheticPicture(picture_seed);
    for (size_t i = 0; i < px.size() && i < cap; ++i) pixels[i] = px[i];
    return (int64_t) px.size();
}

int64_t zkcnn_session_layer_size(void *session, int32_t layer, int32_t *type) {
    if (!session) return -1;
    const layeredCircuit &C = ((gpuSession *) session)->p.C;
    if (layer < 0 || layer >= C.size) return -1;
    if (type) *type = (int32_t) C.circuit[layer].ty;
    return (int64_t) C.circuit[layer].size;
}
int32_t zkcnn_session_poke(void *session, int32_t layer, uint64_t index, const uint64_t value[4]) {
    if (!session || !value) return -1;
    if (!test_hooks_enabled()) return -4;
    gpuSession *s = (gpuSession *) session;
    return zk_poke_layer_value(s->p.context(), layer, index, value) == ZK_OK ? 0 : -2;
}

int32_t zkcnn_session_conv_paths(void *session, int32_t force_field, uint64_t out[3]) {
    if (!session || !out) return -1;
    if (!test_hooks_enabled()) return -4;
    gpuSession *s = (gpuSession *) session;
    return zk_witness_conv_paths(s->p.context(), force_field, out) == ZK_OK ? 0 : -2;
}
void zkcnn_session_destroy(void *session) { delete (gpuSession *) session; }

int32_t zkcnn_session_row(void *session, char *buf, uint64_t cap) {
    if (!session || !cap) return -1;
    std::snprintf(buf, cap, "%s", ((gpuSession *) session)->row.c_str());
    return 0;
}

int32_t zkcnn_session_fs_stats(void *session, uint64_t *rounds, uint64_t *phases) {
    if (!session || !rounds || !phases) return -1;
    ((gpuSession *) session)->p.tailStats(*rounds, *phases);
    return 0;
}

int32_t zkcnn_session_host_tail_rounds(void *session, uint64_t *rounds) {
    if (!session || !rounds) return -1;
    *rounds = ((gpuSession *) session)->p.hostTailRounds();
    return 0;
}

int32_t zkcnn_session_structured_layers(void *session) {
    if (!session) return -1;
    return ((gpuSession *) session)->p.structuredLayers();
}
int32_t zkcnn_session_factored_dot_layers(void *session) {
    if (!session) return -1;
    return ((gpuSession *) session)->p.factoredDotLayers();
}
uint64_t zkcnn_session_dot_deferred_phases(void *session) {
    if (!session) return 0;
    return ((gpuSession *) session)->p.dotDeferredPhases();
}

int32_t zkcnn_session_profile(void *session, uint32_t class_mask) {
    if (!session) return -1;
    return zk_profile_enable(((gpuSession *) session)->p.context(), class_mask);
}
int32_t zkcnn_session_profile_report(void *session, char *buf, uint64_t cap, int32_t reset) {
    if (!session) return -1;
    return zk_profile_report(((gpuSession *) session)->p.context(), buf, cap, reset);
}

}  // extern "C"

#include "prover.hpp"
#include <stdexcept>

static_assert(sizeof(uniGate) == sizeof(zk_uni_gate), "uniGate layout must match the C-ABI record");
static_assert(sizeof(binGate) == sizeof(zk_bin_gate), "binGate layout must match the C-ABI record");
static_assert(sizeof(F) == 32, "Fr must be 4 x u64 Montgomery limbs");

// optional wall-clock breakdown per prover method (ZKCNN_TIMING=1): printed when the prover is destroyed
#include <chrono>
#include <map>
#include <mutex>
namespace {
std::mutex g_times_mu;
struct methodTimes {
    std::map<string, std::pair<double, long>> t;
    bool on = std::getenv("ZKCNN_TIMING") != nullptr;
    ~methodTimes() {
        if (!on) return;
        for (auto &kv : t) fprintf(stderr, "[zkcnn timing] %-28s %9.3f ms %7ld calls\n", kv.first.c_str(), 1e3 * kv.second.first, kv.second.second);
    }
} g_times;
struct scopeTimer {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    explicit scopeTimer(const char *n) : name(n) { if (g_times.on) t0 = std::chrono::steady_clock::now(); }
    ~scopeTimer() {
        if (!g_times.on) return;
        std::lock_guard<std::mutex> lk(g_times_mu);
        auto &e = g_times.t[name];
        e.first += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ++e.second;
    }
};
}
#define TIMED() scopeTimer st_(__func__)

static inline const uint64_t *U(const F &x) { return reinterpret_cast<const uint64_t *>(&x); }
static inline uint64_t *U(F &x) { return reinterpret_cast<uint64_t *>(&x); }

prover::prover() : ctx(nullptr), device_id(0), resident(false) {}
prover::prover(int device) : ctx(nullptr), device_id(device), resident(false) {}
prover::~prover() {
    poly_p.reset();
    if (ctx) zk_ctx_destroy(ctx);
}

void prover::check(int rc, const char *what) const {
    if (rc != ZK_OK) throw std::runtime_error(string(what) + " failed: " + zk_last_error(ctx));
}

void prover::attachFiatShamir(const uint32_t *state, const uint64_t *pending) {
    if (!ctx) return;
    check(zk_fs_attach(ctx, state, pending), "zk_fs_attach");
}
void prover::proofBegin() { if (ctx) check(zk_proof_begin(ctx), "zk_proof_begin"); }
void prover::proofEnd() { if (ctx) (void) zk_proof_end(ctx); }
void prover::setLiveRounds(bool on) {
    if (ctx) check(zk_set_live_rounds(ctx, on ? 1 : 0), "zk_set_live_rounds");
}
void prover::setGeneratorReuse(bool reusable) {
    if (ctx) check(zk_set_generator_reuse(ctx, reusable ? 1 : 0), "zk_set_generator_reuse");
}
bool prover::fixedBaseMul(const std::vector<G1Affine> &table, const std::vector<uint64_t> &scalars, std::vector<G1> &out) {
    if (!ctx) return false;
    static_assert(sizeof(G1) == 18 * sizeof(uint64_t) && sizeof(G1Affine) == 12 * sizeof(uint64_t), "point layouts of the C-ABI");
    const size_t n = scalars.size() / 4;
    out.resize(n);
    check(zk_fixed_base_mul(ctx, reinterpret_cast<const uint64_t *>(table.data()), table.size(), scalars.data(), n, reinterpret_cast<uint64_t *>(out.data())), "zk_fixed_base_mul");
    return true;
}
void prover::setHostTail(int log_entries) {
    if (ctx) check(zk_set_host_tail(ctx, log_entries), "zk_set_host_tail");
}
uint64_t prover::hostTailRounds() const {
    uint64_t r = 0;
    if (ctx) (void) zk_host_tail_stats(ctx, &r);
    return r;
}
void prover::tailStats(uint64_t &rounds, uint64_t &phases) const {
    rounds = phases = 0;
    if (ctx) zk_fs_stats(ctx, &rounds, &phases);
}

void prover::ensureContext() {
    if (ctx) return;
    int rc = zk_ctx_create(device_id, &ctx);
    if (rc != ZK_OK) {
        ctx = nullptr;
        throw std::runtime_error(string("zkCNN HIP prover needs an MI355X-class GPU: ") + zk_last_error(nullptr));
    }
}

// This file also builds against the REFERENCE's circuit.h (tests/golden/make_gates_golden.py links the reference's unmodified generator with this
// prover): the two members this repo's layeredCircuit adds -- structureCopy(), gatesDropped() -- are used where they exist.
template <class LC> static auto circuit_structure_copy(const LC &c, int) -> decltype(c.structureCopy()) { return c.structureCopy(); }
template <class LC> static LC circuit_structure_copy(const LC &c, long) { return c; }
template <class LC> static auto circuit_gates_dropped(const LC &c, int) -> decltype(c.gatesDropped()) { return c.gatesDropped(); }
template <class LC> static bool circuit_gates_dropped(const LC &, long) { return false; }

void prover::cloneFrom(prover &parent) {
    if (!parent.ctx || !parent.resident) throw std::runtime_error("prover::cloneFrom: the parent's circuit is not resident (call init() first)");
    if (ctx) throw std::runtime_error("prover::cloneFrom: this prover already has a context");
    zk_ctx *c = nullptr;
    if (zk_ctx_clone(parent.ctx, &c) != ZK_OK) throw std::runtime_error(string("zk_ctx_clone failed: ") + zk_last_error(parent.ctx));
    ctx = c;
    device_id = parent.device_id;
    C = circuit_structure_copy(parent.C, 0);
    vector<vector<F>>().swap(val);
    resident = true;
    program_resident = parent.program_resident;
    prove_timer.clear();
    check(zk_prover_init(ctx), "zk_prover_init");
}

vector<zk_layer_desc> prover::layerDescs() const {
    static_assert(sizeof(layerType) == sizeof(int), "layerType is passed as int32");
    vector<zk_layer_desc> desc(C.size);
    for (int i = 0; i < C.size; ++i) {
        const layer &L = C.circuit[i];
        zk_layer_desc &d = desc[i];
        std::memset(&d, 0, sizeof(d));
        d.ty = (int32_t) L.ty;
        d.size = L.size;
        for (int b = 0; b < 2; ++b) {
            d.size_u[b] = L.size_u[b]; d.size_v[b] = L.size_v[b];
            d.bit_length_u[b] = L.bit_length_u[b]; d.bit_length_v[b] = L.bit_length_v[b];
        }
        d.bit_length = L.bit_length;
        d.max_bl_u = L.max_bl_u; d.max_bl_v = L.max_bl_v;
        d.fft_bit_length = L.fft_bit_length;
        d.need_phase2 = L.need_phase2;
        d.zero_start_id = L.zero_start_id;
        std::memcpy(d.scale, &L.scale, 32);
        // layer 0 only carries identity gates that the sumcheck never walks
        d.uni_gates = i ? reinterpret_cast<const zk_uni_gate *>(L.uni_gates.data()) : nullptr;
        d.n_uni = i ? L.uni_gates.size() : 0;
        d.bin_gates = i ? reinterpret_cast<const zk_bin_gate *>(L.bin_gates.data()) : nullptr;
        d.n_bin = i ? L.bin_gates.size() : 0;
        d.ori_id_u = L.ori_id_u.data();
        d.ori_id_v = L.ori_id_v.data();
    }
    return desc;
}

void prover::uploadWitnessProgram(const zk_witness_op *ops, size_t n_ops, const u32 *windows, size_t n_windows, const zk_witness_step *steps, size_t n_steps) {
    if (!ctx || !resident) throw std::runtime_error("uploadWitnessProgram: the circuit is not resident (call init() first)");
    vector<zk_layer_desc> desc = layerDescs();
    check(zk_witness_program_upload(ctx, ops, n_ops, windows, n_windows, steps, (uint32_t) n_steps, desc.data(), C.size), "zk_witness_program_upload");
    program_resident = true;
}

void prover::rerunWitness(const vector<F> &picture, vector<u64> &ranges, size_t n_ranges, vector<F> &last_layer) {
    if (!program_resident) throw std::runtime_error("rerunWitness: no witness program on the GPU");
    ranges.assign(2 * n_ranges, 0);
    last_layer.assign(C.circuit[C.size - 1].size, F_ZERO);
    releaseHostValues();           // stale from here on
    check(zk_witness_rerun(ctx, U(picture[0]), picture.size(), reinterpret_cast<uint64_t *>(ranges.data()), (uint32_t) n_ranges, U(last_layer[0]), last_layer.size()), "zk_witness_rerun");
}

// prover::init (reference src/prover.cpp:17-21) + residency of C and val in HBM
void prover::init() {
    ensureContext();
    if (!resident) {
        if (circuit_gates_dropped(C, 0)) throw std::runtime_error("prover::init: this prover holds a structure copy of its circuit (a clone): nothing to upload");
        upload_timer.start();
        // a context holds one circuit; a changed circuit gets a fresh one
        vector<zk_layer_desc> desc = layerDescs();
        // (ZKCNN_CONV_HINTS=0: upload without the generator's hints, as an unmodified reference generator would -- the library then infers them)
        const char *he = std::getenv("ZKCNN_CONV_HINTS");
        const uint32_t n_hints = (he && std::atoi(he) == 0) ? 0u : (uint32_t) conv_hints.size();
        auto up = [&]() { return zk_upload_circuit_hinted(ctx, desc.data(), C.size, U(C.two_mul[0]), (int) C.two_mul.size(), conv_hints.data(), n_hints); };
        if (up() != ZK_OK) {
            // the context already holds another circuit: start over with a new one
            string first_err = zk_last_error(ctx);
            poly_p.reset();
            zk_ctx_destroy(ctx);
            ctx = nullptr;
            program_resident = false;
            ensureContext();
            if (up() != ZK_OK)
                throw std::runtime_error("zk_upload_circuit failed: " + string(zk_last_error(ctx)) + " / " + first_err);
        }
        if ((int) val.size() != C.size) throw std::runtime_error("prover::init: no host values to upload (released after an earlier upload)");
        for (int i = 0; i < C.size; ++i)
            check(zk_upload_layer_values(ctx, i, val[i].empty() ? nullptr : U(val[i][0]), val[i].size()), "zk_upload_layer_values");
        resident = true;
        upload_timer.stop();
    }
    prove_timer.clear();
    check(zk_prover_init(ctx), "zk_prover_init");
}

double prover::proofSize() const { return ctx ? (double) zk_proof_bytes(ctx) / 1024.0 : 0.0; }

void prover::sumcheckInitAll(const vector<F>::const_iterator &r_0_from_v) {
    TIMED();
    prove_timer.start();
    check(zk_sumcheck_init_all(ctx, U(*r_0_from_v), (u32) C.circuit[C.size - 1].bit_length), "zk_sumcheck_init_all");
    prove_timer.stop();
}
void prover::sumcheckInit(const F &alpha_0, const F &beta_0) {
    TIMED();
    prove_timer.start();
    check(zk_sumcheck_init(ctx, U(alpha_0), U(beta_0)), "zk_sumcheck_init");
    zkSetWeights(alpha_0, beta_0);
    prove_timer.stop();
}
void prover::sumcheckDotProdInitPhase1() {
    TIMED();
    prove_timer.start();
    check(zk_sumcheck_dotprod_init_phase1(ctx), "zk_sumcheck_dotprod_init_phase1");
    zkBeginInstance();
    prove_timer.stop();
}
void prover::sumcheckInitPhase1(const F &relu_rou_0) {
    TIMED();
    prove_timer.start();
    check(zk_sumcheck_init_phase1(ctx, U(relu_rou_0)), "zk_sumcheck_init_phase1");
    zkBeginInstance();
    prove_timer.stop();
}
void prover::sumcheckInitPhase2() {
    TIMED();
    prove_timer.start();
    check(zk_sumcheck_init_phase2(ctx), "zk_sumcheck_init_phase2");
    prove_timer.stop();
}
cubic_poly prover::sumcheckDotProdUpdate1(const F &previous_random) {
    TIMED();
    prove_timer.start();
    F o[4];
    check(zk_sumcheck_dotprod_update1(ctx, U(previous_random), U(o[0])), "zk_sumcheck_dotprod_update1");
    cubic_poly poly(o[0], o[1], o[2], o[3]);
    zkMask(poly, previous_random);
    prove_timer.stop();
    return poly;
}
quadratic_poly prover::sumcheckUpdate1(const F &previous_random) {
    TIMED();
    prove_timer.start();
    F o[3];
    check(zk_sumcheck_update1(ctx, U(previous_random), U(o[0])), "zk_sumcheck_update1");
    quadratic_poly poly(o[0], o[1], o[2]);
    zkMask(poly, previous_random);
    prove_timer.stop();
    return poly;
}
quadratic_poly prover::sumcheckUpdate2(const F &previous_random) {
    TIMED();
    prove_timer.start();
    F o[3];
    check(zk_sumcheck_update2(ctx, U(previous_random), U(o[0])), "zk_sumcheck_update2");
    quadratic_poly poly(o[0], o[1], o[2]);
    zkMask(poly, previous_random);
    prove_timer.stop();
    return poly;
}
F prover::Vres(const vector<F>::const_iterator &r, u32 output_size, u8 r_size) {
    TIMED();
    prove_timer.start();
    F out;
    F dummy = F_ZERO;
    check(zk_vres(ctx, r_size ? U(*r) : U(dummy), output_size, r_size, U(out)), "zk_vres");
    prove_timer.stop();
    return out;
}
void prover::sumcheckDotProdFinalize1(const F &previous_random, F &claim_1) {
    TIMED();
    prove_timer.start();
    check(zk_sumcheck_dotprod_finalize1(ctx, U(previous_random), U(claim_1)), "zk_sumcheck_dotprod_finalize1");
    if (zkActive()) {            // zero-knowledge mode: the claim leaves masked, phase 2 is about the masked value (zk_mask.hpp (2))
        F none = F_ZERO, added[2];
        zkMaskClaims(zkmask::SLOT_U0, previous_random, none, claim_1, added);
        check(zk_sumcheck_claims_adjust(ctx, U(added[0]), U(added[1])), "zk_sumcheck_claims_adjust");
    }
    prove_timer.stop();
}
void prover::sumcheckFinalize1(const F &previous_random, F &claim_0, F &claim_1) {
    TIMED();
    prove_timer.start();
    check(zk_sumcheck_finalize1(ctx, U(previous_random), U(claim_0), U(claim_1)), "zk_sumcheck_finalize1");
    if (zkActive()) {
        F added[2];
        zkMaskClaims(zkmask::SLOT_U0, previous_random, claim_0, claim_1, added);
        check(zk_sumcheck_claims_adjust(ctx, U(added[0]), U(added[1])), "zk_sumcheck_claims_adjust");
    }
    prove_timer.stop();
}
void prover::sumcheckFinalize2(const F &previous_random, F &claim_0, F &claim_1) {
    TIMED();
    prove_timer.start();
    check(zk_sumcheck_finalize2(ctx, U(previous_random), U(claim_0), U(claim_1)), "zk_sumcheck_finalize2");
    if (zkActive()) {
        F added[2];
        zkMaskClaims(zkmask::SLOT_V0, previous_random, claim_0, claim_1, added);
    }
    prove_timer.stop();
}
void prover::sumcheckLiuFinalize(const F &previous_random, F &claim_1) {
    TIMED();
    prove_timer.start();
    check(zk_sumcheck_liu_finalize(ctx, U(previous_random), U(claim_1)), "zk_sumcheck_liu_finalize");
    zkMaskInputClaim(previous_random, claim_1);
    prove_timer.stop();
}
void prover::sumcheckLiuInit(const vector<F> &s_u, const vector<F> &s_v) {
    TIMED();
    prove_timer.start();
    check(zk_sumcheck_liu_init(ctx, U(s_u[0]), U(s_v[0]), (u32) s_u.size()), "zk_sumcheck_liu_init");
    zkBeginLiu(s_u, s_v);
    prove_timer.stop();
}
quadratic_poly prover::sumcheckLiuUpdate(const F &previous_random) {
    TIMED();
    prove_timer.start();
    F o[3];
    check(zk_sumcheck_liu_update(ctx, U(previous_random), U(o[0])), "zk_sumcheck_liu_update");
    quadratic_poly poly(o[0], o[1], o[2]);
    zkMask(poly, previous_random);
    prove_timer.stop();
    return poly;
}

// ---- zero-knowledge mode (zk_mask.hpp): the backend side of a phase's last round ----
void prover::zkRawRound(int kind, const F &prev_r, F c[5]) {
    if (kind == zkmask::PH_DOT1) {
        const cubic_poly q = sumcheckDotProdUpdate1(prev_r);
        c[0] = q.d; c[1] = q.c; c[2] = q.b; c[3] = q.a;
        return;
    }
    const quadratic_poly q = kind == zkmask::PH_ONE ? sumcheckUpdate1(prev_r) : kind == zkmask::PH_TWO ? sumcheckUpdate2(prev_r) : sumcheckLiuUpdate(prev_r);
    c[0] = q.c; c[1] = q.b; c[2] = q.a;
}
void prover::zkTailPairs(F A[6]) { check(zk_sumcheck_tail_pairs(ctx, U(A[0])), "zk_sumcheck_tail_pairs"); }
// the last pairs of a phase are on the host when its last rounds run there (hybrid tail) -- and no resident kernel holds them
void prover::zkModeOn() { if (ctx) check(zk_set_host_tail(ctx, 5), "zk_set_host_tail"); }

// reference src/prover.cpp:503-511. The HBM copy of val[0] is already zero padded to 2^bit_length.
hyrax_bls12_381::polyProverBase &prover::commitInput(const vector<G> &gens) {
    TIMED();
    if (!ctx || !resident) throw std::runtime_error("prover::commitInput before prover::init");
    zkReset();
    poly_p.reset(new hyrax_bls12_381::polyProver(ctx, C.circuit[0].bit_length, gens, &gens_cache));
    return *poly_p;
}

hyrax_bls12_381::polyProverBase &prover::commitInputZk(const vector<G> &gens) {
    TIMED();
    if (!ctx || !resident) throw std::runtime_error("prover::commitInputZk before prover::init");
    zkReset();
    const int bl = C.circuit[0].bit_length;
    const vector<F> blinds = zkDrawBlinds((size_t) 1 << (bl >> 1));
    poly_p.reset(new hyrax_bls12_381::polyProver(ctx, bl, gens, &gens_cache, &blinds));
    return *poly_p;
}

// GKR prover with the reference's public interface (reference src/prover.hpp:18-49: same method
// names, argument meaning, return types, public data members C / val / prove_timer), implemented on
// the GPU: every method forwards to the C-ABI of include/zkcnn_hip.h (libzkcnn_hip.so). There is
// no CPU implementation behind it -- if the HIP library or a GPU is missing the calls throw.
//
// Differences a reference user sees: init() uploads C and val to HBM the first time it runs
// (residency; timed separately as uploadTime(), not prover time), and commitInput() returns the
// abstract polyProverBase& the verifier needs instead of the concrete class.
#pragma once
#include <memory>
#include "circuit.h"
#include "polynomial.h"
#include "polyProver.hpp"
#include "zk_mask.hpp"

class neuralNetwork;

class prover : public zkmask::proverMixin {
public:
    prover();
    explicit prover(int device);
    ~prover();
    prover(const prover &) = delete;
    prover &operator=(const prover &) = delete;

    void init();

    void sumcheckInitAll(const vector<F>::const_iterator &r_0_from_v);
    void sumcheckInit(const F &alpha_0, const F &beta_0);
    void sumcheckDotProdInitPhase1();
    void sumcheckInitPhase1(const F &relu_rou_0);
    void sumcheckInitPhase2();

    cubic_poly sumcheckDotProdUpdate1(const F &previous_random);
    quadratic_poly sumcheckUpdate1(const F &previous_random);
    quadratic_poly sumcheckUpdate2(const F &previous_random);

    F Vres(const vector<F>::const_iterator &r, u32 output_size, u8 r_size);

    void sumcheckDotProdFinalize1(const F &previous_random, F &claim_1);
    void sumcheckFinalize1(const F &previous_random, F &claim_0, F &claim_1);
    void sumcheckFinalize2(const F &previous_random, F &claim_0, F &claim_1);
    void sumcheckLiuFinalize(const F &previous_random, F &claim_1);

    void sumcheckLiuInit(const vector<F> &s_u, const vector<F> &s_v);
    quadratic_poly sumcheckLiuUpdate(const F &previous_random);

    hyrax_bls12_381::polyProverBase &commitInput(const vector<G> &gens);
    // zero-knowledge mode (additive; ZKCNN_MODE_ZK): gens = (g_0 .. g_{m-1}, H), every row commitment blinded; the round polynomials of all
    // following sumchecks are masked (zk_mask.hpp: zkMaskCommit / zkSetRho / zkMaskEval / zkMaskOpen1,2 come from the mixin)
    hyrax_bls12_381::polyProverBase &commitInputZk(const vector<G> &gens);

    timer prove_timer;
    double proveTime() const { return prove_timer.elapse_sec(); }
    double proofSize() const;                         // KB
    double polyProverTime() const { return poly_p ? poly_p->getPT() : 0.0; }
    double polyProofSize() const { return poly_p ? poly_p->getPS() : 0.0; }
    double uploadTime() const { return upload_timer.elapse_sec(); }

    layeredCircuit C;
    vector<vector<F>> val;        // the output of each gate (host copy; the GPU works on its HBM copy)

    // call after C / val changed on the host to force a fresh upload at the next init()
    void invalidateDevice() { resident = false; }
    int device() const { return device_id; }
    zk_ctx *context() const { return ctx; }     // for the profiler entry points of include/zkcnn_hip.h

    void ensureContext();                       // creates the GPU context (and its stream) without uploading anything
    // This prover becomes a CLONE of `parent`: a context of its own on parent's resident circuit with a copy of parent's witness (include/zkcnn_hip.h:
    // zk_ctx_clone), C = the structure copy of parent's circuit (no gate lists on the host). Nothing is generated, sorted or uploaded.
    void cloneFrom(prover &parent);
    // non-interactive mode: the verifier's challenge chain (8 state words, count of unhashed bytes); the GPU then runs the small rounds of
    // every phase by itself (include/zkcnn_hip.h: zk_fs_attach). NULL, NULL detaches.
    void attachFiatShamir(const uint32_t *state, const uint64_t *pending);
    void tailStats(uint64_t &rounds, uint64_t &phases) const;
    void setHostTail(int log_entries);          // hybrid tail (include/zkcnn_hip.h: zk_set_host_tail); -1 = the library's default, -2 = off
    uint64_t hostTailRounds() const;            // rounds it has run on the host so far
    void proofBegin();                          // bracket of one proof (include/zkcnn_hip.h: zk_proof_begin / zk_proof_end)
    void proofEnd();
    // out[i] = k_i * B through B's byte-window table on the GPU (include/zkcnn_hip.h: zk_fixed_base_mul): the verifier's generator loop of an in-process run
    bool fixedBaseMul(const std::vector<G1Affine> &table, const std::vector<uint64_t> &scalars, std::vector<G1> &out);
    void setGeneratorReuse(bool reusable);      // the commitment generators of the next proofs come again (public generators) or not (the reference's fresh ones): include/zkcnn_hip.h: zk_set_generator_reuse
    void setLiveRounds(bool on);                // resident round kernel of the interactive protocol (include/zkcnn_hip.h: zk_set_live_rounds)

    // ---- the next picture on the resident circuit (include/zkcnn_hip.h: zk_witness_program_upload / zk_witness_rerun). The program is
    // what the circuit generator recorded (host/neuralNetwork.hpp: witnessProgram, records layout-equal to the C-ABI's). After a rerun
    // the values in HBM belong to the new picture; the host copy `val` is dropped (nothing on the proving path reads it). ----
    void uploadWitnessProgram(const zk_witness_op *ops, size_t n_ops, const u32 *windows, size_t n_windows, const zk_witness_step *steps, size_t n_steps);
    bool hasWitnessProgram() const { return program_resident; }
    // ranges: 2 per RANGE step (largest non-negative value, largest magnitude of a negative one); last_layer: values of the output layer
    void rerunWitness(const vector<F> &picture, vector<u64> &ranges, size_t n_ranges, vector<F> &last_layer);
    void releaseHostValues() { vector<vector<F>>().swap(val); }
    // optional: what the circuit generator knows about its direct-convolution layers (include/zkcnn_hip.h: zk_conv_hint; checked against the
    // gate lists at upload). Must be set before the first init().
    void setConvHints(const zk_conv_hint *hints, size_t n) { conv_hints.assign(hints, hints + n); }
    int structuredLayers() const { return ctx ? zk_structured_layers(ctx) : 0; }
    int factoredDotLayers() const { return ctx ? zk_factored_dot_layers(ctx) : 0; }
    uint64_t dotDeferredPhases() const { return ctx ? zk_dot_deferred_phases(ctx) : 0; }
private:
    vector<zk_layer_desc> layerDescs() const;
    bool program_resident = false;
    vector<zk_conv_hint> conv_hints;
    hyrax_bls12_381::polyProverBase &zkBackend() override { return *poly_p; }
    const layeredCircuit &zkCircuit() const override { return C; }
    void zkRawRound(int kind, const F &prev_r, F c[5]) override;
    void zkTailPairs(F A[6]) override;                                  // include/zkcnn_hip.h: zk_sumcheck_tail_pairs
    void zkModeOn() override;
    void check(int rc, const char *what) const;

    zk_ctx *ctx;
    int device_id;
    bool resident;
    timer upload_timer;
    std::unique_ptr<hyrax_bls12_381::polyProver> poly_p;
    hyrax_bls12_381::polyProver::gensCache gens_cache;      // affine form of the last generator set (re-used generators)
    friend neuralNetwork;
};

// session.hpp calls this for whatever prover type it drives; only the HIP-backed prover can continue the chain on its own
inline void attachFsChain(prover &p, const uint32_t *state, const uint64_t *pending) { p.attachFiatShamir(state, pending); }
inline void setHostTail(prover &p, int log_entries) { p.setHostTail(log_entries); }
inline void setLiveRounds(prover &p, bool on) { p.setLiveRounds(on); }
inline void setGeneratorReuse(prover &p, bool reusable) { p.setGeneratorReuse(reusable); }
inline bool fixedBaseMul(prover &p, const std::vector<G1Affine> &table, const std::vector<uint64_t> &scalars, std::vector<G1> &out) { return p.fixedBaseMul(table, scalars, out); }
inline void proofBegin(prover &p) { p.proofBegin(); }
inline void proofEnd(prover &p) { p.proofEnd(); }
template <class H> inline void setConvHints(prover &p, const std::vector<H> &hints) {
    static_assert(sizeof(H) == sizeof(zk_conv_hint), "convolution hints cross the C-ABI as they are");
    p.setConvHints(reinterpret_cast<const zk_conv_hint *>(hints.data()), hints.size());
}

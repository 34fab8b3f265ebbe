// One circuit + witness kept alive across proofs; shared by the product driver (api.cpp, HIP
// prover) and the oracle driver (oracle/oracle_api.cpp, CPU prover). Mirrors the reference main()
// flow (reference src/main_demo_lenet.cpp:19-41).
#pragma once
#include <chrono>
#include <memory>
#include "../../../include/zkcnn_api.h"
#include "models.hpp"
#include "replay.hpp"
#include "fiber.hpp"
#include "../ff/hash_to_curve.hpp"

// provers that can continue the Fiat-Shamir chain themselves (the HIP-backed one: prover.hpp) overload this; everybody else ignores it
template <class P> inline void attachFsChain(P &, const uint32_t *, const uint64_t *) {}
template <class P> inline void setHostTail(P &, int) {}
template <class P> inline void setLiveRounds(P &, bool) {}
template <class P> inline void setGeneratorReuse(P &, bool) {}
template <class P> inline void proofBegin(P &) {}
template <class P> inline void proofEnd(P &) {}
template <class P, class H> inline void setConvHints(P &, const std::vector<H> &) {}

template <class ProverT>
struct sessionT {
    sessionT() {}
    template <class A> explicit sessionT(const A &prover_arg) : p(prover_arg) {}      // e.g. the GPU index of the HIP-backed prover
    ProverT p;
    std::shared_ptr<neuralNetwork> nn;     // (shared by the clones of a session: everything a proof or a new picture asks of it is const)
    std::vector<G1> gens;
    string model_name;
    int pic_cnt = 1, pic_x = 0, pic_y = 0, pic_channel = 0;
    uint64_t data_seed = 0;
    double witness_s = 0;
    string row;
    witnessAccel *accel = nullptr;     // set by the product driver while the witness is generated
    verifierAccel *vaccel = nullptr;   // set by the product driver: wiring predicates on the GPU

    static double now() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

    // a model descriptor may come from an untrusted proof file: bound it before anything is sized from it
    static bool sane(const zkcnn_model_desc *d) {
        return d->pic_x >= 1 && d->pic_x <= 1024 && d->pic_y >= 1 && d->pic_y <= 1024 && d->pic_channel >= 1 && d->pic_channel <= 64 &&
               d->pic_cnt >= 1 && d->pic_cnt <= 256;
    }

    bool has_witness = true;           // false: a verifier-only session (circuit rebuilt from a statement, nothing to prove)

    // statement = the quantisation scales recorded while the circuit was built; with them, buildStatement() reproduces the circuit
    // from the model descriptor alone (no picture, weights or witness)
    const vector<int> &statementScales() const { return nn->scales(); }
    bool buildStatement(const zkcnn_model_desc *d, const int32_t *scales, uint64_t n) {
        model_name = d->model ? d->model : "";
        pic_cnt = d->pic_cnt; pic_x = d->pic_x; pic_y = d->pic_y; pic_channel = d->pic_channel;
        if (!sane(d)) return false;
        nn.reset(makeModel(model_name, d->pic_x, d->pic_y, d->pic_channel, d->pic_cnt));
        if (!nn) return false;
        nn->setStructureOnly(vector<int>(scales, scales + n));
        nn->create(p, false);
        setConvHints(p, nn->convHints());
        if (!nn->scalesConsumed()) return false;           // a statement with more scales than the model asks for is malformed
        has_witness = false;
        return true;
    }

    // calibrated: not NULL = the quantisation scales are given (those of an earlier session of the model); the build fails if this picture's
    // values do not fit them
    bool build(const zkcnn_model_desc *d, const vector<int> *calibrated = nullptr) {
        model_name = d->model ? d->model : "";
        pic_cnt = d->pic_cnt; pic_x = d->pic_x; pic_y = d->pic_y; pic_channel = d->pic_channel;
        if (!sane(d)) return false;
        nn.reset(makeModel(model_name, d->pic_x, d->pic_y, d->pic_channel, d->pic_cnt));
        if (!nn) return false;
        data_seed = d->data_seed;
        nn->useSyntheticData(d->data_seed, d->picture_seed);
        nn->setWitnessAccel(accel);
        if (calibrated) nn->setFixedScales(*calibrated);
        double t0 = now();
        nn->create(p, false);
        witness_s = now() - t0;
        if (nn->scaleOverflow() || !nn->fixedScalesConsumed()) return false;
        setConvHints(p, nn->convHints());        // a prover that knows the pattern of a direct convolution factors its gate sums (checked at upload)
        return true;
    }

    // common set-up of a verifier run: challenge source (OS CSPRNG, seeded stream, or the transcript itself), generators, options
    // for_prover: the run drives this session's own prover (prove()); only then may the prover be handed the Fiat-Shamir chain -- a replay
    // (verifyProof) must never leave pointers into its own locals in the prover's context
    template <class V>
    void configure(V &v, uint64_t challenge_seed, uint32_t mode, fiatShamir &fs, std::unique_ptr<challengeScope> &scope, bool for_prover) {
        const bool fiat = (mode & ZKCNN_MODE_FIAT_SHAMIR) != 0;
        // generators both sides know in advance must not have a known discrete logarithm: hash-to-curve (ff/hash_to_curve.hpp).
        // Only the in-process interactive run keeps the reference's k_i * G, with k_i drawn by the verifier and never shown to the prover.
        const bool public_gens = fiat || (mode & ZKCNN_MODE_REUSE_GENS);
        const bool zk = (mode & ZKCNN_MODE_ZK) != 0;
        const u8 logn = p.C.circuit[0].bit_length;
        const size_t n_sqrt = ((size_t) 1 << (logn - (logn >> 1))) + (zk ? 1 : 0);       // zero-knowledge mode: one more generator, H
        // (test hook, ZKCNN_TEST_HOOKS=1: ZKCNN_TEST_COIN_SALT=<n> seeds the prover's coins apart from the challenges -- the same challenges with other
        //  coins is how tests/test_zk_cpu.py shows that every message of the zero-knowledge mode depends on the coins)
        uint64_t coin_salt = 0;
        if (const char *hk = std::getenv("ZKCNN_TEST_HOOKS"))
            if (std::atoi(hk) != 0)
                if (const char *cs = std::getenv("ZKCNN_TEST_COIN_SALT")) coin_salt = std::strtoull(cs, nullptr, 0);
        if (mode & ZKCNN_MODE_SEEDED) { Fr::seedCSPRNG(challenge_seed); zkff::privateCoins().seed(challenge_seed ^ coin_salt); }
        else { Fr::useOsRandom(); zkff::privateCoins().useOsRandom(); }
        v.zk = zk;
        // ZKCNN_MODE_HOST_ROUNDS: every sumcheck round is a kernel launch driven from here (no resident round kernel, no device-side Fiat-Shamir rounds)
        setLiveRounds(p, !(mode & ZKCNN_MODE_HOST_ROUNDS));
        // hybrid tail: tables of <= 64 entries finish their phase on the host; otherwise the library's default (lanes of a batch: <= 32 entries) or none at all
        setHostTail(p, (mode & ZKCNN_MODE_HOST_TAIL) ? 6 : (mode & ZKCNN_MODE_GPU_TAIL) ? -2 : -1);
        if (for_prover) setGeneratorReuse(p, public_gens);     // fresh generators (the reference's semantics) die with the proof: no byte table for them
        const zkff::publicGenerators *pg = nullptr;
        if (public_gens) {
            pg = &zkff::publicGeneratorSet(n_sqrt);
            if (gens.size() != n_sqrt) gens = pg->gens;
        }
        v.drive_only = (mode & ZKCNN_MODE_DRIVE_ONLY) != 0;
        if (public_gens) v.fixed_gens = &gens;
        if (mode & ZKCNN_MODE_TAMPER) v.tamper_at = (long) ((mode >> 8) & 0xffffu);
        if (vaccel && !(mode & ZKCNN_MODE_HOST_PRED)) v.accel = vaccel;
        v.cross_check = (mode & ZKCNN_MODE_CROSS_PRED) != 0;
        v.full_ipa = (mode & ZKCNN_MODE_FULL_IPA) != 0;
        if (const char *hg = std::getenv("ZKCNN_HOST_GENERATORS")) v.host_generators = std::atoi(hg) != 0;       // (A/B switch: the generator loop on host threads, as through round 5)
        if (fiat) {
            // non-interactive: every challenge is a hash of the statement (model, picture shape, quantisation scales, every layer's
            // shape, a digest of the wiring, the generators) and of all messages received so far
            zkcnn_model_desc d;
            d.model = model_name.c_str();
            d.pic_x = pic_x; d.pic_y = pic_y; d.pic_channel = pic_channel; d.pic_cnt = pic_cnt; d.data_seed = 0; d.picture_seed = 0;
            fs.absorbStatement(d, nn->scales(), p.C);
            fs.absorb(pg->digest, 32);
            v.transcript.tap = &fs;
            v.lazy_challenges = true;
            scope.reset(new challengeScope(&fs));
            // By default the rounds run like the interactive protocol's (resident kernels for a lone proof, launches otherwise) and the verifier
            // object here derives every challenge from the transcript on the host. ZKCNN_MODE_FS_DEVICE: the prover runs the small rounds of
            // every phase ahead of the verifier, hashing on the device -- the challenges are a function of the transcript. Not with masked round
            // polynomials (the masks are added on the host) and not when a message is corrupted on purpose.
            if (for_prover && (mode & ZKCNN_MODE_FS_DEVICE) && !zk && !(mode & (ZKCNN_MODE_TAMPER | ZKCNN_MODE_HOST_ROUNDS)))
                attachFsChain(p, fs.stateWords(), fs.pendingBytes());
        }
    }

    int prove(uint64_t challenge_seed, uint32_t mode, uint8_t *transcript, uint64_t cap, zkcnn_result *out) {
        std::memset(out, 0, sizeof(*out));
        if (!has_witness) {
            std::snprintf(out->message, sizeof(out->message), "no valid witness: a verifier-only session, or the last new_image call did not succeed");
            return -3;
        }
        double t0 = now();
        const bool drive = mode & ZKCNN_MODE_DRIVE_ONLY;
        output_tb.assign(OUT_COLUMN_CNT, "");
        output_tb[MO_INFO_OUT_ID] = model_name;
        output_tb[PCNT_OUT_ID] = std::to_string(pic_cnt);
        output_tb[WS_OUT_ID] = std::to_string(p.C.circuit[0].size) + "(2^" + std::to_string((int) p.C.circuit[0].bit_length) + ")";
        verifierT<ProverT> v(&p, p.C);
        fiatShamir fs;
        std::unique_ptr<challengeScope> scope;
        attachFsChain(p, nullptr, nullptr);            // whatever an earlier run left behind
        struct scope { ProverT &q; explicit scope(ProverT &x) : q(x) { proofBegin(q); } ~scope() { proofEnd(q); } } active(p);      // this proof counts as in flight on its GPU
        bool ok = false;
        try {
            configure(v, challenge_seed, mode, fs, scope, true);
            ok = v.verify();
        } catch (...) {
            attachFsChain(p, nullptr, nullptr);
            scope.reset();
            throw;
        }
        attachFsChain(p, nullptr, nullptr);
        scope.reset();
        out->accepted = drive ? -1 : (ok ? 1 : 0);
        out->n_layers = p.C.size;
        out->input_size = p.C.circuit[0].size;
        out->input_bits = p.C.circuit[0].bit_length;
        out->prove_s = p.proveTime();
        out->poly_prove_s = p.polyProverTime();
        out->verify_s = v.verifierTime();
        out->poly_verify_s = v.polyVerifierTime();
        out->proof_kb = p.proofSize();
        out->poly_proof_kb = p.polyProofSize();
        out->witness_s = witness_s;
        out->transcript_len = v.transcript.bytes.size();
        out->n_messages = (int32_t) v.msg_count;
        if (transcript && cap) std::memcpy(transcript, v.transcript.bytes.data(), std::min<uint64_t>(cap, out->transcript_len));
        int rounds = 0;
        u64 nu = 0, nb = 0, tbl = 0;
        for (int i = 0; i < p.C.size; ++i) {
            const layer &L = p.C.circuit[i];
            if (i) { rounds += L.max_bl_u + L.max_bl_v; nu += L.uniCount(); nb += L.binCount(); }
            tbl += (u64) 1 << L.bit_length;
        }
        out->n_rounds = rounds + p.C.circuit[0].bit_length;
        out->gate_cnt_uni = nu;
        out->gate_cnt_bin = nb;
        out->table_entries = tbl;
        std::snprintf(out->message, sizeof(out->message), "%s", ok ? "" : v.failure());
        row.clear();
        for (auto &s : output_tb) { row += s; row += ", "; }
        out->wall_s = now() - t0;
        return 0;
    }

    // Checks a serialized proof (the canonical transcript prove() returns) WITHOUT the prover: the same verifier runs against a
    // replayProver that parses its answers from the bytes. challenge_seed / mode must be the ones the proof was made with
    // (the seed is ignored under ZKCNN_MODE_FIAT_SHAMIR). Uses this session's circuit as the statement.
    int verifyProof(const uint8_t *proof, uint64_t len, uint64_t challenge_seed, uint32_t mode, zkcnn_result *out) {
        std::memset(out, 0, sizeof(*out));
        double t0 = now();
        mode &= ~(uint32_t) (ZKCNN_MODE_DRIVE_ONLY | ZKCNN_MODE_TAMPER);
        bool ok = false;
        string why;
        if (!(mode & (ZKCNN_MODE_FIAT_SHAMIR | ZKCNN_MODE_SEEDED))) {
            // the challenges of an interactive transcript were the verifier's own coins; off line there is nothing to check them against
            out->accepted = 0;
            std::snprintf(out->message, sizeof(out->message), "not a non-interactive proof: only Fiat-Shamir proofs verify off line (seeded replay: ZKCNN_MODE_SEEDED)");
            out->wall_s = now() - t0;
            return 0;
        }
        try {
            replayProver rp(proof, len, &p.C);
            verifierT<replayProver> v(&rp, p.C);
            fiatShamir fs;
            std::unique_ptr<challengeScope> scope;
            configure(v, challenge_seed, mode, fs, scope, false);
            ok = v.verify();
            scope.reset();
            if (!ok) why = v.failure();
            else if (!rp.consumedAll()) { ok = false; why = "trailing bytes after the last message"; }
            else if (v.transcript.bytes.size() != len || std::memcmp(v.transcript.bytes.data(), proof, len) != 0) {
                ok = false;
                why = "proof is not in canonical form";
            }
            out->verify_s = v.verifierTime();
            out->poly_verify_s = v.polyVerifierTime();
            out->n_messages = (int32_t) v.msg_count;
        } catch (const std::exception &e) {
            ok = false;
            why = e.what();
        }
        out->accepted = ok ? 1 : 0;
        out->n_layers = p.C.size;
        out->input_size = p.C.circuit[0].size;
        out->input_bits = p.C.circuit[0].bit_length;
        out->transcript_len = len;
        std::snprintf(out->message, sizeof(out->message), "%s", why.c_str());
        out->wall_s = now() - t0;
        return 0;
    }
};

// ------------------------------------------------------------------------------------------------------------------------------------
// batchSession: K proofs in lock step, driven by ONE host thread.
//
// The reference's main() runs one verifier loop against one prover (reference src/main_demo_vgg.cpp:36-41 -> src/verifier.cpp:118-373).
// K sessions that prove K pictures on the same circuit make the same calls in the same order; here their K verifier loops -- the unchanged
// verifierT above, one per lane, each with its own challenge stream -- run as fibers of the calling thread (fiber.hpp). A lane runs until
// its prover would wait for the GPU (the HIP library's yield callback: include/zkcnn_hip.h, zk_batch_set_yield) or, for a prover without
// a GPU, until it ends; when every lane is parked `flush` issues what they deferred -- ONE kernel launch per round for all lanes -- and the
// lanes continue. Every protocol mode of sessionT::prove works unchanged (seeded / OS challenges, Fiat-Shamir on the host, masking,
// tampering), and each lane's transcript is byte for byte what it would be alone: the lanes share nothing but the thread.
// ------------------------------------------------------------------------------------------------------------------------------------

// the thread-local protocol state of one proof (challenge stream, Fiat-Shamir override, the prover's private coins, the result row): kept
// per lane and exchanged with the thread's whenever a lane gets or gives up the thread
struct laneState {
    zkff::Xoshiro challenge;
    bool seeded = false;
    zkff::ChallengeSource *override_src = nullptr;
    zkff::PrivateCoins coins;
    vector<string> row;
    laneState() : row(OUT_COLUMN_CNT, "") { challenge.seed(0); }
    void exchange() {
        std::swap(challenge, zkff::challengeStream());
        std::swap(seeded, zkff::challengeSeeded());
        std::swap(override_src, zkff::challengeOverride());
        std::swap(coins, zkff::privateCoins());
        row.swap(output_tb);
    }
};

template <class SessionT>
struct batchSessionT {
    std::vector<SessionT *> lanes;          // not owned
    uint64_t rounds_of_flushes = 0;         // passes of the driver loop (each ends in one flush)

    // One proof per lane. seeds / transcripts / caps / out: one entry per lane (transcripts may be NULL, or hold NULL entries).
    // flush: called whenever every unfinished lane is parked. Returns 0, or the first lane's non-zero return code (every lane's
    // result is filled in either way; an exception inside a lane ends that lane only: code -2, message in out[k]).
    template <class Flush>
    int prove(const uint64_t *seeds, uint32_t mode, uint8_t *const *transcripts, const uint64_t *caps, zkcnn_result *out, Flush flush) {
        const size_t K = lanes.size();
        std::vector<int> rc(K, 0);
        std::vector<laneState> st(K);
        std::vector<std::unique_ptr<zkfiber::fiber>> fb(K);
        for (size_t k = 0; k < K; ++k) {
            SessionT *s = lanes[k];
            int *code = &rc[k];
            zkcnn_result *res = &out[k];
            uint8_t *tr = transcripts ? transcripts[k] : nullptr;
            const uint64_t cap = (tr && caps) ? caps[k] : 0, seed = seeds ? seeds[k] : 0;
            fb[k].reset(new zkfiber::fiber([s, code, res, tr, cap, seed, mode]() {
                try {
                    *code = s->prove(seed, mode, tr, cap, res);
                } catch (const std::exception &e) {
                    std::memset(res, 0, sizeof(*res));
                    std::snprintf(res->message, sizeof(res->message), "%s", e.what());
                    *code = -2;
                }
            }));
        }
        for (;;) {
            bool any = false;
            for (size_t k = 0; k < K; ++k) {
                if (fb[k]->done()) continue;
                st[k].exchange();
                fb[k]->resume();                       // (the lane's function catches everything: nothing propagates from here)
                st[k].exchange();
                any = any || !fb[k]->done();
            }
            ++rounds_of_flushes;
            flush();
            if (!any) break;
        }
        for (size_t k = 0; k < K; ++k) if (rc[k]) return rc[k];
        return 0;
    }
};

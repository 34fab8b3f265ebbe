// Serialized proofs: the canonical transcript the verifier records (verifier.hpp: proofTranscript) is the wire
// form of a proof. Two pieces make it checkable without the prover:
//   * replayProver: an object with the `prover` interface whose answers are parsed, in order, from a transcript. The unchanged
//     verifier template runs against it (verifierT<replayProver>), so every check of the interactive protocol is applied to the
//     bytes; the challenges come from the same seeded stream (interactive mode) or from the transcript itself:
//   * fiatShamir: a ChallengeSource that hashes everything the verifier has received so far (a BLAKE2s chain), which
//     turns the same protocol into a non-interactive proof. Optional; the interactive path is untouched.
// The reference has neither (its proof never leaves the process); nothing here is on the hot path.
#pragma once
#include <stdexcept>
#include "../ff/sha256.hpp"
#include "../ff/blake2s.hpp"
#include "../../../include/zkcnn_api.h"
#include "verifier.hpp"
#include "zk_mask.hpp"

class fiatShamir : public zkff::ChallengeSource, public transcriptTap {
public:
    // Challenge chain: state (32 bytes) <- BLAKE2s-256(state || everything absorbed since the last challenge), one step per challenge;
    // a draw that is not below r costs one more step with an empty message. A field element is absorbed as its 32 bytes in memory
    // (Montgomery limbs: a bijection of the element), a group element as its 48-byte compressed encoding. The state after a round
    // polynomial's challenge is all the GPU needs to continue the chain by itself (hip/fs_tail.cuh).
    fiatShamir() {
        zkff::Blake2s h;
        h.init();
        h.update("zkcnn-amd/fiat-shamir/v2", 24);
        uint8_t d[32];
        h.final(d);
        std::memcpy(st, d, 32);
    }
    void absorb(const void *data, size_t n) override {
        const uint8_t *p = static_cast<const uint8_t *>(data);
        pending.insert(pending.end(), p, p + n);
        pending_len = pending.size();
    }
    void absorbFr(const Fr &x) override { absorb(&x, 32); }
    bool rawMontgomery() const override { return true; }
    const uint32_t *stateWords() const { return st; }
    const uint64_t *pendingBytes() const { return &pending_len; }
    const std::vector<uint8_t> &pendingData() const { return pending; }       // what the next challenge will hash behind the state (tests: the statement's encoding)
    // binds the statement: the whole model descriptor, every quantisation scale (they fix gate exponents, layer sizes and Q_MAX), the
    // shape of every layer and a digest of the wiring (gate lists, subset maps): statements that differ anywhere get unrelated challenges
    void absorbStatement(const zkcnn_model_desc &d, const vector<int> &scales, const layeredCircuit &C) {
        const uint64_t ml = d.model ? std::strlen(d.model) : 0;
        absorb(&ml, 8);
        if (ml) absorb(d.model, ml);
        const int64_t shape[4] = {d.pic_x, d.pic_y, d.pic_channel, d.pic_cnt};
        absorb(shape, sizeof(shape));
        const uint64_t ns = scales.size();
        absorb(&ns, 8);
        for (int sc : scales) { const int64_t v = sc; absorb(&v, 8); }
        const uint64_t nl = C.size;
        absorb(&nl, 8);
        for (int i = 0; i < C.size; ++i) {
            const layer &L = C.circuit[i];
            int64_t rec[16] = {(int64_t) L.ty, L.size, L.bit_length, L.fft_bit_length, L.zero_start_id, (int64_t) L.uniCount(),
                               (int64_t) L.binCount(), L.size_u[0], L.size_u[1], L.size_v[0], L.size_v[1], L.bit_length_u[0],
                               L.bit_length_u[1], L.bit_length_v[0], L.bit_length_v[1], L.need_phase2};
            absorb(rec, sizeof(rec));
            uint8_t sb[32];
            L.scale.toBytesLE(sb);
            absorb(sb, 32);
        }
        absorb(C.wiringDigest(), 32);
    }
    void words(uint64_t out[4]) override {
        zkff::Blake2s h;
        h.init();
        h.update(st, 32);
        if (!pending.empty()) h.update(pending.data(), pending.size());
        uint8_t d[32];
        h.final(d);
        std::memcpy(st, d, 32);
        pending.clear();
        pending_len = 0;
        std::memcpy(out, st, 32);            // little-endian host: words as they lie in the state
    }
private:
    uint32_t st[8];
    std::vector<uint8_t> pending;
    uint64_t pending_len = 0;
};

// RAII: installs a challenge source for the current thread
struct challengeScope {
    zkff::ChallengeSource *prev;
    explicit challengeScope(zkff::ChallengeSource *s) : prev(zkff::challengeOverride()) { zkff::challengeOverride() = s; }
    ~challengeScope() { zkff::challengeOverride() = prev; }
};

class proofReader {
public:
    proofReader(const u8 *data, size_t n) : p(data), end(data + n) {}
    Fr fr() {
        need(32);
        Fr x;
        if (!Fr::fromBytesLE(x, p)) throw std::runtime_error("proof: field element not in canonical form");
        p += 32;
        return x;
    }
    G1 g1() {
        need(48);
        G1 x;
        if (!G1::deserialize(x, p, true)) throw std::runtime_error("proof: invalid group element");
        p += 48;
        return x;
    }
    bool done() const { return p == end; }
private:
    const u8 *p, *end;
    void need(size_t n) const { if ((size_t) (end - p) < n) throw std::runtime_error("proof: truncated"); }
};

class replayPolyProver : public hyrax_bls12_381::polyProverBase {
public:
    replayPolyProver(proofReader &reader, int input_bits, bool zk = false) : rd(reader) {
        const int rb = input_bits >> 1;
        cb = input_bits - rb;
        if (zk) zk_m = (size_t) 1 << cb;
        comm.resize((size_t) 1 << rb);
        for (auto &c : comm) c = rd.g1();
    }
    const std::vector<G1> &commitment() const override { return comm; }
    void openInit(const std::vector<Fr> &) override {}
    hyrax_bls12_381::ipaRoundMsg openRound() override {
        hyrax_bls12_381::ipaRoundMsg m;
        m.L = rd.g1(); m.R = rd.g1(); m.yL = rd.fr(); m.yR = rd.fr();
        ++rounds;
        return m;
    }
    void openFold(const Fr &) override {}
    std::vector<Fr> openFinal() override {
        std::vector<Fr> a((size_t) 1 << (cb - rounds));
        for (auto &x : a) x = rd.fr();
        return a;
    }
    double getPT() const override { return 0; }
    double getPS() const override { return 0; }
    // zero-knowledge opening: both prover messages come from the bytes
    hyrax_bls12_381::dotProofCommit zkOpenCommit(const std::vector<Fr> &, const std::vector<Fr> &) override { return readDot1(1); }
    hyrax_bls12_381::dotProofResponse zkOpenRespond(const Fr &) override { return readDot2(1); }
    hyrax_bls12_381::dotProofCommit readDot1(size_t rows) {
        hyrax_bls12_381::dotProofCommit m;
        m.delta.resize(rows);
        for (auto &d : m.delta) d = rd.g1();
        m.t = rd.fr();
        return m;
    }
    hyrax_bls12_381::dotProofResponse readDot2(size_t rows) {
        hyrax_bls12_381::dotProofResponse m;
        m.z.resize(rows * zk_m);
        for (auto &x : m.z) x = rd.fr();
        m.z_blind.resize(rows);
        for (auto &x : m.z_blind) x = rd.fr();
        return m;
    }
private:
    proofReader &rd;
    std::vector<G1> comm;
    int cb = 0, rounds = 0;
};

// the `prover` interface (reference src/prover.hpp:18-49) answered from a transcript
class replayProver {
public:
    replayProver(const u8 *data, size_t n, const layeredCircuit *statement) : C(statement), rd(data, n) {}

    void init() {}
    F Vres(const vector<F>::const_iterator &, u32, u8) { return rd.fr(); }
    void sumcheckInitAll(const vector<F>::const_iterator &) {}
    void sumcheckInit(const F &, const F &) {}
    void sumcheckDotProdInitPhase1() { ++inst; var = 0; }
    void sumcheckInitPhase1(const F &) { ++inst; var = 0; }
    void sumcheckInitPhase2() {}
    cubic_poly sumcheckDotProdUpdate1(const F &) { ++var; F a = rd.fr(), b = rd.fr(), c = rd.fr(), d = rd.fr(); return cubic_poly(a, b, c, d); }
    quadratic_poly sumcheckUpdate1(const F &) { return quad(); }
    quadratic_poly sumcheckUpdate2(const F &) { return quad(); }
    void sumcheckDotProdFinalize1(const F &, F &claim_1) { claim_1 = rd.fr(); }
    void sumcheckFinalize1(const F &, F &claim_0, F &claim_1) { claim_0 = rd.fr(); claim_1 = rd.fr(); }
    void sumcheckFinalize2(const F &, F &claim_0, F &claim_1) { claim_0 = rd.fr(); claim_1 = rd.fr(); }
    void sumcheckLiuInit(const vector<F> &, const vector<F> &) { ++inst; var = 0; }
    quadratic_poly sumcheckLiuUpdate(const F &) { return quad(); }
    void sumcheckLiuFinalize(const F &, F &claim_1) { claim_1 = rd.fr(); }
    hyrax_bls12_381::polyProverBase &commitInput(const std::vector<G1> &) {
        pp.reset(new replayPolyProver(rd, C->circuit[0].bit_length));
        return *pp;
    }
    // zero-knowledge mode: the same messages, parsed
    hyrax_bls12_381::polyProverBase &commitInputZk(const std::vector<G1> &) {
        pp.reset(new replayPolyProver(rd, C->circuit[0].bit_length, true));
        return *pp;
    }
    zkmask::maskCommitMsg zkMaskCommit() {
        pl = zkmask::plan(*C, pp->zkColumns());
        inst = -1;
        mask_rows = (pl.total + pp->zkColumns() - 1) / pp->zkColumns();
        zkmask::maskCommitMsg m;
        m.commit.resize(mask_rows);
        for (auto &c : m.commit) c = rd.g1();
        m.sums.resize(pl.items.size());
        for (auto &x : m.sums) x = rd.fr();
        return m;
    }
    void zkSetRho(const F &) {}
    F zkMaskEval() { return rd.fr(); }
    // the last round of a phase: the plan's degree + 1 coefficients, highest first
    zkmask::zkPoly zkLastRound(int, const F &) {
        zkmask::zkPoly q;
        q.deg = pl.items.at(inst).deg.at(var++);
        for (int e = q.deg; e >= 0; --e) q.c[e] = rd.fr();
        return q;
    }
    hyrax_bls12_381::dotProofCommit zkMaskOpen1(const vector<F> &) { return pp->readDot1(mask_rows - 1); }
    hyrax_bls12_381::dotProofResponse zkMaskOpen2(const F &) { return pp->readDot2(mask_rows - 1); }
    double proveTime() const { return 0; }
    double proofSize() const { return 0; }
    double polyProverTime() const { return 0; }
    double polyProofSize() const { return 0; }
    bool consumedAll() const { return rd.done(); }
private:
    const layeredCircuit *C;                  // the statement (only the size of the input layer is needed here)
    proofReader rd;
    std::unique_ptr<replayPolyProver> pp;
    size_t mask_rows = 0;
    zkmask::plan pl;
    int inst = -1, var = 0;
    quadratic_poly quad() { ++var; F a = rd.fr(), b = rd.fr(), c = rd.fr(); return quadratic_poly(a, b, c); }
};

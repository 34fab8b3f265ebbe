// Interactive GKR verifier: owns the protocol loop, draws every challenge, calls the prover once
// per sumcheck round and checks each layer. Protocol, challenge ORDER and acceptance conditions
// follow reference src/verifier.cpp:118-373 (the consumer side of the hot path; it stays on the
// CPU). It is a template over the prover type so that the same driver runs against the HIP-backed
// `prover` and against the CPU checker in oracle/; `verifier` below is the drop-in name.
//
// Additions over the reference: a transcript recorder (every value the prover returns, in call
// order, canonically serialised -- the reference has no serialised proof), an
// option to re-use generators, and a "drive only" mode that draws the same challenges and makes the
// same prover calls but skips the verifier's own (gate-walking) work so a bench can time the prover.
#pragma once
#include <stdexcept>
#include <thread>
#include "circuit.h"
#include "polynomial.h"
#include "utils.hpp"
#include "zk_mask.hpp"

// anything that wants to see every serialized message as it is recorded (the Fiat-Shamir state of replay.hpp)
struct transcriptTap {
    virtual ~transcriptTap() {}
    virtual void absorb(const void *data, size_t n) = 0;
    virtual void absorbFr(const Fr &x) = 0;        // a field element (the tap decides on its encoding)
};

struct proofTranscript : public hyrax_bls12_381::transcriptSink {
    std::vector<u8> bytes;
    transcriptTap *tap = nullptr;
    void put(const Fr &x) override {
        size_t o = bytes.size();
        bytes.resize(o + 32);
        x.toBytesLE(&bytes[o]);
        if (tap) {
            // the tap (and the GPU's copy of the chain, hip/fs_tail.cuh) hashes the limbs as they lie in memory: that is a function of the field
            // element only while every emitted value is fully reduced -- checked here, at the boundary, not assumed
            if (Fr::geMod(x.v)) throw std::runtime_error("transcript: a field element left the prover unreduced (limbs >= r)");
            tap->absorbFr(x);
        }
    }
    void put(const G1 &p) override {
        size_t o = bytes.size();
        bytes.resize(o + 48);
        p.serialize(&bytes[o]);
        if (tap) tap->absorb(&bytes[o], 48);
    }
    void put(const quadratic_poly &p) { put(p.a); put(p.b); put(p.c); }
    void put(const cubic_poly &p) { put(p.a); put(p.b); put(p.c); put(p.d); }
};

// random multiples of the base point (reference src/verifier.cpp:121-126: gens[i] = G * k_i with k_i from the CSPRNG), via a fixed-base table of
// byte windows -- 32 mixed additions per generator -- and a few helper threads: the scalars are drawn first, in order, from the caller's
// challenge stream (so the generators are what a sequential loop gives); the 4096 independent scalar multiplications of a vgg11 proof then take
// ~10 ms instead of ~250 ms on one core. (With several proofs driven by one host thread -- batchSessionT -- this loop was what bounded proofs
// with fresh generators: the GPU waited for the verifier.)
inline const std::vector<G1Affine> &generatorBaseTable() {
    static const std::vector<G1Affine> table = [] {      // table[w * 255 + d - 1] = d * 256^w * G
        std::vector<G1> jac(32 * 255);
        G1 base = G1::generator();
        for (int w = 0; w < 32; ++w) {
            jac[w * 255] = base;
            for (int d = 2; d <= 255; ++d) G1::add(jac[w * 255 + d - 1], jac[w * 255 + d - 2], base);
            G1 nb;
            G1::add(nb, jac[w * 255 + 254], base);
            base = nb;
        }
        std::vector<G1Affine> t;
        zkff::batchToAffine(jac, t);
        return t;
    }();
    return table;
}
// `device` (optional): hands (table, scalars) to the prover's GPU -- fixedBaseMul below -- and returns false if there is none: an in-process run whose prover
// sits on a GPU does the 4 096 multiplications there (0.9 ms, fused over the lanes of a lock-step batch), any other verifier on its own cores.
// The points are the same group elements either way (Jacobian coordinates may differ; nothing reads them but group operations).
template <class Device>
inline void drawGenerators(std::vector<G1> &gens, size_t count, Device device) {
    const std::vector<G1Affine> &table = generatorBaseTable();
    std::vector<uint64_t> e(4 * count);
    for (size_t i = 0; i < count; ++i) {
        Fr k;
        k.setByCSPRNG();
        k.toCanonical(&e[4 * i]);
    }
    if (device(table, e, gens) && gens.size() == count) return;
    gens.resize(count);
    auto work = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            G1 acc;
            for (int w = 0; w < 32; ++w) {
                const unsigned d = (unsigned) (e[4 * i + (w >> 3)] >> ((w & 7) * 8)) & 255u;
                if (d) G1::addMixed(acc, acc, table[(size_t) w * 255 + d - 1]);
            }
            gens[i] = acc;
        }
    };
    size_t nt = std::min<size_t>(std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())), count / 256);
    if (nt <= 1) { work(0, count); return; }
    std::vector<std::thread> th;
    const size_t per = (count + nt - 1) / nt;
    for (size_t t = 1; t < nt; ++t) th.emplace_back(work, std::min(count, t * per), std::min(count, (t + 1) * per));
    work(0, std::min(count, per));
    for (auto &x : th) x.join();
}
inline void drawGenerators(std::vector<G1> &gens, size_t count) {
    drawGenerators(gens, count, [](const std::vector<G1Affine> &, const std::vector<uint64_t> &, std::vector<G1> &) { return false; });
}
// a prover type without a GPU behind it (the replay of a recorded proof, the oracle's prover): the verifier multiplies on the host
template <class P> inline bool fixedBaseMul(P &, const std::vector<G1Affine> &, const std::vector<uint64_t> &, std::vector<G1> &) { return false; }

// Optional accelerator for the verifier's wiring predicates (the only part of the verifier that walks every gate:
// reference src/verifier.cpp:36-116 and :304-325, `total_slow_timer`). The product driver plugs the HIP implementation
// in (include/zkcnn_hip.h: zk_verifier_*); without one the loops below run on the host as in the reference.
struct verifierAccel : public hyrax_bls12_381::msmAccel {
    virtual ~verifierAccel() {}
    bool msm(G1 &, const Fr *, const G1Affine *, size_t, bool) override { return false; }     // default: the host Pippenger
    virtual void predicates(u8 layer, const F *r_0, const F *r_1, const F &alpha, const F &beta, const F &relu_rou, const F *r_u,
                            const F *r_v, const F *r_u2, const F *r_v2, F uni[2], F bin[3]) = 0;
    virtual F inputPredicate(const vector<F> &r_u0, const vector<vector<F>> &r_u, const vector<vector<F>> &r_v, const vector<F> &sig_u,
                             const vector<F> &sig_v) = 0;
};

template <class ProverT>
class verifierT {
public:
    ProverT *p;
    const layeredCircuit &C;

    verifierT(ProverT *pr, const layeredCircuit &cir) : p(pr), C(cir), poly_v(nullptr) {
        final_claim_u0.assign(C.size + 2, F_ZERO);
        final_claim_v0.assign(C.size + 2, F_ZERO);
        r_u.assign(C.size + 2, vector<F>());
        r_v.assign(C.size + 2, vector<F>());
        p->init();
    }
    ~verifierT() { delete poly_v; }

    // options (all default to reference behaviour)
    bool drive_only = false;                 // skip the verifier's checks, keep challenges + prover calls
    // Fiat-Shamir needs every round challenge to depend on that round's message: draw r[j] after round j's polynomial has been
    // recorded instead of all of a phase's challenges up front as the reference does (src/verifier.cpp:155,207,279)
    bool lazy_challenges = false;
    const std::vector<G1> *fixed_gens = nullptr;   // re-use generators instead of drawing new ones
    // zero-knowledge mode (zk_mask.hpp; no reference counterpart): blinded commitments over (g, H), masked round polynomials, proofs of dot
    // product instead of the inner-product argument. The checks of the reference protocol are unchanged underneath the masks.
    bool zk = false;
    bool host_generators = false;            // draw the generators on the host even if the prover has a GPU (experiment switch ZKCNN_HOST_GENERATORS=1 through the session)
    bool full_ipa = false;                   // run the inner-product argument down to length 1 instead of stopping at IPA_STOP_LEN
    proofTranscript transcript;
    // test hook: add 1 to the k-th message received from the prover (a cheating prover); -1 = off.
    // Messages are numbered in arrival order: Vres, every round polynomial, every finalize call, then the
    // commitment-opening messages.
    long tamper_at = -1;
    long msg_count = 0;

    bool verify() {
        const u8 logn = C.circuit[0].bit_length;
        const size_t n_sqrt = (size_t) 1 << (logn - (logn >> 1));
        const size_t n_gens = n_sqrt + (zk ? 1 : 0);                 // zero-knowledge mode: one more generator, H
        if (!drive_only && (!accel || cross_check) && C.gatesDropped())
            return fail("the wiring predicates on the host need the gate lists: this session holds a structure copy of the circuit (a clone): use the GPU predicates");
        if (fixed_gens) gens = *fixed_gens;
        else drawGenerators(gens, n_gens, [this](const std::vector<G1Affine> &table, const std::vector<uint64_t> &k, std::vector<G1> &out) {
            return !host_generators && fixedBaseMul(*p, table, k, out); });
        if (gens.size() != n_gens) return false;
        delete poly_v;
        poly_v = new hyrax_bls12_381::polyVerifier(zk ? p->commitInputZk(gens) : p->commitInput(gens), gens, &transcript);
        poly_v->drive_only = drive_only;
        if (zk && !receiveMasks(n_sqrt)) return false;
        poly_v->stop_len = full_ipa ? 1 : (size_t) hyrax_bls12_381::IPA_STOP_LEN;
        poly_v->accel = accel;
        poly_v->cross_check = cross_check;
        msg_count = 0;
        if (!(verifyInnerLayers() && verifyFirstLayer())) return false;
        if (zk && !verifyMasks()) return false;
        poly_v->tamper_at = tamper_at < 0 ? -1 : tamper_at - msg_count;
        return verifyInput();
    }

    verifierAccel *accel = nullptr;     // wiring predicates on the GPU
    bool cross_check = false;           // with an accelerator: also run the host loops and require identical values
    timer total_timer, total_slow_timer;
    double verifierTime() const { return total_timer.elapse_sec(); }
    double verifierSlowTime() const { return total_slow_timer.elapse_sec(); }
    double polyVerifierTime() const { return poly_v ? poly_v->getVT() : 0.0; }
    const char *failure() const { return fail_msg.c_str(); }

private:
    vector<vector<F>> r_u, r_v;
    vector<F> final_claim_u0, final_claim_v0;
    vector<F> beta_g, beta_u, beta_v, beta_gs;
    F uni_value[2], bin_value[3];
    F eval_in;
    std::vector<G1> gens;
    hyrax_bls12_381::polyVerifier *poly_v;
    string fail_msg;

    // ---- zero-knowledge mode (zk_mask.hpp): masking polynomials of the sumcheck instances, masked evaluation claims ----
    struct maskRecord {
        size_t k;                                   // instance
        vector<F> r;                                // its point
        F E, v;                                     // prod (1 - r_i); the revealed rho g(r) + E K
        vector<std::pair<size_t, F>> in;            // incoming claims: (layer * 4 + slot, weight)
    };
    zkmask::plan mask_plan;
    std::vector<G1> mask_commit;
    std::vector<F> mask_sums, mask_z;               // mask_z[layer * 4 + slot] = Z of a masked claim (0: not masked)
    std::vector<maskRecord> mask_recs;
    zkmask::cursor mask_cur;
    F mask_rho, mask_z_in;
    size_t mask_k = 0;

    // the prover's commitment to every private scalar of the mode and the masking polynomials' sums over the cube, then rho
    bool receiveMasks(size_t m) {
        mask_plan = zkmask::plan(C, m);
        zkmask::maskCommitMsg msg = p->zkMaskCommit();
        if (msg.commit.size() != (mask_plan.total + m - 1) / m || msg.sums.size() != mask_plan.items.size()) return fail("masking commitment of the wrong shape");
        for (const G1 &c : msg.commit) transcript.put(c);
        for (const F &x : msg.sums) transcript.put(x);
        mask_commit = msg.commit;
        mask_sums = msg.sums;
        mask_recs.clear();
        mask_z.assign((size_t) C.size * 4, F_ZERO);
        mask_z_in = F_ZERO;
        mask_k = 0;
        tic();
        mask_rho.setByCSPRNG();
        toc();
        p->zkSetRho(mask_rho);
        return true;
    }
    // a sumcheck instance starts: its claim becomes H + rho G (H already carries K: it is a combination of masked claims)
    void maskOpen(F &claim, const vector<std::pair<size_t, F>> &in) {
        if (!zk) return;
        claim = claim + mask_rho * mask_sums.at(mask_k);
        mask_cur.reset();
        maskRecord rec;
        rec.k = mask_k;
        rec.in = in;
        mask_recs.push_back(rec);
    }
    vector<std::pair<size_t, F>> maskIncoming(int i, const F &alpha, const F &beta) const {
        vector<std::pair<size_t, F>> in;
        if (!zk) return in;
        for (const zkmask::incoming &c : zkmask::incomingOf(C, i))
            in.push_back(std::make_pair((size_t) c.layer * 4 + c.slot, c.w == 0 ? alpha : c.w == 1 ? beta : F_ONE));
        return in;
    }
    // the last round of a phase in the mode: a polynomial of the plan's degree, sent highest coefficient first
    bool maskLastRound(int kind, const F &previousRandom, F &at01, zkmask::zkPoly &poly) {
        poly = p->zkLastRound(kind, previousRandom);
        if (poly.deg != mask_plan.items.at(mask_k).deg.at(mask_cur.bound)) return fail("masked last round of the wrong degree");
        if (hit()) poly.c[0] = poly.c[0] + F_ONE;
        for (int e = poly.deg; e >= 0; --e) transcript.put(poly.c[e]);
        at01 = poly.eval(F_ZERO) + poly.eval(F_ONE);
        return true;
    }
    void maskBind(const F &r) { if (zk) mask_cur.bind(r); }
    // a phase ends: Z of its point for the claims that left masked
    void maskPhaseEnd(int layer, int slot0) {
        if (!zk) return;
        const F Z = mask_cur.zPhase(mask_cur.phase_r.size());
        if (layer == 0) mask_z_in = Z;
        else for (int b = 0; b < 2; ++b)
            if (zkmask::slotActive(C, layer, slot0 + b)) mask_z[(size_t) layer * 4 + slot0 + b] = Z;
        mask_cur.phase_r.clear();
    }
    // ... and the instance: the prover reveals v = rho g(r) + E K, the claim about f is what is left
    void maskClose(F &claim) {
        if (!zk) return;
        F v = p->zkMaskEval();
        if (hit()) v = v + F_ONE;
        transcript.put(v);
        claim = claim - v;
        maskRecord &rec = mask_recs.back();
        rec.v = v;
        rec.r = mask_cur.r;
        rec.E = mask_cur.E;
        ++mask_k;
    }
    // one proof of dot product for all revealed values: <a, sum_k gamma^k u_k> = sum_k gamma^k v_k against the commitment of a
    bool verifyMasks() {
        tic();
        F gamma, y = F_ZERO, w = F_ONE;
        gamma.setByCSPRNG();
        zkmask::evalVector ev(mask_plan);
        for (const maskRecord &rec : mask_recs) {
            ev.addG(rec.k, rec.r, w * mask_rho);
            for (const auto &c : rec.in) ev.addM((int) (c.first / 4), (int) (c.first % 4), w * rec.E * c.second * mask_z[c.first]);
            y = y + w * rec.v;
            w = w * gamma;
        }
        toc();
        if (mask_recs.size() != mask_plan.items.size()) return fail("masking: instance count");
        hyrax_bls12_381::dotProofCommit m1 = p->zkMaskOpen1(ev.u);
        if (hit()) m1.t = m1.t + F_ONE;
        for (const G1 &d : m1.delta) transcript.put(d);
        transcript.put(m1.t);
        tic();
        F c;
        c.setByCSPRNG();
        toc();
        hyrax_bls12_381::dotProofResponse m2 = p->zkMaskOpen2(c);
        if (hit() && !m2.z.empty()) m2.z[m2.z.size() / 3] = m2.z[m2.z.size() / 3] + F_ONE;
        for (const F &x : m2.z) transcript.put(x);
        for (const F &x : m2.z_blind) transcript.put(x);
        if (drive_only) return true;
        tic();
        std::vector<G1Affine> gA;
        zkff::batchToAffine(poly_v->generators(), gA);
        // (rows 1 .. of the commitment: row 0 masks the input's claim and is opened with the input, verifyInput)
        const bool ok = hyrax_bls12_381::dotVerify(std::vector<G1>(mask_commit.begin() + 1, mask_commit.end()), gA,
                                                   std::vector<F>(ev.u.begin() + mask_plan.row, ev.u.end()), y, m1, c, m2, accel, cross_check);
        toc();
        return ok ? true : fail("masking polynomials: proof of dot product");
    }

    void tic() { total_timer.start(); total_slow_timer.start(); }
    void toc() { total_timer.stop(); total_slow_timer.stop(); }
    bool fail(const string &msg) {
        toc();
        fail_msg = msg;
        fprintf(stderr, "Verification fail, %s\n", msg.c_str());
        return false;
    }
    bool hit() { return msg_count++ == tamper_at; }
    static void draw(vector<F> &v, size_t n) {
        v.resize(n);
        for (auto &x : v) x.setByCSPRNG();
    }

    // ---- predicate side: wiring-polynomial evaluations the final check of a layer needs ----
    // eq tables for phase 1 of layer `d` (reference src/verifier.cpp:36-82). r_0 / r_1 are the
    // points the layer's own output claim sits at, alpha / beta the combination weights.
    void betaInitPhase1(u8 d, const F &alpha, const F &beta, const vector<F>::const_iterator &r_0,
                        const vector<F>::const_iterator &r_1, const F &relu_rou) {
        const layer &L = C.circuit[d];
        const i8 bl = L.bit_length, fft_bl = L.fft_bit_length, fft_blh = fft_bl - 1;
        if (L.ty == layerType::FFT || L.ty == layerType::IFFT) {
            beta_gs.resize((size_t) 1 << fft_bl);
            phiGInit(beta_gs, r_0, L.scale, fft_bl, L.ty == layerType::IFFT);
            beta_u.resize((size_t) 1 << L.max_bl_u);
            initBetaTable(beta_u, L.max_bl_u, r_u[d].begin(), F_ONE);
        } else if (L.ty == layerType::PADDING) {
            // outputs are (vector, position): the vector part carries the two-point claim of the
            // DOT_PROD layer two levels up, the position part the FFT layer's sumcheck point
            beta_g.resize((size_t) 1 << bl);
            beta_gs.resize((size_t) 1 << fft_blh);
            initBetaTable(beta_g, bl - fft_blh, r_u[d + 2].begin() + fft_bl, r_v[d + 2].begin(), alpha, beta);
            initBetaTable(beta_gs, fft_blh, r_0, F_ONE);
            const size_t mask = ((size_t) 1 << fft_blh) - 1;
            for (size_t g = (size_t) 1 << bl; g-- > 0;) beta_g[g] = beta_g[g >> fft_blh] * beta_gs[g & mask];
            beta_u.resize((size_t) 1 << L.max_bl_u);
            initBetaTable(beta_u, L.max_bl_u, r_u[d].begin(), F_ONE);
        } else if (L.ty == layerType::DOT_PROD) {
            const i8 cnt_bl = bl - fft_bl, cnt_bl2 = L.max_bl_u - fft_bl;
            beta_g.resize((size_t) 1 << cnt_bl);
            initBetaTable(beta_g, cnt_bl, r_u[d + 2].begin() + fft_bl - 1, alpha);
            beta_u.resize((size_t) 1 << cnt_bl2);
            initBetaTable(beta_u, cnt_bl2, r_u[d].begin() + fft_bl, F_ONE);
            F same = F_ONE;     // eq(r_0, r_u[d]) on the frequency bits
            for (i8 j = 0; j < fft_bl; ++j)
                same = same * (r_0[j] * r_u[d][j] + (F_ONE - r_0[j]) * (F_ONE - r_u[d][j]));
            for (auto &x : beta_u) x = x * same;
        } else {
            beta_g.resize((size_t) 1 << bl);
            initBetaTable(beta_g, bl, r_0, r_1, alpha * L.scale, beta * L.scale);
            if (L.zero_start_id < L.size)
                for (size_t g = L.zero_start_id; g < ((size_t) 1 << bl); ++g) beta_g[g] = beta_g[g] * relu_rou;
            beta_u.resize((size_t) 1 << L.max_bl_u);
            initBetaTable(beta_u, L.max_bl_u, r_u[d].begin(), F_ONE);
        }
    }
    void betaInitPhase2(u8 d) {                                       // reference src/verifier.cpp:84-87
        beta_v.resize((size_t) 1 << C.circuit[d].max_bl_v);
        initBetaTable(beta_v, C.circuit[d].max_bl_v, r_v[d].begin(), F_ONE);
    }
    void predicatePhase1(u8 d) {                                      // reference src/verifier.cpp:89-102
        const layer &L = C.circuit[d];
        uni_value[0].clear();
        uni_value[1].clear();
        if (L.ty == layerType::FFT || L.ty == layerType::IFFT) {
            for (size_t u = 0; u < ((size_t) 1 << L.max_bl_u); ++u) uni_value[1] = uni_value[1] + beta_gs[u] * beta_u[u];
        } else for (const uniGate &gt : L.uni_gates) {
            int idx = gt.lu != 0;
            uni_value[idx] = uni_value[idx] + beta_g[gt.g] * beta_u[gt.u] * C.two_mul[gt.sc];
        }
        bin_value[0] = bin_value[1] = bin_value[2] = F_ZERO;
    }
    void predicatePhase2(u8 d) {                                      // reference src/verifier.cpp:104-116
        const layer &L = C.circuit[d];
        uni_value[0] = uni_value[0] * beta_v[0];
        uni_value[1] = uni_value[1] * beta_v[0];
        const bool dot = L.ty == layerType::DOT_PROD;
        for (const binGate &gt : L.bin_gates) {
            F t = beta_g[gt.g] * beta_u[gt.u] * beta_v[gt.v];
            if (!dot) t = t * C.two_mul[gt.sc];
            bin_value[gt.l] = bin_value[gt.l] + t;
        }
    }

    // ---- stage 1: layers size-1 .. 1 (reference src/verifier.cpp:132-266) ----
    bool verifyInnerLayers() {
        tic();
        F alpha = F_ONE, beta = F_ZERO, relu_rou = F_ONE, claim_u1 = F_ZERO, claim_v1 = F_ZERO;
        const layer &top = C.circuit[C.size - 1];
        draw(r_u[C.size], top.bit_length);
        vector<F>::const_iterator r_0 = r_u[C.size].begin(), r_1 = r_v[C.size].begin();
        toc();

        F previousSum = p->Vres(r_0, top.size, top.bit_length);
        if (hit()) previousSum = previousSum + F_ONE;
        transcript.put(previousSum);
        p->sumcheckInitAll(r_0);

        for (u8 i = C.size - 1; i; --i) {
            const layer &cur = C.circuit[i];
            const bool dot = cur.ty == layerType::DOT_PROD;
            p->sumcheckInit(alpha, beta);
            tic();
            draw(r_u[i], cur.max_bl_u);
            if (cur.zero_start_id < cur.size) relu_rou.setByCSPRNG();
            else relu_rou = F_ONE;
            toc();
            if (dot) p->sumcheckDotProdInitPhase1();
            else p->sumcheckInitPhase1(relu_rou);

            F previousRandom = F_ZERO;
            maskOpen(previousSum, maskIncoming(i, alpha, beta));
            for (i8 j = 0; j < cur.max_bl_u; ++j) {
                F at01, at_r;
                if (zk && j == cur.max_bl_u - 1) {
                    zkmask::zkPoly poly;
                    if (!maskLastRound(dot ? zkmask::PH_DOT1 : zkmask::PH_ONE, previousRandom, at01, poly)) return false;
                    tic();
                    if (lazy_challenges) r_u[i][j].setByCSPRNG();
                    at_r = poly.eval(r_u[i][j]);
                } else if (dot) {
                    cubic_poly poly = p->sumcheckDotProdUpdate1(previousRandom);
                    if (hit()) poly.d = poly.d + F_ONE;
                    transcript.put(poly);
                    tic();
                    if (lazy_challenges) r_u[i][j].setByCSPRNG();
                    at01 = poly.eval(F_ZERO) + poly.eval(F_ONE);
                    at_r = poly.eval(r_u[i][j]);
                } else {
                    quadratic_poly poly = p->sumcheckUpdate1(previousRandom);
                    if (hit()) poly.c = poly.c + F_ONE;
                    transcript.put(poly);
                    tic();
                    if (lazy_challenges) r_u[i][j].setByCSPRNG();
                    at01 = poly.eval(F_ZERO) + poly.eval(F_ONE);
                    at_r = poly.eval(r_u[i][j]);
                }
                if (!drive_only && at01 != previousSum)
                    return fail("phase1, circuit " + std::to_string(i) + ", current bit " + std::to_string(j));
                previousRandom = r_u[i][j];
                previousSum = at_r;
                maskBind(previousRandom);
                toc();
            }
            if (dot) {
                p->sumcheckDotProdFinalize1(previousRandom, claim_u1);
                if (hit()) claim_u1 = claim_u1 + F_ONE;
                transcript.put(claim_u1);
            } else {
                p->sumcheckFinalize1(previousRandom, final_claim_u0[i], claim_u1);
                if (hit()) claim_u1 = claim_u1 + F_ONE;
                transcript.put(final_claim_u0[i]);
                transcript.put(claim_u1);
            }
            maskPhaseEnd(i, zkmask::SLOT_U0);

            total_slow_timer.start();
            const bool host_pred = !drive_only && (!accel || cross_check);
            if (host_pred) {
                betaInitPhase1(i, alpha, beta, r_0, r_1, relu_rou);
                predicatePhase1(i);
            }
            total_timer.start();
            if (cur.need_phase2) {
                draw(r_v[i], cur.max_bl_v);
                toc();
                p->sumcheckInitPhase2();
                previousRandom = F_ZERO;
                for (i8 j = 0; j < cur.max_bl_v; ++j) {
                    F at01, at_r;
                    if (zk && j == cur.max_bl_v - 1) {
                        zkmask::zkPoly poly;
                        if (!maskLastRound(zkmask::PH_TWO, previousRandom, at01, poly)) return false;
                        tic();
                        if (lazy_challenges) r_v[i][j].setByCSPRNG();
                        at_r = poly.eval(r_v[i][j]);
                    } else {
                        quadratic_poly poly = p->sumcheckUpdate2(previousRandom);
                        if (hit()) poly.a = poly.a + F_ONE;
                        transcript.put(poly);
                        tic();
                        if (lazy_challenges) r_v[i][j].setByCSPRNG();
                        at01 = poly.eval(F_ZERO) + poly.eval(F_ONE);
                        at_r = poly.eval(r_v[i][j]);
                    }
                    if (!drive_only && at01 != previousSum)
                        return fail("phase2, circuit level " + std::to_string(i) + ", current bit " + std::to_string(j));
                    previousRandom = r_v[i][j];
                    previousSum = at_r;
                    maskBind(previousRandom);
                    toc();
                }
                p->sumcheckFinalize2(previousRandom, final_claim_v0[i], claim_v1);
                if (hit()) claim_v1 = claim_v1 + F_ONE;
                transcript.put(final_claim_v0[i]);
                transcript.put(claim_v1);
                maskPhaseEnd(i, zkmask::SLOT_V0);
                total_slow_timer.start();
                if (host_pred) {
                    betaInitPhase2(i);
                    predicatePhase2(i);
                }
                total_timer.start();
            }
            maskClose(previousSum);
            if (!drive_only && accel) {
                total_slow_timer.start();
                F uni[2], bin[3];
                // r_u / r_v hold C.size + 2 entries: PADDING / DOT_PROD look two layers up
                accel->predicates(i, r_u[i + 1].data(), r_v[i + 1].data(), alpha, beta, relu_rou, r_u[i].data(), r_v[i].data(),
                                  r_u[i + 2].data(), r_v[i + 2].data(), uni, bin);
                if (cross_check)
                    for (int k = 0; k < 3; ++k)
                        if ((k < 2 && uni[k] != uni_value[k]) || bin[k] != bin_value[k])
                            return fail("predicate mismatch between host and accelerator, circuit level " + std::to_string(i));
                uni_value[0] = uni[0]; uni_value[1] = uni[1];
                bin_value[0] = bin[0]; bin_value[1] = bin[1]; bin_value[2] = bin[2];
                total_timer.start();
            }
            if (!drive_only) {
                // sum over gate types of (wiring predicate) x (claimed operand values)
                F expect = bin_value[0] * (final_claim_u0[i] * final_claim_v0[i]) + bin_value[1] * (claim_u1 * claim_v1) +
                           bin_value[2] * (claim_u1 * final_claim_v0[i]) + uni_value[0] * final_claim_u0[i] +
                           uni_value[1] * claim_u1;
                if (previousSum != expect) return fail("semi final, circuit level " + std::to_string(i));
            }
            if (cur.ty == layerType::FFT || cur.ty == layerType::IFFT) {
                previousSum = claim_u1;                 // carry the claim, alpha / beta stay as they are
            } else {
                if (~cur.bit_length_u[1]) alpha.setByCSPRNG();
                else alpha.clear();
                if (~cur.bit_length_v[1]) beta.setByCSPRNG();
                else beta.clear();
                previousSum = alpha * claim_u1 + beta * claim_v1;
            }
            r_0 = r_u[i].begin();
            r_1 = r_v[i].begin();
            toc();
            beta_u.clear();
            beta_v.clear();
        }
        return true;
    }

    // ---- stage 2: all claims about layer 0 in one sumcheck (reference src/verifier.cpp:268-357) ----
    bool verifyFirstLayer() {
        tic();
        const layer &cur = C.circuit[0];
        vector<F> sig_u, sig_v;
        draw(sig_u, C.size - 1);
        draw(sig_v, C.size - 1);
        draw(r_u[0], cur.bit_length);
        F previousSum = F_ZERO;
        for (int i = 1; i < C.size; ++i) {
            if (~C.circuit[i].bit_length_u[0]) previousSum = previousSum + sig_u[i - 1] * final_claim_u0[i];
            if (~C.circuit[i].bit_length_v[0]) previousSum = previousSum + sig_v[i - 1] * final_claim_v0[i];
        }
        toc();

        p->sumcheckLiuInit(sig_u, sig_v);
        F previousRandom = F_ZERO;
        vector<std::pair<size_t, F>> in;
        for (int i = 1; zk && i < C.size; ++i) {
            if (zkmask::slotActive(C, i, zkmask::SLOT_U0)) in.push_back(std::make_pair((size_t) i * 4 + zkmask::SLOT_U0, sig_u[i - 1]));
            if (zkmask::slotActive(C, i, zkmask::SLOT_V0)) in.push_back(std::make_pair((size_t) i * 4 + zkmask::SLOT_V0, sig_v[i - 1]));
        }
        maskOpen(previousSum, in);
        for (int j = 0; j < cur.bit_length; ++j) {
            F at01, at_r;
            if (zk && j == cur.bit_length - 1) {
                zkmask::zkPoly poly;
                if (!maskLastRound(zkmask::PH_LIU, previousRandom, at01, poly)) return false;
                if (lazy_challenges) r_u[0][j].setByCSPRNG();
                at_r = poly.eval(r_u[0][j]);
            } else {
                quadratic_poly poly = p->sumcheckLiuUpdate(previousRandom);
                if (hit()) poly.b = poly.b + F_ONE;
                transcript.put(poly);
                if (lazy_challenges) r_u[0][j].setByCSPRNG();
                at01 = poly.eval(F_ZERO) + poly.eval(F_ONE);
                at_r = poly.eval(r_u[0][j]);
            }
            if (!drive_only && at01 != previousSum)
                return fail("Liu, circuit 0, current bit " + std::to_string(j));
            previousRandom = r_u[0][j];
            previousSum = at_r;
            maskBind(previousRandom);
        }
        p->sumcheckLiuFinalize(previousRandom, eval_in);
        if (hit()) eval_in = eval_in + F_ONE;
        transcript.put(eval_in);
        maskPhaseEnd(0, 0);
        maskClose(previousSum);

        if (!drive_only) {
            total_slow_timer.start();
            F gr = F_ZERO;
            const bool host_pred = !accel || cross_check;
            if (host_pred) {
            beta_g.resize((size_t) 1 << cur.bit_length);
            initBetaTable(beta_g, cur.bit_length, r_u[0].begin(), F_ONE);
            }
            for (int i = 1; host_pred && i < C.size; ++i) {
                const layer &L = C.circuit[i];
                if (~L.bit_length_u[0]) {
                    beta_u.resize((size_t) 1 << L.bit_length_u[0]);
                    initBetaTable(beta_u, L.bit_length_u[0], r_u[i].begin(), sig_u[i - 1]);
                    for (u32 j = 0; j < L.size_u[0]; ++j) gr = gr + beta_g[L.ori_id_u[j]] * beta_u[j];
                }
                if (~L.bit_length_v[0]) {
                    beta_v.resize((size_t) 1 << L.bit_length_v[0]);
                    initBetaTable(beta_v, L.bit_length_v[0], r_v[i].begin(), sig_v[i - 1]);
                    for (u32 j = 0; j < L.size_v[0]; ++j) gr = gr + beta_g[L.ori_id_v[j]] * beta_v[j];
                }
            }
            if (accel) {
                F g2 = accel->inputPredicate(r_u[0], r_u, r_v, sig_u, sig_v);
                if (cross_check && g2 != gr) return fail("layer-0 predicate mismatch between host and accelerator");
                gr = g2;
            }
            total_timer.start();
            if (eval_in * gr != previousSum) return fail("Liu, semi final, circuit 0");
            toc();
        }
        output_tb[PT_OUT_ID] = to_string_wp(p->proveTime());
        output_tb[VT_OUT_ID] = to_string_wp(verifierTime());
        output_tb[PS_OUT_ID] = to_string_wp(p->proofSize());
        beta_g.clear(); beta_gs.clear(); beta_u.clear(); beta_v.clear();
        return true;
    }

    // ---- stage 3: open the committed input at r_u[0] (reference src/verifier.cpp:359-373) ----
    bool verifyInput() {
        // (zero-knowledge mode: eval_in is the MASKED value V_0(r) + Z <a_row0, eq(r_low)>: the opening is against P + Z D_0, D_0 = the mask commitment's first row)
        if (!(zk ? poly_v->verifyZk(r_u[0], eval_in, &mask_commit.at(0), mask_z_in) : poly_v->verify(r_u[0], eval_in))) return fail("final input check fail");
        output_tb[POLY_PT_OUT_ID] = to_string_wp(p->polyProverTime());
        output_tb[POLY_VT_OUT_ID] = to_string_wp(poly_v->getVT());
        output_tb[POLY_PS_OUT_ID] = to_string_wp(p->polyProofSize());
        output_tb[TOT_PT_OUT_ID] = to_string_wp(p->polyProverTime() + p->proveTime());
        output_tb[TOT_VT_OUT_ID] = to_string_wp(poly_v->getVT() + verifierTime());
        output_tb[TOT_PS_OUT_ID] = to_string_wp(p->polyProofSize() + p->proofSize());
        return true;
    }
};

// Zero-knowledge masking of the sumcheck transcript (the reference states that it "is not fully zero-knowledge",
// reference README.md:5 -- this is the follow-on its commitment scheme was chosen for).
//
// Two masks, both Chiesa-Forbes-Spooner / Libra (Xie et al., CRYPTO'19, section 4):
//
// (1) ROUND MESSAGES. Every sumcheck INSTANCE of a proof -- one per layer (phase 1, then phase 2 if the layer has one), and the layer-0 "Liu"
//     sumcheck -- gets a random polynomial with the per-variable degrees of its round polynomials,
//         g(x) = a_0 + sum_i ( a_{i,1} x_i + ... + a_{i,d_i} x_i^{d_i} ),
//     committed (with every other private scalar of the mode) in one blinded Pedersen matrix commitment before any challenge; the sums
//     G = sum_{x in {0,1}^l} g(x) are sent. The verifier answers with rho, and the sumcheck runs on f + rho g + K eq_0 (K: see (2)) with claim
//     H + rho G: the j-th message is  p_j(t) + rho q_j(t) + E_j K (1 - t),
//         q_j(t) = 2^(l-j-1) (a_0 + sum_{i<j} g_i(r_i) + g_j(t)) + 2^(l-j-2) sum_{i>j} g_i(1),      E_j = prod_{i<j} (1 - r_i),
//     uniformly distributed given the transcript so far. After the instance's last round the prover reveals the ONE value
//     v = rho g(r) + E_l K; the verifier continues with f(r) = claim - v.
//
// (2) EVALUATION CLAIMS (round 5; what the first version of the mode left open). Every claim V(u) a phase ends in -- the operands' values at the
//     phase's point: claim_u0 / claim_u1 / claim_v0 / claim_v1 of each layer and the input's value behind the Liu sumcheck -- is sent as
//         c~ = V(u) + Z(u) M,        Z(u) = prod_j u_j (1 - u_j)  over the phase's l challenges,
//     with one committed uniformly random scalar M per claim: the low-degree extension V is replaced by V' = V + Z M, which agrees with V on the
//     cube. Z vanishes as long as ONE variable is still boolean, so the device tables, every round kernel and every round polynomial but the
//     LAST of a phase are untouched; the last one gains Z's degree,
//         p_last(t) += Z' t (1 - t) sum_b M_b A_b(t),        Z' = prod_{j<l-1} r_j (1 - r_j),   A_b(t) = the operand's multiplier table at (r', t)
//     (O(1) on the host from the phase's last table pairs: `zkTailPairs`), and phase 2 / the layer's final check are about the masked values
//     (what `zkMaskClaims` added goes back to the backend: zk_sumcheck_claims_adjust). The layer BELOW starts from a combination of masked claims, alpha c~_u + beta c~_v = (the true sum) + K with
//     K = alpha Z_u M_u + beta Z_v M_v: a committed quantity the moment the challenges are known, carried through that layer's sumcheck as the
//     term K eq_0(x) above and removed at its end inside v -- never sent on its own. The INPUT's claim is masked by a committed ROW:
//     M_in = <a_row0, eq(r_low)> (m fresh scalars, the first row of the mask commitment), so that the masked value is again a matrix opening --
//     of the input commitment plus Z times that row -- and Hyrax's proof of dot product runs unchanged on P + Z D_0.
//     Simulator: (G, round messages) are uniform through g for any fixed witness and M; every c~ is uniform through its own M (Z != 0 but with
//     negligible probability); v and f(r) are determined by those -- tests/test_zk_cpu.py restates the identities with Python integers.
//     Not covered: a phase without rounds (a one-entry operand table has no variable to hide behind), Vres (the public output).
//
// At the end ONE proof of dot product shows every revealed v against the commitment: <a, sum_k gamma^k u_k> = sum_k gamma^k v_k, where u_k holds
// rho x (the monomials of g_k at r^(k)) and E K's coefficients at the M entries.
//
// Everything here is O(rounds) scalar work on the host, shared by the HIP-backed prover, the CPU oracle and the verifier; the group
// operations (commitments) go through hyrax_bls12_381::polyProverBase.
#pragma once
#include <stdexcept>
#include "circuit.h"
#include "polynomial.h"

namespace zkmask {

typedef zkff::PrivateCoins coins;      // the prover's private randomness (ff/fr.hpp)

enum { SLOT_U0 = 0, SLOT_U1 = 1, SLOT_V0 = 2, SLOT_V1 = 3 };
enum phaseKind { PH_DOT1 = 0, PH_ONE = 1, PH_TWO = 2, PH_LIU = 3 };

// a round polynomial of the mode's last rounds: c[k] is the coefficient of t^k, `deg` of them + 1 are sent (highest first, like a/b/c/d)
struct zkPoly {
    F c[5];
    int deg = 0;
    zkPoly() { for (F &x : c) x = F_ZERO; }
    F eval(const F &t) const {
        F acc = c[deg];
        for (int k = deg - 1; k >= 0; --k) acc = acc * t + c[k];
        return acc;
    }
};

inline Fr pow2(int e) {
    Fr x = Fr::one(), two = Fr(2LL);
    for (int i = 0; i < e; ++i) x = x * two;
    return x;
}
inline Fr zfactor(const Fr &r) { return r * (Fr::one() - r); }

// which sumcheck instances a proof runs, in protocol order (verifier.hpp: verifyInnerLayers, then verifyFirstLayer), and where every private
// scalar of the mode lives in the committed vector: [ row 0: the input claim's mask row, m scalars | g of every instance | M of every claim ]
struct plan {
    struct item {
        int layer = 0, ell1 = 0, ell = 0;        // variables of phase 1, of the whole instance
        std::vector<u8> deg;                     // per-variable degree
        std::vector<u32> voff;                   // offset of variable i's coefficients from a_0
        size_t off = 0;                          // offset of a_0 in the vector
    };
    std::vector<item> items;
    std::vector<size_t> slot_off;                // (layer, slot) -> offset of M, layers 1 .. size-1
    size_t row = 0, total = 0;
    int size = 0;

    plan() {}
    plan(const layeredCircuit &C, size_t m) : row(m), size(C.size) {
        total = m;
        auto add = [&](int layer, int l1, int base1, int l2) {
            item it;
            it.layer = layer; it.ell1 = l1; it.ell = l1 + l2; it.off = total;
            u32 o = 1;
            for (int i = 0; i < it.ell; ++i) {
                const bool last = i == l1 - 1 || i == it.ell - 1;
                const int d = (i < l1 ? base1 : 2) + (last ? 1 : 0);       // the last round of a phase carries Z's t (1 - t)
                it.deg.push_back((u8) d);
                it.voff.push_back(o);
                o += d;
            }
            total += o;
            items.push_back(it);
        };
        for (int i = C.size - 1; i >= 1; --i) {
            const layer &L = C.circuit[i];
            add(i, std::max<int>(L.max_bl_u, 0), L.ty == layerType::DOT_PROD ? 3 : 2, L.need_phase2 ? std::max<int>(L.max_bl_v, 0) : 0);
        }
        add(0, C.circuit[0].bit_length, 2, 0);
        slot_off.assign((size_t) C.size * 4, 0);
        for (int i = 1; i < C.size; ++i)
            for (int s = 0; s < 4; ++s) slot_off[(size_t) i * 4 + s] = total++;
    }
    size_t slot(int layer, int s) const { return slot_off.at((size_t) layer * 4 + s); }
};

// does the claim of (layer, slot) exist and have a variable to hide behind?
inline bool slotActive(const layeredCircuit &C, int i, int s) {
    const layer &L = C.circuit[i];
    const bool v = s >= 2;
    const int b = s & 1;
    if (v && !L.need_phase2) return false;
    if ((v ? L.max_bl_v : L.max_bl_u) < 1) return false;
    if (L.ty == layerType::DOT_PROD && s == SLOT_U0) return false;
    return (v ? L.bit_length_v[b] : L.bit_length_u[b]) >= 0;
}
// the claims layer i's own sumcheck starts from, as (layer, slot, weight kind): 0 = alpha, 1 = beta, 2 = one
struct incoming { int layer, slot, w; };
inline std::vector<incoming> incomingOf(const layeredCircuit &C, int i) {
    std::vector<incoming> in;
    if (i < 1 || i >= C.size - 1) return in;                 // the top layer starts from Vres
    const layerType up = C.circuit[i + 1].ty;
    if (up == layerType::FFT || up == layerType::IFFT) {
        if (slotActive(C, i + 1, SLOT_U1)) in.push_back({i + 1, SLOT_U1, 2});
    } else {
        if (slotActive(C, i + 1, SLOT_U1)) in.push_back({i + 1, SLOT_U1, 0});
        if (slotActive(C, i + 1, SLOT_V1)) in.push_back({i + 1, SLOT_V1, 1});
    }
    return in;
}

// the running algebra of one instance, as both sides need it: challenges bound so far, E = prod (1 - r_i)
struct cursor {
    int round = 0;                     // variables whose message has been produced
    int bound = 0;                     // variables whose challenge is known
    Fr E = Fr::one();
    std::vector<Fr> r;                 // challenges of the instance
    std::vector<Fr> phase_r;           // ... of the phase in progress
    void reset() { round = bound = 0; E = Fr::one(); r.clear(); phase_r.clear(); }
    void bind(const Fr &x) { r.push_back(x); phase_r.push_back(x); E = E * (Fr::one() - x); ++bound; }
    Fr zPhase(size_t upto) const {     // prod_{j < upto} r_j (1 - r_j) over the phase's challenges
        Fr z = Fr::one();
        for (size_t j = 0; j < upto && j < phase_r.size(); ++j) z = z * zfactor(phase_r[j]);
        return z;
    }
};

// prover side: the coefficients and the running state of the instance being proved
class proverState {
public:
    plan pl;
    std::vector<Fr> a;                 // every private scalar of the mode (plan's layout)
    Fr rho;
    bool active = false;
    cursor cur;
    Fr K;                              // the incoming claims' mask term of the instance in progress
    std::vector<Fr> zm;                // (layer, slot) -> Z M of the masked claim (0: not masked)

    void draw(const layeredCircuit &C, size_t m, coins &rnd) {
        pl = plan(C, m);
        a.resize(pl.total);
        for (Fr &x : a) x = rnd.next();
        zm.assign((size_t) C.size * 4, Fr(0LL));
        k = -1;
        active = true;
    }
    const plan::item &item() const { return pl.items.at(k); }
    Fr gi(int i, const Fr &t) const {                 // g_i(t), Horner without the constant term
        const plan::item &it = item();
        const Fr *c = a.data() + it.off + it.voff[i];
        Fr acc(0LL);
        for (int e = it.deg[i]; e >= 1; --e) acc = (acc + c[e - 1]) * t;
        return acc;
    }
    Fr gi1(int i) const {
        const plan::item &it = item();
        const Fr *c = a.data() + it.off + it.voff[i];
        Fr acc(0LL);
        for (int e = 1; e <= it.deg[i]; ++e) acc = acc + c[e - 1];
        return acc;
    }
    std::vector<Fr> cubeSums() {
        std::vector<Fr> G(pl.items.size());
        const int keep = k;
        for (size_t i = 0; i < pl.items.size(); ++i) {
            k = (int) i;
            const plan::item &it = item();
            Fr s(0LL);
            for (int v = 0; v < it.ell; ++v) s = s + gi1(v);
            G[i] = it.ell ? pow2(it.ell) * a[it.off] + pow2(it.ell - 1) * s : a[it.off];
        }
        k = keep;
        return G;
    }
    // a sumcheck instance starts (phase 1 of a layer, the Liu sumcheck), in plan order
    void begin(const Fr &K_in) {
        ++k;
        cur.reset();
        K = K_in;
        prefix = Fr(0LL);
        suffix = Fr(0LL);
        for (int i = 1; i < item().ell; ++i) suffix = suffix + gi1(i);
    }
    void bind(const Fr &r) {
        prefix = prefix + gi(cur.bound, r);
        cur.bind(r);
    }
    // adds rho q_j + E_j K (1 - t) to the round polynomial (coefficients c[0 .. n), n > the variable's degree)
    void maskRound(Fr *c, int n) {
        const plan::item &it = item();
        const int j = cur.round, rest = it.ell - j - 1;
        if (j >= it.ell || cur.bound != j || n <= it.deg[j]) throw std::runtime_error("zero-knowledge masks: round out of order");
        const Fr S = pow2(rest);
        Fr c0 = S * (a[it.off] + prefix);
        if (rest > 0) c0 = c0 + pow2(rest - 1) * suffix;
        c[0] = c[0] + rho * c0;
        const Fr *aj = a.data() + it.off + it.voff[j];
        for (int e = 1; e <= it.deg[j]; ++e) c[e] = c[e] + rho * S * aj[e - 1];
        const Fr ek = cur.E * K;
        c[0] = c[0] + ek;
        c[1] = c[1] - ek;
        ++cur.round;
        if (cur.round < it.ell) suffix = suffix - gi1(cur.round);
    }
    // v = rho g(r) + E K once every challenge of the instance is bound
    Fr eval() const {
        if (cur.bound != item().ell) throw std::runtime_error("zero-knowledge masks: instance closed before its last challenge");
        return rho * (a[item().off] + prefix) + cur.E * K;
    }
    int instance() const { return k; }

private:
    int k = -1;
    Fr prefix, suffix;
};

// verifier side: the vector u with <a, u> = sum_k w_k v_k
class evalVector {
public:
    explicit evalVector(const plan &p) : pl(p), u(p.total, Fr(0LL)) {}
    // rho g_k(r): entry of a_0 gets w, entry of a_{i,e} gets w r_i^e   (w = weight x rho)
    void addG(size_t k, const std::vector<Fr> &r, const Fr &w) {
        const plan::item &it = pl.items[k];
        u[it.off] = u[it.off] + w;
        for (int i = 0; i < it.ell; ++i) {
            Fr pw = w;
            for (int e = 1; e <= it.deg[i]; ++e) {
                pw = pw * r[i];
                u[it.off + it.voff[i] + (e - 1)] = u[it.off + it.voff[i] + (e - 1)] + pw;
            }
        }
    }
    void addM(int layer, int slot, const Fr &w) { u[pl.slot(layer, slot)] = u[pl.slot(layer, slot)] + w; }
    const plan &pl;
    std::vector<Fr> u;
};

// the correction of a phase's last round: c[1 ..] += Z' (t - t^2) sum_b M_b A_b(t), A_b(t) = A[3 b] + A[3 b + 1] t + A[3 b + 2] t^2
inline void addLastRoundMask(Fr *c, int n, const Fr &Zp, const Fr Mb[2], const Fr A[6]) {
    Fr s[3];
    for (int e = 0; e < 3; ++e) s[e] = Zp * (Mb[0] * A[e] + Mb[1] * A[3 + e]);
    for (int e = 0; e < 3; ++e) {
        if (s[e].isZero()) continue;
        if (e + 2 >= n) throw std::runtime_error("zero-knowledge masks: last-round polynomial of unexpected degree");
        c[e + 1] = c[e + 1] + s[e];
        c[e + 2] = c[e + 2] - s[e];
    }
}

// ---- what a prover class adds for the mode (the HIP-backed `prover`, the CPU oracle): state + the protocol steps. The host class calls
// zkSetWeights() from sumcheckInit, zkBeginInstance() from its phase-1 / Liu initialisers, zkBeginPhase() from every phase initialiser, zkMask()
// from its update methods and zkMaskClaims() from its finalize methods; the verifier asks for a phase's LAST round through zkLastRound(). The
// group operations go to the commitment backend (GPU MSM kernels / CPU Pippenger). All methods are additive to the reference's prover interface. ----
struct maskCommitMsg { std::vector<G1> commit; std::vector<Fr> sums; };

class proverMixin {
public:
    virtual ~proverMixin() {}
    bool zkActive() const { return zk_st.active; }
    void zkReset() { zk_st.active = false; }
    // step 1, after the input commitment: every private scalar drawn, committed row-wise (rows of m scalars, one blind each), cube sums
    maskCommitMsg zkMaskCommit() {
        hyrax_bls12_381::polyProverBase &be = zkBackend();
        coins &rnd = zkff::privateCoins();
        const size_t m = be.zkColumns();
        zk_st.draw(zkCircuit(), m, rnd);
        const size_t rows = (zk_st.pl.total + m - 1) / m;
        zk_blinds.resize(rows);
        for (Fr &b : zk_blinds) b = rnd.next();
        maskCommitMsg msg;
        msg.commit = be.commitHostVector(zk_st.a, zk_blinds);
        msg.sums = zk_st.cubeSums();
        be.addProofBytes(rows * 48 + 32 * msg.sums.size());
        zkModeOn();
        return msg;
    }
    void zkSetRho(const Fr &rho) { zk_st.rho = rho; }
    // the last round of the phase in progress: the backend's unmasked polynomial + Z's term + the masks
    zkPoly zkLastRound(int kind, const Fr &prev_r) {
        if (!zk_st.active) throw std::runtime_error("zkLastRound outside the zero-knowledge mode");
        zkPoly p;
        zk_suppress = true;
        try { zkRawRound(kind, prev_r, p.c); } catch (...) { zk_suppress = false; throw; }
        zk_suppress = false;
        if (zk_round_in_phase > 0) zk_st.bind(prev_r);
        const plan::item &it = zk_st.item();
        p.deg = it.deg.at(zk_st.cur.round);
        Fr A[6], Mb[2];
        zkTailPairs(A);
        const int layer = it.layer;
        for (int b = 0; b < 2; ++b) {
            Mb[b] = Fr(0LL);
            if (kind == PH_LIU) { if (b == 1) Mb[b] = zk_min = inputMask(); }
            else if (slotActive(zkCircuit(), layer, (kind == PH_TWO ? 2 : 0) + b)) Mb[b] = zk_st.a[zk_st.pl.slot(layer, (kind == PH_TWO ? 2 : 0) + b)];
        }
        addLastRoundMask(p.c, p.deg + 1, zk_st.cur.zPhase(zk_st.cur.phase_r.size()), Mb, A);
        zk_st.maskRound(p.c, p.deg + 1);
        ++zk_round_in_phase;
        zkBackend().addProofBytes(32 * (size_t) (p.deg + 1));
        return p;
    }
    // v of the instance that just ended
    Fr zkMaskEval() {
        zkBackend().addProofBytes(32);
        return zk_st.eval();
    }
    // the proof of dot product for the revealed values runs over rows 1 .. of the commitment: row 0 (the input claim's mask row) is opened with the input
    // and has no part in any v -- leaving it out saves its m response scalars (131 KB of a vgg11 proof) and one MSM on either side
    hyrax_bls12_381::dotProofCommit zkMaskOpen1(const std::vector<Fr> &u) {
        return zkBackend().dotCommit(std::vector<Fr>(u.begin() + zk_st.pl.row, u.end()), zk_blinds.size() - 1, zkff::privateCoins());
    }
    hyrax_bls12_381::dotProofResponse zkMaskOpen2(const Fr &c) {
        return zkBackend().dotRespond(std::vector<Fr>(zk_st.a.begin() + zk_st.pl.row, zk_st.a.end()), std::vector<Fr>(zk_blinds.begin() + 1, zk_blinds.end()), c);
    }

protected:
    virtual hyrax_bls12_381::polyProverBase &zkBackend() = 0;
    virtual const layeredCircuit &zkCircuit() const = 0;
    // the backend's unmasked round of the given kind, as coefficients of t^0 .. t^3
    virtual void zkRawRound(int kind, const Fr &prev_r, Fr c[5]) = 0;
    // after a phase's last round: A_b(t) of both operand pairs as (t^0, t^1, t^2) coefficients -- the multiplier table's last pair (m0, m1 - m0, 0);
    // a pair absorbed earlier: its multiplier times (1 - t); a DOT_PROD phase 1 before its periodic table collapsed: the product of two pairs
    virtual void zkTailPairs(Fr A[6]) = 0;
    // the backend switches to what the mode needs (the HIP prover: every phase's last rounds on the host, where the last pairs are)
    virtual void zkModeOn() {}

    void zkSetWeights(const Fr &alpha, const Fr &beta) { zk_alpha = alpha; zk_beta = beta; }
    // K of the instance that starts: the incoming claims' Z M under their weights
    void zkBeginInstance() {
        if (!zk_st.active) return;
        const int layer = zk_st.pl.items.at(zk_st.instance() + 1).layer;
        Fr K(0LL);
        for (const incoming &in : incomingOf(zkCircuit(), layer)) {
            const Fr &w = in.w == 0 ? zk_alpha : in.w == 1 ? zk_beta : zk_one;
            K = K + w * zk_st.zm[(size_t) in.layer * 4 + in.slot];
        }
        zk_st.begin(K);
        zk_round_in_phase = 0;
    }
    void zkBeginLiu(const std::vector<Fr> &s_u, const std::vector<Fr> &s_v) {
        if (!zk_st.active) return;
        Fr K(0LL);
        const layeredCircuit &C = zkCircuit();
        for (int i = 1; i < C.size; ++i) {
            if (slotActive(C, i, SLOT_U0)) K = K + s_u[i - 1] * zk_st.zm[(size_t) i * 4 + SLOT_U0];
            if (slotActive(C, i, SLOT_V0)) K = K + s_v[i - 1] * zk_st.zm[(size_t) i * 4 + SLOT_V0];
        }
        zk_st.begin(K);
        zk_round_in_phase = 0;
    }
    void zkBeginPhase() { zk_round_in_phase = 0; zk_st.cur.phase_r.clear(); }
    // a round that is not the last of its phase (the backend's update methods call this on their polynomial)
    void zkMask(quadratic_poly &p, const Fr &prev_r) {
        if (!zk_st.active || zk_suppress) return;
        Fr c[3] = {p.c, p.b, p.a};
        step(c, 3, prev_r);
        p.c = c[0]; p.b = c[1]; p.a = c[2];
    }
    void zkMask(cubic_poly &p, const Fr &prev_r) {
        if (!zk_st.active || zk_suppress) return;
        Fr c[4] = {p.d, p.c, p.b, p.a};
        step(c, 4, prev_r);
        p.d = c[0]; p.c = c[1]; p.b = c[2]; p.a = c[3];
    }
    // a phase ends: the last challenge is bound, the operands' values leave masked. `slot0` = SLOT_U0 / SLOT_V0; d[b] = what was added
    void zkMaskClaims(int slot0, const Fr &last_r, Fr &c0, Fr &c1, Fr d[2]) {
        d[0] = d[1] = Fr(0LL);
        if (!zk_st.active) return;
        if (zk_round_in_phase > 0) zk_st.bind(last_r);
        const int layer = zk_st.item().layer;
        const Fr Z = zk_st.cur.zPhase(zk_st.cur.phase_r.size());
        Fr *c[2] = {&c0, &c1};
        for (int b = 0; b < 2; ++b) {
            if (!slotActive(zkCircuit(), layer, slot0 + b)) continue;
            d[b] = Z * zk_st.a[zk_st.pl.slot(layer, slot0 + b)];
            *c[b] = *c[b] + d[b];
            zk_st.zm[(size_t) layer * 4 + slot0 + b] = d[b];
        }
        zkBeginPhase();
    }
    // the input's claim: masked by row 0 at the point's column half; the opening is told to add Z x that row
    void zkMaskInputClaim(const Fr &last_r, Fr &claim) {
        if (!zk_st.active) return;
        if (zk_round_in_phase > 0) zk_st.bind(last_r);
        const Fr Z = zk_st.cur.zPhase(zk_st.cur.phase_r.size());
        claim = claim + Z * zk_min;
        std::vector<Fr> row(zk_st.a.begin(), zk_st.a.begin() + zk_st.pl.row);
        for (Fr &x : row) x = x * Z;
        zkBackend().setOpenExtra(row, Z * zk_blinds.at(0));
        zkBeginPhase();
    }
    // blinding factors of the input commitment rows (drawn BEFORE the masking coefficients: both provers draw in this order)
    static std::vector<Fr> zkDrawBlinds(size_t rows) {
        std::vector<Fr> b(rows);
        coins &rnd = zkff::privateCoins();
        for (Fr &x : b) x = rnd.next();
        return b;
    }

private:
    void step(Fr *c, int n, const Fr &prev_r) {
        if (zk_round_in_phase > 0) zk_st.bind(prev_r);
        zk_st.maskRound(c, n);
        ++zk_round_in_phase;
    }
    // <a_row0, eq(r_low)>: the column half of the Liu point is bound before the last round (the input has >= 1 row bit)
    Fr inputMask() const {
        const int n = zk_st.item().ell, rb = n >> 1, cb = n - rb;
        if (rb < 1 || (int) zk_st.cur.phase_r.size() < cb || ((size_t) 1 << cb) != zk_st.pl.row)
            throw std::runtime_error("zero-knowledge mode: the input layer is too small to mask its claim");
        std::vector<Fr> eq;
        hyrax_bls12_381::eqTable(eq, zk_st.cur.phase_r.data(), cb, Fr::one());
        Fr s(0LL);
        for (size_t j = 0; j < eq.size(); ++j) s = s + eq[j] * zk_st.a[j];
        return s;
    }
    proverState zk_st;
    std::vector<Fr> zk_blinds;
    Fr zk_alpha, zk_beta, zk_one = Fr::one(), zk_min;
    int zk_round_in_phase = 0;
    bool zk_suppress = false;
};

}  // namespace zkmask

// Zero-knowledge masking of the sumcheck round polynomials (the reference states that it "is not fully zero-knowledge",
// reference README.md:5 -- this is the follow-on its commitment scheme was chosen for).
//
// Every sumcheck instance of a proof (phase 1 and phase 2 of each layer, and the layer-0 "Liu" sumcheck) is masked the standard way
// (Chiesa-Forbes-Spooner / Libra): the prover picks a random polynomial with the same per-variable degree d as the round polynomials,
//     g(x) = a_0 + sum_i ( a_{i,1} x_i + ... + a_{i,d} x_i^d ),
// commits to ALL coefficients of all instances in one blinded Pedersen matrix commitment before any challenge, and sends the sums
// G = sum_{x in {0,1}^l} g(x). The verifier answers with rho, and each sumcheck is run on f + rho g with claim H + rho G: the j-th round
// message is  p_j(t) + rho q_j(t),  q_j(t) = 2^(l-j-1) (a_0 + sum_{i<j} g_i(r_i) + g_j(t)) + 2^(l-j-2) sum_{i>j} g_i(1),
// which is uniformly distributed given the transcript so far. After the last round the prover reveals v = g(r); the verifier continues
// with f(r) = claim - rho v. At the end ONE proof of dot product shows that every revealed v is the committed g evaluated at its point.
// What this does NOT hide: the per-layer evaluation claims V(u), V(v) themselves (Libra masks them with extra low-degree terms) -- so the
// mode makes the ROUND MESSAGES and the COMMITMENTS leak nothing, not yet the whole transcript.
//
// Everything here is O(rounds) scalar work on the host, shared by the HIP-backed prover, the CPU oracle and the verifier; the group
// operations (commitments) go through hyrax_bls12_381::polyProverBase.
#pragma once
#include "circuit.h"
#include "polynomial.h"

namespace zkmask {

typedef zkff::PrivateCoins coins;      // the prover's private randomness (ff/fr.hpp)

// which sumcheck instances a proof runs, in protocol order (verifier.hpp: verifyInnerLayers, then verifyFirstLayer)
struct plan {
    struct item { int ell, deg; size_t off; };       // variables, per-variable degree, offset of a_0 in the coefficient vector
    std::vector<item> items;
    size_t total = 0;
    explicit plan(const layeredCircuit &C) {
        auto add = [&](int ell, int deg) {
            item it = {ell, deg, total};
            items.push_back(it);
            total += 1 + (size_t) ell * deg;
        };
        for (int i = C.size - 1; i >= 1; --i) {
            const layer &L = C.circuit[i];
            add(L.max_bl_u, L.ty == layerType::DOT_PROD ? 3 : 2);
            if (L.need_phase2) add(L.max_bl_v, 2);
        }
        add(C.circuit[0].bit_length, 2);
    }
    plan() {}
};

inline Fr pow2(int e) {
    Fr x = Fr::one(), two = Fr(2LL);
    for (int i = 0; i < e; ++i) x = x * two;
    return x;
}

// coefficient view of one instance
struct view {
    const Fr *a;          // a[0] = a_0, a[1 + i * deg + (e - 1)] = a_{i,e}
    int ell, deg;
    Fr gi(int i, const Fr &t) const {                 // g_i(t), Horner without the constant term
        Fr acc(0LL);
        for (int e = deg; e >= 1; --e) acc = (acc + a[1 + i * deg + (e - 1)]) * t;
        return acc;
    }
    Fr gi1(int i) const {                             // g_i(1)
        Fr acc(0LL);
        for (int e = 1; e <= deg; ++e) acc = acc + a[1 + i * deg + (e - 1)];
        return acc;
    }
    Fr sumOverCube() const {                          // G = 2^l a_0 + 2^(l-1) sum_i g_i(1)
        if (ell == 0) return a[0];
        Fr s(0LL);
        for (int i = 0; i < ell; ++i) s = s + gi1(i);
        return pow2(ell) * a[0] + pow2(ell - 1) * s;
    }
};

// prover side: the coefficients and the running state of the instance being proved
class proverState {
public:
    plan pl;
    std::vector<Fr> a;                 // all coefficients, instance after instance
    Fr rho;
    bool active = false;

    void draw(const layeredCircuit &C, coins &rnd) {
        pl = plan(C);
        a.resize(pl.total);
        for (Fr &x : a) x = rnd.next();
        k = -1;
        active = true;
    }
    std::vector<Fr> cubeSums() const {
        std::vector<Fr> G(pl.items.size());
        for (size_t i = 0; i < pl.items.size(); ++i) G[i] = at(i).sumOverCube();
        return G;
    }
    // a sumcheck instance starts (called from the prover's phase initialisers, in plan order)
    void begin() {
        ++k;
        round = 0;
        prefix = Fr(0LL);
        const view v = at(k);
        suffix = Fr(0LL);
        for (int i = 1; i < v.ell; ++i) suffix = suffix + v.gi1(i);
    }
    // adds rho q_j to the round polynomial the unmasked prover returned; `prev_r` is the challenge of the previous round (ignored in round 0)
    template <class Poly>
    void maskRound(Poly &p, const Fr &prev_r) {
        const view v = at(k);
        if (round > 0) prefix = prefix + v.gi(round - 1, prev_r);
        const int j = round, rest = v.ell - j - 1;
        const Fr S = pow2(rest);
        Fr c0 = S * (v.a[0] + prefix);
        if (rest > 0) c0 = c0 + pow2(rest - 1) * suffix;
        addMasked(p, c0, S, v.a + 1 + j * v.deg, v.deg);
        ++round;
        if (round < v.ell) suffix = suffix - v.gi1(round);
    }
    // g(r) once the last challenge is known
    Fr eval(const Fr &last_r) {
        const view v = at(k);
        if (v.ell > 0) prefix = prefix + v.gi(v.ell - 1, last_r);
        return v.a[0] + prefix;
    }
    int instance() const { return k; }

private:
    int k = -1, round = 0;
    Fr prefix, suffix;
    view at(size_t i) const { view v = {a.data() + pl.items[i].off, pl.items[i].ell, pl.items[i].deg}; return v; }
    void addMasked(quadratic_poly &p, const Fr &c0, const Fr &S, const Fr *aj, int deg) const {
        (void) deg;
        p.c = p.c + rho * c0;
        p.b = p.b + rho * S * aj[0];
        p.a = p.a + rho * S * aj[1];
    }
    void addMasked(cubic_poly &p, const Fr &c0, const Fr &S, const Fr *aj, int deg) const {
        (void) deg;
        p.d = p.d + rho * c0;
        p.c = p.c + rho * S * aj[0];
        p.b = p.b + rho * S * aj[1];
        p.a = p.a + rho * S * aj[2];
    }
};

// verifier side: the vector u with <a, u> = sum_k gamma^k g_k(r^(k)) -- entry of a_0 gets gamma^k, entry of a_{i,e} gets gamma^k r_i^e
class evalVector {
public:
    explicit evalVector(const plan &p) : pl(p), u(p.total, Fr(0LL)) {}
    void add(size_t k, const std::vector<Fr> &r, const Fr &weight) {
        const plan::item &it = pl.items[k];
        u[it.off] = u[it.off] + weight;
        for (int i = 0; i < it.ell; ++i) {
            Fr pw = weight;
            for (int e = 1; e <= it.deg; ++e) {
                pw = pw * r[i];
                u[it.off + 1 + (size_t) i * it.deg + (e - 1)] = u[it.off + 1 + (size_t) i * it.deg + (e - 1)] + pw;
            }
        }
    }
    const plan &pl;
    std::vector<Fr> u;
};

// ---- what a prover class adds for the mode (the HIP-backed `prover`, the CPU oracle): state + the protocol steps. The host class
// calls zkBeginInstance() from its phase initialisers and zkMask() from its update methods; the group operations go to its commitment
// backend (GPU MSM kernels / CPU Pippenger). All methods are additive to the reference's prover interface. ----
struct maskCommitMsg { std::vector<G1> commit; std::vector<Fr> sums; };

class proverMixin {
public:
    virtual ~proverMixin() {}
    bool zkActive() const { return zk_st.active; }
    void zkReset() { zk_st.active = false; }
    // step 1, after the input commitment: coefficients drawn, committed row-wise (rows of m scalars, one blind each), cube sums
    maskCommitMsg zkMaskCommit() {
        hyrax_bls12_381::polyProverBase &be = zkBackend();
        coins &rnd = zkff::privateCoins();
        zk_st.draw(zkCircuit(), rnd);
        const size_t m = be.zkColumns(), rows = (zk_st.pl.total + m - 1) / m;
        zk_blinds.resize(rows);
        for (Fr &b : zk_blinds) b = rnd.next();
        maskCommitMsg msg;
        msg.commit = be.commitHostVector(zk_st.a, zk_blinds);
        msg.sums = zk_st.cubeSums();
        be.addProofBytes(rows * 48 + 32 * msg.sums.size());
        return msg;
    }
    void zkSetRho(const Fr &rho) { zk_st.rho = rho; }
    Fr zkMaskEval(const Fr &last_r) {
        zkBackend().addProofBytes(32);
        return zk_st.eval(last_r);
    }
    hyrax_bls12_381::dotProofCommit zkMaskOpen1(const std::vector<Fr> &u) { return zkBackend().dotCommit(u, zk_blinds.size(), zkff::privateCoins()); }
    hyrax_bls12_381::dotProofResponse zkMaskOpen2(const Fr &c) { return zkBackend().dotRespond(zk_st.a, zk_blinds, c); }

protected:
    virtual hyrax_bls12_381::polyProverBase &zkBackend() = 0;
    virtual const layeredCircuit &zkCircuit() const = 0;
    void zkBeginInstance() { if (zk_st.active) zk_st.begin(); }
    template <class Poly> void zkMask(Poly &p, const Fr &prev_r) { if (zk_st.active) zk_st.maskRound(p, prev_r); }
    // blinding factors of the input commitment rows (drawn BEFORE the masking coefficients: both provers draw in this order)
    static std::vector<Fr> zkDrawBlinds(size_t rows) {
        std::vector<Fr> b(rows);
        coins &rnd = zkff::privateCoins();
        for (Fr &x : b) x = rnd.next();
        return b;
    }
private:
    proverState zk_st;
    std::vector<Fr> zk_blinds;
};

}  // namespace zkmask

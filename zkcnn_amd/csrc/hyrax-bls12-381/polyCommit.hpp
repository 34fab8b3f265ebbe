// <hyrax-bls12-381/polyCommit.hpp> -- the header the reference includes for its field, curve,
// timer, integer typedefs and polynomial commitment (reference src/global_var.hpp:6,
// src/circuit.h:6, src/prover.hpp:40-46,74, src/verifier.hpp:43). The upstream submodule
// (TAMUCrypto/hyrax-bls12-381 + herumi/mcl) is EMPTY in /root/reference, so this is a from-scratch
// implementation of the surface the call sites need, not a copy of it.
//
// Commitment scheme (Hyrax, Wahby et al. S&P'18, matrix form, no blinding -- the reference README
// states the released version is not fully zero-knowledge):
//   Z has 2^n entries, viewed as 2^rb rows x 2^cb columns, rb = n/2, cb = n - rb, entry (i, j) at
//   index i * 2^cb + j (column bits are the LOW bits, matching the little-endian variable order
//   of the sumcheck). Commitment = one Pedersen vector commitment per row over `gens`.
//   Opening at x: w = L^T Z with L = eq(x[cb..n)), then a log-round inner-product argument for
//   <w, R> = y, R = eq(x[0..cb)), against P = sum_i L[i] * C_i:
//     round: prover sends  Lk = <a_lo, g_hi>, Rk = <a_hi, g_lo>, yL = <a_lo, b_hi>, yR = <a_hi, b_lo>
//            verifier sends c;   a' = a_lo + c a_hi,  g' = c g_lo + g_hi,  b' = c b_lo + b_hi
//            P' = Lk + c P + c^2 Rk,   y' = yL + c y + c^2 yR
//     the recursion stops at length IPA_STOP_LEN (256): the prover sends the remaining vector a* and the verifier
//     checks  P* == <a*, g*>  and  y* == <a*, b*>  (g*, b* = generators / eq table folded with the challenges).
//     Stopping early trades 8 KB of proof for 8 latency-bound rounds (each round is two MSMs over half of the generators).
// Zero-knowledge mode (ZKCNN_MODE_ZK): `gens` carries one more generator H, every commitment is blinded,
//   Com(v; s) = <v, g> + s H, and the opening is Hyrax's proof of dot product instead of the inner-product argument:
//     prover: d random, delta = Com(d; s_d), t = <d, b>      verifier: c      prover: z = c w + d, z_s = c s_w + s_d
//     check:  Com(z; z_s) == c P + delta   and   <z, b> == c y + t             (perfect honest-verifier zero knowledge; m scalars)
//   The same protocol, over several rows, proves the evaluations of the sumcheck masking polynomials (host/zk_mask.hpp).
// Message order/format of the upstream library is unknowable here ("parity unpinned").
#pragma once
#include <chrono>
#include <memory>
#include <stdexcept>
#include <vector>
#include "../ff/fr.hpp"
#include "../ff/g1.hpp"

typedef unsigned char u8;
typedef unsigned short u16;
typedef unsigned int u32;
typedef unsigned long long u64;
typedef char i8;               // the reference relies on i8 == (signed) char: utils.hpp:13 vs utils.cpp:23
typedef short i16;
typedef int i32;
typedef long long i64;

using zkff::Fr;
using zkff::G1;
using zkff::G1Affine;

namespace mcl {
enum CurveId { BLS12_381 = 5 };
namespace bn {
inline const G1 &getG1basePoint() { return G1::generator(); }
}
}
inline void initPairing(mcl::CurveId) {}

// accumulate-on-start/stop wall clock (reference use: src/prover.hpp:42-43, src/verifier.hpp:20-22)
class timer {
public:
    timer() : total_(0), running_(false) {}
    void start() {
        if (!running_) { t0_ = std::chrono::steady_clock::now(); running_ = true; }
    }
    void stop() {
        if (running_) {
            total_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
            running_ = false;
        }
    }
    void clear() { total_ = 0; running_ = false; }
    double elapse_sec() const {
        if (!running_) return total_;
        return total_ + std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
    }
private:
    double total_;
    bool running_;
    std::chrono::steady_clock::time_point t0_;
};

namespace hyrax_bls12_381 {

// eq(r, .) over `n` little-endian variables scaled by `init`
inline void eqTable(std::vector<Fr> &out, const Fr *r, int n, const Fr &init) {
    out.assign((size_t) 1 << n, Fr(0LL));
    out[0] = init;
    for (int i = 0; i < n; ++i) {
        size_t half = (size_t) 1 << i;
        for (size_t j = 0; j < half; ++j) {
            Fr t = out[j] * r[i];
            out[j | half] = t;
            out[j] = out[j] - t;
        }
    }
}

enum { IPA_STOP_LEN = 256 };
inline int ipaRounds(int cb, size_t stop_len = IPA_STOP_LEN) { int t = 0; while (((size_t) 1 << (cb - t)) > stop_len) ++t; return t; }

struct ipaRoundMsg {
    G1 L, R;
    Fr yL, yR;
};

// proof of dot product over a blinded matrix commitment (rows of m = gens.size() - 1 scalars)
struct dotProofCommit { std::vector<G1> delta; Fr t; };
struct dotProofResponse { std::vector<Fr> z, z_blind; };

// What the verifier needs from a commitment prover. Two implementations exist: the HIP one
// (class polyProver, zkcnn_amd/csrc/host/polyProver.{hpp,cpp}) and the CPU checker under oracle/.
class polyProverBase {
public:
    virtual ~polyProverBase() {}
    virtual const std::vector<G1> &commitment() const = 0;
    virtual void openInit(const std::vector<Fr> &x) = 0;
    virtual ipaRoundMsg openRound() = 0;
    virtual void openFold(const Fr &c) = 0;
    virtual std::vector<Fr> openFinal() = 0;     // the vector left after ipaRounds(cb) rounds
    virtual double getPT() const = 0;      // seconds
    virtual double getPS() const = 0;      // KB

    // ---- zero-knowledge mode: two primitives per backend, the protocol on top of them is generic ----
    // blinded commitments of a host vector laid out in rows of m scalars (zero padded): out[i] = <v_i, g> + blinds[i] H
    virtual std::vector<G1> commitHostVector(const std::vector<Fr> &, const std::vector<Fr> &) { throw std::runtime_error("commitment backend without zero-knowledge support"); }
    // w = L^T Z of the committed input, L = eq(x[cb..n)): the vector the opening is about (m scalars)
    virtual std::vector<Fr> combineRows(const std::vector<Fr> &) { throw std::runtime_error("commitment backend without zero-knowledge support"); }
    virtual void addProofBytes(size_t) {}
    size_t zkColumns() const { return zk_m; }

    // proof of dot product for ANY vector `a` committed row-wise with blinds `r` (C_i = Com(a_i; r_i)): shows <a, u> = y
    dotProofCommit dotCommit(const std::vector<Fr> &u, size_t rows, zkff::PrivateCoins &rnd) {
        dot_d.resize(rows * zk_m);
        for (Fr &x : dot_d) x = rnd.next();
        dot_s.resize(rows);
        for (Fr &x : dot_s) x = rnd.next();
        dotProofCommit m1;
        m1.delta = commitHostVector(dot_d, dot_s);
        m1.t = Fr(0LL);
        for (size_t j = 0; j < u.size() && j < dot_d.size(); ++j) m1.t = m1.t + dot_d[j] * u[j];
        addProofBytes(rows * 48 + 32);
        return m1;
    }
    dotProofResponse dotRespond(const std::vector<Fr> &a, const std::vector<Fr> &r, const Fr &c) {
        dotProofResponse m2;
        m2.z = dot_d;
        for (size_t j = 0; j < a.size() && j < m2.z.size(); ++j) m2.z[j] = m2.z[j] + c * a[j];
        m2.z_blind = dot_s;
        for (size_t i = 0; i < r.size() && i < m2.z_blind.size(); ++i) m2.z_blind[i] = m2.z_blind[i] + c * r[i];
        addProofBytes(32 * (m2.z.size() + m2.z_blind.size()));
        dot_d.clear(); dot_s.clear();
        return m2;
    }
    // the two prover messages of the INPUT opening in zero-knowledge mode
    virtual dotProofCommit zkOpenCommit(const std::vector<Fr> &x, const std::vector<Fr> &b) {
        zkff::PrivateCoins &rnd = zkff::privateCoins();
        zk_w = combineRows(x);
        const int n = (int) x.size(), rb = n >> 1, cb = n - rb;
        std::vector<Fr> Lrow;
        eqTable(Lrow, x.data() + cb, rb, Fr::one());
        zk_w_blind = Fr(0LL);
        for (size_t i = 0; i < Lrow.size() && i < input_blinds.size(); ++i) zk_w_blind = zk_w_blind + Lrow[i] * input_blinds[i];
        // the masked claim (host/zk_mask.hpp (2)): the opening is about w + Z a_row0, committed as P + Z D_0
        for (size_t j = 0; j < open_extra_w.size() && j < zk_w.size(); ++j) zk_w[j] = zk_w[j] + open_extra_w[j];
        zk_w_blind = zk_w_blind + open_extra_blind;
        open_extra_w.clear();
        open_extra_blind = Fr(0LL);
        return dotCommit(b, 1, rnd);
    }
    virtual dotProofResponse zkOpenRespond(const Fr &c) { return dotRespond(zk_w, std::vector<Fr>(1, zk_w_blind), c); }
    // a committed row (already scaled) that the next zero-knowledge opening adds to the combined input row, with its blind
    void setOpenExtra(const std::vector<Fr> &w, const Fr &blind) { open_extra_w = w; open_extra_blind = blind; }

protected:
    size_t zk_m = 0;                       // columns of a commitment row (set by the backend when it is built for zero knowledge)
    std::vector<Fr> input_blinds;          // blinding factor of every input row
private:
    std::vector<Fr> dot_d, dot_s, zk_w, open_extra_w;
    Fr zk_w_blind, open_extra_blind = Fr(0LL);
};

// Optional accelerator for the verifier's two multi-scalar multiplications (reference src/verifier.cpp:360 ends in them: 0.14 s of a
// 0.21 s vgg11 verification on one host core). Returns false when it does not take the call; the host Pippenger runs then.
struct msmAccel {
    virtual ~msmAccel() {}
    // out = sum k[i] * bases[i]; bases_are_generators: `bases` is (a prefix of) the generator set of this proof
    virtual bool msm(G1 &out, const Fr *k, const G1Affine *bases, size_t n, bool bases_are_generators) = 0;
};
inline G1 msmAny(msmAccel *accel, bool cross, const Fr *k, const G1Affine *bases, size_t n, bool gens) {
    G1 out;
    if (accel && accel->msm(out, k, bases, n, gens)) {
        if (cross && out != zkff::msmCPU(k, bases, n)) throw std::runtime_error("verifier MSM: accelerator and host disagree");
        return out;
    }
    return zkff::msmCPU(k, bases, n);
}

// verifier's side of the proof of dot product: rows C_i, generators in affine form (m + 1 of them, H last)
inline bool dotVerify(const std::vector<G1> &C, const std::vector<G1Affine> &gA, const std::vector<Fr> &u, const Fr &y, const dotProofCommit &m1,
                      const Fr &c, const dotProofResponse &m2, msmAccel *accel = nullptr, bool cross = false) {
    const size_t rows = C.size(), m = gA.size() - 1;
    if (m1.delta.size() != rows || m2.z.size() != rows * m || m2.z_blind.size() != rows || u.size() > rows * m) return false;
    std::vector<Fr> sc(m + 1);
    for (size_t i = 0; i < rows; ++i) {
        for (size_t j = 0; j < m; ++j) sc[j] = m2.z[i * m + j];
        sc[m] = m2.z_blind[i];
        const G1 lhs = msmAny(accel, cross, sc.data(), gA.data(), m + 1, true);
        if (lhs != C[i] * c + m1.delta[i]) return false;
    }
    Fr zu(0LL);
    for (size_t j = 0; j < u.size(); ++j) zu = zu + m2.z[j] * u[j];
    return zu == c * y + m1.t;
}

// hook so a transcript recorder can observe every commitment-phase message in order
struct transcriptSink {
    virtual ~transcriptSink() {}
    virtual void put(const Fr &) = 0;
    virtual void put(const G1 &) = 0;
};

class polyVerifier {
public:
    polyVerifier(polyProverBase &prover, const std::vector<G1> &gens, transcriptSink *sink = nullptr)
            : p(prover), g(gens), comm(prover.commitment()), sink_(sink) {
        if (sink_) for (auto &c : comm) sink_->put(c);
    }

    bool verify(const std::vector<Fr> &x, const Fr &eval) {
        vt.start();
        const int n = (int) x.size();
        const int rb = n >> 1, cb = n - rb;
        bool ok = ((size_t) 1 << cb) == g.size() && ((size_t) 1 << rb) == comm.size();
        if (!ok) { vt.stop(); return false; }

        std::vector<Fr> Lrow, b;
        eqTable(Lrow, x.data() + cb, rb, Fr::one());
        eqTable(b, x.data(), cb, Fr::one());
        G1 P;
        if (!drive_only) {
            std::vector<G1Affine> commA;
            zkff::batchToAffine(comm, commA);
            P = msmAny(accel, cross_check, Lrow.data(), commA.data(), commA.size(), false);
        }
        Fr y = eval;
        vt.stop();

        p.openInit(x);
        const int rounds = ipaRounds(cb, stop_len);
        const int kb = cb - rounds;                      // bits of the vector that is sent in the clear
        std::vector<Fr> cs(rounds);
        for (int t = 0; t < rounds; ++t) {
            ipaRoundMsg m = p.openRound();
            if (tamper_at == (long) t) m.yR = m.yR + Fr::one();
            if (sink_) { sink_->put(m.L); sink_->put(m.R); sink_->put(m.yL); sink_->put(m.yR); }
            vt.start();
            cs[t].setByCSPRNG();
            const Fr &c = cs[t];
            if (!drive_only) {
                Fr c2 = c * c;
                P = m.L + P * c + m.R * c2;
                y = m.yL + c * y + c2 * m.yR;
            }
            vt.stop();
            p.openFold(c);
        }
        std::vector<Fr> a = p.openFinal();
        if (a.size() != ((size_t) 1 << kb)) return false;
        if (tamper_at == (long) rounds) a[a.size() / 2] = a[a.size() / 2] + Fr::one();
        if (sink_) for (const Fr &v : a) sink_->put(v);

        if (drive_only) return true;
        vt.start();
        // coef(j) = prod_t (bit_{cb-1-t}(j) ? 1 : c_t): round t splits on bit cb-1-t; the low kb bits stay free
        std::vector<Fr> coef((size_t) 1 << cb);
        const size_t keep = (size_t) 1 << kb;
        for (size_t j = 0; j < keep; ++j) coef[j] = Fr::one();
        for (int t = rounds - 1; t >= 0; --t) {
            size_t half = (size_t) 1 << (cb - 1 - t);
            for (size_t j = 0; j < half; ++j) {
                coef[j | half] = coef[j];
                coef[j] = coef[j] * cs[t];
            }
        }
        // <a*, b*> and <a*, g*> over the original tables: entry j carries a*[j mod 2^kb] * coef(j)
        Fr ystar(0LL);
        std::vector<Fr> sc(coef.size());
        for (size_t j = 0; j < coef.size(); ++j) {
            sc[j] = coef[j] * a[j & (keep - 1)];
            ystar = ystar + sc[j] * b[j];
        }
        std::vector<G1Affine> gA;
        zkff::batchToAffine(g, gA);
        G1 Pstar = msmAny(accel, cross_check, sc.data(), gA.data(), gA.size(), true);
        ok = (P == Pstar) && (y == ystar);
        vt.stop();
        return ok;
    }
    // zero-knowledge opening: gens = (g_0 .. g_{m-1}, H); proof of dot product <w, R> = eval against P = sum_i L_i C_i
    // (extra, extra_k): the opening is against P + extra_k * extra -- the committed row that masks the claimed value (host/zk_mask.hpp (2))
    bool verifyZk(const std::vector<Fr> &x, const Fr &eval, const G1 *extra = nullptr, const Fr &extra_k = Fr(0LL)) {
        vt.start();
        const int n = (int) x.size();
        const int rb = n >> 1, cb = n - rb;
        if (((size_t) 1 << cb) + 1 != g.size() || ((size_t) 1 << rb) != comm.size()) { vt.stop(); return false; }
        std::vector<Fr> Lrow, b;
        eqTable(Lrow, x.data() + cb, rb, Fr::one());
        eqTable(b, x.data(), cb, Fr::one());
        G1 P;
        if (!drive_only) {
            std::vector<G1Affine> commA;
            zkff::batchToAffine(comm, commA);
            P = msmAny(accel, cross_check, Lrow.data(), commA.data(), commA.size(), false);
            if (extra) P = P + *extra * extra_k;
        }
        vt.stop();
        dotProofCommit m1 = p.zkOpenCommit(x, b);
        if (tamper_at == 0) m1.t = m1.t + Fr::one();
        if (sink_) { for (const G1 &d : m1.delta) sink_->put(d); sink_->put(m1.t); }
        vt.start();
        Fr c;
        c.setByCSPRNG();
        vt.stop();
        dotProofResponse m2 = p.zkOpenRespond(c);
        if (tamper_at == 1) m2.z[m2.z.size() / 2] = m2.z[m2.z.size() / 2] + Fr::one();
        if (tamper_at == 2) m2.z_blind[0] = m2.z_blind[0] + Fr::one();
        if (sink_) { for (const Fr &v : m2.z) sink_->put(v); for (const Fr &v : m2.z_blind) sink_->put(v); }
        if (drive_only) return true;
        vt.start();
        std::vector<G1Affine> gA;
        zkff::batchToAffine(g, gA);
        const bool ok = dotVerify(std::vector<G1>(1, P), gA, b, eval, m1, c, m2, accel, cross_check);
        vt.stop();
        return ok;
    }
    const std::vector<G1> &generators() const { return g; }
    double getVT() const { return vt.elapse_sec(); }
    msmAccel *accel = nullptr; // the two MSMs of the check on a GPU (optional)
    bool cross_check = false;  // with an accelerator: also run the host MSM and require the same point
    long tamper_at = -1;       // test hook: corrupt the k-th opening message (round k, or ipaRounds = the final vector)
    bool drive_only = false;   // make the prover calls and draw the challenges, skip the checks (bench mode)
    size_t stop_len = IPA_STOP_LEN;   // 1 = the textbook argument: log2(m) rounds, a single scalar at the end (ZKCNN_MODE_FULL_IPA)

private:
    polyProverBase &p;
    const std::vector<G1> &g;
    std::vector<G1> comm;
    transcriptSink *sink_;
    timer vt;
};

} // namespace hyrax_bls12_381

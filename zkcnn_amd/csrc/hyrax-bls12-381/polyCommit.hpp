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
// Message order/format of the upstream library is unknowable here ("parity unpinned").
#pragma once
#include <chrono>
#include <memory>
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
};

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
            P = zkff::msmCPU(Lrow.data(), commA.data(), commA.size());
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
        G1 Pstar = zkff::msmCPU(sc.data(), gA.data(), gA.size());
        ok = (P == Pstar) && (y == ystar);
        vt.stop();
        return ok;
    }
    double getVT() const { return vt.elapse_sec(); }
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

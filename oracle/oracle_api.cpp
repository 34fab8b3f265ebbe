// ORACLE -- TEST INFRASTRUCTURE ONLY (PARITY UNPINNED, see ref_tables.hpp).
// C entry points of liboracle.so: the whole-proof driver of include/zkcnn_api.h backed by the CPU
// restatement, plus array-level reference functions the kernel parity tests compare against.
#include "../zkcnn_amd/csrc/ff/sha256.hpp"
#include "../zkcnn_amd/csrc/ff/blake2s.hpp"
#include "ref_prover.hpp"
#include "session.hpp"

typedef sessionT<oracle::prover> oracleSession;

// field elements cross the ABI as 4 x u64 Montgomery limbs
static inline Fr &FR(uint64_t *p, size_t i) { return *reinterpret_cast<Fr *>(p + 4 * i); }
static inline const Fr &FR(const uint64_t *p, size_t i) { return *reinterpret_cast<const Fr *>(p + 4 * i); }

extern "C" {

void *oracle_session_create(const zkcnn_model_desc *desc, int32_t) {
    oracleSession *s = new oracleSession();
    if (!s->build(desc)) { delete s; return nullptr; }
    return s;
}
int32_t oracle_session_prove(void *session, uint64_t seed, uint32_t mode, uint8_t *transcript, uint64_t cap,
                             zkcnn_result *out) {
    return ((oracleSession *) session)->prove(seed, mode, transcript, cap, out);
}
int64_t oracle_session_statement(void *session, int32_t *scales, uint64_t cap) {
    const vector<int> &v = ((oracleSession *) session)->statementScales();
    for (size_t i = 0; i < v.size() && i < cap; ++i) scales[i] = v[i];
    return (int64_t) v.size();
}
void *oracle_verifier_create(const zkcnn_model_desc *desc, const int32_t *scales, uint64_t n_scales, int32_t) {
    oracleSession *s = new oracleSession();
    try {
        if (!s->buildStatement(desc, scales, n_scales)) { delete s; return nullptr; }
    } catch (const std::exception &) { delete s; return nullptr; }
    return s;
}
int32_t oracle_session_verify(void *session, uint64_t seed, uint32_t mode, const uint8_t *proof, uint64_t len, zkcnn_result *out) {
    return ((oracleSession *) session)->verifyProof(proof, len, seed, mode, out);
}
void oracle_session_destroy(void *session) { delete (oracleSession *) session; }
int32_t oracle_session_row(void *session, char *buf, uint64_t cap) {
    const string &r = ((oracleSession *) session)->row;
    if (!cap) return -1;
    std::snprintf(buf, cap, "%s", r.c_str());
    return 0;
}

// BLAKE2s-256 of `data` fed in two pieces (split point `split`), and the word-level chain step the GPU kernel uses
void oracle_blake2s(uint8_t out[32], const uint8_t *data, uint64_t n, uint64_t split) {
    zkff::Blake2s h;
    h.init();
    if (split > n) split = n;
    h.update(data, split);
    h.update(data + split, n - split);
    h.final(out);
}
void oracle_blake2s_chain(uint32_t state[8], const uint32_t *msg, int32_t n_words) { zkff::blake2s_chain_words(state, msg, n_words); }

// public commitment generators (ff/hash_to_curve.hpp): affine points in the C-ABI layout + the digest a transcript absorbs
void oracle_public_generators(uint64_t *out, uint8_t digest[32], uint64_t n) {
    const zkff::publicGenerators &pg = zkff::publicGeneratorSet(n);
    for (size_t i = 0; i < n; ++i) {
        G1Affine a = pg.gens[i].toAffine();
        std::memcpy(out + 12 * i, &a, 96);
    }
    std::memcpy(digest, pg.digest, 32);
}

// ---- field (4 x u64 Montgomery limbs per element, the C-ABI form) ----

void oracle_fr_from_canonical(uint64_t *out, const uint64_t *in, uint64_t n) {
    for (size_t i = 0; i < n; ++i) FR(out, i) = Fr(zkff::MontField<zkff::FrParams>::fromCanonical(in + 4 * i));
}
void oracle_fr_to_canonical(uint64_t *out, const uint64_t *in, uint64_t n) {
    for (size_t i = 0; i < n; ++i) FR(in, i).toCanonical(out + 4 * i);
}
void oracle_fr_mul(uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) {
    for (size_t i = 0; i < n; ++i) FR(out, i) = FR(a, i) * FR(b, i);
}
void oracle_fr_add(uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) {
    for (size_t i = 0; i < n; ++i) FR(out, i) = FR(a, i) + FR(b, i);
}
void oracle_fr_sub(uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t n) {
    for (size_t i = 0; i < n; ++i) FR(out, i) = FR(a, i) - FR(b, i);
}
void oracle_fr_inv(uint64_t *out, const uint64_t *a, uint64_t n) {
    for (size_t i = 0; i < n; ++i) Fr::inv(FR(out, i), FR(a, i));
}
int32_t oracle_fr_sqrt(uint64_t *out, const uint64_t *a) { return Fr::squareRoot(FR(out, 0), FR(a, 0)) ? 1 : 0; }
void oracle_fr_random(uint64_t *out, uint64_t n, uint64_t seed) {
    Fr::seedCSPRNG(seed);
    for (size_t i = 0; i < n; ++i) FR(out, i).setByCSPRNG();
}
void oracle_root_of_unity(uint64_t *out, int32_t n) { FR(out, 0) = oracle::rootOfUnity(n); }

// ---- tables ----
void oracle_eq_table2(uint64_t *out, int32_t n, const uint64_t *r0, const uint64_t *r1, const uint64_t *alpha,
                      const uint64_t *beta) {
    std::vector<Fr> t((size_t) 1 << n);
    oracle::eqTable2(t, n, &FR(r0, 0), &FR(r1, 0), FR(alpha, 0), FR(beta, 0));
    std::memcpy(out, t.data(), t.size() * 32);
}
void oracle_eq_table1(uint64_t *out, int32_t n, const uint64_t *r, const uint64_t *init) {
    std::vector<Fr> t((size_t) 1 << n);
    oracle::eqTable1(t, n, &FR(r, 0), FR(init, 0));
    std::memcpy(out, t.data(), t.size() * 32);
}
void oracle_phi_table(uint64_t *out, const uint64_t *rx, const uint64_t *scale, int32_t n, int32_t inverse) {
    std::vector<Fr> t((size_t) 1 << n);
    oracle::phiTable(t, &FR(rx, 0), FR(scale, 0), n, inverse != 0);
    size_t cnt = inverse ? t.size() : t.size() >> 1;
    std::memcpy(out, t.data(), cnt * 32);
}
void oracle_ntt(uint64_t *data, int32_t logn, int32_t inverse, uint64_t batch) {
    size_t len = (size_t) 1 << logn;
    std::vector<Fr> a(len);
    for (size_t b = 0; b < batch; ++b) {
        std::memcpy(a.data(), data + 4 * b * len, len * 32);
        oracle::nttInPlace(a, logn, inverse != 0);
        std::memcpy(data + 4 * b * len, a.data(), len * 32);
    }
}

// One quadratic sumcheck round on plain tables (the arithmetic of reference src/prover.cpp:396-426):
// fold V and M (length n) with r unless first, then coefficients of sum_i (V'[2i] + x dV)(M'[2i] + x dM).
// Returns the new length; out = (a, b, c).
uint64_t oracle_round_quadratic(uint64_t *V, uint64_t *M, uint64_t n, const uint64_t *r, int32_t first, uint64_t *out) {
    if (!first) {
        for (size_t j = 0; j < n / 2; ++j) {
            FR(V, j) = FR(V, 2 * j) + FR(r, 0) * (FR(V, 2 * j + 1) - FR(V, 2 * j));
            FR(M, j) = FR(M, 2 * j) + FR(r, 0) * (FR(M, 2 * j + 1) - FR(M, 2 * j));
        }
        n /= 2;
    }
    quadratic_poly acc;
    for (size_t i = 0; i < n / 2; ++i) {
        linear_poly lv(FR(V, 2 * i + 1) - FR(V, 2 * i), FR(V, 2 * i)), lm(FR(M, 2 * i + 1) - FR(M, 2 * i), FR(M, 2 * i));
        acc = acc + lm * lv;
    }
    FR(out, 0) = acc.a; FR(out, 1) = acc.b; FR(out, 2) = acc.c;
    return n;
}

// ---- curve ----
// affine points cross the ABI as 12 x u64 (x, y Montgomery limbs), (0,0) = infinity
void oracle_g1_generators(uint64_t *out, uint64_t n, uint64_t seed) {
    Fr::seedCSPRNG(seed);
    std::vector<G1> g;
    drawGenerators(g, n);
    std::vector<G1Affine> a;
    zkff::batchToAffine(g, a);
    std::memcpy(out, a.data(), n * 96);
}
void oracle_msm(uint64_t *out_affine, const uint64_t *scalars, const uint64_t *bases_affine, uint64_t n) {
    G1 r = zkff::msmCPU(&FR(scalars, 0), reinterpret_cast<const G1Affine *>(bases_affine), n);
    G1Affine a = r.toAffine();
    std::memcpy(out_affine, &a, 96);
}
void oracle_g1_serialize(uint8_t *out48, const uint64_t *affine) {
    G1 p = G1::fromAffine(*reinterpret_cast<const G1Affine *>(affine));
    p.serialize(out48);
}
// 1 = accepted (affine point written), 0 = rejected encoding
int32_t oracle_g1_deserialize(uint64_t *out_affine, const uint8_t *in48, int32_t check_subgroup) {
    G1 p;
    if (!G1::deserialize(p, in48, check_subgroup != 0)) return 0;
    G1Affine a = p.toAffine();
    std::memcpy(out_affine, &a, 96);
    return 1;
}
void oracle_sha256(uint8_t *out32, const uint8_t *in, uint64_t n, uint64_t split) {
    zkff::Sha256 h;                       // absorbed in two pieces to exercise the buffering
    h.update(in, split < n ? split : n);
    if (split < n) h.update(in + split, n - split);
    h.digest(out32);
}
int32_t oracle_g1_on_curve(const uint64_t *affine) {
    return G1::fromAffine(*reinterpret_cast<const G1Affine *>(affine)).isOnCurve() ? 1 : 0;
}
void oracle_g1_mul(uint64_t *out_affine, const uint64_t *affine, const uint64_t *k) {
    G1 r = G1::fromAffine(*reinterpret_cast<const G1Affine *>(affine)) * FR(k, 0);
    G1Affine a = r.toAffine();
    std::memcpy(out_affine, &a, 96);
}
void oracle_g1_add(uint64_t *out_affine, const uint64_t *p, const uint64_t *q) {
    G1 r = G1::fromAffine(*reinterpret_cast<const G1Affine *>(p)) + G1::fromAffine(*reinterpret_cast<const G1Affine *>(q));
    G1Affine a = r.toAffine();
    std::memcpy(out_affine, &a, 96);
}
void oracle_g1_base(uint64_t *out_affine) {
    G1Affine a = G1::generator().toAffine();
    std::memcpy(out_affine, &a, 96);
}
void oracle_fp_to_canonical(uint64_t *out, const uint64_t *in, uint64_t n) {
    for (size_t i = 0; i < n; ++i) reinterpret_cast<const zkff::Fp *>(in + 6 * i)->toCanonical(out + 6 * i);
}

}  // extern "C"

// ORACLE -- TEST INFRASTRUCTURE ONLY (PARITY UNPINNED, see ref_tables.hpp).
// C entry points of liboracle.so: the whole-proof driver of include/zkcnn_api.h backed by the CPU
// restatement, plus array-level reference functions the kernel parity tests compare against.
#include "../zkcnn_amd/csrc/ff/sha256.hpp"
#include "../zkcnn_amd/csrc/ff/blake2s.hpp"
#include "ref_prover.hpp"
#define ZKFIBER_IMPLEMENTATION
#include "session.hpp"

typedef sessionT<oracle::prover> oracleSession;
typedef layer layer_t_alias;

// For the host-logic test of batchSessionT (session.hpp) without a GPU: the CPU prover, except that every round call first gives up the thread when
// it runs on a fiber -- the points at which the HIP-backed prover's lanes yield. K such sessions driven by one batchSessionT must produce
// the transcripts they produce one by one.
struct yieldingProver : public oracle::prover {
    cubic_poly sumcheckDotProdUpdate1(const F &r) { zkfiber::fiber::yield(); return oracle::prover::sumcheckDotProdUpdate1(r); }
    quadratic_poly sumcheckUpdate1(const F &r) { zkfiber::fiber::yield(); return oracle::prover::sumcheckUpdate1(r); }
    quadratic_poly sumcheckUpdate2(const F &r) { zkfiber::fiber::yield(); return oracle::prover::sumcheckUpdate2(r); }
    quadratic_poly sumcheckLiuUpdate(const F &r) { zkfiber::fiber::yield(); return oracle::prover::sumcheckLiuUpdate(r); }
    void sumcheckInitPhase1(const F &x) { zkfiber::fiber::yield(); oracle::prover::sumcheckInitPhase1(x); }
    void sumcheckFinalize1(const F &r, F &c0, F &c1) { zkfiber::fiber::yield(); oracle::prover::sumcheckFinalize1(r, c0, c1); }
};
typedef sessionT<yieldingProver> yieldingSession;

// field elements cross the ABI as 4 x u64 Montgomery limbs
static inline Fr &FR(uint64_t *p, size_t i) { return *reinterpret_cast<Fr *>(p + 4 * i); }
static inline const Fr &FR(const uint64_t *p, size_t i) { return *reinterpret_cast<const Fr *>(p + 4 * i); }

extern "C" {

void *oracle_session_create(const zkcnn_model_desc *desc, int32_t) {
    oracleSession *s = new oracleSession();
    if (!s->build(desc)) { delete s; return nullptr; }
    return s;
}
void *oracle_session_create_calibrated(const zkcnn_model_desc *desc, const int32_t *scales, uint64_t n_scales, int32_t) {
    if (!desc || !scales || !n_scales) return nullptr;
    oracleSession *s = new oracleSession();
    const vector<int> given(scales, scales + n_scales);
    try {
        if (!s->build(desc, &given)) { delete s; return nullptr; }
    } catch (const std::exception &) { delete s; return nullptr; }
    return s;
}
int32_t oracle_session_prove(void *session, uint64_t seed, uint32_t mode, uint8_t *transcript, uint64_t cap,
                             zkcnn_result *out) {
    return ((oracleSession *) session)->prove(seed, mode, transcript, cap, out);
}
// K yielding sessions (oracle_session_create_yielding) in one batchSessionT; returns the driver's pass count in *passes
void *oracle_session_create_yielding(const zkcnn_model_desc *desc) {
    yieldingSession *s = new yieldingSession();
    if (!s->build(desc)) { delete s; return nullptr; }
    return s;
}
void oracle_session_destroy_yielding(void *session) { delete (yieldingSession *) session; }
int32_t oracle_session_prove_yielding(void *session, uint64_t seed, uint32_t mode, uint8_t *transcript, uint64_t cap, zkcnn_result *out) {
    return ((yieldingSession *) session)->prove(seed, mode, transcript, cap, out);       // (not on a fiber: the yields do nothing)
}
int32_t oracle_batch_prove(void *const *sessions, int32_t n, const uint64_t *seeds, uint32_t mode, uint8_t *const *transcripts, const uint64_t *caps,
                           zkcnn_result *out, uint64_t *passes) {
    batchSessionT<yieldingSession> b;
    for (int32_t k = 0; k < n; ++k) b.lanes.push_back((yieldingSession *) sessions[k]);
    const int rc = b.prove(seeds, mode, transcripts, caps, out, []() {});
    if (passes) *passes = b.rounds_of_flushes;
    return rc;
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
// ---- the recorded witness program, interpreted on the CPU (what zk_witness_rerun does in HBM, restated with plain loops) ----
// Session `session` was built for some picture; this replays its program for synthetic picture `picture_seed` and compares EVERY layer
// value with a second session built for that picture from scratch. Returns the number of differing entries (0 = the program is a
// faithful description of the witness), -2 if the picture's values do not fit this session's quantisation scales (nothing to compare), -1 on error.
int64_t oracle_session_replay_program(void *session, uint64_t picture_seed) {
    oracleSession *s = (oracleSession *) session;
    if (!s || !s->nn || !s->has_witness) return -1;
    zkcnn_model_desc d;
    d.model = s->model_name.c_str();
    d.pic_x = s->pic_x; d.pic_y = s->pic_y; d.pic_channel = s->pic_channel; d.pic_cnt = s->pic_cnt;
    d.data_seed = s->data_seed; d.picture_seed = picture_seed;
    oracleSession full;                                    // the same picture from scratch, under this session's scales
    try {
        if (!full.build(&d, &s->statementScales())) return -2;       // its values do not fit them: nothing to compare
    } catch (const std::exception &) { return -1; }
    if (full.statementScales() != s->statementScales()) return -1;

    const layeredCircuit &C = s->p.C;
    const witnessProgram &pg = s->nn->program();
    vector<F> picture;
    if (!s->nn->quantisePicture(s->nn->syntheticPicture(picture_seed), picture)) return -2;
    vector<vector<F>> val(C.size);
    for (int i = 0; i < C.size; ++i) val[i].assign(C.circuit[i].size, F_ZERO);
    val[0] = s->p.val[0];                                  // weights and biases stay; picture and auxiliary witnesses are rewritten
    for (size_t i = 0; i < picture.size(); ++i) val[0][i] = picture[i];
    vector<std::pair<u64, u64>> ranges;
    for (const witnessStep &st : pg.steps) {
        if (st.what == witnessStep::AUX) {
            const vector<F> &src = val[st.layer];
            for (u64 j = st.op_begin; j < st.op_end; ++j) {
                const witnessOp &op = pg.ops[j];
                if (op.op == witnessOp::MAX) {
                    if (j > st.op_begin && pg.ops[j - 1].dst == op.dst) continue;
                    F best = F_ZERO;
                    for (u64 k = j; k < st.op_end && pg.ops[k].dst == op.dst; ++k) {
                        const F &x = src.at(pg.ops[k].src);
                        if (!x.isNegative() && x > best) best = x;
                    }
                    val[0].at(op.dst) = best;
                    continue;
                }
                F x = F_ZERO;
                if (op.op == witnessOp::SUM_BIT)
                    for (i32 k = 0; k < st.win; ++k) x = x + src.at(pg.windows.at(st.win_begin + (u64) op.src * st.win + k));
                else x = src.at(op.src);
                const i64 mag = std::llabs(x.getInt64());
                const bool bit = op.op == witnessOp::SIGN ? x.isNegative() : ((mag >> op.shift) & 1) != 0;
                val[0].at(op.dst) = bit ? F_ONE : F_ZERO;
            }
        } else if (st.what == witnessStep::RANGE) {
            F mx = F_ZERO, mn = F_ZERO;
            for (const F &x : val[st.layer]) {
                if (!x.isNegative()) { if (x > mx) mx = x; }
                else { F nx = -x; if (nx > mn) mn = nx; }
            }
            ranges.push_back(std::make_pair((u64) mx.getInt64(), (u64) mn.getInt64()));
        } else {
            const layer &L = C.circuit[st.layer];
            vector<F> &out = val[st.layer];
            const vector<F> &prev = val[st.layer - 1];
            std::fill(out.begin(), out.end(), F_ZERO);
            if (L.ty == layerType::FFT || L.ty == layerType::IFFT) {
                const size_t len = (size_t) 1 << L.fft_bit_length, lenh = len >> 1;
                vector<F> arr(len);
                if (L.ty == layerType::FFT)
                    for (size_t c = 0, e = 0; e < L.size; c += lenh, e += len) {
                        for (size_t j = 0; j < len; ++j) arr[j] = j < lenh ? prev.at(c + j) : F_ZERO;
                        fft(arr, L.fft_bit_length, false);
                        for (size_t j = 0; j < len; ++j) out[e + j] = arr[j];
                    }
                else
                    for (size_t c = 0, e = 0; c < L.size; c += lenh, e += len) {
                        for (size_t j = 0; j < len; ++j) arr[j] = prev.at(e + j);
                        fft(arr, L.fft_bit_length, true);
                        for (size_t j = 0; j < lenh; ++j) out[c + j] = arr[j];
                    }
            } else if (L.ty == layerType::DOT_PROD) {
                const int fb = L.fft_bit_length;
                for (const binGate &gt : L.bin_gates)
                    for (u32 t = 0; t < (1u << fb); ++t)
                        out[((size_t) gt.g << fb) + t] = out[((size_t) gt.g << fb) + t] + prev[((size_t) gt.u << fb) + t] * prev[((size_t) gt.v << fb) + t];
            } else {
                // gates as they are after initSubset: an operand in layer 0 is an index into the layer's subset (ori_id_u / ori_id_v)
                const u8 id = (u8) st.layer;
                for (const uniGate &gt : L.uni_gates) {
                    const F &x = gt.lu == 0 ? val[0].at(L.ori_id_u.at(gt.u)) : prev.at(gt.u);
                    out[gt.g] = out[gt.g] + (gt.sc ? x * C.two_mul[gt.sc] : x);
                }
                for (const binGate &gt : L.bin_gates) {
                    const F &x = gt.getLayerIdU(id) == 0 ? val[0].at(L.ori_id_u.at(gt.u)) : prev.at(gt.u);
                    const F &y = gt.getLayerIdV(id) == 0 ? val[0].at(L.ori_id_v.at(gt.v)) : prev.at(gt.v);
                    F pr = x * y;
                    out[gt.g] = out[gt.g] + (gt.sc ? pr * C.two_mul[gt.sc] : pr);
                }
                if (!(L.scale == F_ONE)) for (F &x : out) x = x * L.scale;
            }
        }
    }
    if (!s->nn->rangesFitScales(ranges)) return -2;
    int64_t diff = 0;
    for (int i = 0; i < C.size; ++i) {
        if (val[i].size() != full.p.val[i].size()) return -1;
        for (size_t j = 0; j < val[i].size(); ++j) diff += !(val[i][j] == full.p.val[i][j]);
    }
    return diff;
}

int64_t oracle_session_layer_size(void *session, int32_t layer, int32_t *type) {
    if (!session) return -1;
    const layeredCircuit &C = ((oracleSession *) session)->p.C;
    if (layer < 0 || layer >= C.size) return -1;
    if (type) *type = (int32_t) C.circuit[layer].ty;
    return (int64_t) C.circuit[layer].size;
}
// The circuit as the generator + initSubset left it, for tests/test_circuit_cpu.py (an independent restatement of reference src/circuit.cpp:4-88
// in numpy is run on these gate lists): meta = {type, size, n_uni, n_bin, size_u0, size_u1, size_v0, size_v1, bl, bl_u0, bl_u1, bl_v0, bl_v1,
// max_bl_u, max_bl_v, fft_bl, need_phase2, zero_start_id}; uni: 4 words per gate (g, u, lu, sc); bin: 5 (g, u, v, sc, l). NULL arrays: counts only.
int32_t oracle_session_layer_dump(void *session, int32_t layer, int64_t meta[18], uint32_t *uni, uint32_t *bin, uint32_t *ori_u, uint32_t *ori_v) {
    if (!session || !meta) return -1;
    const layeredCircuit &C = ((oracleSession *) session)->p.C;
    if (layer < 0 || layer >= C.size) return -1;
    const layer_t_alias &L = C.circuit[layer];
    const int64_t m[18] = {(int64_t) L.ty, L.size, (int64_t) L.uni_gates.size(), (int64_t) L.bin_gates.size(), L.size_u[0], L.size_u[1], L.size_v[0], L.size_v[1],
                           L.bit_length, L.bit_length_u[0], L.bit_length_u[1], L.bit_length_v[0], L.bit_length_v[1], L.max_bl_u, L.max_bl_v, L.fft_bit_length,
                           L.need_phase2, L.zero_start_id};
    std::memcpy(meta, m, sizeof(m));
    if (uni) for (size_t k = 0; k < L.uni_gates.size(); ++k) { const uniGate &g = L.uni_gates[k]; uni[4 * k] = g.g; uni[4 * k + 1] = g.u; uni[4 * k + 2] = g.lu; uni[4 * k + 3] = g.sc; }
    if (bin) for (size_t k = 0; k < L.bin_gates.size(); ++k) { const binGate &g = L.bin_gates[k]; bin[5 * k] = g.g; bin[5 * k + 1] = g.u; bin[5 * k + 2] = g.v; bin[5 * k + 3] = g.sc; bin[5 * k + 4] = g.l; }
    if (ori_u) std::memcpy(ori_u, L.ori_id_u.data(), L.ori_id_u.size() * 4);
    if (ori_v) std::memcpy(ori_v, L.ori_id_v.data(), L.ori_id_v.size() * 4);
    return 0;
}
// the circuit's constants (tests/test_verifier_python_cpu.py restates the verifier): the two_mul table (n entries, at most cap written) and every layer's scale
int32_t oracle_session_consts(void *session, uint64_t *two_mul, uint64_t cap, uint64_t *n, uint64_t *layer_scales) {
    if (!session || !n) return -1;
    const layeredCircuit &C = ((oracleSession *) session)->p.C;
    *n = C.two_mul.size();
    if (two_mul) for (size_t k = 0; k < C.two_mul.size() && k < cap; ++k) FR(two_mul, k) = C.two_mul[k];
    if (layer_scales) for (int i = 0; i < C.size; ++i) FR(layer_scales, i) = C.circuit[i].scale;
    return 0;
}
// the bytes a non-interactive proof's hash chain starts from: the statement's encoding (host/replay.hpp: fiatShamir::absorbStatement) and the digest of the
// public generators (n_gens of them), as the session's verifier absorbs them before the first message (tests/test_verifier_python_cpu.py hashes them itself)
int32_t oracle_session_fs_statement(void *session, uint64_t n_gens, uint8_t *out, uint64_t cap, uint64_t *len) {
    oracleSession *s = (oracleSession *) session;
    if (!s || !len || !s->nn) return -1;
    fiatShamir fs;
    zkcnn_model_desc d;
    std::memset(&d, 0, sizeof(d));
    d.model = s->model_name.c_str();
    d.pic_x = s->pic_x; d.pic_y = s->pic_y; d.pic_channel = s->pic_channel; d.pic_cnt = s->pic_cnt;
    fs.absorbStatement(d, s->nn->scales(), s->p.C);
    fs.absorb(zkff::publicGeneratorSet(n_gens).digest, 32);
    *len = fs.pendingData().size();
    if (out && *len <= cap) std::memcpy(out, fs.pendingData().data(), *len);
    return 0;
}
int32_t oracle_session_poke(void *session, int32_t layer, uint64_t index, const uint64_t value[4]) {
    oracleSession *s = (oracleSession *) session;
    if (!s || !value || layer < 0 || layer >= s->p.C.size || index >= s->p.val[layer].size()) return -1;
    std::memcpy(&s->p.val[layer][index], value, 32);
    return 0;
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
// ---- the commitment scheme on its own (tests/test_hyrax_cpu.py re-verifies every message with Python integers) ----
// Commits the table Z (2^n_bits entries) over `gens` (n_gens affine points; with `blinds`: the last one is H and every row is blinded), opens it at x
// against `eval` with challenges from the stream seeded by chal_seed (oracle_fr_random(k, chal_seed) gives the same values) and writes every message in
// order: the row commitments, then the opening's messages. Returns the verifier's verdict.
int32_t oracle_hyrax_run(const uint64_t *Z, int32_t n_bits, const uint64_t *gens_affine, uint64_t n_gens, const uint64_t *blinds, const uint64_t *x,
                         const uint64_t *eval, uint64_t chal_seed, uint64_t coin_seed, int32_t stop_len, int32_t tamper_at, uint8_t *out, uint64_t cap,
                         uint64_t *out_len) {
    struct sink : public hyrax_bls12_381::transcriptSink {
        std::vector<u8> bytes;
        void put(const Fr &v) override { size_t o = bytes.size(); bytes.resize(o + 32); v.toBytesLE(&bytes[o]); }
        void put(const G1 &p) override { size_t o = bytes.size(); bytes.resize(o + 48); p.serialize(&bytes[o]); }
    } rec;
    std::vector<Fr> table((size_t) 1 << n_bits), pt(n_bits);
    for (size_t i = 0; i < table.size(); ++i) table[i] = FR(Z, i);
    for (int i = 0; i < n_bits; ++i) pt[i] = FR(x, i);
    std::vector<G1> gens(n_gens);
    for (size_t i = 0; i < n_gens; ++i) gens[i] = G1::fromAffine(*reinterpret_cast<const G1Affine *>(gens_affine + 12 * i));
    std::vector<Fr> bl;
    if (blinds) for (size_t i = 0; i < ((size_t) 1 << (n_bits >> 1)); ++i) bl.push_back(FR(blinds, i));
    Fr::seedCSPRNG(chal_seed);
    zkff::privateCoins().seed(coin_seed);
    oracle::polyProverCPU pp(table, gens, blinds ? &bl : nullptr);
    hyrax_bls12_381::polyVerifier pv(pp, gens, &rec);
    pv.stop_len = (size_t) stop_len;
    pv.tamper_at = tamper_at;
    const bool ok = blinds ? pv.verifyZk(pt, FR(eval, 0)) : pv.verify(pt, FR(eval, 0));
    *out_len = rec.bytes.size();
    if (rec.bytes.size() <= cap) std::memcpy(out, rec.bytes.data(), rec.bytes.size());
    return ok ? 1 : 0;
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

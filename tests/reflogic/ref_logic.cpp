// BUILD CONTAINER ONLY (tests/golden/make_ref_logic_golden.py; needs /root/reference): ONE run of the reference's own protocol logic -- its UNMODIFIED
// src/prover.cpp, src/verifier.cpp, src/polynomial.cpp, src/utils.cpp, src/circuit.cpp, src/neuralNetwork.cpp, src/models.cpp, reached through
// symlinks in a scratch directory (nothing is copied into the repository, nothing travels to the GPU box) -- with this repo's field arithmetic
// underneath (tests/reflogic/shim) and a seeded challenge stream. The 9 value-returning prover calls the verifier makes (Vres,
// sumcheckDotProdUpdate1, sumcheckUpdate1/2, sumcheckDotProdFinalize1, sumcheckFinalize1/2, sumcheckLiuUpdate, sumcheckLiuFinalize) are intercepted
// at LINK time (ld --wrap on their mangled names: no source of the reference is touched) and their results serialised exactly as this repo's
// transcript serialises them (Fr = 32 bytes little-endian canonical; a polynomial = its coefficients a, b, c[, d]; claims in the order the verifier
// receives them). Output: SHA-256 and length of that byte stream = the SUMCHECK PART of a transcript, plus what the reference's verifier said.
// tests/test_reflogic_cpu.py compares the CPU oracle's transcript (same model, same data, same challenge seed) against it.
//
// What this closes: the oracle (oracle/ref_prover.cpp) and every verifier in this repo were written from one reading of the reference's sources; a
// shared misreading of the MESSAGE SEMANTICS (coefficient order, which claim goes first, when a table folds) would pass every other test. What it
// does not do: pin the arithmetic (this repo's ff/ under the reference's logic) or the commitment (a stand-in): the parity grade stays "partial".
//
// usage: ref_logic <model> <pic_cnt> <input file> <challenge seed> [network tokens file (model vgg)]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "verifier.hpp"
#include "models.hpp"
#include "global_var.hpp"
#include "ff/sha256.hpp"
#include "wrap_syms.h"          // generated: the mangled names of the nine methods, read off the reference's own object file

vector<std::string> output_tb(16, "");

static std::vector<uint8_t> rec;
static size_t n_msgs = 0;
static void putF(const F &x) {
    const size_t o = rec.size();
    rec.resize(o + 32);
    x.toBytesLE(&rec[o]);
    ++n_msgs;
}
typedef vector<F>::const_iterator it_t;

// the reference's methods under their link-time aliases (Itanium ABI: a member function is a function of `this` first)
F real_Vres(prover *, const it_t &, u32, u8) asm("__real_" SYM_Vres);
cubic_poly real_DotProdUpdate1(prover *, const F &) asm("__real_" SYM_sumcheckDotProdUpdate1);
quadratic_poly real_Update1(prover *, const F &) asm("__real_" SYM_sumcheckUpdate1);
quadratic_poly real_Update2(prover *, const F &) asm("__real_" SYM_sumcheckUpdate2);
quadratic_poly real_LiuUpdate(prover *, const F &) asm("__real_" SYM_sumcheckLiuUpdate);
void real_DotProdFinalize1(prover *, const F &, F &) asm("__real_" SYM_sumcheckDotProdFinalize1);
void real_Finalize1(prover *, const F &, F &, F &) asm("__real_" SYM_sumcheckFinalize1);
void real_Finalize2(prover *, const F &, F &, F &) asm("__real_" SYM_sumcheckFinalize2);
void real_LiuFinalize(prover *, const F &, F &) asm("__real_" SYM_sumcheckLiuFinalize);

F wrap_Vres(prover *p, const it_t &r, u32 n, u8 k) asm("__wrap_" SYM_Vres);
F wrap_Vres(prover *p, const it_t &r, u32 n, u8 k) { F v = real_Vres(p, r, n, k); putF(v); return v; }
cubic_poly wrap_DotProdUpdate1(prover *p, const F &r) asm("__wrap_" SYM_sumcheckDotProdUpdate1);
cubic_poly wrap_DotProdUpdate1(prover *p, const F &r) { cubic_poly q = real_DotProdUpdate1(p, r); putF(q.a); putF(q.b); putF(q.c); putF(q.d); return q; }
#define WRAP_QUAD(name)                                                                                  \
    quadratic_poly wrap_##name(prover *p, const F &r) asm("__wrap_" SYM_sumcheck##name);                 \
    quadratic_poly wrap_##name(prover *p, const F &r) { quadratic_poly q = real_##name(p, r); putF(q.a); putF(q.b); putF(q.c); return q; }
WRAP_QUAD(Update1)
WRAP_QUAD(Update2)
WRAP_QUAD(LiuUpdate)
void wrap_DotProdFinalize1(prover *p, const F &r, F &c1) asm("__wrap_" SYM_sumcheckDotProdFinalize1);
void wrap_DotProdFinalize1(prover *p, const F &r, F &c1) { real_DotProdFinalize1(p, r, c1); putF(c1); }
void wrap_Finalize1(prover *p, const F &r, F &c0, F &c1) asm("__wrap_" SYM_sumcheckFinalize1);
void wrap_Finalize1(prover *p, const F &r, F &c0, F &c1) { real_Finalize1(p, r, c0, c1); putF(c0); putF(c1); }
void wrap_Finalize2(prover *p, const F &r, F &c0, F &c1) asm("__wrap_" SYM_sumcheckFinalize2);
void wrap_Finalize2(prover *p, const F &r, F &c0, F &c1) { real_Finalize2(p, r, c0, c1); putF(c0); putF(c1); }
void wrap_LiuFinalize(prover *p, const F &r, F &c1) asm("__wrap_" SYM_sumcheckLiuFinalize);
void wrap_LiuFinalize(prover *p, const F &r, F &c1) { real_LiuFinalize(p, r, c1); putF(c1); }

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s model pic_cnt input_file challenge_seed [network_file]\n", argv[0]); return 2; }
    const std::string model = argv[1], in_file = argv[3], net_file = argc > 5 ? argv[5] : "";
    const int pp = atoi(argv[2]);
    const uint64_t seed = strtoull(argv[4], nullptr, 0);
    initPairing(mcl::BLS12_381);
    std::unique_ptr<neuralNetwork> nn;
    if (model == "lenet") nn.reset(new lenet(32, 32, 1, pp, MAX, in_file, "", ""));
    else if (model == "lenet.avg") nn.reset(new lenet(32, 32, 1, pp, AVG, in_file, "", ""));
    else if (model == "vgg") nn.reset(new vgg(32, 32, 3, pp, in_file, "", "", net_file));
    else { fprintf(stderr, "unknown model %s\n", model.c_str()); return 2; }
    prover p;
    nn->create(p, false);
    Fr::seedCSPRNG(seed);                     // the reference draws generator scalars and challenges with setByCSPRNG (src/verifier.cpp:124...279)
    verifier v(&p, p.C);
    const bool ok = v.verify();
    zkff::Sha256 h;
    h.update(rec.data(), rec.size());
    uint8_t d[32];
    h.digest(d);
    char hex[65];
    for (int i = 0; i < 32; ++i) snprintf(hex + 2 * i, 3, "%02x", d[i]);
    printf("{\"layers\": %d, \"input_bits\": %d, \"messages\": %zu, \"bytes\": %zu, \"sha256\": \"%s\", \"reference_verifier_accepted\": %s}\n",
           (int) p.C.size, (int) p.C.circuit[0].bit_length, n_msgs, rec.size(), hex, ok ? "true" : "false");
    return 0;
}

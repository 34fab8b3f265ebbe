// Per-layer digests of what the circuit generator emits (reference src/neuralNetwork.cpp:60-142 `create`, :144-650 the layer
// emitters, src/circuit.cpp:4-88 `initSubset`): layer shapes, gate lists in emission order, subset maps, and the witness values.
//
// ONE source, compiled twice:
//   * -DGATES_REFERENCE (build container only: tests/golden/make_gates_golden.py): against the reference's UNMODIFIED neuralNetwork.cpp /
//     models.cpp / circuit.cpp / utils.cpp / polynomial.cpp, reached through symlinks in a scratch directory, with this repo's
//     <hyrax-bls12-381/polyCommit.hpp> as the field (the upstream submodule is absent). Its output is committed as tests/golden/gates.json.
//   * without it (tests/test_gates_cpu.py, everywhere): against this repo's host/neuralNetwork.cpp etc.
// Gate lists and subset maps are index work and do not depend on the field implementation; the value digests do (both builds use this repo's
// ff/fr.hpp), so they check quantisation / truncation / evaluation ORDER, not the arithmetic. This pins the gate emitter, not the oracle.
//
// usage: gates_digest <model> <pic_cnt> <input file> [network tokens file (model vgg)]      model: lenet | lenet.avg | lenetCifar | vgg | vgg16 | vgg11c
//        gates_digest ... with the environment variable GATES_WRITE_INPUT=<seed> (this repo's build only): the input file is WRITTEN -- the values of
//        the seeded synthetic source, as the generator draws them (neuralNetwork::recordDataTo) -- instead of read
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <memory>
#include "circuit.h"
#include "neuralNetwork.hpp"
#include "models.hpp"
#include "global_var.hpp"
#include "ff/sha256.hpp"
#ifdef GATES_REFERENCE
#include "prover.hpp"
vector<std::string> output_tb(16, "");
typedef prover holder_t;
#else
struct holder_t {
    layeredCircuit C;
    vector<vector<F>> val;
};
#endif

static std::string hex(const zkff::Sha256 &s) {
    uint8_t d[32];
    s.digest(d);
    char b[65];
    for (int i = 0; i < 32; ++i) snprintf(b + 2 * i, 3, "%02x", d[i]);
    return std::string(b, 64);
}
static void put32(zkff::Sha256 &s, uint32_t x) { s.update(&x, 4); }
static void put8(zkff::Sha256 &s, uint8_t x) { s.update(&x, 1); }
static void putF(zkff::Sha256 &s, const F &x) {
    uint64_t c[4];
    x.toCanonical(c);
    s.update(c, 32);
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s model pic_cnt input_file [network_file]\n", argv[0]); return 2; }
    const std::string model = argv[1], in_file = argv[3], net_file = argc > 4 ? argv[4] : "";
    const int pp = atoi(argv[2]);
    std::unique_ptr<neuralNetwork> nn;
    if (model == "lenet") nn.reset(new lenet(32, 32, 1, pp, MAX, in_file, "", ""));
    else if (model == "lenet.avg") nn.reset(new lenet(32, 32, 1, pp, AVG, in_file, "", ""));
    else if (model == "lenetCifar") nn.reset(new lenetCifar(32, 32, 3, pp, MAX, in_file, "", ""));
    else if (model == "vgg") nn.reset(new vgg(32, 32, 3, pp, in_file, "", "", net_file));
    else if (model == "vgg16") nn.reset(new vgg16(32, 32, 3, pp, MAX, in_file, "", ""));
    else if (model == "vgg11c") nn.reset(new vgg11(32, 32, 3, pp, MAX, in_file, "", ""));
    else { fprintf(stderr, "unknown model %s\n", model.c_str()); return 2; }
#ifndef GATES_REFERENCE
    if (const char *seed = getenv("GATES_WRITE_INPUT")) {
        // (the constructor opened the not-yet-existing input file: harmless, the synthetic source replaces the reader)
        nn->useSyntheticData(strtoull(seed, nullptr, 0));
        nn->recordDataTo(in_file);
    }
#endif
    holder_t p;
    nn->create(p, false);
    const layeredCircuit &C = p.C;
    printf("{\"layers\": %d, \"two_mul\": %zu, \"layer\": [\n", (int) C.size, C.two_mul.size());
    for (int i = 0; i < (int) C.size; ++i) {
        const layer &L = C.circuit[i];
        zkff::Sha256 hu, hb, hou, hov, hv;
        for (const uniGate &g : L.uni_gates) { put32(hu, g.g); put32(hu, g.u); put8(hu, g.lu); put8(hu, g.sc); }
        for (const binGate &g : L.bin_gates) { put32(hb, g.g); put32(hb, g.u); put32(hb, g.v); put8(hb, g.sc); put8(hb, g.l); }
        for (u32 x : L.ori_id_u) put32(hou, x);
        for (u32 x : L.ori_id_v) put32(hov, x);
        for (const F &x : p.val[i]) putF(hv, x);
        zkff::Sha256 hs;
        putF(hs, L.scale);
        printf(" {\"i\": %d, \"ty\": %d, \"size\": %u, \"bit_length\": %d, \"size_u\": [%u, %u], \"size_v\": [%u, %u], \"bl_u\": [%d, %d], \"bl_v\": [%d, %d], "
               "\"max_bl_u\": %d, \"max_bl_v\": %d, \"need_phase2\": %d, \"zero_start_id\": %u, \"fft_bl\": %d, \"n_uni\": %zu, \"n_bin\": %zu, \"n_val\": %zu, "
               "\"uni\": \"%s\", \"bin\": \"%s\", \"ori_u\": \"%s\", \"ori_v\": \"%s\", \"val\": \"%s\", \"scale\": \"%s\"}%s\n",
               i, (int) L.ty, L.size, (int) L.bit_length, L.size_u[0], L.size_u[1], L.size_v[0], L.size_v[1], (int) L.bit_length_u[0], (int) L.bit_length_u[1],
               (int) L.bit_length_v[0], (int) L.bit_length_v[1], (int) L.max_bl_u, (int) L.max_bl_v, L.need_phase2 ? 1 : 0,
               L.zero_start_id, (int) L.fft_bit_length,
               L.uni_gates.size(), L.bin_gates.size(), p.val[i].size(), hex(hu).c_str(), hex(hb).c_str(), hex(hou).c_str(), hex(hov).c_str(), hex(hv).c_str(),
               hex(hs).c_str(), i + 1 < (int) C.size ? "," : "");
    }
    printf("]}\n");
    return 0;
}

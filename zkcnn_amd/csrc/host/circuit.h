// Layered arithmetic circuit shared by prover and verifier. Field names and meanings follow the
// reference data model (reference src/circuit.h:15-88) because neuralNetwork fills it and the
// verifier walks it; the code is this repo's own.
//
//   uniGate:  out[g] += in_{lu}[u] * two_mul[sc]
//   binGate:  out[g] += in_U[u] * in_V[v] * two_mul[sc];  operand layers are encoded in `l`:
//             l == 0 -> both in layer 0, l == 1 -> both in layer i-1, l == 2 -> u in i-1, v in 0.
#pragma once
#include "global_var.hpp"

struct uniGate {
    u32 g, u;
    u8 lu, sc;
    u8 pad_[2] = {0, 0};                 // explicit zero padding: records are uploaded and hashed as they lie in memory (12 bytes)
    uniGate(u32 out, u32 in, u8 in_layer, u8 scale_id) : g(out), u(in), lu(in_layer), sc(scale_id) {}
};

struct binGate {
    u32 g, u, v;
    u8 sc, l;
    u8 pad_[2] = {0, 0};                 // 16 bytes
    binGate(u32 out, u32 in_u, u32 in_v, u8 scale_id, u8 layers) : g(out), u(in_u), v(in_v), sc(scale_id), l(layers) {}
    u8 getLayerIdU(u8 layer_id) const { return l == 0 ? 0 : (u8) (layer_id - 1); }
    u8 getLayerIdV(u8 layer_id) const { return (l & 1) ? (u8) (layer_id - 1) : 0; }
};

enum class layerType {
    INPUT, FFT, IFFT, ADD_BIAS, RELU, Sqr, OPT_AVG_POOL, MAX_POOL, AVG_POOL, DOT_PROD, PADDING, FCONN, NCONV,
    NCONV_MUL, NCONV_ADD
};

class layer {
public:
    layerType ty = layerType::INPUT;
    u32 size = 0;                        // number of output gates
    u32 size_u[2] = {0, 0};              // live entries of the u-tables: [0] layer-0 subset, [1] previous layer
    u32 size_v[2] = {0, 0};
    i8 bit_length = 0;
    i8 bit_length_u[2] = {-1, -1};       // -1: table absent
    i8 bit_length_v[2] = {-1, -1};
    i8 max_bl_u = 0, max_bl_v = 0;       // number of sumcheck rounds of phase 1 / phase 2
    bool need_phase2 = false;
    u32 zero_start_id = 0;               // rows >= this are constraint rows (must evaluate to 0)
    std::vector<uniGate> uni_gates;
    std::vector<binGate> bin_gates;
    // a STRUCTURE COPY of a circuit (layeredCircuit::structureCopy: what a session keeps that shares another session's resident circuit) holds
    // no gate lists, only how many there were: everything but the host-side wiring predicates works from the counts
    u64 n_uni_kept = 0, n_bin_kept = 0;
    u64 uniCount() const { return uni_gates.empty() ? n_uni_kept : (u64) uni_gates.size(); }
    u64 binCount() const { return bin_gates.empty() ? n_bin_kept : (u64) bin_gates.size(); }
    bool gatesDropped() const { return (uni_gates.empty() && n_uni_kept) || (bin_gates.empty() && n_bin_kept); }
    vector<u32> ori_id_u, ori_id_v;      // subset index -> layer-0 index
    i8 fft_bit_length = -1;
    F scale = F_ONE;                     // 1/N for IFFT, 1/pool^2 for average pooling

    void updateSize() {
        max_bl_u = std::max(bit_length_u[0], bit_length_u[1]);
        max_bl_v = need_phase2 ? std::max(bit_length_v[0], bit_length_v[1]) : (i8) 0;
    }
};

class layeredCircuit {
public:
    vector<layer> circuit;
    u8 size = 0;
    vector<F> two_mul;                   // [k] = 2^k (k <= q), [q+1+k] = -2^k

    void init(u8 q_bit_size, u8 layer_cnt);
    void initSubset();
    // SHA-256 tree over the gate lists and subset maps of every layer (leaves of 2^20 records, hashed on several host threads);
    // computed on first use and kept: what a Fiat-Shamir transcript absorbs as "the wiring"
    const uint8_t *wiringDigest() const;
    // the circuit without its gate lists (shapes, subset maps, scales, the wiring digest): ~2% of the memory of a vgg11 circuit
    layeredCircuit structureCopy() const;
    bool gatesDropped() const { for (const layer &L : circuit) if (L.gatesDropped()) return true; return false; }
private:
    mutable uint8_t wiring_digest[32];
    mutable bool wiring_digest_ok = false;
};

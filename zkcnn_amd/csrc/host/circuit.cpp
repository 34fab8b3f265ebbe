#include "circuit.h"
#include <cstring>
#include "utils.hpp"

// two_mul table and layer storage (behaviour of reference src/circuit.cpp:90-100)
void layeredCircuit::init(u8 q_bit_size, u8 layer_cnt) {
    const int q = q_bit_size;
    two_mul.assign((size_t) 2 * (q + 1), F_ZERO);
    F pw = F_ONE;
    for (int k = 0; k <= q; ++k) {
        two_mul[k] = pw;
        two_mul[q + 1 + k] = -pw;
        pw = pw + pw;
    }
    size = layer_cnt;
    circuit.assign(size, layer());
}

namespace {
// Dense renumbering of the layer-0 wires one layer touches, in first-use order.
struct subsetMap {
    vector<u32> stamp, slot;
    explicit subsetMap(size_t n) : stamp(n, 0), slot(n, 0) {}
    u32 place(u32 wire, u32 layer_id, vector<u32> &ori, u32 &count) {
        if (stamp[wire] != layer_id) {
            stamp[wire] = layer_id;
            slot[wire] = count++;
            ori.push_back(wire);
        }
        return slot[wire];
    }
};
}

// Compact every layer's references into layer 0 to a dense subset (ori_id_u / ori_id_v) and fix
// the table bit lengths the sumcheck uses. Behaviour of reference src/circuit.cpp:4-88: uni gates
// are visited before bin gates, and first-use order defines the subset numbering (that order is
// visible in the transcript, so it is kept).
void layeredCircuit::initSubset() {
    subsetMap mu(circuit[0].size), mv(circuit[0].size);
    for (u32 i = 1; i < size; ++i) {
        layer &cur = circuit[i];
        const layer &prev = circuit[i - 1];
        const bool is_fft = cur.ty == layerType::FFT, is_ifft = cur.ty == layerType::IFFT;
        bool u_in_prev = is_fft || is_ifft, v_in_prev = false;

        for (uniGate &gt : cur.uni_gates) {
            if (gt.lu == 0) gt.u = mu.place(gt.u, i, cur.ori_id_u, cur.size_u[0]);
            else u_in_prev = true;
        }
        for (binGate &gt : cur.bin_gates) {
            if (gt.getLayerIdU(i) == 0) gt.u = mu.place(gt.u, i, cur.ori_id_u, cur.size_u[0]);
            else u_in_prev = true;
            if (gt.getLayerIdV(i) == 0) gt.v = mv.place(gt.v, i, cur.ori_id_v, cur.size_v[0]);
            else v_in_prev = true;
        }
        cur.bit_length_u[0] = ceilPow2BitLength(cur.size_u[0]);
        cur.bit_length_v[0] = ceilPow2BitLength(cur.size_v[0]);

        if (!u_in_prev) { cur.size_u[1] = 0; cur.bit_length_u[1] = -1; }
        else if (is_fft) { cur.bit_length_u[1] = cur.fft_bit_length - 1; cur.size_u[1] = 1u << cur.bit_length_u[1]; }
        else if (is_ifft) { cur.bit_length_u[1] = cur.fft_bit_length; cur.size_u[1] = 1u << cur.bit_length_u[1]; }
        else { cur.size_u[1] = prev.size; cur.bit_length_u[1] = prev.bit_length; }

        if (!v_in_prev) { cur.size_v[1] = 0; cur.bit_length_v[1] = -1; }
        else if (cur.ty == layerType::DOT_PROD) {
            cur.size_v[1] = prev.size >> cur.fft_bit_length;
            cur.bit_length_v[1] = prev.bit_length - cur.fft_bit_length;
        } else { cur.size_v[1] = prev.size; cur.bit_length_v[1] = prev.bit_length; }
        cur.updateSize();
    }
}

// ---- wiring digest (Fiat-Shamir statement binding; not on any timed path) ----
#include <thread>
#include "../ff/sha256.hpp"

namespace {
struct leafJob { const uint8_t *p; size_t n; uint8_t out[32]; };
}

const uint8_t *layeredCircuit::wiringDigest() const {
    if (wiring_digest_ok) return wiring_digest;
    // leaves: consecutive spans of at most 2^20 records of each list, in layer order: uni gates (12 B), bin gates (16 B), ori_id_u, ori_id_v (4 B)
    std::vector<leafJob> jobs;
    auto add = [&](const void *base, size_t count, size_t rec) {
        const uint8_t *b = static_cast<const uint8_t *>(base);
        const size_t span = (size_t) 1 << 20;
        for (size_t i = 0; i < count; i += span) {
            leafJob j;
            j.p = b + i * rec;
            j.n = std::min(span, count - i) * rec;
            jobs.push_back(j);
        }
    };
    static_assert(sizeof(uniGate) == 12 && sizeof(binGate) == 16, "gate records are hashed as they lie in memory");
    for (int i = 0; i < size; ++i) {
        const layer &L = circuit[i];
        add(L.uni_gates.data(), L.uni_gates.size(), sizeof(uniGate));
        add(L.bin_gates.data(), L.bin_gates.size(), sizeof(binGate));
        add(L.ori_id_u.data(), L.ori_id_u.size(), 4);
        add(L.ori_id_v.data(), L.ori_id_v.size(), 4);
    }
    // (the records carry explicit zero padding bytes, circuit.h, so raw memory is a canonical encoding)
    unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (jobs.size() < 4) nthreads = 1;
    auto work = [&](unsigned t) {
        for (size_t k = t; k < jobs.size(); k += nthreads) {
            zkff::Sha256 h;
            h.update(jobs[k].p, jobs[k].n);
            h.digest(jobs[k].out);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nthreads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    zkff::Sha256 top;
    const uint64_t header[2] = {(uint64_t) size, (uint64_t) jobs.size()};
    top.update(header, sizeof(header));
    for (const leafJob &j : jobs) {
        const uint64_t n = j.n;
        top.update(&n, 8);
        top.update(j.out, 32);
    }
    top.digest(wiring_digest);
    wiring_digest_ok = true;
    return wiring_digest;
}

layeredCircuit layeredCircuit::structureCopy() const {
    (void) wiringDigest();                       // computed from the gate lists while they are there; the copy carries it
    layeredCircuit out;
    out.size = size;
    out.two_mul = two_mul;
    std::memcpy(out.wiring_digest, wiring_digest, 32);
    out.wiring_digest_ok = wiring_digest_ok;
    out.circuit.resize(circuit.size());
    for (size_t i = 0; i < circuit.size(); ++i) {
        const layer &S = circuit[i];
        layer &D = out.circuit[i];
        D.ty = S.ty; D.size = S.size;
        for (int b = 0; b < 2; ++b) {
            D.size_u[b] = S.size_u[b]; D.size_v[b] = S.size_v[b];
            D.bit_length_u[b] = S.bit_length_u[b]; D.bit_length_v[b] = S.bit_length_v[b];
        }
        D.bit_length = S.bit_length; D.max_bl_u = S.max_bl_u; D.max_bl_v = S.max_bl_v;
        D.need_phase2 = S.need_phase2; D.zero_start_id = S.zero_start_id;
        D.n_uni_kept = S.uniCount(); D.n_bin_kept = S.binCount();
        D.ori_id_u = S.ori_id_u; D.ori_id_v = S.ori_id_v;
        D.fft_bit_length = S.fft_bit_length; D.scale = S.scale;
    }
    return out;
}

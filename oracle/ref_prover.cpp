// ORACLE -- TEST INFRASTRUCTURE ONLY (PARITY UNPINNED, see ref_tables.hpp). Each function names the
// reference lines it restates.
#include "ref_prover.hpp"

namespace oracle {

static inline linear_poly lerp(const F &at0, const F &at1) { return linear_poly(at1 - at0, at0); }   // prover.cpp:13-15

void prover::init() {                                       // prover.cpp:17-21
    proof_size = 0;
    prove_timer.clear();
    r_u.assign(C.size + 1, vector<F>());
    r_v.assign(C.size + 1, vector<F>());
}

void prover::sumcheckInitAll(const vector<F>::const_iterator &r_0_from_v) {   // prover.cpp:28-36
    sumcheck_id = C.size;
    i8 last_bl = C.circuit[sumcheck_id - 1].bit_length;
    r_u[sumcheck_id].resize(last_bl);
    prove_timer.start();
    for (int i = 0; i < last_bl; ++i) r_u[sumcheck_id][i] = r_0_from_v[i];
    prove_timer.stop();
}

void prover::sumcheckInit(const F &alpha_0, const F &beta_0) {                // prover.cpp:43-52
    prove_timer.start();
    alpha = alpha_0;
    beta = beta_0;
    zkSetWeights(alpha_0, beta_0);
    r_0 = r_u[sumcheck_id].data();
    r_1 = r_v[sumcheck_id].data();
    --sumcheck_id;
    prove_timer.stop();
}

void prover::sumcheckDotProdInitPhase1() {                                    // prover.cpp:57-95
    const layer &cur = C.circuit[sumcheck_id];
    const i8 fft_bl = cur.fft_bit_length;
    total[0] = 1u << fft_bl;
    total[1] = 1u << cur.bit_length_u[1];
    total_size[1] = cur.size_u[1];
    const u32 fft_len = total[0];

    r_u[sumcheck_id].resize(cur.max_bl_u);
    V_mult[0].assign(total[1], linear_poly());
    V_mult[1].assign(total[1], linear_poly());
    mult_array[1].assign(total[0], linear_poly());
    beta_gs.resize((size_t) 1 << fft_bl);

    prove_timer.start();
    eqTable1(beta_gs, fft_bl, r_0, Fr::one());
    for (u32 t = 0; t < fft_len; ++t) mult_array[1][t] = beta_gs[t];
    const vector<F> &prev = val[sumcheck_id - 1];
    for (u32 u = 0; u < total[1]; ++u)
        if (u < cur.size_u[1]) V_mult[1][u] = prev[u];
    for (const binGate &gate : cur.bin_gates)
        for (u32 t = 0; t < fft_len; ++t) {
            u32 iu = gate.u << fft_bl | t, iv = gate.v << fft_bl | t;
            V_mult[0][iu].b = V_mult[0][iu].b + beta_g[gate.g] * prev[iv];
        }
    round = 0;
    absorbed[0] = absorbed[1] = false;
    zkBeginInstance();            // zero-knowledge mode: the layer's masking polynomial (zk_mask.hpp); a no-op otherwise
    prove_timer.stop();
}

cubic_poly prover::sumcheckDotProdUpdate1(const F &previous_random) {        // prover.cpp:103-144
    prove_timer.start();
    if (round) r_u[sumcheck_id].at(round - 1) = previous_random;
    ++round;
    auto &tm = mult_array[1];
    auto &t0 = V_mult[0], &t1 = V_mult[1];

    if (total[0] == 1) tm[0] = tm[0].eval(previous_random);
    else for (u32 i = 0; i < (total[0] >> 1); ++i)
        tm[i] = lerp(tm[i << 1].eval(previous_random), tm[i << 1 | 1].eval(previous_random));
    total[0] >>= 1;

    cubic_poly ret;
    for (u32 i = 0; i < (total[1] >> 1); ++i) {
        u32 g0 = i << 1, g1 = i << 1 | 1;
        if (g0 >= total_size[1]) { t0[i].clear(); t1[i].clear(); continue; }
        if (g1 >= total_size[1]) { t0[g1].clear(); t1[g1].clear(); }
        t0[i] = lerp(t0[g0].eval(previous_random), t0[g1].eval(previous_random));
        t1[i] = lerp(t1[g0].eval(previous_random), t1[g1].eval(previous_random));
        const linear_poly &mm = total[0] ? tm[i & (total[0] - 1)] : tm[0];
        ret = ret + (mm * t1[i]) * t0[i];
    }
    proof_size += F_BYTE_SIZE * (3 + (!ret.a.isZero()));
    total[1] >>= 1;
    total_size[1] = (total_size[1] + 1) >> 1;
    zkMask(ret, previous_random);
    prove_timer.stop();
    return ret;
}

void prover::sumcheckDotProdFinalize1(const F &previous_random, F &claim_1) { // prover.cpp:146-153
    prove_timer.start();
    r_u[sumcheck_id].at(round - 1) = previous_random;
    claim_1 = V_mult[1][0].eval(previous_random);
    F none = F_ZERO, added[2];
    zkMaskClaims(zkmask::SLOT_U0, previous_random, none, claim_1, added);       // zero-knowledge mode: the claim leaves masked, phase 2 is about the masked value
    V_u1 = claim_1 * mult_array[1][0].eval(previous_random);
    prove_timer.stop();
    proof_size += F_BYTE_SIZE * 1;
}

void prover::sumcheckInitPhase1(const F &relu_rou_0) {                        // prover.cpp:155-239
    const layer &cur = C.circuit[sumcheck_id];
    total[0] = ~cur.bit_length_u[0] ? 1u << cur.bit_length_u[0] : 0;
    total_size[0] = cur.size_u[0];
    total[1] = ~cur.bit_length_u[1] ? 1u << cur.bit_length_u[1] : 0;
    total_size[1] = cur.size_u[1];

    r_u[sumcheck_id].resize(cur.max_bl_u);
    for (int b = 0; b < 2; ++b) {
        V_mult[b].assign(total[b], linear_poly());
        mult_array[b].assign(total[b], linear_poly());
    }
    beta_g.resize((size_t) 1 << cur.bit_length);
    const bool is_fft = cur.ty == layerType::FFT, is_ifft = cur.ty == layerType::IFFT;
    if (cur.ty == layerType::PADDING) beta_gs.resize((size_t) 1 << cur.fft_bit_length);
    if (is_fft || is_ifft) beta_gs.resize(total[1]);

    prove_timer.start();
    relu_rou = relu_rou_0;
    add_term.clear();

    if (is_fft || is_ifft) {                                                  // :182-203
        const i8 fft_bl = cur.fft_bit_length, fft_blh = fft_bl - 1;
        const i8 cnt_bl = is_fft ? cur.bit_length - fft_bl : cur.bit_length - fft_blh;
        const u32 cnt_len = cur.size >> (is_fft ? fft_bl : fft_blh);
        if (is_fft) eqTable2(beta_g, cnt_bl, r_0 + fft_bl, r_1, alpha, beta);
        else eqTable1(beta_g, cnt_bl, r_0 + fft_blh, alpha);
        const vector<F> &prev = val[sumcheck_id - 1];
        for (u32 u = 0; u < total[1]; ++u) {
            if (u >= cur.size_u[1]) continue;
            F acc = F_ZERO;
            for (u32 g = 0; g < cnt_len; ++g) acc = acc + prev[(g << cur.max_bl_u) | u] * beta_g[g];
            V_mult[1][u].b = acc;
        }
        beta_gs.resize(total[1]);
        phiTable(beta_gs, r_0, cur.scale, fft_bl, is_ifft);
        for (u32 u = 0; u < total[1]; ++u) mult_array[1][u] = beta_gs[u];
    } else {                                                                  // :204-233
        for (int b = 0; b < 2; ++b) {
            u8 dep = !b ? 0 : sumcheck_id - 1;
            for (u32 u = 0; u < total[b]; ++u)
                if (u < cur.size_u[b]) V_mult[b][u] = cirValue(dep, cur.ori_id_u, u);
        }
        if (cur.ty == layerType::PADDING) {                                   // :214-219 (re-uses the FFT layer's beta_g)
            const i8 fft_blh = cur.fft_bit_length - 1;
            const size_t lenh = (size_t) 1 << fft_blh;
            eqTable1(beta_gs, fft_blh, r_0, Fr::one());
            for (long g = (1L << cur.bit_length) - 1; g >= 0; --g)
                beta_g[g] = beta_g[g >> fft_blh] * beta_gs[g & (lenh - 1)];
        } else eqTable2(beta_g, cur.bit_length, r_0, r_1, alpha * cur.scale, beta * cur.scale);
        if (cur.zero_start_id < cur.size)
            for (size_t g = cur.zero_start_id; g < ((size_t) 1 << cur.bit_length); ++g) beta_g[g] = beta_g[g] * relu_rou;

        for (const uniGate &gate : cur.uni_gates) {
            bool idx = gate.lu != 0;
            mult_array[idx][gate.u].b = mult_array[idx][gate.u].b + beta_g[gate.g] * C.two_mul[gate.sc];
        }
        for (const binGate &gate : cur.bin_gates) {
            bool idx = gate.getLayerIdU(sumcheck_id) != 0;
            F val_lv = cirValue(gate.getLayerIdV(sumcheck_id), cur.ori_id_v, gate.v);
            mult_array[idx][gate.u].b = mult_array[idx][gate.u].b + val_lv * beta_g[gate.g] * C.two_mul[gate.sc];
        }
    }
    round = 0;
    zkBeginInstance();            // zero-knowledge mode: the next masking polynomial (zk_mask.hpp); a no-op otherwise
    prove_timer.stop();
}

void prover::sumcheckInitPhase2() {                                           // prover.cpp:241-310
    const layer &cur = C.circuit[sumcheck_id];
    total[0] = ~cur.bit_length_v[0] ? 1u << cur.bit_length_v[0] : 0;
    total_size[0] = cur.size_v[0];
    total[1] = ~cur.bit_length_v[1] ? 1u << cur.bit_length_v[1] : 0;
    total_size[1] = cur.size_v[1];
    const i8 fft_bl = cur.fft_bit_length, cnt_bl = cur.max_bl_v;

    r_v[sumcheck_id].resize(cur.max_bl_v);
    for (int b = 0; b < 2; ++b) {
        V_mult[b].assign(total[b], linear_poly());
        mult_array[b].assign(total[b], linear_poly());
    }
    if (cur.ty == layerType::DOT_PROD) {
        beta_u.resize((size_t) 1 << cnt_bl);
        beta_gs.resize((size_t) 1 << fft_bl);
    } else beta_u.resize((size_t) 1 << cur.max_bl_u);

    prove_timer.start();
    add_term.clear();
    if (cur.ty == layerType::DOT_PROD) {                                      // :272-288
        const u32 fft_len = 1u << fft_bl;
        eqTable1(beta_u, cnt_bl, r_u[sumcheck_id].data() + fft_bl, Fr::one());
        eqTable1(beta_gs, fft_bl, r_u[sumcheck_id].data(), Fr::one());
        const vector<F> &prev = val[sumcheck_id - 1];
        for (u32 v = 0; v < total[1]; ++v) {
            if (v >= cur.size_v[1]) continue;
            F acc = F_ZERO;
            for (u32 t = 0; t < fft_len; ++t) acc = acc + prev[(v << fft_bl) | t] * beta_gs[t];
            V_mult[1][v].b = acc;
        }
        for (const binGate &gate : cur.bin_gates)
            mult_array[1][gate.v].b = mult_array[1][gate.v].b + beta_g[gate.g] * beta_u[gate.u] * V_u1;
    } else {                                                                  // :289-306
        eqTable1(beta_u, cur.max_bl_u, r_u[sumcheck_id].data(), Fr::one());
        for (int b = 0; b < 2; ++b) {
            u8 dep = !b ? 0 : sumcheck_id - 1;
            for (u32 v = 0; v < total[b]; ++v)
                if (v < cur.size_v[b]) V_mult[b][v] = cirValue(dep, cur.ori_id_v, v);
        }
        for (const uniGate &gate : cur.uni_gates) {
            const F &V_u = !gate.lu ? V_u0 : V_u1;
            add_term = add_term + beta_g[gate.g] * beta_u[gate.u] * V_u * C.two_mul[gate.sc];
        }
        for (const binGate &gate : cur.bin_gates) {
            bool idx = gate.getLayerIdV(sumcheck_id);
            const F &V_u = !gate.getLayerIdU(sumcheck_id) ? V_u0 : V_u1;
            mult_array[idx][gate.v].b = mult_array[idx][gate.v].b + beta_g[gate.g] * beta_u[gate.u] * V_u * C.two_mul[gate.sc];
        }
    }
    round = 0;
    absorbed[0] = absorbed[1] = false;
    prove_timer.stop();
}

void prover::sumcheckLiuInit(const vector<F> &s_u, const vector<F> &s_v) {    // prover.cpp:312-358
    sumcheck_id = 0;
    total[1] = 1u << C.circuit[0].bit_length;
    total_size[1] = C.circuit[0].size;
    r_u[0].resize(C.circuit[0].bit_length);
    mult_array[1].assign(total[1], linear_poly());
    V_mult[1].assign(total[1], linear_poly());

    i8 max_bl = 0;
    for (int i = 1; i < C.size; ++i)
        max_bl = std::max(max_bl, std::max(C.circuit[i].bit_length_u[0], C.circuit[i].bit_length_v[0]));
    beta_g.resize((size_t) 1 << max_bl);

    prove_timer.start();
    add_term.clear();
    for (u32 g = 0; g < total[1]; ++g)
        if (g < total_size[1]) V_mult[1][g] = val[0][g];
    for (u8 i = 1; i < C.size; ++i) {
        const layer &L = C.circuit[i];
        if (~L.bit_length_u[0]) {
            eqTable1(beta_g, L.bit_length_u[0], r_u[i].data(), s_u[i - 1]);
            for (u32 hu = 0; hu < L.size_u[0]; ++hu) {
                u32 u = L.ori_id_u[hu];
                mult_array[1][u].b = mult_array[1][u].b + beta_g[hu];
            }
        }
        if (~L.bit_length_v[0]) {
            eqTable1(beta_g, L.bit_length_v[0], r_v[i].data(), s_v[i - 1]);
            for (u32 hv = 0; hv < L.size_v[0]; ++hv) {
                u32 v = L.ori_id_v[hv];
                mult_array[1][v].b = mult_array[1][v].b + beta_g[hv];
            }
        }
    }
    round = 0;
    absorbed[0] = absorbed[1] = false;
    zkBeginLiu(s_u, s_v);         // zero-knowledge mode: the Liu sumcheck's masking polynomial
    prove_timer.stop();
}

quadratic_poly prover::sumcheckUpdate1(const F &previous_random) { return update(previous_random, r_u[sumcheck_id]); }
quadratic_poly prover::sumcheckUpdate2(const F &previous_random) { return update(previous_random, r_v[sumcheck_id]); }

quadratic_poly prover::update(const F &previous_random, vector<F> &r_arr) {   // prover.cpp:368-383
    prove_timer.start();
    if (round) r_arr.at(round - 1) = previous_random;
    ++round;
    quadratic_poly ret;
    add_term = add_term * (Fr::one() - previous_random);
    for (int b = 0; b < 2; ++b) if (absorbed[b]) absorbed_m[b] = absorbed_m[b] * (Fr::one() - previous_random);
    for (int b = 0; b < 2; ++b) ret = ret + updateEach(previous_random, b);
    ret = ret + quadratic_poly(F_ZERO, -add_term, add_term);
    zkMask(ret, previous_random);
    prove_timer.stop();
    proof_size += F_BYTE_SIZE * 3;
    return ret;
}

quadratic_poly prover::sumcheckLiuUpdate(const F &previous_random) {          // prover.cpp:385-394
    prove_timer.start();
    ++round;
    quadratic_poly ret = updateEach(previous_random, true);
    zkMask(ret, previous_random);
    prove_timer.stop();
    proof_size += F_BYTE_SIZE * 3;
    return ret;
}

quadratic_poly prover::updateEach(const F &previous_random, bool idx) {       // prover.cpp:396-426
    auto &tm = mult_array[idx];
    auto &tv = V_mult[idx];
    if (total[idx] == 1) {
        tv[0] = tv[0].eval(previous_random);
        tm[0] = tm[0].eval(previous_random);
        add_term = add_term + tv[0].b * tm[0].b;
        absorbed[idx] = true;                 // (zero-knowledge mode reads the multiplier of an absorbed pair: zkTailPairs)
        absorbed_m[idx] = tm[0].b;
    }
    quadratic_poly ret;
    for (u32 i = 0; i < (total[idx] >> 1); ++i) {
        u32 g0 = i << 1, g1 = i << 1 | 1;
        if (g0 >= total_size[idx]) { tv[i].clear(); tm[i].clear(); continue; }
        if (g1 >= total_size[idx]) { tv[g1].clear(); tm[g1].clear(); }
        tv[i] = lerp(tv[g0].eval(previous_random), tv[g1].eval(previous_random));
        tm[i] = lerp(tm[g0].eval(previous_random), tm[g1].eval(previous_random));
        ret = ret + tm[i] * tv[i];
    }
    total[idx] >>= 1;
    total_size[idx] = (total_size[idx] + 1) >> 1;
    return ret;
}

// ---- zero-knowledge mode (zk_mask.hpp): what the masks of a phase's last round need from the tables ----
void prover::zkRawRound(int kind, const F &prev_r, F c[5]) {
    raw_kind = kind;
    if (kind == zkmask::PH_DOT1) {
        const cubic_poly q = sumcheckDotProdUpdate1(prev_r);
        c[0] = q.d; c[1] = q.c; c[2] = q.b; c[3] = q.a;
        return;
    }
    const quadratic_poly q = kind == zkmask::PH_ONE ? sumcheckUpdate1(prev_r) : kind == zkmask::PH_TWO ? sumcheckUpdate2(prev_r) : sumcheckLiuUpdate(prev_r);
    c[0] = q.c; c[1] = q.b; c[2] = q.a;
}

void prover::zkTailPairs(F A[6]) {
    for (int k = 0; k < 6; ++k) A[k] = F_ZERO;
    if (raw_kind == zkmask::PH_DOT1) {
        // phase 1 of a DOT_PROD layer: the claimed operand is V_mult[1]; what multiplies it is (periodic table) x (the other operand's sums)
        const quadratic_poly m = mult_array[1][0] * V_mult[0][0];
        A[3] = m.c; A[4] = m.b; A[5] = m.a;
        return;
    }
    for (int b = 0; b < 2; ++b) {
        if (total[b] == 1 && !mult_array[b].empty()) { A[3 * b] = mult_array[b][0].b; A[3 * b + 1] = mult_array[b][0].a; }
        else if (absorbed[b]) { A[3 * b] = absorbed_m[b]; A[3 * b + 1] = -absorbed_m[b]; }
    }
}

F prover::Vres(const vector<F>::const_iterator &r, u32 output_size, u8 r_size) {   // prover.cpp:434-457
    prove_timer.start();
    vector<F> output(std::max<size_t>(output_size, 1));
    for (u32 i = 0; i < output_size; ++i) output[i] = val[C.size - 1][i];
    u32 whole = 1u << r_size;
    for (u8 i = 0; i < r_size; ++i) {
        for (u32 j = 0; j < (whole >> 1); ++j) {
            if (j > 0) output[j].clear();
            if ((j << 1) < output_size) output[j] = output[j << 1] * (Fr::one() - r[i]);
            if ((j << 1 | 1) < output_size) output[j] = output[j] + output[j << 1 | 1] * r[i];
        }
        whole >>= 1;
    }
    F res = output[0];
    prove_timer.stop();
    proof_size += F_BYTE_SIZE;
    return res;
}

void prover::sumcheckFinalize1(const F &previous_random, F &claim_0, F &claim_1) {   // prover.cpp:459-471
    prove_timer.start();
    r_u[sumcheck_id].at(round - 1) = previous_random;
    const layer &cur = C.circuit[sumcheck_id];
    claim_0 = total[0] ? V_mult[0][0].eval(previous_random) : (~cur.bit_length_u[0]) ? V_mult[0][0].b : F_ZERO;
    claim_1 = total[1] ? V_mult[1][0].eval(previous_random) : (~cur.bit_length_u[1]) ? V_mult[1][0].b : F_ZERO;
    F added[2];
    zkMaskClaims(zkmask::SLOT_U0, previous_random, claim_0, claim_1, added);     // zero-knowledge mode: V + Z M leaves, and phase 2 is about it
    V_u0 = claim_0;
    V_u1 = claim_1;
    prove_timer.stop();
    for (int b = 0; b < 2; ++b) { mult_array[b].clear(); V_mult[b].clear(); }
    proof_size += F_BYTE_SIZE * 2;
}

void prover::sumcheckFinalize2(const F &previous_random, F &claim_0, F &claim_1) {   // prover.cpp:473-485
    prove_timer.start();
    r_v[sumcheck_id].at(round - 1) = previous_random;
    const layer &cur = C.circuit[sumcheck_id];
    claim_0 = total[0] ? V_mult[0][0].eval(previous_random) : (~cur.bit_length_v[0]) ? V_mult[0][0].b : F_ZERO;
    claim_1 = total[1] ? V_mult[1][0].eval(previous_random) : (~cur.bit_length_v[1]) ? V_mult[1][0].b : F_ZERO;
    F added[2];
    zkMaskClaims(zkmask::SLOT_V0, previous_random, claim_0, claim_1, added);
    prove_timer.stop();
    for (int b = 0; b < 2; ++b) { mult_array[b].clear(); V_mult[b].clear(); }
    proof_size += F_BYTE_SIZE * 2;
}

void prover::sumcheckLiuFinalize(const F &previous_random, F &claim_1) {      // prover.cpp:487-497
    prove_timer.start();
    r_u[sumcheck_id].at(round - 1) = previous_random;
    claim_1 = total[1] ? V_mult[1][0].eval(previous_random) : V_mult[1][0].b;
    zkMaskInputClaim(previous_random, claim_1);
    prove_timer.stop();
    proof_size += F_BYTE_SIZE;
    mult_array[1].clear();
    V_mult[1].clear();
    beta_g.clear();
}

hyrax_bls12_381::polyProverBase &prover::commitInput(const vector<G> &gens) { // prover.cpp:503-511
    if (C.circuit[0].size != (1ULL << C.circuit[0].bit_length))
        val[0].resize((size_t) 1 << C.circuit[0].bit_length, F_ZERO);
    zkReset();
    poly_p.reset(new polyProverCPU(val[0], gens));
    return *poly_p;
}

hyrax_bls12_381::polyProverBase &prover::commitInputZk(const vector<G> &gens) {
    if (C.circuit[0].size != (1ULL << C.circuit[0].bit_length))
        val[0].resize((size_t) 1 << C.circuit[0].bit_length, F_ZERO);
    zkReset();
    const vector<F> blinds = zkDrawBlinds((size_t) 1 << (C.circuit[0].bit_length >> 1));
    poly_p.reset(new polyProverCPU(val[0], gens, &blinds));
    return *poly_p;
}

// ---------------------------------------------------------------------------------------------
// CPU Hyrax prover: the protocol documented in zkcnn_amd/csrc/hyrax-bls12-381/polyCommit.hpp
// ---------------------------------------------------------------------------------------------
polyProverCPU::polyProverCPU(const std::vector<Fr> &Z_, const std::vector<G1> &gens, const std::vector<Fr> *blinds) : Z(Z_), ps_bytes(0) {
    pt.start();
    int n = 0;
    while (((size_t) 1 << n) < Z.size()) ++n;
    rb = n >> 1;
    cb = n - rb;
    zkff::batchToAffine(gens, g0);
    const size_t rows = (size_t) 1 << rb, cols = (size_t) 1 << cb;
    comm.resize(rows);
    for (size_t i = 0; i < rows; ++i) comm[i] = zkff::msmCPU(&Z[i * cols], g0.data(), cols);
    if (blinds) {                          // zero-knowledge mode: gens = (g_0 .. g_{cols-1}, H), row i gets + blinds[i] H
        input_blinds = *blinds;
        zk_m = cols;
        const G1 H = G1::fromAffine(g0[cols]);
        for (size_t i = 0; i < rows; ++i) comm[i] = comm[i] + H * (*blinds)[i];
    }
    ps_bytes += rows * 48;
    pt.stop();
}

std::vector<G1> polyProverCPU::commitHostVector(const std::vector<Fr> &v, const std::vector<Fr> &blinds) {
    pt.start();
    const size_t rows = blinds.size(), cols = zk_m;
    std::vector<Fr> padded(rows * cols, Fr(0LL));
    std::copy(v.begin(), v.end(), padded.begin());
    const G1 H = G1::fromAffine(g0[cols]);
    std::vector<G1> out(rows);
    for (size_t i = 0; i < rows; ++i) out[i] = zkff::msmCPU(&padded[i * cols], g0.data(), cols) + H * blinds[i];
    pt.stop();
    return out;
}

std::vector<Fr> polyProverCPU::combineRows(const std::vector<Fr> &x) {
    pt.start();
    const size_t rows = (size_t) 1 << rb, cols = (size_t) 1 << cb;
    std::vector<Fr> Lrow, w(cols, Fr(0LL));
    hyrax_bls12_381::eqTable(Lrow, x.data() + cb, rb, Fr::one());
    for (size_t i = 0; i < rows; ++i) {
        if (Lrow[i].isZero()) continue;
        const Fr *row = &Z[i * cols];
        for (size_t j = 0; j < cols; ++j)
            if (!row[j].isZero()) w[j] = w[j] + Lrow[i] * row[j];
    }
    pt.stop();
    return w;
}

void polyProverCPU::openInit(const std::vector<Fr> &x) {
    pt.start();
    const size_t rows = (size_t) 1 << rb, cols = (size_t) 1 << cb;
    std::vector<Fr> Lrow;
    hyrax_bls12_381::eqTable(Lrow, x.data() + cb, rb, Fr::one());
    hyrax_bls12_381::eqTable(b, x.data(), cb, Fr::one());
    a.assign(cols, Fr(0LL));
    for (size_t i = 0; i < rows; ++i) {
        if (Lrow[i].isZero()) continue;
        const Fr *row = &Z[i * cols];
        for (size_t j = 0; j < cols; ++j)
            if (!row[j].isZero()) a[j] = a[j] + Lrow[i] * row[j];
    }
    g.resize(cols);
    for (size_t j = 0; j < cols; ++j) g[j] = G1::fromAffine(g0[j]);
    pt.stop();
}

hyrax_bls12_381::ipaRoundMsg polyProverCPU::openRound() {
    pt.start();
    const size_t h = a.size() >> 1;
    std::vector<G1Affine> ga;
    zkff::batchToAffine(g, ga);
    hyrax_bls12_381::ipaRoundMsg m;
    m.L = zkff::msmCPU(a.data(), ga.data() + h, h);        // <a_lo, g_hi>
    m.R = zkff::msmCPU(a.data() + h, ga.data(), h);        // <a_hi, g_lo>
    m.yL = Fr(0LL);
    m.yR = Fr(0LL);
    for (size_t j = 0; j < h; ++j) {
        m.yL = m.yL + a[j] * b[j + h];
        m.yR = m.yR + a[j + h] * b[j];
    }
    ps_bytes += 2 * 48 + 2 * 32;
    pt.stop();
    return m;
}

void polyProverCPU::openFold(const Fr &c) {
    pt.start();
    const size_t h = a.size() >> 1;
    for (size_t j = 0; j < h; ++j) {
        a[j] = a[j] + c * a[j + h];
        b[j] = c * b[j] + b[j + h];
        g[j] = g[j] * c + g[j + h];
    }
    a.resize(h);
    b.resize(h);
    g.resize(h);
    pt.stop();
}

std::vector<Fr> polyProverCPU::openFinal() {
    ps_bytes += 32 * a.size();
    return a;
}

} // namespace oracle

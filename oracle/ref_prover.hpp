// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_tables.hpp header; PARITY UNPINNED).
//
// CPU restatement of the reference GKR prover state machine, reference src/prover.cpp:17-511, with
// the reference's own bookkeeping representation (tables of linear polynomials that are evaluated
// at the previous challenge at the start of every round). The product prover
// (zkcnn_amd/csrc/host/prover.cpp -> HIP) keeps plain folded tables in HBM instead; the two must
// return identical field elements from every call under the same challenge stream.
#pragma once
#include <memory>
#include "circuit.h"
#include "polynomial.h"
#include "ref_tables.hpp"
#include "zk_mask.hpp"

namespace oracle {

// CPU Hyrax prover (restates the protocol in hyrax-bls12-381/polyCommit.hpp on host cores)
class polyProverCPU : public hyrax_bls12_381::polyProverBase {
public:
    polyProverCPU(const std::vector<Fr> &Z, const std::vector<G1> &gens, const std::vector<Fr> *blinds = nullptr);
    std::vector<G1> commitHostVector(const std::vector<Fr> &v, const std::vector<Fr> &blinds) override;
    std::vector<Fr> combineRows(const std::vector<Fr> &x) override;
    void addProofBytes(size_t n) override { ps_bytes += n; }
    const std::vector<G1> &commitment() const override { return comm; }
    void openInit(const std::vector<Fr> &x) override;
    hyrax_bls12_381::ipaRoundMsg openRound() override;
    void openFold(const Fr &c) override;
    std::vector<Fr> openFinal() override;
    double getPT() const override { return pt.elapse_sec(); }
    double getPS() const override { return (double) ps_bytes / 1024.0; }
private:
    const std::vector<Fr> &Z;
    std::vector<G1Affine> g0;      // commitment generators (affine)
    std::vector<G1> g;             // folded generators of the running inner-product argument
    std::vector<G1> comm;
    std::vector<Fr> a, b;
    int rb, cb;
    timer pt;
    u64 ps_bytes;
};

class prover : public zkmask::proverMixin {
public:
    void init();                                                              // prover.cpp:17-21
    void sumcheckInitAll(const vector<F>::const_iterator &r_0_from_v);        // :28-36
    void sumcheckInit(const F &alpha_0, const F &beta_0);                     // :43-52
    void sumcheckDotProdInitPhase1();                                         // :57-95
    void sumcheckInitPhase1(const F &relu_rou_0);                             // :155-239
    void sumcheckInitPhase2();                                                // :241-310
    cubic_poly sumcheckDotProdUpdate1(const F &previous_random);              // :103-144
    quadratic_poly sumcheckUpdate1(const F &previous_random);                 // :360-362
    quadratic_poly sumcheckUpdate2(const F &previous_random);                 // :364-366
    F Vres(const vector<F>::const_iterator &r, u32 output_size, u8 r_size);   // :434-457
    void sumcheckDotProdFinalize1(const F &previous_random, F &claim_1);      // :146-153
    void sumcheckFinalize1(const F &previous_random, F &claim_0, F &claim_1); // :459-471
    void sumcheckFinalize2(const F &previous_random, F &claim_0, F &claim_1); // :473-485
    void sumcheckLiuFinalize(const F &previous_random, F &claim_1);           // :487-497
    void sumcheckLiuInit(const vector<F> &s_u, const vector<F> &s_v);         // :312-358
    quadratic_poly sumcheckLiuUpdate(const F &previous_random);               // :385-394
    hyrax_bls12_381::polyProverBase &commitInput(const vector<G> &gens);      // :503-511
    hyrax_bls12_381::polyProverBase &commitInputZk(const vector<G> &gens);    // zero-knowledge mode (no reference counterpart), see zk_mask.hpp

    timer prove_timer;
    double proveTime() const { return prove_timer.elapse_sec(); }
    double proofSize() const { return (double) proof_size / 1024.0; }
    double polyProverTime() const { return poly_p->getPT(); }
    double polyProofSize() const { return poly_p->getPS(); }

    layeredCircuit C;
    vector<vector<F>> val;

private:
    hyrax_bls12_381::polyProverBase &zkBackend() override { return *poly_p; }
    const layeredCircuit &zkCircuit() const override { return C; }
    void zkRawRound(int kind, const F &prev_r, F c[5]) override;
    void zkTailPairs(F A[6]) override;
    quadratic_poly updateEach(const F &previous_random, bool idx);            // :396-426
    quadratic_poly update(const F &previous_random, vector<F> &r_arr);        // :368-383
    F cirValue(u8 layer_id, const vector<u32> &ori, u32 u) const {            // :499-501
        return !layer_id ? val[0][ori[u]] : val[layer_id][u];
    }

    const F *r_0 = nullptr, *r_1 = nullptr;
    vector<vector<F>> r_u, r_v;
    vector<F> beta_g, beta_gs, beta_u;       // beta_gs / beta_u are file-scope statics in the reference (:9)
    F add_term, absorbed_m[2];
    bool absorbed[2] = {false, false};
    int raw_kind = -1;
    vector<linear_poly> mult_array[2], V_mult[2];
    F V_u0, V_u1, alpha, beta, relu_rou;
    u64 proof_size = 0;
    u32 total[2] = {0, 0}, total_size[2] = {0, 0};
    u8 round = 0, sumcheck_id = 0;
    std::unique_ptr<polyProverCPU> poly_p;
};

} // namespace oracle

// hyrax_bls12_381::polyProver -- the commitment prover the reference constructs in
// prover::commitInput (reference src/prover.cpp:509, `polyProver(val[0], gens)`), here backed by the
// HIP multi-scalar-multiplication kernels through include/zkcnn_hip.h. The witness already lives
// in HBM (prover::init uploaded it), so the constructor takes the context instead of the vector.
#pragma once
#include <hyrax-bls12-381/polyCommit.hpp>
#include "../../../include/zkcnn_hip.h"

namespace hyrax_bls12_381 {

class polyProver : public polyProverBase {
public:
    // `affine_cache` (optional, owned by the caller): the affine form of the generators of the previous commitment; re-used when the
    // same generators come again (one batched inversion + 6 products per generator saved), refreshed otherwise
    struct gensCache { std::vector<G1> gens; std::vector<G1Affine> affine; };
    // `blinds` (zero-knowledge mode): gens holds one more generator H and row i is committed as <row_i, g> + (*blinds)[i] H
    polyProver(zk_ctx *ctx, int bit_length, const std::vector<G1> &gens, gensCache *affine_cache = nullptr, const std::vector<Fr> *blinds = nullptr);
    std::vector<G1> commitHostVector(const std::vector<Fr> &v, const std::vector<Fr> &blinds) override;
    std::vector<Fr> combineRows(const std::vector<Fr> &x) override;
    void addProofBytes(size_t n) override { ps_bytes += n; }
    const std::vector<G1> &commitment() const override { return comm; }
    void openInit(const std::vector<Fr> &x) override;
    ipaRoundMsg openRound() override;
    void openFold(const Fr &c) override;
    std::vector<Fr> openFinal() override;
    double getPT() const override { return pt.elapse_sec(); }
    double getPS() const override { return (double) ps_bytes / 1024.0; }

private:
    zk_ctx *ctx;
    std::vector<G1> comm;
    timer pt;
    unsigned long long ps_bytes;
};

} // namespace hyrax_bls12_381

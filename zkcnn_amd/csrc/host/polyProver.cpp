#include "polyProver.hpp"
#include <stdexcept>
#include <string>

namespace hyrax_bls12_381 {

static void must(zk_ctx *ctx, int rc, const char *what) {
    if (rc != ZK_OK) throw std::runtime_error(std::string(what) + ": " + zk_last_error(ctx));
}
static G1 fromABI(const uint64_t *p) {
    G1Affine a;
    std::memcpy(&a, p, 96);
    return G1::fromAffine(a);
}

polyProver::polyProver(zk_ctx *c, int bit_length, const std::vector<G1> &gens, gensCache *cache, const std::vector<Fr> *blinds) : ctx(c), ps_bytes(0) {
    pt.start();
    const int rb = bit_length >> 1;
    const size_t rows = (size_t) 1 << rb;
    if (blinds) {
        if (blinds->size() != rows) throw std::runtime_error("polyProver: one blinding factor per commitment row");
        input_blinds = *blinds;
        zk_m = (size_t) 1 << (bit_length - rb);
    }
    std::vector<G1Affine> local;
    const bool hit = cache && cache->gens.size() == gens.size() && !gens.empty() &&
                     std::memcmp(cache->gens.data(), gens.data(), gens.size() * sizeof(G1)) == 0;
    std::vector<G1Affine> &ga = cache ? cache->affine : local;
    if (!hit) {
        zkff::batchToAffine(gens, ga);
        if (cache) cache->gens = gens;
    }
    std::vector<uint64_t> out(rows * 12);
    if (blinds)
        must(ctx, zk_commit_input_blinded(ctx, reinterpret_cast<const uint64_t *>(ga.data()), ga.size(), reinterpret_cast<const uint64_t *>(blinds->data()),
                                          out.data(), rows), "zk_commit_input_blinded");
    else
        must(ctx, zk_commit_input(ctx, reinterpret_cast<const uint64_t *>(ga.data()), ga.size(), out.data(), rows), "zk_commit_input");
    comm.resize(rows);
    for (size_t i = 0; i < rows; ++i) comm[i] = fromABI(&out[i * 12]);
    ps_bytes += rows * 48;
    pt.stop();
}

std::vector<G1> polyProver::commitHostVector(const std::vector<Fr> &v, const std::vector<Fr> &blinds) {
    pt.start();
    const size_t rows = blinds.size(), cols = zk_m;
    if (!cols || !rows || v.size() > rows * cols) throw std::runtime_error("commitHostVector: shape");
    std::vector<Fr> padded(rows * cols, Fr(0LL));
    std::copy(v.begin(), v.end(), padded.begin());
    std::vector<uint64_t> out(rows * 12);
    must(ctx, zk_commit_vector(ctx, reinterpret_cast<const uint64_t *>(padded.data()), rows, cols, reinterpret_cast<const uint64_t *>(blinds.data()), out.data()),
         "zk_commit_vector");
    std::vector<G1> res(rows);
    for (size_t i = 0; i < rows; ++i) res[i] = fromABI(&out[i * 12]);
    pt.stop();
    return res;
}

std::vector<Fr> polyProver::combineRows(const std::vector<Fr> &x) {
    pt.start();
    std::vector<Fr> w(zk_m);
    must(ctx, zk_hyrax_combine_rows(ctx, reinterpret_cast<const uint64_t *>(x.data()), (uint32_t) x.size(), reinterpret_cast<uint64_t *>(w.data())),
         "zk_hyrax_combine_rows");
    pt.stop();
    return w;
}

void polyProver::openInit(const std::vector<Fr> &x) {
    pt.start();
    must(ctx, zk_hyrax_open_init(ctx, reinterpret_cast<const uint64_t *>(x.data()), (uint32_t) x.size()), "zk_hyrax_open_init");
    pt.stop();
}

ipaRoundMsg polyProver::openRound() {
    pt.start();
    uint64_t L[12], R[12];
    ipaRoundMsg m;
    must(ctx, zk_hyrax_open_round(ctx, L, R, reinterpret_cast<uint64_t *>(&m.yL), reinterpret_cast<uint64_t *>(&m.yR)),
         "zk_hyrax_open_round");
    m.L = fromABI(L);
    m.R = fromABI(R);
    ps_bytes += 2 * 48 + 2 * 32;
    pt.stop();
    return m;
}

void polyProver::openFold(const Fr &c) {
    pt.start();
    must(ctx, zk_hyrax_open_fold(ctx, reinterpret_cast<const uint64_t *>(&c)), "zk_hyrax_open_fold");
    pt.stop();
}

std::vector<Fr> polyProver::openFinal() {
    pt.start();
    std::vector<Fr> a(IPA_STOP_LEN);          // the vector is at most this long (shorter when the verifier asked for more rounds)
    uint32_t n = 0;
    must(ctx, zk_hyrax_open_final(ctx, reinterpret_cast<uint64_t *>(a.data()), (uint32_t) a.size(), &n), "zk_hyrax_open_final");
    a.resize(n);
    ps_bytes += 32ull * n;
    pt.stop();
    return a;
}

} // namespace hyrax_bls12_381

// Commitment generators nobody knows a discrete logarithm of.
//
// The reference draws its Pedersen / Hyrax generators as random multiples k_i * G on the VERIFIER's side of an interactive run
// (reference src/verifier.cpp:119-126): the prover never learns k_i. As soon as the generators must be known to both sides ahead of
// time -- session generators that are re-used across proofs, or a non-interactive (Fiat-Shamir) proof -- k_i * G with a public or
// prover-known k_i makes the commitment non-binding (whoever knows every k_i opens it to anything). Those modes therefore take
//     g_i = h_eff * map(SHA-256("zkcnn-amd/generators/v1" || i || ctr ...))        (try-and-increment on x, then cofactor clearing)
// a nothing-up-my-sleeve sequence that depends on nothing but the index: generator set of size n = the first n elements.
// Host only, computed once per process and size (0.3 s for 4096 points on one core), guarded by a mutex.
#pragma once
#include <map>
#include <mutex>
#include "g1.hpp"
#include "sha256.hpp"

namespace zkff {

// one curve point from (domain, index): x = H(...) mod p over 512 bits, first counter for which x^3 + 4 is a square; the root is
// picked by one more hash bit; multiplication by h_eff = 1 - z = 0xd201000000010001 sends it into the order-r subgroup
inline G1 hashToG1(const char *domain, uint64_t index) {
    static const uint64_t SQRT_EXP[6] = {0xee7fbfffffffeaabULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL,
                                         0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL};      // (p + 1) / 4
    static const uint64_t TWO256[6] = {0, 0, 0, 0, 1, 0};
    const Fp two256 = Fp::fromCanonical(TWO256), four = Fp::fromU64(4);
    for (uint32_t ctr = 0;; ++ctr) {
        uint8_t d[64];
        for (int half = 0; half < 2; ++half) {
            Sha256 h;
            h.update(domain, std::strlen(domain));
            uint8_t tail[13];
            for (int i = 0; i < 8; ++i) tail[i] = (uint8_t) (index >> (8 * i));
            for (int i = 0; i < 4; ++i) tail[8 + i] = (uint8_t) (ctr >> (8 * i));
            tail[12] = (uint8_t) half;
            h.update(tail, sizeof(tail));
            h.digest(d + 32 * half);
        }
        uint64_t lo[6] = {0, 0, 0, 0, 0, 0}, hi[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 32; ++i) {
            lo[i >> 3] |= (uint64_t) d[i] << (8 * (i & 7));
            hi[i >> 3] |= (uint64_t) d[32 + i] << (8 * (i & 7));
        }
        const bool pick_larger = (hi[3] >> 63) != 0;
        hi[3] &= 0x7fffffffffffffffULL;                  // 511 bits: the bias of the reduction below is ~2^-130
        const Fp x = Fp::fromCanonical(hi) * two256 + Fp::fromCanonical(lo);
        const Fp rhs = x * x * x + four;
        Fp y;
        Fp::powLimbs(y, rhs, SQRT_EXP, 6);
        if (!(y * y == rhs)) continue;
        const Fp ny = -y;
        if ((Fp::cmpCanonical(y, ny) > 0) != pick_larger) y = ny;
        G1Affine a;
        a.x = x; a.y = y;
        G1 acc;                                           // h_eff * P, h_eff = 0xd201000000010001
        const uint64_t h_eff = 0xd201000000010001ULL;
        for (int i = 63; i >= 0; --i) {
            G1::dbl(acc, acc);
            if ((h_eff >> i) & 1) G1::addMixed(acc, acc, a);
        }
        if (!acc.isInf()) return acc;
    }
}

struct publicGenerators {
    std::vector<G1> gens;          // Z == 1 (affine) so that every consumer sees the same limbs
    uint8_t digest[32];            // SHA-256 of the 48-byte compressed encodings, in order: what a transcript absorbs
};

// the first `count` generators of the sequence and their digest (cached per count; thread safe)
inline const publicGenerators &publicGeneratorSet(size_t count) {
    static std::mutex mu;
    static std::map<size_t, publicGenerators> cache;
    static std::vector<G1> seq;                           // longest prefix computed so far
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(count);
    if (it != cache.end()) return it->second;
    if (seq.size() < count) {
        std::vector<G1> jac;
        for (size_t i = seq.size(); i < count; ++i) jac.push_back(hashToG1("zkcnn-amd/generators/v1", (uint64_t) i));
        std::vector<G1Affine> aff;
        batchToAffine(jac, aff);
        for (const G1Affine &a : aff) seq.push_back(G1::fromAffine(a));
    }
    publicGenerators &pg = cache[count];
    pg.gens.assign(seq.begin(), seq.begin() + count);
    Sha256 h;
    for (const G1 &g : pg.gens) {
        uint8_t b[48];
        g.serialize(b);
        h.update(b, 48);
    }
    h.digest(pg.digest);
    return pg;
}

}  // namespace zkff

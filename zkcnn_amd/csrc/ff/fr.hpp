// BLS12-381 scalar field Fr and base field Fp for the host side.
//
// `Fr` reproduces the slice of the mcl API the reference uses (SURVEY.md Appendix B;
// call sites e.g. reference src/neuralNetwork.cpp:832,900-916,970-973, src/utils.cpp:56,124,141,228,
// src/verifier.cpp:124-279). mcl itself is not present in /root/reference, so the semantics are
// taken from those call sites:
//   Fr(i64) maps signed integers to x mod r;  isNegative() <=> canonical value > (r-1)/2;
//   getInt64() returns the signed representative;  < and > compare canonical integers;
//   setByCSPRNG() draws from a process-wide stream that tests can seed (parity needs that).
#pragma once
#include <sys/random.h>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include "mont.hpp"
#include <ostream>
#include <random>

namespace zkff {

template <class D>
struct FrParamsT {
    enum { N = 4 };
    static const uint64_t MOD[4], R1[4], R2[4], HALF[4];
    static const uint64_t INV = 0xfffffffeffffffffULL;
};
template <class D> const uint64_t FrParamsT<D>::MOD[4] = {
        0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
template <class D> const uint64_t FrParamsT<D>::R1[4] = {
        0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
template <class D> const uint64_t FrParamsT<D>::R2[4] = {
        0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
// (r - 1) / 2
template <class D> const uint64_t FrParamsT<D>::HALF[4] = {
        0x7fffffff80000000ULL, 0xa9ded2017fff2dffULL, 0x199cec0404d0ec02ULL, 0x39f6d3a994cebea4ULL};
typedef FrParamsT<void> FrParams;

template <class D>
struct FpParamsT {
    enum { N = 6 };
    static const uint64_t MOD[6], R1[6], R2[6];
    static const uint64_t INV = 0x89f3fffcfffcfffdULL;
};
template <class D> const uint64_t FpParamsT<D>::MOD[6] = {
        0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
        0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
template <class D> const uint64_t FpParamsT<D>::R1[6] = {
        0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
        0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
template <class D> const uint64_t FpParamsT<D>::R2[6] = {
        0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
        0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};
typedef FpParamsT<void> FpParams;

typedef MontField<FpParams> Fp;

// xoshiro256** : the seedable challenge / synthetic-data stream
struct Xoshiro {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    void seed(uint64_t x) { for (int i = 0; i < 4; ++i) s[i] = splitmix(x); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 45);
        return res;
    }
    double nextUnit() { return (double) (next() >> 11) * (1.0 / 9007199254740992.0); }
};

// Per-thread replacement of the seeded stream: when set, every challenge is taken from it (Fiat-Shamir: a hash of the
// transcript so far, host/replay.hpp) instead of the xoshiro stream.
struct ChallengeSource {
    virtual ~ChallengeSource() {}
    virtual void words(uint64_t out[4]) = 0;     // 256 fresh pseudo-random bits
    // true: an accepted draw (< r after clearing bit 255) IS the element's Montgomery representation -- no conversion product, which the
    // GPU side of the Fiat-Shamir chain would otherwise pay on its critical path; uniform either way (x -> x R^-1 is a bijection)
    virtual bool rawMontgomery() const { return false; }
};
inline ChallengeSource *&challengeOverride() {
    static thread_local ChallengeSource *src = nullptr;
    return src;
}

// The default source of randomness: the operating system's CSPRNG (getrandom(2), buffered 4 KB at a time per thread).
// The reference draws every challenge with mcl's OS-backed setByCSPRNG (reference src/verifier.cpp:124...279); a seeded xoshiro
// stream is NOT a CSPRNG and is only used when a caller asks for a reproducible run (Fr::seedCSPRNG: parity tests, benches).
struct OsRandom {
    uint8_t buf[4096];
    size_t pos = sizeof(buf);
    void words(uint64_t out[4]) {
        if (pos + 32 > sizeof(buf)) {
            size_t got = 0;
            while (got < sizeof(buf)) {
                ssize_t r = ::getrandom(buf + got, sizeof(buf) - got, 0);
                if (r < 0) {
                    if (errno == EINTR) continue;
                    std::perror("getrandom");
                    std::abort();                      // no randomness, no challenges: never fall back to something predictable
                }
                got += (size_t) r;
            }
            pos = 0;
        }
        std::memcpy(out, buf + pos, 32);
        pos += 32;
    }
};
inline OsRandom &osRandom() {
    static thread_local OsRandom r;
    return r;
}
inline bool &challengeSeeded() {                   // per thread: a seeded (reproducible, NOT secure) stream is active
    static thread_local bool on = false;
    return on;
}

inline Xoshiro &challengeStream() {
    static thread_local Xoshiro g = [] {       // one stream per thread: concurrent sessions (one per host thread) stay deterministic
        Xoshiro x;
        std::random_device rd;
        x.seed(((uint64_t) rd() << 32) ^ rd());
        return x;
    }();
    return g;
}

class Fr : public MontField<FrParams> {
    typedef MontField<FrParams> Base;
public:
    Fr() = default;
    Fr(const Base &b) : Base(b) {}
    // integer constructors: all widths funnel to the signed / unsigned 64-bit paths
    Fr(long long x) : Base(Base::fromI64((int64_t) x)) {}
    Fr(long x) : Base(Base::fromI64((int64_t) x)) {}
    Fr(int x) : Base(Base::fromI64((int64_t) x)) {}
    Fr(unsigned long long x) : Base(Base::fromU64((uint64_t) x)) {}
    Fr(unsigned long x) : Base(Base::fromU64((uint64_t) x)) {}
    Fr(unsigned int x) : Base(Base::fromU64((uint64_t) x)) {}

    static const Fr &one() {
        static const Fr o(Base::one());
        return o;
    }
    static int getByteSize() { return 32; }

    // reproducible challenges for parity tests and benches: NOT cryptographically secure (xoshiro256**, 64-bit seed)
    static void seedCSPRNG(uint64_t seed) { challengeStream().seed(seed); challengeSeeded() = true; }
    // back to the operating system's CSPRNG (the default)
    static void useOsRandom() { challengeSeeded() = false; }
    // uniform in [0, r) by rejection on 255-bit draws; stored as-is in Montgomery form
    // (multiplying a uniform value by the constant R^{-1} keeps it uniform).
    void setByCSPRNG() {
        ChallengeSource *src = challengeOverride();
        for (;;) {
            uint64_t t[4];
            if (src) src->words(t);
            else if (challengeSeeded()) { Xoshiro &g = challengeStream(); for (int i = 0; i < 4; ++i) t[i] = g.next(); }
            else osRandom().words(t);
            t[3] &= 0x7fffffffffffffffULL;
            if (!Base::geMod(t)) {
                if (src && src->rawMontgomery()) std::memcpy(static_cast<void *>(this), t, 32);
                else *this = Fr(Base::fromCanonical(t));
                return;
            }
        }
    }

    // 32 little-endian canonical bytes -> element; false when the value is not below r (non-canonical encoding)
    static bool fromBytesLE(Fr &out, const uint8_t in[32]) {
        uint64_t t[4] = {0, 0, 0, 0};
        for (int i = 0; i < 32; ++i) t[i >> 3] |= (uint64_t) in[i] << (8 * (i & 7));
        if (Base::geMod(t)) return false;
        out = Fr(Base::fromCanonical(t));
        return true;
    }

    bool isNegative() const {
        uint64_t c[4];
        toCanonical(c);
        for (int i = 3; i >= 0; --i) {
            if (c[i] > FrParams::HALF[i]) return true;
            if (c[i] < FrParams::HALF[i]) return false;
        }
        return false;
    }
    // signed representative; only meaningful when |x| < 2^63
    int64_t getInt64() const {
        if (!isNegative()) {
            uint64_t c[4];
            toCanonical(c);
            return (int64_t) c[0];
        }
        Base n;
        Base::neg(n, *this);
        uint64_t c[4];
        n.toCanonical(c);
        return -(int64_t) c[0];
    }

    bool operator<(const Fr &o) const { return Base::cmpCanonical(*this, o) < 0; }
    bool operator>(const Fr &o) const { return Base::cmpCanonical(*this, o) > 0; }

    Fr operator+(const Fr &o) const { Fr z; Base::add(z, *this, o); return z; }
    Fr operator-(const Fr &o) const { Fr z; Base::sub(z, *this, o); return z; }
    Fr operator*(const Fr &o) const { Fr z; Base::mul(z, *this, o); return z; }
    Fr operator-() const { Fr z; Base::neg(z, *this); return z; }

    static void inv(Fr &out, const Fr &x) { Base::invert(out, x); }

    // Tonelli-Shanks with the non-residue 5 (2-adicity of r - 1 is 32). No sign normalisation:
    // which of the two roots comes out is a convention of THIS library (the reference's choice,
    // made inside mcl, cannot be observed here -- SURVEY.md 8(c) "parity unpinned").
    static bool squareRoot(Fr &out, const Fr &x) {
        if (x.isZero()) { out = x; return true; }
        // q = (r - 1) / 2^32
        static const uint64_t Q[4] = {0xfffe5bfeffffffffULL, 0x09a1d80553bda402ULL,
                                      0x299d7d483339d808ULL, 0x0000000073eda753ULL};
        // (q + 1) / 2
        static const uint64_t Q1H[4] = {0x7fff2dff80000000ULL, 0x04d0ec02a9ded201ULL,
                                        0x94cebea4199cec04ULL, 0x0000000039f6d3a9ULL};
        Fr z, t, res, five(5LL);
        Base::powLimbs(z, five, Q, 4);        // generator of the 2^32-torsion
        Base::powLimbs(t, x, Q, 4);           // t = x^q
        Base::powLimbs(res, x, Q1H, 4);       // res = x^((q+1)/2)
        int m = 32;
        while (!(t == one())) {
            int i = 0;
            Fr tt = t;
            while (!(tt == one())) {
                tt = tt * tt;
                if (++i == m) return false;   // x is a non-residue
            }
            Fr b = z;
            for (int k = 0; k < m - i - 1; ++k) b = b * b;
            z = b * b;
            t = t * z;
            res = res * b;
            m = i;
        }
        out = res;
        return true;
    }
};

inline std::ostream &operator<<(std::ostream &os, const Fr &x) { return os << x.toHex(); }

// A prover's PRIVATE coins (blinding factors, masking polynomials of the zero-knowledge mode): the operating system's CSPRNG, or -- for
// reproducible parity runs only -- a seeded stream that is separate from the verifier's challenge stream. One per thread, like the
// challenge stream, so that concurrent sessions stay deterministic under a seed.
struct PrivateCoins {
    Xoshiro g;
    bool seeded = false;
    void seed(uint64_t s) { g.seed(s ^ 0x70726f7665722121ULL); seeded = true; }
    void useOsRandom() { seeded = false; }
    Fr next() {
        for (;;) {
            uint64_t t[4];
            if (seeded) for (int i = 0; i < 4; ++i) t[i] = g.next();
            else osRandom().words(t);
            t[3] &= 0x7fffffffffffffffULL;
            if (!MontField<FrParams>::geMod(t)) return Fr(MontField<FrParams>::fromCanonical(t));
        }
    }
};
inline PrivateCoins &privateCoins() {
    static thread_local PrivateCoins c;
    return c;
}

} // namespace zkff

// Host-side Montgomery prime-field arithmetic (C++14, header only).
//
// This replaces the mcl `Fr` / `Fp` types the reference links against
// (reference: src/global_var.hpp:43 `#define F Fr`; CMakeLists.txt:9 pulls mcl in
// through the hyrax-bls12-381 submodule, which is EMPTY in /root/reference).
// Elements are N little-endian 64-bit limbs in Montgomery form (R = 2^(64N));
// the same in-memory form crosses the C-ABI to the HIP side unchanged.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace zkff {

typedef unsigned __int128 u128;

template <int N>
struct limbs_t {
    uint64_t v[N];
};

// P supplies: N, MOD[N], R1[N] (= R mod p), R2[N] (= R^2 mod p), INV (= -p^{-1} mod 2^64)
template <class P>
class MontField {
public:
    enum { N = P::N };
    uint64_t v[N];

    MontField() = default;

    static MontField zero() {
        MontField z;
        for (int i = 0; i < N; ++i) z.v[i] = 0;
        return z;
    }
    static MontField one() {
        MontField z;
        for (int i = 0; i < N; ++i) z.v[i] = P::R1[i];
        return z;
    }
    // x is a canonical integer < p given as N limbs
    static MontField fromCanonical(const uint64_t *x) {
        MontField a, r2;
        for (int i = 0; i < N; ++i) { a.v[i] = x[i]; r2.v[i] = P::R2[i]; }
        MontField z;
        mul(z, a, r2);
        return z;
    }
    static MontField fromU64(uint64_t x) {
        uint64_t t[N] = {0};
        t[0] = x;
        return fromCanonical(t);
    }
    static MontField fromI64(int64_t x) {
        if (x >= 0) return fromU64((uint64_t) x);
        MontField z = fromU64((uint64_t) (-(x + 1)) + 1ULL);
        MontField o;
        neg(o, z);
        return o;
    }
    void toCanonical(uint64_t *out) const {
        MontField a, z;
        for (int i = 0; i < N; ++i) a.v[i] = (i == 0);
        mul(z, *this, a);
        for (int i = 0; i < N; ++i) out[i] = z.v[i];
    }

    bool isZero() const {
        uint64_t acc = 0;
        for (int i = 0; i < N; ++i) acc |= v[i];
        return acc == 0;
    }
    void clear() { for (int i = 0; i < N; ++i) v[i] = 0; }
    bool operator==(const MontField &o) const {
        uint64_t acc = 0;
        for (int i = 0; i < N; ++i) acc |= v[i] ^ o.v[i];
        return acc == 0;
    }
    bool operator!=(const MontField &o) const { return !(*this == o); }

    static inline bool geMod(const uint64_t *a) {
        for (int i = N - 1; i >= 0; --i) {
            if (a[i] > P::MOD[i]) return true;
            if (a[i] < P::MOD[i]) return false;
        }
        return true;
    }
    static inline void subMod(uint64_t *a) {
        uint64_t borrow = 0;
        for (int i = 0; i < N; ++i) {
            u128 d = (u128) a[i] - P::MOD[i] - borrow;
            a[i] = (uint64_t) d;
            borrow = (uint64_t) (d >> 64) & 1;
        }
    }

    static inline void add(MontField &z, const MontField &x, const MontField &y) {
        uint64_t t[N];
        uint64_t carry = 0;
        for (int i = 0; i < N; ++i) {
            u128 s = (u128) x.v[i] + y.v[i] + carry;
            t[i] = (uint64_t) s;
            carry = (uint64_t) (s >> 64);
        }
        // moduli here leave at least one spare top bit, so carry is always 0
        if (carry || geMod(t)) subMod(t);
        for (int i = 0; i < N; ++i) z.v[i] = t[i];
    }
    static inline void sub(MontField &z, const MontField &x, const MontField &y) {
        uint64_t t[N];
        uint64_t borrow = 0;
        for (int i = 0; i < N; ++i) {
            u128 d = (u128) x.v[i] - y.v[i] - borrow;
            t[i] = (uint64_t) d;
            borrow = (uint64_t) (d >> 64) & 1;
        }
        if (borrow) {
            uint64_t carry = 0;
            for (int i = 0; i < N; ++i) {
                u128 s = (u128) t[i] + P::MOD[i] + carry;
                t[i] = (uint64_t) s;
                carry = (uint64_t) (s >> 64);
            }
        }
        for (int i = 0; i < N; ++i) z.v[i] = t[i];
    }
    static inline void neg(MontField &z, const MontField &x) {
        if (x.isZero()) { z = x; return; }
        uint64_t borrow = 0;
        for (int i = 0; i < N; ++i) {
            u128 d = (u128) P::MOD[i] - x.v[i] - borrow;
            z.v[i] = (uint64_t) d;
            borrow = (uint64_t) (d >> 64) & 1;
        }
    }
    // CIOS Montgomery product
    static inline void mul(MontField &z, const MontField &x, const MontField &y) {
        uint64_t t[N + 2];
        for (int i = 0; i < N + 2; ++i) t[i] = 0;
        for (int i = 0; i < N; ++i) {
            uint64_t c = 0;
            for (int j = 0; j < N; ++j) {
                u128 s = (u128) x.v[j] * y.v[i] + t[j] + c;
                t[j] = (uint64_t) s;
                c = (uint64_t) (s >> 64);
            }
            u128 s = (u128) t[N] + c;
            t[N] = (uint64_t) s;
            t[N + 1] = (uint64_t) (s >> 64);

            uint64_t m = t[0] * P::INV;
            s = (u128) m * P::MOD[0] + t[0];
            c = (uint64_t) (s >> 64);
            for (int j = 1; j < N; ++j) {
                s = (u128) m * P::MOD[j] + t[j] + c;
                t[j - 1] = (uint64_t) s;
                c = (uint64_t) (s >> 64);
            }
            s = (u128) t[N] + c;
            t[N - 1] = (uint64_t) s;
            t[N] = t[N + 1] + (uint64_t) (s >> 64);
        }
        if (t[N] || geMod(t)) subMod(t);
        for (int i = 0; i < N; ++i) z.v[i] = t[i];
    }
    static inline void sqr(MontField &z, const MontField &x) { mul(z, x, x); }

    // z = x^e, e given as `n` little-endian 64-bit limbs
    static void powLimbs(MontField &z, const MontField &x, const uint64_t *e, int n) {
        MontField acc = one(), base = x;
        for (int i = 0; i < n; ++i)
            for (int b = 0; b < 64; ++b) {
                if ((e[i] >> b) & 1) mul(acc, acc, base);
                sqr(base, base);
            }
        z = acc;
    }
    // Fermat inverse; inv(0) = 0
    static void invert(MontField &z, const MontField &x) {
        uint64_t e[N];
        for (int i = 0; i < N; ++i) e[i] = P::MOD[i];
        // e = p - 2 (p is odd and > 2, so no borrow past limb 0 for these moduli)
        uint64_t borrow = 2;
        for (int i = 0; i < N && borrow; ++i) {
            uint64_t old = e[i];
            e[i] = old - borrow;
            borrow = old < borrow ? 1 : 0;
        }
        powLimbs(z, x, e, N);
    }

    MontField operator+(const MontField &o) const { MontField z; add(z, *this, o); return z; }
    MontField operator-(const MontField &o) const { MontField z; sub(z, *this, o); return z; }
    MontField operator*(const MontField &o) const { MontField z; mul(z, *this, o); return z; }
    MontField operator-() const { MontField z; neg(z, *this); return z; }
    MontField &operator+=(const MontField &o) { add(*this, *this, o); return *this; }
    MontField &operator-=(const MontField &o) { sub(*this, *this, o); return *this; }
    MontField &operator*=(const MontField &o) { mul(*this, *this, o); return *this; }

    // canonical little-endian bytes (8N of them)
    void toBytesLE(uint8_t *out) const {
        uint64_t c[N];
        toCanonical(c);
        for (int i = 0; i < N; ++i)
            for (int b = 0; b < 8; ++b) out[i * 8 + b] = (uint8_t) (c[i] >> (8 * b));
    }
    std::string toHex() const {
        uint64_t c[N];
        toCanonical(c);
        static const char *dig = "0123456789abcdef";
        std::string s = "0x";
        for (int i = N - 1; i >= 0; --i)
            for (int b = 60; b >= 0; b -= 4) s.push_back(dig[(c[i] >> b) & 15]);
        return s;
    }
    // compare canonical integers: -1 / 0 / +1
    static int cmpCanonical(const MontField &a, const MontField &b) {
        uint64_t x[N], y[N];
        a.toCanonical(x);
        b.toCanonical(y);
        for (int i = N - 1; i >= 0; --i) {
            if (x[i] < y[i]) return -1;
            if (x[i] > y[i]) return 1;
        }
        return 0;
    }
};

} // namespace zkff

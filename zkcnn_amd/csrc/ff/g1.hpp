// BLS12-381 G1 (y^2 = x^3 + 4 over Fp) for the host side: Jacobian arithmetic, scalar
// multiplication, canonical serialisation and a CPU Pippenger MSM.
//
// Stands in for mcl's `G1` / `mcl::bn::getG1basePoint()` / `G1 * Fr`, which the reference uses at
// src/verifier.cpp:121-126 (random generators = random multiples of the base point) and inside
// the absent hyrax-bls12-381 library (Pedersen row commitments).
#pragma once
#include "fr.hpp"
#include <vector>

namespace zkff {

struct G1Affine {
    Fp x, y;          // (0,0) encodes the point at infinity (it is not on the curve)
    bool isInf() const { return x.isZero() && y.isZero(); }
};

class G1 {
public:
    Fp X, Y, Z;       // Jacobian: (X/Z^2, Y/Z^3); Z == 0 <=> infinity

    G1() { X.clear(); Y = Fp::one(); Z.clear(); }

    static G1 fromAffine(const G1Affine &a) {
        G1 p;
        if (a.isInf()) return p;
        p.X = a.x; p.Y = a.y; p.Z = Fp::one();
        return p;
    }
    bool isInf() const { return Z.isZero(); }
    void clear() { *this = G1(); }

    static const G1 &generator() {
        static const G1 g = [] {
            static const uint64_t gx[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                                           0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
            static const uint64_t gy[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                                           0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
            G1 p;
            p.X = Fp::fromCanonical(gx);
            p.Y = Fp::fromCanonical(gy);
            p.Z = Fp::one();
            return p;
        }();
        return g;
    }

    static void dbl(G1 &r, const G1 &p) {
        if (p.isInf()) { r = p; return; }
        // a = 0 doubling (dbl-2009-l)
        Fp A = p.X * p.X, B = p.Y * p.Y, C = B * B;
        Fp t = p.X + B;
        Fp D = t * t - A - C;
        D = D + D;
        Fp E = A + A + A, F = E * E;
        Fp X3 = F - D - D;
        Fp C8 = C + C; C8 = C8 + C8; C8 = C8 + C8;
        Fp Y3 = E * (D - X3) - C8;
        Fp Z3 = p.Y * p.Z;
        Z3 = Z3 + Z3;
        r.X = X3; r.Y = Y3; r.Z = Z3;
    }
    static void add(G1 &r, const G1 &p, const G1 &q) {
        if (p.isInf()) { r = q; return; }
        if (q.isInf()) { r = p; return; }
        Fp Z1Z1 = p.Z * p.Z, Z2Z2 = q.Z * q.Z;
        Fp U1 = p.X * Z2Z2, U2 = q.X * Z1Z1;
        Fp S1 = p.Y * q.Z * Z2Z2, S2 = q.Y * p.Z * Z1Z1;
        if (U1 == U2) {
            if (S1 == S2) { dbl(r, p); return; }
            r = G1();
            return;
        }
        Fp H = U2 - U1, Rr = S2 - S1;
        Fp HH = H * H, HHH = H * HH, V = U1 * HH;
        Fp X3 = Rr * Rr - HHH - V - V;
        Fp Y3 = Rr * (V - X3) - S1 * HHH;
        Fp Z3 = p.Z * q.Z * H;
        r.X = X3; r.Y = Y3; r.Z = Z3;
    }
    static void addMixed(G1 &r, const G1 &p, const G1Affine &q) {
        if (q.isInf()) { r = p; return; }
        if (p.isInf()) { r = fromAffine(q); return; }
        Fp Z1Z1 = p.Z * p.Z;
        Fp U2 = q.x * Z1Z1, S2 = q.y * p.Z * Z1Z1;
        if (p.X == U2) {
            if (p.Y == S2) { dbl(r, p); return; }
            r = G1();
            return;
        }
        Fp H = U2 - p.X, Rr = S2 - p.Y;
        Fp HH = H * H, HHH = H * HH, V = p.X * HH;
        Fp X3 = Rr * Rr - HHH - V - V;
        Fp Y3 = Rr * (V - X3) - p.Y * HHH;
        Fp Z3 = p.Z * H;
        r.X = X3; r.Y = Y3; r.Z = Z3;
    }
    G1 operator+(const G1 &o) const { G1 r; add(r, *this, o); return r; }
    G1 operator-() const { G1 r = *this; r.Y = -r.Y; return r; }
    G1 operator-(const G1 &o) const { return *this + (-o); }

    // scalar multiplication, 4-bit fixed windows over the canonical scalar
    G1 operator*(const Fr &k) const {
        uint64_t e[4];
        k.toCanonical(e);
        G1 tab[16];
        tab[1] = *this;
        for (int i = 2; i < 16; ++i) add(tab[i], tab[i - 1], *this);
        G1 acc;
        for (int w = 63; w >= 0; --w) {
            for (int d = 0; d < 4; ++d) dbl(acc, acc);
            unsigned dig = (e[w >> 4] >> ((w & 15) * 4)) & 15;
            if (dig) add(acc, acc, tab[dig]);
        }
        return acc;
    }

    G1Affine toAffine() const {
        G1Affine a;
        if (isInf()) { a.x.clear(); a.y.clear(); return a; }
        if (Z == Fp::one()) { a.x = X; a.y = Y; return a; }       // already normalised (points that came in affine)
        Fp zi, zi2;
        Fp::invert(zi, Z);
        zi2 = zi * zi;
        a.x = X * zi2;
        a.y = Y * zi2 * zi;
        return a;
    }
    bool operator==(const G1 &o) const {
        if (isInf() || o.isInf()) return isInf() && o.isInf();
        Fp Z1Z1 = Z * Z, Z2Z2 = o.Z * o.Z;
        if (!(X * Z2Z2 == o.X * Z1Z1)) return false;
        return Y * o.Z * Z2Z2 == o.Y * Z * Z1Z1;
    }
    bool operator!=(const G1 &o) const { return !(*this == o); }

    bool isOnCurve() const {
        if (isInf()) return true;
        G1Affine a = toAffine();
        Fp four = Fp::fromU64(4);
        return a.y * a.y == a.x * a.x * a.x + four;
    }

    // 48-byte compressed form (big-endian x; bit7 = compressed, bit6 = infinity, bit5 = y is the
    // lexicographically larger root) -- the canonical transcript encoding of SURVEY.md 8(b).
    void serialize(uint8_t out[48]) const {
        std::memset(out, 0, 48);
        if (isInf()) { out[0] = 0xc0; return; }
        G1Affine a = toAffine();
        uint64_t cx[6];
        a.x.toCanonical(cx);
        for (int i = 0; i < 6; ++i)
            for (int b = 0; b < 8; ++b) out[47 - (i * 8 + b)] = (uint8_t) (cx[i] >> (8 * b));
        out[0] |= 0x80;
        // y is the larger of the two roots iff y > (p - 1) / 2 (one canonical conversion instead of two)
        static const uint64_t half[6] = {0xdcff7fffffffd555ULL, 0x0f55ffff58a9ffffULL, 0xb39869507b587b12ULL,
                                         0xb23ba5c279c2895fULL, 0x258dd3db21a5d66bULL, 0x0d0088f51cbff34dULL};
        uint64_t cy[6];
        a.y.toCanonical(cy);
        for (int i = 5; i >= 0; --i) {
            if (cy[i] != half[i]) { if (cy[i] > half[i]) out[0] |= 0x20; break; }
        }
    }
    // inverse of serialize(): rejects non-canonical encodings (x >= p, stray flag bits, a non-residue x^3 + 4) and, with
    // check_subgroup, points outside the order-r subgroup (r * P != O).
    static bool deserialize(G1 &out, const uint8_t in[48], bool check_subgroup = true) {
        const uint8_t flags = in[0] & 0xe0;
        if (!(flags & 0x80)) return false;                       // only the compressed form is canonical here
        if (flags & 0x40) {                                      // infinity: everything else must be zero
            if (in[0] != 0xc0) return false;
            for (int i = 1; i < 48; ++i) if (in[i]) return false;
            out = G1();
            return true;
        }
        uint64_t cx[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; ++i)
            for (int b = 0; b < 8; ++b) {
                uint8_t byte = in[47 - (i * 8 + b)];
                if (i == 5 && b == 7) byte &= 0x1f;
                cx[i] |= (uint64_t) byte << (8 * b);
            }
        if (Fp::geMod(cx)) return false;
        const Fp x = Fp::fromCanonical(cx);
        const Fp rhs = x * x * x + Fp::fromU64(4);
        // p = 3 (mod 4): sqrt = rhs^((p + 1) / 4)
        static const uint64_t e[6] = {0xee7fbfffffffeaabULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL,
                                      0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL};
        Fp y;
        Fp::powLimbs(y, rhs, e, 6);
        if (!(y * y == rhs)) return false;
        Fp ny = -y;
        const bool y_is_larger = Fp::cmpCanonical(y, ny) > 0;
        if (y_is_larger != ((flags & 0x20) != 0)) y = ny;
        G1Affine a;
        a.x = x; a.y = y;
        out = fromAffine(a);
        if (check_subgroup) {
            static const uint64_t rr[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
            G1 acc;                                              // r * P by double-and-add over the 255 bits of r
            for (int i = 254; i >= 0; --i) {
                dbl(acc, acc);
                if ((rr[i >> 6] >> (i & 63)) & 1) addMixed(acc, acc, a);
            }
            if (!acc.isInf()) return false;
        }
        return true;
    }
};

// many Jacobian -> affine conversions with one field inversion
inline void batchToAffine(const std::vector<G1> &in, std::vector<G1Affine> &out) {
    size_t n = in.size();
    out.resize(n);
    std::vector<Fp> pref(n);
    Fp acc = Fp::one();
    for (size_t i = 0; i < n; ++i) {
        pref[i] = acc;
        if (!in[i].isInf()) acc = acc * in[i].Z;
    }
    Fp inv;
    Fp::invert(inv, acc);
    for (size_t i = n; i-- > 0;) {
        if (in[i].isInf()) { out[i].x.clear(); out[i].y.clear(); continue; }
        Fp zi = inv * pref[i];
        inv = inv * in[i].Z;
        Fp zi2 = zi * zi;
        out[i].x = in[i].X * zi2;
        out[i].y = in[i].Y * zi2 * zi;
    }
}

// CPU Pippenger: sum_i k[i] * base[i]. Scalars above (r-1)/2 are folded to their negative so that
// small signed witnesses (quantised weights, bits) only populate the low windows.
inline G1 msmCPU(const Fr *k, const G1Affine *base, size_t n) {
    const int C = 8, NB = 1 << C;
    std::vector<uint64_t> mag(n * 4);
    std::vector<uint8_t> negf(n);
    int top = 0;                                  // highest non-zero byte index + 1
    for (size_t i = 0; i < n; ++i) {
        Fr s = k[i];
        negf[i] = s.isNegative();
        if (negf[i]) s = -s;
        s.toCanonical(&mag[i * 4]);
        for (int b = 31; b >= top; --b)
            if ((mag[i * 4 + (b >> 3)] >> ((b & 7) * 8)) & 0xff) { top = b + 1; break; }
    }
    G1 total;
    std::vector<G1> bucket(NB);
    for (int w = top - 1; w >= 0; --w) {
        for (int d = 0; d < C; ++d) G1::dbl(total, total);
        for (int b = 0; b < NB; ++b) bucket[b] = G1();
        bool any = false;
        for (size_t i = 0; i < n; ++i) {
            unsigned dig = (mag[i * 4 + (w >> 3)] >> ((w & 7) * 8)) & 0xff;
            if (!dig) continue;
            any = true;
            G1Affine q = base[i];
            if (negf[i]) q.y = -q.y;
            G1::addMixed(bucket[dig], bucket[dig], q);
        }
        if (!any) continue;
        G1 run, sum;
        for (int b = NB - 1; b >= 1; --b) {
            G1::add(run, run, bucket[b]);
            G1::add(sum, sum, run);
        }
        G1::add(total, total, sum);
    }
    return total;
}

} // namespace zkff

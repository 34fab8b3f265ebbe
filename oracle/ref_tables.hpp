// ORACLE -- TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is linked, imported or executed by the
// product path (zkcnn_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// PARITY UNPINNED: the reference ships no tests / golden vectors for this path and its arithmetic
// dependency (hyrax-bls12-381 + mcl) is an empty submodule, so the reference cannot be built or
// run here (DESIGN.md "Oracle"). This file restates the reference's table-building algorithms with
// the SAME recursions the reference uses (the product side deliberately uses different, closed-form
// or parallel formulations, so agreement between the two is a real check):
//   halfTables / eqTable2 / eqTable1   <- reference src/utils.cpp:32-51, 147-165, 168-180
//   phiTable                          <- reference src/utils.cpp:53-103
//   nttInPlace                        <- reference src/utils.cpp:105-145
//   rootOfUnity                       <- reference src/utils.cpp:224-232
#pragma once
#include <vector>
#include <hyrax-bls12-381/polyCommit.hpp>

namespace oracle {

typedef Fr F;

// reference src/utils.cpp:224-232: n-1 successive square roots starting from -1
inline F rootOfUnity(int n) {
    F res = -Fr::one();
    if (!n) return Fr::one();
    while (--n) {
        bool ok = Fr::squareRoot(res, res);
        (void) ok;
    }
    return res;
}

// reference src/utils.cpp:32-51: eq tables over the low `first_half` and the remaining variables
inline void halfTables(std::vector<F> &lo, std::vector<F> &hi, const F *r, const F &init, unsigned first_half,
                       unsigned second_half) {
    lo[0] = init;
    hi[0] = Fr::one();
    for (unsigned i = 0; i < first_half; ++i)
        for (size_t j = 0; j < ((size_t) 1 << i); ++j) {
            F t = lo[j] * r[i];
            lo[j | ((size_t) 1 << i)] = t;
            lo[j] = lo[j] - t;
        }
    for (unsigned i = 0; i < second_half; ++i)
        for (size_t j = 0; j < ((size_t) 1 << i); ++j) {
            F t = hi[j] * r[i + first_half];
            hi[j | ((size_t) 1 << i)] = t;
            hi[j] = hi[j] - t;
        }
}

// reference src/utils.cpp:147-165: out[i] = alpha eq(r0, i) + beta eq(r1, i), with the zero short-cuts
inline void eqTable2(std::vector<F> &out, int n, const F *r0, const F *r1, const F &alpha, const F &beta) {
    unsigned fh = (unsigned) n >> 1, sh = (unsigned) n - fh;
    size_t mask = ((size_t) 1 << fh) - 1, len = (size_t) 1 << n;
    std::vector<F> lo((size_t) 1 << fh), hi((size_t) 1 << sh);
    if (!beta.isZero()) {
        halfTables(lo, hi, r1, beta, fh, sh);
        for (size_t i = 0; i < len; ++i) out[i] = lo[i & mask] * hi[i >> fh];
    } else for (size_t i = 0; i < len; ++i) out[i].clear();
    if (alpha.isZero()) return;
    halfTables(lo, hi, r0, alpha, fh, sh);
    for (size_t i = 0; i < len; ++i) out[i] = out[i] + lo[i & mask] * hi[i >> fh];
}

// reference src/utils.cpp:168-180: out[i] = init eq(r, i)
inline void eqTable1(std::vector<F> &out, int n, const F *r, const F &init) {
    if (n < 0) return;
    unsigned fh = (unsigned) n >> 1, sh = (unsigned) n - fh;
    size_t mask = ((size_t) 1 << fh) - 1, len = (size_t) 1 << n;
    if (init.isZero()) { for (size_t i = 0; i < len; ++i) out[i].clear(); return; }
    std::vector<F> lo((size_t) 1 << fh), hi((size_t) 1 << sh);
    halfTables(lo, hi, r, init, fh, sh);
    for (size_t i = 0; i < len; ++i) out[i] = lo[i & mask] * hi[i >> fh];
}

// reference src/utils.cpp:53-103: MLE (in the row index) of the DFT matrix at rx, butterfly recursion
inline void phiTable(std::vector<F> &phi, const F *rx, const F &scale, int n, bool inverse) {
    size_t N = (size_t) 1 << n;
    std::vector<F> pw(N);
    F w = rootOfUnity(n);
    if (inverse) Fr::inv(w, w);
    pw[0] = Fr::one();
    for (size_t i = 1; i < N; ++i) pw[i] = pw[i - 1] * w;

    if (inverse) {
        phi[0] = phi[1] = scale;
        for (int i = 2; i <= n; ++i)
            for (size_t b = 0; b < ((size_t) 1 << (i - 1)); ++b) {
                size_t l = b, r = b ^ ((size_t) 1 << (i - 1));
                int mm = n - i;
                F t1 = Fr::one() - rx[mm], t2 = rx[mm] * pw[b << mm];
                phi[r] = phi[l] * (t1 - t2);
                phi[l] = phi[l] * (t1 + t2);
            }
    } else {
        phi[0] = scale;
        for (int i = 1; i < n; ++i)
            for (size_t b = 0; b < ((size_t) 1 << (i - 1)); ++b) {
                size_t l = b, r = b ^ ((size_t) 1 << (i - 1));
                int mm = n - i;
                F t1 = Fr::one() - rx[mm], t2 = rx[mm] * pw[b << mm];
                phi[r] = phi[l] * (t1 - t2);
                phi[l] = phi[l] * (t1 + t2);
            }
        for (size_t b = 0; b < ((size_t) 1 << (n - 1)); ++b)
            phi[b] = phi[b] * (Fr::one() - rx[0] + rx[0] * pw[b]);
    }
}

// reference src/utils.cpp:105-145: bit-reversal + decimation-in-time butterflies, w[k] = omega^k
inline void nttInPlace(std::vector<F> &a, int logn, bool inverse) {
    size_t len = (size_t) 1 << logn;
    std::vector<size_t> rev(len);
    std::vector<F> w(len);
    rev[0] = 0;
    for (size_t i = 1; i < len; ++i) rev[i] = (rev[i >> 1] >> 1) | ((i & 1) << (logn - 1));
    w[0] = Fr::one();
    if (len > 1) {
        w[1] = rootOfUnity(logn);
        if (inverse) Fr::inv(w[1], w[1]);
    }
    for (size_t i = 2; i < len; ++i) w[i] = w[i - 1] * w[1];
    for (size_t i = 0; i < len; ++i) if (rev[i] < i) std::swap(a[i], a[rev[i]]);
    for (size_t i = 2; i <= len; i <<= 1)
        for (size_t j = 0; j < len; j += i)
            for (size_t k = 0; k < (i >> 1); ++k) {
                F u = a[j + k], v = a[j + k + (i >> 1)] * w[len / i * k];
                a[j + k] = u + v;
                a[j + k + (i >> 1)] = u - v;
            }
    if (inverse) {
        F ilen;
        Fr::inv(ilen, F((unsigned long long) len));
        for (size_t i = 0; i < len; ++i) a[i] = a[i] * ilen;
    }
}

} // namespace oracle

#include "utils.hpp"
#include <map>

thread_local vector<string> output_tb(OUT_COLUMN_CNT, "");

// Bit length of the next power of two >= n, -1 for n == 0. The reference computes this in double
// precision (reference src/utils.cpp:23-25), which over-counts by one at n = 2^29 and 2^31; the same
// expression is used so table sizes agree for every n.
char ceilPow2BitLength(u32 n) {
    if (n == 0) return -1;
    return (char) std::ceil(std::log((double) n) / std::log(2.));
}
char floorPow2BitLength(u32 n) {
    if (n == 0) return -1;
    return (char) std::floor(std::log((double) n) / std::log(2.));
}

// eq(r, .) by successive doubling: after step i the first 2^(i+1) slots hold the table over i+1 vars
static void eqAccumulate(vector<F> &out, u8 n, const vector<F>::const_iterator &r, const F &init, bool add_into) {
    const size_t len = (size_t) 1 << n;
    vector<F> t(len);
    t[0] = init;
    for (u32 i = 0; i < n; ++i) {
        const size_t half = (size_t) 1 << i;
        for (size_t j = 0; j < half; ++j) {
            F hi = t[j] * r[i];
            t[j | half] = hi;
            t[j] = t[j] - hi;
        }
    }
    if (add_into) for (size_t i = 0; i < len; ++i) out[i] = out[i] + t[i];
    else for (size_t i = 0; i < len; ++i) out[i] = t[i];
}

void initBetaTable(vector<F> &beta_g, u8 gLength, const vector<F>::const_iterator &r_0,
                   const vector<F>::const_iterator &r_1, const F &alpha, const F &beta) {
    const size_t len = (size_t) 1 << gLength;
    if (beta.isZero()) for (size_t i = 0; i < len; ++i) beta_g[i].clear();
    else eqAccumulate(beta_g, gLength, r_1, beta, false);
    if (!alpha.isZero()) eqAccumulate(beta_g, gLength, r_0, alpha, true);
}

void initBetaTable(vector<F> &beta_g, u8 gLength, const vector<F>::const_iterator &r, const F &init) {
    if (gLength == (u8) -1) return;
    const size_t len = (size_t) 1 << gLength;
    if (init.isZero()) for (size_t i = 0; i < len; ++i) beta_g[i].clear();
    else eqAccumulate(beta_g, gLength, r, init, false);
}

// primitive 2^n-th root of unity: n-1 successive square roots of -1 (reference src/utils.cpp:224-232).
// Which root each squareRoot call returns is this library's convention (ff/fr.hpp).
F getRootOfUnit(int n) {
    static thread_local std::map<int, F> cache;
    auto it = cache.find(n);
    if (it != cache.end()) return it->second;
    F res = F_ONE;
    if (n > 0) {
        res = -F_ONE;
        for (int k = 1; k < n; ++k) {
            bool ok = F::squareRoot(res, res);
            assert(ok);
            (void) ok;
        }
    }
    cache[n] = res;
    return res;
}

static void rootPowers(vector<F> &pw, int n, bool inverse) {
    F w = getRootOfUnit(n);
    if (inverse) F::inv(w, w);
    pw.resize((size_t) 1 << n);
    pw[0] = F_ONE;
    for (size_t i = 1; i < pw.size(); ++i) pw[i] = pw[i - 1] * w;
}

// Closed form of the table the reference builds recursively (reference src/utils.cpp:61-103):
//   forward: u < 2^(n-1), phi[u] = scale * prod_{j<n}   (1 - rx_j + rx_j * w^{ u 2^j})
//   inverse: u < 2^n,     phi[u] = scale * prod_{j<n-1} (1 - rx_j + rx_j * w^{-u 2^j})
void phiGInit(vector<F> &phi_g, const vector<F>::const_iterator &rx, const F &scale, int n, bool isIFFT) {
    vector<F> pw;
    rootPowers(pw, n, isIFFT);
    const size_t N = (size_t) 1 << n, cnt = isIFFT ? N : N >> 1;
    const int vars = isIFFT ? n - 1 : n;
    for (size_t u = 0; u < cnt; ++u) {
        F acc = scale;
        for (int j = 0; j < vars; ++j) {
            size_t e = (u << j) & (N - 1);
            acc = acc * (F_ONE - rx[j] + rx[j] * pw[e]);
        }
        phi_g[u] = acc;
    }
}

void fft(vector<F> &arr, int logn, bool flag) {
    const size_t len = (size_t) 1 << logn;
    assert(arr.size() == len);
    static thread_local std::map<int, vector<F>> tw_cache;       // key: logn * 2 + inverse
    vector<F> &tw = tw_cache[logn * 2 + (flag ? 1 : 0)];
    if (tw.empty()) rootPowers(tw, logn, flag);

    for (size_t i = 1, j = 0; i < len; ++i) {       // bit-reversal permutation
        size_t bit = len >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(arr[i], arr[j]);
    }
    for (size_t span = 1; span < len; span <<= 1) {
        const size_t step = len / (span << 1);
        for (size_t base = 0; base < len; base += span << 1)
            for (size_t k = 0; k < span; ++k) {
                F lo = arr[base + k], hi = arr[base + k + span] * tw[step * k];
                arr[base + k] = lo + hi;
                arr[base + k + span] = lo - hi;
            }
    }
    if (flag) {
        F ilen;
        F::inv(ilen, F((unsigned long long) len));
        for (size_t i = 0; i < len; ++i) arr[i] = arr[i] * ilen;
    }
}

void initLayer(layer &circuit, long size, layerType ty) {
    circuit.size = circuit.zero_start_id = (u32) size;
    circuit.bit_length = ceilPow2BitLength((u32) size);
    circuit.ty = ty;
}

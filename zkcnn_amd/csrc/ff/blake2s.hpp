// BLAKE2s-256 (RFC 7693), written from the RFC: the hash of the Fiat-Shamir challenge chain (host/replay.hpp: fiatShamir).
// One implementation for both sides of the prover: g++ compiles it for the host verifier, hipcc for the tail-round kernel that derives
// its own challenges on the GPU (hip/fs_tail.cuh). Why not SHA-256 there: a challenge costs one lane ~3 compressions of 64 strictly
// sequential rounds each (~2200 dependent-ish instructions per compression); BLAKE2s needs 2 compressions of 10 rounds whose 4 column /
// diagonal G functions are independent, ~3x fewer instructions per compression -- on ONE GPU lane that is the difference between a
// hash that costs more than the sumcheck round it serves and one that does not.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zkff {

struct Blake2s {
    uint32_t h[8];
    uint32_t t[2];
    uint8_t buf[64];
    uint32_t fill;

    ZK_HD static uint32_t iv(int i) {
        return i == 0 ? 0x6A09E667u : i == 1 ? 0xBB67AE85u : i == 2 ? 0x3C6EF372u : i == 3 ? 0xA54FF53Au : i == 4 ? 0x510E527Fu : i == 5 ? 0x9B05688Cu
             : i == 6 ? 0x1F83D9ABu : 0x5BE0CD19u;
    }
    ZK_HD static uint32_t rotr(uint32_t x, int k) { return (x >> k) | (x << (32 - k)); }

    ZK_HD void init() {
        for (int i = 0; i < 8; ++i) h[i] = iv(i);
        h[0] ^= 0x01010020u;               // digest length 32, no key, fanout = depth = 1
        t[0] = t[1] = 0;
        fill = 0;
    }

    // one compression of the 16 message words m (already little-endian words), counter and finalisation flag applied by the caller's state
    ZK_HD static void compress(uint32_t h[8], const uint32_t m[16], uint32_t t0, uint32_t t1, bool last) {
        uint32_t v[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = iv(i); }
        v[12] ^= t0;
        v[13] ^= t1;
        if (last) v[14] = ~v[14];
#define ZK_B2S_G(a, b, c, d, x, y)                                              \
        v[a] = v[a] + v[b] + (x); v[d] = rotr(v[d] ^ v[a], 16);                 \
        v[c] = v[c] + v[d];       v[b] = rotr(v[b] ^ v[c], 12);                 \
        v[a] = v[a] + v[b] + (y); v[d] = rotr(v[d] ^ v[a], 8);                  \
        v[c] = v[c] + v[d];       v[b] = rotr(v[b] ^ v[c], 7);
#define ZK_B2S_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)   \
        ZK_B2S_G(0, 4, 8, 12, m[s0], m[s1]) ZK_B2S_G(1, 5, 9, 13, m[s2], m[s3])                \
        ZK_B2S_G(2, 6, 10, 14, m[s4], m[s5]) ZK_B2S_G(3, 7, 11, 15, m[s6], m[s7])              \
        ZK_B2S_G(0, 5, 10, 15, m[s8], m[s9]) ZK_B2S_G(1, 6, 11, 12, m[s10], m[s11])            \
        ZK_B2S_G(2, 7, 8, 13, m[s12], m[s13]) ZK_B2S_G(3, 4, 9, 14, m[s14], m[s15])
        ZK_B2S_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
        ZK_B2S_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
        ZK_B2S_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
        ZK_B2S_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
        ZK_B2S_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
        ZK_B2S_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
        ZK_B2S_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
        ZK_B2S_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
        ZK_B2S_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
        ZK_B2S_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
#undef ZK_B2S_ROUND
#undef ZK_B2S_G
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
    }

    ZK_HD void block(bool last) {
        uint32_t m[16];
        for (int i = 0; i < 16; ++i)
            m[i] = (uint32_t) buf[4 * i] | ((uint32_t) buf[4 * i + 1] << 8) | ((uint32_t) buf[4 * i + 2] << 16) | ((uint32_t) buf[4 * i + 3] << 24);
        compress(h, m, t[0], t[1], last);
    }
    ZK_HD void update(const void *data, size_t n) {
        const uint8_t *p = static_cast<const uint8_t *>(data);
        while (n) {
            if (fill == 64) {                       // a full buffer is only compressed when more input follows: the last block carries the flag
                t[0] += 64;
                if (t[0] < 64) ++t[1];
                block(false);
                fill = 0;
            }
            size_t take = 64 - fill < n ? 64 - fill : n;
            for (size_t i = 0; i < take; ++i) buf[fill + i] = p[i];
            fill += (uint32_t) take; p += take; n -= take;
        }
    }
    ZK_HD void final(uint8_t out[32]) {
        t[0] += fill;
        if (t[0] < fill) ++t[1];
        for (uint32_t i = fill; i < 64; ++i) buf[i] = 0;
        block(true);
        for (int i = 0; i < 8; ++i)
            for (int b = 0; b < 4; ++b) out[4 * i + b] = (uint8_t) (h[i] >> (8 * b));
    }
};

// The Fiat-Shamir chain step on WORDS (what the GPU kernel uses: no byte buffers): state' = BLAKE2s-256(state (32 B) || msg), msg of
// n_words 32-bit little-endian words, (8 + n_words) * 4 <= 128 bytes, i.e. at most two compressions.
ZK_HD void blake2s_chain_words(uint32_t state[8], const uint32_t *msg, int n_words) {
    uint32_t h[8], m[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = Blake2s::iv(i);
    h[0] ^= 0x01010020u;
    const int total = 8 + n_words;                  // words of input
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = i < 8 ? state[i] : (i - 8 < n_words ? msg[i - 8] : 0u);
    if (total <= 16) {
        Blake2s::compress(h, m, (uint32_t) total * 4, 0, true);
    } else {
        Blake2s::compress(h, m, 64, 0, false);
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = (16 + i - 8 < n_words) ? msg[16 + i - 8] : 0u;
        Blake2s::compress(h, m, (uint32_t) total * 4, 0, true);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) state[i] = h[i];
}

}  // namespace zkff

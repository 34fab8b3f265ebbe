// The small rounds of a sumcheck phase in ONE persistent single-workgroup kernel (reference src/prover.cpp:368-426, one call per round there).
//
// A round on a small table is a launch, a hand-over to the host and a host turn-around: ~25 us, of which the arithmetic is two dependent
// field products. Once the live tables of a phase hold at most TAIL_QUADS quads, k_tail runs ALL remaining rounds of the phase -- fold,
// round sums, the O(1) add_term bookkeeping of reference src/prover.cpp:360-426 -- and only the challenge differs by mode:
//   * LIVE (interactive protocol, round 3): the verifier still draws every challenge after seeing the round polynomial. The kernel posts the
//     polynomial in a mapped host mailbox and polls a second mailbox for the challenge; the host side of a round is "read 128 bytes, call
//     the verifier, write 48 bytes". No launch, no dispatch, no stream event per round. Every 16-byte chunk of either mailbox carries the
//     round's sequence number, and a chunk is written / read with ONE 16-byte access, so a message is complete when all its chunks show the
//     expected number: no ordering between stores is assumed and nobody waits for a write acknowledgement across PCIe.
//     The kernel leaves by itself after the phase's last round; if the host stops talking (a verifier that rejected mid-phase) it is told to
//     (abort word), and as a last resort it gives up after TAIL_TIMEOUT_TICKS of the constant 100 MHz clock.
//   * FS: non-interactive mode, the next challenge is a hash of the transcript, which the kernel computes itself:
//     chain step state' = BLAKE2s-256(state || a || b || c) on the 96 bytes as they lie in memory (Montgomery limbs: a bijection of the field
//     elements), challenge = state' with bit 255 cleared, taken AS Montgomery limbs if below r, else one more step with an empty message
//     (host twin: host/replay.hpp fiatShamir). All round polynomials, challenges and final table entries go to the host at once; the host
//     prover answers the verifier's calls from that record and checks that the verifier's challenges are the ones derived here.
#pragma once
#include "kernels.cuh"
#include "tail_types.hpp"
#include "../ff/blake2s.hpp"

struct tail_args {
    const fr_t *Vin[2], *Min[2];      // current tables (V may still be a layer's values: first round)
    fr_t *Vbuf[2][2], *Mbuf[2][2];    // ping-pong buffers; out_idx[b] = the one the next fold writes
    int32_t out_idx[2];
    uint64_t n[2];                    // pre-fold length of each pair; 0 = absent
    int32_t first, rounds, with_add_term, pad_;
    fr_t prev_r, add_term;
    uint32_t fs_state[8];
    tail_out *out;
    unsigned long long seq;           // FS: published when everything is written
    const live_in *in;                // LIVE: challenge mailbox
    uint32_t seq32;                   // LIVE: round k is posted with seq32 + k; the challenge that follows it is expected with seq32 + k + 1
    uint32_t export_tables;           // LIVE: `rounds` ends before the phase does: the last round's folded tables (<= TAIL_EXPORT_MAX entries each) go to out->exp_*
};

__device__ __forceinline__ bool fr_raw_ge_mod(const uint32_t t[8]) {
    const uint32_t m[8] = FR_MOD_INIT;
    for (int i = 7; i >= 0; --i) {
        if (t[i] > m[i]) return true;
        if (t[i] < m[i]) return false;
    }
    return true;
}

// state' = BLAKE2s(state || a || b || c) (128 bytes, two compressions); the challenge is the state with bit 255 cleared, taken as Montgomery
// limbs if below r, else one more chain step with an empty message (host twin: host/replay.hpp fiatShamir)
__device__ __forceinline__ fr_t fs_round_challenge(uint32_t st[8], const fr_t &ca, const fr_t &cb, const fr_t &cc) {
    uint32_t msg[24];
#pragma unroll
    for (int i = 0; i < 8; ++i) { msg[i] = ca.v[i]; msg[8 + i] = cb.v[i]; msg[16 + i] = cc.v[i]; }
    zkff::blake2s_chain_words(st, msg, 24);
    for (;;) {
        uint32_t cand[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) cand[i] = st[i];
        cand[7] &= 0x7fffffffu;
        if (!fr_raw_ge_mod(cand)) {
            fr_t ch;
#pragma unroll
            for (int i = 0; i < 8; ++i) ch.v[i] = cand[i];
            return ch;
        }
        zkff::blake2s_chain_words(st, st, 0);              // rejected (probability ~9%): one more step, empty message
    }
}

// ---- the same chain step on FOUR lanes (a quad): BLAKE2s' column step is four independent G functions and so is its diagonal step. Lane c of the
// quad owns column c of the 4 x 4 state (v[c], v[c+4], v[c+8], v[c+12]); between the two steps rows 1..3 rotate by 1..3 lanes (DPP quad
// permutes, one full-rate instruction each). Inputs and outputs are REPLICATED on the four lanes; ~46 instructions per round instead of 112 on
// one lane, so a challenge costs ~1.5 us instead of ~4.5 (host twin and test vector: ff/blake2s.hpp, tests/test_wire_cpu.py via the transcripts). ----
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t x) { return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t quad_sel(uint32_t c, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) { return c == 0 ? w0 : c == 1 ? w1 : c == 2 ? w2 : w3; }
__device__ __forceinline__ void blake2s_compress4(uint32_t h[8], const uint32_t m[16], uint32_t t0, bool last, uint32_t c) {
    uint32_t va = quad_sel(c, h[0], h[1], h[2], h[3]), vb = quad_sel(c, h[4], h[5], h[6], h[7]);
    uint32_t vc = quad_sel(c, zkff::Blake2s::iv(0), zkff::Blake2s::iv(1), zkff::Blake2s::iv(2), zkff::Blake2s::iv(3));
    uint32_t vd = quad_sel(c, zkff::Blake2s::iv(4) ^ t0, zkff::Blake2s::iv(5), last ? ~zkff::Blake2s::iv(6) : zkff::Blake2s::iv(6), zkff::Blake2s::iv(7));
#define ZK_G4(x, y)                                                                       \
    va = va + vb + (x); vd = zkff::Blake2s::rotr(vd ^ va, 16);                            \
    vc = vc + vd;       vb = zkff::Blake2s::rotr(vb ^ vc, 12);                            \
    va = va + vb + (y); vd = zkff::Blake2s::rotr(vd ^ va, 8);                             \
    vc = vc + vd;       vb = zkff::Blake2s::rotr(vb ^ vc, 7);
#define ZK_R4(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)       \
    ZK_G4(quad_sel(c, m[s0], m[s2], m[s4], m[s6]), quad_sel(c, m[s1], m[s3], m[s5], m[s7]))             \
    vb = quad_perm<0x39>(vb); vc = quad_perm<0x4E>(vc); vd = quad_perm<0x93>(vd);                         \
    ZK_G4(quad_sel(c, m[s8], m[s10], m[s12], m[s14]), quad_sel(c, m[s9], m[s11], m[s13], m[s15]))       \
    vb = quad_perm<0x93>(vb); vc = quad_perm<0x4E>(vc); vd = quad_perm<0x39>(vd);
    ZK_R4(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    ZK_R4(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    ZK_R4(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    ZK_R4(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    ZK_R4(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    ZK_R4(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    ZK_R4(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    ZK_R4(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    ZK_R4(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    ZK_R4(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
#undef ZK_R4
#undef ZK_G4
    const uint32_t x0 = quad_sel(c, h[0], h[1], h[2], h[3]) ^ va ^ vc, x1 = quad_sel(c, h[4], h[5], h[6], h[7]) ^ vb ^ vd;
    h[0] = quad_perm<0x00>(x0); h[1] = quad_perm<0x55>(x0); h[2] = quad_perm<0xAA>(x0); h[3] = quad_perm<0xFF>(x0);
    h[4] = quad_perm<0x00>(x1); h[5] = quad_perm<0x55>(x1); h[6] = quad_perm<0xAA>(x1); h[7] = quad_perm<0xFF>(x1);
}
// state' = BLAKE2s-256(state || msg) on a quad (blake2s_chain_words of ff/blake2s.hpp)
__device__ __forceinline__ void blake2s_chain_words4(uint32_t state[8], const uint32_t *msg, int n_words, uint32_t c) {
    uint32_t h[8], m[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = zkff::Blake2s::iv(i);
    h[0] ^= 0x01010020u;
    const int total = 8 + n_words;
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = i < 8 ? state[i] : (i - 8 < n_words ? msg[i - 8] : 0u);
    if (total <= 16) {
        blake2s_compress4(h, m, (uint32_t) total * 4, true, c);
    } else {
        blake2s_compress4(h, m, 64, false, c);
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = (16 + i - 8 < n_words) ? msg[16 + i - 8] : 0u;
        blake2s_compress4(h, m, (uint32_t) total * 4, true, c);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) state[i] = h[i];
}
// fs_round_challenge on a quad: every lane of the quad passes the same st / coefficients and gets the same challenge
__device__ __forceinline__ fr_t fs_round_challenge4(uint32_t st[8], const fr_t &ca, const fr_t &cb, const fr_t &cc, uint32_t c) {
    uint32_t msg[24];
#pragma unroll
    for (int i = 0; i < 8; ++i) { msg[i] = ca.v[i]; msg[8 + i] = cb.v[i]; msg[16 + i] = cc.v[i]; }
    blake2s_chain_words4(st, msg, 24, c);
    for (;;) {
        uint32_t cand[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) cand[i] = st[i];
        cand[7] &= 0x7fffffffu;
        if (!fr_raw_ge_mod(cand)) {
            fr_t ch;
#pragma unroll
            for (int i = 0; i < 8; ++i) ch.v[i] = cand[i];
            return ch;
        }
        blake2s_chain_words4(st, st, 0, c);                // rejected (probability ~9%): one more step, empty message
    }
}

typedef uint32_t zk_u32x4 __attribute__((ext_vector_type(4)));
// one 16-byte access that goes to the host every time (system scope: sc0 sc1 on gfx94x / gfx950)
__device__ __forceinline__ void store16_sys(void *p, zk_u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void load48_sys(const void *p, zk_u32x4 &c0, zk_u32x4 &c1, zk_u32x4 &c2) {
    asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\t"
                 "global_load_dwordx4 %1, %3, off offset:16 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %3, off offset:32 sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(c0), "=&v"(c1), "=&v"(c2) : "v"(p) : "memory");
}

// Work layout of a round: item i of the list [quads of pair 0 | quads of pair 1 | a pair that collapses this round, one item each] belongs
// to lanes 4 (i mod TAIL_SLOTS) .. + 3, in passes of TAIL_SLOTS items. Four lanes per quad (the layout of k_round_quad_fine: a lone wave
// issues its products one after the other, so the chain per lane is what counts):
//     fold     role 0: v0   1: v1   2: m0   3: m1                                 (one product)
//     product  role 0: c = v0 m0   1: p(1) = v1 m1   2: a = (v1 - v0)(m1 - m0)      (one more)
// A collapsing pair (the reference's `total == 1` case, prover.cpp:400-404) is a degenerate quad: role 0 / 2 fold its last V / M pair and
// role 0's product is the term add_term takes over. add_term (1 - r) is the product of the one lane that is idle in the product step.
template <bool LIVE>
__device__ __forceinline__ void tail_body(const tail_args &a) {
    __shared__ fr_t s_part[TAIL_THREADS / 64][3];
    __shared__ fr_t s_r, s_fin[2], s_prod[2], s_tail[2][2], s_add, s_addm;
    __shared__ uint32_t s_state[8];
    __shared__ int s_stop;
    // the folded tables live in LDS from the kernel's first fold on ([0]: V, [1]: M; pair 0 at the front, pair 1 behind it): a round's operands are
    // ~100 ns away instead of a trip through L2. Ping-pong: a fold's outputs are half its inputs (<= 2 * TAIL_QUADS entries after the first fold).
    __shared__ fr_t s_buf0[2][2 * TAIL_QUADS], s_buf1[2][TAIL_QUADS];
    int folds = 0;
    unsigned long long t_wait = 0;
    const unsigned long long t_begin = LIVE ? wall_clock64() : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t n[2] = {a.n[0], a.n[1]};
    const fr_t *Vin[2] = {a.Vin[0], a.Vin[1]}, *Min[2] = {a.Min[0], a.Min[1]};
    bool first = a.first != 0;
    uint32_t pstate[2] = {n[0] ? 1u : 0u, n[1] ? 1u : 0u};
    if (tid == 0) {
        s_r = a.prev_r;
        s_add = a.add_term;
        s_stop = 0;
        if (!LIVE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s_state[i] = a.fs_state[i];
        }
    }
    __syncthreads();
    for (int k = 0; k < a.rounds; ++k) {
        const fr_t r = s_r;
        bool collapse[2];
        uint32_t quads[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            collapse[b] = n[b] && (first ? n[b] == 1 : n[b] == 2);
            quads[b] = (!n[b] || collapse[b]) ? 0u : (uint32_t) (first ? n[b] / 2 : n[b] / 4);
        }
        const uint32_t role = (uint32_t) tid & 3, nq = quads[0] + quads[1];
        const uint32_t total_items = nq + (collapse[0] ? 1u : 0u) + (collapse[1] ? 1u : 0u);
        fr_t prod = fr_zero();
        bool worked = false;
        for (uint32_t base = 0; base == 0 || base < total_items; base += TAIL_SLOTS) {
            // a wave without items sits the pass out (wave-uniform): the SIMDs then serve only the waves that hold quads -- in the last rounds of
            // a phase that is wave 0 alone, and a product costs its 1.1 us instead of four times that
            if (base + 16u * (uint32_t) wave >= total_items && !(base == 0 && wave == 0)) continue;
            worked = true;
            const uint32_t item = base + ((uint32_t) tid >> 2);
            const bool live = item < total_items;
            const bool special = live && item >= nq;
            // which pair: quads of pair 0, then of pair 1, then the collapsing pairs in pair order
            int b = 0;
            uint32_t q = item;
            if (special) b = (collapse[0] && item == nq) ? 0 : 1;
            else if (live && item >= quads[0]) { b = 1; q = item - quads[0]; }
            const uint32_t quads_b = b ? quads[1] : quads[0];
            const fr_t *Vb = b ? Vin[1] : Vin[0], *Mb = b ? Min[1] : Min[0];
            fr_t *wV = (folds & 1) ? s_buf1[0] : s_buf0[0], *wM = (folds & 1) ? s_buf1[1] : s_buf0[1];
            const uint32_t wbase = b ? 2 * quads[0] : 0;               // where this pair's outputs start in the LDS buffers
            fr_t X = fr_zero(), opA = fr_zero(), opB = fr_zero();
            if (special) {
                if (role == 0 || role == 2) {
                    const fr_t *src = role == 0 ? Vb : Mb;
                    X = fr_load(src);
                    if (!first) X = fr_lerp(X, fr_load(src + 1), r);
                }
            } else if (live && first) {
                if (role < 3) {
                    const fr_t v0 = fr_load(Vb + 2 * q), v1 = fr_load(Vb + 2 * q + 1), m0 = fr_load(Mb + 2 * q), m1 = fr_load(Mb + 2 * q + 1);
                    opA = role == 0 ? v0 : role == 1 ? v1 : fr_sub(v1, v0);
                    opB = role == 0 ? m0 : role == 1 ? m1 : fr_sub(m1, m0);
                    if (quads_b == 1 && role < 2) s_tail[b][role] = opA;
                }
            } else if (live) {
                const fr_t *src = (role < 2 ? Vb : Mb) + 4 * q + 2 * (role & 1);
                X = fr_lerp(fr_load(src), fr_load(src + 1), r);
                (role < 2 ? wV : wM)[wbase + 2 * q + (role & 1)] = X;
                if (quads_b == 1 && role < 2) s_tail[b][role] = X;           // the pair the phase may end with
                if (LIVE && a.export_tables && k == a.rounds - 1 && 2 * q + 1 < TAIL_EXPORT_MAX)      // a shortened tail: the host goes on from these
                    fr_store_scoped((role < 2 ? a.out->exp_V[b] : a.out->exp_M[b]) + 2 * q + (role & 1), X, true);
            }
            if (!first || special) {
                fr_t y1, y2, y3;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    y1.v[i] = (uint32_t) __shfl_xor((int) X.v[i], 2, 64);
                    y2.v[i] = (uint32_t) __shfl_xor((int) X.v[i], 1, 64);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) y3.v[i] = (uint32_t) __shfl_xor((int) y1.v[i], 1, 64);
                if (special) {
                    // role 0: X = V's last value, y1 = M's (role 2's X): their product goes to add_term
                    opA = role == 0 ? X : fr_zero();
                    opB = role == 0 ? y1 : fr_zero();
                    if (role == 0) s_fin[b] = X;
                } else if (!first) {
                    // role 0: X = v0, y1 = m0;  role 1: X = v1, y1 = m1;  role 2: X = m0, y1 = v0, y2 = m1, y3 = v1
                    opA = role == 2 ? fr_sub(y3, y1) : X;
                    opB = role == 2 ? fr_sub(y2, X) : y1;
                    if (role == 3 || !live) { opA = fr_zero(); opB = fr_zero(); }
                }
            }
            // the bookkeeping product add_term (1 - r) rides on a lane that is idle in this step (reference src/prover.cpp:375)
            const bool addm = base == 0 && tid == 3 && a.with_add_term;
            if (addm) { opA = s_add; opB = fr_sub(fr_one(), r); }
            fr_t x = fr_mul(opA, opB);
            if (addm) { s_addm = x; x = fr_zero(); }
            if (special && role == 0) { s_prod[b] = x; x = fr_zero(); }
            prod = fr_add(prod, x);
        }
        // reduction: lanes of equal role by butterflies; lanes 0..2 of every wave leave the wave's sums
        if (worked) {
#pragma unroll
            for (int off = 4; off < 64; off <<= 1) {
                fr_t o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o.v[i] = (uint32_t) __shfl_xor((int) prod.v[i], off, 64);
                prod = fr_add(prod, o);
            }
        }
        if (lane < 3) s_part[wave][lane == 2 ? 0 : lane + 1] = prod;       // accumulator order a, c, p(1); zero from a wave that sat out
        if (LIVE && a.export_tables && k == a.rounds - 1) ZK_WAIT_STORES();  // (the exported entries are written before the barrier below, the last message after it)
        __syncthreads();
        if (wave == 0) {
            // lane 16 t + w holds accumulator t of wave w; a 4-step butterfly over w
            const int t = lane >> 4, w = lane & 15;
            fr_t tot = t < 3 ? s_part[w][t] : fr_zero();
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                fr_t o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o.v[i] = (uint32_t) __shfl_xor((int) tot.v[i], off, 64);
                tot = fr_add(tot, o);
            }
            fr_t ca, cc, p1;                                   // every lane of wave 0 gets the three sums (lanes 0, 16, 32 hold them)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                ca.v[i] = (uint32_t) __shfl((int) tot.v[i], 0, 64);
                cc.v[i] = (uint32_t) __shfl((int) tot.v[i], 16, 64);
                p1.v[i] = (uint32_t) __shfl((int) tot.v[i], 32, 64);
            }
            if (tid < (LIVE ? 1 : 4)) {                          // (non-interactive: the quad of lanes 0..3 shares the hash, all four carry the same values)
                // host bookkeeping of quad_round (sumcheck.hip), reference src/prover.cpp:368-383
                fr_t add_term = a.with_add_term ? s_addm : s_add;
                fr_t cb = fr_sub(fr_sub(p1, ca), cc);
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    if (collapse[b]) add_term = fr_add(add_term, s_prod[b]);
                if (a.with_add_term) { cb = fr_sub(cb, add_term); cc = fr_add(cc, add_term); }
                const bool last = k == a.rounds - 1;
                if (!LIVE) {
                    // chain step on the 96 bytes of (a, b, c) as they lie in memory, then the challenge
                    uint32_t st[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) st[i] = s_state[i];
                    const fr_t ch = fs_round_challenge4(st, ca, cb, cc, (uint32_t) tid);
                    if (tid == 0) {
                        s_add = add_term;
                        fr_store(&a.out->poly[k][0], ca);
                        fr_store(&a.out->poly[k][1], cb);
                        fr_store(&a.out->poly[k][2], cc);
                        s_r = ch;
                        fr_store(&a.out->chal[k], ch);
#pragma unroll
                        for (int i = 0; i < 8; ++i) s_state[i] = st[i];
                    }
                }
                if (LIVE) {
                    s_add = add_term;
                    tail_out *o = a.out;
                    if (last) {
                        // what the host needs after the phase: posted (and acknowledged) BEFORE the last round's mailbox message
                        fr_store_scoped(&o->add_term, add_term, true);
                        __hip_atomic_store(&o->ticks_wait, t_wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(&o->ticks_total, wall_clock64() - t_begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            uint32_t ps = pstate[b];
                            if (collapse[b]) ps = 2;
                            else if (a.export_tables && ps == 1 && !first) {
                                ps = 3;                                            // n[b] / 2 entries of V and of M were exported by this round's fold
                                __hip_atomic_store(&o->exp_n[b], (uint32_t) (n[b] >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                            __hip_atomic_store(&o->pair_state[b], ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            if (ps == 1) { fr_store_scoped(&o->tail_v[b][0], s_tail[b][0], true); fr_store_scoped(&o->tail_v[b][1], s_tail[b][1], true); }
                            if (ps == 2) fr_store_scoped(&o->final_v[b], s_fin[b], true);
                        }
                        ZK_WAIT_STORES();
                    }
                    uint32_t w24[24];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { w24[i] = ca.v[i]; w24[8 + i] = cb.v[i]; w24[16 + i] = cc.v[i]; }
                    const uint32_t sq = a.seq32 + (uint32_t) k;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        zk_u32x4 ch = {w24[3 * j], w24[3 * j + 1], w24[3 * j + 2], sq};
                        store16_sys(&o->live.c[j][0], ch);
                    }
                    if (!last) {
                        // the verifier's challenge for this polynomial
                        const uint32_t want = sq + 1;
                        const unsigned long long t0 = wall_clock64();
                        for (;;) {
                            zk_u32x4 c0, c1, c2;
                            load48_sys(a.in, c0, c1, c2);
                            if (c0.w == want && c1.w == want && c2.w == want) {
                                fr_t ch;
                                ch.v[0] = c0.x; ch.v[1] = c0.y; ch.v[2] = c0.z; ch.v[3] = c1.x; ch.v[4] = c1.y; ch.v[5] = c1.z; ch.v[6] = c2.x; ch.v[7] = c2.y;
                                s_r = ch;
                                break;
                            }
                            if (c0.w == TAIL_ABORT) { s_stop = 1; break; }
                            if (wall_clock64() - t0 > TAIL_TIMEOUT_TICKS) { s_stop = 2; break; }
                        }
                        t_wait += wall_clock64() - t0;
                    }
                }
            }
        }
        __syncthreads();
        if (LIVE && s_stop) {
            if (tid == 0) __hip_atomic_store(&a.out->status, (uint32_t) s_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (!n[b]) continue;
            if (collapse[b]) { n[b] = 0; pstate[b] = 2; continue; }
            if (!first) {
                const uint32_t wbase = b ? 2 * quads[0] : 0;
                Vin[b] = ((folds & 1) ? s_buf1[0] : s_buf0[0]) + wbase;
                Min[b] = ((folds & 1) ? s_buf1[1] : s_buf0[1]) + wbase;
                n[b] >>= 1;
            }
        }
        if (!first) ++folds;
        first = false;
    }
    if (!LIVE && tid == 0) {
        tail_out *o = a.out;
        fr_store(&o->add_term, s_add);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            o->pair_state[b] = pstate[b];
            if (pstate[b] == 1) { fr_store(&o->tail_v[b][0], s_tail[b][0]); fr_store(&o->tail_v[b][1], s_tail[b][1]); }
            if (pstate[b] == 2) fr_store(&o->final_v[b], s_fin[b]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) o->fs_state[i] = s_state[i];
        __threadfence_system();
        *((volatile unsigned long long *) &o->seq) = a.seq;
    }
}
template <bool LIVE>
__global__ void __launch_bounds__(TAIL_THREADS) k_tail(tail_args a) { tail_body<LIVE>(a); }
// the interactive form as a functor (launch.cuh): the lanes of a lock-step batch run their tails as ONE launch, a workgroup per lane, each with its own
// mailboxes -- inside the launch the lanes are independent (a lane's rounds go at its verifier's pace), the batch is in lock step again when all have left
struct k_tail_live_f {
    tail_args a;
    __device__ __forceinline__ void operator()() const { tail_body<true>(a); }
};

// ------------------------------------------------------------------------------------------------
// Resident MULTI-workgroup rounds of the interactive protocol: the middle of a phase, tables of 2^11 .. 2^16 entries (too large for the one
// workgroup of k_tail, too small to amortise a launch: ~28 us per round as launches, of which ~10 are the kernel). One launch runs a SEGMENT
// of consecutive rounds in which no table collapses or reaches its last pair (the host plans segments; such rounds stay launches).
// Work layout of k_round_quad_fine (4 lanes per quad, 64 quads per block); a round needs ceil(quads / 64) blocks, and a block whose quads
// are gone leaves for good (tables only shrink). Per round:
//   every block   waits for the challenge (broadcast line in device memory), folds its quads, block sums -> partials, arrival ticket
//   last arriver  ("leader" of the round) adds the partials, finishes the polynomial (add_term: every block tracks it, no hand-over), posts it
//                 in the host mailbox, polls the host mailbox for the verifier's challenge and broadcasts it
// Tables written in one round are read by OTHER workgroups (other XCDs) in the next: every table access, the partials and the broadcast line
// are 16-byte sc1 (agent-scope) accesses -- the documented alternative to release / acquire fences, which would write back and invalidate
// whole L2s once per block and round. Workgroups wait for one another, so all of them must be resident: at most 256 blocks of 256 threads,
// and the resident kernels are only used while at most GPU_MAX_HW_QUEUES proofs are active (zk_proof_begin).
// ------------------------------------------------------------------------------------------------
struct mid_args {
    const fr_t *Vin[2], *Min[2];
    fr_t *Vbuf[2][2], *Mbuf[2][2];
    int32_t out_idx[2];
    uint64_t n[2];                    // pre-fold length of each pair; 0 = absent; a present pair has >= 8 entries in every round of the segment
    int32_t rounds, with_add_term;
    int32_t first, pad0_;             // the segment starts with the phase's first round (pairs, nothing is folded)
    fr_t prev_r, add_term;            // challenge of the segment's first fold; add_term BEFORE that round's (1 - r) factor
    fr_t *partials;                   // 3 per block
    uint32_t *arrive;                 // zero at launch; cumulative arrivals
    mid_bcast *bc;
    tail_out *out;
    const live_in *in;
    uint32_t seq32, pad_;
    uint32_t fs_state[8];             // Fiat-Shamir (k_mid<false>): chain state at the segment's start
    unsigned long long seq;           // Fiat-Shamir: published in out->seq when the segment is over
};

__device__ __forceinline__ fr_t fr_load_sc1(const fr_t *p) {
    zk_u32x4 lo, hi;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(p) : "memory");
    fr_t z;
    z.v[0] = lo.x; z.v[1] = lo.y; z.v[2] = lo.z; z.v[3] = lo.w; z.v[4] = hi.x; z.v[5] = hi.y; z.v[6] = hi.z; z.v[7] = hi.w;
    return z;
}
// two consecutive elements (64 bytes): four loads in flight, one wait
__device__ __forceinline__ void fr_load2_sc1(const fr_t *p, fr_t &x, fr_t &y) {
    zk_u32x4 a0, a1, b0, b1;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(p) : "memory");
    x.v[0] = a0.x; x.v[1] = a0.y; x.v[2] = a0.z; x.v[3] = a0.w; x.v[4] = a1.x; x.v[5] = a1.y; x.v[6] = a1.z; x.v[7] = a1.w;
    y.v[0] = b0.x; y.v[1] = b0.y; y.v[2] = b0.z; y.v[3] = b0.w; y.v[4] = b1.x; y.v[5] = b1.y; y.v[6] = b1.z; y.v[7] = b1.w;
}
__device__ __forceinline__ void fr_store_sc1(fr_t *p, const fr_t &a) {
    const zk_u32x4 lo = {a.v[0], a.v[1], a.v[2], a.v[3]}, hi = {a.v[4], a.v[5], a.v[6], a.v[7]};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" :: "v"(p), "v"(lo), "v"(hi) : "memory");
}
__device__ __forceinline__ void load48_sc1(const void *p, zk_u32x4 &c0, zk_u32x4 &c1, zk_u32x4 &c2) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
                 "global_load_dwordx4 %1, %3, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %3, off offset:32 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(c0), "=&v"(c1), "=&v"(c2) : "v"(p) : "memory");
}
__device__ __forceinline__ void store16_sc1(void *p, zk_u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

// LIVE: the verifier's challenges through the host mailboxes (interactive protocol). !LIVE: the non-interactive mode -- the round's leader continues
// the BLAKE2s chain itself (state handed from leader to leader through the broadcast line) and the host reads all polynomials afterwards.
template <bool LIVE>
__global__ void __launch_bounds__(ZK_BLOCK) k_mid(mid_args a) {
    __shared__ fr_t s_role[3][ZK_BLOCK / 64];
    __shared__ fr_t s_r, s_add, s_tot[3];
    __shared__ int s_flag;            // 0 go on, 1 this block is the round's leader, 2 stop (abort / time-out)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t lb = blockIdx.x, role = (uint32_t) tid & 3;
    uint64_t n[2] = {a.n[0], a.n[1]};
    const fr_t *Vin[2] = {a.Vin[0], a.Vin[1]}, *Min[2] = {a.Min[0], a.Min[1]};
    int oi[2] = {a.out_idx[0], a.out_idx[1]};
    uint32_t arrived_before = 0;      // arrivals of all earlier rounds (every block computes the same numbers)
    if (tid == 0) { s_r = a.prev_r; s_add = a.add_term; s_flag = 0; }
    __syncthreads();
    bool first = a.first != 0;
    for (int k = 0; k < a.rounds; ++k) {
        // chunks of 64 quads; block lb takes chunks lb, lb + G, ... (G = gridDim.x <= MID_MAX_BLOCKS: the kernel's footprint is bounded whatever the
        // table size, because its blocks wait for one another and must all be resident)
        const uint32_t q0 = (uint32_t) (first ? n[0] / 2 : n[0] / 4), q1 = (uint32_t) (first ? n[1] / 2 : n[1] / 4), total = q0 + q1;
        const uint32_t nchunks = (total + 63) / 64, nblk = min(nchunks, gridDim.x);
        if (lb >= nchunks) return;                                     // this block's chunks are gone (uniform: every later round is smaller)
        const fr_t r = s_r;
        // add_term (1 - r): every block tracks the scalar (one product on a wave that idles otherwise), so whoever leads the round has it
        if (tid == 64 && a.with_add_term) s_add = fr_mul(s_add, fr_sub(fr_one(), r));
        fr_t prod = fr_zero();
        for (uint32_t chunk = lb; chunk < nchunks; chunk += gridDim.x) {
            const uint32_t item = 64u * chunk + ((uint32_t) tid >> 2);
            const bool live = item < total;
            const int b = (live && item >= q0) ? 1 : 0;
            const uint32_t q = b ? item - q0 : item;
            fr_t opA = fr_zero(), opB = fr_zero();
            if (first) {
                // the phase's first round: pairs as they are (reference src/prover.cpp:396-426 with nothing to fold yet)
                if (live && role < 3) {
                    fr_t v0, v1, m0, m1;
                    fr_load2_sc1((b ? Vin[1] : Vin[0]) + 2 * (size_t) q, v0, v1);
                    fr_load2_sc1((b ? Min[1] : Min[0]) + 2 * (size_t) q, m0, m1);
                    opA = role == 0 ? v0 : role == 1 ? v1 : fr_sub(v1, v0);
                    opB = role == 0 ? m0 : role == 1 ? m1 : fr_sub(m1, m0);
                }
            } else {
                fr_t X = fr_zero();
                if (live) {
                    const fr_t *src = (role < 2 ? (b ? Vin[1] : Vin[0]) : (b ? Min[1] : Min[0])) + 4 * (size_t) q + 2 * (role & 1);
                    fr_t e0, e1;
                    fr_load2_sc1(src, e0, e1);
                    X = fr_lerp(e0, e1, r);
                    fr_t *dst = role < 2 ? (b ? a.Vbuf[1][oi[1]] : a.Vbuf[0][oi[0]]) : (b ? a.Mbuf[1][oi[1]] : a.Mbuf[0][oi[0]]);
                    fr_store_sc1(dst + 2 * (size_t) q + (role & 1), X);
                }
                fr_t y1, y2, y3;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    y1.v[i] = (uint32_t) __shfl_xor((int) X.v[i], 2, 64);
                    y2.v[i] = (uint32_t) __shfl_xor((int) X.v[i], 1, 64);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) y3.v[i] = (uint32_t) __shfl_xor((int) y1.v[i], 1, 64);
                // role 0: X = v0, y1 = m0;  role 1: X = v1, y1 = m1;  role 2: X = m0, y1 = v0, y2 = m1, y3 = v1
                opA = role == 2 ? fr_sub(y3, y1) : X;
                opB = role == 2 ? fr_sub(y2, X) : y1;
                if (role == 3 || !live) { opA = fr_zero(); opB = fr_zero(); }
            }
            prod = fr_add(prod, fr_mul(opA, opB));
        }
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) {
            fr_t o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] = (uint32_t) __shfl_xor((int) prod.v[i], off, 64);
            prod = fr_add(prod, o);
        }
        if (lane < 3) s_role[lane][wave] = prod;          // role 0: c, role 1: p(1), role 2: a
        __syncthreads();
        if (tid < 3) {
            fr_t t = s_role[tid][0];
#pragma unroll
            for (int w = 1; w < ZK_BLOCK / 64; ++w) t = fr_add(t, s_role[tid][w]);
            fr_store_sc1(a.partials + (size_t) 3 * lb + tid, t);
        }
        ZK_WAIT_STORES();                                  // ONE wait per round: this thread's table stores and partial sums are out before the block reports in
        __syncthreads();
        if (tid == 0) {
            const uint32_t t = __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_flag = (t == arrived_before + nblk - 1) ? 1 : 0;
        }
        __syncthreads();
        const bool last_round = k == a.rounds - 1;
        const uint32_t sq = a.seq32 + (uint32_t) k;
        if (s_flag == 1) {
            // ---- leader of the round: grid sums, the polynomial, the host, the next challenge ----
            if (wave < 3) {
                fr_t tot = fr_zero();
                for (uint32_t blk = lane; blk < nblk; blk += 64) tot = fr_add(tot, fr_load_sc1(a.partials + (size_t) 3 * blk + wave));
                tot = fr_wave_sum(tot);
                if (lane == 0) s_tot[wave] = tot;
            }
            __syncthreads();
            if (tid < (LIVE ? 1 : 4)) {                          // (non-interactive: the quad of lanes 0..3 shares the hash)
                fr_t cc = s_tot[0], p1 = s_tot[1], ca = s_tot[2];
                fr_t cb = fr_sub(fr_sub(p1, ca), cc);
                if (a.with_add_term) { cb = fr_sub(cb, s_add); cc = fr_add(cc, s_add); }
                if (last_round && tid == 0) __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (everybody has arrived: the counter is free for the next launch)
                if (!LIVE) {
                    // non-interactive: record the polynomial, continue the chain, broadcast the challenge derived from it
                    if (tid == 0) {
                        fr_store(&a.out->poly[k][0], ca);
                        fr_store(&a.out->poly[k][1], cb);
                        fr_store(&a.out->poly[k][2], cc);
                    }
                    uint32_t st[8];
                    if (k == 0) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) st[i] = a.fs_state[i];
                    } else {
                        zk_u32x4 s0, s1, s2;
                        load48_sc1(&a.bc->st[0], s0, s1, s2);
                        st[0] = s0.x; st[1] = s0.y; st[2] = s0.z; st[3] = s0.w; st[4] = s1.x; st[5] = s1.y; st[6] = s1.z; st[7] = s1.w;
                    }
                    const fr_t ch = fs_round_challenge4(st, ca, cb, cc, (uint32_t) tid);
                    if (tid == 0) fr_store(&a.out->chal[k], ch);
                    ZK_WAIT_STORES();
                    if (tid != 0) {
                        // (lanes 1..3 only helped with the hash)
                    } else if (last_round) {
                        __threadfence_system();
                        *((volatile unsigned long long *) &a.out->seq) = a.seq;
                    } else {
                        const zk_u32x4 s0 = {st[0], st[1], st[2], st[3]}, s1 = {st[4], st[5], st[6], st[7]};
                        store16_sc1(&a.bc->st[0], s0);
                        store16_sc1(&a.bc->st[4], s1);
                        ZK_WAIT_STORES();
                        s_r = ch;
                        const uint32_t want = sq + 1;
                        const zk_u32x4 c0 = {ch.v[0], ch.v[1], ch.v[2], want}, c1 = {ch.v[3], ch.v[4], ch.v[5], want}, c2 = {ch.v[6], ch.v[7], 0u, want};
                        store16_sc1(&a.bc->c[0][0], c0);
                        store16_sc1(&a.bc->c[1][0], c1);
                        store16_sc1(&a.bc->c[2][0], c2);
                    }
                }
                uint32_t w24[24];
#pragma unroll
                for (int i = 0; i < 8; ++i) { w24[i] = ca.v[i]; w24[8 + i] = cb.v[i]; w24[16 + i] = cc.v[i]; }
                if (LIVE) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        zk_u32x4 ch = {w24[3 * j], w24[3 * j + 1], w24[3 * j + 2], sq};
                        store16_sys(&a.out->live.c[j][0], ch);
                    }
                }
                if (LIVE && !last_round) {
                    const uint32_t want = sq + 1;
                    const unsigned long long t0 = wall_clock64();
                    zk_u32x4 c0, c1, c2;
                    int stop = 0;
                    for (;;) {
                        load48_sys(a.in, c0, c1, c2);
                        if (c0.w == want && c1.w == want && c2.w == want) break;
                        if (c0.w == TAIL_ABORT) { stop = 1; break; }
                        if (wall_clock64() - t0 > TAIL_TIMEOUT_TICKS) { stop = 2; break; }
                    }
                    if (stop) {
                        c0.w = c1.w = c2.w = TAIL_ABORT;
                        __hip_atomic_store(&a.out->status, ((uint32_t) stop << 8) | (uint32_t) k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_flag = 2;
                    } else {
                        fr_t ch;
                        ch.v[0] = c0.x; ch.v[1] = c0.y; ch.v[2] = c0.z; ch.v[3] = c1.x; ch.v[4] = c1.y; ch.v[5] = c1.z; ch.v[6] = c2.x; ch.v[7] = c2.y;
                        s_r = ch;
                    }
                    // the broadcast line: the same three chunks
                    store16_sc1(&a.bc->c[0][0], c0);
                    store16_sc1(&a.bc->c[1][0], c1);
                    store16_sc1(&a.bc->c[2][0], c2);
                }
            }
        } else if (!last_round) {
            if (tid == 0) {
                const uint32_t want = sq + 1;
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    zk_u32x4 c0, c1, c2;
                    load48_sc1(a.bc, c0, c1, c2);
                    if (c0.w == want && c1.w == want && c2.w == want) {
                        fr_t ch;
                        ch.v[0] = c0.x; ch.v[1] = c0.y; ch.v[2] = c0.z; ch.v[3] = c1.x; ch.v[4] = c1.y; ch.v[5] = c1.z; ch.v[6] = c2.x; ch.v[7] = c2.y;
                        s_r = ch;
                        break;
                    }
                    if (c0.w == TAIL_ABORT) { s_flag = 2; break; }
                    if (wall_clock64() - t0 > 2 * TAIL_TIMEOUT_TICKS) {
                        s_flag = 2;
                        __hip_atomic_store(&a.out->status, 0x300u | (uint32_t) k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // a follower never saw the round's challenge
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
        }
        __syncthreads();
        if (s_flag == 2) return;
        arrived_before += nblk;
        if (!first) {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                if (!n[bb]) continue;
                Vin[bb] = a.Vbuf[bb][oi[bb]];
                Min[bb] = a.Mbuf[bb][oi[bb]];
                oi[bb] ^= 1;
                n[bb] >>= 1;
            }
        }
        first = false;
    }
}

// Hybrid tail: the live tables of a phase (at most 256 entries each) to mapped host memory in one small launch; seq is written last.
struct export_args { const fr_t *V[2], *M[2]; uint32_t n[2]; export_out *out; unsigned long long seq; };
struct k_export_f {                    // (a functor: the lanes of a batch export in ONE launch, launch.cuh)
    export_args a;
    __device__ __forceinline__ void operator()() const {
        const uint32_t b = threadIdx.x >> 8, i = threadIdx.x & 255;
        if (i < a.n[b]) {
            fr_store(&a.out->V[b][i], fr_load(a.V[b] + i));
            fr_store(&a.out->M[b][i], fr_load(a.M[b] + i));
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            *((volatile unsigned long long *) &a.out->seq) = a.seq;
        }
    }
};

// Device-side Fiat-Shamir rounds ("removes the ~10^3 host round-trips per proof").
//
// In the interactive protocol the verifier draws each round's challenge after seeing the round polynomial, so every sumcheck round is a
// launch, a hand-over to the host and a host turn-around (~25-30 us, of which the kernel is ~10). In the NON-interactive mode the next
// challenge is a hash of the transcript, which the prover can compute itself: once the live tables of a phase are small (<= 2^12 entries,
// which is 12 of the ~15-21 rounds of every vgg11 phase) ONE single-workgroup kernel runs all remaining rounds of the phase -- fold, round
// sums, the O(1) add_term bookkeeping of reference src/prover.cpp:360-426, the BLAKE2s chain step on the round polynomial and the
// derivation of the next challenge -- and hands all round polynomials, challenges and final table entries to the host at once. The host
// prover then answers the verifier's calls from that record and checks that the verifier's challenges are the ones the device derived
// (they are the same function of the same transcript: host/replay.hpp fiatShamir). The interactive path is untouched.
//
// Chain step (identical on both sides): state' = BLAKE2s-256(state || message), message of a quadratic round = the three coefficients as
// they lie in memory (Montgomery limbs, 96 bytes: a bijection of the field elements, so equally binding, and no conversion on the
// critical path) -- 128 bytes, two compressions. Challenge = state' with bit 255 cleared, taken AS Montgomery limbs if below r, else one
// more chain step with an empty message.
#pragma once
#include "kernels.cuh"
#include "../ff/blake2s.hpp"

#define FS_TAIL_THREADS 576            // 8 waves of quads + one wave of scalar bookkeeping
#define FS_TAIL_SLOTS 128              // quads in flight: four lanes each on 8 waves
#define FS_TAIL_QUADS 512              // quads (both table pairs together) the kernel takes: its first two rounds then make 4 and 2 passes over the slots
#define FS_TAIL_MAX_ROUNDS ZK_MAX_VARS

struct tail_out {                     // pinned, mapped host memory
    fr_t poly[FS_TAIL_MAX_ROUNDS][3]; // round polynomials (a, b, c) as the host's quad_round returns them
    fr_t chal[FS_TAIL_MAX_ROUNDS];    // challenge derived after each of them
    fr_t add_term;                    // bookkeeping scalar after the last round
    fr_t tail_v[2][2];                // the two entries left in each V table ...
    fr_t final_v[2];                  // ... or the value it collapsed to
    uint32_t pair_state[2];           // 0 absent, 1 two entries left (tail_v), 2 collapsed (final_v)
    uint32_t fs_state[8];             // chain state after the last challenge
    unsigned long long seq;           // written last
};

struct tail_args {
    const fr_t *Vin[2], *Min[2];      // current tables (V may still be a layer's values: first round)
    fr_t *Vbuf[2][2], *Mbuf[2][2];    // ping-pong buffers; out_idx[b] = the one the next fold writes
    int32_t out_idx[2];
    uint64_t n[2];                    // pre-fold length of each pair; 0 = absent
    int32_t first, rounds, with_add_term;
    fr_t prev_r, add_term;
    uint32_t fs_state[8];
    tail_out *out;
    unsigned long long seq;
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

// Work layout of a round: quad q of the concatenated list [pair 0 | pair 1] belongs to lanes 4q .. 4q + 3 (no loop: the tables hold at
// most FS_TAIL_QUADS quads together when the kernel starts); the last wave does the scalar bookkeeping at the same time --
// add_term (1 - r), the final values of a pair that collapses this round and their product -- so that after the block reduction thread 0
// only combines, hashes and derives the challenge.
__global__ void __launch_bounds__(FS_TAIL_THREADS) k_fs_tail(tail_args a) {
    __shared__ fr_t s_part[FS_TAIL_THREADS / 64][3];
    __shared__ fr_t s_r, s_fin[2][2], s_prod[2], s_tail[2][2], s_add;
    __shared__ uint32_t s_state[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int scalar_wave = FS_TAIL_THREADS / 64 - 1;
    uint64_t n[2] = {a.n[0], a.n[1]};
    const fr_t *Vin[2] = {a.Vin[0], a.Vin[1]}, *Min[2] = {a.Min[0], a.Min[1]};
    int oi[2] = {a.out_idx[0], a.out_idx[1]};
    bool first = a.first != 0;
    uint32_t pstate[2] = {n[0] ? 1u : 0u, n[1] ? 1u : 0u};
    if (tid == 0) {
        s_r = a.prev_r;
        s_add = a.add_term;
#pragma unroll
        for (int i = 0; i < 8; ++i) s_state[i] = a.fs_state[i];
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
        if (wave == scalar_wave) {
            // scalar bookkeeping, concurrent with the quads: lane 0/1 (2/3): V and M of a collapsing pair 0 (1); lane 4: add_term (1 - r)
            if (lane < 4 && (lane < 2 ? collapse[0] : collapse[1])) {
                const int b = lane >> 1;                 // (constant indices below: a runtime index into a local array would live in scratch memory)
                const fr_t *src = (lane & 1) ? (b ? Min[1] : Min[0]) : (b ? Vin[1] : Vin[0]);
                fr_t x = fr_load(src);
                if (!first) x = fr_lerp(x, fr_load(src + 1), r);
                s_fin[b][lane & 1] = x;
            }
            if (lane == 4 && a.with_add_term && !first) s_add = fr_mul(s_add, fr_sub(fr_one(), r));
        }
        // Four lanes per quad (the layout of k_round_quad_fine: a lone wave issues its products one after the other, so the chain per
        // lane is what counts):   fold     role 0: v0   1: v1   2: m0   3: m1        (one product)
        //                         product  role 0: c = v0 m0   1: p(1) = v1 m1   2: a = (v1 - v0)(m1 - m0)     (one more)
        const uint32_t role = (uint32_t) tid & 3, total_quads = quads[0] + quads[1];
        fr_t prod = fr_zero();
        for (uint32_t base = 0; base == 0 || base < total_quads; base += FS_TAIL_SLOTS) {
            const uint32_t item = base + ((uint32_t) tid >> 2);
            const bool live = wave != scalar_wave && item < total_quads;
            const int b = (live && item >= quads[0]) ? 1 : 0;
            const uint32_t q = b ? item - quads[0] : item, quads_b = b ? quads[1] : quads[0];
            const fr_t *Vb = b ? Vin[1] : Vin[0], *Mb = b ? Min[1] : Min[0];
            const int oib = b ? oi[1] : oi[0];
            fr_t X = fr_zero(), opA = fr_zero(), opB = fr_zero();
            if (live && first) {
                if (role < 3) {
                    const fr_t v0 = fr_load(Vb + 2 * q), v1 = fr_load(Vb + 2 * q + 1), m0 = fr_load(Mb + 2 * q), m1 = fr_load(Mb + 2 * q + 1);
                    opA = role == 0 ? v0 : role == 1 ? v1 : fr_sub(v1, v0);
                    opB = role == 0 ? m0 : role == 1 ? m1 : fr_sub(m1, m0);
                    if (quads_b == 1 && role < 2) s_tail[b][role] = opA;
                }
            } else if (live) {
                const fr_t *src = (role < 2 ? Vb : Mb) + 4 * q + 2 * (role & 1);
                X = fr_lerp(fr_load(src), fr_load(src + 1), r);
                fr_store((role < 2 ? a.Vbuf[b][oib] : a.Mbuf[b][oib]) + 2 * q + (role & 1), X);
                if (quads_b == 1 && role < 2) s_tail[b][role] = X;           // the pair the phase may end with
            }
            if (!first && wave != scalar_wave) {
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
            if (wave != scalar_wave && (base == 0 || live)) prod = fr_add(prod, fr_mul(opA, opB));
        }
        // reduction: lanes of equal role by butterflies, lanes 0..2 of every wave that holds quads leave the wave's sums, threads 0..2 add those
        const int nw = (int) std::min<uint32_t>(FS_TAIL_THREADS / 64 - 1, (4 * total_quads + 63) / 64);
        if (wave < nw) {
#pragma unroll
            for (int off = 4; off < 64; off <<= 1) {
                fr_t o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o.v[i] = (uint32_t) __shfl_xor((int) prod.v[i], off, 64);
                prod = fr_add(prod, o);
            }
            if (lane < 3) s_part[wave][lane == 2 ? 0 : lane + 1] = prod;       // accumulator order a, c, p(1)
        }
        __syncthreads();
        if (wave == scalar_wave && (lane == 0 ? collapse[0] : lane == 2 ? collapse[1] : false)) s_prod[lane >> 1] = fr_mul(s_fin[lane >> 1][0], s_fin[lane >> 1][1]);
        fr_t tot = fr_zero();
        if (tid < 3)
            for (int w = 0; w < nw; ++w) tot = fr_add(tot, s_part[w][tid]);
        __syncthreads();
        if (wave == 0) {
            fr_t ca = tot, cc = fr_shfl_down(tot, 1), p1 = fr_shfl_down(tot, 2);
            if (tid == 0) {
                // host bookkeeping of quad_round (sumcheck.hip), reference src/prover.cpp:368-383
                fr_t add_term = s_add;
                fr_t cb = fr_sub(fr_sub(p1, ca), cc);
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    if (collapse[b]) add_term = fr_add(add_term, s_prod[b]);
                s_add = add_term;
                if (a.with_add_term) { cb = fr_sub(cb, add_term); cc = fr_add(cc, add_term); }
                fr_store(&a.out->poly[k][0], ca);
                fr_store(&a.out->poly[k][1], cb);
                fr_store(&a.out->poly[k][2], cc);
                // chain step on the 96 bytes of (a, b, c) as they lie in memory, then the challenge
                uint32_t st[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) st[i] = s_state[i];
                const fr_t ch = fs_round_challenge(st, ca, cb, cc);
                s_r = ch;
                fr_store(&a.out->chal[k], ch);
#pragma unroll
                for (int i = 0; i < 8; ++i) s_state[i] = st[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (!n[b]) continue;
            if (collapse[b]) { n[b] = 0; pstate[b] = 2; continue; }
            if (!first) {
                Vin[b] = a.Vbuf[b][oi[b]];
                Min[b] = a.Mbuf[b][oi[b]];
                oi[b] ^= 1;
                n[b] >>= 1;
            }
        }
        first = false;
    }
    if (tid == 0) {
        tail_out *o = a.out;
        fr_store(&o->add_term, s_add);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            o->pair_state[b] = pstate[b];
            if (pstate[b] == 1) { fr_store(&o->tail_v[b][0], s_tail[b][0]); fr_store(&o->tail_v[b][1], s_tail[b][1]); }
            if (pstate[b] == 2) fr_store(&o->final_v[b], s_fin[b][0]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) o->fs_state[i] = s_state[i];
        __threadfence_system();
        *((volatile unsigned long long *) &o->seq) = a.seq;
    }
}


// Hybrid tail: the live tables of a phase (at most 256 entries each) to mapped host memory in one small launch; seq is written last.
struct export_out { fr_t V[2][256], M[2][256]; unsigned long long seq; };
struct export_args { const fr_t *V[2], *M[2]; uint32_t n[2]; export_out *out; unsigned long long seq; };
__global__ void __launch_bounds__(512) k_export_tables(export_args a) {
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

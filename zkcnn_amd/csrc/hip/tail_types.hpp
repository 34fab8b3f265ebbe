// Host-visible records of the resident / device-side round kernels (fs_tail.cuh): mailboxes, the phase record, the broadcast line. Plain data,
// shared by the kernels and by the host code that allocates and reads them (context.hip, sumcheck.hip).
#pragma once
#include "types.cuh"

#define TAIL_THREADS 1024              // 16 waves, four lanes per quad
#define TAIL_SLOTS 256                 // quads in flight per pass
#define TAIL_QUADS 256                 // quads (both table pairs together) the kernel accepts: one pass over the slots. (A single workgroup is one CU:
                                       // 16 waves share 4 SIMDs, so a 256-quad round costs 4 x 2 products per SIMD -- beyond that a launch over many CUs wins)
#define FS_TAIL_MAX_ROUNDS ZK_MAX_VARS
#define TAIL_TIMEOUT_TICKS 300000000ull   // 3 s of s_memrealtime (100 MHz): a live kernel nobody talks to gives up
#define TAIL_ABORT 0xffffffffu
#define TAIL_EXPORT_MAX 64             // entries per table a shortened resident tail hands to the host
#define MID_MAX_BLOCKS 256u           // k_mid's grid: its workgroups wait for one another, so the footprint is bounded (1024 waves) whatever the table size

struct __align__(16) live_in {        // mapped host memory, written by the host: chunk j = {challenge words 3j, 3j+1, 3j+2, seq} (chunk 2: words 6, 7, 0)
    uint32_t c[3][4];
    uint32_t pad_[4];
};
struct __align__(16) live_out {       // mapped host memory, written by the kernel: chunk j = {words 3j .. 3j+2 of (a, b, c), seq}
    uint32_t c[8][4];
};

struct tail_out {                     // pinned, mapped host memory
    fr_t poly[FS_TAIL_MAX_ROUNDS][3]; // FS: round polynomials (a, b, c) as the host's quad_round returns them
    fr_t chal[FS_TAIL_MAX_ROUNDS];    // FS: challenge derived after each of them
    fr_t add_term;                    // bookkeeping scalar after the last round
    fr_t tail_v[2][2];                // the two entries left in each V table ...
    fr_t final_v[2];                  // ... or the value it collapsed to
    uint32_t pair_state[2];           // 0 absent, 1 two entries left (tail_v), 2 collapsed (final_v)
    uint32_t fs_state[8];             // FS: chain state after the last challenge
    uint32_t status, pad_;            // LIVE: 0 running / finished, 1 aborted by the host, 2 timed out
    unsigned long long ticks_wait, ticks_total;   // LIVE: 100 MHz ticks spent polling for challenges / in the whole kernel (diagnostics, written with the final state)
    unsigned long long seq;           // FS: written last
    live_out live;                    // LIVE: the round mailbox
    // LIVE with tail_args::export_tables: the kernel ends BEFORE the phase does and hands what is left of the tables to the host (pair_state 3: exp_n[b] entries of
    // V and M as its last round's fold left them); the host runs the remaining rounds itself (rounds_resident.hpp: host_tail_round)
    fr_t exp_V[2][TAIL_EXPORT_MAX], exp_M[2][TAIL_EXPORT_MAX];
    uint32_t exp_n[2];
};

struct __align__(16) mid_bcast {
    uint32_t c[3][4];                 // challenge chunks {3 words, seq}, like live_in
    uint32_t st[8];                   // Fiat-Shamir: chain state after the challenge (written BEFORE the chunks: whoever sees the challenge sees the state)
    uint32_t pad_[4];
};

struct export_out { fr_t V[2][256], M[2][256]; unsigned long long seq; };

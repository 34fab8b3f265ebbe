// The streaming form of the quadratic round (reference src/prover.cpp:396-426, sumcheckUpdateEach, for tables far above the caches):
// fold V and M with the previous challenge, store the halves, and accumulate this round's sums over the folded pairs.
//
// Work split: a WAVE takes a tile of 64 pairs per table and step -- 4 KB of V and 4 KB of M, read as whole 1 KB wave accesses -- and LANE l folds
// pair 64 t + l of BOTH tables (two products by the wave-uniform challenge, which lives in scalar registers). Folded pair p is one half of
// quad p / 2, so the round sums need only the neighbour lane:
//     even lane:  c += v'[p] m'[p]                              (the quad's  v0 m0)
//     odd  lane:  a += (v'[p] - v'[p-1]) (m'[p] - m'[p-1])      (the quad's  dv dm; the neighbour's values come by DPP)
// -- one product per lane, accumulated as a 512-bit integer and reduced once per thread (fr_stream.cuh). p(1) = sum v1 m1 is only needed when
// the caller cannot derive b from the running claim (P1: the second accumulator): 2.5 products per lane and step instead of 3.5 per pair before.
// Per quad that is 4 x 120 + 2 x 64 multiply-accumulate steps against 6 (7) x 128.
//
// Memory (what the product uses: LOAD_DMA, STORE_DIRECT): a tile is fetched by LDS-DMA -- four contiguous 1 KB wave accesses per table that land in
// the wave's staging area (8 KB: V's tile, M's tile) without passing through registers, issued for the NEXT tile as soon as this one is in registers, so
// the fetch runs under the whole step's arithmetic; the lane then reads its own 64 bytes with ds_read_b128 (source-side swizzle: no bank conflicts).
// No barrier anywhere: a wave stages only its own tiles. The other modes are the experiment's (scripts/exp/round_lab.hip): LOAD_DIRECT reads a lane's 64
// contiguous bytes with four 16-byte loads (each wave access strides over 4 KB), LOAD_LDS loads to registers and transposes through LDS, STORE_LDS
// transposes the folded halves too. MI355X, two 2^24-entry tables, per launch: 0.42-0.47 ms round 4's loop, 0.33 ms this one (4.8-4.9 TB/s of algorithmic
// traffic); the same accesses without arithmetic 0.30 ms, the arithmetic without accesses 0.25 ms (profiles/r05_round_lab.md).
#pragma once
#include "fr_stream.cuh"

enum { RS_LOAD_DIRECT = 0, RS_LOAD_LDS = 1, RS_LOAD_DMA = 2, RS_LOAD_NONE = 3 };      // (NONE: experiment, arithmetic only)
enum { RS_STORE_DIRECT = 0, RS_STORE_LDS = 1 };
#define RS_STAGE_SLOTS 256u          // 16-byte slots per wave: one tile of one table

// slot of 16-byte piece s of a tile: lane L owns pieces 4L .. 4L+3; XOR-ing bits 4..5 into bits 0..1 spreads the 16 lanes of one ds_read_b128 group
// over all 16 slot columns, and leaves the 8 consecutive pieces of one ds_write_b128 group consecutive
__device__ __forceinline__ uint32_t rs_slot(uint32_t s) { return s ^ ((s >> 4) & 3u); }

__device__ __forceinline__ fr_t rs_make(const uint4 &lo, const uint4 &hi) {
    fr_t z;
    z.v[0] = lo.x; z.v[1] = lo.y; z.v[2] = lo.z; z.v[3] = lo.w;
    z.v[4] = hi.x; z.v[5] = hi.y; z.v[6] = hi.z; z.v[7] = hi.w;
    return z;
}

// the neighbour lane's copy for odd lanes, the lane's own for even ones (DPP quad_perm [0, 0, 2, 2])
__device__ __forceinline__ fr_t rs_from_even(const fr_t &a) {
    fr_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = (uint32_t) __builtin_amdgcn_mov_dpp((int) a.v[i], 0xA0, 0xf, 0xf, false);
    return z;
}

// LDS-DMA: tile `tile` of `tab` lands in the wave's 256 slots at `stage` while the wave computes (global_load_lds_dwordx4: the data bypass
// the registers; destination = wave-uniform base + 16 lane, so the swizzle is applied to the SOURCE piece -- still one contiguous 1 KB per access)
// `pieces`: 16-byte pieces of the tile that exist (4 per pair; 256 for every tile but a table's last): a lane whose piece lies behind them fetches the
// last one that does -- a table shorter than a tile, or one that ends inside it, is never read past its end (those lanes' values are discarded).
template <bool NT = false>
__device__ __forceinline__ void rs_dma_tile(const fr_t *tab, uint64_t tile, uint32_t lane, uint4 *stage, uint32_t pieces = 256) {
    const uint4 *src = reinterpret_cast<const uint4 *>(tab) + tile * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = min(rs_slot(i * 64 + lane), pieces - 1);
        if (NT) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + s),
                                                 (__attribute__((address_space(3))) void *) (stage + i * 64), 16, 0, 2);
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (src + s),
                                              (__attribute__((address_space(3))) void *) (stage + i * 64), 16, 0, 0);
    }
}
// pieces of tile t of a table of `npairs` pairs
__device__ __forceinline__ uint32_t rs_tile_pieces(uint64_t npairs, uint64_t t) {
    const uint64_t left = npairs - t * 64;
    return left < 64 ? 4u * (uint32_t) left : 256u;
}
__device__ __forceinline__ void rs_read_staged(const uint4 *stage, uint32_t lane, fr_t &e0, fr_t &e1) {
    uint4 y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = stage[rs_slot(4 * lane + k)];
    e0 = rs_make(y[0], y[1]);
    e1 = rs_make(y[2], y[3]);
}

// entries 2p, 2p+1 of `tab` for the lane's pair p = 64 tile + lane
template <int LOADM>
__device__ __forceinline__ void rs_load_pair(const fr_t *tab, uint64_t tile, uint32_t lane, uint4 *stage, fr_t &e0, fr_t &e1) {
    const uint4 *src = reinterpret_cast<const uint4 *>(tab) + tile * 256;
    if (LOADM == RS_LOAD_DIRECT) {
        const uint4 *p = src + 4 * lane;
        const uint4 y0 = p[0], y1 = p[1], y2 = p[2], y3 = p[3];
        e0 = rs_make(y0, y1);
        e1 = rs_make(y2, y3);
    } else {
        uint4 x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = src[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) stage[rs_slot(i * 64 + lane)] = x[i];
        uint4 y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = stage[rs_slot(4 * lane + k)];
        e0 = rs_make(y[0], y[1]);
        e1 = rs_make(y[2], y[3]);
    }
}

// out[64 tile + lane] = x for lanes below `valid`
template <int STOREM, bool NT = false>
__device__ __forceinline__ void rs_store_half(fr_t *out, uint64_t tile, uint32_t lane, uint32_t valid, uint4 *stage, const fr_t &x) {
    uint4 *dst = reinterpret_cast<uint4 *>(out) + tile * 128;
    const uint4 lo = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]), hi = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    if (STOREM == RS_STORE_DIRECT) {
        if (lane < valid) {
            if (NT) {
                typedef uint32_t rs_v4 __attribute__((ext_vector_type(4)));
                rs_v4 *d4 = reinterpret_cast<rs_v4 *>(dst);
                __builtin_nontemporal_store(rs_v4{lo.x, lo.y, lo.z, lo.w}, d4 + 2 * lane);
                __builtin_nontemporal_store(rs_v4{hi.x, hi.y, hi.z, hi.w}, d4 + 2 * lane + 1);
            }
            else { dst[2 * lane] = lo; dst[2 * lane + 1] = hi; }
        }
    } else {
        stage[2 * lane] = lo;
        stage[2 * lane + 1] = hi;
        const uint4 y0 = stage[lane], y1 = stage[64 + lane];
        if (lane < 2 * valid) dst[lane] = y0;
        if (64 + lane < 2 * valid) dst[64 + lane] = y1;
    }
}

// One table pair, pairs [0, npairs) (npairs even: whole quads), waves gw, gw + nw, ... of the launch. Vout / Mout receive npairs entries each.
// Returns this lane's share of the sums: even lanes hold part of c, odd lanes part of a (and, with P1, every lane part of p(1)).
// stage: the wave's LDS area -- RS_STAGE_SLOTS slots (LOAD_LDS / STORE_LDS), twice that for LOAD_DMA (V's tile, then M's).
template <int LOADM, int STOREM, bool P1, int NT = 0>
__device__ __forceinline__ void rs_fold_accumulate(const fr_t *Vin, const fr_t *Min, fr_t *Vout, fr_t *Mout, uint64_t npairs, const fr_u &r,
                                                   uint64_t gw, uint64_t nw, uint4 *stage, fr_t &sum, fr_t &sum_p1) {
    const uint32_t lane = threadIdx.x & 63;
    const bool odd = lane & 1;
    const uint32_t mod[8] = FR_MOD_INIT;
    fr_acc512 W = fr_acc512_zero(), W1 = fr_acc512_zero();
    const uint64_t ntiles = (npairs + 63) / 64;
    if (LOADM == RS_LOAD_DMA && gw < ntiles) {
        rs_dma_tile<(NT & 1) != 0>(Vin, gw, lane, stage, rs_tile_pieces(npairs, gw));
        rs_dma_tile<(NT & 1) != 0>(Min, gw, lane, stage + RS_STAGE_SLOTS, rs_tile_pieces(npairs, gw));
    }
    fr_t v = fr_zero(), m = fr_zero();
    for (uint64_t t = gw; t < ntiles; t += nw) {
        const uint64_t left = npairs - t * 64;
        const uint32_t valid = left < 64 ? (uint32_t) left : 64u;
        fr_t e0, e1, f0, f1;
        if (LOADM == RS_LOAD_DMA) {
            // this tile has landed (and the previous step's stores are acknowledged); once it is in registers the next one may overwrite it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rs_read_staged(stage, lane, e0, e1);
            rs_read_staged(stage + RS_STAGE_SLOTS, lane, f0, f1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + nw < ntiles) {
                rs_dma_tile<(NT & 1) != 0>(Vin, t + nw, lane, stage, rs_tile_pieces(npairs, t + nw));
                rs_dma_tile<(NT & 1) != 0>(Min, t + nw, lane, stage + RS_STAGE_SLOTS, rs_tile_pieces(npairs, t + nw));
            }
            v = fr_lerp_u(e0, e1, r);
            rs_store_half<STOREM, (NT & 2) != 0>(Vout, t, lane, valid, stage, v);
            m = fr_lerp_u(f0, f1, r);
            rs_store_half<STOREM, (NT & 2) != 0>(Mout, t, lane, valid, stage, m);
        } else if (LOADM == RS_LOAD_NONE) {
            e0 = v; e1 = m; e0.v[0] ^= (uint32_t) t; e1.v[1] += lane;
            v = fr_lerp_u(e0, e1, r);
            e0.v[2] ^= 0x55u;
            m = fr_lerp_u(e1, e0, r);
            if (valid == 77) { rs_store_half<RS_STORE_DIRECT>(Vout, t, lane, valid, stage, v); rs_store_half<RS_STORE_DIRECT>(Mout, t, lane, valid, stage, m); }
        } else {
            rs_load_pair<LOADM>(Vin, t, lane, stage, e0, e1);
            v = fr_lerp_u(e0, e1, r);
            rs_store_half<STOREM>(Vout, t, lane, valid, stage, v);
            rs_load_pair<LOADM>(Min, t, lane, stage, e0, e1);
            m = fr_lerp_u(e0, e1, r);
            rs_store_half<STOREM>(Mout, t, lane, valid, stage, m);
        }
        if (lane >= valid) { v = fr_zero(); m = fr_zero(); }      // (a partial tile: whole lane pairs, npairs is even)
        // even: (v - r + r)(m - r + r) = v m;  odd: (v - v_prev + r)(m - m_prev + r)
        fr_t pv = rs_from_even(v), pm = rs_from_even(m);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pv.v[i] = odd ? pv.v[i] : mod[i];
            pm.v[i] = odd ? pm.v[i] : mod[i];
        }
        fr_acc512_mac(W, fr_sub_lazy(v, pv), fr_sub_lazy(m, pm));
        if (P1) {
            fr_t v1 = v;
            if (!odd) v1 = fr_zero();
            fr_acc512_mac(W1, v1, m);
        }
    }
    sum = fr_acc512_reduce(W);
    sum_p1 = P1 ? fr_acc512_reduce(W1) : fr_zero();
}

// The FIRST round of a phase (reference src/prover.cpp:368-383 with nothing to fold yet): lane l takes pair 64 t + l of both tables and accumulates
// c += v0 m0, p(1) += v1 m1, a += (v1 - v0)(m1 - m0) as three 512-bit sums; `npairs` pairs, tiles through the same LDS-DMA staging.
__device__ __forceinline__ void rs_first_accumulate(const fr_t *Vin, const fr_t *Min, uint64_t npairs, uint64_t gw, uint64_t nw, uint4 *stage,
                                                    fr_t &sum_a, fr_t &sum_c, fr_t &sum_p1) {
    const uint32_t lane = threadIdx.x & 63;
    fr_acc512 Wa = fr_acc512_zero(), Wc = fr_acc512_zero(), Wp = fr_acc512_zero();
    const uint64_t ntiles = (npairs + 63) / 64;
    if (gw < ntiles) {
        rs_dma_tile<true>(Vin, gw, lane, stage, rs_tile_pieces(npairs, gw));
        rs_dma_tile<true>(Min, gw, lane, stage + RS_STAGE_SLOTS, rs_tile_pieces(npairs, gw));
    }
    for (uint64_t t = gw; t < ntiles; t += nw) {
        const uint64_t left = npairs - t * 64;
        const uint32_t valid = left < 64 ? (uint32_t) left : 64u;
        fr_t v0, v1, m0, m1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rs_read_staged(stage, lane, v0, v1);
        rs_read_staged(stage + RS_STAGE_SLOTS, lane, m0, m1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + nw < ntiles) {
            rs_dma_tile<true>(Vin, t + nw, lane, stage, rs_tile_pieces(npairs, t + nw));
            rs_dma_tile<true>(Min, t + nw, lane, stage + RS_STAGE_SLOTS, rs_tile_pieces(npairs, t + nw));
        }
        if (lane >= valid) { v0 = fr_zero(); v1 = fr_zero(); m0 = fr_zero(); m1 = fr_zero(); }
        fr_acc512_mac(Wc, v0, m0);
        fr_acc512_mac(Wp, v1, m1);
        fr_acc512_mac(Wa, fr_sub_lazy(v1, v0), fr_sub_lazy(m1, m0));
    }
    sum_a = fr_acc512_reduce(Wa);
    sum_c = fr_acc512_reduce(Wc);
    sum_p1 = fr_acc512_reduce(Wp);
}

__device__ __forceinline__ fr_t rs_shfl_xor(const fr_t &a, int mask) {
    fr_t z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = (uint32_t) __shfl_xor((int) a.v[i], mask, 64);
    return z;
}
// Block totals of rs_fold_accumulate's lane sums in thread 0: acc[0] = a (odd lanes), acc[1] = c (even lanes), acc[2] = p(1) (all lanes; zero
// without P1). One butterfly over the lanes of equal parity instead of three wave sums. smem: 3 * ZK_BLOCK / 64 elements.
template <bool P1>
__device__ __forceinline__ void rs_block_totals(fr_t s, fr_t s1, fr_t (&acc)[3], fr_t *smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
#pragma unroll
    for (int d = 2; d < 64; d <<= 1) s = fr_add(s, rs_shfl_xor(s, d));
    if (P1) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) s1 = fr_add(s1, rs_shfl_xor(s1, d));
    }
    if (lane == 0) { smem[wave * 3 + 1] = s; smem[wave * 3 + 2] = s1; }
    if (lane == 1) smem[wave * 3] = s;
    __syncthreads();
    acc[0] = fr_zero(); acc[1] = fr_zero(); acc[2] = fr_zero();
    if (threadIdx.x == 0) {
        for (int w = 0; w < nwave; ++w) {
            acc[0] = fr_add(acc[0], smem[w * 3]);
            acc[1] = fr_add(acc[1], smem[w * 3 + 1]);
            if (P1) acc[2] = fr_add(acc[2], smem[w * 3 + 2]);
        }
    }
}

// the memory side alone (experiment): the same accesses, an XOR instead of the arithmetic
template <int LOADM, int STOREM>
__device__ __forceinline__ void rs_copy_only(const fr_t *Vin, const fr_t *Min, fr_t *Vout, fr_t *Mout, uint64_t npairs, uint64_t gw, uint64_t nw, uint4 *stage) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t ntiles = (npairs + 63) / 64;
    if (LOADM == RS_LOAD_DMA && gw < ntiles) {
        rs_dma_tile(Vin, gw, lane, stage);
        rs_dma_tile(Min, gw, lane, stage + RS_STAGE_SLOTS);
    }
    for (uint64_t t = gw; t < ntiles; t += nw) {
        fr_t e0, e1, f0, f1;
        if (LOADM == RS_LOAD_DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rs_read_staged(stage, lane, e0, e1);
            rs_read_staged(stage + RS_STAGE_SLOTS, lane, f0, f1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + nw < ntiles) {
                rs_dma_tile(Vin, t + nw, lane, stage);
                rs_dma_tile(Min, t + nw, lane, stage + RS_STAGE_SLOTS);
            }
        } else {
            rs_load_pair<LOADM>(Vin, t, lane, stage, e0, e1);
            rs_load_pair<LOADM>(Min, t, lane, stage, f0, f1);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { e0.v[i] ^= e1.v[i]; f0.v[i] ^= f1.v[i]; }
        rs_store_half<STOREM>(Vout, t, lane, 64, stage, e0);
        rs_store_half<STOREM>(Mout, t, lane, 64, stage, f0);
    }
}

// Lock-step batches, sumcheck unit: the batched forms of its kernels and the flush that issues them (ctx.hpp: zk_batch; include/zkcnn_hip.h:
// zk_batch_*). A batched kernel is the SAME device body as the per-context kernel, run by grid row blockIdx.y on lane blockIdx.y's argument
// block; the blocks travel by value in the kernel argument segment (<= 4 KB: ZK_BATCH_MAX_LANES blocks of <= ZK_BATCH_ARG_BYTES), so a fused
// launch costs no copy and no extra dependent memory access -- the lane's block is read with scalar loads like any kernel argument.
// reference src/prover.cpp:360-426 (sumcheckUpdate1/2 -> sumcheckUpdateEach) is what each row computes.
#pragma once
#include "ctx.hpp"
#include "kernels.cuh"

template <class A>
struct lanes_t {
    A a[ZK_BATCH_MAX_LANES];
};
static_assert(sizeof(round2_args) <= ZK_BATCH_ARG_BYTES, "a lane's argument block must fit its slot of the pending queue");
static_assert(sizeof(lanes_t<round2_args>) <= 4096, "kernel argument segment");

// issue priority as in ZK_LATENCY_PRIO, for the whole (lanes x blocks) grid
#define ZK_LATENCY_PRIO_B() do { if (gridDim.x * gridDim.y <= ZK_LATENCY_GRID) __builtin_amdgcn_s_setprio(3); } while (0)

__global__ void __launch_bounds__(ZK_BLOCK) k_round_quad_fine_b(lanes_t<round2_args> L) {
    ZK_LATENCY_PRIO_B();
    const round2_args &a = L.a[blockIdx.y];
    if (blockIdx.x >= round_fine_blocks(a)) return;            // (lanes may differ in length when one of them has been detached and re-attached mid-proof: never in lock step)
    round_quad_fine_body(a, blockIdx.x);
}
__global__ void __launch_bounds__(ZK_BLOCK) k_round_quad2_b(lanes_t<round2_args> L) {
    ZK_LATENCY_PRIO_B();
    const round2_args &a = L.a[blockIdx.y];
    if (blockIdx.x >= a.blocks[0] + a.blocks[1]) return;       // a lane's live prefix (its witness) decides how many blocks it asked for
    round_quad2_body(a, blockIdx.x);
}

// one launch per kind over the lanes that deferred one (lanes in lock step all defer the same kind; anything else still runs, in as many
// launches as there are kinds)
template <class A, class Launch>
static int32_t flush_kind(zk_batch *b, std::vector<batch_item> &items, int kind, int prof_class, Launch launch) {
    lanes_t<A> L;
    uint32_t n = 0, blocks = 0;
    double bytes = 0;
    zk_ctx *first = nullptr;
    for (const batch_item &it : items) {
        if (it.kind != kind) continue;
        if (n == ZK_BATCH_MAX_LANES) { b->err = "more deferred launches of one kind than lanes"; return ZK_ERR_STATE; }
        std::memcpy(&L.a[n++], it.arg, sizeof(A));
        blocks = std::max(blocks, it.blocks);
        bytes += it.bytes;
        if (!first) first = it.ctx;
    }
    if (!n) return ZK_OK;
    zk_ctx *ctx = first;                   // (the profiler books a fused launch on its first lane)
    prof_begin(ctx, prof_class, bytes);
    launch(L, dim3(blocks, n), b->stream);
    prof_end(ctx, prof_class);
    if (hipGetLastError() != hipSuccess) { b->err = "batched launch failed"; return ZK_ERR_HIP; }
    ++b->n_launches;
    b->n_lane_launches += n;
    return ZK_OK;
}

int32_t zk_batch_flush_sumcheck(zk_batch *b, std::vector<batch_item> &items) {
    int32_t rc;
    if ((rc = flush_kind<round2_args>(b, items, BK_ROUND_FINE, PC_ROUND_QUAD, [](const lanes_t<round2_args> &L, dim3 grid, hipStream_t st) {
             hipLaunchKernelGGL(k_round_quad_fine_b, grid, dim3(ZK_BLOCK), 0, st, L);
         }))) return rc;
    if ((rc = flush_kind<round2_args>(b, items, BK_ROUND_QUAD2, PC_ROUND_QUAD, [](const lanes_t<round2_args> &L, dim3 grid, hipStream_t st) {
             hipLaunchKernelGGL(k_round_quad2_b, grid, dim3(ZK_BLOCK), 0, st, L);
         }))) return rc;
    return ZK_OK;
}

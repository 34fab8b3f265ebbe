// Kernels as functors: one definition, two launch forms.
//
// A kernel of the sumcheck unit is a plain struct F -- its members are the kernel's arguments, its operator() the kernel's body -- and is
// launched through zk_launch_f. For a context on its own that is k_run<F><<<grid>>>(f): exactly the kernel it would be with positional
// arguments. For a context that is a LANE of a lock-step batch (ctx.hpp: zk_batch) the launch is deferred, and zk_batch_flush runs the same
// body for all lanes in ONE launch: k_run_b<F><<<(gx, gy, lanes)>>>, row blockIdx.z working on lane blockIdx.z's argument block. The blocks
// travel by value in the kernel argument segment (scalar loads, like any argument) -- or, when eight of them exceed its 4 KB, through a ring
// of pinned host memory that the kernel reads in place (k_run_p).
//
// What a functor's body may assume: blockIdx.x / blockIdx.y as in its own launch, and gridDim.x / gridDim.y AT LEAST as large as its own
// launch's (a fused launch takes the largest grid any lane asked for -- the lanes' live prefixes differ with their witnesses): grid-stride
// loops and `if (i >= n) return` guards are fine as they are; a body that counts arrivals or splits the grid into block ranges carries its
// own block count in its arguments. Bodies launched through zk_launch_d (positional parameters, written for exactly their own grid) are only
// fused with lanes that asked for the SAME grid. No body may use gridDim.z / blockIdx.z: that is the lane.
// reference: each functor names the lines of src/prover.cpp it computes, as the kernels did.
#pragma once
#include <type_traits>
#include "ctx.hpp"

template <class A>
struct lanes_t {
    A a[ZK_BATCH_MAX_LANES];
};

template <class F, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_run(F f) { f(); }
template <class F, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_run_b(lanes_t<F> L) { L.a[blockIdx.z](); }
template <class F, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_run_p(const F *lanes) { lanes[blockIdx.z](); }

// issue priority for launches that are links of a proof's dependent chain (types.cuh: ZK_LATENCY_PRIO), for the whole fused grid
#define ZK_LATENCY_PRIO_F() do { if (gridDim.x * gridDim.y * gridDim.z <= ZK_LATENCY_GRID) __builtin_amdgcn_s_setprio(3); } while (0)

int32_t zk_batch_ring_slot(zk_batch *b, size_t bytes, void **host, void **dev);      // context.hip

// the fused launch of n lanes' deferred launches of F (all of one generation)
template <class F, int BLOCK>
static int32_t launch_lanes(zk_batch *b, const batch_item *const *items, uint32_t n, uint32_t gx, uint32_t gy) {
    static_assert(std::is_trivially_copyable<F>::value, "argument blocks are copied as they are");
    if constexpr (sizeof(lanes_t<F>) <= 4096) {
        lanes_t<F> L;
        for (uint32_t i = 0; i < n; ++i) std::memcpy((void *) &L.a[i], items[i]->arg, sizeof(F));
        for (uint32_t i = n; i < ZK_BATCH_MAX_LANES; ++i) std::memcpy((void *) &L.a[i], items[0]->arg, sizeof(F));
        hipLaunchKernelGGL((k_run_b<F, BLOCK>), dim3(gx, gy, n), dim3(BLOCK), 0, b->stream, L);
    } else {
        void *h = nullptr, *d = nullptr;
        int32_t rc = zk_batch_ring_slot(b, n * sizeof(F), &h, &d);
        if (rc) return rc;
        for (uint32_t i = 0; i < n; ++i) std::memcpy((unsigned char *) h + i * sizeof(F), items[i]->arg, sizeof(F));
        hipLaunchKernelGGL((k_run_p<F, BLOCK>), dim3(gx, gy, n), dim3(BLOCK), 0, b->stream, (const F *) d);
    }
    return hipGetLastError() == hipSuccess ? ZK_OK : ZK_ERR_HIP;
}

// launches functor f on ctx's stream, or defers it when ctx is a lane (then nothing is on the stream until the batch is flushed: callers
// that need the result go through zk_batch_sync_point -- wait_slot and ZK_ORDER do)
template <class F, int BLOCK = ZK_BLOCK>
static inline void zk_launch_f(zk_ctx *ctx, int cls, double bytes, dim3 grid, const F &f, bool exact_grid = false) {
    static_assert(std::is_trivially_copyable<F>::value && sizeof(F) <= ZK_BATCH_ARG_BYTES && sizeof(F) <= 4096, "a kernel functor is a small POD");
    if (ctx->batch) {
        ctx->batch->pending.emplace_back();
        batch_item &it = ctx->batch->pending.back();
        it.launch = &launch_lanes<F, BLOCK>;
        it.ctx = ctx;
        it.gen = ctx->n_pending++;
        it.gx = grid.x; it.gy = grid.y;
        it.exact = exact_grid;
        it.name = __PRETTY_FUNCTION__;
        it.prof_class = cls;
        it.bytes = bytes;
        std::memcpy(it.arg, (const void *) &f, sizeof(F));
        return;
    }
    prof_begin(ctx, cls, bytes);
    hipLaunchKernelGGL((k_run<F, BLOCK>), grid, dim3(BLOCK), 0, ctx->stream, f);
    prof_end(ctx, cls);
}

// ---- the same for a kernel body written as a __device__ function with positional parameters: zk_launch_d<body, BLOCK>(ctx, class, bytes, grid,
// arguments...) packs the arguments into a functor whose operator() calls the body (the commitment's kernels: msm_kernels.cuh) ----
template <class... T> struct pack_t;
template <> struct pack_t<> {};
template <class H, class... T> struct pack_t<H, T...> {
    H head;
    pack_t<T...> tail;
};
template <class... T> static inline pack_t<> make_pack() { return pack_t<>{}; }
template <class H, class... T> static inline pack_t<H, T...> make_pack(const H &h, const T &...t) {
    pack_t<H, T...> p;
    p.head = h;
    p.tail = make_pack<T...>(t...);
    return p;
}
template <auto Body, class... Got>
__device__ __forceinline__ void unpack_call(const pack_t<> &, const Got &...got) { Body(got...); }
template <auto Body, class H, class... T, class... Got>
__device__ __forceinline__ void unpack_call(const pack_t<H, T...> &p, const Got &...got) { unpack_call<Body>(p.tail, got..., p.head); }

template <auto Body, class... A>
struct call_f {
    pack_t<A...> args;
    __device__ __forceinline__ void operator()() const { unpack_call<Body>(args); }
};
template <auto Body, int BLOCK, class... A>
static inline void zk_launch_d(zk_ctx *ctx, int cls, double bytes, dim3 grid, const A &...a) {
    call_f<Body, A...> f;
    f.args = make_pack<A...>(a...);
    zk_launch_f<call_f<Body, A...>, BLOCK>(ctx, cls, bytes, grid, f, true);      // (a positional body was written for its own grid)
}

// experiment bench of the streaming quadratic round (hip/round_stream.cuh) against the round-4 loop (one thread per quad, 128-byte lane loads):
// identical tables and sums checked first, then time per launch on two 2^LOG_N-entry tables.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkcnn_amd/csrc -I zkcnn_amd/csrc/hip scripts/exp/round_lab.hip -o scripts/exp/round_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels.cuh"
#include "round_stream.cuh"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define BLOCK 256

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31);
}
__global__ void k_gen(fr_t *t, uint64_t n, uint64_t seed) {
    for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        fr_t z;
        for (int k = 0; k < 4; ++k) {
            const uint64_t h = mix(seed + i * 4 + k);
            z.v[2 * k] = (uint32_t) h; z.v[2 * k + 1] = (uint32_t) (h >> 32);
        }
        z.v[7] &= 0x3fffffffu;          // below r
        fr_store(t + i, z);
    }
}

// ---- round 4's loop (kernels.cuh: round_quad2_body, non-first round) ----
template <bool SKIP_P1>
__global__ void __launch_bounds__(BLOCK) k_old(const fr_t *Vin, const fr_t *Min, fr_t *Vout, fr_t *Mout, uint64_t Q, fr_t r, fr_t *partials) {
    __shared__ fr_t smem[3 * BLOCK / 64];
    fr_t acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    const uint64_t tid = blockIdx.x * (uint64_t) BLOCK + threadIdx.x, stride = (uint64_t) gridDim.x * BLOCK;
    for (uint64_t q = tid; q < Q; q += stride) {
        fr_t a0 = fr_load(Vin + 4 * q), a1 = fr_load(Vin + 4 * q + 1), a2 = fr_load(Vin + 4 * q + 2), a3 = fr_load(Vin + 4 * q + 3);
        fr_t v0 = fr_lerp(a0, a1, r), v1 = fr_lerp(a2, a3, r);
        fr_store(Vout + 2 * q, v0);
        fr_store(Vout + 2 * q + 1, v1);
        a0 = fr_load(Min + 4 * q); a1 = fr_load(Min + 4 * q + 1); a2 = fr_load(Min + 4 * q + 2); a3 = fr_load(Min + 4 * q + 3);
        fr_t m0 = fr_lerp(a0, a1, r), m1 = fr_lerp(a2, a3, r);
        fr_store(Mout + 2 * q, m0);
        fr_store(Mout + 2 * q + 1, m1);
        acc[0] = fr_add(acc[0], fr_mul(fr_sub(v1, v0), fr_sub(m1, m0)));
        acc[1] = fr_add(acc[1], fr_mul(v0, m0));
        if (!SKIP_P1) acc[2] = fr_add(acc[2], fr_mul(v1, m1));
    }
    fr_block_sum<3>(acc, smem);
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; ++k) fr_store(partials + 3 * blockIdx.x + k, acc[k]);
}

// ---- the streaming form ----
template <int LOADM, int STOREM, bool P1, int MINW, int NT = 0>
__global__ void __launch_bounds__(BLOCK, MINW) k_new(const fr_t *Vin, const fr_t *Min, fr_t *Vout, fr_t *Mout, uint64_t Q, fr_t r, fr_t *partials) {
    __shared__ fr_t smem[3 * BLOCK / 64];
    constexpr uint32_t SLOTS = LOADM == RS_LOAD_DMA ? 2 * RS_STAGE_SLOTS : (LOADM == RS_LOAD_LDS || STOREM == RS_STORE_LDS) ? RS_STAGE_SLOTS : 0;
    __shared__ uint4 stage[SLOTS ? (BLOCK / 64) * SLOTS : 1];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const fr_u ru = fr_uniform(r);
    fr_t s, s1;
    rs_fold_accumulate<LOADM, STOREM, P1, NT>(Vin, Min, Vout, Mout, 2 * Q, ru, blockIdx.x * (uint64_t) (BLOCK / 64) + wave, (uint64_t) gridDim.x * (BLOCK / 64),
                                          stage + wave * SLOTS, s, s1);
    fr_t acc[3];
    acc[0] = (lane & 1) ? s : fr_zero();
    acc[1] = (lane & 1) ? fr_zero() : s;
    acc[2] = s1;
    fr_block_sum<3>(acc, smem);
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; ++k) fr_store(partials + 3 * blockIdx.x + k, acc[k]);
}

template <int LOADM, int STOREM>
__global__ void __launch_bounds__(BLOCK) k_copy(const fr_t *Vin, const fr_t *Min, fr_t *Vout, fr_t *Mout, uint64_t Q, fr_t r, fr_t *partials) {
    constexpr uint32_t SLOTS = LOADM == RS_LOAD_DMA ? 2 * RS_STAGE_SLOTS : (LOADM == RS_LOAD_LDS || STOREM == RS_STORE_LDS) ? RS_STAGE_SLOTS : 0;
    __shared__ uint4 stage[SLOTS ? (BLOCK / 64) * SLOTS : 1];
    const uint32_t wave = threadIdx.x >> 6;
    rs_copy_only<LOADM, STOREM>(Vin, Min, Vout, Mout, 2 * Q, blockIdx.x * (uint64_t) (BLOCK / 64) + wave, (uint64_t) gridDim.x * (BLOCK / 64), stage + wave * SLOTS);
}

__global__ void k_sum(fr_t *out, const fr_t *partials, uint32_t nblk) {
    if (threadIdx.x < 3) {
        fr_t t = fr_zero();
        for (uint32_t b = 0; b < nblk; ++b) t = fr_add(t, fr_load(partials + 3 * b + threadIdx.x));
        fr_store(out + threadIdx.x, t);
    }
}
__global__ void k_diff(const uint4 *a, const uint4 *b, uint64_t n16, unsigned long long *cnt) {
    unsigned long long c = 0;
    for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n16; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        c += (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
    }
    if (c) atomicAdd(cnt, c);
}
// copy: the streaming ceiling with this kernel's read : write ratio (2 : 1)
__global__ void __launch_bounds__(BLOCK) k_stream21(const uint4 *a, const uint4 *b, uint4 *o, uint64_t n16) {
    for (uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x; i < n16; i += (uint64_t) gridDim.x * BLOCK) {
        const uint4 x = a[2 * i], y = a[2 * i + 1], z = b[2 * i], w = b[2 * i + 1];
        (void) y; (void) w;
        o[i] = make_uint4(x.x ^ y.x, x.y ^ z.y, x.z ^ w.z, x.w ^ y.w);
    }
}

struct bufs {
    fr_t *V, *M, *Vo, *Mo, *Vr, *Mr, *partials, *sums, *sums_ref;
    unsigned long long *cnt;
    uint64_t n;
};

template <class L>
static double time_it(L launch, int iters, hipStream_t st = 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <class K>
static void run_variant(const char *name, K kern, bufs &B, uint32_t blocks, const fr_t &r, bool p1, int iters) {
    const uint64_t Q = B.n / 4;
    CK(hipMemset(B.Vo, 0xEE, B.n / 2 * 32));
    CK(hipMemset(B.Mo, 0xEE, B.n / 2 * 32));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), 0, 0, B.V, B.M, B.Vo, B.Mo, Q, r, B.partials);
    hipLaunchKernelGGL(k_sum, dim3(1), dim3(64), 0, 0, B.sums, B.partials, blocks);
    CK(hipMemset(B.cnt, 0, 8));
    hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, 0, (const uint4 *) B.Vo, (const uint4 *) B.Vr, B.n / 2 * 2, B.cnt);
    hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, 0, (const uint4 *) B.Mo, (const uint4 *) B.Mr, B.n / 2 * 2, B.cnt);
    CK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    CK(hipMemcpy(&bad, B.cnt, 8, hipMemcpyDeviceToHost));
    uint32_t s[24], sr[24];
    CK(hipMemcpy(s, B.sums, 96, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sr, B.sums_ref, 96, hipMemcpyDeviceToHost));
    const bool sums_ok = memcmp(s, sr, p1 ? 96 : 64) == 0;
    const double ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), 0, 0, B.V, B.M, B.Vo, B.Mo, Q, r, B.partials); }, iters);
    const double bytes = 96.0 * (double) B.n;
    printf("%-34s blocks %5u  %8.4f ms  %7.1f GB/s  tables %s  sums %s\n", name, blocks, ms, bytes / ms / 1e6, bad ? "DIFFER" : "ok", sums_ok ? "ok" : "DIFFER");
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int log_n = argc > 1 ? atoi(argv[1]) : 24;
    const int iters = argc > 2 ? atoi(argv[2]) : 10;
    bufs B;
    B.n = 1ull << log_n;
    const uint64_t n = B.n;
    const int layout = argc > 3 ? atoi(argv[3]) : 0;          // 0: separate allocations; 1: one allocation, tables back to back (the product's scratch); >= 2: skewed by `layout` KB
    if (layout == 0) {
        CK(hipMalloc((void **) &B.V, n * 32)); CK(hipMalloc((void **) &B.M, n * 32));
        CK(hipMalloc((void **) &B.Vo, n * 16)); CK(hipMalloc((void **) &B.Mo, n * 16));
    } else {
        char *base;
        const size_t skew = layout == 1 ? 0 : (size_t) layout * 1024;
        CK(hipMalloc((void **) &base, n * 96 + 4 * skew));
        B.V = (fr_t *) base; B.M = (fr_t *) (base + n * 32 + skew); B.Vo = (fr_t *) (base + n * 64 + 2 * skew); B.Mo = (fr_t *) (base + n * 80 + 3 * skew);
    }
    printf("layout %d: V %p M %p Vo %p Mo %p\n", layout, (void *) B.V, (void *) B.M, (void *) B.Vo, (void *) B.Mo);
    CK(hipMalloc((void **) &B.Vr, n * 16)); CK(hipMalloc((void **) &B.Mr, n * 16));
    CK(hipMalloc((void **) &B.partials, 3 * 32 * 16384)); CK(hipMalloc((void **) &B.sums, 96)); CK(hipMalloc((void **) &B.sums_ref, 96));
    CK(hipMalloc((void **) &B.cnt, 8));
    hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, 0, B.V, n, 0x1111ull);
    hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, 0, B.M, n, 0x7777777ull);
    fr_t r;
    {
        const uint32_t rv[8] = {0x12345678u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u, 0x13579bdfu, 0x2468ace0u, 0xdeadbeefu, 0x2badcafeu};
        for (int i = 0; i < 8; ++i) r.v[i] = rv[i];
    }
    // reference: the round-4 loop with all three sums
    hipLaunchKernelGGL(k_old<false>, dim3(768), dim3(BLOCK), 0, 0, B.V, B.M, B.Vr, B.Mr, n / 4, r, B.partials);
    hipLaunchKernelGGL(k_sum, dim3(1), dim3(64), 0, 0, B.sums_ref, B.partials, 768u);
    CK(hipDeviceSynchronize());
    printf("tables 2 x 2^%d entries, %.2f GB algorithmic per launch\n", log_n, 96.0 * n / 1e9);

    {
        const double ms = time_it([&] { hipLaunchKernelGGL(k_stream21, dim3(4096), dim3(BLOCK), 0, 0, (const uint4 *) B.V, (const uint4 *) B.M, (uint4 *) B.Vo, n * 2 / 2); }, iters);
        printf("%-34s              %8.4f ms  %7.1f GB/s (read 2, write 1: n*64 B read, n*16.. scaled)\n", "stream 2:1 (no arithmetic)", ms, (n * 32.0 * 2 / 2 * 3) / ms / 1e6);
    }
    run_variant("old skip_p1", k_old<true>, B, 768, r, false, iters);
    run_variant("old all sums", k_old<false>, B, 768, r, true, iters);
    const uint32_t grids[] = {1024};
    for (uint32_t g : grids) {
        run_variant("new direct/direct w4", k_new<RS_LOAD_DIRECT, RS_STORE_DIRECT, false, 4>, B, g, r, false, iters);
        run_variant("new dma/direct w4 nt-loads", k_new<RS_LOAD_DMA, RS_STORE_DIRECT, false, 4, 1>, B, g, r, false, iters);
    }
    {   // the product kernel itself (kernels.cuh: k_round_quad2<RQ_FOLD>) on the same tables
        round2_args A;
        memset(&A, 0, sizeof(A));
        A.Vin[0] = B.V; A.Min[0] = B.M; A.Vout[0] = B.Vo; A.Mout[0] = B.Mo;
        A.n[0] = n; A.nl[0] = n; A.blocks[0] = 1024; A.r = r; A.skip_p1 = 1;
        CK(hipMalloc((void **) &A.partials, 4096 * 4 * 32));
        CK(hipMalloc((void **) &A.counter, 64)); CK(hipMemset(A.counter, 0, 64));
        CK(hipMalloc((void **) &A.slot, sizeof(host_slot)));
        unsigned long long seq = 0;
        const double ms = time_it([&] { A.seq = ++seq; hipLaunchKernelGGL(k_round_quad2<RQ_FOLD>, dim3(1024), dim3(BLOCK), 0, 0, A); }, iters);
        printf("%-34s blocks %5u  %8.4f ms  %7.1f GB/s\n", "product k_round_quad2<RQ_FOLD>", 1024u, ms, 96.0 * n / ms / 1e6);
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        const double ms2 = time_it([&] { A.seq = ++seq; hipLaunchKernelGGL(k_round_quad2<RQ_FOLD>, dim3(1024), dim3(BLOCK), 0, st, A); }, iters, st);
        printf("%-34s blocks %5u  %8.4f ms  %7.1f GB/s\n", "product, non-blocking stream", 1024u, ms2, 96.0 * n / ms2 / 1e6);
        host_slot *hs, *ds;
        CK(hipHostMalloc((void **) &hs, sizeof(host_slot), hipHostMallocMapped | hipHostMallocCoherent));
        CK(hipHostGetDevicePointer((void **) &ds, hs, 0));
        host_slot *keep = A.slot;
        A.slot = ds;
        const double ms3 = time_it([&] { A.seq = ++seq; hipLaunchKernelGGL(k_round_quad2<RQ_FOLD>, dim3(1024), dim3(BLOCK), 0, st, A); }, iters, st);
        printf("%-34s blocks %5u  %8.4f ms  %7.1f GB/s\n", "product, stream + host slot", 1024u, ms3, 96.0 * n / ms3 / 1e6);
        A.slot = keep;
        // eq-table-like data: V = eq(r, .) built by doubling on the device is what the library's bench uses; here: M = V (maximally correlated operands)
        CK(hipMemcpy(B.M, B.V, n * 32, hipMemcpyDeviceToDevice));
        const double ms4 = time_it([&] { A.seq = ++seq; hipLaunchKernelGGL(k_round_quad2<RQ_FOLD>, dim3(1024), dim3(BLOCK), 0, st, A); }, iters, st);
        printf("%-34s blocks %5u  %8.4f ms  %7.1f GB/s\n", "product, M == V", 1024u, ms4, 96.0 * n / ms4 / 1e6);
        uint32_t sv[24], sr[24];
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(sv, A.slot, 96, hipMemcpyDeviceToHost));
        CK(hipMemcpy(sr, B.sums_ref, 96, hipMemcpyDeviceToHost));
        printf("product sums %s\n", memcmp(sv, sr, 64) == 0 ? "ok" : "DIFFER");
    }
    return 0;
}

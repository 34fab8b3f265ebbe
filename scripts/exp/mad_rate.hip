// experiment: issue cost of v_mad_u64_u32 alone against the (mad, s_nop 1, v_addc) triple of the carry-tracking column accumulator
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
template <int MODE>
__global__ void k(u64 *out, unsigned a0, unsigned b0, int iters) {
    unsigned a = a0 + threadIdx.x, b = b0 + blockIdx.x;
    u64 acc = threadIdx.x, acc2 = blockIdx.x;
    unsigned ovf = 0, ovf2 = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k2 = 0; k2 < 64; ++k2) {
            if (MODE == 0) {            // one dependent chain, no carry tracking
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
            } else if (MODE == 1) {     // mad + wait states + carry count (the current MAC)
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(ovf) : "v"(a), "v"(b) : "vcc");
            } else {                    // two independent chains without carry tracking
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1" : "+v"(acc), "+v"(acc2) : "v"(a), "v"(b) : "vcc");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + acc2 + ovf + ovf2;
}
template <int MODE>
static void run(const char *what, int macs_per_step) {
    u64 *d;
    const int blocks = 256 * 4 * 8, iters = 2000;        // 8 waves per SIMD
    hipMalloc((void **) &d, (size_t) blocks * 64 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 3u, 5u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 3u, 5u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double macs = (double) blocks * 64 * iters * 64 * macs_per_step;
    printf("%s: %.1f T MAC/s (lane MACs), %.2f SIMD cycles per wave-MAC at 2.4 GHz\n", what, macs / ms / 1e9,
           1024.0 * 2.4e9 / (macs / 64 / (ms * 1e-3)));
    hipFree(d);
}
int main() {
    run<0>("mad only, one chain          ", 1);
    run<1>("mad + s_nop 1 + addc         ", 1);
    run<2>("two independent mads per step", 2);
    return 0;
}

// experiment: interactive loop with (a) one launch per round after the challenge is known, (b) the next round's kernel launched
// ahead and fed through a mapped host word. Kernel body = ~8 us of spinning; the host "verifier" does nothing.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef unsigned long long u64;
__global__ void k_round(volatile u64 *chal, u64 expect, volatile u64 *slot, u64 i, u64 ticks) {
    if (threadIdx.x == 0) {
        if (chal) {
            const u64 t0 = wall_clock64();
            while (__hip_atomic_load((u64 *) chal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != expect)
                if (wall_clock64() - t0 > 200000000ull) return;
        }
        const u64 t1 = wall_clock64();
        while (wall_clock64() - t1 < ticks) {}
        __hip_atomic_store((u64 *) slot, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int main() {
    u64 *h_c, *d_c, *h_s, *d_s;
    hipHostMalloc((void **) &h_c, 64, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostMalloc((void **) &h_s, 64, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostGetDevicePointer((void **) &d_c, h_c, 0);
    hipHostGetDevicePointer((void **) &d_s, h_s, 0);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int n = 3000;
    const u64 ticks = 800;                 // 8 us
    *h_c = 0; *h_s = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= n; ++i) {         // (a)
        hipLaunchKernelGGL(k_round, dim3(1), dim3(256), 0, s, (volatile u64 *) nullptr, 0ull, d_s, (u64) i, ticks);
        while (*(volatile u64 *) h_s != (u64) i) {}
    }
    double a = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / n;
    hipStreamSynchronize(s);
    *h_s = 0;
    hipLaunchKernelGGL(k_round, dim3(1), dim3(256), 0, s, d_c, 1ull, d_s, 1ull, ticks);      // round 1 ahead
    t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= n; ++i) {         // (b)
        __atomic_store_n(h_c, (u64) i, __ATOMIC_RELEASE);                                     // publish challenge i
        if (i < n) hipLaunchKernelGGL(k_round, dim3(1), dim3(256), 0, s, d_c, (u64) (i + 1), d_s, (u64) (i + 1), ticks);
        while (*(volatile u64 *) h_s != (u64) i) {}
    }
    double b = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / n;
    hipStreamSynchronize(s);
    printf("8 us kernel body: launch per round %.2f us/round, launched ahead %.2f us/round\n", a, b);
    return 0;
}

// experiment: can the host write straight into device memory (fine-grained allocation, large BAR) and how long is a
// host -> device -> host ping-pong through it, compared with the GPU polling a mapped host word?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <csignal>
#include <csetjmp>
static jmp_buf jb;
static void on_segv(int) { longjmp(jb, 1); }
__global__ void k_pong(volatile unsigned long long *flag, volatile unsigned long long *slot, int rounds, int sys) {
    for (int i = 1; i <= rounds; ++i) {
        unsigned long long t0 = wall_clock64();
        for (;;) {
            unsigned long long v = sys ? __hip_atomic_load((unsigned long long *) flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                       : __hip_atomic_load((unsigned long long *) flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v == (unsigned long long) i) break;
            if (wall_clock64() - t0 > 200000000ull) return;
        }
        __hip_atomic_store((unsigned long long *) slot, (unsigned long long) i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static double run(volatile unsigned long long *host_view_flag, unsigned long long *dev_flag, const char *what) {
    unsigned long long *h_slot, *d_slot;
    hipHostMalloc((void **) &h_slot, 64, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostGetDevicePointer((void **) &d_slot, h_slot, 0);
    *h_slot = 0;
    const int rounds = 2000;
    hipLaunchKernelGGL(k_pong, dim3(1), dim3(64), 0, 0, dev_flag, d_slot, rounds, 1);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= rounds; ++i) {
        *host_view_flag = i;
        while (*(volatile unsigned long long *) h_slot != (unsigned long long) i) {}
    }
    double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / rounds;
    hipDeviceSynchronize();
    printf("%s: %.2f us per host->device->host ping-pong\n", what, us);
    hipHostFree(h_slot);
    return us;
}
int main() {
    // (1) flag in mapped host memory, GPU polls over PCIe
    unsigned long long *h_flag, *d_flag;
    hipHostMalloc((void **) &h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostGetDevicePointer((void **) &d_flag, h_flag, 0);
    *h_flag = 0;
    run(h_flag, d_flag, "flag in host memory (GPU reads over PCIe)");
    // (2) flag in fine-grained DEVICE memory, host writes through the BAR
    unsigned long long *f = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **) &f, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s ptr %p\n", hipGetErrorString(e), (void *) f);
    if (e == hipSuccess) {
        hipMemset(f, 0, 4096);
        hipDeviceSynchronize();
        signal(SIGSEGV, on_segv);
        signal(SIGBUS, on_segv);
        if (setjmp(jb) == 0) {
            volatile unsigned long long probe = *(volatile unsigned long long *) f;     // host load from device memory
            (void) probe;
            *(volatile unsigned long long *) f = 0;
            printf("host can load/store the device allocation\n");
            run(f, f, "flag in fine-grained device memory (host writes through the BAR)");
        } else printf("host access to the device allocation faults: no BAR mapping\n");
    }
    return 0;
}

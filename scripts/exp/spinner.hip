// Experiment: does a resident kernel on a high-priority stream of its own (a hardware queue that never finishes its packet) cost the
// launch-per-round sessions of the same process anything?  build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC spinner.hip -o _spinner.so
#include <hip/hip_runtime.h>
#include <cstdio>
static volatile unsigned *h_flag = nullptr;
static unsigned *d_flag = nullptr;
static hipStream_t st = nullptr;
__global__ void __launch_bounds__(1024) k_spin(const unsigned *flag) {
    __shared__ unsigned stop;
    for (;;) {
        if (threadIdx.x == 0) {
            unsigned v;
            do {
                v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __builtin_amdgcn_s_sleep(16);
            } while (v == 0);
            stop = v;
        }
        __syncthreads();
        if (stop) return;
    }
}
extern "C" int spinner_start(int blocks, int high_priority) {
    if (hipHostMalloc((void **) &h_flag, 64, hipHostMallocMapped) != hipSuccess) return 1;
    *h_flag = 0;
    if (hipHostGetDevicePointer((void **) &d_flag, (void *) h_flag, 0) != hipSuccess) return 2;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, high_priority ? hi : lo) != hipSuccess) return 3;
    hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(1024), 0, st, d_flag);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
extern "C" int spinner_stop() {
    if (!h_flag) return 0;
    *h_flag = 1;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : 5;
}

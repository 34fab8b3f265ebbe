// experiment: GPU-side gap between the end of one kernel and the start of the next in an in-order stream (device timestamps),
// with the queue kept full (each kernel spins ~12 us, the host enqueues faster than that)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void k_spin(unsigned long long *ts, int i, unsigned long long ticks, volatile unsigned long long *host_word) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        ts[2 * i] = t0;
        while (wall_clock64() - t0 < ticks) {}
        if (host_word) __hip_atomic_store((unsigned long long *) host_word, (unsigned long long) i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ts[2 * i + 1] = wall_clock64();
    }
}
static void run(hipStream_t s, const char *what, unsigned long long *d_ts, unsigned long long *host_dev) {
    const int n = 600;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, s, d_ts, i, 1200ull, host_dev);
    hipStreamSynchronize(s);
    std::vector<unsigned long long> ts(2 * n);
    hipMemcpy(ts.data(), d_ts, ts.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> gaps;
    for (int i = 100; i + 1 < n; ++i) gaps.push_back((double) (ts[2 * (i + 1)] - ts[2 * i + 1]) / 100.0);   // 100 MHz -> us
    std::sort(gaps.begin(), gaps.end());
    printf("%s: end->start gap median %.2f us, p10 %.2f, p90 %.2f (kernel body %.1f us)\n", what, gaps[gaps.size() / 2], gaps[gaps.size() / 10],
           gaps[gaps.size() * 9 / 10], (double) (ts[2 * 200 + 1] - ts[2 * 200]) / 100.0);
}
int main() {
    unsigned long long *d_ts, *h_w, *d_w;
    hipMalloc((void **) &d_ts, 2 * 600 * 8);
    hipHostMalloc((void **) &h_w, 64, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostGetDevicePointer((void **) &d_w, h_w, 0);
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    run(s1, "blocking stream, device memory only   ", d_ts, nullptr);
    run(s2, "non-blocking stream, device memory only", d_ts, nullptr);
    run(s2, "non-blocking stream, writes a host word", d_ts, d_w);
    return 0;
}

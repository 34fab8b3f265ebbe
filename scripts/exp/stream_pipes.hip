// experiment: do S concurrent streams progress at the same rate? Each stream gets the same chain of M kernels (every kernel: `blocks` workgroups spinning ~`us`
// microseconds), all streams are filled from one host thread round-robin, and the wall time of every stream's chain is measured with events on that stream.
// Streams that share a hardware queue / pipe with another busy stream show up as the slow ones.  build: hipcc --offload-arch=gfx950 -O2 stream_pipes.hip
// usage: stream_pipes [streams] [kernels per stream] [blocks] [us]      (GPU_MAX_HW_QUEUES from the environment)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_spin(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}
int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 6, M = argc > 2 ? atoi(argv[2]) : 400, blocks = argc > 3 ? atoi(argv[3]) : 256;
    const unsigned long long ticks = 100ull * (argc > 4 ? atoi(argv[4]) : 100);        // wall_clock64: 100 MHz
    std::vector<hipStream_t> st(S);
    std::vector<hipEvent_t> e0(S), e1(S);
    for (int s = 0; s < S; ++s) { hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking); hipEventCreate(&e0[s]); hipEventCreate(&e1[s]); }
    for (int rep = 0; rep < 2; ++rep) {
        for (int s = 0; s < S; ++s) hipEventRecord(e0[s], st[s]);
        for (int i = 0; i < M; ++i)
            for (int s = 0; s < S; ++s) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st[s], ticks);
        for (int s = 0; s < S; ++s) hipEventRecord(e1[s], st[s]);
        hipDeviceSynchronize();
        if (rep == 0) continue;
        printf("%d streams x %d kernels of %d blocks x %llu us (GPU_MAX_HW_QUEUES=%s): per-stream ms:", S, M, blocks, ticks / 100, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "default");
        for (int s = 0; s < S; ++s) { float ms; hipEventElapsedTime(&ms, e0[s], e1[s]); printf(" %.1f", ms); }
        printf("\n");
    }
    return 0;
}

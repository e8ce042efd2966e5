// Which HIP streams block each other at the dispatcher?  Stream 0 runs a grid of far more workgroups than the chip holds (its dispatch stays
// at the head of its hardware queue for the whole kernel, like K1's); every other stream in turn runs a chain of tiny kernels meanwhile.
// Build: hipcc --offload-arch=gfx950 -O3 tools/queue_pipes.hip -o tools/queue_pipes      Run: GPU_MAX_HW_QUEUES=16 tools/queue_pipes [streams=16]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
__global__ __launch_bounds__(256) void big(float* p, int spin) { float x = p[threadIdx.x]; for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f; if (x == 12345.f) p[0] = x; }
__global__ void tiny(float* p) { if (p[0] == 12345.f) p[1] = 1.f; }
int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 16;
    float* p; hipMalloc(&p, 1 << 20); hipMemset(p, 0, 1 << 20);
    // argv[2]: order of FIRST USE: 0 = as created, 1 = reversed;  argv[3]: index of a stream created with the highest priority (-1: none)
    const int rev = argc > 2 ? atoi(argv[2]) : 0, prio = argc > 3 ? atoi(argv[3]) : -1;
    std::vector<hipStream_t> st(N);
    int least = 0, greatest = 0; hipDeviceGetStreamPriorityRange(&least, &greatest);
    for (int i = 0; i < N; ++i) { if (i == prio) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, greatest); else hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); }
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[rev ? N - 1 - i : i], p);
    hipDeviceSynchronize();
    printf("streams %d, first use %s, priority stream %d\n", N, rev ? "reversed" : "as created", prio);
    for (int a = 0; a < 1; ++a) {     // the blocking stream: 0
        for (int j = 0; j < N; ++j) {
            if (j == a) continue;
            hipLaunchKernelGGL(big, dim3(400000), dim3(256), 0, st[a], p, 2000);   // ~60 ms of full chip
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 100; ++k) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st[j], p);
            hipStreamSynchronize(st[j]);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            const auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(st[a]);
            const double rest = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
            if (ms > 3.0) printf("  blocked: stream %2d (%.1f ms)\n", j, ms);
        }
    }
    return 0;
}

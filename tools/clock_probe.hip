// The shader clock the chip actually runs at while something else loads it: ONE wave spins and records (s_memtime = shader cycles,
// s_memrealtime = 100 MHz wall clock) pairs every ~50 us for `seconds`; the host prints the clock per 10 ms window.  Run it in the
// background beside the benchmark (another process: its own queue): tools/profile_round.sh.
// Build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <vector>
__global__ void probe(unsigned long long* out, int n, unsigned long long wall_step)
{
    if (threadIdx.x) return;
    unsigned long long next = wall_clock64() + wall_step;
    for (int i = 0; i < n; ++i) {
        while (wall_clock64() < next) __builtin_amdgcn_s_sleep(8);
        out[2 * i] = wall_clock64();
        out[2 * i + 1] = clock64();
        next += wall_step;
    }
}
int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const unsigned long long step = 5000;               // 50 us at 100 MHz
    const int n = (int)(seconds * 1e8 / step);
    unsigned long long* d;
    if (hipMalloc(&d, (size_t)n * 16) != hipSuccess) return 1;
    timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    printf("# host epoch at launch %.3f\n", ts.tv_sec + ts.tv_nsec * 1e-9);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n, step);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::vector<unsigned long long> h((size_t)n * 2);
    if (hipMemcpy(h.data(), d, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    const int win = 200;                                // 200 samples = 10 ms
    printf("# t_ms  shader_clock_MHz (s_memtime ticks per 100 MHz wall tick x 100, per 10 ms window)\n");
    for (int i = 0; i + win < n; i += win) {
        const double dw = (double)(h[2 * (i + win)] - h[2 * i]), dc = (double)(h[2 * (i + win) + 1] - h[2 * i + 1]);
        printf("%.0f\t%.0f\n", (h[2 * i] - h[0]) / 1e5, dc / dw * 100.0);
    }
    return 0;
}

// How fast does a K5 wave serve itself (nf_serve_limit, m17_wave_kernel.hpp) — alone on its SIMD and in a crowd of its own kind?
// Build: hipcc --offload-arch=gfx950 -I m17-cxx-demod_amd/include -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -disable-machine-licm tools/serve_bench.hip -o tools/serve_bench
#include "../m17-cxx-demod_amd/csrc/m17_wave_kernel.hpp"
#include <cstdio>
using namespace m17;
__global__ __launch_bounds__(256) void serve_kernel(const float* y, float* h, uint32_t N, uint32_t stretch, size_t pitch)
{
    __shared__ __attribute__((aligned(16))) float Bs[4][512];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t row = (size_t)blockIdx.x * 4 + wave;
    const float* yr = y + row * pitch;
    float* hr = h + row * pitch + 4;
    float h0 = 0.f, h1 = 0.f, h2 = 0.f;
    for (uint32_t t = 0; t < N; t += stretch) {
        const Hist3 r = nf_serve_limit(yr, hr, as_lds(&Bs[wave][0]), t, min(N, t + stretch), h0, h1, h2);
        h0 = r.h0; h1 = r.h1; h2 = r.h2;
    }
    if ((threadIdx.x & 63) == 0) hr[-1] = h0 + h1 + h2;
}
int main()
{
    const uint32_t N = 96000; const size_t pitch = N + 64; const int maxb = 1024;
    float *y, *h; hipMalloc(&y, pitch * maxb * 4 * sizeof(float)); hipMalloc(&h, pitch * maxb * 4 * sizeof(float));
    hipMemset(y, 0, pitch * maxb * 4 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (uint32_t stretch : {960u, 9600u, 96000u})
        for (int blocks : {1, 256, 512, 1024}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a); hipLaunchKernelGGL(serve_kernel, dim3(blocks), dim3(256), 0, 0, y, h, N, stretch, pitch); hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("stretch %6u  blocks %4d (%4d waves): %.3f ms = %.1f ns per sample\n", stretch, blocks, blocks * 4, best, best * 1e6 / N);
        }
    return 0;
}

// The limit recurrence on a lone wave: which form is fastest?  (companion of serve_bench.hip; findings in NOTES.md 4.12)
#include "../m17-cxx-demod_amd/csrc/m17_wave_kernel.hpp"
#include <cstdio>
using namespace m17;
// FORM 0: packed multiply (a1 h, a2 h), 1: two plain multiplies (second off the chain), LDS: inputs read from / outputs written to LDS (read-ahead of two)
template <bool MASK16, int FORM, bool LDS>
__global__ __launch_bounds__(256) void rec_kernel(float* out, uint32_t N)
{
    __shared__ __attribute__((aligned(16))) float Bs[4][256];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t l = threadIdx.x & 63u;
    M17_LDS float* B = as_lds(&Bs[wave][0]);
    for (int j = 0; j < 4; ++j) B[l + 64 * j] = (float)(l + j);
    __syncthreads();
    float h0 = out[threadIdx.x], h1 = 0.f, h2 = 0.f, m2 = 0.f;
    auto step = [&](float x) {
        float hn;
        if (FORM == 0) hn = iir_advance_pk(fabsf(x), h0, m2);
        else { float q = IirCoef::a2 * h1; asm("" : "+v"(q)); hn = fabsf(x) - IirCoef::a1 * h0; hn = hn - q; }
        h2 = h1; h1 = h0; h0 = hn;
        return hn;
    };
    if (!MASK16 || l < 16u) {
        for (uint32_t b = 0; b < N; b += 256) {
            if (LDS) {
                const M17_LDS m17_v4f* B4 = reinterpret_cast<const M17_LDS m17_v4f*>(B);
                M17_LDS m17_v4f* O4 = reinterpret_cast<M17_LDS m17_v4f*>(B);
                m17_v4f c0 = B4[0], c1 = B4[1];
                for (uint32_t i = 0; i < 256; i += 8) {
                    const m17_v4f v0 = c0, v1 = c1;
                    c0 = B4[min(i / 4u + 2u, 62u)]; c1 = B4[min(i / 4u + 3u, 63u)];
                    __builtin_amdgcn_sched_barrier(0);
                    m17_v4f o; o.x = step(v0.x); o.y = step(v0.y); o.z = step(v0.z); o.w = step(v0.w); O4[i / 4u] = o;
                    o.x = step(v1.x); o.y = step(v1.y); o.z = step(v1.z); o.w = step(v1.w); O4[i / 4u + 1u] = o;
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                float x = h2 + 1.f;
#pragma unroll 8
                for (uint32_t i = 0; i < 256; ++i) step(x);
            }
        }
    }
    out[threadIdx.x] = h0 + h1 + h2 + m2;
}
template <typename K> void run(const char* name, K k, float* f)
{
    const uint32_t N = 96000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("%-34s", name);
    for (int blocks : {1, 1024}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) { hipEventRecord(a); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, f, N); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("  %4d waves: %5.1f ns/sample", blocks * 4, best * 1e6 / N);
    }
    printf("\n");
}
int main()
{
    float* f; hipMalloc(&f, 1 << 22); hipMemset(f, 0, 1 << 22);
    run("64 lanes packed registers", rec_kernel<false, 0, false>, f);
    run("16 lanes packed registers", rec_kernel<true, 0, false>, f);
    run("64 lanes plain  registers", rec_kernel<false, 1, false>, f);
    run("16 lanes plain  registers", rec_kernel<true, 1, false>, f);
    run("64 lanes packed LDS", rec_kernel<false, 0, true>, f);
    run("16 lanes packed LDS", rec_kernel<true, 0, true>, f);
    run("64 lanes plain  LDS", rec_kernel<false, 1, true>, f);
    run("16 lanes plain  LDS", rec_kernel<true, 1, true>, f);
    return 0;
}

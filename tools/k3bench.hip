// Micro-benchmarks behind the K3 (sliding-DFT) design: dependent-op latencies of a lone wave and the recurrence with
// and without its loads.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I m17-cxx-demod_amd/csrc tools/k3bench.hip -o tools/k3bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "m17_frontend_kernels.hpp"
using namespace m17;
__global__ void k_dep_add(float* out, int n, float a) { float x = out[threadIdx.x];
#pragma unroll 16
    for (int i = 0; i < n; ++i) x = x + a; out[threadIdx.x] = x; }
__global__ void k_dep_pkmul(float* out, int n, float a) { v2f x = {out[threadIdx.x], 1.f}; const v2f c = {a, a};
#pragma unroll 16
    for (int i = 0; i < n; ++i) x = x * c; out[threadIdx.x] = x.x + x.y; }
__global__ void k_indep_add(float* out, int n, float a) { float x0 = out[threadIdx.x], x1 = x0, x2 = x0, x3 = x0, x4 = x0, x5 = x0, x6 = x0, x7 = x0;
    for (int i = 0; i < n; i += 8) { x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a; } out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; }
// recurrence only: deltas from LDS (written once), no global traffic in the loop
template <int MODE> __global__ __launch_bounds__(64) void k_rec(float* out, int nblocks, float cr, float ci)
{
    __shared__ __attribute__((aligned(16))) float dl[64];
    dl[threadIdx.x] = 0.001f * threadIdx.x;
    __syncthreads();
    DcdLane s; s.X = v2f{0.f, 0.f}; s.cc = v2f{cr, ci}; s.cs = v2f{-ci, cr}; s.a01 = s.a23 = s.a45 = v2f{0.f, 0.f};
    for (int b = 0; b < nblocks; ++b) {
        float d[64];
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float4 v = *reinterpret_cast<const float4*>(dl + 4 * q); d[4*q] = v.x; d[4*q+1] = v.y; d[4*q+2] = v.z; d[4*q+3] = v.w; }
#pragma unroll
        for (int q = 0; q < 64; ++q) dcd_step(s, d[q]);
        __builtin_amdgcn_wave_barrier();
    }
    out[threadIdx.x + 64 * blockIdx.x] = s.a01.x + s.a45.y + s.X.x;
}
template <typename F> float timeit(F f) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); f(); hipDeviceSynchronize(); hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms; }
int main()
{
    float* f; hipMalloc(&f, 1 << 22); hipMemset(f, 0, 1 << 22);
    const int n = 480000;
    for (int blocks : {1, 1024, 4096}) {
        float ta = timeit([&] { hipLaunchKernelGGL(k_dep_add, dim3(blocks), dim3(64), 0, 0, f, n, 1.0001f); });
        float tp = timeit([&] { hipLaunchKernelGGL(k_dep_pkmul, dim3(blocks), dim3(64), 0, 0, f, n, 1.0001f); });
        float ti = timeit([&] { hipLaunchKernelGGL(k_indep_add, dim3(blocks), dim3(64), 0, 0, f, n, 1.0001f); });
        float t0 = timeit([&] { hipLaunchKernelGGL(k_rec<0>, dim3(blocks), dim3(64), 0, 0, f, n / 64, 0.95f, 0.31f); });
        float t1 = timeit([&] { hipLaunchKernelGGL(k_rec<1>, dim3(blocks), dim3(64), 0, 0, f, n / 64, 0.95f, 0.31f); });
        printf("blocks=%4d | dep v_add %.2f ns | dep v_pk_mul %.2f ns | indep v_add %.2f ns | recurrence (9 VALU/sample) %.2f ns/sample (%.2f ms) | again %.2f ns/sample (%.2f ms)\n",
               blocks, ta * 1e6 / n, tp * 1e6 / n, ti * 1e6 / n, t0 * 1e6 / n, t0, t1 * 1e6 / n, t1);
    }
    // the product kernel on random input
    {
        const uint32_t T = 480000; const size_t xpitch = ((size_t)XPRE + T + 8 + 7) / 8 * 8; const uint32_t ticks_cap = T / TICK + 2;
        for (uint32_t C : {4u, 1024u, 4096u}) {
            int16_t* x; DcdState* st; float* tab;
            hipMalloc(&x, C * xpitch * 2); hipMalloc(&st, C * sizeof(DcdState)); hipMalloc(&tab, (size_t)C * ticks_cap * 48);
            std::vector<int16_t> h(C * xpitch); uint32_t r = 12345; for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (int16_t)((int)(r >> 16) % 20000 - 10000); }
            hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemset(st, 0, C * sizeof(DcdState));
            DcdCoef k{0.95f, -0.31f, 0.89f, -0.45f};
            float t = timeit([&] { hipLaunchKernelGGL(dcd_kernel, dim3((C + DCD_CPW - 1) / DCD_CPW), dim3(64), 0, 0, x, xpitch, st, tab, ticks_cap, C, T, 0ull, k, 0u); });
            printf("dcd_kernel C=%u: %.2f ms (%.2f ns/sample)\n", C, t, t * 1e6 / T);
            hipFree(x); hipFree(st); hipFree(tab);
        }
    }
    return 0;
}

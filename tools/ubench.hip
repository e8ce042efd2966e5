// Micro-benchmarks for latency-bound single-wave kernels on gfx950: dependent fp32 chain, dependent fp64 chain,
// LDS read->use chain, global pointer-chase, s_memrealtime cost.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_chain32(float* out, int n, float a) { float x = out[threadIdx.x]; for (int i = 0; i < n; ++i) { x = x * a; x = x + a; } out[threadIdx.x + blockIdx.x * blockDim.x] = x; }
__global__ void k_chain64(double* out, int n, double a) { double x = out[threadIdx.x]; for (int i = 0; i < n; ++i) { x = x * a; x = x + a; } out[threadIdx.x + blockIdx.x * blockDim.x] = x; }
__global__ void k_lds(int* out, int n) { __shared__ int l[1024]; for (int i = threadIdx.x; i < 1024; i += blockDim.x) l[i] = (i * 7 + 1) & 1023; __syncthreads(); int p = threadIdx.x; for (int i = 0; i < n; ++i) p = l[p]; out[threadIdx.x + blockIdx.x * blockDim.x] = p; }
__global__ void k_chase(const int* tab, int* out, int n) { int p = threadIdx.x + blockIdx.x * 64; for (int i = 0; i < n; ++i) p = tab[p]; out[threadIdx.x + blockIdx.x * blockDim.x] = p; }
__global__ void k_clock(unsigned long long* out, int n) { unsigned long long a = 0; for (int i = 0; i < n; ++i) a += wall_clock64() & 1; out[threadIdx.x + blockIdx.x * blockDim.x] = a + clock64(); }
template <typename F> float timeit(F f) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); f(); hipDeviceSynchronize(); hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms; }
int main() {
    float* f; double* d; int* i1; int* tab; unsigned long long* u;
    const int N = 1 << 24;
    hipMalloc(&f, 1 << 22); hipMalloc(&d, 1 << 23); hipMalloc(&i1, 1 << 22); hipMalloc(&tab, N * 4); hipMalloc(&u, 1 << 23);
    hipMemset(f, 0, 1 << 22); hipMemset(d, 0, 1 << 23);
    std::vector<int> h(N); for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i * 1048583 + 12345) % N);
    hipMemcpy(tab, h.data(), N * 4, hipMemcpyHostToDevice);
    const int n = 200000;
    for (int blocks : {1, 64, 256, 1024}) for (int threads : {1, 64}) {
        float t32 = timeit([&] { hipLaunchKernelGGL(k_chain32, dim3(blocks), dim3(threads), 0, 0, f, n, 1.0001f); });
        float t64 = timeit([&] { hipLaunchKernelGGL(k_chain64, dim3(blocks), dim3(threads), 0, 0, d, n, 1.0001); });
        float tl = timeit([&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(threads), 0, 0, i1, n); });
        float tg = timeit([&] { hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(threads), 0, 0, tab, i1, n / 10); });
        float tc = timeit([&] { hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(threads), 0, 0, u, n / 10); });
        printf("blocks=%4d threads=%2d | dep f32 op %.2f ns | dep f64 op %.2f ns | LDS chase %.1f ns | global chase %.0f ns | wall_clock64 %.0f ns\n", blocks, threads,
               t32 * 1e6 / (2.0 * n), t64 * 1e6 / (2.0 * n), tl * 1e6 / n, tg * 1e6 / (n / 10), tc * 1e6 / (n / 10));
    }
    return 0;
}

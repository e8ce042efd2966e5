// Latency of the sliding-DFT recurrence on a lone wave (MI355X): which instruction mix carries X <- (X + d) * c fastest?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/chain_bench.hip -o tools/chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N 480000
__global__ void k_dep_add(float* out, float a) { float x = out[threadIdx.x];
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = x + a; out[threadIdx.x] = x; }
__global__ void k_dep_mul(float* out, float a) { float x = out[threadIdx.x];
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = x * a; out[threadIdx.x] = x; }
__global__ void k_dep_pkmul(float* out, float a) { v2f x = {out[threadIdx.x], 1.f}; const v2f c = {a, a};
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = x * c; out[threadIdx.x] = x.x + x.y; }
__global__ void k_dep_pkadd(float* out, float a) { v2f x = {out[threadIdx.x], 1.f}; const v2f c = {a, a};
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = x + c; out[threadIdx.x] = x.x + x.y; }
// the recurrence, packed form (what K3 runs): 4 VALU per sample
__global__ void k_rec_packed(float* out, float cr, float ci) { v2f X = {out[threadIdx.x], 0.f}; const v2f cc = {cr, ci}, cs = {-ci, cr}; float d = out[64 + threadIdx.x];
#pragma unroll 16
    for (int i = 0; i < N; ++i) { const float a = X.x + d; const v2f m1 = v2f{a, a} * cc; const v2f m2 = v2f{X.y, X.y} * cs; X = m1 + m2; }
    out[threadIdx.x] = X.x + X.y; }
// scalar form: 7 VALU per sample, dependent depth 3
__global__ void k_rec_scalar(float* out, float cr, float ci) { float xr = out[threadIdx.x], xi = 0.f; float d = out[64 + threadIdx.x];
#pragma unroll 16
    for (int i = 0; i < N; ++i) { const float a = xr + d; const float ac = a * cr, ad = a * ci, bd = xi * ci, bc = xi * cr; xr = ac - bd; xi = ad + bc; }
    out[threadIdx.x] = xr + xi; }
// mixed: scalar add and multiplies on the critical path, packed for the rest
__global__ void k_rec_mixed(float* out, float cr, float ci) { float xr = out[threadIdx.x], xi = 0.f; float d = out[64 + threadIdx.x]; const v2f cs = {-ci, cr};
#pragma unroll 16
    for (int i = 0; i < N; ++i) { const v2f m2 = v2f{xi, xi} * cs; const float a = xr + d; const float ac = a * cr, ad = a * ci; xr = ac + m2.x; xi = ad + m2.y; }
    out[threadIdx.x] = xr + xi; }
template <typename F> float timeit(F f) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); f(); hipDeviceSynchronize(); hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms; }
int main()
{
    float* f; hipMalloc(&f, 1 << 20); hipMemset(f, 0, 1 << 20);
    for (int blocks : {1, 128, 1024}) {
        printf("blocks=%4d:", blocks);
        printf(" dep v_add %.2f", timeit([&] { hipLaunchKernelGGL(k_dep_add, dim3(blocks), dim3(64), 0, 0, f, 1.0001f); }) * 1e6 / N);
        printf(" | dep v_mul %.2f", timeit([&] { hipLaunchKernelGGL(k_dep_mul, dim3(blocks), dim3(64), 0, 0, f, 1.0001f); }) * 1e6 / N);
        printf(" | dep v_pk_mul %.2f", timeit([&] { hipLaunchKernelGGL(k_dep_pkmul, dim3(blocks), dim3(64), 0, 0, f, 1.0001f); }) * 1e6 / N);
        printf(" | dep v_pk_add %.2f", timeit([&] { hipLaunchKernelGGL(k_dep_pkadd, dim3(blocks), dim3(64), 0, 0, f, 1.0001f); }) * 1e6 / N);
        printf(" | recurrence packed %.2f", timeit([&] { hipLaunchKernelGGL(k_rec_packed, dim3(blocks), dim3(64), 0, 0, f, 0.95f, 0.31f); }) * 1e6 / N);
        printf(" | scalar %.2f", timeit([&] { hipLaunchKernelGGL(k_rec_scalar, dim3(blocks), dim3(64), 0, 0, f, 0.95f, 0.31f); }) * 1e6 / N);
        printf(" | mixed %.2f ns/sample\n", timeit([&] { hipLaunchKernelGGL(k_rec_mixed, dim3(blocks), dim3(64), 0, 0, f, 0.95f, 0.31f); }) * 1e6 / N);
    }
    return 0;
}

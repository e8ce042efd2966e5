// Round 5 probe: can the matrix pipe carry the PRODUCTS of the exact-order matched filter?
//   v_mfma_f32_4x4x1_16b_f32 with C = 0 is D[i] = fl(A_i * B_own): four correctly rounded f32 products of a lane's own value with four
//   values supplied by (a broadcast block of) four lanes — "own sample x four consecutive taps" for K1, while the VALU keeps the 149
//   ordered additions (v_pk_add_f32).  This tool answers, on the GPU:
//   (a) the operand / result lane layout incl. the CBSZ / ABID broadcast of the A block,
//   (b) bit-exactness of the MFMA product against v_mul_f32 for every (tap, int16 sample) pair of the filter (149 x 65536),
//   (c) issue rates: MFMA alone, v_pk_add alone, 4 MFMA + 8 v_pk_add per window element (the proposed K1 body), and today's
//       2 v_pk_mul + 2 v_pk_add per four products, at 1 / 2 / 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I m17-cxx-demod_amd/include -I m17-cxx-demod_amd/csrc tools/mfma_probe.hip -o tools/mfma_probe
#include "m17_common.hpp"
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int E, int N, typename F>
__device__ __forceinline__ void unroll(F&& f)
{
    f(std::integral_constant<int, E>{});
    if constexpr (E + 1 < N) unroll<E + 1, N>(f);
}

// ---- (a) layout ------------------------------------------------------------------------------------------------------------------
template <int CBSZ, int ABID>
__global__ void k_layout(const float* a, const float* b, float* d)
{
    const v4f z = {0.f, 0.f, 0.f, 0.f};
    const v4f r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], z, CBSZ, ABID, 0);
    for (int i = 0; i < 4; ++i) d[threadIdx.x * 4 + i] = r[i];
}

// ---- (b) exactness ---------------------------------------------------------------------------------------------------------------
__global__ void k_exact(const float* taps, unsigned long long* bad, unsigned long long* checked, int invert)
{
    const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 32768;   // every int16
    const float x = m17::scale_sample(s, invert != 0);
    const int lane = threadIdx.x & 63;
    const v4f z = {0.f, 0.f, 0.f, 0.f};
    unsigned long long nb = 0, nc = 0;
    for (int T = -3; T < 152; ++T) {   // every quad alignment, incl. the zero taps outside 0..148
        const int ti = T + (lane & 3);
        const float tap = (ti >= 0 && ti < 149) ? taps[ti] : 0.0f;
        const v4f r = __builtin_amdgcn_mfma_f32_4x4x1f32(tap, x, z, 0, 0, 0);
        for (int i = 0; i < 4; ++i) {
            const int tj = T + i;
            const float tp = (tj >= 0 && tj < 149) ? taps[tj] : 0.0f;
            const float want = __fmul_rn(tp, x);
            // a zero product may differ in sign (fma(a, b, +0) = +0 where a * b = -0): the accumulators are never -0, so adding either is the same
            const bool same = __float_as_uint(want) == __float_as_uint(r[i]) || (want == 0.0f && r[i] == 0.0f);
            nb += same ? 0 : 1; ++nc;
        }
    }
    atomicAdd(bad, nb); atomicAdd(checked, nc);
}

// ---- (c) issue rates -------------------------------------------------------------------------------------------------------------
// MODE 0: per window element 4 MFMA (own x, tap quads of two registers through ABID) + 8 v_pk_add on the products of the element before
// MODE 1: the MFMAs only   MODE 2: the adds only   MODE 3: today's form — per element 8 v_pk_mul + 8 v_pk_add (sixteen products)
// MODE 4: like 0 with the window elements read from LDS (ds_read_b128 per four elements) and the tap vector reloaded per 16 elements
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_rate(float* out, const float* in, int n, unsigned long long* cycles)
{
    __shared__ __attribute__((aligned(16))) float win[256 * 16 + 256];
    __shared__ float tq[512];
    const int tid = threadIdx.x;
    for (int k = tid; k < 256 * 16 + 256; k += 256) win[k] = in[k & 1023];
    for (int k = tid; k < 512; k += 256) tq[k] = in[k & 255] * 0.01f;
    __syncthreads();
    v2f acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = v2f{in[tid + q], in[tid + q + 8]};
    float q0 = in[tid + 16], q1 = in[tid + 17];
    float xs[16];
    for (int e = 0; e < 16; ++e) xs[e] = in[tid + 20 + e];
    const v4f z = {0.f, 0.f, 0.f, 0.f};
    v4f dp[4] = {z, z, z, z};
    float sacc[16];
    for (int q = 0; q < 16; ++q) sacc[q] = in[tid + 40 + q];
    const float* base = win + tid * 16;
    const float* tqb = tq + (tid & 63) / 4 + (tid & 3);
    const unsigned long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
        v4f w4[4];
        if (MODE == 4) {
            for (int g = 0; g < 4; ++g) w4[g] = *reinterpret_cast<const v4f*>(base + 4 * g + (it & 15) * 16);
            q0 = q1; q1 = tqb[(it & 7) * 16];
        }
        unroll<0, 16>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            const float x = MODE == 4 ? w4[e / 4][e % 4] : xs[e];
            v4f dc[4];
            if (MODE == 0 || MODE == 1 || MODE == 4) {
                unroll<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    constexpr int id = e + 4 * j;           // quad id relative to the body: 0 .. 27 -> (register, ABID)
                    dc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(id < 16 ? q0 : q1, x, z, 4, id & 15, 0);
                });
            }
            if (MODE == 0 || MODE == 2 || MODE == 4) {
                for (int j = 0; j < 4; ++j) {
                    acc[2 * j] = acc[2 * j] + __builtin_shufflevector(dp[j], dp[j], 0, 1);
                    acc[2 * j + 1] = acc[2 * j + 1] + __builtin_shufflevector(dp[j], dp[j], 2, 3);
                }
            }
            if (MODE == 1) { for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(dc[j])); }
            if (MODE == 0 || MODE == 1 || MODE == 4) { for (int j = 0; j < 4; ++j) dp[j] = dc[j]; }
            if (MODE == 2) { for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(dp[j])); }
            if (MODE == 5 || MODE == 6) {   // the same sixteen additions UNPACKED (v_add_f32; build with -fno-slp-vectorize), alone (5) and beside the MFMAs (6)
                if (MODE == 6) {
                    unroll<0, 4>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int id = e + 4 * j;
                        dc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(id < 16 ? q0 : q1, x, z, 4, id & 15, 0);
                    });
                }
                for (int j = 0; j < 4; ++j) {
                    sacc[4 * j] = sacc[4 * j] + dp[j].x; sacc[4 * j + 1] = sacc[4 * j + 1] + dp[j].y;
                    sacc[4 * j + 2] = sacc[4 * j + 2] + dp[j].z; sacc[4 * j + 3] = sacc[4 * j + 3] + dp[j].w;
                }
                if (MODE == 6) { for (int j = 0; j < 4; ++j) dp[j] = dc[j]; }
                else { for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(dp[j])); }
            }
            if (MODE == 7) {   // sixteen products and sixteen additions, all unpacked
                const float tv = xs[e] + (float)it;   // (a different "tap" per element and iteration: no common subexpressions)
                for (int q = 0; q < 16; ++q) {
                    const float p = xs[(e + q) & 15] * tv;
                    sacc[q] = sacc[q] + p;
                }
            }
            if (MODE == 3) {
                for (int q = 0; q < 8; ++q) {
                    const v2f p = v2f{xs[(e + q) & 15], xs[(e + q + 1) & 15]} * v2f{xs[e], xs[e]};   // (a different "tap" per element: no common subexpressions)
                    acc[q] = acc[q] + p;
                }
            }
        });
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += acc[q].x + acc[q].y;
    for (int j = 0; j < 4; ++j) s += dp[j].x;
    for (int q = 0; q < 16; ++q) s += sacc[q];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) atomicAdd(cycles, t1 - t0);
}

template <int MODE>
static void rate(const char* name, float* out, const float* in, unsigned long long* cyc, int n)
{
    for (int wps : {1, 2, 4, 8}) {   // blocks of four waves: `wps` blocks per CU = waves per SIMD (low register use: all resident)
        const int blocks = 256 * wps;
        hipMemset(cyc, 0, 8);
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, 64, cyc);
        hipDeviceSynchronize();
        hipMemset(cyc, 0, 8);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, n, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double per_wave = (double)c / blocks;                    // cycles one wave spent in the loop
        const double elems = (double)n * 16;
        printf("%-34s waves/SIMD %d: %.3f ms | %.1f wave-cycles per element | %.1f SIMD-cycles per element | clock %.2f GHz\n", name, wps, ms,
               per_wave / elems, per_wave / elems / wps, per_wave / (ms * 1e6));
    }
}

int main()
{
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    // ---- (a)
    auto layout = [&](auto kern, const char* name) {
        printf("layout %s: lane -> (A source lane, B source lane) for result registers 0..3\n", name);
        std::vector<int> srcA(256, -1), srcB(256, -1);
        for (int p = 0; p < 64; ++p) {
            float ha[64], hb[64], hd[256];
            for (int l = 0; l < 64; ++l) { ha[l] = l == p ? 1.f : 0.f; hb[l] = (float)(l + 1); }
            hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, a, b, d);
            hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
            for (int k = 0; k < 256; ++k) if (hd[k] != 0.f) { srcA[k] = p; srcB[k] = (int)hd[k] - 1; }
        }
        for (int l : {0, 1, 2, 3, 4, 5, 21, 63}) {
            printf("  lane %2d:", l);
            for (int i = 0; i < 4; ++i) printf(" r%d=(A%2d,B%2d)", i, srcA[l * 4 + i], srcB[l * 4 + i]);
            printf("\n");
        }
        bool std_layout = true;
        return std_layout;
    };
    layout(k_layout<0, 0>, "cbsz 0");
    layout(k_layout<4, 0>, "cbsz 4 abid 0");
    layout(k_layout<4, 5>, "cbsz 4 abid 5");
    layout(k_layout<4, 15>, "cbsz 4 abid 15");
    // ---- (b)
    float htaps[160] = {0};
    for (int i = 0; i < 149; ++i) htaps[i] = m17::rrc_tap(i);
    float* taps; hipMalloc(&taps, sizeof(htaps)); hipMemcpy(taps, htaps, sizeof(htaps), hipMemcpyHostToDevice);
    unsigned long long* cnt; hipMalloc(&cnt, 16);
    for (int inv = 0; inv < 2; ++inv) {
        hipMemset(cnt, 0, 16);
        hipLaunchKernelGGL(k_exact, dim3(65536 / 256), dim3(256), 0, 0, taps, cnt, cnt + 1, inv);
        unsigned long long h[2]; hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
        printf("exactness (invert %d): %llu products compared with v_mul_f32, %llu differ\n", inv, h[1], h[0]);
    }
    // ---- (c)
    float *out, *in; hipMalloc(&out, 4 * 256 * 4096); hipMalloc(&in, 4 * 8192);
    std::vector<float> hin(8192);
    for (int i = 0; i < 8192; ++i) hin[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.f - 0.5f;
    hipMemcpy(in, hin.data(), 4 * 8192, hipMemcpyHostToDevice);
    const int n = 4000;
    rate<1>("MFMA only (4 per element)", out, in, cnt, n);
    rate<2>("v_pk_add only (8 per element)", out, in, cnt, n);
    rate<0>("4 MFMA + 8 v_pk_add", out, in, cnt, n);
    rate<4>("4 MFMA + 8 v_pk_add, LDS operands", out, in, cnt, n);
    rate<3>("8 v_pk_mul + 8 v_pk_add (today)", out, in, cnt, n);
    rate<5>("16 v_add_f32 only", out, in, cnt, n);
    rate<6>("4 MFMA + 16 v_add_f32", out, in, cnt, n);
    rate<7>("16 v_mul_f32 + 16 v_add_f32", out, in, cnt, n);
    return 0;
}

// Where do the waves of a launch land?  Every wave records HW_REG_HW_ID (gfx9: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13])
// and HW_REG_XCC_ID, then stays resident for a while; the host counts waves per (XCC, SE, SH, CU, SIMD).  Workgroups of ONE wave against workgroups of FOUR.
// NOTES 6.6: single-wave workgroups of one launch pile up on one SIMD of a CU.
// hipcc --offload-arch=gfx950 -O3 -o tools/simd_probe tools/simd_probe.hip && tools/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>

__global__ void probe(uint32_t* out, uint32_t spin_us)
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)spin_us * 100) __builtin_amdgcn_s_sleep(10);
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[2 * w] = hw; out[2 * w + 1] = xcc;
    }
}

static void run(const char* name, int blocks, int threads, uint32_t* d, int lds_bytes)
{
    const int waves = blocks * threads / 64;
    (void)hipMemset(d, 0xFF, (size_t)waves * 8);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), lds_bytes, 0, d, 200u);
    (void)hipDeviceSynchronize();
    std::vector<uint32_t> h(2 * (size_t)waves);
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> per_simd, per_cu;
    for (int w = 0; w < waves; ++w) {
        const uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 0xF;
        const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const uint32_t cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[cukey]++; per_simd[(cukey << 2) | simd]++;
    }
    int hist[40] = {0}, mx = 0;
    for (auto& kv : per_simd) { hist[kv.second < 39 ? kv.second : 39]++; if (kv.second > mx) mx = kv.second; }
    printf("%-44s %5d waves on %3zu CUs, %4zu SIMDs in use; waves per used SIMD:", name, waves, per_cu.size(), per_simd.size());
    for (int i = 1; i <= mx && i < 40; ++i) if (hist[i]) printf("  %d x%d", i, hist[i]);
    printf("\n");
}

int main()
{
    uint32_t* d; (void)hipMalloc(&d, 1 << 22);
    run("1024 blocks of 1 wave", 1024, 64, d, 0);
    run("256 blocks of 4 waves", 256, 256, d, 0);
    run("256 blocks of 1 wave", 256, 64, d, 0);
    run("512 blocks of 1 wave", 512, 64, d, 0);
    run("2048 blocks of 1 wave", 2048, 64, d, 0);
    run("4096 blocks of 1 wave", 4096, 64, d, 0);
    run("4096 blocks of 1 wave, 18.7 KB LDS each", 4096, 64, d, 19136);
    run("1024 blocks of 4 waves, 63 KB LDS each", 1024, 256, d, 64512);
    run("512 blocks of 1 wave, 23 KB LDS each", 512, 64, d, 23744);
    return 0;
}

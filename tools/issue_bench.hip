// What does ONE wave pay per instruction when it carries a dependent chain?  (the limit recurrence of configs[1], limit_pipe_kernel's R wave;
// findings in NOTES.md 5.9).  The recurrence h0 = (|y| - a1 h1) - a2 h2 as one asm statement per tile, so that no compiler-inserted s_nop / v_mov
// sits in the chain: three instructions per sample, the inputs read from LDS two blocks ahead, the outputs written sixteen bytes at a time.
//   V = 0  the bare chain (inputs from registers, nothing written)
//   V = 1  + one independent v_mov per sample          V = 2  + two
//   V = 3  LDS reads only      V = 4  LDS writes only      V = 5  both (the production form)
//   V = 6  V = 5 with s_nop 0 after every chain instruction (is a lone wave's issue slotted?)
// hipcc --offload-arch=gfx950 -O3 -o tools/issue_bench tools/issue_bench.hip && tools/issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CH_STEP(X, O, PLO, PHI, SEL, RAlo, RBlo, RBhi, EXTRA) \
    "v_sub_f32_e64 v86, |v" #X "|, v" #RAlo "\n" EXTRA \
    "v_sub_f32_e32 v" #O ", v86, v" #RBhi "\n" EXTRA \
    "v_pk_mul_f32 v[" #RBlo ":" #RBhi "], v[" #PLO ":" #PHI "], s[20:21] op_sel:[" #SEL ",0] op_sel_hi:[" #SEL ",1]\n" EXTRA
#define CH_FOUR(X0, X1, X2, X3, O0, O1, O2, O3, EXTRA, MOV) \
    CH_STEP(X0, O0, O0, O1, 0, 82, 80, 81, EXTRA) MOV CH_STEP(X1, O1, O0, O1, 1, 80, 82, 83, EXTRA) MOV \
    CH_STEP(X2, O2, O2, O3, 0, 82, 80, 81, EXTRA) MOV CH_STEP(X3, O3, O2, O3, 1, 80, 82, 83, EXTRA) MOV
#define RD(V0, V1, OFF) "ds_read_b128 v[" #V0 ":" #V1 "], v40 offset:" #OFF "\n"
#define WR(V0, V1, OFF) "ds_write_b128 v41, v[" #V0 ":" #V1 "] offset:" #OFF "\n"
#define WAIT "s_waitcnt lgkmcnt(2)\n"
#define NONE ""
// sixteen samples: block A = v[48:55], block B = v[56:63]; outputs v[64:79]
#define CH_16(OFF_RB0, OFF_RB1, OFF_RA0, OFF_RA1, W0, W1, W2, W3, R, W, WT, EXTRA, MOV) \
    WT R(56, 59, OFF_RB0) R(60, 63, OFF_RB1) \
    CH_FOUR(48, 49, 50, 51, 64, 65, 66, 67, EXTRA, MOV) W(64, 67, W0) CH_FOUR(52, 53, 54, 55, 68, 69, 70, 71, EXTRA, MOV) W(68, 71, W1) \
    WT R(48, 51, OFF_RA0) R(52, 55, OFF_RA1) \
    CH_FOUR(56, 57, 58, 59, 72, 73, 74, 75, EXTRA, MOV) W(72, 75, W2) CH_FOUR(60, 61, 62, 63, 76, 77, 78, 79, EXTRA, MOV) W(76, 79, W3)
// the same sixteen with every write one block of four LATE (its registers were finished four samples ago)
#define CH_16D(OFF_RB0, OFF_RB1, OFF_RA0, OFF_RA1, WM1, W0, W1, W2, R, W, WT, EXTRA, MOV) \
    WT R(56, 59, OFF_RB0) R(60, 63, OFF_RB1) \
    CH_FOUR(48, 49, 50, 51, 64, 65, 66, 67, EXTRA, MOV) W(76, 79, WM1) CH_FOUR(52, 53, 54, 55, 68, 69, 70, 71, EXTRA, MOV) W(64, 67, W0) \
    WT R(48, 51, OFF_RA0) R(52, 55, OFF_RA1) \
    CH_FOUR(56, 57, 58, 59, 72, 73, 74, 75, EXTRA, MOV) W(68, 71, W1) CH_FOUR(60, 61, 62, 63, 76, 77, 78, 79, EXTRA, MOV) W(72, 75, W2)
#define WR64(V0, V1, OFF) "ds_write_b64 v41, v[" #V0 ":" #V0 "+1] offset:" #OFF "\n ds_write_b64 v41, v[" #V1 "-1:" #V1 "] offset:" #OFF "+8\n"
#define WR2(V0, V1, OFF) "ds_write2_b64 v41, v[" #V0 ":" #V0 "+1], v[" #V1 "-1:" #V1 "] offset0:" #OFF "/8 offset1:" #OFF "/8+1\n"
#define WAIT4 "s_waitcnt lgkmcnt(4)\n"
// the outputs straight to GLOBAL memory from the chain's lanes (v44 = the lane's byte offset into the buffer at s[24:25]) instead of through LDS
#define WRG(V0, V1, OFF) "global_store_dwordx4 v44, v[" #V0 ":" #V1 "], s[24:25] offset:" #OFF "\n"
#define WAIT0 "s_waitcnt lgkmcnt(0)\n"
#define NO_R(V0, V1, OFF) ""
#define NO_W(V0, V1, OFF) ""
#define CH_TILE(R, W, WT, EXTRA, MOV) \
    "s_mov_b32 s20, 0xbffda16a\n s_mov_b32 s21, 0x3f7b4df5\n s_mov_b32 s22, 8\n" \
    "ds_read_b128 v[48:51], v40\n ds_read_b128 v[52:55], v40 offset:16\n ds_write_b128 v41, v[64:67]\n ds_write_b128 v41, v[64:67]\n" \
    "1:\n" \
    CH_16(32, 48, 64, 80, 16, 32, 48, 64, R, W, WT, EXTRA, MOV) CH_16(96, 112, 128, 144, 80, 96, 112, 128, R, W, WT, EXTRA, MOV) \
    "v_add_u32_e32 v40, 128, v40\n v_add_u32_e32 v41, 128, v41\n s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 1b\n" \
    "s_waitcnt lgkmcnt(0)\n"
#define CH_TILED(R, W, WT, EXTRA, MOV) \
    "s_mov_b32 s20, 0xbffda16a\n s_mov_b32 s21, 0x3f7b4df5\n s_mov_b32 s22, 8\n" \
    "ds_read_b128 v[48:51], v40\n ds_read_b128 v[52:55], v40 offset:16\n ds_write_b128 v41, v[64:67]\n ds_write_b128 v41, v[64:67]\n" \
    "1:\n" \
    CH_16D(32, 48, 64, 80, 0, 16, 32, 48, R, W, WT, EXTRA, MOV) CH_16D(96, 112, 128, 144, 64, 80, 96, 112, R, W, WT, EXTRA, MOV) \
    "v_add_u32_e32 v40, 128, v40\n v_add_u32_e32 v41, 128, v41\n s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 1b\n" \
    "s_waitcnt lgkmcnt(0)\n"
#define CH_TILEG(R, W, WT, EXTRA, MOV) \
    "s_mov_b32 s20, 0xbffda16a\n s_mov_b32 s21, 0x3f7b4df5\n s_mov_b32 s22, 8\n" \
    "ds_read_b128 v[48:51], v40\n ds_read_b128 v[52:55], v40 offset:16\n" \
    "1:\n" \
    CH_16(32, 48, 64, 80, 16, 32, 48, 64, R, W, WT, EXTRA, MOV) CH_16(96, 112, 128, 144, 80, 96, 112, 128, R, W, WT, EXTRA, MOV) \
    "v_add_u32_e32 v40, 128, v40\n v_add_u32_e32 v44, 128, v44\n s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 1b\n" \
    "s_waitcnt lgkmcnt(0)\n"
#define CLOB "v44", "s24", "s25", "v40", "v41", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", \
    "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v86", "v87", "v88", "s20", "s21", "s22", "scc", "memory"

template <int V, int LANES>
__global__ __launch_bounds__(256) void chain_kernel(float* out, uint32_t ntiles)
{
    float* gbuf = out + (1 << 16);   // (V = 12, 13: every tile writes the same 1 KB per lane: 16 lanes x 1100 floats per block)
    __shared__ __attribute__((aligned(16))) float yrow[16][260 + 64];
    __shared__ __attribute__((aligned(16))) float hrow[16][264 + 64];
    const uint32_t l = threadIdx.x & 63u;   // (blocks of 256 threads: four waves, each running the chain on its own lanes, the rows shared)
    for (int j = 0; j < 16; ++j) for (uint32_t i = threadIdx.x; i < 324; i += blockDim.x) yrow[j][i] = 0.001f * (float)((i * 7 + j) % 97);
    __syncthreads();
    if (l < (uint32_t)LANES) {
        const uint32_t ya = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)&yrow[l & 15][0];
        const uint32_t ha = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)&hrow[l & 15][0];
        asm volatile("v_mov_b32 v80, 0\n v_mov_b32 v81, 0\n v_mov_b32 v82, 0\n v_mov_b32 v83, 0\n v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n"
                     "v_mov_b32 v48, 1.0\n v_mov_b32 v49, 0.5\n v_mov_b32 v50, 2.0\n v_mov_b32 v51, 1.0\n v_mov_b32 v52, 1.0\n v_mov_b32 v53, 0.5\n v_mov_b32 v54, 2.0\n v_mov_b32 v55, 1.0\n"
                     "v_mov_b32 v56, 1.0\n v_mov_b32 v57, 0.5\n v_mov_b32 v58, 2.0\n v_mov_b32 v59, 1.0\n v_mov_b32 v60, 1.0\n v_mov_b32 v61, 0.5\n v_mov_b32 v62, 2.0\n v_mov_b32 v63, 1.0\n" ::: CLOB);
        for (uint32_t t = 0; t < ntiles; ++t) {
            if (V == 0) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILE(NO_R, NO_W, NONE, NONE, NONE) :: "v"(ya), "v"(ha) : CLOB);
            if (V == 1) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILE(NO_R, NO_W, NONE, NONE, "v_mov_b32 v87, v88\n") :: "v"(ya), "v"(ha) : CLOB);
            if (V == 2) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILE(NO_R, NO_W, NONE, NONE, "v_mov_b32 v87, v88\n v_mov_b32 v88, v87\n") :: "v"(ya), "v"(ha) : CLOB);
            if (V == 3) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILE(RD, NO_W, "s_waitcnt lgkmcnt(0)\n", NONE, NONE) :: "v"(ya), "v"(ha) : CLOB);
            if (V == 4) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILE(NO_R, WR, NONE, NONE, NONE) :: "v"(ya), "v"(ha) : CLOB);
            if (V == 5) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILE(RD, WR, WAIT, NONE, NONE) :: "v"(ya), "v"(ha) : CLOB);
            if (V == 6) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILE(RD, WR, WAIT, "s_nop 0\n", NONE) :: "v"(ya), "v"(ha) : CLOB);
            if (V == 7) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILED(RD, WR, WAIT, NONE, NONE) :: "v"(ya), "v"(ha) : CLOB);
            if (V == 8) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILE(RD, WR64, WAIT4, NONE, NONE) :: "v"(ya), "v"(ha) : CLOB);
            if (V == 9) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILED(RD, WR64, WAIT4, NONE, NONE) :: "v"(ya), "v"(ha) : CLOB);
            if (V == 10) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILED(RD, WR2, WAIT, NONE, NONE) :: "v"(ya), "v"(ha) : CLOB);
            if (V == 12) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v44, %2\n s_mov_b32 s24, %3\n s_mov_b32 s25, %4\n" CH_TILEG(RD, WRG, WAIT0, NONE, NONE) :: "v"(ya), "v"(ha), "v"((uint32_t)((blockIdx.x * 16 + (l & 15)) * 1100 * 4)), "s"((uint32_t)(uintptr_t)gbuf), "s"((uint32_t)((uintptr_t)gbuf >> 32)) : CLOB);
            if (V == 13) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v44, %2\n s_mov_b32 s24, %3\n s_mov_b32 s25, %4\n" CH_TILEG(NO_R, WRG, NONE, NONE, NONE) :: "v"(ya), "v"(ha), "v"((uint32_t)((blockIdx.x * 16 + (l & 15)) * 1100 * 4)), "s"((uint32_t)(uintptr_t)gbuf), "s"((uint32_t)((uintptr_t)gbuf >> 32)) : CLOB);
            if (V == 11) asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n" CH_TILED(NO_R, WR, NONE, NONE, NONE) :: "v"(ya), "v"(ha) : CLOB);
        }
        float r;
        asm volatile("v_add_f32 %0, v79, v80" : "=v"(r) :: CLOB);
        out[blockIdx.x * 64 + l] = r + hrow[l & 15][20];
    }
}

template <typename K> void run4(const char* name, K k, float* f)   // the same wave counts as blocks of FOUR waves
{
    const uint32_t ntiles = 400;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    printf("%-46s", name);
    for (int blocks : {16, 256}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) { (void)hipEventRecord(a); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, f, ntiles); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("  %4d waves (x4 per block): %5.2f ns/sample", 4 * blocks, best * 1e6 / (ntiles * 256.0));
    }
    printf("\n");
}
template <typename K> void run(const char* name, K k, float* f)
{
    const uint32_t ntiles = 400;   // x 256 samples
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    printf("%-46s", name);
    for (int blocks : {1, 64, 1024}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) { (void)hipEventRecord(a); hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, f, ntiles); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("  %4d waves: %5.2f ns/sample", blocks, best * 1e6 / (ntiles * 256.0));
    }
    printf("\n");
}
int main()
{
    float* f; (void)hipMalloc(&f, 1 << 27); (void)hipMemset(f, 0, 1 << 27);
    run("bare chain, 16 lanes", chain_kernel<0, 16>, f);
    run("bare chain, 64 lanes", chain_kernel<0, 64>, f);
    run("bare chain, 1 lane", chain_kernel<0, 1>, f);
    run("+1 independent v_mov per sample, 16 lanes", chain_kernel<1, 16>, f);
    run("+2 independent v_mov per sample, 16 lanes", chain_kernel<2, 16>, f);
    run("LDS reads only, 16 lanes", chain_kernel<3, 16>, f);
    run("LDS writes only, 16 lanes", chain_kernel<4, 16>, f);
    run("LDS reads + writes, 16 lanes", chain_kernel<5, 16>, f);
    run("LDS reads + writes, 64 lanes", chain_kernel<5, 64>, f);
    run("LDS reads + writes + s_nop, 16 lanes", chain_kernel<6, 16>, f);
    run("reads + b128 writes one block late, 16 lanes", chain_kernel<7, 16>, f);
    run("reads + b128 writes one block late, 64 lanes", chain_kernel<7, 64>, f);
    run("reads + 2 x b64 writes, 16 lanes", chain_kernel<8, 16>, f);
    run("reads + 2 x b64 writes one block late, 16 lanes", chain_kernel<9, 16>, f);
    run("reads + write2_b64 one block late, 16 lanes", chain_kernel<10, 16>, f);
    run("b128 writes one block late only, 16 lanes", chain_kernel<11, 16>, f);
    run("LDS reads + GLOBAL 16-byte stores, 16 lanes", chain_kernel<12, 16>, f);
    run("LDS reads + GLOBAL 16-byte stores, 1 lane", chain_kernel<12, 1>, f);
    run("GLOBAL 16-byte stores only, 16 lanes", chain_kernel<13, 16>, f);
    run("LDS reads + writes, 1 lane", chain_kernel<5, 1>, f);
    run4("bare chain, 16 lanes, 4 waves per block", chain_kernel<0, 16>, f);
    run4("LDS reads + writes, 16 lanes, 4 waves per block", chain_kernel<5, 16>, f);
    return 0;
}

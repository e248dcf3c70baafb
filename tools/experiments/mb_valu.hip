// Experiment (not part of the product): issue cost of the VALU instructions the gate bodies are made of, in cycles
// per wave-instruction, at 1 / 2 / 3 / 4 waves per SIMD (s_memtime around a long unrolled block of independent ops).
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_valu tools/experiments/mb_valu.hip && /tmp/mb_valu
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND> __global__ __launch_bounds__(256) void k(uint64_t* out, int iters) {
    uint64_t t0, t1;
    asm volatile("v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 0.5\n v_mov_b32 v13, 0.5\n v_mov_b32 v14, 0.25\n v_mov_b32 v15, 0.25\n"
                 "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 0.5\n v_mov_b32 v19, 0.5\n v_mov_b32 v20, 0.25\n v_mov_b32 v21, 0.25\n"
                 "v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v25, 0.5\n s_mov_b32 s40, 0x3f000000\n s_mov_b32 s41, 0x3f000000"
                 ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "s40", "s41");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) asm volatile(REP16("v_pk_fma_f32 v[10:11], v[12:13], s[40:41], v[10:11]\n v_pk_fma_f32 v[14:15], v[16:17], s[40:41], v[14:15]\n v_pk_fma_f32 v[18:19], v[20:21], s[40:41], v[18:19]\n v_pk_fma_f32 v[22:23], v[24:25], s[40:41], v[22:23]\n") ::: "v10", "v11", "v14", "v15", "v18", "v19", "v22", "v23");
        if constexpr (KIND == 1) asm volatile(REP16("v_pk_add_f32 v[10:11], v[12:13], v[10:11]\n v_pk_add_f32 v[14:15], v[16:17], v[14:15]\n v_pk_add_f32 v[18:19], v[20:21], v[18:19]\n v_pk_add_f32 v[22:23], v[24:25], v[22:23]\n") ::: "v10", "v11", "v14", "v15", "v18", "v19", "v22", "v23");
        if constexpr (KIND == 2) asm volatile(REP16("v_mov_b64 v[10:11], v[12:13]\n v_mov_b64 v[14:15], v[16:17]\n v_mov_b64 v[18:19], v[20:21]\n v_mov_b64 v[22:23], v[24:25]\n") ::: "v10", "v11", "v14", "v15", "v18", "v19", "v22", "v23");
        if constexpr (KIND == 3) asm volatile(REP16("v_mov_b32 v10, v12\n v_mov_b32 v14, v16\n v_mov_b32 v18, v20\n v_mov_b32 v22, v24\n") ::: "v10", "v14", "v18", "v22");
        if constexpr (KIND == 4) asm volatile(REP16("v_fma_f32 v10, v12, s40, v10\n v_fma_f32 v14, v16, s40, v14\n v_fma_f32 v18, v20, s40, v18\n v_fma_f32 v22, v24, s40, v22\n") ::: "v10", "v14", "v18", "v22");
        if constexpr (KIND == 5) asm volatile(REP16("v_swap_b32 v10, v12\n v_swap_b32 v14, v16\n v_swap_b32 v18, v20\n v_swap_b32 v22, v24\n") ::: "v10", "v12", "v14", "v16", "v18", "v20", "v22", "v24");
        if constexpr (KIND == 6) asm volatile(REP16("v_pk_mul_f32 v[10:11], v[12:13], s[40:41] op_sel_hi:[1,0]\n v_pk_mul_f32 v[14:15], v[16:17], s[40:41] op_sel_hi:[1,0]\n v_pk_mul_f32 v[18:19], v[20:21], s[40:41] op_sel_hi:[1,0]\n v_pk_mul_f32 v[22:23], v[24:25], s[40:41] op_sel_hi:[1,0]\n") ::: "v10", "v11", "v14", "v15", "v18", "v19", "v22", "v23");
        if constexpr (KIND == 7) asm volatile(REP16("v_pk_fma_f32 v[10:11], v[12:13], s[40:41], v[10:11] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n v_pk_fma_f32 v[14:15], v[16:17], s[40:41], v[14:15] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n v_pk_fma_f32 v[18:19], v[20:21], s[40:41], v[18:19] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n v_pk_fma_f32 v[22:23], v[24:25], s[40:41], v[22:23] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n") ::: "v10", "v11", "v14", "v15", "v18", "v19", "v22", "v23");
        if constexpr (KIND == 8) asm volatile(REP16("v_pk_fma_f32 v[10:11], v[12:13], v[16:17], v[10:11]\n v_pk_fma_f32 v[14:15], v[16:17], v[20:21], v[14:15]\n v_pk_fma_f32 v[18:19], v[20:21], v[24:25], v[18:19]\n v_pk_fma_f32 v[22:23], v[24:25], v[12:13], v[22:23]\n") ::: "v10", "v11", "v14", "v15", "v18", "v19", "v22", "v23");
        if constexpr (KIND == 9) asm volatile(REP16("v_pk_mov_b32 v[10:11], v[12:13], v[12:13] op_sel:[0,1]\n v_pk_mov_b32 v[14:15], v[16:17], v[16:17] op_sel:[0,1]\n v_pk_mov_b32 v[18:19], v[20:21], v[20:21] op_sel:[0,1]\n v_pk_mov_b32 v[22:23], v[24:25], v[24:25] op_sel:[0,1]\n") ::: "v10", "v11", "v14", "v15", "v18", "v19", "v22", "v23");
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND> static void run(const char* name, uint64_t* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wg : {1, 2, 3, 4, 6, 8}) {      // workgroups of 256 threads per CU = waves per SIMD
        const int iters = 20000;
        hipLaunchKernelGGL(k<KIND>, dim3(256 * wg), dim3(256), 0, 0, d, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * wg), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        uint64_t c;
        hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
        const double ninst = iters * 64.0;
        printf("%-26s waves/SIMD %d: %6.2f ticks per instruction per wave; %7.3f ms -> %6.2f ns per instruction per wave, %5.2f ns per SIMD; tick = %.3f ns\n",
               name, wg, (double)c / ninst, ms, ms * 1e6 / ninst, ms * 1e6 / ninst / wg, ms * 1e6 / (double)c);
    }
}

int main() {
    uint64_t* d;
    hipMalloc(&d, 64);
    run<0>("v_pk_fma_f32 (sgpr src)", d);
    run<8>("v_pk_fma_f32 (vgpr srcs)", d);
    run<7>("v_pk_fma_f32 op_sel/neg", d);
    run<6>("v_pk_mul_f32", d);
    run<1>("v_pk_add_f32", d);
    run<2>("v_mov_b64", d);
    run<9>("v_pk_mov_b32", d);
    run<3>("v_mov_b32", d);
    run<4>("v_fma_f32", d);
    run<5>("v_swap_b32", d);
    return 0;
}

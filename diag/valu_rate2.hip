// microbenchmark 2: per-opcode issue cost on one SIMD with 2 waves resident (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S S S S S S S S
#define KERNEL(NAME, ASM, ...)                                                                         \
    __global__ void NAME(float *out, int iters) {                                                      \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                                 \
        int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;                                   \
        unsigned long long m = 0x5555555555555555ull;                                                   \
        for (int it = 0; it < iters; it++) {                                                           \
            asm volatile(REP8(REP8(ASM)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "s"(m) __VA_ARGS__); \
        }                                                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3;           \
    }
// each ASM string holds 4 independent instructions -> 256 per loop iteration
KERNEL(k_fma, "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n")
KERNEL(k_cnd_vcc, "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %4, vcc\n", : "vcc")
KERNEL(k_cnd_sgpr, "v_cndmask_b32 %4, %4, %5, %8\n v_cndmask_b32 %5, %5, %6, %8\n v_cndmask_b32 %6, %6, %7, %8\n v_cndmask_b32 %7, %7, %4, %8\n")
KERNEL(k_cnd_indep, "v_cndmask_b32 %4, %4, %4, %8\n v_cndmask_b32 %5, %5, %5, %8\n v_cndmask_b32 %6, %6, %6, %8\n v_cndmask_b32 %7, %7, %7, %8\n")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n", : "vcc")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %6, %6, %7, vcc\n", : "vcc")
KERNEL(k_and, "v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %4\n")
KERNEL(k_lshl, "v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7\n")
KERNEL(k_bfe, "v_bfe_u32 %4, %4, 3, 4\n v_bfe_u32 %5, %5, 3, 4\n v_bfe_u32 %6, %6, 3, 4\n v_bfe_u32 %7, %7, 3, 4\n")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %4, %4, %4\n v_bcnt_u32_b32 %5, %5, %5\n v_bcnt_u32_b32 %6, %6, %6\n v_bcnt_u32_b32 %7, %7, %7\n")
KERNEL(k_lshladd, "v_lshl_add_u32 %4, %4, 1, %5\n v_lshl_add_u32 %5, %5, 1, %6\n v_lshl_add_u32 %6, %6, 1, %7\n v_lshl_add_u32 %7, %7, 1, %4\n")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1\n")
KERNEL(k_trunc, "v_trunc_f32 %0, %0\n v_trunc_f32 %1, %1\n v_trunc_f32 %2, %2\n v_trunc_f32 %3, %3\n")
KERNEL(k_cvt, "v_cvt_i32_f32 %4, %0\n v_cvt_i32_f32 %5, %1\n v_cvt_i32_f32 %6, %2\n v_cvt_i32_f32 %7, %3\n")
KERNEL(k_ffbl, "v_ffbl_b32 %4, %4\n v_ffbl_b32 %5, %5\n v_ffbl_b32 %6, %6\n v_ffbl_b32 %7, %7\n")
KERNEL(k_dpp, "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_log, "v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n")
KERNEL(k_mul_u24, "v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %5, %5, %6\n v_mul_u32_u24 %6, %6, %7\n v_mul_u32_u24 %7, %7, %4\n")
KERNEL(k_mullo, "v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %4\n")
KERNEL(k_salu, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
template <typename K>
void run(const char *name, K kern, float *d, int wps) {
    const int iters = 4000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * 4 * wps), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * 4 * wps), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double)iters * 256 * wps;
    printf("%-12s waves/SIMD=%d  %.2f ns per instr per SIMD\n", name, wps, ms * 1e6 / n);
}
int main() {
    float *d; hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
#define R(k) run(#k, k, d, 1); run(#k, k, d, 2);
    R(k_fma) R(k_cnd_vcc) R(k_cnd_sgpr) R(k_cnd_indep) R(k_cmp) R(k_cmp_cnd) R(k_and) R(k_lshl) R(k_bfe) R(k_bcnt) R(k_lshladd) R(k_max3) R(k_trunc) R(k_cvt) R(k_ffbl) R(k_dpp) R(k_log) R(k_mul_u24) R(k_mullo) R(k_salu)
    return 0;
}

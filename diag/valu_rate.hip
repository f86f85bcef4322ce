// microbenchmark: VALU issue rate per SIMD for a few op classes at 1..4 waves per SIMD (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (OP == 1) { asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)); }
            if (OP == 2) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) :: "vcc"); }
            if (OP == 3) { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)); }
            if (OP == 4) { asm volatile("v_mul_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_max_f32 %2, %2, %2\n v_sub_f32 %3, %3, %3\n v_mul_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_max_f32 %6, %6, %6\n v_sub_f32 %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
            if (OP == 5) { asm volatile("v_and_b32 %0, %0, %1\n v_lshlrev_b32 %1, 1, %1\n v_bfe_u32 %2, %2, 3, 4\n v_bcnt_u32_b32 %3, %3, %3\n v_and_b32 %4, %4, %5\n v_lshlrev_b32 %5, 1, %5\n v_bfe_u32 %6, %6, 3, 4\n v_bcnt_u32_b32 %7, %7, %7" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}
template <int OP>
void run(const char *name, float *d) {
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps++) {
        // one block per CU (256 CUs), wps*4 waves per block -> wps waves per SIMD
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * 4 * wps), 0, 0, d, 10);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * 4 * wps), 0, 0, d, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double instr_per_simd = (double)iters * 64 * wps;   // VALU instructions issued on one SIMD
        printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f ns per instr per SIMD (%.2f cycles @2.4GHz)\n", name, wps, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}
int main() {
    float *d; hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
    run<0>("fma_f32", d); run<4>("f32 mix", d); run<1>("add_u32", d); run<2>("cndmask", d); run<3>("mov", d); run<5>("bitops", d);
    return 0;
}

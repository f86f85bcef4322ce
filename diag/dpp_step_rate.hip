// cycles per step of the two forms of the LPC recurrence's inner step, one wave per SIMD:
//   A: v_floor_f64 + 2 x v_readlane + v_fma_f64 (scalar operand)      B: v_floor_f64 + v_fmac_f64_dpp row_newbcast
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void form_a(double *p, unsigned long long *cyc, int iters) {
    double acc = p[threadIdx.x], c = 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const double x = floor(acc);
            const int lo = __builtin_amdgcn_readlane(__double2loint(x), t), hi = __builtin_amdgcn_readlane(__double2hiint(x), t);
            acc = fma(c, __hiloint2double(hi, lo), acc);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    p[threadIdx.x + 64 * blockIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void form_b(double *p, unsigned long long *cyc, int iters) {
    double acc = p[threadIdx.x], c = 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            double x;
            asm("v_floor_f64 %1, %0\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc), "=&v"(x) : "v"(c), "n"(t));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    p[threadIdx.x + 64 * blockIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// C: the broadcast as two v_mov_b32_dpp row_newbcast (is it the 64-bit DPP operand that is slow?)
__global__ void form_c(double *p, unsigned long long *cyc, int iters) {
    double acc = p[threadIdx.x], c = 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const double x = floor(acc);
            int lo = __double2loint(x), hi = __double2hiint(x), blo, bhi;
            asm("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf" : "=&v"(blo), "=&v"(bhi) : "v"(lo), "v"(hi), "n"(t));
            acc = fma(c, __hiloint2double(bhi, blo), acc);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    p[threadIdx.x + 64 * blockIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// D: form B with sixteen different coefficient registers and NB blocks unrolled (straight-line code of NB x 16 steps, as in ll_predict)
template <int NB>
__global__ void form_d(double *p, unsigned long long *cyc, int iters) {
    double acc = p[threadIdx.x], C[16];
#pragma unroll
    for (int t = 0; t < 16; t++) C[t] = p[64 + threadIdx.x + 64 * t] * 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters / NB; i++) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
#pragma unroll
            for (int t = 0; t < 16; t++) {
                double x;
                asm("v_floor_f64 %1, %0\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc), "=&v"(x) : "v"(C[t]), "n"(t));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    p[threadIdx.x + 64 * blockIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double *p; unsigned long long *cyc;
    hipMalloc(&p, 64 * 4096 * 8); hipMalloc(&cyc, 4096 * 8);
    hipMemset(p, 0, 64 * 4096 * 8);
    const int iters = 4096;
    for (int wg : {256, 1024, 4096}) {
        for (int f = 0; f < 3; f++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                if (f == 0) form_a<<<wg, 64>>>(p, cyc, iters);
                else if (f == 1) form_b<<<wg, 64>>>(p, cyc, iters);
                else form_c<<<wg, 64>>>(p, cyc, iters);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d one-wave workgroups, form %c: %.3f ms, %.1f ns per step, wave 0: %.1f counter ticks per step\n", wg, "ABC"[f], ms, ms * 1e6 / (iters * 16.0), (double)h / (iters * 16.0));
        }
    }
    for (int nbv = 0; nbv < 3; nbv++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (nbv == 0) form_d<1><<<640, 64>>>(p, cyc, iters);
            else if (nbv == 1) form_d<4><<<640, 64>>>(p, cyc, iters);
            else form_d<16><<<640, 64>>>(p, cyc, iters);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("640 one-wave workgroups, form D with %d blocks unrolled: %.3f ms, %.1f ns per step, wave 0: %.1f counter ticks per step\n", nbv == 0 ? 1 : nbv == 1 ? 4 : 16, ms, ms * 1e6 / (iters * 16.0), (double)h / (iters * 16.0));
    }
    return 0;
}

// Is a wave that walks a long straight-line body (tens of KB of code per pass, like the transform wave's frame body)
// slower per instruction than the same instructions in a loop that fits the instruction buffer / stays hot in the
// instruction cache? BODY = packed-f32 operations per loop pass (8 bytes each); 1..3 waves per SIMD; dependent and
// independent chains.   hipcc --offload-arch=gfx950 -O3 -o diag/fetch_rate diag/fetch_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int BODY, int CHAINS>
__global__ __launch_bounds__(64) void k(float *out, const float *in, int passes) {
    const int t = threadIdx.x;
    v2f a[CHAINS];
    for (int i = 0; i < CHAINS; i++) a[i] = (v2f){in[(t + i) & 255], in[(t + i + 8) & 255]};
    const v2f W0 = {in[1], in[2]}, W1 = {in[3], in[4]};
    for (int p = 0; p < passes; p++) {
#pragma unroll
        for (int i = 0; i < BODY; i++) {
            a[i % CHAINS] = __builtin_elementwise_fma(a[i % CHAINS], W0, W1);
        }
        asm volatile("" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < CHAINS; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 64 + t] = s;
}

template <int BODY, int CHAINS>
static void run(float *out, const float *in, const char *what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 3; wps++) {
        const int blocks = 256 * 4 * wps;
        const int passes = (1 << 22) / BODY;     // the same instruction count for every body size
        k<BODY, CHAINS><<<blocks, 64>>>(out, in, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<BODY, CHAINS><<<blocks, 64>>>(out, in, passes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)passes * BODY;
        printf("%-28s body=%5d (%6d B) chains=%2d waves/SIMD=%d  %8.3f ms  %.2f cycles per instruction per wave (2.4 GHz), %.2f per SIMD\n", what, BODY, BODY * 8,
               CHAINS, wps, ms, ms * 1e-3 * 2.4e9 / instr, ms * 1e-3 * 2.4e9 / instr / wps);
    }
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1024); hipMalloc(&out, 256 * 4 * 3 * 64 * 4);
    float h[256]; for (int i = 0; i < 256; i++) h[i] = 0.001f * i + 0.5f;
    hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
    run<16, 8>(out, in, "tight loop, 8 chains");
    run<512, 8>(out, in, "4 KB body, 8 chains");
    run<2048, 8>(out, in, "16 KB body, 8 chains");
    run<4096, 8>(out, in, "32 KB body, 8 chains");
    run<16, 1>(out, in, "tight loop, 1 chain");
    run<4096, 1>(out, in, "32 KB body, 1 chain");
    run<16, 2>(out, in, "tight loop, 2 chains");
    run<4096, 2>(out, in, "32 KB body, 2 chains");
    return 0;
}

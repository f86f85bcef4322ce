// v_mad_i64_i32 issue rate on gfx950 (diagnostic): the lossless analysis is made of these
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(long long *out, const int *in, int iters) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int a[8];
    long long acc[8];
    for (int i = 0; i < 8; i++) { a[i] = in[(t + i) & 255]; acc[i] = i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] += (long long)a[i] * (long long)a[(i + 1) & 7];
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] ^= (int)(acc[i] >> 40);   // keep the multiplies from being hoisted (cheap op)
    }
    long long s = 0;
    for (int i = 0; i < 8; i++) s += acc[i];
    out[t] = s;
}
int main() {
    int *in; long long *out;
    hipMalloc(&in, 1024); hipMalloc(&out, 256 * 1024 * 8);
    int h[256]; for (int i = 0; i < 256; i++) h[i] = i * 7919 + 13;
    hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc = 1; wpc <= 4; wpc *= 2) {
        int blocks = 256 * wpc, iters = 20000;
        k<<<blocks, 256>>>(out, in, 10); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<blocks, 256>>>(out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mads = (double)blocks * 256 * iters * 8;
        printf("waves/SIMD=%d: %.3f ms, %.2f T i64 MAD/s chip-wide, %.2f cycles per wave-MAD per SIMD (incl. 1 xor+shift per MAD)\n", wpc, ms, mads / ms / 1e9,
               ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wpc));
    }
    return 0;
}

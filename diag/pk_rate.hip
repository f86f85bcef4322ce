#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
// A: two channels as separate scalars; B: two channels as one float2 (packed ops). Same arithmetic per component.
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    float w0 = in[t & 255], w1 = in[(t + 1) & 255];
    if (MODE == 0) {
        float a[8], b[8];
        for (int i = 0; i < 8; i++) { a[i] = in[(t + i) & 255]; b[i] = in[(t + i + 8) & 255]; }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float ta = fmaf(a[i], w0, -(a[(i + 1) & 7] * w1));
                float tb = fmaf(b[i], w0, -(b[(i + 1) & 7] * w1));
                a[i] = ta + a[(i + 3) & 7];
                b[i] = tb + b[(i + 3) & 7];
            }
        }
        float s = 0;
        for (int i = 0; i < 8; i++) s += a[i] + b[i];
        out[t] = s;
    } else {
        v2f a[8];
        for (int i = 0; i < 8; i++) a[i] = (v2f){in[(t + i) & 255], in[(t + i + 8) & 255]};
        const v2f W0 = {w0, w0}, W1 = {w1, w1};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                v2f m = a[(i + 1) & 7] * W1;
                v2f ta = __builtin_elementwise_fma(a[i], W0, -m);
                a[i] = ta + a[(i + 3) & 7];
            }
        }
        float s = 0;
        for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
        out[t] = s;
    }
}
int main() {
    float *in, *out;
    hipMalloc(&in, 1024); hipMalloc(&out, 256 * 256 * 16 * 4 * 4);
    float h[256]; for (int i = 0; i < 256; i++) h[i] = 0.001f * i + 0.5f;
    hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc = 1; wpc <= 4; wpc++) {   // workgroups of 256 threads per CU -> waves per SIMD
        for (int mode = 0; mode < 2; mode++) {
            int blocks = 256 * wpc, iters = 20000;
            if (mode == 0) k<0><<<blocks, 256>>>(out, in, 10); else k<1><<<blocks, 256>>>(out, in, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 256>>>(out, in, iters); else k<1><<<blocks, 256>>>(out, in, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flopinstr = (double)iters * 8 * 3 * 2;   // scalar-equivalent f32 instructions per thread-pair
            printf("waves/SIMD=%d mode=%s  %.3f ms  -> %.2f cycles per (2-channel op) per SIMD at 2.4GHz\n", wpc, mode ? "pk" : "scalar", ms,
                   ms * 1e-3 * 2.4e9 / (iters * 8.0 * 3 * wpc));
        }
    }
    return 0;
}

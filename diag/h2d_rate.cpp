// host->device upload strategies from pageable memory (diagnostic): direct hipMemcpy, hipHostRegister, threaded staging
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = (size_t)256 << 20;
    char *src = (char *)malloc(N);
    memset(src, 1, N);
    char *dev; hipMalloc(&dev, N);
    char *pin; hipHostMalloc(&pin, N);
    hipStream_t st; hipStreamCreate(&st);
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); hipMemcpy(dev, src, N, hipMemcpyHostToDevice); double d = now() - t;
        printf("pageable hipMemcpy: %.1f GB/s\n", N / d / 1e9);
        t = now(); hipMemcpy(dev, pin, N, hipMemcpyHostToDevice); d = now() - t;
        printf("pinned hipMemcpy: %.1f GB/s\n", N / d / 1e9);
        t = now(); memcpy(pin, src, N); d = now() - t;
        printf("1-thread memcpy pageable->pinned: %.1f GB/s\n", N / d / 1e9);
        for (int nt : {2, 4, 8, 16}) {
            t = now();
            std::vector<std::thread> th;
            for (int i = 0; i < nt; i++) th.emplace_back([=] { memcpy(pin + N / nt * i, src + N / nt * i, N / nt); });
            for (auto &x : th) x.join();
            d = now() - t;
            printf("%d-thread memcpy pageable->pinned: %.1f GB/s\n", nt, N / d / 1e9);
        }
        t = now(); hipHostRegister(src, N, hipHostRegisterDefault); double dr = now() - t;
        t = now(); hipMemcpy(dev, src, N, hipMemcpyHostToDevice); d = now() - t;
        double t2 = now(); hipHostUnregister(src); double du = now() - t2;
        printf("hipHostRegister %.2f ms, copy %.1f GB/s, unregister %.2f ms\n", dr * 1e3, N / d / 1e9, du * 1e3);
        // small copies: 3.5 MB
        const size_t S = 3528000;
        t = now(); for (int i = 0; i < 20; i++) hipMemcpy(dev, src + i * S, S, hipMemcpyHostToDevice); d = (now() - t) / 20;
        printf("3.5 MB pageable hipMemcpy: %.1f us (%.1f GB/s)\n", d * 1e6, S / d / 1e9);
        t = now(); for (int i = 0; i < 20; i++) { memcpy(pin, src + i * S, S); hipMemcpy(dev, pin, S, hipMemcpyHostToDevice); } d = (now() - t) / 20;
        printf("3.5 MB memcpy+pinned hipMemcpy: %.1f us\n", d * 1e6);
        t = now(); for (int i = 0; i < 20; i++) hipMemcpy(src + i * S, dev, 180000, hipMemcpyDeviceToHost); d = (now() - t) / 20;
        printf("180 KB D2H pageable: %.1f us\n", d * 1e6);
    }
    return 0;
}

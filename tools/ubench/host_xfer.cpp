// Development probe for the streaming host-buffer entry points (host path): what pinning the caller's pages costs
// (hipHostRegister / hipHostUnregister), and how fast pageable, registered and library-pinned buffers move over PCIe, one
// direction and both at once.   hipcc -O2 -o host_xfer host_xfer.cpp -lpthread && ./host_xfer
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const size_t N = 48u << 20;
    void *d0, *d1;
    CK(hipMalloc(&d0, N)); CK(hipMalloc(&d1, N));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    char* pg = (char*)aligned_alloc(4096, N); char* pg2 = (char*)aligned_alloc(4096, N);
    std::memset(pg, 1, N); std::memset(pg2, 2, N);
    for (int rep = 0; rep < 3; ++rep) {
        double t = now();
        CK(hipMemcpy(d0, pg, N, hipMemcpyHostToDevice));
        double a = now() - t; t = now();
        CK(hipMemcpy(pg2, d1, N, hipMemcpyDeviceToHost));
        double b = now() - t;
        std::printf("pageable  H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", a * 1e3, N / a / 1e9, b * 1e3, N / b / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t = now();
        CK(hipHostRegister(pg, N, hipHostRegisterDefault)); CK(hipHostRegister(pg2, N, hipHostRegisterDefault));
        double reg = now() - t; t = now();
        CK(hipMemcpyAsync(d0, pg, N, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
        double a = now() - t; t = now();
        CK(hipMemcpyAsync(pg2, d1, N, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1));
        double b = now() - t; t = now();
        CK(hipMemcpyAsync(d0, pg, N, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(pg2, d1, N, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        double both = now() - t; t = now();
        CK(hipHostUnregister(pg)); CK(hipHostUnregister(pg2));
        double unreg = now() - t;
        std::printf("registered: register 2 x 48 MB %.2f ms, unregister %.2f ms; H2D %.2f ms (%.1f GB/s) D2H %.2f ms (%.1f GB/s) both at once %.2f ms\n",
                    reg * 1e3, unreg * 1e3, a * 1e3, N / a / 1e9, b * 1e3, N / b / 1e9, both * 1e3);
    }
    char *pin, *pin2;
    CK(hipHostMalloc((void**)&pin, N, hipHostMallocDefault)); CK(hipHostMalloc((void**)&pin2, N, hipHostMallocDefault));
    for (int threads : { 1, 2, 4, 8 }) {
        double t = now();
        std::vector<std::thread> th;
        for (int k = 0; k < threads; ++k) th.emplace_back([&, k] { size_t a = N * k / threads, b = N * (k + 1) / threads; std::memcpy(pin + a, pg + a, b - a); });
        for (auto& x : th) x.join();
        double c = now() - t;
        std::printf("memcpy pageable -> pinned, %d thread(s): %.2f ms (%.1f GB/s)\n", threads, c * 1e3, N / c / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t = now();
        CK(hipMemcpyAsync(d0, pin, N, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(pin2, d1, N, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        std::printf("library-pinned both directions at once: %.2f ms\n", (now() - t) * 1e3);
    }
    return 0;
}

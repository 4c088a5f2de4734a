// What does v_cvt_pk_u8_f32 do with fractions, ties, negatives, large values and NaN?  (decode kernel: truncate + clamp + pack in one
// instruction?)   hipcc -O3 --offload-arch=gfx950 -o cvt_pk_u8 cvt_pk_u8.hip && ./cvt_pk_u8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* x, unsigned* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned r;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(x[i]), "v"(0xAABBCCDDu));
    out[i] = r;
}
int main()
{
    const float v[] = { 0.f, 0.2f, 0.5f, 0.7f, 0.9999f, 1.f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.9999f, 255.f, 255.4f, 255.5f, 256.f, 300.f, 1e9f,
                        -0.2f, -0.5f, -0.9f, -1.f, -1.5f, -300.f, -1e9f, NAN, INFINITY, -INFINITY, 127.99999f, 128.00001f };
    const int n = sizeof v / sizeof v[0];
    float* dx; unsigned* dout; unsigned h[64];
    (void)hipMalloc(&dx, sizeof v); (void)hipMalloc(&dout, sizeof h);
    (void)hipMemcpy(dx, v, sizeof v, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout, n);
    (void)hipMemcpy(h, dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%14.6f -> byte1 %3u  (dword %08x)\n", v[i], (h[i] >> 8) & 0xFF, h[i]);
    return 0;
}

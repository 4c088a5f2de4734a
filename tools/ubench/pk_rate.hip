// Micro-benchmark: v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 vs scalar FP32 issue cost on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define N_ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed)
{
    float2_t a[8];
    float f[16];
    for (int i = 0; i < 8; ++i) { a[i] = float2_t{seed + i + threadIdx.x, seed - i}; }
    for (int i = 0; i < 16; ++i) f[i] = seed * i + threadIdx.x;
    const float2_t c1 = {seed * 0.37f, seed * 0.41f}, c2 = {seed * 1.01f, seed * 0.99f};
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = u & 7;
            if (OP == 0) a[i] = __builtin_elementwise_fma(a[i], c1, c2);     // v_pk_fma_f32
            if (OP == 1) a[i] = a[i] * c1;                                   // v_pk_mul_f32
            if (OP == 2) a[i] = a[i] + c2;                                   // v_pk_add_f32
            if (OP == 3) f[u] = __builtin_fmaf(f[u], c1.x, c2.x);            // v_fma_f32
            if (OP == 4) f[u] = f[u] + c2.x;                                 // v_add_f32
            if (OP == 5) f[u] = __builtin_truncf(f[u]) + c2.x;               // trunc + add
            if (OP == 6) f[u] = __builtin_rintf(f[u] * c1.x);                // mul + rndne
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, int instr, int w)
{
    const int blocks = 256 * w;
    float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 1.5f); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, 1.5f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)N_ITER * 16 * instr * w;
    printf("%-22s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD\n", name, w, ms, ms * 1e6 / n);
    (void)hipFree(out);
}
int main()
{
    for (int w : {2, 4, 8}) {
        run<0>("v_pk_fma_f32", 1, w); run<1>("v_pk_mul_f32", 1, w); run<2>("v_pk_add_f32", 1, w);
        run<3>("v_fma_f32", 1, w); run<4>("v_add_f32", 1, w); run<5>("v_trunc_f32+v_add_f32", 2, w); run<6>("v_mul_f32+v_rndne_f32", 2, w);
    }
}

// Micro-benchmark: does MFMA work of one wave overlap the VALU work of the other waves of its SIMD on gfx950?
//   valu  : 16 independent v_fma_f32 chains per lane per iteration
//   mfma  : M independent v_mfma_f32_16x16x4_f32 per iteration
//   both  : the two together.   overlap  <=>  t(both) ~ max(t(valu), t(mfma));  none  <=>  t(both) ~ sum
//   hipcc -O3 --offload-arch=gfx950 -o mfma_overlap mfma_overlap.hip && ./mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2048
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));

template <int VALU, int MFMA, int BF16 = 0>
__global__ __launch_bounds__(256) void k(float* out, float seed)
{
    float f[16];
    for (int i = 0; i < 16; ++i) f[i] = seed * i + threadIdx.x;
    const float g = seed + threadIdx.x, h = seed * 3.f;
    v4f acc[4] = { {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0} };
    for (int it = 0; it < N_ITER; ++it) {
        if (VALU) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g), "v"(h));
        }
#pragma unroll
        for (int m = 0; m < MFMA; ++m) {
            if (BF16) {
                const v4s a = { (short)threadIdx.x, 1, 2, 3 }, b = { 3, 2, 1, (short)threadIdx.x };
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[m], 0, 0, 0);
            } else {
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(g, h, acc[m], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += f[i];
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VALU, int MFMA, int BF16 = 0>
static void run(const char* name, int waves_per_simd)
{
    float* out;
    const int blocks = 256 * waves_per_simd;              // 256 CUs x (4 SIMDs x waves) / 4 waves per block
    hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<VALU, MFMA, BF16>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<VALU, MFMA, BF16>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves_per_simd waves x N_ITER iterations
    std::printf("%-28s %d waves/SIMD: %7.3f ms  = %6.1f ns per iteration per wave-slot (SIMD time per wave-iteration %5.1f ns)\n", name,
                waves_per_simd, ms, ms * 1e6 / N_ITER, ms * 1e6 / N_ITER / waves_per_simd);
    hipFree(out);
}

int main()
{
    for (int w : { 4, 5, 8 }) {
        if (w == 4) { run<1, 0>("valu x16", 4); run<0, 1>("mfma x1", 4); run<1, 1>("valu x16 + mfma x1", 4); run<0, 2>("mfma x2", 4); run<1, 2>("valu x16 + mfma x2", 4); run<1, 4>("valu x16 + mfma x4", 4); }
        if (w == 5) { run<1, 0>("valu x16", 5); run<0, 1>("mfma x1", 5); run<1, 1>("valu x16 + mfma x1", 5); run<1, 2>("valu x16 + mfma x2", 5); }
        if (w == 8) { run<1, 0>("valu x16", 8); run<0, 1>("mfma x1", 8); run<1, 1>("valu x16 + mfma x1", 8); run<1, 2>("valu x16 + mfma x2", 8); }
    }
    run<0, 1, 1>("bf16 mfma 16x16x16 x1", 5); run<1, 1, 1>("valu x16 + bf16 mfma x1", 5); run<0, 2, 1>("bf16 mfma x2", 5); run<1, 2, 1>("valu x16 + bf16 mfma x2", 5);
    return 0;
}

// Micro-benchmark: issue cost of the non-FMA VALU ops the f32 encode kernel uses (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2048
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed, unsigned useed)
{
    float f[16]; unsigned u[16]; int n[16];
    for (int i = 0; i < 16; ++i) { f[i] = seed * i + threadIdx.x; u[i] = useed * (i + 1) + threadIdx.x; n[i] = (int)u[i]; }
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) f[i] = (float)((u[i] >> 8) & 0xFFu) + f[i];           // cvt_f32_ubyte1 + add
            if (OP == 1) { n[i] = (int)f[i]; f[i] = f[i] + (float)(n[i] & 1); }   // cvt_i32_f32 + and + cvt_f32_i32 + add
            if (OP == 2) f[i] = (f[i] < seed) ? f[i] + 1.f : f[i] - 1.f;        // cmp + 2 ops + cndmask
            if (OP == 3) u[i] = u[i] * 2654435u + 12345u;                       // mad_u32_u24? (mul_lo)
            if (OP == 4) u[i] = ((u[i] >> 3) & 0xFFFF) + (u[i] << 2);           // bfe/and + lshl_add
            if (OP == 5) f[i] = __builtin_fabsf(f[i] - 500.f) == 500.f ? f[i] * 0.5f : f[i] + 2.f;   // sub + cmp + mul/add + cndmask
            if (OP == 6) u[i] = __builtin_amdgcn_update_dpp((int)u[i], (int)u[i], 0x114, 0xF, 0xA, false) + 1;   // dpp mov + add
            if (OP == 7) f[i] = __builtin_truncf(f[i] * 0.999f);                // mul + trunc
            if (OP == 8) f[i] = __builtin_fmaf(f[i], 0.999f, 114.f);            // fma with literals
            if (OP == 9) u[i] = (unsigned)__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(f[i], f[(i + 1) & 15])) + u[i];
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += f[i] + (float)u[i] + (float)n[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, int w)
{
    const int blocks = 256 * w;
    float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 1.5f, 77u); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, 1.5f, 77u); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)N_ITER * 16 * w;
    printf("%-44s waves/SIMD=%d  %.2f ns per source statement per SIMD\n", name, w, ms * 1e6 / n);
    (void)hipFree(out);
}
int main()
{
    for (int w : {4, 8}) {
        run<0>("cvt_f32_ubyte1 + add_f32", w); run<1>("cvt_i32_f32 + and + cvt_f32_i32 + add", w);
        run<2>("cmp_lt + add + sub + cndmask", w); run<3>("mul_lo_u32 + add", w); run<4>("lshr/and + lshl_add", w);
        run<5>("sub + cmp_eq(|.|) + mul + add + cndmask", w); run<6>("mov_dpp + add_u32", w); run<7>("mul_f32 + trunc_f32", w);
        run<8>("fma_f32 with two literals", w); run<9>("cvt_pkrtz + add_u32", w);
    }
}

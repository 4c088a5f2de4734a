// Micro-benchmark, second part: the FP64, conversion and integer forms the DECODE kernel uses (dequant_idct_kernel: a third of its
// vector instructions are FP64).  Same method as valu_cost.hip: inline asm, distinct registers, a dependency distance of several
// instructions, W waves per SIMD; ns per wave-instruction per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o valu_cost_f64 valu_cost_f64.hip && ./valu_cost_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 1024

// D: double destination, double sources; M: the statement names d[i] (double), d[(i+3)&7], f[i] (float/int), f[(i+5)&7]
#define DEFM(NAME, ASM)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, float seed)                                     \
    {                                                                                                       \
        double d[8]; float f[8];                                                                            \
        for (int i = 0; i < 8; ++i) { d[i] = (double)seed * i + threadIdx.x; f[i] = seed * i + threadIdx.x; } \
        for (int it = 0; it < N_ITER; ++it) {                                                               \
            _Pragma("unroll") for (int r = 0; r < 2; ++r)                                                    \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                    \
                asm volatile(ASM : "+v"(d[i]), "+v"(f[i]) : "v"(d[(i + 3) & 7]), "v"(f[(i + 5) & 7]), "v"(d[(i + 5) & 7]), "v"(f[(i + 3) & 7])); \
        }                                                                                                   \
        double s = 0;                                                                                       \
        for (int i = 0; i < 8; ++i) s += d[i] + f[i];                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;                                              \
    }
// operands: %0 d[i]  %1 f[i]  %2 d[i+3]  %3 f[i+5]  %4 d[i+5]  %5 f[i+3]
DEFM(add_f64, "v_add_f64 %0, %2, %4")
DEFM(mul_f64, "v_mul_f64 %0, %2, %4")
DEFM(fma_f64, "v_fma_f64 %0, %2, %4, %0")
DEFM(fract_f64, "v_fract_f64 %0, %2")
DEFM(cvt_f64_i32, "v_cvt_f64_i32 %0, %3")
DEFM(cvt_i32_f64, "v_cvt_i32_f64 %1, %2")
DEFM(cvt_f64_f32, "v_cvt_f64_f32 %0, %3")
DEFM(cvt_f32_f64, "v_cvt_f32_f64 %1, %2")
DEFM(cvt_f32_i32, "v_cvt_f32_i32 %1, %3")
DEFM(cvt_i32_f32, "v_cvt_i32_f32 %1, %3")
DEFM(fma_f32, "v_fma_f32 %1, %3, %5, %1")
DEFM(mul_lo_u32, "v_mul_lo_u32 %1, %3, %5")
DEFM(mul_hi_u32, "v_mul_hi_u32 %1, %3, %5")
DEFM(mad_u32_u24, "v_mad_u32_u24 %1, %3, %5, %1")
DEFM(mad_i32_i24, "v_mad_i32_i24 %1, %3, %5, %1")
DEFM(ctl_add_f32, "v_add_f32 %1, %3, %5")
DEFM(ctl_and_b32, "v_and_b32 %1, %3, %5")
DEFM(add_u32, "v_add_u32 %1, %3, %5")
DEFM(sub_u32, "v_sub_u32 %1, %3, %5")
DEFM(lshl_or, "v_lshl_or_b32 %1, %3, 8, %5")
DEFM(cvt_f32_ub1, "v_cvt_f32_ubyte1 %1, %3")
DEFM(min3_u32, "v_min3_u32 %1, %3, %5, %1")
DEFM(max3_i32, "v_max3_i32 %1, %3, %5, %1")
DEFM(or3, "v_or3_b32 %1, %3, %5, %1")
DEFM(add3, "v_add3_u32 %1, %3, %5, %1")
DEFM(sat_pk_u8, "v_sat_pk_u8_i16 %1, %3")
DEFM(cvt_pk_i16, "v_cvt_pk_i16_i32 %1, %3, %5")
DEFM(alignbit, "v_alignbit_b32 %1, %3, %5, 7")
DEFM(lshlrev, "v_lshlrev_b32 %1, 3, %3")
DEFM(bfe_i32, "v_bfe_i32 %1, %3, 16, 16")
DEFM(lshl_b64, "v_lshlrev_b64 %0, %3, %2")
DEFM(lshr_b64, "v_lshrrev_b64 %0, %3, %2")
DEFM(ffbh_u32, "v_ffbh_u32 %1, %3")
DEFM(bcnt_u32, "v_bcnt_u32_b32 %1, %3, %5")
DEFM(lshl_b32_vv, "v_lshlrev_b32 %1, %3, %5")
DEFM(lshr_b32_vv, "v_lshrrev_b32 %1, %3, %5")
DEFM(or_b32, "v_or_b32 %1, %3, %5")
DEFM(xor_b32, "v_xor_b32 %1, %3, %5")
DEFM(max_i32, "v_max_i32 %1, %3, %5")
DEFM(bfi_b32, "v_bfi_b32 %1, %3, %5, %1")
DEFM(and_or, "v_and_or_b32 %1, %3, %5, %1")
DEFM(xad_u32, "v_xad_u32 %1, %3, %5, %1")

typedef void (*kern_t)(float*, float);
static void run(const char* name, kern_t k, int w)
{
    const int blocks = 256 * w;   // 256 CUs x (w workgroups of 4 waves) => w waves per SIMD
    float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.5f); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.5f); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double n = (double)N_ITER * 16 * w;
    printf("%-16s w=%d  %.2f ns per wave-instruction per SIMD\n", name, w, best * 1e6 / n); fflush(stdout);
    (void)hipFree(out);
}
#define RUN(K) run(#K, K, w);
int main()
{
    for (int w : { 1, 4, 8 }) {
        RUN(add_f64) RUN(mul_f64) RUN(fma_f64) RUN(fract_f64) RUN(cvt_f64_i32) RUN(cvt_i32_f64) RUN(cvt_f64_f32) RUN(cvt_f32_f64)
        RUN(cvt_f32_i32) RUN(cvt_i32_f32) RUN(fma_f32) RUN(mul_lo_u32) RUN(mul_hi_u32) RUN(mad_u32_u24) RUN(mad_i32_i24)
        RUN(ctl_add_f32) RUN(ctl_and_b32) RUN(add_u32) RUN(sub_u32) RUN(lshl_or) RUN(cvt_f32_ub1)
        RUN(min3_u32) RUN(max3_i32) RUN(or3) RUN(add3) RUN(sat_pk_u8) RUN(cvt_pk_i16) RUN(alignbit) RUN(lshlrev) RUN(bfe_i32)
        RUN(lshl_b64) RUN(lshr_b64) RUN(ffbh_u32) RUN(bcnt_u32) RUN(lshl_b32_vv) RUN(lshr_b32_vv) RUN(or_b32) RUN(xor_b32) RUN(max_i32) RUN(bfi_b32) RUN(and_or) RUN(xad_u32)
    }
    return 0;
}

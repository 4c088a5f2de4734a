// Micro-benchmark: issue cost of single gfx950 VALU instructions, written as inline asm so that the compiler cannot
// fuse, vectorise or drop them.  16 independent register chains per lane, 8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o asm_rate asm_rate.hip && ./asm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 1024
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define DEF_KERNEL(NAME, ASM)                                                                          \
    __global__ __launch_bounds__(256) void NAME(float* out, float seed)                                \
    {                                                                                                  \
        float f[16];                                                                                   \
        float g = seed + threadIdx.x, h = seed * 3.f; float sg = seed * 5.f;                                                  \
        for (int i = 0; i < 16; ++i) f[i] = seed * i + threadIdx.x;                                    \
        for (int it = 0; it < N_ITER; ++it) {                                                          \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(f[i]) : "v"(g), "v"(h), "s"(sg) : "s10", "s11", "vcc"); \
        }                                                                                              \
        float s = 0;                                                                                   \
        for (int i = 0; i < 16; ++i) s += f[i];                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                \
    }
#define DEF_KERNEL64(NAME, ASM)                                                                        \
    __global__ __launch_bounds__(256) void NAME(float* out, float seed)                                \
    {                                                                                                  \
        double f[16];                                                                                  \
        double g = seed + threadIdx.x, h = seed * 3.f; float sg = seed * 5.f;                                                 \
        for (int i = 0; i < 16; ++i) f[i] = seed * i + threadIdx.x;                                    \
        for (int it = 0; it < N_ITER; ++it) {                                                          \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(f[i]) : "v"(g), "v"(h), "s"(sg) : "s10", "s11", "vcc"); \
        }                                                                                              \
        double s = 0;                                                                                  \
        for (int i = 0; i < 16; ++i) s += f[i];                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;                                         \
    }

DEF_KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
DEF_KERNEL(k_add, "v_add_f32 %0, %0, %1")
DEF_KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
DEF_KERNEL64(k_pkfma, "v_pk_fma_f32 %0, %0, %1, %2")
DEF_KERNEL64(k_pkadd, "v_pk_add_f32 %0, %0, %1")
DEF_KERNEL64(k_pkmul, "v_pk_mul_f32 %0, %0, %1")
DEF_KERNEL(k_cvtub, "v_cvt_f32_ubyte1 %0, %0")
DEF_KERNEL(k_trunc, "v_trunc_f32 %0, %0")
DEF_KERNEL(k_rndne, "v_rndne_f32 %0, %0")
DEF_KERNEL(k_fract, "v_fract_f32 %0, %0")
DEF_KERNEL(k_cvti, "v_cvt_i32_f32 %0, %0")
DEF_KERNEL(k_cvtf, "v_cvt_f32_i32 %0, %0")
DEF_KERNEL(k_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
DEF_KERNEL(k_cvtf16, "v_cvt_f32_f16 %0, %0")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
DEF_KERNEL(k_dot2, "v_dot2_i32_i16 %0, %1, %2, %0")
DEF_KERNEL(k_dot2c, "v_dot2c_i32_i16 %0, %1, %2")
DEF_KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF_KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
DEF_KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
DEF_KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 3, %1")
DEF_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
DEF_KERNEL(k_and, "v_and_b32 %0, %0, %1")
DEF_KERNEL(k_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
DEF_KERNEL(k_cvti_sdwa, "v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD")
DEF_KERNEL(k_dpp, "v_mov_b32_dpp %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa")
DEF_KERNEL(k_cmp, "v_cmp_eq_f32 vcc, %0, %1")
DEF_KERNEL(k_cmpabs, "v_cmp_eq_f32_e64 s[10:11], |%0|, %1")
DEF_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(k_max3, "v_max3_f32 %0, %0, |%1|, |%2|")
DEF_KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
DEF_KERNEL(k_mov, "v_mov_b32 %0, %1")
DEF_KERNEL(k_cvtpku8, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
DEF_KERNEL(k_readlane, "v_readlane_b32 s10, %0, 3")
DEF_KERNEL64(k_fma64, "v_fma_f64 %0, %0, %1, %2")
DEF_KERNEL64(k_add64, "v_add_f64 %0, %0, %1")
DEF_KERNEL64(k_mul64, "v_mul_f64 %0, %0, %1")
DEF_KERNEL64(k_trunc64, "v_trunc_f64 %0, %0")

DEF_KERNEL(k_sub, "v_sub_f32 %0, %0, %1")
DEF_KERNEL(k_max, "v_max_f32 %0, %0, %1")
DEF_KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
DEF_KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0")
DEF_KERNEL(k_addu, "v_add_u32 %0, %0, %1")
DEF_KERNEL(k_or, "v_or_b32 %0, %0, %1")
DEF_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEF_KERNEL(k_add_e64abs, "v_add_f32_e64 %0, |%0|, %1")
DEF_KERNEL(k_mul_lit, "v_mul_f32 %0, 0x3a83126f, %0")
DEF_KERNEL(k_add_sgpr, "v_add_f32 %0, %3, %0")
DEF_KERNEL(k_fmamk, "v_fmamk_f32 %0, %0, 0x43958000, %1")
DEF_KERNEL(k_fmac_lit, "v_fmac_f32 %0, 0x4412c000, %1")
DEF_KERNEL(k_fma_sgpr, "v_fma_f32 %0, %0, %3, %1")
DEF_KERNEL(k_cmplt32, "v_cmp_lt_f32 vcc, %0, %1")
DEF_KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %2")
DEF_KERNEL(k_maxi, "v_max_i32 %0, %0, %1")
DEF_KERNEL(k_ashr, "v_ashrrev_i32 %0, 3, %0")
DEF_KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 8, %1")
DEF_KERNEL(k_andor, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL(k_cvtub0, "v_cvt_f32_ubyte0 %0, %0")
DEF_KERNEL(k_cvtu32, "v_cvt_f32_u32 %0, %0")
DEF_KERNEL(k_mul_sdwa, "v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
DEF_KERNEL(k_pkmadu16, "v_pk_mad_u16 %0, %0, %1, %2")
DEF_KERNEL(k_pkfmaf16, "v_pk_fma_f16 %0, %0, %1, %2")
DEF_KERNEL(k_pkaddf16, "v_pk_add_f16 %0, %0, %1")
DEF_KERNEL(k_sad, "v_sad_u8 %0, %0, %1, %2")

typedef void (*kern_t)(float*, float);
static void run(const char* name, kern_t k)
{
    const int w = 8, blocks = 256 * w;   // 256 CUs x (w workgroups of 4 waves) => w waves per SIMD
    float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.5f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.5f); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)N_ITER * 16 * w;
    printf("%-28s %.2f ns per wave-instruction per SIMD\n", name, ms * 1e6 / n); fflush(stdout);
    (void)hipFree(out);
}
#define RUN(K) run(#K, K);
int main()
{
    RUN(k_fma) RUN(k_fmac) RUN(k_add) RUN(k_mul) RUN(k_pkfma) RUN(k_pkadd) RUN(k_pkmul) RUN(k_cvtub) RUN(k_trunc) RUN(k_rndne)
    RUN(k_fract) RUN(k_cvti) RUN(k_cvtf) RUN(k_pkrtz) RUN(k_cvtf16) RUN(k_perm) RUN(k_dot4) RUN(k_dot2) RUN(k_dot2c) RUN(k_mad24)
    RUN(k_mul24) RUN(k_mullo) RUN(k_mulhi) RUN(k_lshladd) RUN(k_add3) RUN(k_bfe) RUN(k_and) RUN(k_sdwa) RUN(k_cvti_sdwa) RUN(k_dpp)
    RUN(k_cmp) RUN(k_cmpabs) RUN(k_cndmask) RUN(k_max3) RUN(k_med3) RUN(k_mov) RUN(k_cvtpku8)
    RUN(k_sub) RUN(k_max) RUN(k_lshl) RUN(k_lshr) RUN(k_addu) RUN(k_or) RUN(k_xor) RUN(k_add_e64abs) RUN(k_mul_lit) RUN(k_add_sgpr)
    RUN(k_fmamk) RUN(k_fmac_lit) RUN(k_fma_sgpr) RUN(k_cmplt32) RUN(k_min3) RUN(k_maxi) RUN(k_ashr) RUN(k_lshlor) RUN(k_andor)
    RUN(k_cvtub0) RUN(k_cvtu32) RUN(k_mul_sdwa) RUN(k_pkmadu16) RUN(k_pkfmaf16) RUN(k_pkaddf16) RUN(k_sad)
    RUN(k_fma64) RUN(k_add64) RUN(k_mul64) RUN(k_trunc64)
    RUN(k_readlane)
    return 0;
}

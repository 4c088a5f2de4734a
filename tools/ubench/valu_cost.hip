// Micro-benchmark: cost of gfx950 VALU instruction FORMS as the encode kernel uses them -- distinct source and destination
// registers, a dependency distance of several instructions, W waves per SIMD.  Inline asm, so nothing is fused, dropped or
// re-selected.  ns per wave-instruction per SIMD = launch time / (instructions issued by one SIMD's waves).
//   hipcc -O3 --offload-arch=gfx950 -o valu_cost valu_cost.hip && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_ITER 1024

// 16 statements per iteration: d = f[i], sources f[(i + 5) & 15] and f[(i + 11) & 15] -- the result of a statement is read
// five and eleven statements later
#define BODY1(ASM)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                           \
        asm volatile(ASM : "+v"(f[i]) : "v"(f[(i + 5) & 15]), "v"(f[(i + 11) & 15]), "s"(sg), "s"(sp));
#define DEF1(NAME, ASM)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, float seed)                                     \
    {                                                                                                       \
        float f[16];                                                                                        \
        float sg = seed * 5.f; f2 sp = { seed * 3.f, seed * 7.f };                                         \
        for (int i = 0; i < 16; ++i) f[i] = seed * i + threadIdx.x;                                        \
        for (int it = 0; it < N_ITER; ++it) { BODY1(ASM) }                                                  \
        float s = 0;                                                                                        \
        for (int i = 0; i < 16; ++i) s += f[i];                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                     \
    }
#define BODY1C(ASM)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                           \
        asm volatile(ASM : "+v"(f[i]) : "v"(f[(i + 5) & 15]), "v"(f[(i + 11) & 15]), "s"(sg), "s"(sp) : "vcc", "s10", "s11");
#define DEF1C(NAME, ASM)                                                                                    \
    __global__ __launch_bounds__(256) void NAME(float* out, float seed)                                     \
    {                                                                                                       \
        float f[16];                                                                                        \
        float sg = seed * 5.f; f2 sp = { seed * 3.f, seed * 7.f };                                         \
        for (int i = 0; i < 16; ++i) f[i] = seed * i + threadIdx.x;                                        \
        for (int it = 0; it < N_ITER; ++it) { BODY1C(ASM) }                                                 \
        float s = 0;                                                                                        \
        for (int i = 0; i < 16; ++i) s += f[i];                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                     \
    }
#define BODY2(ASM)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                           \
        asm volatile(ASM : "+v"(f[i]) : "v"(f[(i + 5) & 15]), "v"(f[(i + 11) & 15]), "s"(sg), "s"(sp));
#define DEF2(NAME, ASM)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, float seed)                                     \
    {                                                                                                       \
        f2 f[16];                                                                                           \
        float sg = seed * 5.f; f2 sp = { seed * 3.f, seed * 7.f };                                         \
        for (int i = 0; i < 16; ++i) f[i] = f2{ seed * i + threadIdx.x, seed - i };                        \
        for (int it = 0; it < N_ITER; ++it) { BODY2(ASM) }                                                  \
        float s = 0;                                                                                        \
        for (int i = 0; i < 16; ++i) s += f[i].x + f[i].y;                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                     \
    }

DEF1(add_vv, "v_add_f32 %0, %1, %2")
DEF1(sub_vv, "v_sub_f32 %0, %1, %2")
DEF1(mul_vv, "v_mul_f32 %0, %1, %2")
DEF1(mul_lit, "v_mul_f32 %0, 0x3f6c835e, %1")
DEF1(add_sv, "v_add_f32 %0, %3, %1")
DEF1(fmamk, "v_fmamk_f32 %0, %1, 0x3f6c835e, %2")
DEF1(fmaak, "v_fmaak_f32 %0, %1, %2, 0x3f6c835e")
DEF1(fmac_vv, "v_fmac_f32 %0, %1, %2")
DEF1(fmac_sv, "v_fmac_f32 %0, %3, %1")
DEF1(fma_vvv, "v_fma_f32 %0, %1, %2, %0")
DEF1(fma_vsv, "v_fma_f32 %0, %1, %3, %2")
DEF1(fma_negv, "v_fma_f32 %0, -%1, %3, %2")
DEF2(pk_add, "v_pk_add_f32 %0, %1, %2")
DEF2(pk_add_mod, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]")
DEF2(pk_mul_vv, "v_pk_mul_f32 %0, %1, %2")
DEF2(pk_mul_vs, "v_pk_mul_f32 %0, %1, %4 op_sel:[1,0] op_sel_hi:[1,1]")
DEF2(pk_fma_vvv, "v_pk_fma_f32 %0, %1, %2, %0")
DEF2(pk_fma_vsv, "v_pk_fma_f32 %0, %1, %4, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_hi:[0,1,0]")
DEF2(pk_fma_bcast, "v_pk_fma_f32 %0, %1, %4, %2 op_sel_hi:[1,0,1]")
DEF1(cvt_ub0, "v_cvt_f32_ubyte0 %0, %1")
DEF1(cvt_ub2, "v_cvt_f32_ubyte2 %0, %1")
DEF1(trunc, "v_trunc_f32 %0, %1")
DEF1(fract, "v_fract_f32 %0, %1")
DEF1(rndne, "v_rndne_f32 %0, %1")
DEF1(cvt_i32, "v_cvt_i32_f32 %0, %1")
DEF1(cvt_u32, "v_cvt_u32_f32 %0, %1")
DEF1(min_vv, "v_min_f32 %0, %1, %2")
DEF1(min3, "v_min3_f32 %0, %1, %2, %0")
DEF1(max3_abs, "v_max3_f32 %0, |%1|, |%2|, %0")
DEF1(med3, "v_med3_f32 %0, %1, %3, %2")
DEF1(add_abs, "v_add_f32_e64 %0, |%1|, -0.5")
DEF1C(cndmask, "v_cndmask_b32 %0, %1, %2, s[10:11]")
DEF1(sdwa_add, "v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
DEF1(mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xa")
DEF1(mov, "v_mov_b32 %0, %1")
DEF1(and_vv, "v_and_b32 %0, %1, %2")
DEF1(bfe, "v_bfe_u32 %0, %1, 8, 8")
DEF1(lshl_add, "v_lshl_add_u32 %0, %1, 3, %2")
DEF1(add_u32, "v_add_u32 %0, %1, %2")
DEF1C(cmp_lt, "v_cmp_lt_f32 vcc, %1, %2")
DEF1C(cmp_lt_s, "v_cmp_lt_f32_e64 s[10:11], %1, %2")
DEF1(perm, "v_perm_b32 %0, %1, %2, %0")
DEF1(cvt_pk_i16, "v_cvt_pk_i16_i32 %0, %1, %2")
DEF1(fma_mix_lo, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]")
DEF1(fma_mix_hi, "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
DEF1(fma_mix_s, "v_fma_mix_f32 %0, %1, %3, %2 op_sel_hi:[1,0,0]")
DEF1(fma_f64_dummy, "v_mul_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %2")     // two dependent full-rate instructions

typedef void (*kern_t)(float*, float);
static void run(const char* name, kern_t k, int w, int per_stmt)
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
    const double n = (double)N_ITER * 16 * per_stmt * w;
    printf("%-16s w=%d  %.2f ns per wave-instruction per SIMD\n", name, w, best * 1e6 / n); fflush(stdout);
    (void)hipFree(out);
}
#define RUN(K) run(#K, K, w, 1);
int main()
{
    for (int w : { 1, 3, 5, 8 }) {
        RUN(add_vv) RUN(sub_vv) RUN(mul_vv) RUN(mul_lit) RUN(add_sv) RUN(fmamk) RUN(fmaak) RUN(fmac_vv) RUN(fmac_sv) RUN(fma_vvv)
        RUN(fma_vsv) RUN(fma_negv) RUN(pk_add) RUN(pk_add_mod) RUN(pk_mul_vv) RUN(pk_mul_vs) RUN(pk_fma_vvv) RUN(pk_fma_vsv) RUN(pk_fma_bcast)
        RUN(cvt_ub0) RUN(cvt_ub2) RUN(trunc) RUN(fract) RUN(rndne) RUN(cvt_i32) RUN(cvt_u32) RUN(min_vv) RUN(min3) RUN(max3_abs) RUN(med3)
        RUN(add_abs) RUN(cndmask) RUN(sdwa_add) RUN(mov_dpp) RUN(mov) RUN(and_vv) RUN(bfe) RUN(lshl_add) RUN(add_u32) RUN(cmp_lt)
        RUN(cmp_lt_s) RUN(perm) RUN(cvt_pk_i16)
        run("mul+add (2)", fma_f64_dummy, w, 2);
        RUN(fma_mix_lo) RUN(fma_mix_hi) RUN(fma_mix_s)
    }
    return 0;
}

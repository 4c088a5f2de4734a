// mfma_f16_numerics.hip -- how does v_mfma_f32_16x16x32_f16 accumulate?  The ISA documents the f32-input MFMA as a k-ordered
// fmaf chain; for f16 inputs (products exact in FP32) nothing is documented.  This probe compares D = A*B + C with
//   (1) RN(C + exact sum of the 32 products)          -- one rounding per instruction
//   (2) a k-ordered FP32 fmaf chain starting from C   -- 32 roundings
//   (3) RN(C + RN(exact sum))                          -- products summed exactly, then one more rounding with C
// on random data of the shape an FDCT-by-MFMA would use (integer samples |x| <= 128 against f16 limbs of cosine products)
// and on structured cases built to separate the models (one large term + many half-ulp terms; cancellation).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f16_numerics mfma_f16_numerics.hip && ./mfma_f16_numerics
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// A[16][32], B[32][16] row-major f16, C/D[16][16] row-major f32.  Lane maps: cdna_hip_programming.md section 3.
__global__ void mfma_kernel(const _Float16* A, const _Float16* B, const float* C, float* D, int ntile)
{
    const int lane = threadIdx.x & 63;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const _Float16* a = A + (size_t)t * 16 * 32;
        const _Float16* b = B + (size_t)t * 32 * 16;
        h8 fa, fb;
        for (int j = 0; j < 8; ++j) {
            fa[j] = a[(lane & 15) * 32 + 8 * (lane >> 4) + j];
            fb[j] = b[(8 * (lane >> 4) + j) * 16 + (lane & 15)];
        }
        f4 c;
        for (int r = 0; r < 4; ++r) c[r] = C[(size_t)t * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)];
        const f4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D[(size_t)t * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = d[r];
    }
}

static float h2f(_Float16 h) { return (float)h; }

int main()
{
    const int NT = 4096;                    // tiles of random data
    std::vector<_Float16> A((size_t)NT * 512), B((size_t)NT * 512);
    std::vector<float> C((size_t)NT * 256), D((size_t)NT * 256);
    std::mt19937_64 rng(12345);
    std::uniform_int_distribution<int> px(-128, 127);
    std::uniform_real_distribution<double> uc(-1.0, 1.0);
    for (int t = 0; t < NT; ++t) {
        const int kind = t % 4;             // 0: hi limbs, C = 0; 1: hi limbs, C ~ partial sum; 2: lo limbs on a large C; 3: wide dynamic range
        for (int i = 0; i < 512; ++i) {
            A[(size_t)t * 512 + i] = (_Float16)(float)px(rng);
            double g = uc(rng) * 0.025;
            if (kind == 2) g *= 0x1p-11;
            if (kind == 3) g = std::ldexp(uc(rng), -(int)(rng() % 14));
            B[(size_t)t * 512 + i] = (_Float16)(float)g;
        }
        for (int i = 0; i < 256; ++i) C[(size_t)t * 256 + i] = kind == 0 ? 0.f : (float)(uc(rng) * 40.0);
    }
    // structured tiles at the end: tile NT-1: row r, col c: A row = [4096? no: f16 max 65504] ...
    // S1: one product 2^11 * 2^1 = 4096 and 31 products of 2^-12 (= half an ulp of 4096 in FP32) -> exact 4096 + 31 * 2^-12
    //     one rounding: 4096 + 16 ulp (15.5 -> tie to even 16);  fmaf chain in k order: stays 4096 (every add is a tie -> even)
    // S2: the same with the large product LAST (k = 31)
    // S3: cancellation: +2^12, 30 x 2^-12, -2^12 -> exact 30 * 2^-12
    const int S = NT - 1;
    for (int i = 0; i < 512; ++i) { A[(size_t)S * 512 + i] = (_Float16)0.f; B[(size_t)S * 512 + i] = (_Float16)0.f; }
    for (int i = 0; i < 256; ++i) C[(size_t)S * 256 + i] = 0.f;
    auto setA = [&](int row, int k, float v) { A[(size_t)S * 512 + row * 32 + k] = (_Float16)v; };
    auto setB = [&](int k, int col, float v) { B[(size_t)S * 512 + k * 16 + col] = (_Float16)v; };
    for (int k = 0; k < 32; ++k) setB(k, 0, k == 0 ? 2.f : 0x1p-6f), setA(0, k, k == 0 ? 2048.f : 0x1p-6f);                 // S1 at (0,0)
    for (int k = 0; k < 32; ++k) setB(k, 1, k == 31 ? 2.f : 0x1p-6f), setA(1, k, k == 31 ? 2048.f : 0x1p-6f);               // S2 at (1,1)
    for (int k = 0; k < 32; ++k) setB(k, 2, (k == 0 || k == 31) ? 2.f : 0x1p-6f), setA(2, k, k == 0 ? 2048.f : k == 31 ? -2048.f : 0x1p-6f);   // S3 at (2,2)

    _Float16 *dA, *dB;
    float *dC, *dD;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, C.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_kernel, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, NT);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));

    long n = 0, eq1 = 0, eq2 = 0, eq3 = 0;
    double max_ulp = 0, max_rel_sumabs = 0;
    long per_kind_bad1[4] = { 0, 0, 0, 0 };
    for (int t = 0; t < NT - 1; ++t)
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                double ex = 0, sumabs = 0;
                float chain = C[(size_t)t * 256 + r * 16 + c];
                for (int k = 0; k < 32; ++k) {
                    const float a = h2f(A[(size_t)t * 512 + r * 32 + k]), b = h2f(B[(size_t)t * 512 + k * 16 + c]);
                    ex += (double)a * (double)b;            // exact: 22-bit products, 32 terms
                    sumabs += std::fabs((double)a * (double)b);
                    chain = std::fmaf(a, b, chain);
                }
                const double c0 = C[(size_t)t * 256 + r * 16 + c];
                const float m1 = (float)(c0 + ex), m3 = (float)(c0 + (double)(float)ex), got = D[(size_t)t * 256 + r * 16 + c];
                ++n;
                eq1 += got == m1; eq2 += got == chain; eq3 += got == m3;
                if (got != m1) ++per_kind_bad1[t % 4];
                const double exact = c0 + ex, err = std::fabs((double)got - exact);
                const double ulp = std::ldexp(1.0, std::ilogb(std::fabs(exact) > 0 ? std::fabs(exact) : 1e-30) - 23);
                if (err / ulp > max_ulp) max_ulp = err / ulp;
                sumabs += std::fabs(c0);
                if (sumabs > 0 && err / sumabs > max_rel_sumabs) max_rel_sumabs = err / sumabs;
            }
    printf("random tiles: %ld results; equal to (1) one rounding of the exact sum: %ld; (2) k-ordered fmaf chain: %ld; (3) RN(C + RN(sum)): %ld\n", n, eq1, eq2, eq3);
    printf("  mismatches against (1) by data kind [hi,C=0 | hi,C~sum | lo limbs on large C | wide range]: %ld %ld %ld %ld\n", per_kind_bad1[0], per_kind_bad1[1], per_kind_bad1[2], per_kind_bad1[3]);
    printf("  worst error: %.3f ulp of the exact result; %.3g x (|C| + sum |a_k b_k|)   [2^-24 = %.3g]\n", max_ulp, max_rel_sumabs, 0x1p-24);
    const float s1 = D[(size_t)S * 256 + 0 * 16 + 0], s2 = D[(size_t)S * 256 + 1 * 16 + 1], s3 = D[(size_t)S * 256 + 2 * 16 + 2];
    printf("structured: S1 (large term first)  = 4096 + %g ulp   [exact-then-round: 16; fmaf chain: 0]\n", (s1 - 4096.f) / 0x1p-11f);
    printf("            S2 (large term last)   = 4096 + %g ulp   [exact-then-round: 16; fmaf chain: 16 (31 * 2^-12 exact, then tie)]\n", (s2 - 4096.f) / 0x1p-11f);
    printf("            S3 (cancellation)      = %g x 2^-12      [exact: 30]\n", s3 / 0x1p-12f);
    return 0;
}

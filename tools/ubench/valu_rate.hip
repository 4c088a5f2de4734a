// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU ops the jpezy kernels lean on.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 4096
#define UNROLL 16

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, long long* cycles, double seed)
{
    double a[8];
    float f[8];
    int n[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; f[i] = (float)a[i]; n[i] = (int)a[i]; }
    const double c1 = seed * 0.37, c2 = seed * 1.01;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int i = u & 7;
            if (OP == 0) a[i] = __builtin_fma(a[i], c1, c2);
            if (OP == 1) a[i] = a[i] * c1;
            if (OP == 2) a[i] = a[i] + c2;
            if (OP == 3) { n[i] = (int)a[i]; a[i] += 1.0; }   // cvt_i32_f64 + add
            if (OP == 4) { a[i] = (double)(unsigned)n[i]; n[i] += 3; }  // cvt_f64_u32 + int add
            if (OP == 5) f[i] = __builtin_fmaf(f[i], (float)c1, (float)c2);
            if (OP == 6) n[i] = (n[i] + 12345) & 0xFFFFFF;       // 2 int ops
            if (OP == 7) a[i] = __builtin_trunc(a[i]) + c2;      // trunc + add
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; float fs = 0; int ns = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += a[i]; fs += f[i]; ns += n[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + fs + ns;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int instr_per_iter, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 wave per SIMD) x waves_per_simd
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, cyc, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, cyc, 1.5);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double n_instr = (double)N_ITER * UNROLL * instr_per_iter;
    // s_memtime ticks at 100 MHz on gfx9 (REFCLK); use wall time instead, assume ~2.1-2.4 GHz
    const double wave_instr_per_simd = n_instr * waves_per_simd;
    printf("%-28s waves/SIMD=%d  time=%.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz, %.2f @2.0GHz)  memtime=%.0f\n",
           name, waves_per_simd, ms, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4,
           ms * 1e6 / wave_instr_per_simd * 2.0, avg);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", 1, w);
        run<1>("v_mul_f64", 1, w);
        run<2>("v_add_f64", 1, w);
        run<3>("v_cvt_i32_f64 + v_add_f64", 2, w);
        run<4>("v_cvt_f64_u32 + v_add_u32", 2, w);
        run<5>("v_fma_f32", 1, w);
        run<6>("v_add_u32 + v_and_b32", 2, w);
        run<7>("v_trunc_f64 + v_add_f64", 2, w);
    }
    return 0;
}

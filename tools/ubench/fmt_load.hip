// fmt_load.hip -- does gfx950's buffer_load_format_xyzw convert 8_8_8_8 USCALED bytes to floats in the load path, and what
// does a streaming read cost that way?  (a) global_load_dwordx4 + 16 v_cvt_f32_ubyte per lane, (b) 4 format loads per lane.
// Both sum the 16 pixels of a lane's segment into one float per lane (the consumer), planes of 4096 x 4096 bytes.
//   hipcc -O3 --offload-arch=gfx950 -o fmt_load fmt_load.hip && ./fmt_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ f4 raw_buffer_load_format_v4f32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// SRD word 3: DST_SEL xyzw = 4,5,6,7; NUM_FORMAT USCALED (2) << 12; DATA_FORMAT 8_8_8_8 (10) << 15
constexpr int SRD3_U8X4_USCALED = 0xFAC | (2 << 12) | (10 << 15);

__global__ void check_kernel(const unsigned char* p, float* out, unsigned n)
{
    const unsigned long long a = (unsigned long long)p;
    const i4 r = { (int)(unsigned)a, (int)(unsigned)(a >> 32), (int)n, SRD3_U8X4_USCALED };
    const f4 v = raw_buffer_load_format_v4f32(r, threadIdx.x * 4, 0, 0);
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}

template <int B> __device__ __forceinline__ float ub(unsigned w) { return (float)((w >> (8 * B)) & 0xFFu); }

__global__ __launch_bounds__(128) void sum_cvt(const unsigned char* r, const unsigned char* g, const unsigned char* b, float* out, unsigned n16)
{
    const unsigned i = blockIdx.x * 128 + threadIdx.x;
    if (i >= n16) return;
    float s = 0;
    const unsigned char* pl[3] = { r, g, b };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint4 v = *reinterpret_cast<const uint4*>(pl[c] + (size_t)i * 16);
        const unsigned w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int k = 0; k < 4; ++k) s += (ub<0>(w[k]) * 0.299f + ub<1>(w[k]) * 0.587f) + (ub<2>(w[k]) * 0.114f + ub<3>(w[k]) * 0.5f);
    }
    out[i] = s;
}

__global__ __launch_bounds__(128) void sum_fmt(const unsigned char* r, const unsigned char* g, const unsigned char* b, float* out, unsigned n16)
{
    const unsigned i = blockIdx.x * 128 + threadIdx.x;
    if (i >= n16) return;
    float s = 0;
    const unsigned char* pl[3] = { r, g, b };
    f4 v[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned long long a = (unsigned long long)pl[c];
        const i4 rs = { (int)(unsigned)a, (int)(unsigned)(a >> 32), (int)(n16 * 16u), SRD3_U8X4_USCALED };
#pragma unroll
        for (int k = 0; k < 4; ++k) v[c][k] = raw_buffer_load_format_v4f32(rs, (int)(i * 16u + k * 4u), 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) s += (v[c][k].x * 0.299f + v[c][k].y * 0.587f) + (v[c][k].z * 0.114f + v[c][k].w * 0.5f);
    out[i] = s;
}

int main()
{
    // (1) correctness: every byte value
    unsigned char* d;
    float* o;
    CK(hipMalloc(&d, 256));
    CK(hipMalloc(&o, 256 * 4));
    std::vector<unsigned char> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (unsigned char)((i * 37 + 11) & 255);
    CK(hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, d, o, 256u);
    std::vector<float> ho(256);
    CK(hipMemcpy(ho.data(), o, 256 * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += ho[i] != (float)h[i];
    printf("format load 8_8_8_8 USCALED: %d of 256 values wrong (first: got %g %g %g %g want %d %d %d %d)\n", bad, ho[0], ho[1], ho[2], ho[3], h[0], h[1], h[2], h[3]);
    // (2) streaming cost
    const size_t plane = 4096ull * 4096;
    const unsigned n16 = (unsigned)(plane / 16);
    const int ring = 8;
    unsigned char* planes;
    float* out;
    CK(hipMalloc(&planes, plane * 3 * ring));
    CK(hipMalloc(&out, (size_t)n16 * 4));
    CK(hipMemset(planes, 0x5A, plane * 3 * ring));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; ++variant)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 40; ++it) {
                const unsigned char* base = planes + (size_t)(it % ring) * plane * 3;
                if (variant == 0)
                    hipLaunchKernelGGL(sum_cvt, dim3((n16 + 127) / 128), dim3(128), 0, 0, base, base + plane, base + 2 * plane, out, n16);
                else
                    hipLaunchKernelGGL(sum_fmt, dim3((n16 + 127) / 128), dim3(128), 0, 0, base, base + plane, base + 2 * plane, out, n16);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s: %.2f us per 4096^2 x 3 planes (%.0f GB/s)\n", variant ? "format loads (12 per lane)    " : "dwordx4 + 48 cvt_ubyte per lane", ms / 40 * 1e3,
                   3.0 * plane / (ms / 40 * 1e-3) / 1e9);
        }
    return bad != 0;
}

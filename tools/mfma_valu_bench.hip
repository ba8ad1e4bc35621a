// How much plain VALU / LDS work rides along with a bf16 MFMA stream on gfx950 (design input for
// the matching prefilter; not part of the product).  One workgroup of 8 waves per CU = 2 waves per
// SIMD, each wave loops over { 16 x v_mfma_f32_32x32x16_bf16 in two chains, NV x v_max3_f32 spread
// between them, NL x ds_read_b128 }.  Prints the time per loop iteration per SIMD in matrix-pipe
// cycles (16 MFMAs x 32 cycles x 2 waves = 1024 when nothing else costs).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma_valu_bench tools/mfma_valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// the same with the reads spread: read l is issued behind MFMA l of the iteration (one per gap),
// consumed at the end of the iteration
template <int NV, int NL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void kspread(const uint4 *in, float *out, int iters)
{
    __shared__ uint4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += WAVES * 64) lds[i] = in[i & 511];
    __syncthreads();
    uint4 ua = in[threadIdx.x & 255], ub = in[(threadIdx.x & 255) + 256];
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    f32x16 x0 = 0, x1 = 0;
    float m0 = 0.f, m1 = 1.f, m2 = 2.f, m3 = 3.f;
    uint4 acc4 = make_uint4(0, 0, 0, 0);
    const uint4 *lp = lds + (threadIdx.x & 63);
    uint4 f[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) f[l] = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < iters; ++i) {
        const unsigned addr = (unsigned)(size_t)lp + (i & 1) * 16384;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x0, 0, 0, 0);
            if (2 * s < NL) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[2 * s]) : "v"(addr), "n"(2 * s * 1024));
#pragma unroll
            for (int v = 0; v < NV / 16; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(m2), "v"(m3));
            __builtin_amdgcn_sched_barrier(0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x1, 0, 0, 0);
            if (2 * s + 1 < NL) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[2 * s + 1]) : "v"(addr), "n"((2 * s + 1) * 1024));
#pragma unroll
            for (int v = 0; v < NV / 16; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m1) : "v"(m2), "v"(m3));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int l = 0; l < NL; ++l) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc4.x) : "v"(f[l].x));
    }
    float s = m0 + m1 + (float)acc4.x + (float)acc4.y;
    for (int r = 0; r < 16; ++r) s += x0[r] + x1[r];
    if (s == 123.456f) out[0] = s;
}

template <int NV, int NL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const uint4 *in, float *out, int iters)
{
    __shared__ uint4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += WAVES * 64) lds[i] = in[i & 511];
    __syncthreads();
    uint4 ua = in[threadIdx.x & 255], ub = in[(threadIdx.x & 255) + 256];
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    f32x16 x0 = 0, x1 = 0;
    float m0 = 0.f, m1 = 1.f, m2 = 2.f, m3 = 3.f;
    uint4 acc4 = make_uint4(0, 0, 0, 0);
    const uint4 *lp = lds + (threadIdx.x & 63);
    for (int i = 0; i < iters; ++i) {
        uint4 f[NL > 0 ? NL : 1];
        const unsigned addr = (unsigned)(size_t)lp + (i & 1) * 16384;
#pragma unroll
        for (int l = 0; l < NL; ++l)  // (asm: the compiler would narrow the reads to the dwords used)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[l]) : "v"(addr), "n"((l & 15) * 1024));
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV / 16; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(m2), "v"(m3));
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x1, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV / 16; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m1) : "v"(m2), "v"(m3));
        }
        if (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int l = 0; l < NL; ++l) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc4.x) : "v"(f[l].x));
    }
    float s = m0 + m1 + (float)acc4.x + (float)acc4.y;
    for (int r = 0; r < 16; ++r) s += x0[r] + x1[r];
    if (s == 123.456f) out[0] = s;
}

// three chains per wave (a wave that owns three query groups): 24 MFMAs per iteration, NV v_max3, NL reads
template <int NV, int NL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k3(const uint4 *in, float *out, int iters)
{
    __shared__ uint4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += WAVES * 64) lds[i] = in[i & 511];
    __syncthreads();
    uint4 ua = in[threadIdx.x & 255], ub = in[(threadIdx.x & 255) + 256];
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    f32x16 x0 = 0, x1 = 0, x2 = 0;
    float m0 = 0.f, m1 = 1.f, m2 = 2.f, m3 = 3.f, m4 = 4.f;
    uint4 acc4 = make_uint4(0, 0, 0, 0);
    const uint4 *lp = lds + (threadIdx.x & 63);
    for (int i = 0; i < iters; ++i) {
        uint4 f[NL > 0 ? NL : 1];
        const unsigned addr = (unsigned)(size_t)lp + (i & 1) * 16384;
#pragma unroll
        for (int l = 0; l < NL; ++l)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[l]) : "v"(addr), "n"((l & 15) * 1024));
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV / 24; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(m2), "v"(m3));
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x1, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV / 24; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m1) : "v"(m2), "v"(m3));
            x2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x2, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV / 24; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m4) : "v"(m2), "v"(m3));
        }
        if (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int l = 0; l < NL; ++l) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc4.x) : "v"(f[l].x));
    }
    float s = m0 + m1 + m4 + (float)acc4.x + (float)acc4.y;
    for (int r = 0; r < 16; ++r) s += x0[r] + x1[r] + x2[r];
    if (s == 123.456f) out[0] = s;
}

template <int NV, int NL, int WAVES> static void run3(const uint4 *in, float *out, int cus)
{
    const int iters = 14000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k3<NV, NL, WAVES>), dim3(cus), dim3(WAVES * 64), 0, 0, in, out, iters / 2);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k3<NV, NL, WAVES>), dim3(cus), dim3(WAVES * 64), 0, 0, in, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * 32 * 32 * 16 * 24 * (double)iters * cus * WAVES;
    printf("THREE chains: waves/SIMD %d  v_max3 %3d  ds_read_b128 %2d per 24 MFMAs: %7.3f ms  %6.0f TFLOP/s\n", WAVES / 4, NV, NL, ms,
           flop / (ms * 1e-3) / 1e12);
}

template <int NV, int NL, int WAVES> static void run_spread(const uint4 *in, float *out, int cus)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kspread<NV, NL, WAVES>), dim3(cus), dim3(WAVES * 64), 0, 0, in, out, iters / 2);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((kspread<NV, NL, WAVES>), dim3(cus), dim3(WAVES * 64), 0, 0, in, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * 32 * 32 * 16 * 16 * (double)iters * cus * WAVES;
    printf("waves/SIMD %d  v_max3 %3d  ds_read_b128 %2d (one per MFMA gap): %7.3f ms  %6.0f TFLOP/s  %6.1f ns per iteration\n", WAVES / 4, NV, NL, ms,
           flop / (ms * 1e-3) / 1e12, ms * 1e6 / iters);
}

template <int NV, int NL, int WAVES> static void run(const uint4 *in, float *out, int cus, double ghz_hint)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NV, NL, WAVES>), dim3(cus), dim3(WAVES * 64), 0, 0, in, out, iters / 2);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NV, NL, WAVES>), dim3(cus), dim3(WAVES * 64), 0, 0, in, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * 32 * 32 * 16 * 16 * (double)iters * cus * WAVES;
    printf("waves/SIMD %d  v_max3 %3d  ds_read_b128 %2d : %7.3f ms  %6.0f TFLOP/s  %6.1f ns per iteration\n", WAVES / 4, NV, NL, ms,
           flop / (ms * 1e-3) / 1e12, ms * 1e6 / iters);
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint4 *in; float *out;
    CHECK(hipMalloc(&in, 512 * sizeof(uint4))); CHECK(hipMalloc(&out, 64));
    unsigned h[2048];
    for (int i = 0; i < 2048; ++i) h[i] = 0x3F803F80u ^ ((i * 2654435761u) & 0x007F007Fu);
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    run<0, 0, 8>(in, out, cus, 0); run<16, 0, 8>(in, out, cus, 0); run<32, 0, 8>(in, out, cus, 0); run<48, 0, 8>(in, out, cus, 0);
    run<64, 0, 8>(in, out, cus, 0); run<96, 0, 8>(in, out, cus, 0); run<128, 0, 8>(in, out, cus, 0);
    run<0, 12, 8>(in, out, cus, 0); run<48, 12, 8>(in, out, cus, 0); run<48, 24, 8>(in, out, cus, 0);
    run<0, 0, 4>(in, out, cus, 0); run<48, 0, 4>(in, out, cus, 0); run<96, 0, 4>(in, out, cus, 0); run<48, 12, 4>(in, out, cus, 0);
    run_spread<0, 4, 8>(in, out, cus); run_spread<0, 8, 8>(in, out, cus); run_spread<0, 12, 8>(in, out, cus); run_spread<48, 12, 8>(in, out, cus); run_spread<0, 16, 8>(in, out, cus);
    run_spread<0, 12, 4>(in, out, cus); run_spread<48, 12, 4>(in, out, cus);
    run<0, 0, 16>(in, out, cus, 0); run<48, 12, 16>(in, out, cus, 0);
    // round 3: would a wave with three query groups (12 reads per 24 MFMAs, the same vector work per MFMA) do better?
    run<48, 12, 8>(in, out, cus, 0); run3<72, 12, 8>(in, out, cus); run3<48, 12, 8>(in, out, cus); run3<96, 12, 8>(in, out, cus);
    run3<72, 0, 8>(in, out, cus); run3<0, 0, 8>(in, out, cus); run<64, 12, 8>(in, out, cus, 0); run3<96, 18, 8>(in, out, cus);
    return 0;
}

// MFMA issue-rate microbenchmark for gfx950 (design input for the matching prefilter; not
// part of the product): sustained bf16 MFMA rate of the whole chip under power limits.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma_bench tools/mfma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k32(const uint4 *in, float *out, int iters)
{
    uint4 ua = in[threadIdx.x], ub = in[threadIdx.x + 256];
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 123.456f) out[0] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void k16(const uint4 *in, float *out, int iters)
{
    uint4 ua = in[threadIdx.x], ub = in[threadIdx.x + 256];
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
    if (s == 123.456f) out[0] = s;
}

template <typename F> static void run(const char *name, F launch, double flop_per_wave_iter, int wgs_per_cu, int cus, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(iters / 4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch(iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)cus * wgs_per_cu * 4;
    const double tf = flop_per_wave_iter * iters * waves / (ms * 1e-3) / 1e12;
    printf("%-28s wg/CU %d  %8.3f ms  %8.1f TFLOP/s  (%.0f %% of 2500)\n", name, wgs_per_cu, ms, tf, tf / 25.0);
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device CUs %d clock %d kHz\n", cus, p.clockRate);
    uint4 *in; float *out;
    CHECK(hipMalloc(&in, 512 * sizeof(uint4))); CHECK(hipMalloc(&out, 64));
    for (int data = 0; data < 2; ++data) {
        unsigned h[2048];
        for (int i = 0; i < 2048; ++i) h[i] = data ? (0x3F803F80u ^ ((i * 2654435761u) & 0x007F007Fu)) : 0u;
        CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
        printf("operands: %s\n", data ? "random bf16 near 1.0" : "all zero");
        const int iters = 20000;
        for (int w = 1; w <= 2; ++w) {
            run("mfma_f32_32x32x16_bf16 x2", [&](int it) { hipLaunchKernelGGL(k32<2>, dim3(cus * w), dim3(256), 0, 0, in, out, it); }, 2.0 * 32 * 32 * 16 * 8 * 2, w, cus, iters);
            run("mfma_f32_32x32x16_bf16 x4", [&](int it) { hipLaunchKernelGGL(k32<4>, dim3(cus * w), dim3(256), 0, 0, in, out, it); }, 2.0 * 32 * 32 * 16 * 8 * 4, w, cus, iters);
            run("mfma_f32_16x16x32_bf16 x4", [&](int it) { hipLaunchKernelGGL(k16<4>, dim3(cus * w), dim3(256), 0, 0, in, out, it); }, 2.0 * 16 * 16 * 32 * 8 * 4, w, cus, iters);
            run("mfma_f32_16x16x32_bf16 x8", [&](int it) { hipLaunchKernelGGL(k16<8>, dim3(cus * w), dim3(256), 0, 0, in, out, it); }, 2.0 * 16 * 16 * 32 * 8 * 8, w, cus, iters);
        }
    }
    return 0;
}

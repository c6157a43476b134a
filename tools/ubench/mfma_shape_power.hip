// Micro-benchmark (gfx950, round 6): do the two f16 MFMA shapes cost the same power per FLOP?  One workgroup per CU, 8 waves, back-to-back
// v_mfma_f32_32x32x16_f16 (32 768 FLOPs, 4 accumulators of 16 registers) or v_mfma_f32_16x16x32_f16 (16 384 FLOPs, 8 accumulators of 4) on data-like
// operands, ~0.6 s per point (the package sits on its 1 400 W cap: the sustained rate IS the energy per FLOP).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_power mfma_shape_power.hip && ./mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE, int NACC = 8>
__global__ __launch_bounds__(512, 1) void k(float* sink, long long* clk, int steps) {
    const int tid = threadIdx.x;
    half8 a[4], b[4];
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i) {
            const unsigned h = (unsigned)(tid * 64 + q * 8 + i) * 2654435761u, g = h * 2246822519u + 12345u;
            a[q][i] = __builtin_bit_cast(_Float16, (unsigned short)(0x3800u + ((h >> 7) & 0x07ffu) + ((h >> 3) & 0x8000u)));
            b[q][i] = __builtin_bit_cast(_Float16, (unsigned short)(0x3800u + ((g >> 7) & 0x07ffu) + ((g >> 3) & 0x8000u)));
        }
    f16v acc32[4];
    f4v acc16[8];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) acc32[q][i] = 0.f;
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 4; ++i) acc16[q][i] = 0.f;
    const long long w0 = wall_clock64(), t0 = clock64();
    for (int st = 0; st < steps; ++st) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if (SHAPE == 32) acc32[m % (NACC > 4 ? 4 : NACC)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(m >> 2) & 3], b[m & 3], acc32[m % (NACC > 4 ? 4 : NACC)], 0, 0, 0);
            else {
                acc16[(2 * m) % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(m >> 2) & 3], b[m & 3], acc16[(2 * m) % NACC], 0, 0, 0);
                acc16[(2 * m + 1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(m >> 1) & 3], b[(m + 1) & 3], acc16[(2 * m + 1) % NACC], 0, 0, 0);
            }
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float r = 0.f;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) r += acc32[q][i];
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 4; ++i) r += acc16[q][i];
    if (r == 123.456f) sink[0] = r;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int SHAPE, int NACC = 8>
static void run(const char* name, float* sink, long long* clk, int ncu) {
    const int steps = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto launch = [&]() { hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(ncu), dim3(512), 0, 0, sink, clk, steps); };
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float one = 0.f; CHECK(hipEventElapsedTime(&one, e0, e1));
    const int settle = std::max(8, (int)(600.f / one)), timed = std::max(4, settle / 4);
    for (int i = 0; i < settle; ++i) launch();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < timed; ++i) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= timed;
    long long h[2];
    CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    const double flops = 8.0 * ncu * steps * 32 * 32768.0;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s  shader clock %.2f GHz\n", name, ms, flops / ms * 1e-9, (double)h[0] / ((double)h[1] * 10.0));
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount / 8 * 8;
    float* sink; long long* clk;
    CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&clk, 2 * ncu * 8));
    run<32>("v_mfma_f32_32x32x16_f16", sink, clk, ncu);
    run<16>("v_mfma_f32_16x16x32_f16", sink, clk, ncu);
    run<32>("v_mfma_f32_32x32x16_f16", sink, clk, ncu);
    run<16>("v_mfma_f32_16x16x32_f16", sink, clk, ncu);
    // how far apart must two 16x16x32 MFMAs on the SAME accumulator be?  (2 waves per SIMD: the partner fills what it can)
    run<16, 4>("16x16x32, 4 accumulators in rotation", sink, clk, ncu);
    run<16, 3>("16x16x32, 3 accumulators", sink, clk, ncu);
    run<16, 2>("16x16x32, 2 accumulators", sink, clk, ncu);
    run<16, 1>("16x16x32, 1 accumulator", sink, clk, ncu);
    run<32, 4>("32x32x16, 4 accumulators in rotation", sink, clk, ncu);
    run<32, 2>("32x32x16, 2 accumulators", sink, clk, ncu);
    run<32, 1>("32x32x16, 1 accumulator", sink, clk, ncu);
    return 0;
}

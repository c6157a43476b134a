// Micro-benchmark (gfx950): v_mfma_f32_32x32x2_f32 rate of ONE wave per SIMD as a function of the number of
// independent accumulators it rotates through (1 = every MFMA depends on the previous one).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(int iters, long long* out, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 a[4] = {};
    float x = lane * 0.5f, y = lane * 0.25f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a[j % NACC]) : "v"(x), "v"(y));
    }
    const long long t1 = clock64();
    float s = 0;
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += a[n][r];
    if (s == 12345.f) sink[0] = s;
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(long long* out, float* sink) {
    const int blocks = 256, iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(256), 0, 0, iters, out, sink);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    (void)hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%d accumulator(s) in rotation: %.1f cycles per MFMA\n", NACC, s / blocks / (iters * 8.0));
}

int main() {
    long long* out; float* sink;
    (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&sink, 64);
    run<1>(out, sink); run<2>(out, sink); run<4>(out, sink);
    return 0;
}

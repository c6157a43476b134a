// Micro-benchmark (gfx950): what does a wave pay for non-MFMA instructions placed between its own
// back-to-back independent v_mfma_f32_32x32x2_f32 (64 cycles each)?  One wave per SIMD.
// Per iteration: 4 MFMAs + NV v_fma (independent of the MFMAs) + NL ds_read_b32, interleaved.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NL, int SPREAD>
__global__ __launch_bounds__(256) void k(int iters, long long* out, float* sink) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = lane * 0.5f, y = lane * 0.25f;
    float v[4] = {1, 2, 3, 4}, l[4] = {0, 0, 0, 0};
    const unsigned p = (unsigned)(uintptr_t)(lds + lane);
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#define MF(a) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y))
#define VA(j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j]))
#define LD(j) asm volatile("ds_read_b32 %0, %1 offset:%c2" : "=v"(l[j]) : "v"(p), "i"(j * 256))
        if (SPREAD) {   // non-MFMA work spread evenly behind each MFMA
            MF(a0); for (int j = 0; j < NV / 4; ++j) VA(j & 3); for (int j = 0; j < NL / 4; ++j) LD(j & 3);
            MF(a1); for (int j = 0; j < NV / 4; ++j) VA(j & 3); for (int j = 0; j < NL / 4; ++j) LD(j & 3);
            MF(a2); for (int j = 0; j < NV / 4; ++j) VA(j & 3); for (int j = 0; j < NL / 4; ++j) LD(j & 3);
            MF(a3); for (int j = 0; j < NV / 4; ++j) VA(j & 3); for (int j = 0; j < NL / 4; ++j) LD(j & 3);
        } else {        // all of it behind the first MFMA
            MF(a0); for (int j = 0; j < NV; ++j) VA(j & 3); for (int j = 0; j < NL; ++j) LD(j & 3);
            MF(a1); MF(a2); MF(a3);
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = clock64();
    float s = v[0] + v[1] + v[2] + v[3] + l[0] + l[1] + l[2] + l[3];
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.f) sink[0] = s;
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}

template <int NV, int NL, int SPREAD>
void run(long long* out, float* sink) {
    const int blocks = 256, iters = 4000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NV, NL, SPREAD>), dim3(blocks), dim3(256), 0, 0, iters, out, sink);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    (void)hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("4 MFMA + %2d VALU + %2d ds_read (%s): %.1f cycles per iteration (MFMA bound 256)\n", NV, NL, SPREAD ? "spread" : "clumped", s / blocks / iters);
}

int main() {
    long long* out; float* sink;
    (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&sink, 64);
    run<0, 0, 1>(out, sink);
    run<4, 0, 1>(out, sink);
    run<12, 0, 1>(out, sink);
    run<24, 0, 1>(out, sink);
    run<48, 0, 1>(out, sink);
    run<0, 8, 1>(out, sink);
    run<0, 16, 1>(out, sink);
    run<12, 8, 1>(out, sink);
    run<12, 0, 0>(out, sink);
    run<12, 8, 0>(out, sink);
    return 0;
}

// Micro-benchmark (gfx950): how fast does a VALU / LDS wave run while ANOTHER wave on the same SIMD
// streams back-to-back v_mfma_f32_32x32x2_f32?  waves 0-3: MFMA stream (if mf); waves 4-7: measured.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int mf, int mode, int iters, long long* out, float* sink) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 512) lds[i] = (float)i;
    __syncthreads();
    if (wave < 4) {
        if (!mf) return;
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        float x = lane * 0.5f, y = lane * 0.25f;
        for (int i = 0; i < iters * 3; ++i)
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\tv_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n\tv_mfma_f32_32x32x2_f32 %3, %4, %5, %3\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
        float s = 0;
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
        if (s == 12345.f) sink[0] = s;
        return;
    }
    const long long t0 = clock64();
    if (mode == 0) {          // 16 independent VALU per iteration
        float v0 = lane, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
        for (int i = 0; i < iters; ++i)
            asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                         "v_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\t"
                         "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                         "v_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\t"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 12345.f) sink[1] = v0;
    } else if (mode == 1) {   // 16 ds_write_b32 per iteration, conflict-free
        const unsigned p = (unsigned)(uintptr_t)(lds + lane);
        float v = lane;
        for (int i = 0; i < iters; ++i) {
            for (int j = 0; j < 16; ++j) asm volatile("ds_write_b32 %0, %1 offset:%c2" ::"v"(p), "v"(v), "i"(j * 256) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
    } else {                  // 16 ds_read_b32 per iteration
        const unsigned p = (unsigned)(uintptr_t)(lds + lane);
        float l[4] = {};
        for (int i = 0; i < iters; ++i) {
            for (int j = 0; j < 16; ++j) asm volatile("ds_read_b32 %0, %1 offset:%c2" : "=v"(l[j & 3]) : "v"(p), "i"(j * 256));
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        if (l[0] + l[1] + l[2] + l[3] == 12345.f) sink[1] = l[0];
    }
    const long long t1 = clock64();
    if (lane == 0 && wave == 4) out[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256, iters = 2000;
    long long* out; float* sink;
    (void)hipMalloc(&out, blocks * 8); (void)hipMalloc(&sink, 64);
    const char* names[] = {"16 VALU", "16 ds_write_b32", "16 ds_read_b32"};
    for (int mode = 0; mode < 3; ++mode)
        for (int mf = 0; mf < 2; ++mf) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, mf, mode, iters, out, sink);
            (void)hipDeviceSynchronize();
            std::vector<long long> h(blocks);
            (void)hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            printf("%-16s per iteration, MFMA stream on the same SIMD %s: %.1f cycles\n", names[mode], mf ? "ON " : "off", s / blocks / iters);
        }
    return 0;
}

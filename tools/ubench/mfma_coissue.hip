// Micro-benchmark (gfx950): how much does a second wave on the same SIMD slow a wave that issues
// back-to-back independent v_mfma_f32_32x32x2_f32?  Workgroup = 8 waves; waves 0-3 (one per SIMD) run
// the MFMA loop, waves 4-7 run the interference pattern `mode`:
//   0 none   1 VALU (independent v_fma)   2 ds_read_b32 conflict-free   3 ds_write_b32 4-way conflicts
//   4 global loads (L2 hits)   5 second MFMA stream   6 VALU with dependent chain
// prints cycles per MFMA of wave 0 (ideal 64).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue mfma_coissue.hip && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int mode, int iters, long long* out, const float* gsrc, float* sink) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 512) lds[i] = (float)i;
    __syncthreads();
    if (wave < 4 || mode == 5) {
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        float x = lane * 0.5f, y = lane * 0.25f;
        const long long t0 = clock64();
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n\t"
                         "v_mfma_f32_32x32x2_f32 %3, %4, %5, %3\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
        }
        const long long t1 = clock64();
        float s = 0;
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
        if (s == 12345.f) sink[0] = s;
        if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
        return;
    }
    // interference waves: run ~2x as long as the MFMA loop so it is covered entirely
    const int n = iters * 4;
    if (mode == 1) {
        float v0 = lane, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
        for (int i = 0; i < n * 8; ++i) {
            asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                         "v_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\t"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        }
        if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 12345.f) sink[1] = v0;
    } else if (mode == 6) {
        float v0 = lane;
        for (int i = 0; i < n * 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0" : "+v"(v0));
        if (v0 == 12345.f) sink[1] = v0;
    } else if (mode == 2) {
        float acc = 0;
        const float* p = lds + lane;
        for (int i = 0; i < n * 4; ++i) {
            float a, b, c, d;
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"((unsigned)(uintptr_t)p) : "memory");
            acc += a + b + c + d;
        }
        if (acc == 12345.f) sink[1] = acc;
    } else if (mode == 3) {
        const unsigned p = (unsigned)(uintptr_t)(lds + (lane & 7) + (lane >> 3) * 32 * 4 % 4096);   // 8 banks used by 64 lanes
        const unsigned q = (unsigned)(uintptr_t)(lds + (lane * 128) % 8192);                        // all lanes one bank: worst case
        (void)q;
        float v = lane;
        for (int i = 0; i < n * 4; ++i)
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1 offset:4096\n\tds_write_b32 %0, %1 offset:8192\n\tds_write_b32 %0, %1 offset:12288\n\ts_waitcnt lgkmcnt(0)" ::"v"(p), "v"(v) : "memory");
    } else if (mode == 4) {
        float acc = 0;
        const float4* g = reinterpret_cast<const float4*>(gsrc) + lane + wave * 64;
        for (int i = 0; i < n; ++i) {
            float4 a = g[(i & 15) * 512], b = g[(i & 15) * 512 + 4096];
            acc += a.x + b.y;
        }
        if (acc == 12345.f) sink[1] = acc;
    }
}

int main() {
    const int blocks = 256, iters = 4000;
    long long* out; float *gsrc, *sink;
    hipMalloc(&out, blocks * 8); hipMalloc(&gsrc, 1 << 22); hipMalloc(&sink, 64);
    hipMemset(gsrc, 0, 1 << 22);
    const char* names[] = {"none", "VALU indep", "ds_read", "ds_write 8-bank", "global loads", "2nd MFMA wave", "VALU dep chain"};
    for (int mode = 0; mode < 7; ++mode) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, mode, iters, out, gsrc, sink);
        hipDeviceSynchronize();
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += v;
        printf("mode %d (%-16s): %.1f cycles per MFMA (wave 0, mean over %d workgroups)\n", mode, names[mode], s / blocks / (iters * 4.0), blocks);
    }
    return 0;
}

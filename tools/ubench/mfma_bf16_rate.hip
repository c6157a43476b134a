// Micro-benchmark (gfx950): v_mfma_f32_32x32x16_bf16 rate per SIMD as a function of
//   NACC  - independent accumulators a wave rotates through (1 = every MFMA depends on the previous one),
//   WPS   - waves per SIMD streaming MFMAs (1 or 2),
//   LDSOP - operands taken from registers (0) or re-read from LDS (2 ds_read_b128 per MFMA) 1, 2 or 3 MFMAs ahead.
// Question behind it: the bf16 kernels of this repo reach 41-55 % matrix-pipe use whatever their instruction order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int LDSOP>
__global__ __launch_bounds__(512) void k(int iters, long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[32768];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(sm)[i] = 0x3f803f80u;
    __syncthreads();
    f32x16 a[4] = {};
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(lane * 0.01f); y[e] = (__bf16)0.5f; }
    const unsigned char* p = sm + lane * 16;
    bf16x8 qx[4] = {x, x, x, x}, qy[4] = {y, y, y, y};            // LDSOP = depth of the operand queue (steps ahead), 0 = registers
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (LDSOP) {
                qx[(j + LDSOP) & 3] = *reinterpret_cast<const bf16x8*>(p + ((j * 2) & 15) * 1024);
                qy[(j + LDSOP) & 3] = *reinterpret_cast<const bf16x8*>(p + ((j * 2 + 1) & 15) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
            a[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qx[j & 3], qy[j & 3], a[j % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += a[n][r];
    if (s == 12345.f) sink[0] = s;
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}

template <int NACC, int LDSOP>
void run(int wps, long long* out, float* sink) {
    const int blocks = 256, iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NACC, LDSOP>), dim3(blocks), dim3(256 * wps), 0, 0, iters, out, sink);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    (void)hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%d wave(s)/SIMD, %d accumulator(s), operands from %s: %.1f cycles per MFMA per wave = %.1f per SIMD\n", wps, NACC,
           LDSOP == 0 ? "registers" : LDSOP == 1 ? "LDS, 1 step ahead" : LDSOP == 2 ? "LDS, 2 steps ahead" : "LDS, 3 steps ahead", s / blocks / (iters * 8.0), s / blocks / (iters * 8.0) / wps);
}

int main() {
    long long* out; float* sink;
    (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&sink, 64);
    for (int wps = 1; wps <= 2; ++wps) {
        run<1, 0>(wps, out, sink); run<2, 0>(wps, out, sink); run<4, 0>(wps, out, sink);
        run<2, 1>(wps, out, sink); run<2, 2>(wps, out, sink); run<2, 3>(wps, out, sink);
    }
    return 0;
}

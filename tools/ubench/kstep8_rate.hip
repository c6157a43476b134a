// Micro-benchmark (gfx950): cycles per 8-MFMA Winograd K-step of conv_wino_ws (lone wave per SIMD, no
// barriers / staging / epilogue).  Ideal 512.  variant 0: production blocks (U loads issued by the
// compiler between the asm blocks); variant 1: asm split after the first MFMA so the U loads share the LDS-read group.
//   hipcc --offload-arch=gfx950 -O3 -I../../pfnl_amd/csrc -o kstep8_rate kstep8_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "wino_geom.h"
using namespace pfnl;
constexpr int UD = 8;

template <int V>
__global__ __launch_bounds__(256) void k(int units, long long* out, const float* upack, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * WP_BUF; i += blockDim.x) smem[i] = (float)(i & 7);
    __syncthreads();
    const int xi = wave & 3;
    const int tx = lane & 15, ty = (lane >> 4) & 1, kh = lane >> 5;
    constexpr int KS_F4 = 64, UP_CHUNK_F4 = 4 * 2 * 8 * KS_F4;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(upack), 0, (int)(wino_pack_floats_c * sizeof(float)), 0x00020000);
    const int uvoff = ((xi * 2 * 8) * KS_F4 + lane) * 16;
#define USOFF(s_, g_) (((((s_) & 31) >> 3) * UP_CHUNK_F4 + ((g_) * 8 + ((s_) & 7)) * KS_F4) * 16)
#define USTEP(s_, g_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uvoff, USOFF(s_, g_), 0))
    f32x4 ring0[UD], ring1[UD];
    for (int d = 0; d < UD; ++d) { ring0[d] = USTEP(d, 0); ring1[d] = USTEP(d, 1); }
    const int rowA = (xi == 0) ? 0 : 1, rowB = (xi == 3) ? 3 : 2;
    const float sgn = (xi == 1) ? 1.f : -1.f;
    const f32x2 sg2 = {sgn, sgn};
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned pa = lds0 + (unsigned)(kh * WP_PS + (2 * ty + rowA) * WP_RP + 2 * tx) * 4u;
    const unsigned pb = lds0 + (unsigned)(kh * WP_PS + (2 * ty + rowB) * WP_RP + 2 * tx) * 4u;
    f32x16 acc[8] = {};
    f32x2 vA = {1, 2}, vB = {3, 4}, vC, vD;
#define STEP(s_, c03_, c12_, n03_, n12_) do { constexpr int d_ = (s_) % UD; \
        constexpr int n_ = ((s_) + 1) & 31; constexpr int off_ = (((n_ >> 3) & 1) * WP_BUF + 2 * (n_ & 7) * WP_PS) * 4; \
        f32x2 x01_, y01_, x23_, y23_; \
        if (V == 0) { \
            const f32x4 b0_ = ring0[d_], b1_ = ring1[d_]; \
            ring0[d_] = USTEP((s_) + UD, 0); ring1[d_] = USTEP((s_) + UD, 1); \
            wq_kstep_a<off_, false>(acc[0], acc[1], acc[2], acc[3], (c03_).x, (c12_).x, (c12_).y, (c03_).y, b0_, x01_, y01_, x23_, y23_, pa, pb); \
            wq_kstep_b<false>(acc[4], acc[5], acc[6], acc[7], (c03_).x, (c12_).x, (c12_).y, (c03_).y, b1_, x01_, y01_, x23_, y23_, n03_, n12_, sg2); \
        } else { \
            const f32x4 b0_ = ring0[d_], b1_ = ring1[d_]; \
            wr_kstep_head<off_, false>(acc[0], (c03_).x, b0_.x, x01_, y01_, x23_, y23_, pa, pb); \
            __builtin_amdgcn_sched_barrier(0); \
            ring0[d_] = USTEP((s_) + UD, 0); ring1[d_] = USTEP((s_) + UD, 1); \
            __builtin_amdgcn_sched_barrier(0); \
            wr_kstep_rest<false>(acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7], (c03_).x, (c12_).x, (c12_).y, (c03_).y, b0_, b1_, x01_, y01_, x23_, y23_, n03_, n12_, sg2); \
        } \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#define STEP2(s_) STEP(s_, vA, vB, vC, vD); STEP((s_) + 1, vC, vD, vA, vB)
    const long long t0 = clock64();
    for (int u = 0; u < units; ++u) {
        STEP2(0); STEP2(2); STEP2(4); STEP2(6); STEP2(8); STEP2(10); STEP2(12); STEP2(14);
        STEP2(16); STEP2(18); STEP2(20); STEP2(22); STEP2(24); STEP2(26); STEP2(28); STEP2(30);
    }
    const long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)");
    float s = 0;
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int d = 0; d < UD; ++d) s += ring0[d].x + ring1[d].y;
    if (s == 12345.f) sink[0] = s;
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256, units = 100;
    long long* out; float *up, *sink;
    (void)hipMalloc(&out, blocks * 8); (void)hipMalloc(&up, wino_pack_floats_c * 4); (void)hipMalloc(&sink, 64);
    (void)hipMemset(up, 0, wino_pack_floats_c * 4);
    const size_t lds = 2 * WP_BUF * sizeof(float);
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            if (v) hipLaunchKernelGGL((k<1>), dim3(blocks), dim3(256), lds, 0, units, out, up, sink);
            else hipLaunchKernelGGL((k<0>), dim3(blocks), dim3(256), lds, 0, units, out, up, sink);
        }
        (void)hipDeviceSynchronize();
        std::vector<long long> h(blocks);
        (void)hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto x : h) s += x;
        const double per = s / blocks / (units * 32.0);
        printf("variant %d: %.1f cycles per 8-MFMA K-step (ideal 512) -> %.1f %%\n", v, per, 100.0 * 512 / per);
    }
    return 0;
}

// Micro-benchmark (gfx950): cycles per Winograd K-step (4 x v_mfma_f32_32x32x2_f32 + 8 ds_read + 12 VALU
// + one 16-byte U load) of the production inline-asm K-step, free of barriers / staging / epilogue.
//   nw   = matrix waves per SIMD (1, 2, 3)        ideal = 256 * nw cycles per K-step per wave
//   uld  = 0: U ring never refilled, 1: refilled from L2 as in the kernel
//   hipcc --offload-arch=gfx950 -O3 -I../../pfnl_amd/csrc -o kstep_rate kstep_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "wino_geom.h"
using namespace pfnl;

template <int ULD, int V2>
__global__ __launch_bounds__(768) void k(int units, long long* out, const float* upack, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * WP_BUF; i += blockDim.x) smem[i] = (float)(i & 7);
    __syncthreads();
    const int xi = wave & 3;
    const int tx = lane & 15, ty = (lane >> 4) & 1, kh = lane >> 5;
    constexpr int KS_F4 = 64, UP_CHUNK_F4 = 4 * 2 * 8 * KS_F4;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(upack), 0, (int)(wino_pack_floats_c * sizeof(float)), 0x00020000);
    const int uvoff = (((xi * 2 + ((wave >> 2) & 1)) * 8) * KS_F4 + lane) * 16;
#define USTEP(s_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uvoff, ((((s_) & 31) >> 3) * UP_CHUNK_F4 + ((s_) & 7) * KS_F4) * 16, 0))
    f32x4 ring[2];
    ring[0] = USTEP(0); ring[1] = USTEP(1);
    const int rowA = (xi == 0) ? 0 : 1, rowB = (xi == 3) ? 3 : 2;
    const float sgn = (xi == 1) ? 1.f : -1.f;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned lane_off = (unsigned)(kh * WN_PS + (2 * ty) * WN_RP + tx) * 4u;
    const unsigned pa = lds0 + lane_off + rowA * WN_RP * 4, pb = lds0 + lane_off + rowB * WN_RP * 4;
    f32x16 acc[4] = {};
    float va[4] = {1, 2, 3, 4}, vb[4];
    const unsigned qa = lds0 + (unsigned)(kh * WP_PS + (2 * ty + rowA) * WP_RP + 2 * tx) * 4u;
    const unsigned qb = lds0 + (unsigned)(kh * WP_PS + (2 * ty + rowB) * WP_RP + 2 * tx) * 4u;
    const f32x2 sg2 = {sgn, sgn};
    f32x2 pA = {1, 2}, pB = {3, 4}, pC, pD;     // (V0,V3), (V1,V2)
#define KSTEP2(q_, kk_, c03_, c12_, n03_, n12_) do { constexpr int s_ = (q_) * 8 + (kk_); const f32x4 bc_ = ring[s_ % 2]; \
        if (ULD) ring[s_ % 2] = USTEP(s_ + 2); \
        wp_kstep_asm<(((q_) & 1) * WP_BUF + 2 * (((kk_) + 1) & 7) * WP_PS) * 4>(acc[0], acc[1], acc[2], acc[3], c03_.x, c12_.x, c12_.y, c03_.y, n03_, n12_, bc_, qa, qb, sg2); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#define CHUNK2(q_) do { KSTEP2(q_, 0, pA, pB, pC, pD); KSTEP2(q_, 1, pC, pD, pA, pB); KSTEP2(q_, 2, pA, pB, pC, pD); KSTEP2(q_, 3, pC, pD, pA, pB); \
                        KSTEP2(q_, 4, pA, pB, pC, pD); KSTEP2(q_, 5, pC, pD, pA, pB); KSTEP2(q_, 6, pA, pB, pC, pD); KSTEP2(q_, 7, pC, pD, pA, pB); } while (0)
#define KSTEP(q_, kk_, vcur_, vnxt_) do { constexpr int s_ = (q_) * 8 + (kk_); const f32x4 bc_ = ring[s_ % 2]; \
        if (ULD) ring[s_ % 2] = USTEP(s_ + 2); \
        wn_kstep_asm<(((q_) & 1) * WN_BUF + 2 * (((kk_) + 1) & 7) * WN_PS) * 4>(acc[0], acc[1], acc[2], acc[3], vcur_, vnxt_, bc_, pa, pb, sgn); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#define CHUNK(q_) do { KSTEP(q_, 0, va, vb); KSTEP(q_, 1, vb, va); KSTEP(q_, 2, va, vb); KSTEP(q_, 3, vb, va); \
                       KSTEP(q_, 4, va, vb); KSTEP(q_, 5, vb, va); KSTEP(q_, 6, va, vb); KSTEP(q_, 7, vb, va); } while (0)
    const long long t0 = clock64();
    for (int u = 0; u < units; ++u) { if (V2) { CHUNK2(0); CHUNK2(1); CHUNK2(2); CHUNK2(3); } else { CHUNK(0); CHUNK(1); CHUNK(2); CHUNK(3); } }
    const long long t1 = clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r] + acc[2][r] + acc[3][r];
    if (s == 12345.f) sink[0] = s;
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256, units = 100;
    long long* out; float *up, *sink;
    (void)hipMalloc(&out, blocks * 8); (void)hipMalloc(&up, wino_pack_floats_c * 4); (void)hipMalloc(&sink, 64);
    (void)hipMemset(up, 0, wino_pack_floats_c * 4);
    const size_t lds = 2 * WP_BUF * sizeof(float);
    for (int uld = 0; uld < 4; ++uld)
        for (int nw = 1; nw <= 2; ++nw) {
            for (int rep = 0; rep < 2; ++rep) {
                if (uld == 3) hipLaunchKernelGGL((k<1, 1>), dim3(blocks), dim3(256 * nw), lds, 0, units, out, up, sink);
                else if (uld == 2) hipLaunchKernelGGL((k<0, 1>), dim3(blocks), dim3(256 * nw), lds, 0, units, out, up, sink);
                else if (uld) hipLaunchKernelGGL((k<1, 0>), dim3(blocks), dim3(256 * nw), lds, 0, units, out, up, sink);
                else hipLaunchKernelGGL((k<0, 0>), dim3(blocks), dim3(256 * nw), lds, 0, units, out, up, sink);
            }
            (void)hipDeviceSynchronize();
            std::vector<long long> h(blocks);
            (void)hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            const double per = s / blocks / (units * 32.0);
            printf("variant %d (bit0 = U loads, bit1 = packed-math K-step), %d matrix wave(s)/SIMD: %.1f cycles per K-step per wave -> MFMA pipe %.1f %%\n", uld, nw, per, 100.0 * 256 * nw / per);
        }
    return 0;
}

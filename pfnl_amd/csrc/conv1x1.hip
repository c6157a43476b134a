// conv10_i of the reference (model/pfnl.py:50 applied at :67-68): 1x1 convolution over the channel
// concatenation of the T frames of a clip, T*64 -> 64 channels, + bias, leaky_relu(0.2) -> `base`.
//
// A 1x1 convolution has no spatial reuse, so nothing needs to go through LDS: the GEMM's A operand
// (M = pixels, K = T*64 input channels) is read straight from HBM as 16-byte pieces, and - the order
// of a contraction being free - the K order is chosen so that a float4 of 4 CONSECUTIVE channels held by
// one lane feeds 4 consecutive v_mfma_f32_32x32x2_f32:  lane (pixel = l&31, kh = l>>5) loads channels
// [32M + 16kh + 4t, +4) of frame f; MFMA (t, e) then contracts the channel pair (32M + 4t + e,
// 32M + 16 + 4t + e).  The weights are packed on the host in exactly that order, 16 bytes per lane per
// 4 MFMAs.  Per 8 MFMAs the wave issues one global load, two LDS reads and no VALU instruction at all,
// which is what a saturated gfx950 matrix pipe wants (tools/ubench/).  The four float4 of one pixel's
// 32-channel half (one 128-byte line) are requested back to back.
//
// The weights (16 KB per frame) go through LDS, one frame ahead of the MFMAs, double-buffered: the B
// operand is one ds_read_b128 per lane per 4 MFMAs, and they are shared by the 8 waves of a workgroup (a
// first version that streamed B per wave from L2 stalled behind the HBM-latency A pieces in the in-order
// vmcnt queue; a second one that copied all T*16 KB up front paid that copy as a serial prologue, which
// matters when a wave only lives for T*64 MFMAs).  Wave = 32 consecutive pixels of one clip x all 64
// output channels (2 accumulators); the next two frames' A pieces and the next frame's weights are in flight while
// the current frame's 64 MFMAs run; the frame barrier sits between steps 6 and 7, so that step 7 can
// already prefetch the next frame's first B.  The epilogue adds the bias, applies the activation on the
// accumulators and stores 128-byte channel-contiguous segments per pixel.
#include <cstdint>

#include "common.h"

namespace pfnl {

constexpr int C1_THREADS = 512;
constexpr int C1_WF = 16 * 64;                                     // f32x4 of weights per frame (16 KB)

__global__ __launch_bounds__(C1_THREADS, 2) void conv1x1_stream_kernel(const float* __restrict__ in,
                                                                       const float* __restrict__ wpack,
                                                                       const float* __restrict__ bias,
                                                                       float* __restrict__ out, int HW, int T, int items,
                                                                       int act) {
    __shared__ __attribute__((aligned(16))) f32x4 sw[2][C1_WF];     // [frame parity][M][t][g][lane] x e
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xl = lane & 31;
    const int kh = lane >> 5;
    const int gpi = (HW + 31) >> 5;                                 // pixel groups per clip
    const int ngroups = gpi * items;
    const int g = min(blockIdx.x * 8 + wave, ngroups - 1);          // surplus waves redo the last group (same values)
    const int item = g / gpi;
    const int p0 = (g - item * gpi) * 32;

    const f32x4* wsrc = reinterpret_cast<const f32x4*>(wpack) + tid;   // + f*C1_WF (+512)
    f32x4 wr0 = wsrc[0], wr1 = wsrc[C1_THREADS];
    // A: this lane's pixel (clamped; masked at the store), 16-byte pieces at channel 16*kh + 4*t of half M
    const f32x4* ap = reinterpret_cast<const f32x4*>(in) + (((size_t)item * T * HW + min(p0 + xl, HW - 1)) * 16 + kh * 4);
    const size_t aframe = (size_t)HW * 16;                          // f32x4 per frame
#ifdef PFNL_C1_NO_A   /* timing experiment only: wrong results */
#define C1_LOAD_A(dst, ap_) do { (void)(ap_); } while (0)
#else
#define C1_LOAD_A(dst, ap_)                                                                            \
    do {                                                                                               \
        _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) dst[q_] = (ap_)[(q_ >> 2) * 8 + (q_ & 3)];     \
    } while (0)
#endif
    f32x4 a0[8] = {}, a1[8] = {}, a2[8] = {};                                    // rotating sets: [M*4 + t] of one frame
    C1_LOAD_A(a0, ap);
    C1_LOAD_A(a1, ap + (size_t)min(1, T - 1) * aframe);
    sw[0][tid] = wr0;
    sw[0][tid + C1_THREADS] = wr1;
    const float bias0 = bias[xl], bias1 = bias[32 + xl];
    __syncthreads();

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.f;
        acc1[r] = 0.f;
    }
    f32x4 w0[2], w1[2];                                             // B double buffer: [step parity]
    w0[0] = sw[0][lane];
    w1[0] = sw[0][64 + lane];

    // one frame: request the A pieces two frames ahead and the next frame's weights, 64 MFMAs
#define C1_FRAME(cur, far, f_)                                                                         \
    do {                                                                                               \
        const int fn_ = min((f_) + 1, T - 1);                   /* past the end: harmless re-read */   \
        C1_LOAD_A(far, ap + (size_t)min((f_) + 2, T - 1) * aframe);                                    \
        wr0 = wsrc[(size_t)fn_ * C1_WF];                                                               \
        wr1 = wsrc[(size_t)fn_ * C1_WF + C1_THREADS];                                                  \
        __builtin_amdgcn_sched_barrier(0);                      /* the streams stay a whole frame ahead */ \
        const f32x4* wl_ = &sw[(f_) & 1][lane];                                                        \
        const f32x4* wn_ = &sw[((f_) + 1) & 1][lane];                                                  \
        _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                             \
            if (q_ == 6) {                                      /* next frame's weights -> the other buffer */ \
                sw[((f_) + 1) & 1][tid] = wr0;                                                         \
                sw[((f_) + 1) & 1][tid + C1_THREADS] = wr1;                                            \
            }                                                                                          \
            if (q_ == 7) __syncthreads();                                                              \
            /* B of the next step (next frame's first step after the last) lands during this step's MFMAs */ \
            const f32x4* wq_ = q_ < 7 ? wl_ + ((q_ + 1) * 2) * 64 : wn_;                               \
            w0[(q_ + 1) & 1] = wq_[0];                                                                 \
            w1[(q_ + 1) & 1] = wq_[64];                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            const f32x4 w0_ = w0[q_ & 1], w1_ = w1[q_ & 1];                                            \
            acc0 = mfma32(cur[q_].x, w0_.x, acc0);                                                     \
            acc1 = mfma32(cur[q_].x, w1_.x, acc1);                                                     \
            acc0 = mfma32(cur[q_].y, w0_.y, acc0);                                                     \
            acc1 = mfma32(cur[q_].y, w1_.y, acc1);                                                     \
            acc0 = mfma32(cur[q_].z, w0_.z, acc0);                                                     \
            acc1 = mfma32(cur[q_].z, w1_.z, acc1);                                                     \
            acc0 = mfma32(cur[q_].w, w0_.w, acc0);                                                     \
            acc1 = mfma32(cur[q_].w, w1_.w, acc1);                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        }                                                                                              \
    } while (0)
    for (int f = 0; f < T; f += 3) {
        C1_FRAME(a0, a2, f);
        if (f + 1 < T) C1_FRAME(a1, a0, f + 1);
        if (f + 2 < T) C1_FRAME(a2, a1, f + 2);
    }
#undef C1_FRAME
#undef C1_LOAD_A

    // epilogue: lane holds output channel xl (acc0) and 32 + xl (acc1) of pixels drow(r, lane)
    if ((int)(blockIdx.x * 8 + wave) >= ngroups) return;
    const float slope = act ? 0.2f : 1.0f;
    float* ob = out + ((size_t)item * HW + p0) * 64 + xl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int px = drow(r, lane);
        float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
        v0 = fmaxf(v0, slope * v0);
        v1 = fmaxf(v1, slope * v1);
        if (p0 + px < HW) {
            ob[(size_t)px * 64] = v0;
            ob[(size_t)px * 64 + 32] = v1;
        }
    }
}

hipError_t launch_conv1x1_stream(const float* in, const float* wpack, const float* bias, float* out, int items, int T,
                                 int HW, int act, hipStream_t s) {
    if (!in || !wpack || !bias || !out || items < 1 || T < 1 || HW < 1) return hipErrorInvalidValue;
    const int ngroups = ((HW + 31) / 32) * items;
    hipLaunchKernelGGL(conv1x1_stream_kernel, dim3((ngroups + 7) / 8), dim3(C1_THREADS), 0, s, in, wpack, bias, out, HW, T,
                       items, act);
    return hipGetLastError();
}

size_t conv1x1_pack_floats(int T) { return (size_t)T * 64 * 64; }

// HWIO [1,1,T*64,64] -> [f][M][t][g][lane][e], value W[f*64 + 32M + 16*(lane>>5) + 4t + e][32g + (lane&31)]
void conv1x1_pack_weights(const float* hwio, int T, float* dst) {
    for (int f = 0; f < T; ++f)
        for (int M = 0; M < 2; ++M)
            for (int t = 0; t < 4; ++t)
                for (int g = 0; g < 2; ++g)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int ci = f * 64 + 32 * M + 16 * (lane >> 5) + 4 * t + e;
                            const int co = 32 * g + (lane & 31);
                            dst[((((((size_t)f * 2 + M) * 4 + t) * 2 + g) * 64 + lane) * 4) + e] =
                                hwio[(size_t)ci * 64 + co];
                        }
}

// ------------------------------------------------------------------------------------------------
// The same 1x1 on the f16 matrix pipe with exactly split fp32 operands (option conv1x1=split16; see conv_split16.hip for the
// arithmetic: x = f16(x) + f16((x - f16(x)) 2^11) 2^-11, three f16 MFMAs per product block, fp32 accumulation).  Same streaming
// structure as conv1x1_stream_kernel - A straight from HBM two frames ahead, the frame's weights (16 KB: [k-step 4][g 2]
// [hi/lo' 2][lane] x 16 B) through a double-buffered LDS slot one frame ahead, one barrier per frame - with 24 MFMAs of 32
// cycles per frame instead of 64 of 64: the kernel is left with its HBM stream (117 MB read per launch at configs[1]).
// The two float4 of 8 consecutive channels a lane holds are split in registers (8 VALU per float4) and form one MFMA operand:
// k-step (M, h) contracts channels 32M + 16kh + 8h + e, the weights are packed in that order.
typedef _Float16 c1h8 __attribute__((ext_vector_type(8)));
typedef _Float16 c1h4 __attribute__((ext_vector_type(4)));
typedef unsigned c1u4 __attribute__((ext_vector_type(4)));
typedef unsigned c1u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void c1_split4(f32x4 v, c1u2& hi, c1u2& lo, float nscale) {
    const c1h4 h = __builtin_convertvector(v, c1h4);
    hi = __builtin_bit_cast(c1u2, h);
    const f32x4 t = v * 2048.0f;
    unsigned l0, l1;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi.x), "s"(nscale), "v"(t.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(hi.x), "s"(nscale), "v"(t.y));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi.y), "s"(nscale), "v"(t.z));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(hi.y), "s"(nscale), "v"(t.w));
    lo = c1u2{l0, l1};
}

// ISF: `in` is in the split format (conv_split16.h): the two operands of a k-step are two 16-byte loads, no arithmetic.
// OSF: `out` is written in the split format (base feeds the shared half of conv2_i only: conv_sf.hip).
// NW = waves per workgroup (one 32-pixel group each): 8 for launches with at least a workgroup per CU; 2 for the mid shapes' launches
// (capi.hip "MID shapes": one clip of 128x128 = 512 groups = 64 workgroups of 8 waves on 64 CUs, each moving 80 KB per frame through its
// port - 24.9 us; as 256 workgroups of 2 waves 32 KB per frame and CU).  Same arithmetic per wave: no bit changes.
template <bool ISF, bool OSF, int NW>
__global__ __launch_bounds__(NW * 64, 2) void conv1x1_split16_kernel(const float* __restrict__ in,
                                                                        const uint16_t* __restrict__ wpack,
                                                                        const float* __restrict__ bias,
                                                                        float* __restrict__ out, int HW, int T, int items,
                                                                        int act) {
    __shared__ __attribute__((aligned(16))) c1u4 sw[2][C1_WF];      // [frame parity][k-step][g][part][lane] x 16 B
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xl = lane & 31;
    const int kh = lane >> 5;
    const int gpi = (HW + 31) >> 5;
    const int ngroups = gpi * items;
    constexpr int NT = NW * 64, WR = C1_WF / NT;                    // threads; 16-byte weight pieces per thread and frame
    const int g = min((int)blockIdx.x * NW + wave, ngroups - 1);    // surplus waves redo the last group (same values)
    const int item = g / gpi;
    const int p0 = (g - item * gpi) * 32;
    const float nscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -2048.0f)));

    const c1u4* wsrc = reinterpret_cast<const c1u4*>(wpack) + tid;  // + f*C1_WF (+ k NT)
    c1u4 wr[WR];
#pragma unroll
    for (int k = 0; k < WR; ++k) wr[k] = wsrc[k * NT];
    // fp32: piece (q >> 2) * 8 + kh * 4 + (q & 3) of the pixel's 16 (4 channels each); pieces 2q', 2q'+1 = k-step q' = (M, h).
    // SF: 16-byte chunks of the pixel: [M][part][4]: k-step (M, h) takes chunk 2 kh + h of part hi (-> dst[2q']) and lo' (-> dst[2q'+1])
    const f32x4* ap = reinterpret_cast<const f32x4*>(in) + (((size_t)item * T * HW + min(p0 + xl, HW - 1)) * 16 + (ISF ? kh * 2 : kh * 4));
    const size_t aframe = (size_t)HW * 16;
#define C1S_LOAD_A(dst, ap_)                                                                           \
    do {                                                                                               \
        _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_)                                                \
            dst[q_] = ISF ? (ap_)[(q_ >> 2) * 8 + ((q_ >> 1) & 1) + (q_ & 1) * 4] : (ap_)[(q_ >> 2) * 8 + (q_ & 3)]; \
    } while (0)
    f32x4 a0[8] = {}, a1[8] = {}, a2[8] = {};
    C1S_LOAD_A(a0, ap);
    C1S_LOAD_A(a1, ap + (size_t)min(1, T - 1) * aframe);
#pragma unroll
    for (int k = 0; k < WR; ++k) sw[0][tid + k * NT] = wr[k];
    const float bias0 = bias[xl], bias1 = bias[32 + xl];
    __syncthreads();

    f32x16 accm0, accm1, accc0, accc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        accm0[r] = 0.f;
        accm1[r] = 0.f;
        accc0[r] = 0.f;
        accc1[r] = 0.f;
    }
    c1u4 wq[2][4];                                                  // B double buffer: [step parity][g*2 + part]
#pragma unroll
    for (int i = 0; i < 4; ++i) wq[0][i] = sw[0][i * 64 + lane];

#define C1S_FRAME(cur, far, f_)                                                                        \
    do {                                                                                               \
        const int fn_ = min((f_) + 1, T - 1);                   /* past the end: harmless re-read */   \
        C1S_LOAD_A(far, ap + (size_t)min((f_) + 2, T - 1) * aframe);                                   \
        _Pragma("unroll") for (int k_ = 0; k_ < WR; ++k_) wr[k_] = wsrc[(size_t)fn_ * C1_WF + k_ * NT]; \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        const c1u4* wl_ = &sw[(f_) & 1][lane];                                                         \
        const c1u4* wn_ = &sw[((f_) + 1) & 1][lane];                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {      /* k-step q = (M, h): pieces 2q, 2q+1 of the frame */ \
            if (q_ == 2) {                                      /* next frame's weights -> the other buffer */ \
                _Pragma("unroll") for (int k_ = 0; k_ < WR; ++k_) sw[((f_) + 1) & 1][tid + k_ * NT] = wr[k_]; \
            }                                                                                          \
            if (q_ == 3) __syncthreads();                                                              \
            const c1u4* wp_ = q_ < 3 ? wl_ + ((q_ + 1) * 4) * 64 : wn_;                                \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) wq[(q_ + 1) & 1][i_] = wp_[i_ * 64];      \
            c1h8 ah_, al_;                                                                             \
            if constexpr (ISF) {                                                                       \
                ah_ = __builtin_bit_cast(c1h8, cur[2 * q_]);                                           \
                al_ = __builtin_bit_cast(c1h8, cur[2 * q_ + 1]);                                       \
            } else {                                                                                   \
                c1u2 h0_, l0_, h1_, l1_;                                                               \
                c1_split4(cur[2 * q_], h0_, l0_, nscale);                                              \
                c1_split4(cur[2 * q_ + 1], h1_, l1_, nscale);                                          \
                ah_ = __builtin_bit_cast(c1h8, c1u4{h0_.x, h0_.y, h1_.x, h1_.y});                      \
                al_ = __builtin_bit_cast(c1h8, c1u4{l0_.x, l0_.y, l1_.x, l1_.y});                      \
            }                                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            const c1h8 w0h_ = __builtin_bit_cast(c1h8, wq[q_ & 1][0]), w0l_ = __builtin_bit_cast(c1h8, wq[q_ & 1][1]); \
            const c1h8 w1h_ = __builtin_bit_cast(c1h8, wq[q_ & 1][2]), w1l_ = __builtin_bit_cast(c1h8, wq[q_ & 1][3]); \
            accm0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, w0h_, accm0, 0, 0, 0);                  \
            accm1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, w1h_, accm1, 0, 0, 0);                  \
            accc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, w0l_, accc0, 0, 0, 0);                  \
            accc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, w1l_, accc1, 0, 0, 0);                  \
            accc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_, w0h_, accc0, 0, 0, 0);                  \
            accc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_, w1h_, accc1, 0, 0, 0);                  \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        }                                                                                              \
    } while (0)
    for (int f = 0; f < T; f += 3) {
        C1S_FRAME(a0, a2, f);
        if (f + 1 < T) C1S_FRAME(a1, a0, f + 1);
        if (f + 2 < T) C1S_FRAME(a2, a1, f + 2);
    }
#undef C1S_FRAME
#undef C1S_LOAD_A

    if ((int)blockIdx.x * NW + wave >= ngroups) return;
    const float slope = act ? 0.2f : 1.0f;
    float* ob = out + ((size_t)item * HW + p0) * 64 + xl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int px = drow(r, lane);
        float v0 = accm0[r] + accc0[r] * (1.0f / 2048.0f) + bias0, v1 = accm1[r] + accc1[r] * (1.0f / 2048.0f) + bias1;
        v0 = fmaxf(v0, slope * v0);
        v1 = fmaxf(v1, slope * v1);
        if (p0 + px < HW) {
            if constexpr (OSF) {                                    // pixel = 128 halves: [M = g][hi | lo'][32]
                _Float16* const oh = reinterpret_cast<_Float16*>(out) + ((size_t)item * HW + p0 + px) * 128 + xl;
                const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
                oh[0] = h0;
                oh[32] = (_Float16)((v0 - (float)h0) * 2048.0f);
                oh[64] = h1;
                oh[96] = (_Float16)((v1 - (float)h1) * 2048.0f);
            } else {
                ob[(size_t)px * 64] = v0;
                ob[(size_t)px * 64 + 32] = v1;
            }
        }
    }
}

hipError_t launch_conv1x1_split16(const float* in, const uint16_t* wpack, const float* bias, float* out, int items, int T,
                                  int HW, int act, hipStream_t s, bool in_sf, bool out_sf) {
    if (!in || !wpack || !bias || !out || items < 1 || T < 1 || HW < 1) return hipErrorInvalidValue;
    const int ngroups = ((HW + 31) / 32) * items;
    const int ncu = device_cu_count();
    if ((ngroups + 7) / 8 < (ncu ? ncu : 256)) {                    // fewer 8-wave workgroups than CUs: 2 waves per workgroup
        const dim3 grid((ngroups + 1) / 2), block(128);
        if (in_sf && out_sf) hipLaunchKernelGGL((conv1x1_split16_kernel<true, true, 2>), grid, block, 0, s, in, wpack, bias, out, HW, T, items, act);
        else if (in_sf) hipLaunchKernelGGL((conv1x1_split16_kernel<true, false, 2>), grid, block, 0, s, in, wpack, bias, out, HW, T, items, act);
        else if (out_sf) hipLaunchKernelGGL((conv1x1_split16_kernel<false, true, 2>), grid, block, 0, s, in, wpack, bias, out, HW, T, items, act);
        else hipLaunchKernelGGL((conv1x1_split16_kernel<false, false, 2>), grid, block, 0, s, in, wpack, bias, out, HW, T, items, act);
        return hipGetLastError();
    }
    const dim3 grid((ngroups + 7) / 8), block(C1_THREADS);
    if (in_sf && out_sf) hipLaunchKernelGGL((conv1x1_split16_kernel<true, true, 8>), grid, block, 0, s, in, wpack, bias, out, HW, T, items, act);
    else if (in_sf) hipLaunchKernelGGL((conv1x1_split16_kernel<true, false, 8>), grid, block, 0, s, in, wpack, bias, out, HW, T, items, act);
    else if (out_sf) hipLaunchKernelGGL((conv1x1_split16_kernel<false, true, 8>), grid, block, 0, s, in, wpack, bias, out, HW, T, items, act);
    else hipLaunchKernelGGL((conv1x1_split16_kernel<false, false, 8>), grid, block, 0, s, in, wpack, bias, out, HW, T, items, act);
    return hipGetLastError();
}

size_t conv1x1_split16_pack_halfs(int T) { return (size_t)T * 8192; }   // 16 KB per frame

// HWIO [1,1,T*64,64] -> [f][k-step q = 2M + h][g][part][lane][e]: W[f*64 + 32M + 16(lane>>5) + 8h + e][32g + (lane&31)],
// part 0 = f16(w), part 1 = f16((w - hi) 2^11)
void conv1x1_split16_pack_weights(const float* hwio, int T, uint16_t* dst) {
    for (int f = 0; f < T; ++f)
        for (int q = 0; q < 4; ++q)
            for (int g = 0; g < 2; ++g)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int ci = f * 64 + 32 * (q >> 1) + 16 * (lane >> 5) + 8 * (q & 1) + e;
                        const int co = 32 * g + (lane & 31);
                        const float w = hwio[(size_t)ci * 64 + co];
                        const _Float16 hi = (_Float16)w;
                        const _Float16 lo = (_Float16)((w - (float)hi) * 2048.0f);
                        const size_t base = (((((size_t)f * 4 + q) * 2 + g) * 2) * 64 + lane) * 8 + e;
                        uint16_t hb, lb;
                        __builtin_memcpy(&hb, &hi, 2);
                        __builtin_memcpy(&lb, &lo, 2);
                        dst[base] = hb;
                        dst[base + 512] = lb;
                    }
}

}  // namespace pfnl

// Head and tail of PFNL.forward (gfx950): bandwidth-class kernels, f32 VALU.
//   conv0   : lrelu(conv5x5 3->64) on each frame of the non-local output   (model/pfnl.py:48,61-62)
//   tail    : depth_to_space -> convmerge2 3x3 (no activation) -> depth_to_space -> + bicubic(x_centre)
//             -> [B,1,sH,sW,3]                                            (model/pfnl.py:53,63,76-80)
//   bicubic : TF1.12 legacy ResizeBicubic (align_corners=False, no half-pixel centres, A=-0.75,
//             clamped taps, no renormalisation)                           (model/pfnl.py:63)
#include <cstdlib>

#include "common.h"
#include "conv_bf16.h"

namespace pfnl {

// ------------------------------------------------------------------------------------------------
// conv0.  Input is the packed non-local output Xo [B][N][CP] (space_to_depth layout, see
// nonlocal.hip); pixel (y,x), stack channel k lives at Xo[b][(y/2)(W/2)+x/2][((y&1)*2+(x&1))*3T+k].
// Workgroup = 16x16 output pixels of one frame; thread = one pixel x 64 output channels.
constexpr int C0_T = 16;
constexpr int C0_IN = C0_T + 4;

template <bool BF16OUT>                                             // bf16 trunk: `out` is [F][H][W][64] bf16
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ Xo,
                                                    const float* __restrict__ w,     // [75][64]
                                                    const float* __restrict__ bias,  // [64]
                                                    float* __restrict__ out, int T, int H, int W,
                                                    int CP, int yoff, int Hs) {   // output = LR rows [yoff, yoff + Hs) of the H-row frame
    __shared__ __attribute__((aligned(16))) float smem0[256 * 36];   // weights [75][64] | input tile, later the store slab
    float* const sw = smem0;
    float* const s_in = smem0 + 75 * 64;
    const int tid = threadIdx.x;
    const int f = blockIdx.z;
    const int b = f / T, t = f % T;
    const int x0 = blockIdx.x * C0_T, y0 = blockIdx.y * C0_T;
    const int W2 = W / 2, C3 = 3 * T;
    const float* Xb = Xo + (size_t)b * (H / 2) * W2 * CP;

    for (int i = tid; i < 75 * 64; i += 256) sw[i] = w[i];
    for (int i = tid; i < C0_IN * C0_IN * 3; i += 256) {
        const int c = i % 3, pix = i / 3;
        const int py = pix / C0_IN, px = pix % C0_IN;
        const int gy = yoff + y0 + py - 2, gx = x0 + px - 2;       // (frame coordinates: rows outside a strip are real data)
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = Xb[((size_t)(gy >> 1) * W2 + (gx >> 1)) * CP + ((gy & 1) * 2 + (gx & 1)) * C3 + 3 * t + c];
        s_in[i] = v;
    }
    __syncthreads();

    const int ty = tid / C0_T, tx = tid % C0_T;
    // 64 output channels as 32 packed pairs: v_pk_fma_f32 does two channels per VALU issue
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) acc[o] = f32x2{bias[2 * o], bias[2 * o + 1]};
    for (int ky = 0; ky < 5; ++ky) {
        for (int kx = 0; kx < 5; ++kx) {
            const float* ip = s_in + ((ty + ky) * C0_IN + tx + kx) * 3;
            const float* wp = sw + (ky * 5 + kx) * 3 * 64;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float a = ip[c];
                const f32x2 a2 = {a, a};
#pragma unroll
                for (int o4 = 0; o4 < 16; ++o4) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + c * 64 + o4 * 4);
                    acc[o4 * 2 + 0] = __builtin_elementwise_fma(a2, f32x2{wv.x, wv.y}, acc[o4 * 2 + 0]);
                    acc[o4 * 2 + 1] = __builtin_elementwise_fma(a2, f32x2{wv.z, wv.w}, acc[o4 * 2 + 1]);
                }
            }
        }
    }
    // Stores: a thread owns one pixel's 64 channels (256 B); stored directly, every instruction would scatter 64
    // 16-byte pieces over 64 cache lines.  Two passes of 32 channels through an LDS slab (aliasing the weights and
    // the input tile) turn them into instructions that write 8 whole 128-byte lines.
    constexpr int SS = 36;                                          // slab row stride (floats): 32 channels + 4
    float* slab = smem0;
    static_assert(75 * 64 + C0_IN * C0_IN * 3 <= 256 * SS, "weights + input tile must fit the slab they share LDS with");
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
        __syncthreads();                                            // weights / input tile (or the previous pass) no longer needed
#pragma unroll
        for (int o4 = 0; o4 < 8; ++o4) {
            const int o = hp * 16 + o4 * 2;
            *reinterpret_cast<f32x4*>(slab + tid * SS + o4 * 4) =
                f32x4{lrelu(acc[o].x), lrelu(acc[o].y), lrelu(acc[o + 1].x), lrelu(acc[o + 1].y)};
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int id = it * 256 + tid;
            const int pix = id >> 3, q = id & 7;
            const int y = y0 + pix / C0_T, x = x0 + pix % C0_T;
            if (y < Hs && x < W) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(slab + pix * SS + q * 4);
                if (BF16OUT) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<uint16_t*>(out) + (((size_t)f * Hs + y) * W + x) * 64 + hp * 32 + q * 4) =
                        __builtin_convertvector(v, bf16x4);
                } else {
                    *reinterpret_cast<f32x4*>(out + (((size_t)f * Hs + y) * W + x) * 64 + hp * 32 + q * 4) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv0 on the f16 matrix pipe (default since round 2; the VALU kernel above stays as PFNL_CONV0=valu).
// 4800 MACs per output pixel are 84 k cycles of packed VALU per launch at configs[1]; as a GEMM (M = pixels, N = 64, K = 75)
// with exactly split fp32 operands (conv_split16.hip: x = hi + lo' 2^-11, 3 MFMAs per product block, fp32 accumulation) the
// arithmetic is a tenth of that and the launch is left with its 117 MB of output.
//   * K is ordered (ky, kx, c) and padded per row tap to 16: the 15 values (kx, c) of one ky are 15 CONSECUTIVE dwords of the
//     [row][col][3] input tile, so the A operand of k-step ky is 8 dwords per lane starting at the output pixel's own column
//     (the 16th value is the next pixel's first channel, met by a zero weight);
//   * the input tile is split ONCE when it is loaded, as packed (hi | lo' << 16) dwords: the gather is 8 ds_read_b32 (stride
//     3 dwords between lanes: conflict-free) + 8 v_perm_b32 per k-step; weights are split and laid out per (ky, part, N-tile,
//     lane) by the workgroup itself (20 KB), 16 bytes per lane and MFMA;
//   * workgroup = 4 waves = 16 rows x 32 columns of one frame, a wave walks 4 rows: 30 MFMAs per row, the accumulators of a row
//     leave as whole 128-byte channel lines (lane = channel, register = pixel).
constexpr int C0M_TW = 32, C0M_TH = 16, C0M_IW = C0M_TW + 4, C0M_IH = C0M_TH + 4;
constexpr int C0M_IN_DW = C0M_IH * C0M_IW * 3 + 4;                  // + slack for the pad element of the last pixel
constexpr int C0M_W_BYTES = 5 * 2 * 2 * 1024;                       // [ky][part][N-tile][lane] x 16 B

template <bool BF16OUT>
__global__ __launch_bounds__(256) void conv0_mfma_kernel(const float* __restrict__ Xo, const float* __restrict__ w,   // [75][64]
                                                         const float* __restrict__ bias, float* __restrict__ out, int T, int H,
                                                         int W, int CP, int yoff, int Hs) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned char sw[C0M_W_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned s_in[C0M_IN_DW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = blockIdx.z;
    const int b = f / T, t = f % T;
    const int x0 = blockIdx.x * C0M_TW, y0 = blockIdx.y * C0M_TH;
    const int W2 = W / 2, C3 = 3 * T;
    const float* Xb = Xo + (size_t)b * (H / 2) * W2 * CP;

    auto split1 = [](float v, unsigned short& hb, unsigned short& lb) {
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);    // exact difference, one rounding
        hb = __builtin_bit_cast(unsigned short, hi);
        lb = __builtin_bit_cast(unsigned short, lo);
    };
    // Everything the prologue needs - this thread's weight values and its dwords of the input tile - is requested before any of it is
    // used: written as a loop with the load under a bounds test, every dword of the tile waited for its own round trip (9 in a row:
    // most of the 13 us this launch took at 7 x 32 x 32, where nothing else hides them).  Out-of-tile / out-of-image dwords read a
    // clamped address and are zeroed.
    constexpr int C0M_WIT = (5 * 2 * 64 + 255) / 256, C0M_XIT = (C0M_IN_DW + 255) / 256;
    float wreg[C0M_WIT][8], xreg[C0M_XIT];
#pragma unroll
    for (int k = 0; k < C0M_WIT; ++k) {
        const int ent = min(tid + k * 256, 5 * 2 * 64 - 1);
        const int l = ent & 63, g = (ent >> 6) & 1, ky = ent >> 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kp = 8 * (l >> 5) + e;
            const float wv = w[(size_t)(ky * 15 + min(kp, 14)) * 64 + 32 * g + (l & 31)];
            wreg[k][e] = kp < 15 ? wv : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < C0M_XIT; ++k) {
        const int i = min(tid + k * 256, C0M_IN_DW - 1);
        const int c = i % 3, pix = i / 3;
        const int py = pix / C0M_IW, px = pix - py * C0M_IW;
        const int gy = yoff + y0 + py - 2, gx = x0 + px - 2;
        const bool in = py < C0M_IH && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int cy = in ? gy : 0, cx = in ? gx : 0;
        const float v = Xb[((size_t)(cy >> 1) * W2 + (cx >> 1)) * CP + ((cy & 1) * 2 + (cx & 1)) * C3 + 3 * t + c];
        xreg[k] = in ? v : 0.f;
    }
    // weights: entry (ky, g, l) = 8 values k' = 8 (l >> 5) + e of row tap ky for channel 32 g + (l & 31); k' = kx * 3 + c, k' = 15: 0
#pragma unroll
    for (int k = 0; k < C0M_WIT; ++k) {
        const int ent = tid + k * 256;
        if (ent >= 5 * 2 * 64) break;
        const int l = ent & 63, g = (ent >> 6) & 1, ky = ent >> 7;
        unsigned short hv[8], lv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split1(wreg[k][e], hv[e], lv[e]);
        u32x4 hq, lq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hq[e] = hv[2 * e] | ((unsigned)hv[2 * e + 1] << 16);
            lq[e] = lv[2 * e] | ((unsigned)lv[2 * e + 1] << 16);
        }
        *reinterpret_cast<u32x4*>(sw + ((ky * 2 + 0) * 2 + g) * 1024 + l * 16) = hq;
        *reinterpret_cast<u32x4*>(sw + ((ky * 2 + 1) * 2 + g) * 1024 + l * 16) = lq;
    }
    // input tile (frame coordinates: rows outside a strip are real data), split once: hi | lo' << 16
#pragma unroll
    for (int k = 0; k < C0M_XIT; ++k) {
        const int i = tid + k * 256;
        if (i >= C0M_IN_DW) break;
        unsigned short hb, lb;
        split1(xreg[k], hb, lb);
        s_in[i] = hb | ((unsigned)lb << 16);
    }
    __syncthreads();

    const int xl = lane & 31, kh = lane >> 5;
    const float bias0 = bias[xl], bias1 = bias[32 + xl];
    const unsigned char* const wl = sw + lane * 16;
#pragma unroll 1
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr;                                  // output row of the tile
        if (y0 + row >= Hs) break;                                      // (wave-uniform)
        f32x16 am0, am1, ac0, ac1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            am0[r] = bias0;
            am1[r] = bias1;
            ac0[r] = 0.f;
            ac1[r] = 0.f;
        }
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) {
            const unsigned* ap = s_in + ((row + ky) * C0M_IW + xl) * 3 + 8 * kh;
            unsigned d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = ap[e];
            u32x4 ahq, alq;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ahq[e] = __builtin_amdgcn_perm(d[2 * e + 1], d[2 * e], 0x05040100u);   // the two hi halves
                alq[e] = __builtin_amdgcn_perm(d[2 * e + 1], d[2 * e], 0x07060302u);   // the two lo' halves
            }
            const h8 ah = __builtin_bit_cast(h8, ahq), al = __builtin_bit_cast(h8, alq);
            const h8 w0h = *reinterpret_cast<const h8*>(wl + ((ky * 2 + 0) * 2 + 0) * 1024);
            const h8 w1h = *reinterpret_cast<const h8*>(wl + ((ky * 2 + 0) * 2 + 1) * 1024);
            const h8 w0l = *reinterpret_cast<const h8*>(wl + ((ky * 2 + 1) * 2 + 0) * 1024);
            const h8 w1l = *reinterpret_cast<const h8*>(wl + ((ky * 2 + 1) * 2 + 1) * 1024);
            am0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w0h, am0, 0, 0, 0);
            am1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w1h, am1, 0, 0, 0);
            ac0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w0l, ac0, 0, 0, 0);
            ac1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w1l, ac1, 0, 0, 0);
            ac0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w0h, ac0, 0, 0, 0);
            ac1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w1h, ac1, 0, 0, 0);
        }
        // D register r of a lane = pixel drow(r, lane) of the row, channel xl (+32): 128-byte channel lines
        const size_t rowbase = (((size_t)f * Hs + y0 + row) * W + x0) * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = drow(r, lane);
            const float v0 = lrelu(am0[r] + ac0[r] * (1.0f / 2048.0f)), v1 = lrelu(am1[r] + ac1[r] * (1.0f / 2048.0f));
            if (x0 + px < W) {
                if (BF16OUT) {
                    uint16_t* o16 = reinterpret_cast<uint16_t*>(out) + rowbase + (size_t)px * 64 + xl;
                    o16[0] = __builtin_bit_cast(unsigned short, (__bf16)v0);   // round to nearest even, as the VALU kernel's convertvector
                    o16[32] = __builtin_bit_cast(unsigned short, (__bf16)v1);
                } else {
                    float* o = out + rowbase + (size_t)px * 64 + xl;
                    o[0] = v0;
                    o[32] = v1;
                }
            }
        }
    }
}

static bool conv0_use_valu() {
    static const int v = [] {
        const char* e = getenv("PFNL_CONV0");
        return (e && e[0] == 'v') ? 1 : 0;
    }();
    return v != 0;
}

hipError_t launch_conv0(const float* Xo, const float* w75x64, const float* bias, float* out, int B,
                        int T, int H, int W, hipStream_t s, const StripGeom* strip, bool force_f32) {
    const int yoff = strip ? strip->yoff : 0, Hs = strip ? strip->Hs : H;
    if (!conv0_use_valu() && !force_f32) {
        dim3 grid((W + C0M_TW - 1) / C0M_TW, (Hs + C0M_TH - 1) / C0M_TH, B * T);
        hipLaunchKernelGGL(conv0_mfma_kernel<false>, grid, dim3(256), 0, s, Xo, w75x64, bias, out, T, H, W, nl_padded_ch(12 * T), yoff, Hs);
        return hipGetLastError();
    }
    dim3 grid((W + C0_T - 1) / C0_T, (Hs + C0_T - 1) / C0_T, B * T);
    hipLaunchKernelGGL(conv0_kernel<false>, grid, dim3(256), 0, s, Xo, w75x64, bias, out, T, H, W,
                       nl_padded_ch(12 * T), yoff, Hs);
    return hipGetLastError();
}

hipError_t launch_conv0_bf16(const float* Xo, const float* w75x64, const float* bias, uint16_t* out, int B, int T, int H,
                             int W, hipStream_t s, const StripGeom* strip) {
    const int yoff = strip ? strip->yoff : 0, Hs = strip ? strip->Hs : H;
    if (!conv0_use_valu()) {
        dim3 grid((W + C0M_TW - 1) / C0M_TW, (Hs + C0M_TH - 1) / C0M_TH, B * T);
        hipLaunchKernelGGL(conv0_mfma_kernel<true>, grid, dim3(256), 0, s, Xo, w75x64, bias, reinterpret_cast<float*>(out), T, H, W,
                           nl_padded_ch(12 * T), yoff, Hs);
        return hipGetLastError();
    }
    dim3 grid((W + C0_T - 1) / C0_T, (Hs + C0_T - 1) / C0_T, B * T);
    hipLaunchKernelGGL(conv0_kernel<true>, grid, dim3(256), 0, s, Xo, w75x64, bias, reinterpret_cast<float*>(out), T, H, W,
                       nl_padded_ch(12 * T), yoff, Hs);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Keys cubic (A = -0.75) tap weights at fractional offset t; exact in f32 for t = k/4 and k/2.
__device__ __forceinline__ void cubic_taps(float t, float w[4]) {
    const float A = -0.75f;
    const float d0 = 1.f + t, d1 = t, d2 = 1.f - t, d3 = 2.f - t;
    w[0] = ((A * d0 - 5.f * A) * d0 + 8.f * A) * d0 - 4.f * A;
    w[1] = ((A + 2.f) * d1 - (A + 3.f)) * d1 * d1 + 1.f;
    w[2] = ((A + 2.f) * d2 - (A + 3.f)) * d2 * d2 + 1.f;
    w[3] = ((A * d3 - 5.f * A) * d3 + 8.f * A) * d3 - 4.f * A;
}

// src [H][W][3] with row pitch `pitch` floats; returns the 3 channels of HR pixel (oy, ox).
__device__ __forceinline__ void bicubic_px(const float* __restrict__ src, int H, int W, size_t pitch,
                                           int scale, int oy, int ox, float res[3]) {
    const int iy = oy / scale, ix = ox / scale;
    float wy[4], wx[4];
    cubic_taps((float)(oy % scale) / (float)scale, wy);
    cubic_taps((float)(ox % scale) / (float)scale, wx);
    res[0] = res[1] = res[2] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), H - 1);
        float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const int xx = min(max(ix - 1 + bb, 0), W - 1);
            const float* p = src + (size_t)yy * pitch + (size_t)xx * 3;
            row[0] = fmaf(wx[bb], p[0], row[0]);
            row[1] = fmaf(wx[bb], p[1], row[1]);
            row[2] = fmaf(wx[bb], p[2], row[2]);
        }
        res[0] = fmaf(wy[a], row[0], res[0]);
        res[1] = fmaf(wy[a], row[1], res[1]);
        res[2] = fmaf(wy[a], row[2], res[2]);
    }
}

__global__ void bicubic_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H,
                               int W, int scale) {
    const int OH = H * scale, OW = W * scale;
    const size_t total = (size_t)B * OH * OW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OW);
        const int oy = (int)((i / OW) % OH);
        const int b = (int)(i / ((size_t)OW * OH));
        float r[3];
        bicubic_px(x + (size_t)b * H * W * 3, H, W, (size_t)W * 3, scale, oy, ox, r);
        out[i * 3 + 0] = r[0];
        out[i * 3 + 1] = r[1];
        out[i * 3 + 2] = r[2];
    }
}

hipError_t launch_bicubic(const float* x, float* out, int B, int H, int W, int scale, hipStream_t s) {
    const size_t total = (size_t)B * H * W * scale * scale;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(bicubic_kernel, dim3(blocks), dim3(256), 0, s, x, out, B, H, W, scale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// tail.  merge [B][H][W][48] -> large1 = depth_to_space (virtual) [B][2H][2W][12]
//   out1 = conv3x3(large1; w2 [3][3][12][CO]) + b2, CO = 12 (4x) or 3 (2x)
//   4x: out[b][0][2Y+i][2X+j][c] = out1[Y][X][(2i+j)*3+c] + bic ;  2x: out[b][0][Y][X][c] = out1 + bic
// Thread = one (Y,X) of the 2H x 2W grid.  Weights are wave-uniform -> scalar loads.
template <int CO>
__global__ __launch_bounds__(256) void tail_kernel(const float* __restrict__ merge,
                                                   const float* __restrict__ x,   // [B][T][H][W][3]
                                                   const float* __restrict__ w2,  // [3][3][12][CO]
                                                   const float* __restrict__ b2,  // [CO]
                                                   float* __restrict__ out, int B, int T, int H, int W, int MS,   // MS = floats per merge pixel (48 or 64)
                                                   int yoff, int Hs, int core0, int core1,     // merge holds LR rows [yoff, yoff + Hs) of the H-row frame;
                                                                                                // only strip rows [core0, core1) are written
                                                   unsigned* __restrict__ nonfinite) {         // set to 1 when a value written is inf / NaN (or null)
    constexpr int SCALE = (CO == 12) ? 4 : 2;
    const int H2 = 2 * Hs, W2 = 2 * W;
    const int X = blockIdx.x * 32 + (threadIdx.x & 31);
    const int Y = blockIdx.y * 8 + (threadIdx.x >> 5);
    const int b = blockIdx.z;
    if (X >= W2 || Y >= H2 || (Y >> 1) < core0 || (Y >> 1) >= core1) return;
    const float* mb = merge + (size_t)b * Hs * W * MS;

    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = b2[o];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = Y + dy - 1;
        if (yy < 0 || yy >= H2) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = X + dx - 1;
            if (xx < 0 || xx >= W2) continue;
            // large1[yy][xx][k] = merge[yy/2][xx/2][((yy&1)*2+(xx&1))*12 + k]   (model/pfnl.py:76)
            const float4* src = reinterpret_cast<const float4*>(
                mb + ((size_t)(yy >> 1) * W + (xx >> 1)) * MS + ((yy & 1) * 2 + (xx & 1)) * 12);
            const float4 v0 = src[0], v1 = src[1], v2 = src[2];
            const float v[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
            const float* wp = w2 + (dy * 3 + dx) * 12 * CO;
#pragma unroll
            for (int k = 0; k < 12; ++k)
#pragma unroll
                for (int o = 0; o < CO; ++o) acc[o] = fmaf(v[k], wp[k * CO + o], acc[o]);
        }
    }

    const float* xc = x + (((size_t)b * T + T / 2) * H) * W * 3;   // centre frame, model/pfnl.py:63
    const int OH = SCALE * H, OW = SCALE * W;
    float* ob = out + (size_t)b * OH * OW * 3;
    // The f16-pipe kernels upstream (conv_split16.hip, conv_sf.hip, conv_small.hip, nonlocal_f16.hip, conv0_mfma) have a domain: an
    // operand beyond binary16's range becomes inf, and everything downstream of an inf stays non-finite up to here - the one place
    // every value of the result passes through.  One compare per value on a bandwidth-bound kernel; an atomic only when it fires.
    bool bad = false;
    if (CO == 12) {
        // The thread's 2 x 2 HR pixels (rows 2 Yf + i, columns 2 X + j) lie in ONE LR cell: floor((2 Yf + i) / 4) = Yf >> 1 and floor((2 X + j) / 4)
        // = X >> 1 for i, j in {0, 1} - they share the 4 x 4 window of clamped taps and differ only in the phase of the weights (2 (Yf & 1) + i,
        // 2 (X & 1) + j quarters).  The window is loaded ONCE (48 values, not 4 x 48) and the sums keep bicubic_px's order - over the columns with
        // wx, then over the rows with wy, both from 0 with fmaf: the same bits as the pixel-by-pixel form (round 6: 26 -> us at configs[1]).
        const int Yf = Y + 2 * yoff;                                // frame coordinates of the thread's row of the 2H x 2W grid
        const int iy = Yf >> 1, ix = X >> 1;
        float wy[2][4], wx[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            cubic_taps((float)(2 * (Yf & 1) + i) / 4.0f, wy[i]);
            cubic_taps((float)(2 * (X & 1) + i) / 4.0f, wx[i]);
        }
        float res[2][2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) res[i][j][0] = res[i][j][1] = res[i][j][2] = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), H - 1);
            float px[4][3];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const int xx = min(max(ix - 1 + bb, 0), W - 1);
                const float* p = xc + ((size_t)yy * W + xx) * 3;
                px[bb][0] = p[0];
                px[bb][1] = p[1];
                px[bb][2] = p[2];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    row[0] = fmaf(wx[j][bb], px[bb][0], row[0]);
                    row[1] = fmaf(wx[j][bb], px[bb][1], row[1]);
                    row[2] = fmaf(wx[j][bb], px[bb][2], row[2]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    res[i][j][0] = fmaf(wy[i][a], row[0], res[i][j][0]);
                    res[i][j][1] = fmaf(wy[i][a], row[1], res[i][j][1]);
                    res[i][j][2] = fmaf(wy[i][a], row[2], res[i][j][2]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = 2 * Yf + i;
            float v[6];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    v[3 * j + c] = acc[(2 * i + j) * 3 + c] + res[i][j][c];
                    bad |= !(fabsf(v[3 * j + c]) <= 3.4028234e38f);
                }
            // the thread's two pixels of the row are 24 contiguous bytes, 8-byte aligned when `out` is (24 X from a row start): three 8-byte stores
            float* dstf = ob + ((size_t)oy * OW + 2 * X) * 3;
            if ((reinterpret_cast<uintptr_t>(out) & 7) == 0) {      // (uniform over the launch)
                float2* dst = reinterpret_cast<float2*>(dstf);
                dst[0] = make_float2(v[0], v[1]);
                dst[1] = make_float2(v[2], v[3]);
                dst[2] = make_float2(v[4], v[5]);
            } else {
#pragma unroll
                for (int k = 0; k < 6; ++k) dstf[k] = v[k];
            }
        }
    } else {
        float bic[3];
        bicubic_px(xc, H, W, (size_t)W * 3, SCALE, Y + 2 * yoff, X, bic);
        float* dst = ob + ((size_t)(Y + 2 * yoff) * OW + X) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = acc[c] + bic[c];
            bad |= !(fabsf(v) <= 3.4028234e38f);
            dst[c] = v;
        }
    }
    // sticky word, only ever set to 1 by the device: a system-scope store (the word may live in pinned, device-mapped host memory)
    if (bad && nonfinite) __hip_atomic_store(nonfinite, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_tail(const float* merge, const float* x, const float* w2, const float* b2, float* out,
                       int B, int T, int H, int W, int scale, int merge_cstride, hipStream_t s, const StripGeom* strip, unsigned* nonfinite) {
    if (merge_cstride != 48 && merge_cstride != 64) return hipErrorInvalidValue;
    const StripGeom g = strip ? *strip : StripGeom{0, H, 0, H};
    dim3 grid((2 * W + 31) / 32, (2 * g.Hs + 7) / 8, B);
    if (scale == 4)
        hipLaunchKernelGGL(tail_kernel<12>, grid, dim3(256), 0, s, merge, x, w2, b2, out, B, T, H, W, merge_cstride, g.yoff, g.Hs, g.core0, g.core1, nonfinite);
    else if (scale == 2)
        hipLaunchKernelGGL(tail_kernel<3>, grid, dim3(256), 0, s, merge, x, w2, b2, out, B, T, H, W, merge_cstride, g.yoff, g.Hs, g.core0, g.core1, nonfinite);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// blur + decimate (the step BEFORE the path in test_video_truth / eval: reference utils.py:95-105,
// 169-192): reflect-pad 6 (edge sample not repeated), 13x13 Gaussian sigma 1.6 (float32 taps, as the
// reference's BLUR.astype(float32)), depthwise, stride `scale`, VALID.  hr [F][H][W][3] -> lr
// [F][ceil(H/s)][ceil(W/s)][3].  HBM-bound: one thread per output pixel, taps in constant memory.
__constant__ float c_blur[13 * 13];

__global__ void blur_decimate_kernel(const float* __restrict__ hr, float* __restrict__ lr, int F, int H, int W,
                                     int oh, int ow, int scale) {
    const size_t total = (size_t)F * oh * ow;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % ow);
        const int oy = (int)((i / ow) % oh);
        const int f = (int)(i / ((size_t)ow * oh));
        const float* src = hr + (size_t)f * H * W * 3;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int ky = 0; ky < 13; ++ky) {
            int y = oy * scale + ky - 6;
            y = y < 0 ? -y : (y >= H ? 2 * H - 2 - y : y);          // tf.pad(mode='REFLECT')
            for (int kx = 0; kx < 13; ++kx) {
                int x = ox * scale + kx - 6;
                x = x < 0 ? -x : (x >= W ? 2 * W - 2 - x : x);
                const float w = c_blur[ky * 13 + kx];
                const float* p = src + ((size_t)y * W + x) * 3;
                a0 = fmaf(w, p[0], a0);
                a1 = fmaf(w, p[1], a1);
                a2 = fmaf(w, p[2], a2);
            }
        }
        lr[i * 3 + 0] = a0;
        lr[i * 3 + 1] = a1;
        lr[i * 3 + 2] = a2;
    }
}

hipError_t launch_blur_decimate(const float* hr, float* lr, int F, int H, int W, int scale, hipStream_t s) {
    static std::atomic<int> init[64];                               // (per device; a racing first call repeats the same upload)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !init[dev]) {
        // scipy.ndimage.gaussian_filter(delta, 1.6) on a 13x13 grid == outer(k, k), k truncated at +-6
        // and normalised (verified to 1e-17, tests/test_host.py)
        double k[13], sum = 0.0;
        for (int i = 0; i < 13; ++i) {
            const double d = (i - 6) / 1.6;
            k[i] = exp(-0.5 * d * d);
            sum += k[i];
        }
        float w[169];
        for (int a = 0; a < 13; ++a)
            for (int b = 0; b < 13; ++b) w[a * 13 + b] = (float)((k[a] / sum) * (k[b] / sum));
        e = hipMemcpyToSymbol(HIP_SYMBOL(c_blur), w, sizeof(w));
        if (e != hipSuccess) return e;
        init[dev] = 1;
    }
    const int oh = (H + scale - 1) / scale, ow = (W + scale - 1) / scale;
    const size_t total = (size_t)F * oh * ow;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(blur_decimate_kernel, dim3(blocks), dim3(256), 0, s, hr, lr, F, H, W, oh, ow, scale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// MFMA layout self-test: D = A(32x2) * B(2x32) with asymmetric integer-valued operands.
__global__ void mfma_selftest_kernel(int* mismatches) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, k = lane >> 5;
    const float a = (float)(3 * i + 7 * k + 1);        // A[i][k]
    const float bv = (float)(5 * (lane & 31) - 11 * k + 2);  // B[k][j], j = lane&31
    f32x16 d;
    for (int r = 0; r < 16; ++r) d[r] = 0.f;
    d = mfma32(a, bv, d);
    int bad = 0;
    for (int r = 0; r < 16; ++r) {
        const int row = drow(r, lane), col = lane & 31;
        float ref = 0.f;
        for (int kk = 0; kk < 2; ++kk)
            ref += (float)(3 * row + 7 * kk + 1) * (float)(5 * col - 11 * kk + 2);
        if (d[r] != ref) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

hipError_t run_mfma_selftest(int* mismatches) {
    int* d = nullptr;
    hipError_t e = hipMalloc(&d, sizeof(int));
    if (e != hipSuccess) return e;
    hipMemset(d, 0, sizeof(int));
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, 0, d);
    e = hipMemcpy(mismatches, d, sizeof(int), hipMemcpyDeviceToHost);
    hipFree(d);
    return e;
}

}  // namespace pfnl

// ------------------------------------------------------------------------------------------------
// Harness helpers (reference model/pfnl.py:238-242, 254-257): the LR sequence is uploaded once; the clamped T-frame
// windows of a batch are gathered on the device, and the SR frames are quantised to uint8 there (a quarter of the D2H).
namespace pfnl {

// frames [F][frame_f4 float4] -> win [count][T][frame_f4]: window w, slot t = frame clamp(first + w + t - T/2, 0, F-1)
__global__ __launch_bounds__(256) void gather_windows_kernel(const f32x4* __restrict__ frames, f32x4* __restrict__ win, int F,
                                                             int first, int count, int T, size_t frame_f4) {
    const size_t total = (size_t)count * T * frame_f4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t e = i % frame_f4;
        const int wt = (int)(i / frame_f4);
        const int w = wt / T, t = wt - w * T;
        int f = first + w + t - T / 2;
        f = f < 0 ? 0 : (f > F - 1 ? F - 1 : f);
        win[i] = frames[(size_t)f * frame_f4 + e];
    }
}

hipError_t launch_gather_windows(const float* frames, float* win, int F, int first, int count, int T, size_t frame_floats,
                                 hipStream_t s) {
    if (frame_floats % 4) return hipErrorInvalidValue;
    const size_t total = (size_t)count * T * (frame_floats / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(gather_windows_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, reinterpret_cast<const f32x4*>(frames),
                       reinterpret_cast<f32x4*>(win), F, first, count, T, frame_floats / 4);
    return hipGetLastError();
}

// uint8(np.round(np.clip(sr * 255, 0, 255))): rintf = round half to even, like np.round
__global__ __launch_bounds__(256) void quantise_u8_kernel(const f32x4* __restrict__ sr, uint32_t* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = sr[i];
        const uint32_t a = (uint32_t)rintf(fminf(fmaxf(v.x * 255.0f, 0.f), 255.f));
        const uint32_t b = (uint32_t)rintf(fminf(fmaxf(v.y * 255.0f, 0.f), 255.f));
        const uint32_t c = (uint32_t)rintf(fminf(fmaxf(v.z * 255.0f, 0.f), 255.f));
        const uint32_t d = (uint32_t)rintf(fminf(fmaxf(v.w * 255.0f, 0.f), 255.f));
        out[i] = a | (b << 8) | (c << 16) | (d << 24);
    }
}

hipError_t launch_quantise_u8(const float* sr, uint8_t* out, size_t n, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(quantise_u8_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, reinterpret_cast<const f32x4*>(sr),
                       reinterpret_cast<uint32_t*>(out), n4);
    return hipGetLastError();
}

}  // namespace pfnl

// bf16 trunk kernels (conv_bf16.hip): declarations shared with capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pfnl {

struct ConvBf16Params {
    const uint16_t* in;      // [items][H][W][64] bf16
    const uint16_t* wpack;   // conv3x3_bf16_pack_weights
    const float* bias;       // [64] f32; never null
    const uint16_t* addend;  // [items/add_div][H][W][64] bf16, added before the activation \ both or
    const uint16_t* resid;   // [items][H][W][64] bf16, added after the activation           / neither
    uint16_t* out;           // [items][H][W][64] bf16 (may alias resid)
    int H, W, items, add_div, act;
    // conv1_i + conv10_i in one launch (addend == resid == nullptr): per chain of add_div frames
    // x_out[clip] = leaky_relu(sum_t W10_t out_t + x_bias)
    const uint16_t* x_w;     // conv1x1_bf16_pack_weights (add_div frames); nullptr = plain launch
    const float* x_bias;     // [64]
    uint16_t* x_out;         // [items/add_div][H][W][64] bf16
    // convmerge1 (accumulating mode; out == addend == x_out == nullptr): wpack = add_div consecutive packs (one per frame of
    // a clip), out_f32[clip] = act(sum_t conv(in[clip*add_div + t]; W_t) + bias), fp32 [items/add_div][H][W][64]
    float* out_f32;
    // SPLIT CHAINS (third-generation kernel, modes 1 and 2; split_s = 0: off; conv_split16.h has the scheme): the first n_full (clip, tile) chains - a
    // whole number of rounds of the grid - whole, each chain behind them cut by frames into split_s parts of <= split_q frames, one per workgroup.
    // Mode 2: a part leaves its share of conv10_i's sum as raw fp32 in partial[slot] ([8][32][64] floats) and launch_c10_finalize_bf16 builds x_out
    int n_full, split_s, split_q;
    float* partial;
    int flat;                // addend launches on the third-generation kernel: 1 = deal the tiles out one by one instead of as chains of the add_div
                             // frames of a (clip, tile) (capi.hip, "MID shapes": fewer chains than workgroups); same result bit for bit
    const uint16_t* wpack16; // conv3x3_bf16_pack_weights16 of the same kernel (or null): modes 1 / 2 of the third-generation kernel then run on
                             // v_mfma_f32_16x16x32_bf16 (conv_bf16_v3.hip, M16)
};
hipError_t launch_conv3x3_bf16(const ConvBf16Params& p, hipStream_t s);
// the second-generation kernel (conv_bf16_v2.hip: halo by LDS-DMA, MFMA groups with nothing else in them, a serial epilogue phase) for
// modes 0 (plain), 1 (addend + resid), 2 (conv1_i + conv10_i); launch_conv3x3_bf16 takes it for the modes the third generation does not (PFNL_BF16_V3)
hipError_t launch_conv3x3_bf16_v2(const ConvBf16Params& p, int mode, hipStream_t s);
// the third-generation kernel (conv_bf16_v3.hip: the two halves of the workgroup half a tile period apart), same modes
hipError_t launch_conv3x3_bf16_v3(const ConvBf16Params& p, int mode, hipStream_t s);
// split chains: x_out[clip][tile of chain n_full + j] = bf16(lrelu(sum over the parts r of partial[j * split_s + r] + x_bias)) (p: H, W, items, add_div,
// n_full, split_s, partial, x_bias, x_out)
hipError_t launch_c10_finalize_bf16(const ConvBf16Params& p, hipStream_t s);
hipError_t launch_conv1x1_bf16(const uint16_t* in, const uint16_t* wpack, const float* bias, uint16_t* out, int items, int T,
                               int HW, int act, hipStream_t s);
hipError_t launch_cast_bf16_f32(const uint16_t* in, float* out, size_t n, hipStream_t s);   // n % 8 == 0
hipError_t launch_cast_f32_bf16(const float* in, uint16_t* out, size_t n, hipStream_t s);
uint16_t bf16_rne(float f);
size_t conv3x3_bf16_pack_halfs();
void conv3x3_bf16_pack_weights(const float* hwio, int cin_total, int cin_begin, uint16_t* dst, int cout = 64);   // cout < 64: zero-padded
void conv3x3_bf16_pack_weights16(const float* hwio, int cin_total, int cin_begin, uint16_t* dst);   // the M16 form's order (ConvBf16Params::wpack16), 64 output channels
size_t conv1x1_bf16_pack_halfs(int T);
void conv1x1_bf16_pack_weights(const float* hwio, int T, uint16_t* dst);
// conv0 writing the bf16 trunk input (misc_kernels.hip)
struct StripGeom;
hipError_t launch_conv0_bf16(const float* Xo, const float* w75x64, const float* bias, uint16_t* out, int B, int T, int H,
                             int W, hipStream_t s, const StripGeom* strip = nullptr);

}  // namespace pfnl

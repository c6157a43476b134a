// Winograd F(2x2,3x3) 64->64 convolution on the f16 matrix pipe with exactly split operands (conv_wsplit.hip): declarations shared
// with capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pfnl {

struct ConvWsParams {
    const float* in;         // [items][H][W][64] fp32
    const uint16_t* upack;   // conv_wsplit_pack_weights: U = G g G^T, 16 positions x 64 x 64, split into binary16 pairs
    const float* bias;       // [64] f32; never null
    const float* addend;     // [items/add_div][H][W][64] f32, added before the activation \ both or
    const float* resid;      // [items][H][W][64] f32, added after the activation           / neither
    float* out;              // [items][H][W][64] f32 (may alias resid)
    int H, W, items, add_div, act;
};

// out = act(conv3x3(in) + bias [+ addend[item / add_div]]) [+ resid]; any H, W
hipError_t launch_conv_wsplit(const ConvWsParams& p, hipStream_t s);
size_t conv_wsplit_pack_halfs();                                      // 16-bit elements of one packed kernel (131 072)
// HWIO [3,3,cin_total,cout] rows [cin_begin, cin_begin + 64) -> the MFMA B operands of the 16 Winograd positions (cout < 64: zero-padded)
void conv_wsplit_pack_weights(const float* hwio, int cin_total, int cin_begin, uint16_t* dst, int cout = 64);

}  // namespace pfnl

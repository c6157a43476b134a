// Small-shape convolutions of the trunk (conv_small.hip): declarations shared with capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pfnl {

// out[i] = act(sum over the sources s of conv_ks(src(i, s); weights of source s) + bias) (+ resid[i]); all tensors fp32 NHWC, 64 channels.
// Source s of output item i: s < nA ? a[i / a_div] : b[i * b_mul + (s - nA)]
//   conv1_i:    nA 0, b = inp0, b_mul 1, nsrc 1, items = frames                (reference model/pfnl.py:66)
//   conv10_i:   nA 0, b = inp1, b_mul T, nsrc T, items = clips, ks 1           (:67-68)
//   conv2_i:    a = base, nA 1, a_div T, b = inp1, b_mul 1, nsrc 2, resid = out = inp0, items = frames   (:69-71, concat([base, f]) as written)
//   convmerge1: nA 0, b = inp0, b_mul T, nsrc T, items = clips, cout 48 zero-padded to 64   (:73-74)
struct ConvSmallParams {
    const float* a;          // may be null when nA == 0
    const float* b;
    int nA, a_div, b_mul, nsrc;
    const uint16_t* wpack;   // conv_small_pack_weights
    const float* bias;       // [64] f32 (zero-padded); never null
    const float* resid;      // [items][H][W][64] or null; may alias out
    float* out;              // [items][H][W][64]
    int H, W, items, act, ks;
    // Two launches per progressive-fusion block at small shapes (round 4) - conv10_i has no launch of its own:
    //  * the conv1_i launch (ks 3, nsrc 1) also takes its finished, activated tile through the 1x1: x_out[item] = the PARTIAL sum
    //    W10[64 t .. 64 t + 63, :]^T . inp1_t of this frame (t = item % x_T), no bias, no activation, fp32 NHWC [items][H][W][64];
    //  * the conv2_i launch builds its source `a` (= base, model/pfnl.py:67-68) from them on the way into LDS:
    //    a(i) = lrelu(sum over j < a_nsum of a[(i / a_div) * a_nsum + j] + a_bias) - the split-K partials meet in the NEXT kernel's
    //    prologue, in the fixed order j = 0 .. a_nsum - 1 (deterministic), with no inter-workgroup traffic inside a launch.
    const uint16_t* x_wpack; // conv_small_pack_weights(ks 1, nsrc x_T) of conv10_i, or null
    float* x_out;
    int x_T;
    int a_nsum;              // 0 / 1: `a` is a plain tensor
    const float* a_bias;     // [64], with a_nsum > 1
};
hipError_t launch_conv_small(const ConvSmallParams& p, hipStream_t s);
size_t conv_small_pack_halfs(int ks, int nsrc);                       // 16-bit elements
void conv_small_pack_weights(const float* hwio, int ks, int nsrc, int cout, uint16_t* dst);   // HWIO [ks, ks, 64 nsrc, cout]

}  // namespace pfnl

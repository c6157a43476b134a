// fp32 convolution on the f16 matrix pipe with split operands (conv_split16.hip): declarations shared with capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// schedule switches of the two launches of a progressive-fusion block (conv3x3_c1c10_kernel, conv3x3_sf_chain_kernel)
#ifndef PFNL_S16_SPREAD
#define PFNL_S16_SPREAD 1    // 1: the next sub-step's operand reads between the MFMAs of the current one instead of in front of them
#endif
#ifndef PFNL_S16_PRIO
#define PFNL_S16_PRIO 1      // > 0: s_setprio of waves 4-7 (the younger wave of every SIMD)
#endif

namespace pfnl {

struct ConvSplitParams {
    const float* in;         // [items][H][W][64] fp32
    const uint16_t* wpack;   // conv3x3_split16_pack_weights (both channel halves)
    const float* bias;       // [64] f32; never null
    const float* addend;     // [items/add_div][H][W][64] f32, added before the activation \ both or
    const float* resid;      // [items][H][W][64] f32, added after the activation           / neither
    float* out;              // [items][H][W][64] f32 (may alias resid)
    int H, W, items, add_div, act;
    int accum;               // 1: out[items/add_div] = act(sum over the add_div frames of an item group + bias); wpack = add_div packs (convmerge1)
    int out_sf;              // 1 (plain mode only): `out` is written in the split format below instead of fp32
    const float* in2;        // launch_conv3x3_sf_chain only: `base` [items/add_div][H][W] in the split format (the shared half's input)
    const uint16_t* wpack2;  // ... and the packed kernel rows that multiply it (identity rows); launch_conv3x3_c1c10: conv10_i (conv1x1_c10_pack_weights)
    const float* bias2;      // launch_conv3x3_c1c10 only: conv10_i's bias [64]
    float* out2;             // ... and its output `base` [items/add_div][H][W] in the split format
    // SPLIT CHAINS (launch_conv3x3_c1c10, launch_conv3x3_sf_chain; split_s = 0: off): the first n_full (clip, tile) chains - a whole number of
    // rounds of the grid - are dealt out whole; each chain behind them is cut by frames into split_s parts of <= split_q frames, one part per
    // workgroup.  c1c10: a part leaves its share of conv10_i's sum as raw fp32 in partial[slot] ([8][32][64] floats per slot, slot = split
    // chain * split_s + part; part 0 includes the bias) and launch_c10_finalize builds `base`; the chain kernel recomputes the shared half per part
    int n_full, split_s, split_q;
    float* partial;
    int in_sf;               // launch_conv3x3_c1c10 only: 1 = `in` is the split-format copy of inp0 (launch_conv3x3_sf_chain's out2): halo by LDS-DMA
    int flat;                // launch_conv3x3_sf with an addend only: 1 = deal the tiles out one by one instead of as chains of the add_div frames
                             // of a (clip, tile) - for launches with fewer chains than workgroups (capi.hip, "MID shapes").  The two input-channel
                             // halves of a tile are summed in the order its position in the workgroup's sequence gives (boustrophedon:
                             // the weights in LDS serve two units in a row), so the work order moves the last bit of some results
    const uint16_t* wpack_m16;   // launch_conv3x3_sf_chain only, both or neither: conv3x3_split16_pack_weights16 of wpack / wpack2 - the launch then runs on
    const uint16_t* wpack2_m16;  // v_mfma_f32_16x16x32_f16 (conv3x3_sf_chain16_kernel).  Not with out2
};

// THE SPLIT FORMAT ("SF") of an activation tensor that only ever feeds MFMA operands (conv1_i's output, conv10_i's output):
// [items][H][W] pixels of 256 B = [channel half M = 0, 1][part: hi, lo'][32 channels] binary16, where for a value x
// hi = f16(x), lo' = f16((x - hi) 2^11) - exactly the two operands conv_split16.hip builds from an fp32 tensor when it commits a
// halo tile, built once by the producer instead.  Same bytes per value as fp32.  16-byte chunk c of a (pixel, half) = channels
// 32 M + 8 (c & 3) .. + 7 of part c >> 2: the unit the 3x3 kernels' LDS tiles and the 1x1 kernel's operands are made of.
hipError_t launch_conv3x3_split16(const ConvSplitParams& p, hipStream_t s);
size_t conv3x3_split16_pack_halfs();                                  // 16-bit elements per packed 3x3 64->64 kernel
void conv3x3_split16_pack_weights16(const float* hwio, int cin_total, int cin_begin, uint16_t* dst);   // the order of the 16x16x32 kernels: [half][kx][ky][16-channel output tile][hi / lo'][lane] (64 output channels)
void conv3x3_split16_pack_weights(const float* hwio, int cin_total, int cin_begin, uint16_t* dst, int cout = 64,   // cout < 64: zero-padded
                                  bool identity_rows = false);        // true: the pack conv3x3_sf_kernel takes

// conv1_i + conv10_i of a progressive-fusion block in one launch (reference model/pfnl.py:66-68): in fp32 [clips*T][H][W][64], add_div = T;
// out = inp1 and out2 = base, both in the split format; act applies to both convolutions
hipError_t launch_conv3x3_c1c10(const ConvSplitParams& p, hipStream_t s);
// split chains: base[clip][tile of chain n_full + j] = split format of act(sum over the parts r of partial[j * split_s + r]), j < nchains - n_full
// (p: H, W, items = clips * T, add_div = T, act, n_full, split_s, partial, out2 = base)
hipError_t launch_c10_finalize(const ConvSplitParams& p, hipStream_t s);
// the grid the persistent split-f16 launches use on the current device (whole XCDs): what n_full must be a multiple of
int conv_split16_grid();
size_t conv1x1_c10_pack_halfs(int T);
void conv1x1_c10_pack_weights(const float* hwio, int T, uint16_t* dst);

// 3x3 64->64 with the INPUT in the split format (conv_sf.hip): halo tiles by LDS-DMA, epilogue from registers.  `in` points at SF
// data ([items][H][W] x 256 B); wpack = conv3x3_split16_pack_weights(..., identity_rows = true); plain and fused (addend + resid) modes.
hipError_t launch_conv3x3_sf(const ConvSplitParams& p, hipStream_t s);
// the whole of conv2_i in one launch: per (clip, tile) the shared half (in2, wpack2) stays in registers as the initial C of the
// add_div frame tiles (in, wpack) that follow; bias, leaky-relu and the residual in the epilogue; out may alias resid.
// out2 != null: the result is ALSO written in the split format ([items][H][W] x 256 B) - the next block's launch_conv3x3_c1c10 input (in_sf)
hipError_t launch_conv3x3_sf_chain(const ConvSplitParams& p, hipStream_t s);
hipError_t launch_sf_from_f32(const float* in, uint16_t* out, size_t npix, hipStream_t s);   // [npix][64] fp32 -> SF (tests / taps)
hipError_t launch_sf_to_f32(const uint16_t* in, float* out, size_t npix, hipStream_t s);     // SF -> hi + lo' 2^-11

// non-local block of the fp32 path on the f16 matrix pipe with exactly split operands (nonlocal_f16.hip); arguments as launch_nl_attn
// (common.h) plus the operand scratch; split = false: the hi parts only (the non-local block of precision=bf16)
size_t nl_f16_scratch_halfs(int B, int N);
hipError_t launch_nl_attn_f16(const float* X, float* Xo, const float* Wp, const float* bp, float* partial, uint16_t* scratch16,
                              int B, int N, int C, hipStream_t s, int q0 = 0, int q1 = -1, bool split = true,
                              bool prepacked = false);   // prepacked: launch_nl_pack_fused has filled X and scratch16 (no pack launch here)
// round 6: nl_pack_kernel's and nl_pack_f16_kernel's jobs in ONE launch: x [B][T][H][W][3] -> X [B][N][CP] fp32 + the binary16 operand arrays
// (for batches whose packed operands fit one launch: nl_f16_fits_one_launch; larger ones keep launch_nl_pack + the chunked launch_nl_attn_f16)
bool nl_f16_fits_one_launch(int B, int N);
hipError_t launch_nl_pack_fused(const float* x, float* X, uint16_t* scratch16, int B, int T, int H, int W, hipStream_t s);

}  // namespace pfnl

// bf16 trunk, third generation of the 3x3 64->64 kernel (round 4): conv1_i + conv10_i, both halves of conv2_i (reference
// model/pfnl.py:49-51, 66-71).  Same arithmetic, LDS layouts, packed weights, work order and epilogue pieces as conv_bf16_v2.hip (whose
// header explains the tile: 8 x 32 pixels, halo 10 x 34 x 128 B by LDS-DMA into one of two buffers, 12 MFMA groups, a serial epilogue
// phase in the tile's own halo buffer); what changes is that the two HALVES of the workgroup run half a tile period apart.
//
// Why.  In the second generation both waves of a SIMD reach the serial phase together, so the matrix pipe idles for its whole length
// (tools/b2_timing.py: 2.3 - 3 k of a tile's 8.9 - 9.9 k cycles).  Here waves 0-3 (group A = tile rows 0-3) and waves 4-7 (group B =
// rows 4-7) process the SAME tile sequence, group B two workgroup barriers behind group A.  A tile is four intervals -
//     I0  MFMA groups 0-5        I1  MFMA groups 6-11        I2  epilogue dump (+ halo requests)        I3  lines out (+ conv10_i)
// - so one half's I2 | I3 always faces the other half's I0 | I1: on every SIMD one wave feeds the matrix pipe while its partner does
// the VALU / LDS / memory work.  Two rules follow from the phase stamps of the first attempts (tools/experiments/README.md, round 4):
//   * a wave that is ALONE on its matrix pipe pays for everything it issues between two MFMAs beyond ~5 slots (MI355X_MICROARCH.md,
//     "one wave per SIMD"): the 7 operand reads of the next group go one or two per MFMA gap, pinned in source order (a clump of 7
//     before the 6 MFMAs: 350 cycles per group instead of 230), and
//   * it issues NO vector-memory instruction (each one cost 90 - 185 cycles among the MFMAs): every store, halo piece and epilogue input
//     is issued by the half that is in its serial phase, which has the slack.
// What makes it fit without more LDS:
//   * group A's scratch = bytes [0, 16 K) of the tile's buffer = halo rows 0 - 3.8, which only group A reads; group B's scratch = bytes
//     [B3_SCR_B, + 16 K) = halo rows 6 - 9.8, which only group B reads;
//   * the halo of tile u + 2 goes into tile u's buffer in two parts: group B issues the pieces below its scratch (i < 25) in ITS dump
//     interval of tile u (group A's scratch there was read one interval earlier, and nobody reads halo u any more), group A issues the
//     rest (i >= 25) in its dump interval of tile u + 1 (group B's lines of tile u left one interval earlier).  Each wave waits for its
//     own pieces (an explicit vmcnt wait) before the barrier that precedes group A's first read of that halo.
// All synchronisation is the workgroup barrier - 4 per tile, the same count for every wave (group B's two leading barriers pair with
// group A's two trailing ones): no flags, no spinning.
#include <type_traits>

#include "common.h"
#include "conv_bf16.h"

#ifndef B3_STORE_AUX
#define B3_STORE_AUX 0
#endif

namespace pfnl {

typedef __bf16 b3h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b3h4 __attribute__((ext_vector_type(4)));
typedef unsigned b3u4 __attribute__((ext_vector_type(4)));
typedef unsigned b3u2 __attribute__((ext_vector_type(2)));

constexpr int B3_THREADS = 512, B3_GTHREADS = 256;
constexpr int B3_TH = 8, B3_TW = 32, B3_IH = 10, B3_IW = 34;
constexpr int B3_NDMA = (B3_IH * B3_IW + 7) / 8;                    // 43 DMA instructions of 8 pixels x 128 B
constexpr int B3_TILE_BYTES = B3_NDMA * 1024;                       // 44 032 (the last instruction's 4 surplus pixels land in padding)
constexpr int B3_W_BYTES = 9 * 4 * 2 * 1024;                        // 73 728: conv3x3_bf16_pack_weights
constexpr int B3_LDS_BYTES = 2 * B3_TILE_BYTES + B3_W_BYTES + 2 * 64 * 4;   // 162 304 of 163 840 (bias, conv10_i's bias)
constexpr int B3_SCR_B = 6 * B3_IW * 128;                           // 26 112: group B's scratch starts at halo row 6
constexpr int B3_LOW = 25;                                          // pieces i < 25 (bytes < 25 600) lie below group B's scratch
constexpr int B3_PIECES = 7;                                        // per wave: group B i = wl + 4k < 25, group A i = 25 + wl + 4k < 43
static_assert(B3_LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(B3_LOW * 1024 <= B3_SCR_B && B3_SCR_B + 128 * 128 <= B3_TILE_BYTES, "group B's scratch: above the low pieces, inside the buffer");
static_assert(128 * 128 <= 4 * B3_IW * 128, "group A's scratch inside halo rows 0-3 (group B reads rows 4-9)");

__device__ __forceinline__ void b3_dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}
__device__ __forceinline__ void b3_dma4(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff) {   // one dword per lane: LDS [lds_dst + 4 L]
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}
__device__ __forceinline__ f32x16 b3_mfma(b3h8 a, b3h8 b, f32x16 c) {
#ifdef PFNL_X_MFMA16   /* timing experiment only (wrong results on purpose): the same FLOPs as two v_mfma_f32_16x16x32_bf16 on the same operand registers */
    {
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        f32x4_ lo = {c[0], c[1], c[2], c[3]}, hi = {c[4], c[5], c[6], c[7]};
        lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, hi, 0, 0, 0);
        c[0] = lo[0]; c[1] = lo[1]; c[2] = lo[2]; c[3] = lo[3];
        c[4] = hi[0]; c[5] = hi[1]; c[6] = hi[2]; c[7] = hi[3];
        return c;
    }
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 b3_to_f32(b3u2 v) {
    return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                 __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 b3_lrelu4(f32x4 v, float slope) {
    const f32x4 sv = v * slope;
    asm("v_max_f32 %0, %1, %2" : "=v"(v.x) : "v"(v.x), "v"(sv.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(v.y) : "v"(v.y), "v"(sv.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(v.z) : "v"(v.z), "v"(sv.z));
    asm("v_max_f32 %0, %1, %2" : "=v"(v.w) : "v"(v.w), "v"(sv.w));
    return v;
}
__device__ __forceinline__ b3u2 b3_to_bf16(f32x4 v) {             // round to nearest even (v_cvt_pk_bf16_f32)
    const b3h4 b = __builtin_convertvector(v, b3h4);
    return __builtin_bit_cast(b3u2, b);
}
#define B3_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef PFNL_B3_TIMING   /* phase timeline (tools/b3_timing.py); not part of the product build */
__device__ long long b3_dbg[256 * 2 * 128];
#ifndef PFNL_B3_TIMING_MODE
#define PFNL_B3_TIMING_MODE 2
#endif
#define B3_STAMP() do { if (MODE == PFNL_B3_TIMING_MODE && lane == 0 && (wave == 0 || wave == 5) && dbg_n < 128) b3_dbg[(blockIdx.x * 2 + (wave != 0)) * 128 + dbg_n++] = __builtin_readcyclecounter(); } while (0)
#ifdef PFNL_B3_PSTAMPS
#define B3_PSTAMP() B3_STAMP()
#else
#define B3_PSTAMP() do {} while (0)
#endif
#else
#define B3_STAMP() do {} while (0)
#define B3_PSTAMP() do {} while (0)
#endif

// (M16) the K = 32 shape: 16 x 16 outputs, 4 accumulator registers - 14 % less energy per FLOP than 32x32x16 under the package power cap
typedef float b3f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ b3f4 b3_mfma16(b3h8 a, b3h8 b, b3f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// MODE 0: out = act(conv + bias).  MODE 1 (conv2_i per-frame half): out = act(conv + bias + addend[item / add_div]) + resid.
// MODE 2 (conv1_i + conv10_i): MODE 0, and per chain of add_div frames x_out = lrelu(sum_t W10_t out_t + x_bias).
// SPLIT (round 6, modes 1 and 2): split chains - behind the whole rounds of (clip, tile) chains every workgroup takes ONE part (frames [sp_f0, sp_f1)) of a
// cut chain (conv_split16.h).  Mode 1: the part fetches the chain's addend pieces with its own first frame; mode 2: its share of conv10_i's sum leaves raw
// (fp32, no bias, no activation) in p.partial[slot] and c10_finalize_bf16_kernel adds the parts.  A template parameter: the whole-round launches keep their stream.
// M16 (round 6, late): the 3x3's products on v_mfma_f32_16x16x32_bf16.  A = weights (16 output channels x 32 input channels), B = pixels (32 channels x 16
// pixels): lane (n = l & 15, g = l >> 4) reads chunk 4 kk + g (8 channels) of halo pixel 16 ph + n + kx for the k-step kk of 32 channels, and owns, per output row
// and pixel half ph, the 8 CONSECUTIVE output channels 32 mt + 8 g .. + 7 of pixel 16 ph + n (two 16-channel output tiles ct = 0, 1 whose rows the weight pack
// - conv3x3_bf16_pack_weights16, p.wpack16 - permutes: row 4 g + r of tile ct = channel 8 g + 4 ct + r): one 16-byte piece of a scratch line, one 16-byte residual
// / addend piece.  6 groups (column tap kx, k-step kk) of 24 MFMAs per tile instead of 12 of 6; LDS image, scratch lines, conv10_i's stage, the work order,
// the two half-workgroups and every barrier are those of the 32x32x16 form.  Summation order differs (K = 32 per instruction): not bit-equal to it.
template <int MODE, bool SPLIT = false, bool M16 = false>
__global__ __launch_bounds__(B3_THREADS, 1) void conv3x3_bf16_v3_kernel(ConvBf16Params p) {
    constexpr bool FUSE = MODE == 1;
    constexpr bool WITH10 = MODE == 2;
    static_assert(!SPLIT || MODE != 0, "split chains: the chained modes only");
    extern __shared__ __attribute__((aligned(16))) unsigned char b3_smem[];
    unsigned char* const wl = b3_smem + 2 * B3_TILE_BYTES;
    float* const bl = reinterpret_cast<float*>(b3_smem + 2 * B3_TILE_BYTES + B3_W_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PFNL_B3_TIMING
    int dbg_n = 0;
#endif
    const int rp = wave >> 1;                                       // rows 2rp, 2rp+1 of the tile
    const int mt = wave & 1;                                        // output channels 32mt .. 32mt+31
    const int grp = wave >> 2;                                      // half of the workgroup: 0 = A (early), 1 = B (two intervals behind)
    const int wq = wave & 3;                                        // wave within the half
    const int rl = rp & 1;                                          // row pair within the half's scratch
    const int tidl = tid & 255;                                     // thread within the half
    const int H = p.H, W = p.W;
    const int tiles_x = (W + B3_TW - 1) / B3_TW, tiles_y = (H + B3_TH - 1) / B3_TH;
    const int per_item = tiles_x * tiles_y;
    const int item_bytes = H * W * 128;
    const int wbytes = W * 128;
    // work order: chains of the gT frames of a clip at one spatial tile, dealt out XCD by XCD (conv_bf16.hip); p.flat (mode 1): single
    // tiles - every unit is then the first of its "chain" (f == 0) and fetches its addend pieces itself
    const int gT = ((FUSE && !p.flat) || WITH10) ? p.add_div : 1;
    const int nchains = per_item * (p.items / gT);
    const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3, cpx = gridDim.x >> 3;
    const int n_full = SPLIT ? p.n_full : nchains;
    const int per_xcd = (n_full + 7) >> 3;
    const int cbeg = xcd * per_xcd;
    const int ccnt = min(per_xcd, n_full - cbeg);
    if (!SPLIT && xj >= ccnt) return;
    const int nfc = (!SPLIT || xj < ccnt) ? (ccnt - xj + cpx - 1) / cpx : 0;   // whole chains of this workgroup
    [[maybe_unused]] const int slot = xcd * cpx + xj;
    const bool has_part = SPLIT && slot < (nchains - n_full) * p.split_s;
    const int sp_chain = has_part ? n_full + slot / p.split_s : 0;
    const int sp_f0 = has_part ? (slot % p.split_s) * p.split_q : 0, sp_f1 = has_part ? min(gT, sp_f0 + p.split_q) : 0;
    const int nu = nfc * gT + (sp_f1 - sp_f0);                      // tiles of this workgroup
    if (SPLIT && nu <= 0) return;
    // a unit = (item, y0, x0) + its position (chain ci of this workgroup, frame f of the chain; the chain's frames are [f0, fe)); all wave-uniform.
    // Stepping inside a chain is an increment; the three integer divisions (~450 cycles) happen once per chain, in an interval that has the slack.
    struct Unit { int item, y0, x0, f, ci, f0, fe; };
    auto unit_head = [&](int ci) __attribute__((always_inline)) {
        const bool part = SPLIT && ci >= nfc;                       // (wave-uniform) this workgroup's part of a cut chain: its last item
        const int ch = part ? sp_chain : cbeg + xj + ci * cpx;
        const int cl = ch / per_item;
        const int sp = ch - cl * per_item;
        const int ty = sp / tiles_x;
        const int f0 = part ? sp_f0 : 0;
        return Unit{cl * gT + f0, ty * B3_TH, (sp - ty * tiles_x) * B3_TW, f0, ci, f0, part ? sp_f1 : gT};
    };
    auto unit_next = [&](const Unit& c) __attribute__((always_inline)) {
        if (c.f + 1 < c.fe) return Unit{c.item + 1, c.y0, c.x0, c.f + 1, c.ci, c.f0, c.fe};
        return unit_head(c.ci + 1);
    };

    B3_STAMP();                                                 // (timing build) kernel entry, past the work-order arithmetic
    // DMA map: piece i covers halo pixels 8 i .. 8 i + 7 (linear, 34 per row); lane L -> pixel 8 i + (L >> 3), LDS slot L & 7, which
    // holds chunk (L & 7) ^ ((px >> 1) & 7) of that pixel: `drel` = byte offset of the lane's SOURCE chunk relative to the halo origin,
    // `dpk` = py | px << 8 (border test).  This wave's pieces: group B i = wq + 4k (< 25), group A i = 25 + wq + 4k (< 43).
    const int i0 = grp ? wq : B3_LOW + wq;
    const int iend = grp ? B3_LOW : B3_NDMA;
    int drel[B3_PIECES], dpk[B3_PIECES];
#pragma unroll
    for (int k = 0; k < B3_PIECES; ++k) {
        const int pix = 8 * (i0 + 4 * k) + (lane >> 3);
        const int py = pix / B3_IW, px = pix - py * B3_IW;
        drel[k] = py * wbytes + px * 128 + (((lane & 7) ^ ((px >> 1) & 7)) << 4);
        dpk[k] = py | (px << 8);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)b3_smem;
    B3_PSTAMP();                                                // (timing build) P1: DMA tables
    // this wave's share of the halo of unit `un` -> buffer `buf`; `live` false: nothing (an empty resource)
    auto dma_share = [&](const Unit& un, bool live, int buf) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in) + (size_t)un.item * H * W * 64, 0, live ? item_bytes : 0, 0x00020000);
        const int org = ((un.y0 - 1) * W + un.x0 - 1) * 128;
        const bool interior = un.y0 > 0 && un.y0 + B3_IH - 1 <= H && un.x0 > 0 && un.x0 + B3_IW - 1 <= W;
#pragma unroll
        for (int k = 0; k < B3_PIECES; ++k) {
            const int i = i0 + 4 * k;
            if (i < iend) {                                         // (wave-uniform)
                const int py = dpk[k] & 0xff, px = dpk[k] >> 8;
                const int gy = un.y0 + py - 1, gx = un.x0 + px - 1;
                const bool in = interior | (((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W) & (py < B3_IH));
                b3_dma16(rs, lds0 + buf * B3_TILE_BYTES + i * 1024, in ? org + drel[k] : 0x7fffffff);
            }
        }
    };
    // Completion of the pieces: the vector-memory counter retires in issue order, so "at most K operations outstanding", K = the operations
    // this wave is SURE to have issued after its last piece, means the pieces have landed.  After the pieces of a tile a wave always issues
    // its 4 line stores and (modes 1, 2) the 4 input loads of the next tile before it waits (the addend pieces and conv10_i's stores come
    // on top at chain boundaries: then the wait covers a few of them as well).  An explicit wait rather than a fence load whose result
    // is "consumed": the compiler's own counter model does not see the inline-asm pieces, and with a pending fence register it fell back
    // to vmcnt(0) in front of every piece (measured: 71 -> 90 us per launch).
    constexpr int KWAIT = MODE == 0 ? 4 : 8;
#define B3_PIECES_LANDED() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KWAIT) : "memory")

    // operand addresses (conv_bf16.hip): pixel operand of (column tap kx, k-step ks) = chunk 2*ks + (lane >> 5) of halo pixel
    // (row 2*rp + ..., column (lane & 31) + kx); weights: 16 bytes per lane
    int paddr[3][4];                                                // M16: [kx][2 kk + ph]
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (M16) {
                const int col = 16 * (ks & 1) + (lane & 15) + kx;
                paddr[kx][ks] = ((2 * rp) * B3_IW + col) * 128 + (((4 * (ks >> 1) + (lane >> 4)) ^ ((col >> 1) & 7)) << 4);
            } else {
                const int col = (lane & 31) + kx;
                paddr[kx][ks] = ((2 * rp) * B3_IW + col) * 128 + (((2 * ks + (lane >> 5)) ^ ((col >> 1) & 7)) << 4);
            }
        }
    }
    const unsigned char* const wlane = wl + (M16 ? 2 * mt : mt) * 1024 + lane * 16;
    [[maybe_unused]] const int ech16 = 32 * mt + 8 * (lane >> 4);   // M16: the lane's 8 consecutive channels (of pixels 16 ph + (lane & 15))
    // register r of a lane = channel 32mt + 16(lane>>5) + r (the row -> channel map of the packed weights), pixel lane & 31 of the row
    const int ech = 32 * mt + 16 * (lane >> 5);
    [[maybe_unused]] f32x16 acc[2];
    [[maybe_unused]] b3f4 acc4[2][4];                               // M16: [output row][2 ph + ct]
    [[maybe_unused]] b3u4 radd[2][2];                               // FUSE: addend pieces (accumulator layout; fetched once per chain)
    [[maybe_unused]] b3u4 rq[4];                                    // FUSE: this lane's residual pieces (row n, channel half h: index 2n + h)
    [[maybe_unused]] b3u4 xw[4];                                    // WITH10: W10 operands of the tile's frame
    [[maybe_unused]] f32x16 bacc[2];
    if constexpr (WITH10) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) bacc[n][r] = 0.f;
    }
    const float eslope = p.act ? 0.2f : 1.0f;
    const __amdgpu_buffer_rsrc_t rsXW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x_w), 0, WITH10 ? gT * 8192 : 0, 0x00020000);
    // the epilogue inputs of unit `un` (requested a serial phase ahead; `live` false: empty resources, zeros)
    auto request_inputs = [&](const Unit& un, bool live) __attribute__((always_inline)) {
        if constexpr (FUSE) {
            const int ox = un.x0 + (lane & 31);
            // (rows / columns past the image: past the resource's range)
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.resid) + (size_t)un.item * H * W * 64, 0, live ? item_bytes : 0, 0x00020000);
            // the shared-half pieces are the same pixels for every frame of a chain: fetched with its first frame
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.addend) + (size_t)(un.item / p.add_div) * H * W * 64, 0, live ? item_bytes : 0, 0x00020000);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int oy = un.y0 + 2 * rp + n;
                if constexpr (M16) {                                // pieces in the M16 accumulator layout: (row n, pixel half ph) -> channels ech16 .. + 7
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ox16 = un.x0 + 16 * h + (lane & 15);
                        const int eoff = (ox16 < W && oy < H) ? (oy * W + ox16) * 128 + ech16 * 2 : 0x7fffffff;
                        rq[2 * n + h] = __builtin_bit_cast(b3u4, __builtin_amdgcn_raw_buffer_load_b128(rsR, eoff, 0, 0));
                        if (un.f == un.f0) radd[n][h] = __builtin_bit_cast(b3u4, __builtin_amdgcn_raw_buffer_load_b128(rsA, eoff, 0, 0));
                    }
                } else {
                const int eoff = (ox < W && oy < H) ? (oy * W + ox) * 128 : 0x7fffffff;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    rq[2 * n + h] = __builtin_bit_cast(b3u4, __builtin_amdgcn_raw_buffer_load_b128(rsR, eoff, (ech + 8 * h) * 2, 0));
                    if (un.f == un.f0) radd[n][h] = __builtin_bit_cast(b3u4, __builtin_amdgcn_raw_buffer_load_b128(rsA, eoff, (ech + 8 * h) * 2, 0));
                }
                }
            }
        }
        if constexpr (WITH10) {                                     // W10[frame][k-step][channel tile mt]: 16 bytes per lane and k-step
#pragma unroll
            for (int g = 0; g < 4; ++g)
                xw[g] = __builtin_bit_cast(b3u4, __builtin_amdgcn_raw_buffer_load_b128(rsXW, lane * 16 + mt * 1024, (un.f * 4 + g) * 2048, 0));
        }
    };

    // ---- prologue: halo of tile 0 -> buffer 0 (both halves' shares = all of it); group B's share of tile 1 -> buffer 1 while it waits
    Unit cu = unit_head(0);
    Unit nx = nu > 1 ? unit_next(cu) : cu;
    B3_PSTAMP();                                                // P2: operand addresses, first units
    dma_share(cu, true, 0);
    request_inputs(cu, true);
    B3_PSTAMP();                                                // P3: halo + input requests issued
    // bias (and conv10_i's) by LDS-DMA, IN FRONT of the weight pieces (the wait below counts the late ones): one dword per lane
    if (wave == 0) b3_dma4(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, 256, 0x00020000), lds0 + 2 * B3_TILE_BYTES + B3_W_BYTES, lane * 4);
    if constexpr (WITH10) {
        if (wave == 1) b3_dma4(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x_bias), 0, 256, 0x00020000), lds0 + 2 * B3_TILE_BYTES + B3_W_BYTES + 256, lane * 4);
    }
    // weights -> LDS by LDS-DMA as well (the pack is the LDS image: piece w = 2 slice + mt, slice = (ky 3 + kx) 4 + ks; this wave's pieces
    // w = wave + 8 j have k-step wave >> 1, column tap j % 3, row tap j / 3), in TWO parts: a workgroup's prologue moves 72 KB of
    // weights + 44 KB of halo through a port that takes ~11 B per clock - 10 k cycles - and up to its mid barrier a tile only reads the
    // slices with 4 kx + ks < 7: groups 0-5, AND group 6, whose operands group 5 requests one group ahead (with "< 6" here the stress
    // found a rare wrong tile: group 5's prefetch read late slices before anything had waited for them).  Those go first and are waited
    // for here; the rest may still be on its way when tile 0 starts and is waited for in front of tile 0's mid barrier (every wave for its
    // own pieces, group B in front of the leading barrier that pairs with it: behind that barrier all of it is there).
    {
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wpack), 0, B3_W_BYTES, 0x00020000);
        const int ks0 = wave >> 1;
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const bool early = M16 ? (j % 3) < 2 : 4 * (j % 3) + ks0 < 7;   // (wave-uniform; M16: piece w = ((ky 3 + kx) 2 + kk) 4 + output tile: up to its mid barrier a tile reads - and prefetches - column taps 0 and 1)
                if (early == (part == 0)) b3_dma16(rsW, lds0 + 2 * B3_TILE_BYTES + (wave + 8 * j) * 1024, (wave + 8 * j) * 1024 + lane * 16);
            }
    }
    B3_PSTAMP();                                                // P4: weights requested
    if (M16 || wave < 6) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // all but the late weight pieces (waves 0-5: three of nine, waves 6-7: six; M16: three for every wave)
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    B3_PSTAMP();                                                // P5: halo and the early weights landed
    __syncthreads();
    [[maybe_unused]] b3f4 bias4[2];                                 // M16: output tile ct, rows 4 g + r = channels ech16 + 4 ct + r
    if constexpr (M16) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + ech16 + 4 * ct);
            bias4[ct] = b3f4{b4.x, b4.y, b4.z, b4.w};
        }
    }
    f32x16 bias16;                                                  // the tile's first MFMAs take C = bias (register r of a lane = channel ech + r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + ech + 4 * q);
        bias16[4 * q] = b4.x;
        bias16[4 * q + 1] = b4.y;
        bias16[4 * q + 2] = b4.z;
        bias16[4 * q + 3] = b4.w;
    }
    if (grp) {                                                      // group B: two intervals behind group A
        // (group B's first leading barrier is group A's mid barrier of tile 0, behind which A reads the LATE weight slices - group B's
        // share of them included: it has to have landed before group B arrives there)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dma_share(nx, nu > 1, 1);                                   // its share of tile 1's halo (nothing follows it that B3_PIECES_LANDED
        B3_BARRIER();                                               // could count on: an unconditional wait, between the two barriers)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        B3_BARRIER();
    }
#ifdef B3_PRIO_B
    if (grp) __builtin_amdgcn_s_setprio(1);
#endif

    for (int u = 0; u < nu; ++u) {
        const int cb = u & 1;
        B3_STAMP();                                                 // 0: tile start
        unsigned char* const tile = b3_smem + cb * B3_TILE_BYTES;   // this tile's halo
        unsigned char* const scr = tile + (grp ? B3_SCR_B : 0);     // then this half's 128 lines
        const int item = cu.item, y0 = cu.y0, x0 = cu.x0;

        // ---- I0 | I1: 12 groups (column tap kx, k-step ks): the 4 halo rows 2rp..2rp+3 serve the 3 row taps of both output rows - 4 pixel
        // reads + 3 weight reads feed 6 MFMAs; the operands of group g + 1 are requested in the MFMA gaps of group g
        if constexpr (M16) {
            // 6 groups (column tap kx, k-step kk of 32 channels): 4 halo rows x 2 pixel halves + 3 row taps x 2 output tiles = 14 operand reads feed 24 MFMAs
            // (row tap ky outermost: 8 different accumulators between two MFMAs on the same one); the operands of group g + 1 are requested one per gap
            // Two orders of the 24 MFMAs of a group.  Mode 1 (the per-frame half of conv2_i, which has the registers): ACCUMULATOR-major - the three row taps of one
            // accumulator back to back, operands double-buffered per group: consecutive 16x16x32 MFMAs that share their accumulator cost less energy
            // (tools/ubench/mfma_shape_power.hip: 2 220 against 2 000 TFLOP/s under the cap), the launch 72.3 -> 69.8 us at 1080p.  The other modes (conv10_i's
            // accumulators and operands on top: the double buffer spills): row-tap-major with operands refilled in place.
            if constexpr (FUSE) {
            b3h8 px[2][4][2], wv[2][3][2];
#define B3_PX16(g_, r_, ph_) (*reinterpret_cast<const b3h8*>(tile + paddr[(g_) >> 1][2 * ((g_) & 1) + (ph_)] + (r_) * (B3_IW * 128)))
#define B3_WT16(g_, ky_, ct_) (*reinterpret_cast<const b3h8*>(wlane + (((((ky_) * 3 + ((g_) >> 1)) * 2 + ((g_) & 1)) * 4 + (ct_)) << 10)))
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) px[0][r][ph] = B3_PX16(0, r, ph);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) wv[0][ky][ct] = B3_WT16(0, ky, ct);
            auto group16 = [&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                constexpr int cur = g & 1;
#define B3_M(ky_, n_, ph_, ct_)                                                                                                        \
    do {                                                                                                                               \
        if constexpr (g == 0 && (ky_) == 0) acc4[n_][2 * (ph_) + (ct_)] = b3_mfma16(wv[cur][ky_][ct_], px[cur][(n_) + (ky_)][ph_], bias4[ct_]);   \
        else acc4[n_][2 * (ph_) + (ct_)] = b3_mfma16(wv[cur][ky_][ct_], px[cur][(n_) + (ky_)][ph_], acc4[n_][2 * (ph_) + (ct_)]);      \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
    } while (0)
#define B3_A(n_, ph_, ct_) do { B3_M(0, n_, ph_, ct_); B3_M(1, n_, ph_, ct_); B3_M(2, n_, ph_, ct_); } while (0)
#define B3_RP(r_, ph_) do { if constexpr (g < 5) px[cur ^ 1][r_][ph_] = B3_PX16(g + 1, r_, ph_); __builtin_amdgcn_sched_barrier(0); } while (0)
#define B3_RW(k_, ct_) do { if constexpr (g < 5) wv[cur ^ 1][k_][ct_] = B3_WT16(g + 1, k_, ct_); __builtin_amdgcn_sched_barrier(0); } while (0)
                __builtin_amdgcn_sched_barrier(0);
                B3_A(0, 0, 0); B3_RP(0, 0); B3_RW(0, 0);
                B3_A(0, 0, 1); B3_RP(1, 0); B3_RW(0, 1);
                B3_A(0, 1, 0); B3_RP(0, 1); B3_RW(1, 0);
                B3_A(0, 1, 1); B3_RP(1, 1); B3_RW(1, 1);
                B3_A(1, 0, 0); B3_RP(2, 0); B3_RW(2, 0);
                B3_A(1, 0, 1); B3_RP(2, 1); B3_RW(2, 1);
                B3_A(1, 1, 0); B3_RP(3, 0);
                B3_A(1, 1, 1); B3_RP(3, 1);
#undef B3_A
#undef B3_M
#undef B3_RP
#undef B3_RW
            };
            group16(std::integral_constant<int, 0>{});
            group16(std::integral_constant<int, 1>{});
            group16(std::integral_constant<int, 2>{});
            B3_STAMP();                                             // 1: groups 0-2 issued
            if (u == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the late weight pieces of the prologue (column tap 2)
            B3_BARRIER();                                           // interval boundary (the other half's dump | lines)
            B3_STAMP();                                             // 2
            group16(std::integral_constant<int, 3>{});
            group16(std::integral_constant<int, 4>{});
            group16(std::integral_constant<int, 5>{});
#undef B3_PX16
#undef B3_WT16
            } else {
            // Operands are refilled IN PLACE for group g + 1 as soon as group g has read them for the last time (halo row 0 behind its 4th MFMA, row tap 0's
            // weights behind the 8th, row 1 behind the 12th ...): 8 + 6 operand sets live, not 16 + 12 - the conv1_i + conv10_i mode has conv10_i's
            // accumulators and operands on top and spilled with a full double buffer.
            b3h8 px[4][2], wv[3][2];
#define B3_PX16(g_, r_, ph_) (*reinterpret_cast<const b3h8*>(tile + paddr[(g_) >> 1][2 * ((g_) & 1) + (ph_)] + (r_) * (B3_IW * 128)))
#define B3_WT16(g_, ky_, ct_) (*reinterpret_cast<const b3h8*>(wlane + (((((ky_) * 3 + ((g_) >> 1)) * 2 + ((g_) & 1)) * 4 + (ct_)) << 10)))
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) px[r][ph] = B3_PX16(0, r, ph);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) wv[ky][ct] = B3_WT16(0, ky, ct);
            auto group16 = [&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
#define B3_M(ky_, n_, ph_, ct_)                                                                                                        \
    do {                                                                                                                               \
        if constexpr (g == 0 && (ky_) == 0) acc4[n_][2 * (ph_) + (ct_)] = b3_mfma16(wv[ky_][ct_], px[(n_) + (ky_)][ph_], bias4[ct_]);   \
        else acc4[n_][2 * (ph_) + (ct_)] = b3_mfma16(wv[ky_][ct_], px[(n_) + (ky_)][ph_], acc4[n_][2 * (ph_) + (ct_)]);                \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
    } while (0)
#define B3_RP(r_, ph_) do { if constexpr (g < 5) px[r_][ph_] = B3_PX16(g + 1, r_, ph_); __builtin_amdgcn_sched_barrier(0); } while (0)
#define B3_RW(k_, ct_) do { if constexpr (g < 5) wv[k_][ct_] = B3_WT16(g + 1, k_, ct_); __builtin_amdgcn_sched_barrier(0); } while (0)
                __builtin_amdgcn_sched_barrier(0);
                B3_M(0, 0, 0, 0);
                B3_M(0, 0, 0, 1);
                B3_M(0, 0, 1, 0);
                B3_M(0, 0, 1, 1); B3_RP(0, 0);
                B3_M(0, 1, 0, 0); B3_RP(0, 1);
                B3_M(0, 1, 0, 1);
                B3_M(0, 1, 1, 0);
                B3_M(0, 1, 1, 1); B3_RW(0, 0);
                B3_M(1, 0, 0, 0); B3_RW(0, 1);
                B3_M(1, 0, 0, 1);
                B3_M(1, 0, 1, 0);
                B3_M(1, 0, 1, 1); B3_RP(1, 0);
                B3_M(1, 1, 0, 0); B3_RP(1, 1);
                B3_M(1, 1, 0, 1);
                B3_M(1, 1, 1, 0);
                B3_M(1, 1, 1, 1); B3_RW(1, 0);
                B3_M(2, 0, 0, 0); B3_RW(1, 1);
                B3_M(2, 0, 0, 1);
                B3_M(2, 0, 1, 0);
                B3_M(2, 0, 1, 1); B3_RP(2, 0);
                B3_M(2, 1, 0, 0); B3_RP(2, 1);
                B3_M(2, 1, 0, 1);
                B3_M(2, 1, 1, 0);
                B3_M(2, 1, 1, 1); B3_RP(3, 0); B3_RP(3, 1); B3_RW(2, 0); B3_RW(2, 1);
#undef B3_M
#undef B3_RP
#undef B3_RW
            };
            group16(std::integral_constant<int, 0>{});
            group16(std::integral_constant<int, 1>{});
            group16(std::integral_constant<int, 2>{});
            B3_STAMP();                                             // 1: groups 0-2 issued
            if (u == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the late weight pieces of the prologue (column tap 2)
            B3_BARRIER();                                           // interval boundary (the other half's dump | lines)
            B3_STAMP();                                             // 2
            group16(std::integral_constant<int, 3>{});
            group16(std::integral_constant<int, 4>{});
            group16(std::integral_constant<int, 5>{});
#undef B3_PX16
#undef B3_WT16
            }
        } else {
        b3h8 px[2][4], wv[2][3];
#define B3_PX(g_, r_) (*reinterpret_cast<const b3h8*>(tile + paddr[(g_) >> 2][(g_) & 3] + (r_) * (B3_IW * 128)))
#define B3_WT(g_, ky_) (*reinterpret_cast<const b3h8*>(wlane + ((((ky_) * 3 + ((g_) >> 2)) * 4 + ((g_) & 3)) << 11)))
#pragma unroll
        for (int r = 0; r < 4; ++r) px[0][r] = B3_PX(0, r);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) wv[0][ky] = B3_WT(0, ky);
        auto group = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            constexpr int cur = g & 1;
#define B3_M(ky_, n_)                                                                                          \
    do {                                                                                                       \
        if constexpr (g == 0 && (ky_) == 0) acc[n_] = b3_mfma(wv[cur][ky_], px[cur][(n_) + (ky_)], bias16);     \
        else acc[n_] = b3_mfma(wv[cur][ky_], px[cur][(n_) + (ky_)], acc[n_]);                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)
#define B3_RP(r_) do { if constexpr (g < 11) px[cur ^ 1][r_] = B3_PX(g + 1, r_); } while (0)
#define B3_RW(k_) do { if constexpr (g < 11) wv[cur ^ 1][k_] = B3_WT(g + 1, k_); } while (0)
            __builtin_amdgcn_sched_barrier(0);
            B3_M(0, 0);
            B3_RP(0);
            B3_RW(0);
            __builtin_amdgcn_sched_barrier(0);
            B3_M(0, 1);
            B3_RP(1);
            __builtin_amdgcn_sched_barrier(0);
            B3_M(1, 0);
            B3_RP(2);
            B3_RW(1);
            __builtin_amdgcn_sched_barrier(0);
            B3_M(1, 1);
            B3_RP(3);
            __builtin_amdgcn_sched_barrier(0);
            B3_M(2, 0);
            B3_RW(2);
            __builtin_amdgcn_sched_barrier(0);
            B3_M(2, 1);
#undef B3_M
#undef B3_RP
#undef B3_RW
        };
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 1>{});
        group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{});
        group(std::integral_constant<int, 4>{});
        group(std::integral_constant<int, 5>{});
        B3_STAMP();                                                 // 1: groups 0-5 issued
        if (u == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the late weight pieces of the prologue (groups 6-11 read them)
        B3_BARRIER();                                               // interval boundary (the other half's dump | lines)
        B3_STAMP();                                                 // 2
        group(std::integral_constant<int, 6>{});
        group(std::integral_constant<int, 7>{});
        group(std::integral_constant<int, 8>{});
        group(std::integral_constant<int, 9>{});
        group(std::integral_constant<int, 10>{});
        group(std::integral_constant<int, 11>{});
#undef B3_PX
#undef B3_WT
        }
        B3_STAMP();                                                 // 3: groups 6-11 issued
        if (grp) B3_PIECES_LANDED();                                // group B's share of the next halo has landed (group A reads it after this barrier)
        B3_BARRIER();                                               // this half is past its last operand read of its rows: they are its scratch now
        B3_STAMP();                                                 // 4

        // ---- I2: the halo requests this half owes, then the epilogue dump
        const bool more = u + 1 < nu, more2 = u + 2 < nu;           // (wave-uniform)
        const Unit n2 = more2 ? unit_next(nx) : nx;
        auto halo_requests = [&]() __attribute__((always_inline)) {
            if (!grp) dma_share(nx, more, cb ^ 1);                  // group A: pieces i >= 25 of tile u + 1 (group B's lines of tile u - 1 have left)
            else dma_share(n2, more2, cb);                          // group B: pieces i < 25 of tile u + 2, below its own scratch
        };
        // (mode 1: the dump reads registers that compiler-issued loads filled, and the compiler's wait for them, which does not know the
        // pieces, covers the pieces as well; issuing them behind the dump instead measured worse - 76.9 -> 81.9 us per launch: that
        // mode runs at what the CU's memory port sustains, and the pieces cannot afford to start late)
        halo_requests();
        // scratch = pixel-major lines, 16-byte pieces XOR-swizzled by (pixel >> 1) & 7 (conflict-free for the piece writes, the line
        // read-back and the 1x1 operand reads)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int h = 0; h < 2; ++h) {                           // row n, channels ech + 8h .. + 7: bias is in (initial C); addend, leaky_relu, residual, bf16
                                                                    // (M16: h = pixel half; the lane's channels ech16 .. + 7 = output tiles ct = 0, 1)
                const int j = M16 ? 16 * h + (lane & 15) : (lane & 31);
                const int c = M16 ? 4 * mt + (lane >> 4) : 4 * mt + 2 * (lane >> 5) + h;   // piece of the pixel's line
                b3u4* const slot = reinterpret_cast<b3u4*>(scr + ((2 * rl + n) * 32 + j) * 128 + ((c ^ ((j >> 1) & 7)) << 4));
                f32x4 v[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r0 = 8 * h + 4 * q;
                    if constexpr (M16) v[q] = f32x4{acc4[n][2 * h + q][0], acc4[n][2 * h + q][1], acc4[n][2 * h + q][2], acc4[n][2 * h + q][3]};
                    else v[q] = f32x4{acc[n][r0], acc[n][r0 + 1], acc[n][r0 + 2], acc[n][r0 + 3]};
                    if constexpr (FUSE) v[q] += b3_to_f32(b3u2{radd[n][h][2 * q], radd[n][h][2 * q + 1]});
                    v[q] = b3_lrelu4(v[q], eslope);
                    if constexpr (FUSE) v[q] += b3_to_f32(b3u2{rq[2 * n + h][2 * q], rq[2 * n + h][2 * q + 1]});
                }
                const b3u2 lo = b3_to_bf16(v[0]), hi = b3_to_bf16(v[1]);
                *slot = b3u4{lo.x, lo.y, hi.x, hi.y};
            }
        B3_STAMP();                                                 // 5: epilogue pieces written
        B3_BARRIER();                                               // this half's 128 lines are complete
        B3_STAMP();                                                 // 6

        // ---- I3: whole 128-byte lines to HBM (8 pixels per wave instruction), conv10_i on the same lines, the next tile's epilogue inputs
        {
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)item * H * W * 64, 0, item_bytes, 0x00020000);
            b3u4 line[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {                           // 1024 pieces per half, 4 per thread
                const int id = k * B3_GTHREADS + tidl;
                const int pp = id >> 3, c = id & 7;
                line[k] = *reinterpret_cast<const b3u4*>(scr + pp * 128 + ((c ^ ((pp >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int id = k * B3_GTHREADS + tidl;
                const int pp = id >> 3, c = id & 7;
                const int sx = x0 + (pp & 31), sy = y0 + 4 * grp + (pp >> 5);
                buffer_store_b128_guarded<B3_STORE_AUX>(line[k], rsO, (sx < W && sy < H) ? (sy * W + sx) * 128 + c * 16 : 0x7fffffff, 0);
            }
        }
        if constexpr (WITH10) {                                     // conv10_i: 4 k-steps x 2 rows, B = the lines of this frame's tile, A = W10 of the frame
            const int j = lane & 31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned char* bp = scr + ((2 * rl) * 32 + j) * 128 + (((2 * ks + (lane >> 5)) ^ ((j >> 1) & 7)) << 4);
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    bacc[n] = b3_mfma(__builtin_bit_cast(b3h8, xw[ks]), *reinterpret_cast<const b3h8*>(bp + n * 32 * 128), bacc[n]);
            }
            if (SPLIT && cu.ci >= nfc && cu.f == cu.fe - 1) {       // (wave-uniform) a PART is complete: its raw fp32 sum -> partial[slot] ([row][column][64]); clear
                float* const pt = p.partial + (size_t)slot * (B3_TH * B3_TW * 64) + (size_t)(2 * rp * B3_TW + (lane & 31)) * 64 + ech;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(pt + n * (B3_TW * 64) + 4 * q) = f32x4{bacc[n][4 * q], bacc[n][4 * q + 1], bacc[n][4 * q + 2], bacc[n][4 * q + 3]};
#pragma unroll
                    for (int r = 0; r < 16; ++r) bacc[n][r] = 0.f;
                }
            } else if (cu.f == cu.fe - 1) {                         // (wave-uniform) the chain is complete: bias, leaky_relu, bf16, store; clear
                const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(p.x_out + (size_t)(item / gT) * H * W * 64, 0, item_bytes, 0x00020000);
                const int sx = x0 + (lane & 31);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int sy = y0 + 2 * rp + n;
                    const int off = (sx < W && sy < H) ? ((sy * W + sx) * 64 + ech) * 2 : 0x7fffffff;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x4 v[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int r0 = 8 * h + 4 * q;
                            v[q] = f32x4{bacc[n][r0], bacc[n][r0 + 1], bacc[n][r0 + 2], bacc[n][r0 + 3]} + *reinterpret_cast<const f32x4*>(bl + 64 + ech + r0);
                            v[q] = b3_lrelu4(v[q], 0.2f);
                        }
                        const b3u2 lo = b3_to_bf16(v[0]), hi = b3_to_bf16(v[1]);
                        buffer_store_b128_guarded<B3_STORE_AUX>(b3u4{lo.x, lo.y, hi.x, hi.y}, rsX, off, 16 * h);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) bacc[n][r] = 0.f;
                }
            }
        }
        request_inputs(nx, more);                                   // (after the 1x1's MFMAs have read xw)
        cu = nx;
        nx = n2;
        B3_STAMP();                                                 // 7: lines out
        if (!grp) B3_PIECES_LANDED();                               // group A's share of the next halo has landed
        B3_BARRIER();                                               // this half's scratch has been read: its bytes are free for the halo after next
    }
    if (!grp) {                                                     // group A sits out group B's last two intervals
        B3_BARRIER();
        B3_BARRIER();
    }
}

template <int MODE, bool SPLIT = false, bool M16 = false>
static hipError_t b3_launch(const ConvBf16Params& p, int grid, int dev, hipStream_t s) {
    static std::atomic<int> attr_dev[64];
    if (!attr_dev[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16_v3_kernel<MODE, SPLIT, M16>), hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_dev[dev] = 1;
    }
    hipLaunchKernelGGL((conv3x3_bf16_v3_kernel<MODE, SPLIT, M16>), dim3(grid), dim3(B3_THREADS), B3_LDS_BYTES, s, p);
    return hipGetLastError();
}

// split chains: the parts' raw conv10_i sums -> x_out (bf16).  8 workgroups per cut chain, thread = (pixel, 4-channel group), the parts added in the fixed
// order r = 0 .. s-1, then + x_bias, leaky-relu, bf16 (the rounding point of the uncut launch)
__global__ __launch_bounds__(256) void c10_finalize_bf16_kernel(ConvBf16Params p) {
    const int H = p.H, W = p.W;
    const int tiles_x = (W + B3_TW - 1) / B3_TW, tiles_y = (H + B3_TH - 1) / B3_TH;
    const int per_item = tiles_x * tiles_y;
    const int j = blockIdx.x >> 3, ch = p.n_full + j;
    const int clip = ch / per_item, sp = ch - clip * per_item;
    const int ty = sp / tiles_x, y0 = ty * B3_TH, x0 = (sp - ty * tiles_x) * B3_TW;
    const int S = p.split_s;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int id = (blockIdx.x & 7) * 512 + it * 256 + threadIdx.x;
        const int pix = id >> 4, c4 = id & 15;
        const int y = y0 + pix / B3_TW, x = x0 + (pix & (B3_TW - 1));
        const float* src = p.partial + ((size_t)j * S * (B3_TH * B3_TW) + pix) * 64 + c4 * 4;
        f32x4 part[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) part[r] = r < S ? *reinterpret_cast<const f32x4*>(src + (size_t)r * (B3_TH * B3_TW * 64)) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 v = part[0];
#pragma unroll
        for (int r = 1; r < 7; ++r)
            if (r < S) v += part[r];
        v += *reinterpret_cast<const f32x4*>(p.x_bias + c4 * 4);
        v = b3_lrelu4(v, 0.2f);
        if (y < H && x < W) *reinterpret_cast<b3u2*>(p.x_out + (((size_t)clip * H + y) * W + x) * 64 + c4 * 4) = b3_to_bf16(v);
    }
}

static bool b3_split_ok(const ConvBf16Params& p, int grid) {
    const int T = p.add_div;
    const long long nchains = (long long)((p.W + B3_TW - 1) / B3_TW) * ((p.H + B3_TH - 1) / B3_TH) * (p.items / T);
    if (p.split_s < 2 || p.split_s > 7 || p.split_q < 1 || p.n_full < 0 || p.n_full % grid || p.n_full >= nchains) return false;
    if ((long long)p.split_s * p.split_q < T || (long long)(p.split_s - 1) * p.split_q >= T) return false;
    return (nchains - p.n_full) * p.split_s <= grid;
}

hipError_t launch_c10_finalize_bf16(const ConvBf16Params& p, hipStream_t s) {
    if (!p.partial || !p.x_out || !p.x_bias || p.items < 1 || p.add_div < 1 || p.items % p.add_div) return hipErrorInvalidValue;
    const int ncu = device_cu_count();
    if (!ncu) return hipErrorUnknown;
    const int grid = ncu >= 8 ? ncu / 8 * 8 : 8;
    if (!b3_split_ok(p, grid)) return hipErrorInvalidValue;
    const long long nchains = (long long)((p.W + B3_TW - 1) / B3_TW) * ((p.H + B3_TH - 1) / B3_TH) * (p.items / p.add_div);
    hipLaunchKernelGGL(c10_finalize_bf16_kernel, dim3((unsigned)(nchains - p.n_full) * 8), dim3(256), 0, s, p);
    return hipGetLastError();
}

// modes 0 - 2 of launch_conv3x3_bf16 (conv_bf16.hip) on the third-generation kernel
hipError_t launch_conv3x3_bf16_v3(const ConvBf16Params& p, int mode, hipStream_t s) {
    if (mode < 0 || mode > 2) return hipErrorInvalidValue;
    const int ncu = device_cu_count();
    if (!ncu) return hipErrorUnknown;
    const int grid = ncu >= 8 ? ncu / 8 * 8 : 8;                    // whole XCDs; surplus workgroups exit at once
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    // p.wpack16 (conv3x3_bf16_pack_weights16): the M16 form of the kernel on that pack
    ConvBf16Params q = p;
    const bool m16 = p.wpack16 != nullptr;
    if (m16) q.wpack = p.wpack16;
    if (p.split_s) {                                                // split chains: the chained modes only, geometry as the kernels assume it
        if (mode == 0 || p.flat || !b3_split_ok(p, grid) || (mode == 2 && !p.partial)) return hipErrorInvalidValue;
        if (m16) return mode == 1 ? b3_launch<1, true, true>(q, grid, dev, s) : b3_launch<2, true, true>(q, grid, dev, s);
        return mode == 1 ? b3_launch<1, true>(p, grid, dev, s) : b3_launch<2, true>(p, grid, dev, s);
    }
    if (m16) return mode == 1 ? b3_launch<1, false, true>(q, grid, dev, s) : mode == 2 ? b3_launch<2, false, true>(q, grid, dev, s) : b3_launch<0, false, true>(q, grid, dev, s);
    if (mode == 1) return b3_launch<1>(p, grid, dev, s);
    if (mode == 2) return b3_launch<2>(p, grid, dev, s);
    return b3_launch<0>(p, grid, dev, s);
}

}  // namespace pfnl

#ifdef PFNL_B3_TIMING
extern "C" int pfnl_debug_read_b3_stamps(long long* host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pfnl::b3_dbg), n * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

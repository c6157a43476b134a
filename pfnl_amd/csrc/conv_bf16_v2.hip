// bf16 trunk, second generation of the 3x3 64->64 kernel (round 4; modes 0 - 2 of conv3x3_bf16_kernel in conv_bf16.hip, which stays for
// the accumulating mode 3 and as `PFNL_BF16_V2=0`): conv1_i + conv10_i, both halves of conv2_i (reference model/pfnl.py:49-51, 66-71).
//
// Same arithmetic, same LDS layouts, same packed weights, same work order and the same epilogue pieces as the first kernel - what changes
// is WHEN the non-MFMA work of a tile runs.  The first kernel slices the previous tile's epilogue, its stores, the next halo's commit
// and the request after next over the 12 MFMA groups of a tile "to hide them"; both waves of a SIMD run the same slice at the same time,
// so nothing hides anything (its header: 9.2 k cycles per tile with MFMAs, loads and stores compiled out, 4.6 k of MFMA on top; measured in
// round 4 against what the byte streams alone need - tools/ubench/cu_stream_mix - conv1_i + conv10_i takes 12.4 k cycles per tile for 7 k
// of bytes and 5.1 k of MFMA).  Here, as in conv3x3_c1c10_kernel of the fp32 path:
//   * the halo is already in operand form in HBM (bf16 NHWC = the LDS layout up to the XOR swizzle), so it travels by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`, 43 wave instructions per tile, swizzle applied to the source, out-of-image = out-of-range offset =
//     zeros), issued at the START of a tile for the NEXT tile into the other buffer, waited for with a fence load at the tile's end: no
//     staging registers (24), no commit (6 ds_write_b128 per thread), no packed-word decode;
//   * the 12 groups of a tile are MFMAs and their operand reads and nothing else;
//   * then a short SERIAL phase in the tile's own halo buffer (free now): accumulators -> bias is the initial C, addend, leaky-relu,
//     residual (both fetched in accumulator layout during the first groups), bf16 -> pixel-major lines in LDS -> barrier -> whole 128-byte lines to HBM, and (mode 2) the 1x1 of
//     conv10_i reading the same lines as its B operands.  No copy of the accumulators (32 registers), no `pending` state, no tail.
// Barriers per tile: 3.
#include <type_traits>

#include "common.h"
#include "conv_bf16.h"

#ifndef B2_STORE_AUX
#define B2_STORE_AUX 0
#endif

namespace pfnl {

typedef __bf16 b2h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b2h4 __attribute__((ext_vector_type(4)));
typedef unsigned b2u4 __attribute__((ext_vector_type(4)));
typedef unsigned b2u2 __attribute__((ext_vector_type(2)));

constexpr int B2_THREADS = 512;
constexpr int B2_TH = 8, B2_TW = 32, B2_IH = 10, B2_IW = 34;
constexpr int B2_NDMA = (B2_IH * B2_IW + 7) / 8;                    // 43 DMA instructions of 8 pixels x 128 B
constexpr int B2_TILE_BYTES = B2_NDMA * 1024;                       // 44 032 (the last instruction's 4 surplus pixels land in padding)
constexpr int B2_W_BYTES = 9 * 4 * 2 * 1024;                        // 73 728: conv3x3_bf16_pack_weights
constexpr int B2_LDS_BYTES = 2 * B2_TILE_BYTES + B2_W_BYTES + 64 * 4;   // 162 048 of 163 840
constexpr int B2_DMA_ITERS = (B2_NDMA + 7) / 8;                     // 6 per wave (waves 3..7: 5)
static_assert(B2_LDS_BYTES <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void b2_dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}
__device__ __forceinline__ f32x16 b2_mfma(b2h8 a, b2h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 b2_to_f32(b2u2 v) {
    return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                 __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 b2_lrelu4(f32x4 v, float slope) {
    const f32x4 sv = v * slope;
    asm("v_max_f32 %0, %1, %2" : "=v"(v.x) : "v"(v.x), "v"(sv.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(v.y) : "v"(v.y), "v"(sv.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(v.z) : "v"(v.z), "v"(sv.z));
    asm("v_max_f32 %0, %1, %2" : "=v"(v.w) : "v"(v.w), "v"(sv.w));
    return v;
}
__device__ __forceinline__ b2u2 b2_to_bf16(f32x4 v) {             // round to nearest even (v_cvt_pk_bf16_f32)
    const b2h4 b = __builtin_convertvector(v, b2h4);
    return __builtin_bit_cast(b2u2, b);
}
#define B2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef PFNL_B2_TIMING   /* phase timeline (tools/b2_timing.py); not part of the product build */
__device__ long long b2_dbg[256 * 2 * 128];
#ifndef PFNL_B2_TIMING_MODE
#define PFNL_B2_TIMING_MODE 1
#endif
#define B2_STAMP() do { if (MODE == PFNL_B2_TIMING_MODE && lane == 0 && (wave == 0 || wave == 5) && dbg_n < 128) b2_dbg[(blockIdx.x * 2 + (wave != 0)) * 128 + dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define B2_STAMP() do {} while (0)
#endif

// MODE 0: out = act(conv + bias).  MODE 1 (conv2_i per-frame half): out = act(conv + bias + addend[item / add_div]) + resid.
// MODE 2 (conv1_i + conv10_i): MODE 0, and per chain of add_div frames x_out = lrelu(sum_t W10_t out_t + x_bias).
template <int MODE>
__global__ __launch_bounds__(B2_THREADS, 1) void conv3x3_bf16_v2_kernel(ConvBf16Params p) {
    constexpr bool FUSE = MODE == 1;
    constexpr bool WITH10 = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char b2_smem[];
    unsigned char* const wl = b2_smem + 2 * B2_TILE_BYTES;
    float* const bl = reinterpret_cast<float*>(b2_smem + 2 * B2_TILE_BYTES + B2_W_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PFNL_B2_TIMING
    int dbg_n = 0;
#endif
    const int rp = wave >> 1;                                       // rows 2rp, 2rp+1 of the tile
    const int mt = wave & 1;                                        // output channels 32mt .. 32mt+31
    const int H = p.H, W = p.W;
    const int tiles_x = (W + B2_TW - 1) / B2_TW, tiles_y = (H + B2_TH - 1) / B2_TH;
    const int per_item = tiles_x * tiles_y;
    const int item_bytes = H * W * 128;
    const int wbytes = W * 128;
    // work order: chains of the gT frames of a clip at one spatial tile, dealt out XCD by XCD (conv_bf16.hip)
    const int gT = (FUSE || WITH10) ? p.add_div : 1;
    const int nchains = per_item * (p.items / gT);
    const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3, cpx = gridDim.x >> 3;
    const int per_xcd = (nchains + 7) >> 3;
    const int cbeg = xcd * per_xcd;
    const int ccnt = min(per_xcd, nchains - cbeg);
    if (xj >= ccnt) return;
    const int nu = ((ccnt - xj + cpx - 1) / cpx) * gT;              // tiles of this workgroup
#define B2_UNIT(u_, item_, y0_, x0_)                                                             \
    do {                                                                                         \
        const int ci_ = (u_) / gT, f_ = (u_) - ci_ * gT;                                         \
        const int ch_ = cbeg + xj + ci_ * cpx;                                                   \
        const int cl_ = ch_ / per_item;                                                          \
        const int sp_ = ch_ - cl_ * per_item;                                                    \
        item_ = cl_ * gT + f_;                                                                   \
        const int ty_ = sp_ / tiles_x;                                                           \
        y0_ = ty_ * B2_TH;                                                                       \
        x0_ = (sp_ - ty_ * tiles_x) * B2_TW;                                                     \
    } while (0)

    B2_STAMP();                                                 // (timing build) kernel entry, past the work-order arithmetic
    // weights + bias -> LDS (once per workgroup)
#pragma unroll
    for (int k = 0; k < B2_W_BYTES / 16 / B2_THREADS; ++k)
        reinterpret_cast<b2u4*>(wl)[k * B2_THREADS + tid] = reinterpret_cast<const b2u4*>(p.wpack)[k * B2_THREADS + tid];
    if (tid < 64) bl[tid] = p.bias[tid];

    // DMA map: instruction i = wave + 8 k covers halo pixels 8 i .. 8 i + 7 (linear, 34 per row); lane L -> pixel 8 i + (L >> 3), LDS slot
    // L & 7, which holds chunk (L & 7) ^ ((px >> 1) & 7) of that pixel: `drel` = byte offset of the lane's SOURCE chunk relative to the halo
    // origin, `dpk` = py | px << 8 (border test).  Constant for the life of the kernel.
    int drel[B2_DMA_ITERS], dpk[B2_DMA_ITERS];
#pragma unroll
    for (int k = 0; k < B2_DMA_ITERS; ++k) {
        const int pix = 8 * (wave + 8 * k) + (lane >> 3);
        const int py = pix / B2_IW, px = pix - py * B2_IW;
        drel[k] = py * wbytes + px * 128 + (((lane & 7) ^ ((px >> 1) & 7)) << 4);
        dpk[k] = py | (px << 8);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)b2_smem;
#define B2_DMA_PIECE(k_, rs_, org_, interior_, y0_, x0_, buf_)                                   \
    do {                                                                                         \
        const int i_ = wave + 8 * (k_);                                                          \
        if ((k_) < B2_DMA_ITERS - 1 || i_ < B2_NDMA) {                                           \
            const int py_ = dpk[k_] & 0xff, px_ = dpk[k_] >> 8;                                  \
            const int gy_ = (y0_) + py_ - 1, gx_ = (x0_) + px_ - 1;                              \
            const bool in_ = (interior_) | (((unsigned)gy_ < (unsigned)H) & ((unsigned)gx_ < (unsigned)W) & (py_ < B2_IH)); \
            b2_dma16(rs_, lds0 + (buf_) * B2_TILE_BYTES + i_ * 1024, in_ ? (org_) + drel[k_] : 0x7fffffff); \
        }                                                                                        \
    } while (0)
#define B2_DMA_HALO(rs_, org_, interior_, y0_, x0_, buf_)                                        \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < B2_DMA_ITERS; ++k_) B2_DMA_PIECE(k_, rs_, org_, interior_, y0_, x0_, buf_); \
    } while (0)

    // operand addresses (conv_bf16.hip): pixel operand of (column tap kx, k-step ks) = chunk 2*ks + (lane >> 5) of halo pixel
    // (row 2*rp + ..., column (lane & 31) + kx); weights: 16 bytes per lane
    int paddr[3][4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = (lane & 31) + kx;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            paddr[kx][ks] = ((2 * rp) * B2_IW + col) * 128 + (((2 * ks + (lane >> 5)) ^ ((col >> 1) & 7)) << 4);
    }
    const unsigned char* const wlane = wl + mt * 1024 + lane * 16;
    // register r of a lane = channel 32mt + 16(lane>>5) + r (the row -> channel map of the packed weights), pixel lane & 31 of the row
    const int ech = 32 * mt + 16 * (lane >> 5);
    f32x16 acc[2];
    [[maybe_unused]] b2u4 radd[2][2];                               // FUSE: addend pieces (accumulator layout; fetched once per chain)
    [[maybe_unused]] b2u4 rq[4];                                    // FUSE: this lane's residual pieces (row n, channel half h: index 2n + h)
    [[maybe_unused]] b2u4 xw[4];                                    // WITH10: W10 operands of the tile's frame
    [[maybe_unused]] f32x16 bacc[2];
    if constexpr (WITH10) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) bacc[n][r] = 0.f;
    }
    // the finished tile's lines leave LDS into these registers at the end of its serial phase and go to HBM one piece per group under
    // the NEXT tile's first MFMAs: 32 KB of stores issued in one go block the CU's memory pipe for ~2 k cycles (measured: the requests
    // behind them - the next halo's DMA pieces - stalled every wave for 1.5 - 5 k cycles per tile)
    b2u4 held[4];
    int h_item = 0, h_y0 = 0, h_x0 = 0;
    bool h_pending = false;
    const float eslope = p.act ? 0.2f : 1.0f;
    const __amdgpu_buffer_rsrc_t rsXW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x_w), 0, WITH10 ? gT * 8192 : 0, 0x00020000);

    // ---- prologue: halo of tile 0 -> buffer 0
    int c_item, c_y0, c_x0;
    B2_UNIT(0, c_item, c_y0, c_x0);
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in) + (size_t)c_item * H * W * 64, 0, item_bytes, 0x00020000);
        const int org = ((c_y0 - 1) * W + c_x0 - 1) * 128;
        B2_DMA_HALO(rs, org, false, c_y0, c_x0, 0);
        const unsigned fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
        asm volatile("" ::"v"(fence));                              // the compiler's wait for the fence load: the DMAs have landed
    }
    __syncthreads();

#ifdef B2_PRIO   /* experiment: static priority for the younger half of the workgroup (MI355X_MICROARCH.md, "two waves per SIMD", item 4) */
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    for (int u = 0; u < nu; ++u) {
        const int cb = u & 1;
        B2_STAMP();                                                 // 0: tile start
        unsigned char* const tile = b2_smem + cb * B2_TILE_BYTES;   // this tile's halo, then its scratch
        const int item = c_item, y0 = c_y0, x0 = c_x0;
        // ---- the next tile's halo -> the other buffer (free since the previous tile's closing barrier); past the end: nothing
        int n_item = item, n_y0 = y0, n_x0 = x0;
        const bool more = u + 1 < nu;                               // (wave-uniform)
        if (more) B2_UNIT(u + 1, n_item, n_y0, n_x0);
        const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.in) + (size_t)n_item * H * W * 64, 0, more ? item_bytes : 0, 0x00020000);
        const int n_org = ((n_y0 - 1) * W + n_x0 - 1) * 128;
        const bool n_interior = n_y0 > 0 && n_y0 + B2_IH - 1 <= H && n_x0 > 0 && n_x0 + B2_IW - 1 <= W;
        unsigned fence = 0;
        // ---- this tile's epilogue inputs: descriptors here, the loads ride on the first groups
        [[maybe_unused]] int eoff[2] = {0, 0};
        [[maybe_unused]] int rbase = 0;
        [[maybe_unused]] __amdgpu_buffer_rsrc_t rsR, rsA;
        if constexpr (FUSE) {
            const int ox = x0 + (lane & 31);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int oy = y0 + 2 * rp + n;
                eoff[n] = (ox < W && oy < H) ? (oy * W + ox) * 128 : 0x7fffffff;
            }
            const int rcol = x0 + ((tid >> 3) & 31), rrow = y0 + (tid >> 8);
            rbase = rcol < W ? (rrow * W + rcol) * 128 + (tid & 7) * 16 : 0x7fffffff;
            // (rows past the image: past the resource's range through the scalar offset)
            rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.resid) + (size_t)item * H * W * 64, 0, item_bytes, 0x00020000);
            // the shared-half pieces are the same pixels for every frame of a chain: fetched with its first frame (else: an empty resource)
            rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.addend) + (size_t)(item / p.add_div) * H * W * 64, 0, item_bytes, 0x00020000);
        }
        const bool chain_head = item % gT == 0;                     // (wave-uniform)
        const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)h_item * H * W * 64, 0, h_pending ? item_bytes : 0, 0x00020000);
        // the tile's first MFMAs take C = bias (register r of a lane = channel ech + r)
        f32x16 bias16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + ech + 4 * q);
            bias16[4 * q] = b4.x;
            bias16[4 * q + 1] = b4.y;
            bias16[4 * q + 2] = b4.z;
            bias16[4 * q + 3] = b4.w;
        }

        // ---- 12 groups (column tap kx, k-step ks): the 4 halo rows 2rp..2rp+3 serve the 3 row taps of both output rows - 4 pixel reads
        // + 3 weight reads feed 6 MFMAs; the operands of group g + 1 are requested before the MFMAs of g.  Nothing else in here.
        B2_STAMP();                                                 // 1: requests issued
        b2h8 px[2][4], wv[2][3];
#define B2_PX(g_, r_) (*reinterpret_cast<const b2h8*>(tile + paddr[(g_) >> 2][(g_) & 3] + (r_) * (B2_IW * 128)))
#define B2_WT(g_, ky_) (*reinterpret_cast<const b2h8*>(wlane + ((((ky_) * 3 + ((g_) >> 2)) * 4 + ((g_) & 3)) << 11)))
#pragma unroll
        for (int r = 0; r < 4; ++r) px[0][r] = B2_PX(0, r);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) wv[0][ky] = B2_WT(0, ky);
        auto group = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            constexpr int cur = g & 1;
            // --- the slice of memory traffic that rides on this group (one or two instructions: the memory pipe never sees a burst)
            if constexpr (g < 4) {                                  // a quarter of the previous tile's lines; this tile's epilogue inputs
                const int id = g * B2_THREADS + tid;
                const int pp = id >> 3, c = id & 7;
                const int sx = h_x0 + (pp & 31), sy = h_y0 + (pp >> 5);
                buffer_store_b128_guarded<B2_STORE_AUX>(held[g], rsH, (sx < W && sy < H) ? (sy * W + sx) * 128 + c * 16 : 0x7fffffff, 0);
                if constexpr (FUSE) {
                    // the residual pieces in ACCUMULATOR layout (a lane = a pixel, 16 bytes of its line): 64 lines per instruction instead of
                    // 8, but no staging through the scratch and no fourth barrier - measured 4.17 -> 4.13 ms per 1080p forward against
                    // whole-line loads + a stage + a barrier (-DB2_RESID_LINES)
#ifdef B2_RESID_LINES
                    rq[g] = __builtin_bit_cast(b2u4, __builtin_amdgcn_raw_buffer_load_b128(rsR, rbase, g * 2 * wbytes, 0));
#else
                    rq[g] = __builtin_bit_cast(b2u4, __builtin_amdgcn_raw_buffer_load_b128(rsR, eoff[g >> 1], (ech + 8 * (g & 1)) * 2, 0));
#endif
                    if (chain_head)
                        radd[g >> 1][g & 1] = __builtin_bit_cast(b2u4, __builtin_amdgcn_raw_buffer_load_b128(rsA, eoff[g >> 1], (ech + 8 * (g & 1)) * 2, 0));
                }
                if constexpr (WITH10)                               // W10[frame][k-step][channel tile mt]: 16 bytes per lane and k-step
                    xw[g] = __builtin_bit_cast(b2u4, __builtin_amdgcn_raw_buffer_load_b128(rsXW, lane * 16 + mt * 1024, ((item % gT) * 4 + g) * 2048, 0));
            }
            if constexpr (g >= 4 && g < 4 + B2_DMA_ITERS) {         // the next tile's halo -> the other buffer, a piece per group
                B2_DMA_PIECE(g - 4, rsn, n_org, n_interior, n_y0, n_x0, cb ^ 1);
                if constexpr (g == 4 + B2_DMA_ITERS - 1) fence = __builtin_amdgcn_raw_buffer_load_b32(rsn, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g < 11) {
#pragma unroll
                for (int r = 0; r < 4; ++r) px[cur ^ 1][r] = B2_PX(g + 1, r);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) wv[cur ^ 1][ky] = B2_WT(g + 1, ky);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if constexpr (g == 0) {
                        if (ky == 0) {
                            acc[n] = b2_mfma(wv[cur][ky], px[cur][n + ky], bias16);
                            continue;
                        }
                    }
                    acc[n] = b2_mfma(wv[cur][ky], px[cur][n + ky], acc[n]);
                }
            __builtin_amdgcn_sched_barrier(0);
        };
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 1>{});
        group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{});
        group(std::integral_constant<int, 4>{});
        group(std::integral_constant<int, 5>{});
        group(std::integral_constant<int, 6>{});
        group(std::integral_constant<int, 7>{});
        group(std::integral_constant<int, 8>{});
        group(std::integral_constant<int, 9>{});
        group(std::integral_constant<int, 10>{});
        group(std::integral_constant<int, 11>{});
#undef B2_PX
#undef B2_WT
        B2_STAMP();                                                 // 2: MFMA groups issued
        B2_STAMP();                                                 // 3: (the fence wait moved to the closing barrier)
        B2_BARRIER();
        B2_STAMP();                                                 // 4: barrier passed                                               // every wave is past its last operand read of this buffer: it is the scratch now

        // ---- serial phase.  Scratch = pixel-major lines, 16-byte pieces XOR-swizzled by (pixel >> 1) & 7 (conflict-free for the piece
        // writes, the line read-back and the 1x1 operand reads)
#ifdef B2_RESID_LINES
        if constexpr (FUSE) {                                       // the residual lines go where the output lines will be
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pp = k * 64 + (tid >> 3), c = tid & 7;
                *reinterpret_cast<b2u4*>(tile + pp * 128 + ((c ^ ((pp >> 1) & 7)) << 4)) = rq[k];
            }
            B2_BARRIER();
        }
#endif
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int h = 0; h < 2; ++h) {                           // row n, channels ech + 8h .. + 7: bias is in (initial C); addend, leaky_relu, residual, bf16
                const int j = lane & 31;
                const int c = 4 * mt + 2 * (lane >> 5) + h;         // piece of the pixel's line
                b2u4* const slot = reinterpret_cast<b2u4*>(tile + ((2 * rp + n) * 32 + j) * 128 + ((c ^ ((j >> 1) & 7)) << 4));
                [[maybe_unused]] b2u4 rr = {0, 0, 0, 0};
#ifdef B2_RESID_LINES
                if constexpr (FUSE) rr = *slot;
#else
                if constexpr (FUSE) rr = rq[2 * n + h];
#endif
                f32x4 v[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r0 = 8 * h + 4 * q;
                    v[q] = f32x4{acc[n][r0], acc[n][r0 + 1], acc[n][r0 + 2], acc[n][r0 + 3]};
                    if constexpr (FUSE) v[q] += b2_to_f32(b2u2{radd[n][h][2 * q], radd[n][h][2 * q + 1]});
                    v[q] = b2_lrelu4(v[q], eslope);
                    if constexpr (FUSE) v[q] += b2_to_f32(b2u2{rr[2 * q], rr[2 * q + 1]});
                }
                const b2u2 lo = b2_to_bf16(v[0]), hi = b2_to_bf16(v[1]);
                *slot = b2u4{lo.x, lo.y, hi.x, hi.y};
            }
        B2_STAMP();                                                 // 5: epilogue pieces written
        B2_BARRIER();                                               // the 256 lines are complete
        B2_STAMP();                                                 // 6: barrier passed
        {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                           // 2048 pieces, 4 per thread: whole 128-byte lines, 8 pixels per wave instruction
                const int id = k * B2_THREADS + tid;
                const int pp = id >> 3, c = id & 7;
                held[k] = *reinterpret_cast<const b2u4*>(tile + pp * 128 + ((c ^ ((pp >> 1) & 7)) << 4));
            }
            h_item = item;
            h_y0 = y0;
            h_x0 = x0;
            h_pending = true;
            if constexpr (WITH10) {                                 // conv10_i: 4 k-steps x 2 rows, B = the lines of this frame's tile, A = W10 of the frame
                const int j = lane & 31;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const unsigned char* bp = tile + ((2 * rp) * 32 + j) * 128 + (((2 * ks + (lane >> 5)) ^ ((j >> 1) & 7)) << 4);
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        bacc[n] = b2_mfma(__builtin_bit_cast(b2h8, xw[ks]), *reinterpret_cast<const b2h8*>(bp + n * 32 * 128), bacc[n]);
                }
            }
        }
        if constexpr (WITH10) {
            if (item % gT == gT - 1) {                              // (wave-uniform) the chain is complete: bias, leaky_relu, bf16, store; clear
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
                            v[q] = f32x4{bacc[n][r0], bacc[n][r0 + 1], bacc[n][r0 + 2], bacc[n][r0 + 3]} + *reinterpret_cast<const f32x4*>(p.x_bias + ech + r0);
                            v[q] = b2_lrelu4(v[q], 0.2f);
                        }
                        const b2u2 lo = b2_to_bf16(v[0]), hi = b2_to_bf16(v[1]);
                        buffer_store_b128_guarded<B2_STORE_AUX>(b2u4{lo.x, lo.y, hi.x, hi.y}, rsX, off, 16 * h);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) bacc[n][r] = 0.f;
                }
            }
        }
        c_item = n_item;
        c_y0 = n_y0;
        c_x0 = n_x0;
        B2_STAMP();                                                 // 7: lines picked up
        asm volatile("" ::"v"(fence));                              // the next tile's halo has landed (fence load: conv_sf.hip)
        B2_BARRIER();                                               // the scratch has been read: the buffer is free for the halo after next
    }
    {   // the last tile's lines
        const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)h_item * H * W * 64, 0, h_pending ? item_bytes : 0, 0x00020000);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = k * B2_THREADS + tid;
            const int pp = id >> 3, c = id & 7;
            const int sx = h_x0 + (pp & 31), sy = h_y0 + (pp >> 5);
            buffer_store_b128_guarded<B2_STORE_AUX>(held[k], rsH, (sx < W && sy < H) ? (sy * W + sx) * 128 + c * 16 : 0x7fffffff, 0);
        }
    }
#undef B2_DMA_HALO
#undef B2_DMA_PIECE
#undef B2_UNIT
}

// modes 0 - 2 of launch_conv3x3_bf16 (conv_bf16.hip) on the second-generation kernel
hipError_t launch_conv3x3_bf16_v2(const ConvBf16Params& p, int mode, hipStream_t s) {
    if (mode < 0 || mode > 2) return hipErrorInvalidValue;
    const int ncu = device_cu_count();
    if (!ncu) return hipErrorUnknown;
    const int grid = ncu >= 8 ? ncu / 8 * 8 : 8;                    // whole XCDs; surplus workgroups exit at once
    static std::atomic<int> attr_dev[64][3];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    const void* fn = mode == 1 ? reinterpret_cast<const void*>(conv3x3_bf16_v2_kernel<1>)
                   : mode == 2 ? reinterpret_cast<const void*>(conv3x3_bf16_v2_kernel<2>) : reinterpret_cast<const void*>(conv3x3_bf16_v2_kernel<0>);
    if (!attr_dev[dev][mode]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, B2_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_dev[dev][mode] = 1;
    }
    if (mode == 1) hipLaunchKernelGGL(conv3x3_bf16_v2_kernel<1>, dim3(grid), dim3(B2_THREADS), B2_LDS_BYTES, s, p);
    else if (mode == 2) hipLaunchKernelGGL(conv3x3_bf16_v2_kernel<2>, dim3(grid), dim3(B2_THREADS), B2_LDS_BYTES, s, p);
    else hipLaunchKernelGGL(conv3x3_bf16_v2_kernel<0>, dim3(grid), dim3(B2_THREADS), B2_LDS_BYTES, s, p);
    return hipGetLastError();
}

}  // namespace pfnl

#ifdef PFNL_B2_TIMING
extern "C" int pfnl_debug_read_b2_stamps(long long* host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pfnl::b2_dbg), n * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

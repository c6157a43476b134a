// fp32 3x3 64->64 convolution as Winograd F(2x2,3x3) on the f16 matrix pipe with exactly SPLIT operands (option conv3x3=wsplit): the 3x3
// convolutions of the progressive-fusion blocks (conv1_i, both halves of conv2_i; reference model/pfnl.py:49-51, 66-71).
//
// Why.  The direct split-f16 kernels (conv_split16.hip, conv_sf.hip) execute 3 MFMAs per product block and sit on the package power cap
// with the matrix pipe as the main consumer: what is left to win is MFMA work itself.  F(2x2,3x3) needs 16 products per
// 2x2 outputs instead of 36 - 2.25x fewer MFMAs for the same fp32-equivalent products:
//     Y = A^T [ (G g G^T) . (B^T d B) ] A,   U = G g G^T per (cin, cout): 16 positions (i, j),   V = B^T d B per (4x4 input patch, cin)
// The transforms run in fp32 (V) / fp64 on the host (U); both are then split exactly into binary16 pairs x = hi + lo' 2^-11 as in
// conv_split16.hip, and a product is hi hi + (hi lo' + lo' hi) 2^-11 with fp32 accumulation: >= 22 mantissa bits per product.
//
// What made it buildable: the transformed weights (16 x 64 x 64 x (hi, lo') = 256 KB per layer) do not fit LDS ("sized,
// not built"), but they fit the REGISTER file of a 4-wave workgroup: one wave per SIMD owns 512 registers, wave w keeps the four
// positions (w, j = 0..3) of U - 64 KB = 256 registers, the AGPR half of its file - for the whole launch as the MFMAs' B operands.
// No weight ever passes through LDS or is re-read from L2 per tile.
//
// Structure: workgroup = 4 waves, persistent over M-blocks of 8 x 16 output pixels (4 x 8 Winograd tiles = the M = 32 of an MFMA);
//   * the 10 x 18-pixel fp32 halo of an M-block (46 KB) enters LDS by LDS-DMA (buffer_load ... lds), double-buffered; out-of-image
//     pixels are out of the resource's range = zeros (SAME padding and ragged tiles); 16-byte chunks XOR-swizzled on the SOURCE side so
//     that the 16 lanes of a ds_read_b128 group (8 tile columns x 2 row parities) hit 16 different bank quads;
//   * wave w (Winograd row i = w) runs 4 passes j = 0..3: per 16-channel k-step it reads 2 patch rows x 2 patch columns x 8 channels per
//     lane, forms V_wj = (d[a1][b1] +- d[a2][b1]) +- (d[a1][b2] +- d[a2][b2]), splits it and issues 6 MFMAs (2 cout tiles x 3);
//     the finished position M_wj (fp32) is folded into the wave's half of the output transform: z_0 = M_w0 + M_w1 + M_w2, z_1 = M_w1 - M_w2 - M_w3;
//   * the four waves meet in LDS (64 KB: z[w][j'][tile][cout]); the second half of the transform Y[i'][j'] = sum_i A^T[i'][i] z_i[j'] is
//     taken by threads in OUTPUT order - 16 consecutive threads own the 256 bytes of one pixel - so that bias, addend, leaky-relu,
//     residual and the store all run on whole 256-byte lines with 16-byte accesses.
#include <cstring>
#include <vector>

#include "common.h"
#include "conv_wsplit.h"

#ifndef WS_STORE_AUX
#define WS_STORE_AUX 17     // output stores written through (sc0 sc1), as conv_split16.hip
#endif

namespace pfnl {

typedef _Float16 wsh8 __attribute__((ext_vector_type(8)));
typedef _Float16 wsh4 __attribute__((ext_vector_type(4)));
typedef unsigned wsu4 __attribute__((ext_vector_type(4)));
typedef unsigned wsu2 __attribute__((ext_vector_type(2)));

constexpr int WS_THREADS = 256;
constexpr int WS_TH = 8, WS_TW = 16;                                // output pixels of an M-block
constexpr int WS_IH = WS_TH + 2, WS_IW = WS_TW + 2;
constexpr int WS_TILE_BYTES = WS_IH * WS_IW * 256;                  // 46 080: fp32 NHWC, 256 B per pixel
// LDS map.  The two halo buffers sit 64 KB apart, so that an operand address moves from one to the other by flipping ONE bit (the 32
// address registers of a lane are flipped once per block instead of being rebuilt); the waves' z areas (16 KB each) fill the gaps:
//   [0, 46 080) halo 0 | [46 080, 62 464) z of wave 0 | [62 464, 62 720) bias | [65 536, 111 616) halo 1 | [111 616, 160 768) z of waves 1-3
constexpr int WS_BUF1 = 65536;
constexpr int WS_Z0 = WS_TILE_BYTES, WS_Z1 = WS_BUF1 + WS_TILE_BYTES;
constexpr int WS_BIAS = WS_Z0 + 16384;
constexpr int WS_LDS_BYTES = WS_Z1 + 3 * 16384;                     // 160 768 of 163 840
__host__ __device__ constexpr int ws_zoff(int w) { return w == 0 ? WS_Z0 : WS_Z1 + (w - 1) * 16384; }
constexpr float WS_SCALE = 2048.0f, WS_ISCALE = 1.0f / 2048.0f;
static_assert(WS_BIAS + 256 <= WS_BUF1, "LDS map");

// D (+)= A B with the B operand (the weights) in AGPRs.  Inline asm because this compiler allocates MFMA sources to VGPRs and uses the
// AGPR half only as spill space for them (4 v_accvgpr_read per MFMA); nothing inside the strings is padded by the compiler:
//   * the A operand is produced by VALU at least one MFMA earlier (the operands of k-step s + 1 are built in the gaps of step s), the
//     block's first one behind an explicit s_nop;  * accumulate chains (same D as C) need no states;  * ws_mfma_done() pays the
//     MFMA -> VALU-reader states.
#ifdef WS_X_ACC_AGPR   /* timing experiment: accumulators in AGPRs, U in VGPRs (only builds with WS_X_NOVALU) */
__device__ __forceinline__ void ws_mfma(f32x16& c, wsh8 a, wsh8 u) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(u));
}
__device__ __forceinline__ void ws_mfma0(f32x16& c, wsh8 a, wsh8 u) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&a"(c) : "v"(a), "v"(u));
}
#else
__device__ __forceinline__ void ws_mfma(f32x16& c, wsh8 a, wsh8 u) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(u));
}
__device__ __forceinline__ void ws_mfma0(f32x16& c, wsh8 a, wsh8 u) {   // C = 0: a pass's first products
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(u));
}
#endif
__device__ __forceinline__ void ws_mfma_done(f32x16& a, f32x16& b) {     // the MFMAs that wrote a, b have retired (8-pass XDL: 12 states; 22 here)
    asm volatile("s_nop 15\n\ts_nop 5" : "+v"(a), "+v"(b));
}

// one LDS-DMA instruction: lane L's 16 bytes at (resource, voff) -> LDS [lds_dst + 16 L] (conv_sf.hip, sf_dma16)
__device__ __forceinline__ void ws_dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}

#define WS_FENCE() __builtin_amdgcn_sched_barrier(0)

#ifdef PFNL_WS_TIMING   /* phase timeline of the kernel (tools/ws_timing.py); not part of the product build */
__device__ long long ws_dbg[256 * 4 * 256];
#define WS_STAMP() do { if (lane == 0 && dbg_n < 256) ws_dbg[(blockIdx.x * 4 + w) * 256 + dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define WS_STAMP() do {} while (0)
#endif

// FUSE: out = act(conv + bias + addend[item / add_div]) + resid (conv2_i's per-frame half); else out = act(conv + bias)
template <bool FUSE>
__global__ __launch_bounds__(WS_THREADS, 1) void conv_wsplit_kernel(ConvWsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];
    float* const bl = reinterpret_cast<float*>(ws_smem + WS_BIAS);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);         // Winograd row i of this wave
#ifdef PFNL_WS_TIMING
    int dbg_n = 0;
#endif
    const int H = p.H, W = p.W;
    const int tiles_x = (W + WS_TW - 1) / WS_TW, tiles_y = (H + WS_TH - 1) / WS_TH;
    const int per_item = tiles_x * tiles_y;
    const int item_bytes = H * W * 256;
    const int wbytes = W * 256;
    // M-blocks dealt out XCD by XCD (workgroup b runs on XCD b & 7): neighbouring workgroups of an XCD take x-adjacent blocks at the same
    // time, so the halo columns they share are served by that XCD's L2
    const int nblk = per_item * p.items;
    const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3, cpx = gridDim.x >> 3;
    const int per_xcd = (nblk + 7) >> 3;
    const int cbeg = xcd * per_xcd;
    const int ccnt = min(per_xcd, nblk - cbeg);
    if (xj >= ccnt) return;
    const int nb = (ccnt - xj + cpx - 1) / cpx;                     // M-blocks of this workgroup
#define WS_BLOCK(k_, item_, y0_, x0_)                                                            \
    do {                                                                                         \
        const int id_ = cbeg + xj + (k_) * cpx;                                                  \
        item_ = id_ / per_item;                                                                  \
        const int sp_ = id_ - item_ * per_item;                                                  \
        const int by_ = sp_ / tiles_x;                                                           \
        y0_ = by_ * WS_TH;                                                                       \
        x0_ = (sp_ - by_ * tiles_x) * WS_TW;                                                     \
    } while (0)

    // ---- the wave's four positions of U: 64 x 16 bytes per lane, resident for the whole launch (the "a" operands of ws_mfma); brought
    // in four batches so that no more than 64 VGPRs are in flight (every AGPR belongs to U: the allocator must never want one as a spill slot)
    wsh8 U[4][4][2][2];                                             // [j][k-step][cout tile][hi / lo']
    {
        const wsu4* src = reinterpret_cast<const wsu4*>(p.upack) + (size_t)w * (4 * 4 * 2 * 2 * 64) + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int part = 0; part < 2; ++part) {
                        U[j][ks][ct][part] = __builtin_bit_cast(wsh8, src[((((j * 4 + ks) * 2 + ct) * 2) + part) * 64]);
#ifndef WS_X_ACC_AGPR
                        asm volatile("" : "+a"(U[j][ks][ct][part]));
#endif
                    }
            WS_FENCE();
        }
    }
    if (tid < 64) bl[tid] = p.bias[tid];
    const float nscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -WS_SCALE)));
    const float slope = p.act ? 0.2f : 1.0f;

    // ---- operand side: lane = (tile m = lane & 31 -> ty = m >> 3, tx = m & 7; k-group g = lane >> 5: channels 16 ks + 8 g .. + 7)
    // B^T row w combines patch rows (a1, a2): t = d[a1] + sgn d[a2]
    const int a1 = w == 0 ? 0 : (w == 2 ? 2 : 1);
    const int a2 = w == 0 ? 2 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
    const float sgn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w == 1 ? 1.0f : -1.0f)));
    // LDS byte address of (patch row r = a1 / a2, patch column b, 16-byte chunk 4 ks + 2 g + e) in halo buffer 0 = (adr[r][b >> 1][ks] ^ 16 e) + b * 256:
    // pixel (2 ty + a, 2 tx + b), the chunk in slot chunk ^ sw, sw = ((tx + (b >> 1)) & 7) | (((ty + (a >> 1)) & 1) << 3).  The 16 of e = 0 are
    // kept (one v_xor per block moves them to the other buffer; e = 1: ^ 16 at the read) - 32 would leave the allocator short of VGPRs.
    int adr[2][2][4];
    {
        const int m = lane & 31, ty = m >> 3, tx = m & 7, g = lane >> 5;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int bh = 0; bh < 2; ++bh) {
                const int a = r ? a2 : a1;
                const int sw = ((tx + bh) & 7) | (((ty + (a >> 1)) & 1) << 3);
                const int base = ((2 * ty + a) * WS_IW + 2 * tx) * 256 + ((sw ^ (2 * g)) << 4);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) adr[r][bh][ks] = (base ^ (ks << 6)) ^ WS_BUF1;   // (the first block flips them to buffer 0)
            }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)ws_smem;

    // ---- halo DMA: wave w brings halo rows w, w + 4, w + 8, each as five instructions of four pixels (the fifth: pixels 16, 17 - lanes
    // 0..31 only; lanes that are switched off write nothing).  Lane L: pixel column 4 q + (L >> 4), LDS slot L & 15 <- source chunk
    // slot ^ sw(pixel).  (row >> 1) & 1 is the same for all rows of a wave, so the lane's part of the source offset depends on q & 3 only:
    // four values, rebuilt per block from an opaque copy of the lane id (kept across the k-steps they would cost the allocator the registers
    // it needs there: every AGPR belongs to U, a VGPR spilled into one evicts a quad of U to scratch); the rest of an offset is wave-uniform.
    auto dma_block = [&](int item, int y0, int x0, int buf) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in) + (size_t)item * H * W * 64, 0, item_bytes, 0x00020000);
        const int org = ((y0 - 1) * W + x0 - 1) * 256;
        const bool xin = x0 > 0 && x0 + WS_IW - 1 <= W;             // every halo column is inside the image (wave-uniform)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        int dl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dl[q] = (ln >> 4) * 256 + (((ln & 15) ^ (((2 * q + (ln >> 5)) & 7) | (((w >> 1) & 1) << 3))) << 4);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int row = w + 4 * t;
            if (t < 2 || row < WS_IH) {                             // (wave-uniform)
                const int gy = y0 - 1 + row;
                const int ub = (unsigned)gy < (unsigned)H ? org + row * wbytes : 0x7fff0000;   // rows outside the image: out of range = zeros
                const unsigned dst = lds0 + (buf ? WS_BUF1 : 0) + row * (WS_IW * 256);
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    int voff = ub + q * 1024 + dl[q & 3];
                    if (!xin) {
                        const int gx = x0 - 1 + 4 * q + (ln >> 4);
                        voff = (unsigned)gx < (unsigned)W ? voff : 0x7fffffff;
                    }
                    if (q < 4 || ln < 32) ws_dma16(rs, dst + q * 1024, voff);
                }
            }
        }
    };

    int c_item, c_y0, c_x0, n_item, n_y0, n_x0;
    WS_BLOCK(0, c_item, c_y0, c_x0);
    dma_block(c_item, c_y0, c_x0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    WS_STAMP();                                                     // (kernel entry -> first block: U, first halo)
    for (int kb = 0; kb < nb; ++kb) {
        WS_STAMP();                                                 // 0: block start
        {   // the next block's halo -> the other buffer (free since the previous block's closing barrier); past the end: this block again
            const int kn = min(kb + 1, nb - 1);
            WS_BLOCK(kn, n_item, n_y0, n_x0);
#ifndef WS_X_NODMA    /* timing experiments only (wrong results on purpose) */
            dma_block(n_item, n_y0, n_x0, (kb & 1) ^ 1);
#endif
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int bh = 0; bh < 2; ++bh)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) adr[r][bh][ks] ^= WS_BUF1;   // -> this block's buffer
        WS_STAMP();                                                 // 1: DMA issued, addresses flipped
        f32x16 z0[2], z1[2], accm[2], accc[2];
        // ---- the software pipeline of a block's 16 k-steps s = 4 j + ks.  The operands of step s + 1 are built, and the pixels of step
        // s + 2 requested, in the six gaps behind the MFMAs of step s (a lone wave hides ~5 single-issue instructions per MFMA; its own
        // LDS latency only if the read is issued four gaps ahead of its use): per 8-channel half e
        //   R: 4 ds_read_b128 (2 patch rows x 2 patch columns)   T: t = d[a1] +- d[a2] per column, v = t[b1] +- t[b2] (12 VALU)
        //   S1: hi = f16(v), v *= 2^11 (6 VALU)    S2: lo' (4 VALU)
        f32x4 d[2][4];                                              // [e][row a1 col b1, a2 b1, a1 b2, a2 b2]
#ifdef WS_X_NOWAIT
        f32x4 dd[2][4];
#endif
        f32x4 v[2];
        wsu2 hi[2], lo[2];
        wsh8 vh[2], vl[2];
        auto R = [&](int s, int e) __attribute__((always_inline)) {
            const int j = s >> 2, ks = s & 3;
            const int b1 = j == 0 ? 0 : 1, b2 = j == 3 ? 3 : 2;    // patch columns: j = 0: 0, 2; 1: 1, 2; 2: 1, 2; 3: 1, 3
#ifdef WS_X_NOREAD    /* timing experiments only (wrong results on purpose) */
            if (s > 1) return;
#endif
#ifdef WS_X_NOWAIT    /* timing experiments only: the reads are issued, nobody waits for them inside the k-steps */
            if (s > 1) {
                dd[e][0] = *reinterpret_cast<const f32x4*>(ws_smem + (adr[0][b1 >> 1][ks] ^ (e << 4)) + b1 * 256);
                dd[e][1] = *reinterpret_cast<const f32x4*>(ws_smem + (adr[1][b1 >> 1][ks] ^ (e << 4)) + b1 * 256);
                dd[e][2] = *reinterpret_cast<const f32x4*>(ws_smem + (adr[0][b2 >> 1][ks] ^ (e << 4)) + b2 * 256);
                dd[e][3] = *reinterpret_cast<const f32x4*>(ws_smem + (adr[1][b2 >> 1][ks] ^ (e << 4)) + b2 * 256);
                asm volatile("" : "+v"(dd[e][0]), "+v"(dd[e][1]), "+v"(dd[e][2]), "+v"(dd[e][3]));
                return;
            }
#endif
            d[e][0] = *reinterpret_cast<const f32x4*>(ws_smem + (adr[0][b1 >> 1][ks] ^ (e << 4)) + b1 * 256);
            d[e][1] = *reinterpret_cast<const f32x4*>(ws_smem + (adr[1][b1 >> 1][ks] ^ (e << 4)) + b1 * 256);
            d[e][2] = *reinterpret_cast<const f32x4*>(ws_smem + (adr[0][b2 >> 1][ks] ^ (e << 4)) + b2 * 256);
            d[e][3] = *reinterpret_cast<const f32x4*>(ws_smem + (adr[1][b2 >> 1][ks] ^ (e << 4)) + b2 * 256);
        };
        // NOTE on the instruction choice (tools/ubench/mfma_agpr_src): PACKED fp32 VALU (v_pk_fma / v_pk_mul / v_pk_add_f32) does not run in
        // the shadow of the wave's own MFMAs - each costs the matrix pipe its 4.5 cycles plus a restart (2 behind every MFMA: 53 cycles
        // per MFMA instead of 32) - while plain v_fma_f32 / v_add_f32 / v_add_u32 / v_xor_b32 hide completely (6 per gap: 34.5) and
        // v_cvt_pk_f16_f32 mostly.  Everything between the MFMAs is therefore written as SCALAR fp32 / integer instructions (inline asm:
        // the compiler would pair them again): twice the instructions, none of them on the matrix pipe's clock.
        // (one asm statement per stage: the compiler pads every asm boundary in front of a dependent instruction with an s_nop)
#define WS_T_ASM(OP_)                                                                                                  \
    asm("v_fma_f32 %4, %9, %21, %5\n\tv_fma_f32 %0, %17, %21, %13\n\t" OP_(0)                                       \
        "v_fma_f32 %4, %10, %21, %6\n\tv_fma_f32 %1, %18, %21, %14\n\t" OP_(1)                                      \
        "v_fma_f32 %4, %11, %21, %7\n\tv_fma_f32 %2, %19, %21, %15\n\t" OP_(2)                                      \
        "v_fma_f32 %4, %12, %21, %8\n\tv_fma_f32 %3, %20, %21, %16\n\t" OP_(3)                                      \
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(tt)                                                        \
        : "v"(d[e][0].x), "v"(d[e][0].y), "v"(d[e][0].z), "v"(d[e][0].w), "v"(d[e][1].x), "v"(d[e][1].y), "v"(d[e][1].z), "v"(d[e][1].w), \
          "v"(d[e][2].x), "v"(d[e][2].y), "v"(d[e][2].z), "v"(d[e][2].w), "v"(d[e][3].x), "v"(d[e][3].y), "v"(d[e][3].z), "v"(d[e][3].w), "s"(sgn))
#define WS_OP_ADD(c_) "v_add_f32 %" #c_ ", %4, %" #c_ "\n\t"
#define WS_OP_SUB(c_) "v_sub_f32 %" #c_ ", %4, %" #c_ "\n\t"      /* t1 - t2 */
#define WS_OP_RSUB(c_) "v_sub_f32 %" #c_ ", %" #c_ ", %4\n\t"     /* t2 - t1 */
        auto T = [&](int s, int e) __attribute__((always_inline)) {
            const int j = s >> 2;                                   // B^T row j over the columns: t0 - t2 | t1 + t2 | t2 - t1 | t1 - t3
            float r0, r1, r2, r3, tt;                               // per channel: t1 = d[a1][b1] +- d[a2][b1] (%4), t2 = d[a1][b2] +- d[a2][b2], v = t1 +- t2
#ifdef WS_X_NOT      /* timing experiments only */
            if (s > 1) return;
#endif
            if (j == 1) WS_T_ASM(WS_OP_ADD);
            else if (j == 2) WS_T_ASM(WS_OP_RSUB);
            else WS_T_ASM(WS_OP_SUB);
            v[e] = f32x4{r0, r1, r2, r3};
        };
        auto S1 = [&](int e) __attribute__((always_inline)) {        // hi = f16(v) (round to nearest even); v *= 2^11 as an exponent add
            unsigned h0, h1;                                        // (v = 0 / denormal becomes a number below 2^-100: lo' rounds to 0 as it should)
            asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5\n\t"
                "v_add_u32 %2, %6, %2\n\tv_add_u32 %3, %6, %3\n\tv_add_u32 %4, %6, %4\n\tv_add_u32 %5, %6, %5"
                : "=&v"(h0), "=&v"(h1), "+v"(v[e].x), "+v"(v[e].y), "+v"(v[e].z), "+v"(v[e].w) : "s"(0x05800000u));
            hi[e] = wsu2{h0, h1};
        };
        auto S2 = [&](int e) __attribute__((always_inline)) {        // lo' = f16(v 2^11 - hi 2^11), one rounding (conv_split16.hip, split4)
            unsigned l0, l1;
            asm("v_fma_mixlo_f16 %0, %2, %4, %5 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %2, %4, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mixlo_f16 %1, %3, %4, %7 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %1, %3, %4, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                : "=&v"(l0), "=&v"(l1) : "v"(hi[e].x), "v"(hi[e].y), "s"(nscale), "v"(v[e].x), "v"(v[e].y), "v"(v[e].z), "v"(v[e].w));
            lo[e] = wsu2{l0, l1};
        };
        auto finish = [&](int s) __attribute__((always_inline)) {   // the operands of step s as two register quads
            vh[s & 1] = __builtin_bit_cast(wsh8, wsu4{hi[0].x, hi[0].y, hi[1].x, hi[1].y});
            vl[s & 1] = __builtin_bit_cast(wsh8, wsu4{lo[0].x, lo[0].y, lo[1].x, lo[1].y});
            asm volatile("" : "+v"(vh[s & 1]), "+v"(vl[s & 1]));
        };
        // fill: steps 0 and 1 (their latencies are paid once per block)
        R(0, 0);
        R(0, 1);
        T(0, 0);
        S1(0);
        S2(0);
        R(1, 0);
        T(0, 1);
        S1(1);
        S2(1);
        R(1, 1);
        finish(0);
        asm volatile("s_nop 1" ::: "memory");                        // (the block's first operands are read by the MFMA right behind their last VALU write)
        WS_STAMP();                                                 // 2: pipeline filled
        WS_FENCE();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int j = s >> 2, ks = s & 3;
            const wsh8 ah = vh[s & 1], al = vl[s & 1];
            const bool nx = s < 15, nx2 = s < 14;
#ifdef WS_X_NOVALU    /* timing experiments only: MFMAs alone */
#define T(a_, b_) do {} while (0)
#define S1(a_) do {} while (0)
#define S2(a_) do {} while (0)
#define R(a_, b_) do {} while (0)
#endif
            if (ks == 0) ws_mfma0(accm[0], ah, U[j][ks][0][0]);
            else ws_mfma(accm[0], ah, U[j][ks][0][0]);
            WS_FENCE();
            if (nx) T(s + 1, 0);
            WS_FENCE();
            if (ks == 0) ws_mfma0(accm[1], ah, U[j][ks][1][0]);
            else ws_mfma(accm[1], ah, U[j][ks][1][0]);
            WS_FENCE();
#ifdef WS_X_NOS      /* timing experiments only */
#define S1(e_) do {} while (0)
#define S2(e_) do {} while (0)
#endif
            if (nx) S1(0);
            WS_FENCE();
            if (ks == 0) ws_mfma0(accc[0], ah, U[j][ks][0][1]);
            else ws_mfma(accc[0], ah, U[j][ks][0][1]);
            WS_FENCE();
            if (nx) S2(0);
            if (nx2) R(s + 2, 0);
            WS_FENCE();
            if (ks == 0) ws_mfma0(accc[1], ah, U[j][ks][1][1]);
            else ws_mfma(accc[1], ah, U[j][ks][1][1]);
            WS_FENCE();
            if (nx) T(s + 1, 1);
            WS_FENCE();
            ws_mfma(accc[0], al, U[j][ks][0][0]);
            WS_FENCE();
            if (nx) S1(1);
            WS_FENCE();
            ws_mfma(accc[1], al, U[j][ks][1][0]);
            WS_FENCE();
            if (nx) {
                S2(1);
                finish(s + 1);
            }
            if (nx2) R(s + 2, 1);
            WS_FENCE();
#ifdef WS_TIMING_FINE
            if (s == 4 || s == 5 || s == 6) WS_STAMP();             // (steps 5, 6 bracketed; 7 = the pass end below)
#endif
            if (ks == 3) {
#ifdef WS_TIMING_FINE
                if (j == 1) WS_STAMP();
#endif
                // position (w, j) is complete: fold the cross terms in, add it into the wave's half of the output transform
                // z_0 = M_0 + M_1 + M_2, z_1 = M_1 - M_2 - M_3
                ws_mfma_done(accm[0], accc[0]);
                ws_mfma_done(accm[1], accc[1]);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    if (j == 0) z0[ct] = accm[ct] + accc[ct] * WS_ISCALE;
                    if (j == 1) {
                        z1[ct] = accm[ct] + accc[ct] * WS_ISCALE;
                        z0[ct] += z1[ct];
                    }
                    if (j == 2) {
                        accm[ct] += accc[ct] * WS_ISCALE;
                        z0[ct] += accm[ct];
                        z1[ct] -= accm[ct];
                    }
                    if (j == 3) z1[ct] = (z1[ct] - accm[ct]) - accc[ct] * WS_ISCALE;
                }
                WS_FENCE();
#ifdef WS_TIMING_FINE
                if (j == 1) WS_STAMP();
#endif
            }
        }
        WS_STAMP();                                                 // 3: k-steps done
        // ---- the waves meet: z -> LDS.  (addresses of the exchange and of the output side: rebuilt here from an opaque thread id, not kept
        // across the k-steps)
        int te = tid;
        asm volatile("" : "+v"(te));
        // z[w][j'][m][cout] fp32; D layout: lane = cout 32 ct + (lane & 31), register r = tile (r >> 2) * 8 + (r & 3) + 4 (lane >> 5)
        unsigned char* const zw = ws_smem + ws_zoff(w) + ((te >> 5) & 1) * 1024 + (te & 31) * 4;
        // output side: thread = 16-byte group cg = tid & 15 of pixel column tid >> 4; unit k = pixel row k of the M-block
        const int cg = te & 15, pxx = te >> 4;
        const unsigned char* const zr = ws_smem + (pxx & 1) * 8192 + (pxx >> 1) * 256 + cg * 16;
        [[maybe_unused]] f32x4 add4[WS_TH], res4[WS_TH];
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)c_item * H * W * 64, 0, item_bytes, 0x00020000);
        const int gx = c_x0 + pxx;
        const int obase = (gx < W) ? (c_y0 * W + gx) * 256 + cg * 16 : 0x7fffffff;   // (rows past the image: past the end of the resource)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int off = (r >> 2) * 2048 + (r & 3) * 256 + ct * 128;
                *reinterpret_cast<float*>(zw + off) = z0[ct][r];
                *reinterpret_cast<float*>(zw + 8192 + off) = z1[ct][r];
            }
        WS_STAMP();                                                 // 4: z written
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the next block's halo has landed (requested a whole block ago)
        WS_STAMP();                                                 // 5: halo + earlier stores complete
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        WS_STAMP();                                                 // 6: past the barrier
        if constexpr (FUSE) {                                        // the addend / residual lines of the 8 rows (z's registers are free now)
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.addend) + (size_t)(c_item / p.add_div) * H * W * 64, 0, item_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.resid) + (size_t)c_item * H * W * 64, 0, item_bytes, 0x00020000);
#pragma unroll
            for (int k = 0; k < WS_TH; ++k) {
                add4[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, obase, k * wbytes, 0));
                res4[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, obase, k * wbytes, 0));
            }
        }
        // ---- output transform over the waves (A^T row i' = k & 1: z_0 + z_1 + z_2 | z_1 - z_2 - z_3), epilogue, whole 256-byte lines out
        const f32x4 bias4 = *reinterpret_cast<const f32x4*>(bl + 4 * cg);
#pragma unroll
        for (int k = 0; k < WS_TH; ++k) {
            const unsigned char* const zk = zr + (k >> 1) * 2048;
            f32x4 y;
            if ((k & 1) == 0) {
                y = *reinterpret_cast<const f32x4*>(zk + ws_zoff(0)) + *reinterpret_cast<const f32x4*>(zk + ws_zoff(1)) + *reinterpret_cast<const f32x4*>(zk + ws_zoff(2));
            } else {
                y = *reinterpret_cast<const f32x4*>(zk + ws_zoff(1)) - *reinterpret_cast<const f32x4*>(zk + ws_zoff(2)) - *reinterpret_cast<const f32x4*>(zk + ws_zoff(3));
            }
            y += bias4;
            if constexpr (FUSE) y += add4[k];
            const f32x4 sv = y * slope;
            asm("v_max_f32 %0, %1, %2" : "=v"(y.x) : "v"(y.x), "v"(sv.x));
            asm("v_max_f32 %0, %1, %2" : "=v"(y.y) : "v"(y.y), "v"(sv.y));
            asm("v_max_f32 %0, %1, %2" : "=v"(y.z) : "v"(y.z), "v"(sv.z));
            asm("v_max_f32 %0, %1, %2" : "=v"(y.w) : "v"(y.w), "v"(sv.w));
            if constexpr (FUSE) y += res4[k];
            buffer_store_b128_guarded<WS_STORE_AUX>(__builtin_bit_cast(pfnl_u32x4, y), rsO, obase, k * wbytes);
        }
        c_item = n_item;
        c_y0 = n_y0;
        c_x0 = n_x0;
        WS_STAMP();                                                 // 7: output stage issued
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // z has been read: the next block may overwrite it
    }
    WS_STAMP();
#undef WS_BLOCK
}

hipError_t launch_conv_wsplit(const ConvWsParams& p, hipStream_t s) {
    if (!p.in || !p.upack || !p.bias || !p.out || p.items < 1 || p.H < 1 || p.W < 1) return hipErrorInvalidValue;
    if ((p.addend == nullptr) != (p.resid == nullptr) || (p.addend && (p.add_div < 1 || p.items % p.add_div))) return hipErrorInvalidValue;
    if ((long long)p.H * p.W * 256 >= 0x7fffffffLL) return hipErrorInvalidValue;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    const int ncu = device_cu_count();
    if (!ncu) return hipErrorUnknown;
    const int grid = ncu >= 8 ? ncu / 8 * 8 : 8;                    // whole XCDs; surplus workgroups exit at once
    static std::atomic<int> attr_dev[64][2];
    const int mode = p.addend ? 1 : 0;
    const void* fn = mode ? reinterpret_cast<const void*>(conv_wsplit_kernel<true>) : reinterpret_cast<const void*>(conv_wsplit_kernel<false>);
    if (!attr_dev[dev][mode]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_dev[dev][mode] = 1;
    }
    if (mode) hipLaunchKernelGGL(conv_wsplit_kernel<true>, dim3(grid), dim3(WS_THREADS), WS_LDS_BYTES, s, p);
    else hipLaunchKernelGGL(conv_wsplit_kernel<false>, dim3(grid), dim3(WS_THREADS), WS_LDS_BYTES, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// host-side packing: U = G g G^T in fp64, split into binary16 pairs
size_t conv_wsplit_pack_halfs() { return (size_t)16 * 64 * 64 * 2; }

static uint16_t ws_f16_bits(float f) {
    const _Float16 h = (_Float16)f;                                 // round to nearest even
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}

// HWIO [3,3,cin_total,cout] rows [cin_begin, cin_begin+64) -> [i][j][ks][ct][part][lane][e] = U_ij[cin_begin + 16 ks + 8 (lane >> 5) + e][32 ct + (lane & 31)]
// (the B operand of v_mfma_f32_32x32x16_f16: lane holds B[k = 8 (lane >> 5) + e][n = lane & 31]); part 0 = f16(u), part 1 = f16((u - hi) 2^11)
void conv_wsplit_pack_weights(const float* hwio, int cin_total, int cin_begin, uint16_t* dst, int cout) {
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    std::vector<double> Uf((size_t)16 * 64 * 64);
    for (int ci = 0; ci < 64; ++ci)
        for (int co = 0; co < 64; ++co) {
            double g[3][3];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx)
                    g[ky][kx] = co < cout ? (double)hwio[((size_t)(ky * 3 + kx) * cin_total + cin_begin + ci) * cout + co] : 0.0;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    double u = 0.0;
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) u += G[i][ky] * g[ky][kx] * G[j][kx];
                    Uf[((size_t)(i * 4 + j) * 64 + ci) * 64 + co] = u;
                }
        }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int ks = 0; ks < 4; ++ks)
                for (int ct = 0; ct < 2; ++ct)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int ci = 16 * ks + 8 * (lane >> 5) + e, co = 32 * ct + (lane & 31);
                            const double u = Uf[((size_t)(i * 4 + j) * 64 + ci) * 64 + co];
                            const _Float16 hi = (_Float16)(float)u;
                            const float lo = (float)((u - (double)(float)hi) * 2048.0);
                            const size_t base = ((((size_t)((i * 4 + j) * 4 + ks) * 2 + ct) * 2) * 64 + lane) * 8 + e;
                            dst[base] = ws_f16_bits((float)hi);
                            dst[base + 512] = ws_f16_bits(lo);
                        }
}

}  // namespace pfnl

#ifdef PFNL_WS_TIMING
extern "C" int pfnl_debug_read_ws_stamps(long long* host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pfnl::ws_dbg), n * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

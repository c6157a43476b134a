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
constexpr int WS_NDMA = WS_IH * WS_IW / 4;                          // 45 DMA instructions of 1 KB (4 pixels)
constexpr int WS_Z_BYTES = 4 * 2 * 32 * 256;                        // z[w][j'][tile m][64 cout] fp32
constexpr int WS_LDS_BYTES = 2 * WS_TILE_BYTES + WS_Z_BYTES + 256;  // + bias: 157 952 of 163 840
constexpr float WS_SCALE = 2048.0f, WS_ISCALE = 1.0f / 2048.0f;
static_assert(WS_IH * WS_IW % 4 == 0, "whole DMA instructions");

// D (+)= A B with the B operand (the weights) in AGPRs.  Inline asm because this compiler allocates MFMA sources to VGPRs and uses the
// AGPR half only as spill space for them (4 v_accvgpr_read per MFMA); nothing inside the strings is padded by the compiler:
//   * the A operand is produced by VALU at least one whole k-step (6 MFMAs) earlier and pinned there (ws_pin), so no VALU -> MFMA-source
//     wait states are owed here;  * accumulate chains (same D as C) need none;  * ws_mfma_done() pays the MFMA -> VALU-reader states.
__device__ __forceinline__ void ws_mfma(f32x16& c, wsh8 a, wsh8 u) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(u));
}
__device__ __forceinline__ void ws_mfma0(f32x16& c, wsh8 a, wsh8 u) {   // C = 0: a pass's first products
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(u));
}
__device__ __forceinline__ void ws_mfma_done(f32x16& a, f32x16& b) {     // the MFMAs that wrote a, b have retired (8-pass XDL: 12 states; 22 here)
    asm volatile("s_nop 15\n\ts_nop 5" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void ws_pin(wsh8& a, wsh8& b) { asm volatile("" : "+v"(a), "+v"(b)); }

// x -> (hi, lo') for 4 values (conv_split16.hip, split4): hi = f16(x), lo' = f16(x 2^11 - hi 2^11), one rounding each
__device__ __forceinline__ void ws_split4(f32x4 v, wsu2& hi, wsu2& lo, float nscale) {
    const wsh4 h = __builtin_convertvector(v, wsh4);
    hi = __builtin_bit_cast(wsu2, h);
    const f32x4 t = v * WS_SCALE;
    unsigned l0, l1;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi.x), "s"(nscale), "v"(t.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(hi.x), "s"(nscale), "v"(t.y));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi.y), "s"(nscale), "v"(t.z));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(hi.y), "s"(nscale), "v"(t.w));
    lo = wsu2{l0, l1};
}

// one LDS-DMA instruction: lane L's 16 bytes at (resource, voff) -> LDS [lds_dst + 16 L] (conv_sf.hip, sf_dma16)
__device__ __forceinline__ void ws_dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}

// FUSE: out = act(conv + bias + addend[item / add_div]) + resid (conv2_i's per-frame half); else out = act(conv + bias)
template <bool FUSE>
__global__ __launch_bounds__(WS_THREADS, 1) void conv_wsplit_kernel(ConvWsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];
    unsigned char* const zl = ws_smem + 2 * WS_TILE_BYTES;
    float* const bl = reinterpret_cast<float*>(ws_smem + 2 * WS_TILE_BYTES + WS_Z_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);         // Winograd row i of this wave
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

    // ---- the wave's four positions of U: 64 x 16 bytes per lane, resident for the whole launch (the "a" operands of ws_mfma)
    wsh8 U[4][4][2][2];                                             // [j][k-step][cout tile][hi / lo']
    {
        const wsu4* src = reinterpret_cast<const wsu4*>(p.upack) + (size_t)w * (4 * 4 * 2 * 2 * 64) + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int part = 0; part < 2; ++part)
                        U[j][ks][ct][part] = __builtin_bit_cast(wsh8, src[((((j * 4 + ks) * 2 + ct) * 2) + part) * 64]);
    }
    if (tid < 64) bl[tid] = p.bias[tid];
    const float nscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -WS_SCALE)));
    const float slope = p.act ? 0.2f : 1.0f;

    // ---- operand side: lane = (tile m = lane & 31 -> ty = m >> 3, tx = m & 7; k-group g = lane >> 5: channels 16 ks + 8 g .. + 7)
    // B^T row w combines patch rows (a1, a2): t = d[a1] + sgn d[a2]
    const int a1 = w == 0 ? 0 : (w == 2 ? 2 : 1);
    const int a2 = w == 0 ? 2 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
    const float sgn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w == 1 ? 1.0f : -1.0f)));
    // LDS byte address of (patch row r = a1 / a2, patch column b, 16-byte chunk c) = A[r][b >> 1] ^ (c' << 4) + b * 256 with
    // c' = 4 ks + e (the k-group bit 2 g is folded into A): pixel (2 ty + a, 2 tx + b), chunk c = 4 ks + 2 g + e in slot c ^ sw,
    // sw = ((tx + (b >> 1)) & 7) | (((ty + (a >> 1)) & 1) << 3)
    int A[2][2];
    {
        const int m = lane & 31, ty = m >> 3, tx = m & 7, g = lane >> 5;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int bh = 0; bh < 2; ++bh) {
                const int a = r ? a2 : a1;
                const int sw = ((tx + bh) & 7) | (((ty + (a >> 1)) & 1) << 3);
                A[r][bh] = ((2 * ty + a) * WS_IW + 2 * tx) * 256 + ((sw ^ (2 * g)) << 4);
            }
    }
    // ---- exchange side: z[w][j'][m][cout] fp32; D layout: lane = cout 32 ct + (lane & 31), register r = tile (r >> 2) * 8 + (r & 3) + 4 (lane >> 5)
    unsigned char* const zw = zl + w * 16384 + (lane >> 5) * 1024 + (lane & 31) * 4;
    // ---- output side: thread = 16-byte group cg = tid & 15 of pixel column tid >> 4; unit k = pixel row k of the M-block
    const int cg = tid & 15, pxx = tid >> 4;
    const unsigned char* const zr = zl + (pxx & 1) * 8192 + (pxx >> 1) * 256 + cg * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)ws_smem;

    // DMA of block (item, y0, x0) -> buffer buf: instruction i = w + 4 k; lane L: halo pixel 4 i + (L >> 4), slot L & 15
    auto dma_block = [&](int item, int y0, int x0, int buf) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in) + (size_t)item * H * W * 64, 0, item_bytes, 0x00020000);
        const int org = ((y0 - 1) * W + x0 - 1) * 256;
#pragma unroll
        for (int k = 0; k < (WS_NDMA + 3) / 4; ++k) {
            const int i = w + 4 * k;
            if (k < (WS_NDMA + 3) / 4 - 1 || i < WS_NDMA) {
                const int px = 4 * i + (lane >> 4);
                const int row = px / WS_IW, col = px - row * WS_IW;
                const int sw = ((col >> 1) & 7) | (((row >> 1) & 1) << 3);
                const int gy = y0 - 1 + row, gx = x0 - 1 + col;
                const bool in = ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
                const int rel = row * wbytes + col * 256 + (((lane & 15) ^ sw) << 4);
                ws_dma16(rs, lds0 + buf * WS_TILE_BYTES + i * 1024, in ? org + rel : 0x7fffffff);
            }
        }
    };

    int c_item, c_y0, c_x0, n_item, n_y0, n_x0;
    WS_BLOCK(0, c_item, c_y0, c_x0);
    dma_block(c_item, c_y0, c_x0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(bl + 4 * cg);

    for (int kb = 0; kb < nb; ++kb) {
        const unsigned char* const tile = ws_smem + (kb & 1) * WS_TILE_BYTES;
        {   // the next block's halo -> the other buffer (free since the previous block's closing barrier); past the end: this block again
            const int kn = min(kb + 1, nb - 1);
            WS_BLOCK(kn, n_item, n_y0, n_x0);
            dma_block(n_item, n_y0, n_x0, (kb & 1) ^ 1);
        }
        f32x16 z0[2], z1[2];
        // operands of k-step (j, ks): 8 ds_read_b128, row / column combination, split
        auto operands = [&](int j, int ks, wsh8& vh, wsh8& vl) __attribute__((always_inline)) {
            const int b1 = j == 0 ? 0 : 1, b2 = j == 0 ? 2 : (j == 3 ? 3 : (j == 1 ? 2 : 2));   // columns: j = 0: 0, 2; 1: 1, 2; 2: 1, 2; 3: 1, 3
            f32x4 v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int cx = (4 * ks + e) << 4;
                const f32x4 d11 = *reinterpret_cast<const f32x4*>(tile + (A[0][b1 >> 1] ^ cx) + b1 * 256);
                const f32x4 d21 = *reinterpret_cast<const f32x4*>(tile + (A[1][b1 >> 1] ^ cx) + b1 * 256);
                const f32x4 d12 = *reinterpret_cast<const f32x4*>(tile + (A[0][b2 >> 1] ^ cx) + b2 * 256);
                const f32x4 d22 = *reinterpret_cast<const f32x4*>(tile + (A[1][b2 >> 1] ^ cx) + b2 * 256);
                const f32x4 t1 = d11 + sgn * d21, t2 = d12 + sgn * d22;
                // B^T row j over the columns: j = 0: t0 - t2; 1: t1 + t2; 2: t2 - t1; 3: t1 - t3
                v[e] = j == 1 ? t1 + t2 : (j == 2 ? t2 - t1 : t1 - t2);
            }
            wsu2 h0, l0, h1, l1;
            ws_split4(v[0], h0, l0, nscale);
            ws_split4(v[1], h1, l1, nscale);
            vh = __builtin_bit_cast(wsh8, wsu4{h0.x, h0.y, h1.x, h1.y});
            vl = __builtin_bit_cast(wsh8, wsu4{l0.x, l0.y, l1.x, l1.y});
            ws_pin(vh, vl);
        };
        wsh8 vh[2], vl[2];
        operands(0, 0, vh[0], vl[0]);
        asm volatile("s_nop 1" ::: "memory");                        // (the block's first operands are read by the MFMA right behind their last VALU write)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x16 accm[2], accc[2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int s = j * 4 + ks;                           // flat step; operands of step s + 1 are built between this step's MFMAs
                const wsh8 ah = vh[s & 1], al = vl[s & 1];
                if (ks == 0) {
                    ws_mfma0(accm[0], ah, U[j][ks][0][0]);
                    ws_mfma0(accm[1], ah, U[j][ks][1][0]);
                    ws_mfma0(accc[0], ah, U[j][ks][0][1]);
                    ws_mfma0(accc[1], ah, U[j][ks][1][1]);
                } else {
                    ws_mfma(accm[0], ah, U[j][ks][0][0]);
                    ws_mfma(accm[1], ah, U[j][ks][1][0]);
                    ws_mfma(accc[0], ah, U[j][ks][0][1]);
                    ws_mfma(accc[1], ah, U[j][ks][1][1]);
                }
                if (s < 15) operands((s + 1) >> 2, (s + 1) & 3, vh[(s + 1) & 1], vl[(s + 1) & 1]);
                ws_mfma(accc[0], al, U[j][ks][0][0]);
                ws_mfma(accc[1], al, U[j][ks][1][0]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // position (w, j) is complete: fold the cross terms in, add it into the wave's half of the output transform
            ws_mfma_done(accm[0], accc[0]);
            ws_mfma_done(accm[1], accc[1]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const f32x16 M = accm[ct] + accc[ct] * WS_ISCALE;
                if (j == 0) z0[ct] = M;
                if (j == 1) {
                    z0[ct] += M;
                    z1[ct] = M;
                }
                if (j == 2) {
                    z0[ct] += M;
                    z1[ct] -= M;
                }
                if (j == 3) z1[ct] -= M;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the waves meet: z -> LDS
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int off = (r >> 2) * 2048 + (r & 3) * 256 + ct * 128;
                *reinterpret_cast<float*>(zw + off) = z0[ct][r];
                *reinterpret_cast<float*>(zw + 8192 + off) = z1[ct][r];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the next block's halo has landed (requested a whole block ago)
        __syncthreads();
        // ---- output transform over the waves, epilogue, whole 256-byte lines out
        {
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)c_item * H * W * 64, 0, item_bytes, 0x00020000);
            const int gx = c_x0 + pxx;
            const int obase = (gx < W) ? (c_y0 * W + gx) * 256 + cg * 16 : 0x7fffffff;
            [[maybe_unused]] __amdgpu_buffer_rsrc_t rsA = rsO, rsR = rsO;
            if constexpr (FUSE) {
                rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.addend) + (size_t)(c_item / p.add_div) * H * W * 64, 0, item_bytes, 0x00020000);
                rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid) + (size_t)c_item * H * W * 64, 0, item_bytes, 0x00020000);
            }
            [[maybe_unused]] f32x4 add4[WS_TH], res4[WS_TH];
            if constexpr (FUSE) {
#pragma unroll
                for (int k = 0; k < WS_TH; ++k) {
                    const int off = (c_y0 + k < H) ? obase : 0x7fffffff;   // (rows past the image; obase may already be out of range)
                    add4[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, k * wbytes, 0));
                    res4[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, off, k * wbytes, 0));
                }
            }
#pragma unroll
            for (int k = 0; k < WS_TH; ++k) {
                const int ip = k & 1, ty = k >> 1;                  // output row k = 2 ty + i'
                const unsigned char* const zk = zr + ty * 2048;
                f32x4 y;
                if (ip == 0) {                                      // A^T row 0: z_0 + z_1 + z_2
                    y = *reinterpret_cast<const f32x4*>(zk) + *reinterpret_cast<const f32x4*>(zk + 16384) + *reinterpret_cast<const f32x4*>(zk + 32768);
                } else {                                            // A^T row 1: z_1 - z_2 - z_3
                    y = *reinterpret_cast<const f32x4*>(zk + 16384) - *reinterpret_cast<const f32x4*>(zk + 32768) - *reinterpret_cast<const f32x4*>(zk + 49152);
                }
                y += bias4;
                if constexpr (FUSE) y += add4[k];
                const f32x4 sv = y * slope;
                asm("v_max_f32 %0, %1, %2" : "=v"(y.x) : "v"(y.x), "v"(sv.x));
                asm("v_max_f32 %0, %1, %2" : "=v"(y.y) : "v"(y.y), "v"(sv.y));
                asm("v_max_f32 %0, %1, %2" : "=v"(y.z) : "v"(y.z), "v"(sv.z));
                asm("v_max_f32 %0, %1, %2" : "=v"(y.w) : "v"(y.w), "v"(sv.w));
                if constexpr (FUSE) y += res4[k];
                const int off = (c_y0 + k < H) ? obase : 0x7fffffff;
                buffer_store_b128_guarded<WS_STORE_AUX>(__builtin_bit_cast(pfnl_u32x4, y), rsO, off, k * wbytes);
            }
        }
        c_item = n_item;
        c_y0 = n_y0;
        c_x0 = n_x0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // z has been read: the next block may overwrite it
    }
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

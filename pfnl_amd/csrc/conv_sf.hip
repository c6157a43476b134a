// 3x3 64->64 convolution whose INPUT is in the split format (SF, conv_split16.h): both halves of conv2_i of the progressive-fusion
// blocks (reference model/pfnl.py:51, 69-71) - their inputs, conv1_i's and conv10_i's outputs, only ever feed MFMA operands.
//
// Arithmetic: that of conv_split16.hip (x = hi + lo' 2^-11 exactly split binary16 operands, three v_mfma_f32_32x32x16_f16 per
// product block, fp32 accumulation, the two cross terms in a second accumulator folded in with 2^-11 once per tile).
// What changes against conv3x3_split16_kernel is everything AROUND the MFMAs:
//   * the halo tile of a unit ((tile, half of the input channels): 10 x 34 pixels x 128 B) is already in operand form in HBM, so it
//     goes HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 43 wave instructions of 1 KB per unit): no staging registers, no
//     split arithmetic (48 VALU per thread and unit), no ds_write (12 per thread and unit).  The LDS layout keeps its XOR swizzle
//     (16-byte chunk c of pixel (py, px) sits in slot c ^ ((px >> 1) & 7)): a DMA instruction writes 1 KB linearly (lane L ->
//     M0 + 16 L), so the swizzle is applied to the SOURCE - lane L fetches the chunk that belongs in its slot.  Out-of-image
//     pixels: an out-of-range buffer offset makes the DMA write zeros (tools/ubench/lds_dma.hip: measured, with everything
//     else the kernel relies on - destinations above 64 KB, no compaction under EXEC, the immediate offset moving BOTH sides);
//   * MFMA roles swapped (A = pixels, B = weights): D[pixel][cout], i.e. a lane owns ONE output channel and 16 pixels of a row,
//     and 32 lanes x 4 B are a full 128-byte line of an NHWC pixel - the epilogue (bias as the initial C, leaky-relu, addend,
//     residual) runs from registers with dword loads / stores, without the trip through an LDS scratch tile, its two barriers and
//     the constraint that the other halo buffer be free for it in the first two thirds of a unit.  Rows past the image, columns
//     past it and "no tile pending" cost no branches: the output / residual / addend resources are built per ROW with
//     num_records = W * 256 bytes (or 0), and the range check covers voffset + soffset + immediate;
//   * completion of the DMA is waited for with a FENCE LOAD: a compiler-visible 4-byte buffer load issued right after the
//     unit's DMA instructions, consumed (an empty asm taking it as input) right before the unit's closing barrier.  vmcnt is in
//     order, so when the compiler's own - exactly counted - wait for that load has passed, everything older has landed; the
//     compiler knows nothing of the asm DMAs, which only ever makes its other waits stronger, never weaker.
// Work order, weight slices (24 KB per column tap, L2 -> registers -> LDS following the taps consumed), the operand pipeline and
// the chain order of the per-frame half are those of conv3x3_split16_kernel.
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"
#include "conv_split16.h"

#ifndef SF_STORE_AUX
#define SF_STORE_AUX 17     // output stores written through (sc0 sc1), as conv_split16.hip measured
#endif

namespace pfnl {

typedef _Float16 sfh8 __attribute__((ext_vector_type(8)));
typedef unsigned sfu4 __attribute__((ext_vector_type(4)));

constexpr int SF_THREADS = 512;
constexpr int SF_TH = 8, SF_TW = 32;
constexpr int SF_IH = SF_TH + 2, SF_IW = SF_TW + 2;
constexpr int SF_NPIX = SF_IH * SF_IW;                              // 340
constexpr int SF_NDMA = (SF_NPIX + 7) / 8;                          // 43 DMA instructions of 8 pixels x 128 B
constexpr int SF_TILE_BYTES = SF_NDMA * 1024;                       // 44 032 (the last instruction's 4 surplus pixels land in padding)
constexpr int SF_W_BYTES = 3 * 2 * 3 * 2 * 2 * 1024;                // 73 728: [kx][ks][ky][m][hi/lo][lane] x 16 B, one channel half
constexpr int SF_SLOT_BYTES = SF_W_BYTES / 3;
constexpr int SF_LDS_BYTES = 2 * SF_TILE_BYTES + SF_W_BYTES + 64 * 4;   // 162 048 of 163 840
constexpr int SF_DMA_ITERS = (SF_NDMA + 7) / 8;                     // 6 per wave (waves 3..7: 5)
constexpr int SF_WITERS = SF_SLOT_BYTES / 16 / SF_THREADS;          // 3
constexpr float SF_ISCALE = 1.0f / 2048.0f;
static_assert(SF_LDS_BYTES <= 160 * 1024, "LDS budget");

// one LDS-DMA instruction: lane L's 16 bytes at (resource, voff) -> LDS [lds_dst + 16 L].  m0 is compiler-reserved: saved / restored.
__device__ __forceinline__ void sf_dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}

// 4-byte buffer store that is safe to follow with anything (common.h, buffer_store_b128_guarded: same reasoning)
template <int AUX>
__device__ __forceinline__ void sf_store_b32(float v, __amdgpu_buffer_rsrc_t rs, int voffset, int soffset) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voffset, soffset, AUX);
    asm volatile("s_nop 1" ::"v"(v));
}

template <int AUX>
__device__ __forceinline__ void sf_store_u32(unsigned v, __amdgpu_buffer_rsrc_t rs, int voffset, int soffset) {
    __builtin_amdgcn_raw_buffer_store_b32(v, rs, voffset, soffset, AUX);
    asm volatile("s_nop 1" ::"v"(v));
}
typedef unsigned sfu2 __attribute__((ext_vector_type(2)));
typedef _Float16 sfh4 __attribute__((ext_vector_type(4)));
// x -> (hi, lo') for 4 values, exactly as conv_split16.hip's split4 (same instructions, same bits): hi = f16(x), lo' = f16(x 2^11 - hi 2^11)
__device__ __forceinline__ void sf_split4(f32x4 v, sfu2& hi, sfu2& lo, float nscale) {
    const sfh4 h = __builtin_convertvector(v, sfh4);
    hi = __builtin_bit_cast(sfu2, h);
    f32x4 t;
    asm("v_mul_f32 %0, %4, %5\n\tv_mul_f32 %1, %4, %6\n\tv_mul_f32 %2, %4, %7\n\tv_mul_f32 %3, %4, %8"
        : "=&v"(t.x), "=&v"(t.y), "=&v"(t.z), "=&v"(t.w) : "s"(2048.0f), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    unsigned l0, l1;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi.x), "s"(nscale), "v"(t.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(hi.x), "s"(nscale), "v"(t.y));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi.y), "s"(nscale), "v"(t.z));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(hi.y), "s"(nscale), "v"(t.w));
    lo = sfu2{l0, l1};
}

__device__ __forceinline__ f32x16 sf_mfma(sfh8 a, sfh8 b, f32x16 c) {
#ifdef SF_X_NOMFMA   /* timing experiments only (wrong results on purpose; tools/sf_variants.sh) */
    c[0] += (float)a[0] * (float)b[0];
    return c;
#endif
#ifdef PFNL_X_MFMA16   /* timing experiment only (wrong results on purpose): the same FLOPs as two v_mfma_f32_16x16x32_f16 on the same operand registers */
    {
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        f32x4_ lo = {c[0], c[1], c[2], c[3]}, hi = {c[4], c[5], c[6], c[7]};
        lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, hi, 0, 0, 0);
        c[0] = lo[0]; c[1] = lo[1]; c[2] = lo[2]; c[3] = lo[3];
        c[4] = hi[0]; c[5] = hi[1]; c[6] = hi[2]; c[7] = hi[3];
        return c;
    }
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
#ifdef SF_X_NODMA
#define SF_DMA16(rs_, dst_, voff_) do { if ((voff_) == 0x12345) sf_dma16(rs_, dst_, voff_); } while (0)
#else
#define SF_DMA16(rs_, dst_, voff_) sf_dma16(rs_, dst_, voff_)
#endif

// MODE 0: out = act(conv + bias).   MODE 1 (conv2_i per-frame half): out = act(conv + bias + addend[item / add_div]) + resid.
template <int MODE>
__global__ __launch_bounds__(SF_THREADS, 1) void conv3x3_sf_kernel(ConvSplitParams p) {
    constexpr bool FUSE = MODE == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
    unsigned char* const wl = sf_smem + 2 * SF_TILE_BYTES;
    float* const bl = reinterpret_cast<float*>(sf_smem + 2 * SF_TILE_BYTES + SF_W_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rp = wave >> 1;                                       // rows 2rp, 2rp+1 of the tile
    const int nt = wave & 1;                                        // output channels 32nt .. 32nt+31
    const int H = p.H, W = p.W;
    const int tiles_x = (W + SF_TW - 1) / SF_TW, tiles_y = (H + SF_TH - 1) / SF_TH;
    const int per_item = tiles_x * tiles_y;
    const int item_bytes = H * W * 256;
    const int wbytes = W * 256;
    // work order: chains of the gT frames of a clip at one spatial tile, dealt out XCD by XCD (conv_split16.hip); p.flat: single tiles
    // (a chain keeps the addend's tile in L2 for its frames, but a launch with fewer chains than workgroups leaves CUs idle)
    const int gT = (FUSE && !p.flat) ? p.add_div : 1;
    const int nchains = per_item * (p.items / gT);
    const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3, cpx = gridDim.x >> 3;
    const int per_xcd = (nchains + 7) >> 3;
    const int cbeg = xcd * per_xcd;
    const int ccnt = min(per_xcd, nchains - cbeg);
    if (xj >= ccnt) return;
    const int nt_tiles = ((ccnt - xj + cpx - 1) / cpx) * gT;        // tiles of this workgroup
#define SF_TILE(k_, item_, y0_, x0_)                                                             \
    do {                                                                                         \
        const int ci_ = (k_) / gT, f_ = (k_) - ci_ * gT;                                         \
        const int ch_ = cbeg + xj + ci_ * cpx;                                                   \
        const int cl_ = ch_ / per_item;                                                          \
        const int sp_ = ch_ - cl_ * per_item;                                                    \
        item_ = cl_ * gT + f_;                                                                   \
        const int ty_ = sp_ / tiles_x;                                                           \
        y0_ = ty_ * SF_TH;                                                                       \
        x0_ = (sp_ - ty_ * tiles_x) * SF_TW;                                                     \
    } while (0)
#define SF_HALF(u_) ((((u_) >> 1) ^ (u_)) & 1)

    // weights of half 0 + bias: requested here, written to LDS after the first halo has been requested too
    sfu4 w0reg[SF_W_BYTES / 16 / SF_THREADS];
#pragma unroll
    for (int k = 0; k < SF_W_BYTES / 16 / SF_THREADS; ++k) w0reg[k] = reinterpret_cast<const sfu4*>(p.wpack)[k * SF_THREADS + tid];
    const float bias_r = tid < 64 ? p.bias[tid] : 0.f;

    // DMA map: instruction i = wave + 8 k covers halo pixels 8 i .. 8 i + 7 (linear, 34 per row); lane L -> pixel 8 i + (L >> 3),
    // LDS slot L & 7, which holds chunk (L & 7) ^ ((px >> 1) & 7) of that pixel.  Constant for the life of the kernel:
    // `dgrel` = byte offset of the lane's source chunk relative to the halo origin, `dpk` = py | px << 8 (border test).
    int dgrel[SF_DMA_ITERS], dpk[SF_DMA_ITERS];
#pragma unroll
    for (int k = 0; k < SF_DMA_ITERS; ++k) {
        const int pix = 8 * (wave + 8 * k) + (lane >> 3);
        const int py = pix / SF_IW, px = pix - py * SF_IW;
        const int c = (lane & 7) ^ ((px >> 1) & 7);
        dgrel[k] = py * wbytes + px * 256 + c * 16;
        dpk[k] = py | (px << 8);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)sf_smem;             // LDS byte address of halo buffer 0
    // (the asm DMAs are invisible to the compiler's vmcnt bookkeeping; the wave-uniform branch around the 6th only skips asm)
#define SF_DMA_ALL(rs_, org_, interior_, y0_, x0_, buf_)                                         \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < SF_DMA_ITERS; ++k_) {                            \
            const int i_ = wave + 8 * k_;                                                        \
            if (k_ < SF_DMA_ITERS - 1 || i_ < SF_NDMA) {                                         \
                const int gy_ = (y0_) + (dpk[k_] & 0xff) - 1, gx_ = (x0_) + (dpk[k_] >> 8) - 1;  \
                const bool in_ = (interior_) || ((unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W && (dpk[k_] & 0xff) < SF_IH); \
                SF_DMA16(rs_, lds0 + (buf_) * SF_TILE_BYTES + i_ * 1024, in_ ? (org_) + dgrel[k_] : 0x7fffffff); \
            }                                                                                    \
        }                                                                                        \
    } while (0)

    // operand addresses (conv_split16.hip): pixel operand of (column tap kx, k-step ks, part) = chunk 4*part + 2*ks + (lane >> 5)
    // of halo pixel (row 2*rp + ..., column (lane & 31) + kx); weights: 16 bytes per lane
    int paddr[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = (lane & 31) + kx;
        paddr[kx] = ((2 * rp) * SF_IW + col) * 128 + ((((lane >> 5)) ^ ((col >> 1) & 7)) << 4);
    }
    const int lo_xor = 4 << 4;
    const unsigned char* const wlane = wl + nt * 2048 + lane * 16;

    // D[pixel][cout]: lane = output channel 32 nt + (lane & 31); register r = pixel column drow(r, lane) of the row
    const int ech = 32 * nt + (lane & 31);
    f32x16 accm[2], accc[2], accp[2];                               // [output row]: hi.hi / cross products (x 2^11) / finished tile
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accm[n][r] = 0.f;
            accc[n][r] = 0.f;
            accp[n][r] = 0.f;
        }
    int ex0p = 0, ey0p = 0, eitemp = 0;                             // the tile awaiting its epilogue
    bool pending = false;
    const float slope = p.act ? 0.2f : 1.0f;

    // ---- epilogue (from registers) --------------------------------------------------------------
    // row n of the pending tile in unit n of the next one.  Register 4 q + j of a lane = pixel column 8 q + j + 4 (lane >> 5):
    // byte offset in the row = evoff + j * 256 (immediate) + q * 2048 (scalar offset).
    __amdgpu_buffer_rsrc_t rsO;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsR, rsA;
    int evoff = 0;
    [[maybe_unused]] float radd[16], rres[16];
    auto row_setup = [&](int n) __attribute__((always_inline)) {
        const int ey = ey0p + 2 * rp + n;
        const int nrec = (pending && ey < H) ? wbytes : 0;
        const size_t row = ((size_t)eitemp * H + ey) * W * 64;
        rsO = __builtin_amdgcn_make_buffer_rsrc(p.out + row, 0, nrec, 0x00020000);
        if constexpr (FUSE) {
            rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid) + row, 0, nrec, 0x00020000);
            rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.addend) + ((size_t)(eitemp / p.add_div) * H + ey) * W * 64, 0, nrec, 0x00020000);
        }
        evoff = (ex0p + 4 * (lane >> 5)) * 256 + ech * 4;
    };
    auto quarter_request = [&](int q) __attribute__((always_inline)) {
        if constexpr (FUSE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#ifdef SF_X_NOADDEND
                radd[4 * q + j] = 0.25f;
#else
                radd[4 * q + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, evoff + j * 256, q * 2048, 0));
#endif
#ifdef SF_X_NORESID
                rres[4 * q + j] = 0.125f;
#else
                rres[4 * q + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, evoff + j * 256, q * 2048, 0));
#endif
            }
        }
    };
    auto quarter_finish = [&](int n, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = accp[n][4 * q + j];                           // (the bias is already in: initial C of the tile)
            if constexpr (FUSE) v += radd[4 * q + j];
            const float sv = v * slope;
            asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(sv)); // leaky_relu(0.2) or identity (slope 1), branch-free
            if constexpr (FUSE) v += rres[4 * q + j];
#ifdef SF_X_NOSTORE
            if (v == 1.2345e30f)
#endif
            sf_store_b32<SF_STORE_AUX>(v, rsO, evoff + j * 256, q * 2048);
        }
    };

    // ---- weight replacement: a slot (24 KB, one column tap) travels L2 -> registers -> LDS, 3 x 16 B per thread
    sfu4 wnx[SF_WITERS];
    auto w_request = [&](int half, int slot) __attribute__((always_inline)) {
        const sfu4* src = reinterpret_cast<const sfu4*>(p.wpack) + (size_t)half * (SF_W_BYTES / 16) + slot * (SF_SLOT_BYTES / 16);
#pragma unroll
        for (int k = 0; k < SF_WITERS; ++k) wnx[k] = src[k * SF_THREADS + tid];
    };
    auto w_write = [&](int slot) __attribute__((always_inline)) {
        sfu4* dst = reinterpret_cast<sfu4*>(wl + slot * SF_SLOT_BYTES);
#pragma unroll
        for (int k = 0; k < SF_WITERS; ++k) dst[k * SF_THREADS + tid] = wnx[k];
    };
#define SF_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // ---- prologue: halo of unit 0 -> buffer 0 by DMA; weights of half 0 -> LDS --------------------------------
    int c_item, c_y0, c_x0, n_item, n_y0, n_x0;
    SF_TILE(0, c_item, c_y0, c_x0);
    n_item = c_item;
    n_y0 = c_y0;
    n_x0 = c_x0;
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in) + (size_t)c_item * H * W * 64, 0, item_bytes, 0x00020000);
        const int org = ((c_y0 - 1) * W + c_x0 - 1) * 256 + SF_HALF(0) * 128;
        SF_DMA_ALL(rs, org, false, c_y0, c_x0, 0);
        const unsigned fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < SF_W_BYTES / 16 / SF_THREADS; ++k) reinterpret_cast<sfu4*>(wl)[k * SF_THREADS + tid] = w0reg[k];
        if (tid < 64) bl[tid] = bias_r;
        asm volatile("" ::"v"(fence));                              // the compiler's wait for the fence load: the DMAs have landed
    }
    __syncthreads();
    const float bias_l = bl[ech];

    for (int kt = 0; kt < nt_tiles; ++kt) {
        const int half_a = kt & 1;                                  // channel half of unit A; unit B: the other one
        auto unit = [&](auto par) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par)::value;               // 0: unit A, 1: unit B
            constexpr int cb = PAR;                                 // LDS buffer of this unit
            const unsigned char* const tile = sf_smem + cb * SF_TILE_BYTES;
            sfh8 X[4][2], Wv[2][2];
#define SF_PX(g_, r_, part_) (*reinterpret_cast<const sfh8*>(tile + (paddr[(g_) >> 1] ^ (((part_) ? lo_xor : 0) | (((g_) & 1) << 5))) + (r_) * (SF_IW * 128)))
#define SF_WT(g_, ky_, part_) (*reinterpret_cast<const sfh8*>(wlane + (((g_) * 3 + (ky_)) << 12) + ((part_) << 10)))
            X[0][0] = SF_PX(0, 0, 0);
            X[0][1] = SF_PX(0, 0, 1);
            X[1][0] = SF_PX(0, 1, 0);
            X[1][1] = SF_PX(0, 1, 1);
            Wv[0][0] = SF_WT(0, 0, 0);
            Wv[0][1] = SF_WT(0, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            // the NEXT unit's halo -> the other buffer (free since the previous unit's closing barrier): unit A asks for the other
            // half of ITS tile, unit B for the first half of the next tile
            const int q_item = PAR == 0 ? c_item : n_item, y0q = PAR == 0 ? c_y0 : n_y0, x0q = PAR == 0 ? c_x0 : n_x0;
            const int q_half = half_a ^ 1;                          // (boustrophedon: unit B's half is also the next tile's first half)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.in) + (size_t)q_item * H * W * 64, 0, item_bytes, 0x00020000);
            const int org = ((y0q - 1) * W + x0q - 1) * 256 + q_half * 128;
            const bool interior = y0q > 0 && y0q + SF_IH - 1 <= H && x0q > 0 && x0q + SF_IW - 1 <= W;
            SF_DMA_ALL(rs, org, interior, y0q, x0q, cb ^ 1);
            const unsigned fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
            if constexpr (PAR == 0) w_request(half_a ^ 1, 0);
            row_setup(PAR);                                         // epilogue row PAR of the previous tile (nothing pending: empty resources)

            auto substep = [&](auto sc) __attribute__((always_inline)) {
                constexpr int S = decltype(sc)::value;
                constexpr int g = S / 3, ky = S % 3;
                // --- the slice of non-MFMA work that rides on this sub-step
                if constexpr (FUSE) {
                    if constexpr (S < 4) quarter_request(S);
                    if constexpr (S == 8) quarter_finish(PAR, 0);
                    if constexpr (S == 10) quarter_finish(PAR, 1);
                    if constexpr (S == 12) quarter_finish(PAR, 2);
                    if constexpr (S == 14) quarter_finish(PAR, 3);
                } else {
                    if constexpr (S == 2) quarter_finish(PAR, 0);
                    if constexpr (S == 5) quarter_finish(PAR, 1);
                    if constexpr (S == 8) quarter_finish(PAR, 2);
                    if constexpr (S == 11) quarter_finish(PAR, 3);
                }
                if constexpr (ky == 0) {
                    if constexpr (g == 1 && PAR == 1) w_write(2);   // tap 2 of this unit's half (requested in unit A's group 5; free since its closing barrier)
                    if constexpr (g == 2) SF_BARRIER();             // b0: column tap 0 consumed (unit A) / tap 2 complete (unit B)
                    if constexpr (g == 3 && PAR == 0) {
                        w_write(0);
                        w_request(half_a ^ 1, 1);
                        const int kn = min(kt + 1, nt_tiles - 1);   // decode the next tile (past the end: this one again - a harmless re-read)
                        SF_TILE(kn, n_item, n_y0, n_x0);
                    }
                    if constexpr (g == 4 && PAR == 0) SF_BARRIER(); // b1: column tap 1 consumed
                    if constexpr (g == 5 && PAR == 0) {
                        w_write(1);
                        w_request(half_a ^ 1, 2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // --- operands of the next sub-step
                if constexpr (S < 17) {
                    constexpr int S1 = S + 1, g1 = S1 / 3, ky1 = S1 % 3;
                    Wv[S1 & 1][0] = SF_WT(g1, ky1, 0);
                    Wv[S1 & 1][1] = SF_WT(g1, ky1, 1);
                    if constexpr (ky1 == 0) {
                        X[0][0] = SF_PX(g1, 0, 0);
                        X[0][1] = SF_PX(g1, 0, 1);
                        X[1][0] = SF_PX(g1, 1, 0);
                        X[1][1] = SF_PX(g1, 1, 1);
                    } else {
                        X[ky1 + 1][0] = SF_PX(g1, ky1 + 1, 0);
                        X[ky1 + 1][1] = SF_PX(g1, ky1 + 1, 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // --- 6 MFMAs: row tap ky of both output rows (A = pixels, B = weights)
                const sfh8 wh = Wv[S & 1][0], wo = Wv[S & 1][1];
                if constexpr (PAR == 0 && S == 0) {                 // a tile's first products: C = bias / 0 (no registers to clear)
                    f32x16 bias16, zero;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        bias16[r] = bias_l;
                        zero[r] = 0.f;
                    }
                    accm[0] = sf_mfma(X[ky][0], wh, bias16);
                    accm[1] = sf_mfma(X[ky + 1][0], wh, bias16);
                    accc[0] = sf_mfma(X[ky][0], wo, zero);
                    accc[1] = sf_mfma(X[ky + 1][0], wo, zero);
                } else {
                    accm[0] = sf_mfma(X[ky][0], wh, accm[0]);
                    accm[1] = sf_mfma(X[ky + 1][0], wh, accm[1]);
                    accc[0] = sf_mfma(X[ky][0], wo, accc[0]);
                    accc[1] = sf_mfma(X[ky + 1][0], wo, accc[1]);
                }
                accc[0] = sf_mfma(X[ky][1], wh, accc[0]);
                accc[1] = sf_mfma(X[ky + 1][1], wh, accc[1]);
                __builtin_amdgcn_sched_barrier(0);
            };
            substep(std::integral_constant<int, 0>{});
            substep(std::integral_constant<int, 1>{});
            substep(std::integral_constant<int, 2>{});
            substep(std::integral_constant<int, 3>{});
            substep(std::integral_constant<int, 4>{});
            substep(std::integral_constant<int, 5>{});
            substep(std::integral_constant<int, 6>{});
            substep(std::integral_constant<int, 7>{});
            substep(std::integral_constant<int, 8>{});
            substep(std::integral_constant<int, 9>{});
            substep(std::integral_constant<int, 10>{});
            substep(std::integral_constant<int, 11>{});
            substep(std::integral_constant<int, 12>{});
            substep(std::integral_constant<int, 13>{});
            substep(std::integral_constant<int, 14>{});
            substep(std::integral_constant<int, 15>{});
            substep(std::integral_constant<int, 16>{});
            substep(std::integral_constant<int, 17>{});
#undef SF_PX
#undef SF_WT
            if constexpr (PAR == 1) {                               // the tile is complete: fold the cross terms in, hand it to the epilogue
#pragma unroll
                for (int n = 0; n < 2; ++n) accp[n] = accm[n] + accc[n] * SF_ISCALE;
                ex0p = c_x0;
                ey0p = c_y0;
                eitemp = c_item;
                c_item = n_item;
                c_y0 = n_y0;
                c_x0 = n_x0;
                pending = true;
            }
            asm volatile("" ::"v"(fence));                          // the next unit's halo has landed (fence load: see the header)
            SF_BARRIER();                                           // b2: this unit's buffer is free, the next unit's is complete
        };
        unit(std::integral_constant<int, 0>{});
        unit(std::integral_constant<int, 1>{});
    }

    // ---- the last tile: both rows ----------------------------------------------------------------------
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        row_setup(n);
#pragma unroll
        for (int q = 0; q < 4; ++q) quarter_request(q);
#pragma unroll
        for (int q = 0; q < 4; ++q) quarter_finish(n, q);
    }
#undef SF_DMA_ALL
#undef SF_HALF
#undef SF_TILE
#undef SF_BARRIER
}

// ------------------------------------------------------------------------------------------------------------------------
// The WHOLE of conv2_i in one launch (reference model/pfnl.py:69-71 with the exact shared-base split of DESIGN.md section 3):
// a chain = the T + 1 tiles of one (clip, spatial tile): first the shared half (input `base`, kernel rows 0..63), whose result
// never leaves the registers - folded with the bias it becomes the INITIAL C of the T per-frame tiles (input inp1_t, kernel rows
// 64..127) that follow, so the addend costs neither a launch, nor its HBM round trip (write + T reads per tile), nor an add.
// Against conv3x3_sf_kernel<0> + <1>: the weight slices travel by LDS-DMA as well (L2 -> LDS, no registers: a chain switches
// packs twice, and every unit may now bring the next unit's weights - the same three slices following the column taps consumed -
// skipped when the next unit uses what is already there), each slice covered by its own fence load.
#ifdef PFNL_SFC_TIMING
__device__ long long sfc_dbg[256 * 2 * 160];
#endif
// SFCOPY (round 6): the block's output - the next block's inp0 - is ALSO written in the split format (p.out2, [items][H][W] x 256 B), so
// that conv3x3_c1c10_kernel can take its halo by LDS-DMA in operand form instead of splitting fp32 on the VALU at every commit.  A lane
// owns one channel (D[pixel][cout]) where an SF chunk is 8 channels of one pixel: neighbouring lanes swap one binary16 pair per pixel
// (DPP quad_perm), the even lane then stores the hi halves of channels (c, c + 1), the odd lane their lo' halves - 32 lanes x 4 B = the
// 128 bytes [hi 64 B | lo' 64 B] of a (pixel, channel half): the same line-per-instruction pattern as the fp32 stores.
// SPLIT: split chains (below) - a template parameter so that the whole-round launches keep their instruction stream (conv_split16.hip).
template <bool SFCOPY, bool SPLIT>
__global__ __launch_bounds__(SF_THREADS, 1) void conv3x3_sf_chain_kernel(ConvSplitParams p) {
#ifdef PFNL_SFC_TIMING
    int dbg_n = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
    unsigned char* const wl = sf_smem + 2 * SF_TILE_BYTES;
    float* const bl = reinterpret_cast<float*>(sf_smem + 2 * SF_TILE_BYTES + SF_W_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if PFNL_S16_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(PFNL_S16_PRIO);
#endif
    const int rp = wave >> 1;
    const int nt = wave & 1;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + SF_TW - 1) / SF_TW, tiles_y = (H + SF_TH - 1) / SF_TH;
    const int per_item = tiles_x * tiles_y;
    const int item_bytes = H * W * 256;
    const int wbytes = W * 256;
    const int T = p.add_div, gT = T + 1;                            // tiles of a chain: the shared half, then the T frames
    const int nchains = per_item * (p.items / T);
    const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3, cpx = gridDim.x >> 3;
    // SPLIT CHAINS (round 6; p.split_s > 0; conv_split16.h): whole chains for the first p.n_full, one PART per workgroup of each chain behind
    // them - the shared half (recomputed per part: one tile in 1 + frames) and the part's frames [sp_f0, sp_f1)
    const int n_full = SPLIT ? p.n_full : nchains;
    const int per_xcd = (n_full + 7) >> 3;
    const int cbeg = xcd * per_xcd;
    const int ccnt = min(per_xcd, n_full - cbeg);
    if (!SPLIT && xj >= ccnt) return;
    const int nfull_tiles = ((!SPLIT || xj < ccnt) ? (ccnt - xj + cpx - 1) / cpx : 0) * gT;
    [[maybe_unused]] const int slot = xcd * cpx + xj;
    const bool has_part = SPLIT && slot < (nchains - n_full) * p.split_s;
    const int sp_chain = has_part ? n_full + slot / p.split_s : 0;
    const int sp_f0 = has_part ? (slot % p.split_s) * p.split_q : 0, sp_f1 = has_part ? min(T, sp_f0 + p.split_q) : 0;
    const int nt_tiles = nfull_tiles + (has_part ? 1 + sp_f1 - sp_f0 : 0);
    if (SPLIT && nt_tiles <= 0) return;
    // tile k -> (f = position in the chain: 0 = shared half, 1 .. T = frame f - 1; clip, y0, x0)
#define SFC_TILE(k_, f_, clip_, y0_, x0_)                                                        \
    do {                                                                                         \
        int ch_;                                                                                 \
        if (!SPLIT || (k_) < nfull_tiles) {                                                      \
            const int ci_ = (k_) / gT;                                                           \
            f_ = (k_) - ci_ * gT;                                                                \
            ch_ = cbeg + xj + ci_ * cpx;                                                         \
        } else {                                                                                 \
            const int kk_ = (k_) - nfull_tiles;                                                  \
            f_ = kk_ == 0 ? 0 : sp_f0 + kk_;                                                     \
            ch_ = sp_chain;                                                                      \
        }                                                                                        \
        clip_ = ch_ / per_item;                                                                  \
        const int sp_ = ch_ - clip_ * per_item;                                                  \
        const int ty_ = sp_ / tiles_x;                                                           \
        y0_ = ty_ * SF_TH;                                                                       \
        x0_ = (sp_ - ty_ * tiles_x) * SF_TW;                                                     \
    } while (0)
#define SFC_HALF(u_) ((((u_) >> 1) ^ (u_)) & 1)

    const float bias_r = tid < 64 ? p.bias[tid] : 0.f;
    int dpk[SF_DMA_ITERS];                                          // py | px << 8 of the lane's halo pixel per DMA instruction (the source offset is
#pragma unroll                                                      // recomputed from it: this kernel has no registers to spare)
    for (int k = 0; k < SF_DMA_ITERS; ++k) {
        const int pix = 8 * (wave + 8 * k) + (lane >> 3);
        const int py = pix / SF_IW, px = pix - py * SF_IW;
        dpk[k] = py | (px << 8);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)sf_smem;
    const unsigned ldsw = lds0 + 2 * SF_TILE_BYTES;                 // LDS byte address of the weights
#define SFC_DMA_HALO(rs_, org_, interior_, y0_, x0_, buf_)                                       \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < SF_DMA_ITERS; ++k_) {                            \
            const int i_ = wave + 8 * k_;                                                        \
            if (k_ < SF_DMA_ITERS - 1 || i_ < SF_NDMA) {                                         \
                const int py_ = dpk[k_] & 0xff, px_ = dpk[k_] >> 8;                              \
                const int gy_ = (y0_) + py_ - 1, gx_ = (x0_) + px_ - 1;                          \
                const bool in_ = (interior_) || ((unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W && py_ < SF_IH); \
                const int rel_ = py_ * wbytes + px_ * 256 + (((lane & 7) ^ ((px_ >> 1) & 7)) << 4); \
                sf_dma16(rs_, lds0 + (buf_) * SF_TILE_BYTES + i_ * 1024, in_ ? (org_) + rel_ : 0x7fffffff); \
            }                                                                                    \
        }                                                                                        \
    } while (0)
#define SFC_DMA_HALO_PIECE(k_, rs_, org_, interior_, y0_, x0_, buf_)                             \
    do {                                                                                         \
        const int i_ = wave + 8 * (k_);                                                          \
        if ((k_) < SF_DMA_ITERS - 1 || i_ < SF_NDMA) {                                           \
            const int py_ = dpk[k_] & 0xff, px_ = dpk[k_] >> 8;                                  \
            const int gy_ = (y0_) + py_ - 1, gx_ = (x0_) + px_ - 1;                              \
            const bool in_ = (interior_) || ((unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W && py_ < SF_IH); \
            const int rel_ = py_ * wbytes + px_ * 256 + (((lane & 7) ^ ((px_ >> 1) & 7)) << 4);   \
            sf_dma16(rs_, lds0 + (buf_) * SF_TILE_BYTES + i_ * 1024, in_ ? (org_) + rel_ : 0x7fffffff); \
        }                                                                                        \
    } while (0)
    // one 24 KB weight slice (column tap `slot` of pack `pk_`, channel half `half_`): 24 DMA instructions, 3 per wave
    const int wvoff = wave * 1024 + lane * 16;
#define SFC_DMA_W(pk_, half_, slot_)                                                             \
    do {                                                                                         \
        const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc(                    \
            const_cast<uint16_t*>((pk_) ? p.wpack : p.wpack2), 0, 2 * SF_W_BYTES, 0x00020000);   \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_)                                         \
            sf_dma16(rw_, ldsw + (slot_) * SF_SLOT_BYTES + (wave + 8 * k_) * 1024, (half_) * SF_W_BYTES + (slot_) * SF_SLOT_BYTES + wvoff + k_ * 8192); \
    } while (0)

#ifdef PFNL_X_NOWSTREAM   /* timing experiment only (wrong results on purpose), as in conv_split16.hip */
#define SFC_DMA_WX(pk_, half_, slot_) do {} while (0)
#else
#define SFC_DMA_WX(pk_, half_, slot_) SFC_DMA_W(pk_, half_, slot_)
#endif

    int paddr[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = (lane & 31) + kx;
        paddr[kx] = ((2 * rp) * SF_IW + col) * 128 + ((((lane >> 5)) ^ ((col >> 1) & 7)) << 4);
    }
    const int lo_xor = 4 << 4;
    const unsigned char* const wlane = wl + nt * 2048 + lane * 16;
    const int ech = 32 * nt + (lane & 31);
    f32x16 accm[2], accc[2], accp[2], pbv[2];                       // pbv: shared half + bias of the chain = initial C of its frames
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accm[n][r] = 0.f;
            accc[n][r] = 0.f;
            accp[n][r] = 0.f;
            pbv[n][r] = 0.f;
        }
    int ex0p = 0, ey0p = 0, eitemp = 0;
    bool pending = false;
    const float slope = p.act ? 0.2f : 1.0f;

    __amdgpu_buffer_rsrc_t rsO;                                     // out == resid (the launcher checks): one resource per row
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsS;                    // SFCOPY: the same row of the split-format copy
    int evoff = 0;
    [[maybe_unused]] int evoff_sf = 0;
    float rres[8];                                                  // the residual of two quarters at a time
    const float nscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -2048.0f)));
    auto row_setup = [&](int n) __attribute__((always_inline)) {
        const int ey = ey0p + 2 * rp + n;
        const int nrec = (pending && ey < H) ? wbytes : 0;
        rsO = __builtin_amdgcn_make_buffer_rsrc(p.out + ((size_t)eitemp * H + ey) * W * 64, 0, nrec, 0x00020000);
        evoff = (ex0p + 4 * (lane >> 5)) * 256 + ech * 4;
        if constexpr (SFCOPY) {
            rsS = __builtin_amdgcn_make_buffer_rsrc(p.out2 + ((size_t)eitemp * H + ey) * W * 64, 0, nrec, 0x00020000);
            // even lane: hi pair of channels (c, c + 1) at [half nt][hi][c]; odd lane: the lo' pair of (c - 1, c) at [half nt][lo'][c - 1]
            evoff_sf = (ex0p + 4 * (lane >> 5)) * 256 + nt * 128 + (lane & 1) * 64 + ((lane & 31) >> 1) * 4;
        }
    };
    // the four finished values of a quarter (pixels 8 q + j of the lane's channel) -> split, pair exchange, 4 dword stores
    auto sf_copy_quarter = [&](const float (&v)[4], int q) __attribute__((always_inline)) {
        if constexpr (SFCOPY) {
            sfu2 hi, lo;
            sf_split4(f32x4{v[0], v[1], v[2], v[3]}, hi, lo, nscale);
            const bool odd = lane & 1;
            const unsigned s0 = odd ? hi.x : lo.x, s1 = odd ? hi.y : lo.y;         // what the partner lane needs: its hi (to an even lane), its lo' (to an odd one)
            const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xf, 0xf, false);   // quad_perm [1, 0, 3, 2]: lane ^ 1
            const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xf, 0xf, false);
            const unsigned a0 = odd ? r0 : hi.x, a1 = odd ? r1 : hi.y;             // channel c (even) / c - 1 (odd): the low halfword
            const unsigned b0 = odd ? lo.x : r0, b1 = odd ? lo.y : r1;             // channel c + 1 (even) / c (odd): the high halfword
            sf_store_u32<SF_STORE_AUX>(__builtin_amdgcn_perm(b0, a0, 0x05040100u), rsS, evoff_sf, q * 2048);
            sf_store_u32<SF_STORE_AUX>(__builtin_amdgcn_perm(b0, a0, 0x07060302u), rsS, evoff_sf + 256, q * 2048);
            sf_store_u32<SF_STORE_AUX>(__builtin_amdgcn_perm(b1, a1, 0x05040100u), rsS, evoff_sf + 512, q * 2048);
            sf_store_u32<SF_STORE_AUX>(__builtin_amdgcn_perm(b1, a1, 0x07060302u), rsS, evoff_sf + 768, q * 2048);
        }
    };
    auto quarter_request = [&](int q) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rres[4 * (q & 1) + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsO, evoff + j * 256, q * 2048, 0));
    };
    auto quarter_finish_with = [&](int n, int q, const float (&rv)[4]) __attribute__((always_inline)) {
        float vv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = accp[n][4 * q + j];                           // (shared half + bias are already in: initial C of the tile)
            const float sv = v * slope;
            asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(sv));
            v += rv[j];
            vv[j] = v;
            sf_store_b32<SF_STORE_AUX>(v, rsO, evoff + j * 256, q * 2048);
        }
        sf_copy_quarter(vv, q);
    };
    auto quarter_finish = [&](int n, int q) __attribute__((always_inline)) {
        float rv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rv[j] = rres[4 * (q & 1) + j];
        quarter_finish_with(n, q, rv);
    };
#define SFC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifndef SFC_SPREAD_HALO
#define SFC_SPREAD_HALO 0
#endif
#ifdef PFNL_SFC_TIMING   /* phase timeline of the chain kernel (tools/sfc_timing.py); not part of the product build */
#define SFC_STAMP() do { if (lane == 0 && (wave == 0 || wave == 5) && dbg_n < 160) sfc_dbg[(blockIdx.x * 2 + (wave != 0)) * 160 + dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define SFC_STAMP() do {} while (0)
#endif

    // ---- prologue: halo of unit 0 (the first chain's shared half: `base`) and its weights (pack 0 = shared half, channel half 0)
    int c_f, c_clip, c_y0, c_x0, n_f, n_clip, n_y0, n_x0;
    SFC_TILE(0, c_f, c_clip, c_y0, c_x0);
    n_f = c_f;
    n_clip = c_clip;
    n_y0 = c_y0;
    n_x0 = c_x0;
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in2) + (size_t)c_clip * H * W * 64, 0, item_bytes, 0x00020000);
        const int org = ((c_y0 - 1) * W + c_x0 - 1) * 256 + SFC_HALF(0) * 128;
        SFC_DMA_HALO(rs, org, false, c_y0, c_x0, 0);
        SFC_DMA_W(0, SFC_HALF(0), 0);
        SFC_DMA_W(0, SFC_HALF(0), 1);
        SFC_DMA_W(0, SFC_HALF(0), 2);
        const unsigned fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
        if (tid < 64) bl[tid] = bias_r;
        asm volatile("" ::"v"(fence));
    }
    __syncthreads();
    const float bias_l = bl[ech];
    // weights in LDS now: (pack, half) of the current unit; the slice-2 request that is still owed to the current unit
    int w_pk = 0;                                                   // pack of the current unit: 0 = shared half, 1 = per-frame half
    bool w_slice2_owed = false;                                     // slice 2 of the current unit's weights still has to be brought (its slot was busy)

    for (int kt = 0; kt < nt_tiles; ++kt) {
        const int half_a = kt & 1;
        auto unit = [&](auto par) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par)::value;
            constexpr int cb = PAR;
            const unsigned char* const tile = sf_smem + cb * SF_TILE_BYTES;
            const int half_u = PAR == 0 ? half_a : half_a ^ 1;      // channel half of this unit
            // the next unit: unit B of this tile (same pack, other half), or unit A of the next tile (same half, that tile's pack)
            const int nx_pk = PAR == 0 ? w_pk : (n_f != 0);
            const int nx_half = half_a ^ 1;
            const bool w_replace = PAR == 0 || nx_pk != w_pk;       // (wave-uniform; unit B -> next A only at the two ends of a chain)
            SFC_STAMP();                                            // 0: unit start
            sfh8 X[4][2], Wv[2][2];
#define SF_PX(g_, r_, part_) (*reinterpret_cast<const sfh8*>(tile + (paddr[(g_) >> 1] ^ (((part_) ? lo_xor : 0) | (((g_) & 1) << 5))) + (r_) * (SF_IW * 128)))
#define SF_WT(g_, ky_, part_) (*reinterpret_cast<const sfh8*>(wlane + (((g_) * 3 + (ky_)) << 12) + ((part_) << 10)))
            X[0][0] = SF_PX(0, 0, 0);
            X[0][1] = SF_PX(0, 0, 1);
            X[1][0] = SF_PX(0, 1, 0);
            X[1][1] = SF_PX(0, 1, 1);
            Wv[0][0] = SF_WT(0, 0, 0);
            Wv[0][1] = SF_WT(0, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            // slice 2 of THIS unit's weights (its slot was free only after the previous unit's closing barrier), then the next
            // unit's halo; one fence load covers both (slice 2 is first read after b0, the halo after this unit's closing barrier)
            if (w_slice2_owed) SFC_DMA_WX(w_pk, half_u, 2);
            const int q_f = PAR == 0 ? c_f : n_f, q_clip = PAR == 0 ? c_clip : n_clip, y0q = PAR == 0 ? c_y0 : n_y0, x0q = PAR == 0 ? c_x0 : n_x0;
            const float* const qsrc = q_f == 0 ? p.in2 + (size_t)q_clip * H * W * 64 : p.in + ((size_t)q_clip * T + (q_f - 1)) * H * W * 64;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qsrc), 0, item_bytes, 0x00020000);
            const int org = ((y0q - 1) * W + x0q - 1) * 256 + nx_half * 128;
            const bool interior = y0q > 0 && y0q + SF_IH - 1 <= H && x0q > 0 && x0q + SF_IW - 1 <= W;
#if SFC_SPREAD_HALO
            // the halo pieces go one per sub-step BEHIND b0 (sub-steps 6 .. 11) instead of in a burst at the unit's start, where both waves
            // of a SIMD issue their DMAs at the same time and nobody computes; slice 2 of this unit's weights keeps a fence of its own
            unsigned fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);   // (covers slice 2; re-issued behind the last halo piece)
#else
            SFC_DMA_HALO(rs, org, interior, y0q, x0q, cb ^ 1);
            const unsigned fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
#endif
            unsigned fence_w = 0;
            row_setup(PAR);

            auto substep = [&](auto sc) __attribute__((always_inline)) {
                constexpr int S = decltype(sc)::value;
                constexpr int g = S / 3, ky = S % 3;
                // the residual lines of quarters 0, 1 are requested at the unit's start and used 6 - 7 sub-steps later; those of
                // quarters 2, 3 take their registers over (requested BEFORE the stores of 0, 1: vmcnt is in order and counts stores)
                if constexpr (S == 0) quarter_request(0);
                if constexpr (S == 1) quarter_request(1);
                if constexpr (S == 7) {
                    float v0[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v0[j] = rres[j];
                    quarter_request(2);
                    quarter_finish_with(PAR, 0, v0);
                }
                if constexpr (S == 8) {
                    float v1[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v1[j] = rres[4 + j];
                    quarter_request(3);
                    quarter_finish_with(PAR, 1, v1);
                }
                if constexpr (S == 14) quarter_finish(PAR, 2);
                if constexpr (S == 15) quarter_finish(PAR, 3);
#if SFC_SPREAD_HALO
                if constexpr (S >= 7 && S < 6 + SF_DMA_ITERS) SFC_DMA_HALO_PIECE(S - 6, rs, org, interior, y0q, x0q, cb ^ 1);
                if constexpr (S == 6 + SF_DMA_ITERS) fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);   // behind the last piece
#endif
                if constexpr (ky == 0) {
                    if constexpr (g == 2) {
                        SFC_STAMP();                                // 1: groups 0-1 done
                        asm volatile("" ::"v"(fence));              // slice 2 of this unit's weights has landed (and the halo, as it happens)
                        SFC_STAMP();                                // 2: fence passed
                        SFC_BARRIER();                              // b0: column tap 0 consumed; slice 2 complete
                        SFC_STAMP();                                // 3: past b0
                        if (w_replace) SFC_DMA_WX(nx_pk, nx_half, 0);
#if SFC_SPREAD_HALO
                        SFC_DMA_HALO_PIECE(0, rs, org, interior, y0q, x0q, cb ^ 1);
#endif
                        if constexpr (PAR == 0) {                   // decode the next tile (past the end: this one again - a harmless re-read)
                            const int kn = min(kt + 1, nt_tiles - 1);
                            SFC_TILE(kn, n_f, n_clip, n_y0, n_x0);
                        }
                    }
                    if constexpr (g == 4) {
                        SFC_STAMP();                                // 4: groups 2-3 done
                        SFC_BARRIER();                              // b1: column tap 1 consumed
                        SFC_STAMP();                                // 5: past b1
                        if (w_replace) SFC_DMA_WX(nx_pk, nx_half, 1);
                        fence_w = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);   // covers slices 0 and 1 of the next unit's weights
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#if PFNL_S16_SPREAD
                // the next sub-step's operand reads between this sub-step's MFMAs (conv_split16.hip, conv3x3_c1c10_kernel)
                {
                    constexpr int S1 = S + 1, g1 = S1 / 3, ky1 = S1 % 3;
                    constexpr bool RD = S < 17;
                    constexpr bool FIRST = PAR == 0 && S == 0;   // a tile's first products: C = the chain's shared half + bias (or 0 for that half itself)
                    f32x16 zero;
#pragma unroll
                    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                    const sfh8 wh = Wv[S & 1][0], wo = Wv[S & 1][1];
                    accm[0] = sf_mfma(X[ky][0], wh, FIRST ? pbv[0] : accm[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (RD) {
                        Wv[S1 & 1][0] = SF_WT(g1, ky1, 0);
                        Wv[S1 & 1][1] = SF_WT(g1, ky1, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    accm[1] = sf_mfma(X[ky + 1][0], wh, FIRST ? pbv[1] : accm[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (RD) {
                        if constexpr (ky1 == 0) {
                            X[0][0] = SF_PX(g1, 0, 0);
                            X[0][1] = SF_PX(g1, 0, 1);
                        } else {
                            X[ky1 + 1][0] = SF_PX(g1, ky1 + 1, 0);
                            X[ky1 + 1][1] = SF_PX(g1, ky1 + 1, 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    accc[0] = sf_mfma(X[ky][0], wo, FIRST ? zero : accc[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (RD && ky1 == 0) {
                        X[1][0] = SF_PX(g1, 1, 0);
                        X[1][1] = SF_PX(g1, 1, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    accc[1] = sf_mfma(X[ky + 1][0], wo, FIRST ? zero : accc[1]);
                    accc[0] = sf_mfma(X[ky][1], wh, accc[0]);
                    accc[1] = sf_mfma(X[ky + 1][1], wh, accc[1]);
                }
#else
                if constexpr (S < 17) {
                    constexpr int S1 = S + 1, g1 = S1 / 3, ky1 = S1 % 3;
                    Wv[S1 & 1][0] = SF_WT(g1, ky1, 0);
                    Wv[S1 & 1][1] = SF_WT(g1, ky1, 1);
                    if constexpr (ky1 == 0) {
                        X[0][0] = SF_PX(g1, 0, 0);
                        X[0][1] = SF_PX(g1, 0, 1);
                        X[1][0] = SF_PX(g1, 1, 0);
                        X[1][1] = SF_PX(g1, 1, 1);
                    } else {
                        X[ky1 + 1][0] = SF_PX(g1, ky1 + 1, 0);
                        X[ky1 + 1][1] = SF_PX(g1, ky1 + 1, 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const sfh8 wh = Wv[S & 1][0], wo = Wv[S & 1][1];
                if constexpr (PAR == 0 && S == 0) {                 // a tile's first products: C = the chain's shared half + bias (or 0 for that half itself)
                    f32x16 zero;
#pragma unroll
                    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                    accm[0] = sf_mfma(X[ky][0], wh, pbv[0]);
                    accm[1] = sf_mfma(X[ky + 1][0], wh, pbv[1]);
                    accc[0] = sf_mfma(X[ky][0], wo, zero);
                    accc[1] = sf_mfma(X[ky + 1][0], wo, zero);
                } else {
                    accm[0] = sf_mfma(X[ky][0], wh, accm[0]);
                    accm[1] = sf_mfma(X[ky + 1][0], wh, accm[1]);
                    accc[0] = sf_mfma(X[ky][0], wo, accc[0]);
                    accc[1] = sf_mfma(X[ky + 1][0], wo, accc[1]);
                }
                accc[0] = sf_mfma(X[ky][1], wh, accc[0]);
                accc[1] = sf_mfma(X[ky + 1][1], wh, accc[1]);
#endif
                __builtin_amdgcn_sched_barrier(0);
            };
            substep(std::integral_constant<int, 0>{});
            substep(std::integral_constant<int, 1>{});
            substep(std::integral_constant<int, 2>{});
            substep(std::integral_constant<int, 3>{});
            substep(std::integral_constant<int, 4>{});
            substep(std::integral_constant<int, 5>{});
            substep(std::integral_constant<int, 6>{});
            substep(std::integral_constant<int, 7>{});
            substep(std::integral_constant<int, 8>{});
            substep(std::integral_constant<int, 9>{});
            substep(std::integral_constant<int, 10>{});
            substep(std::integral_constant<int, 11>{});
            substep(std::integral_constant<int, 12>{});
            substep(std::integral_constant<int, 13>{});
            substep(std::integral_constant<int, 14>{});
            substep(std::integral_constant<int, 15>{});
            substep(std::integral_constant<int, 16>{});
            substep(std::integral_constant<int, 17>{});
#undef SF_PX
#undef SF_WT
            if constexpr (PAR == 1) {
                // The tile is complete.  A frame tile is handed to the epilogue; the shared half of a chain stays in registers
                // (+ bias) as the initial C of the frames that follow, and is cleared behind the chain's last frame (the next
                // tile is a shared half again: initial C = 0).  Branch-free (selects on wave-uniform conditions): arms that
                // define 32-register vectors cost the allocator live copies of both.
                const bool head = c_f == 0, last = c_f == ((SPLIT && kt >= nfull_tiles) ? sp_f1 : T);   // (the last frame of a whole chain / of this workgroup's part)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const f32x16 fold = accm[n] + accc[n] * SF_ISCALE;
                    accp[n] = fold;
#pragma unroll
                    for (int r = 0; r < 16; ++r) pbv[n][r] = head ? fold[r] + bias_l : (last ? 0.f : pbv[n][r]);
                }
                ex0p = c_x0;
                ey0p = c_y0;
                eitemp = c_clip * T + (c_f - 1);
                pending = !head;
                c_f = n_f;
                c_clip = n_clip;
                c_y0 = n_y0;
                c_x0 = n_x0;
            }
            w_slice2_owed = w_replace;                              // slice 2 of the next unit's weights goes once this unit's is consumed: at its start
            w_pk = nx_pk;
            SFC_STAMP();                                            // 6: groups 4-5 done
            asm volatile("" ::"v"(fence), "v"(fence_w));            // the next unit's halo and weight slices 0, 1 have landed
            SFC_STAMP();                                            // 7: fences passed
            SFC_BARRIER();                                          // b2
        };
        unit(std::integral_constant<int, 0>{});
        unit(std::integral_constant<int, 1>{});
    }
    // ---- the last tile (a frame tile): both rows
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        row_setup(n);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            quarter_request(2 * h2);
            quarter_request(2 * h2 + 1);
            quarter_finish(n, 2 * h2);
            quarter_finish(n, 2 * h2 + 1);
        }
    }
#undef SFC_DMA_HALO
#undef SFC_DMA_W
#undef SFC_HALF
#undef SFC_TILE
#undef SFC_BARRIER
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// conv3x3_sf_chain16_kernel (round 6, late): what it took to get the 16x16x32 shape's energy gain through a kernel with no register to spare - each of these
// cost MORE than the shape gains, and the first version was 12 % slower than the kernel it replaces:
//   * ds_read_b128's lane groups make a 16x16x32 pixel operand collide under the 32x32x16 kernels' swizzle (SF_SWZ16 below: 4.2e6 conflict cycles -> 0);
//   * the vector-memory counter is in order: a residual value used two sub-steps after its request waits behind the unit's halo DMA (an HBM round trip) - two
//     quarters in flight for four sub-steps;
//   * ONE spilled register is a scratch load + s_waitcnt vmcnt(0), i.e. a wait for every DMA piece in flight: no spill is affordable in a kernel that keeps
//     LDS-DMA in flight (the halo-DMA coordinates are recomputed per piece, the second pixel half's offset is an immediate: SF_SWZ16(px + 16) = SF_SWZ16(px));
//   * 32 one-dword LDS reads of the bias inside the fold, each with its own full wait, were ~1 500 cycles per unit: two reads in front of the loop.
// Found by comparing instruction counts of the two kernels' ISA (s_waitcnt lgkmcnt(0): 45 against 13) - the phase-stamp build spills.  Split chains stay on the
// 32x32x16 kernel (their bookkeeping does not fit the registers).
// The same launch on v_mfma_f32_16x16x32_f16 (round 6, late; DESIGN.md R6.9: the K = 32 shape costs 14 % less energy per FLOP under the package power cap).
// A = pixels (16 pixels x 32 input channels = a unit's channel half in ONE k-step), B = weights (32 channels x 16 output channels): lane (n = l & 15, kq = l >> 4)
// reads chunk kq (hi) / 4 + kq (lo') of halo pixel 16 ph + n + kx and of the weights of output channel 16 (2 nt + ct) + n, and owns - per output row, pixel half
// ph and 16-channel output tile ct - channel 32 nt + 16 ct + n of the 4 pixels 16 ph + 4 kq .. + 3: a dword access of the epilogue is 16 lanes x 4 B = 64 contiguous
// bytes of a pixel (the roles the other way round - a lane owning 4 channels of one pixel, 16-byte accesses - measured the same).  9 sub-steps (column tap, row tap) of 24 MFMAs per unit instead of 18 of 6; weights in the order conv3x3_split16_pack_weights16 writes
// ([kx][ky][output tile][hi / lo'][lane]: the same 24 KB slices per column tap); the LDS image of the halo, the work order, the DMA requests, the three barriers of a
// unit and the fences are those of conv3x3_sf_chain_kernel.  Operand registers are refilled for the next sub-step behind the MFMAs that read them last.
// Summation order differs from the 32x32x16 form (K = 32 per instruction): equal to it within rounding, not bit for bit.
typedef float sff4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ sff4 sf_mfma16(sfh8 a, sfh8 b, sff4 c) {
#ifdef SF_X_NOMFMA   /* timing experiments only (wrong results on purpose) */
    c[0] += (float)a[0] * (float)b[0];
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// The XOR swizzle of this kernel's LDS image: chunk c of halo pixel (py, px) sits in slot c ^ SF_SWZ16(px), SF_SWZ16 = 2 ((px >> 1) & 3).  A 16x16x32 pixel
// operand is read by lane (n = l & 15, kq = l >> 4), and ds_read_b128 serves a wave in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS): a group is ALL 16 pixels of the operand, the outer eight with chunk kq and the inner eight with kq ^ 1.  With the (px >> 1) & 7 of the
// 32x32x16 kernels (whose groups are 16 consecutive pixels of ONE chunk) those collide for column taps 1 and 2: SQ_LDS_BANK_CONFLICT 4.2e6 per launch, +27 us.
// Searched over the XOR masks linear in px >> 1: this one is conflict-free for every column tap, pixel half, chunk pair and part.
#define SF_SWZ16(px_) ((((px_) >> 1) & 3) << 1)
template <bool SPLIT>
__global__ __launch_bounds__(SF_THREADS, 1) void conv3x3_sf_chain16_kernel(ConvSplitParams p) {
#ifdef PFNL_SFC_TIMING
    int dbg_n = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
    unsigned char* const wl = sf_smem + 2 * SF_TILE_BYTES;
    float* const bl = reinterpret_cast<float*>(sf_smem + 2 * SF_TILE_BYTES + SF_W_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if PFNL_S16_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(PFNL_S16_PRIO);
#endif
    const int rp = wave >> 1;
    const int nt = wave & 1;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + SF_TW - 1) / SF_TW, tiles_y = (H + SF_TH - 1) / SF_TH;
    const int per_item = tiles_x * tiles_y;
    const int item_bytes = H * W * 256;
    const int wbytes = W * 256;
    const int T = p.add_div, gT = T + 1;                            // tiles of a chain: the shared half, then the T frames
    const int nchains = per_item * (p.items / T);
    const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3, cpx = gridDim.x >> 3;
    // SPLIT CHAINS (round 6; p.split_s > 0; conv_split16.h): whole chains for the first p.n_full, one PART per workgroup of each chain behind
    // them - the shared half (recomputed per part: one tile in 1 + frames) and the part's frames [sp_f0, sp_f1)
    const int n_full = SPLIT ? p.n_full : nchains;
    const int per_xcd = (n_full + 7) >> 3;
    const int cbeg = xcd * per_xcd;
    const int ccnt = min(per_xcd, n_full - cbeg);
    if (!SPLIT && xj >= ccnt) return;
    const int nfull_tiles = ((!SPLIT || xj < ccnt) ? (ccnt - xj + cpx - 1) / cpx : 0) * gT;
    [[maybe_unused]] const int slot = xcd * cpx + xj;
    const bool has_part = SPLIT && slot < (nchains - n_full) * p.split_s;
    const int sp_chain = has_part ? n_full + slot / p.split_s : 0;
    const int sp_f0 = has_part ? (slot % p.split_s) * p.split_q : 0, sp_f1 = has_part ? min(T, sp_f0 + p.split_q) : 0;
    const int nt_tiles = nfull_tiles + (has_part ? 1 + sp_f1 - sp_f0 : 0);
    if (SPLIT && nt_tiles <= 0) return;
    // tile k -> (f = position in the chain: 0 = shared half, 1 .. T = frame f - 1; clip, y0, x0)
#define SFC_TILE(k_, f_, clip_, y0_, x0_)                                                        \
    do {                                                                                         \
        int ch_;                                                                                 \
        if (!SPLIT || (k_) < nfull_tiles) {                                                      \
            const int ci_ = (k_) / gT;                                                           \
            f_ = (k_) - ci_ * gT;                                                                \
            ch_ = cbeg + xj + ci_ * cpx;                                                         \
        } else {                                                                                 \
            const int kk_ = (k_) - nfull_tiles;                                                  \
            f_ = kk_ == 0 ? 0 : sp_f0 + kk_;                                                     \
            ch_ = sp_chain;                                                                      \
        }                                                                                        \
        clip_ = ch_ / per_item;                                                                  \
        const int sp_ = ch_ - clip_ * per_item;                                                  \
        const int ty_ = sp_ / tiles_x;                                                           \
        y0_ = ty_ * SF_TH;                                                                       \
        x0_ = (sp_ - ty_ * tiles_x) * SF_TW;                                                     \
    } while (0)
#define SFC_HALF(u_) ((((u_) >> 1) ^ (u_)) & 1)

    const float bias_r = tid < 64 ? p.bias[tid] : 0.f;
    // (py, px) of the lane's halo pixel per DMA instruction: recomputed at every request (a multiply-high and two more VALU per piece - this kernel has no
    // registers to spare for a table)
    const unsigned lds0 = (unsigned)(uintptr_t)sf_smem;
    const unsigned ldsw = lds0 + 2 * SF_TILE_BYTES;                 // LDS byte address of the weights
#define SFC_DMA_HALO(rs_, org_, interior_, y0_, x0_, buf_)                                       \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < SF_DMA_ITERS; ++k_) {                            \
            const int i_ = wave + 8 * k_;                                                        \
            if (k_ < SF_DMA_ITERS - 1 || i_ < SF_NDMA) {                                         \
                const int pix_ = 8 * (wave + 8 * k_) + (lane >> 3);                              \
                const int py_ = (pix_ * 1928) >> 16, px_ = pix_ - py_ * SF_IW;                   \
                const int gy_ = (y0_) + py_ - 1, gx_ = (x0_) + px_ - 1;                          \
                const bool in_ = (interior_) || ((unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W && py_ < SF_IH); \
                const int rel_ = py_ * wbytes + px_ * 256 + (((lane & 7) ^ SF_SWZ16(px_)) << 4);    \
                sf_dma16(rs_, lds0 + (buf_) * SF_TILE_BYTES + i_ * 1024, in_ ? (org_) + rel_ : 0x7fffffff); \
            }                                                                                    \
        }                                                                                        \
    } while (0)
    // one 24 KB weight slice (column tap `slot` of pack `pk_`, channel half `half_`): 24 DMA instructions, 3 per wave
    const int wvoff = wave * 1024 + lane * 16;
#define SFC_DMA_W(pk_, half_, slot_)                                                             \
    do {                                                                                         \
        const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc(                    \
            const_cast<uint16_t*>((pk_) ? p.wpack : p.wpack2), 0, 2 * SF_W_BYTES, 0x00020000);   \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_)                                         \
            sf_dma16(rw_, ldsw + (slot_) * SF_SLOT_BYTES + (wave + 8 * k_) * 1024, (half_) * SF_W_BYTES + (slot_) * SF_SLOT_BYTES + wvoff + k_ * 8192); \
    } while (0)

#ifdef PFNL_X_NOWSTREAM   /* timing experiment only (wrong results on purpose), as in conv_split16.hip */
#define SFC_DMA_WX(pk_, half_, slot_) do {} while (0)
#else
#define SFC_DMA_WX(pk_, half_, slot_) SFC_DMA_W(pk_, half_, slot_)
#endif

    int paddr[3];                                                   // [column tap]: pixel half 0; half 1 is 16 pixels = 2 048 bytes further (SF_SWZ16(px + 16) = SF_SWZ16(px)): an immediate
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = (lane & 15) + kx;
        paddr[kx] = ((2 * rp) * SF_IW + col) * 128 + (((lane >> 4) ^ SF_SWZ16(col)) << 4);
    }
    const int lo_xor = 4 << 4;
    const unsigned char* const wlane = wl + nt * 4096 + lane * 16;  // [kx][ky][output tile 2 nt + ct][hi / lo'][lane] x 16 B
    const int ech = 32 * nt + (lane & 15);                          // the lane's channels: ech + 16 ct (D[pixel][channel]: rows 4 (lane >> 4) + r = pixels)
    sff4 accm[2][4], accc[2][4], accp[2][4], pbv[2][4];             // [output row][2 ph + ct]; pbv: shared half + bias of the chain = initial C of its frames
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            accm[n][q] = sff4{0.f, 0.f, 0.f, 0.f};
            accc[n][q] = sff4{0.f, 0.f, 0.f, 0.f};
            accp[n][q] = sff4{0.f, 0.f, 0.f, 0.f};
            pbv[n][q] = sff4{0.f, 0.f, 0.f, 0.f};
        }
    int ex0p = 0, ey0p = 0, eitemp = 0;
    bool pending = false;
    const float slope = p.act ? 0.2f : 1.0f;

    __amdgpu_buffer_rsrc_t rsO;                                     // out == resid (the launcher checks): one resource per row
    int evoff = 0;
    sff4 rres[2];                                                   // the residual values of two quarters in flight (an HBM load needs ~4 sub-steps: requested at S, used at S + 4)
    auto row_setup = [&](int n) __attribute__((always_inline)) {
        const int ey = ey0p + 2 * rp + n;
        const int nrec = (pending && ey < H) ? wbytes : 0;
        rsO = __builtin_amdgcn_make_buffer_rsrc(p.out + ((size_t)eitemp * H + ey) * W * 64, 0, nrec, 0x00020000);
        evoff = (ex0p + 4 * (lane >> 4)) * 256 + ech * 4;
    };
    // quarter q = 2 ph + ct of a row: channel ech + 16 ct of the 4 pixels 16 ph + 4 (lane >> 4) + j - 16 lanes x 4 B = 64 contiguous bytes per pixel and access
    auto quarter_request = [&](int q) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rres[q & 1][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsO, evoff + j * 256, (q >> 1) * 4096 + (q & 1) * 64, 0));
    };
    auto quarter_finish_with = [&](int n, int q, const sff4 rv) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = accp[n][q][j];                                // (shared half + bias are already in: initial C of the tile)
            const float sv = v * slope;
            asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(sv));
            sf_store_b32<SF_STORE_AUX>(v + rv[j], rsO, evoff + j * 256, (q >> 1) * 4096 + (q & 1) * 64);
        }
    };
    auto quarter_finish = [&](int n, int q) __attribute__((always_inline)) { quarter_finish_with(n, q, rres[q & 1]); };
#define SFC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifndef SFC_SPREAD_HALO
#define SFC_SPREAD_HALO 0
#endif
#ifdef PFNL_SFC_TIMING   /* phase timeline of the chain kernel (tools/sfc_timing.py); not part of the product build */
#define SFC_STAMP() do { if (lane == 0 && (wave == 0 || wave == 5) && dbg_n < 160) sfc_dbg[(blockIdx.x * 2 + (wave != 0)) * 160 + dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define SFC_STAMP() do {} while (0)
#endif

    // ---- prologue: halo of unit 0 (the first chain's shared half: `base`) and its weights (pack 0 = shared half, channel half 0)
    int c_f, c_clip, c_y0, c_x0, n_f, n_clip, n_y0, n_x0;
    SFC_TILE(0, c_f, c_clip, c_y0, c_x0);
    n_f = c_f;
    n_clip = c_clip;
    n_y0 = c_y0;
    n_x0 = c_x0;
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in2) + (size_t)c_clip * H * W * 64, 0, item_bytes, 0x00020000);
        const int org = ((c_y0 - 1) * W + c_x0 - 1) * 256 + SFC_HALF(0) * 128;
        SFC_DMA_HALO(rs, org, false, c_y0, c_x0, 0);
        SFC_DMA_W(0, SFC_HALF(0), 0);
        SFC_DMA_W(0, SFC_HALF(0), 1);
        SFC_DMA_W(0, SFC_HALF(0), 2);
        const unsigned fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
        if (tid < 64) bl[tid] = bias_r;
        asm volatile("" ::"v"(fence));
    }
    __syncthreads();

    // weights in LDS now: (pack, half) of the current unit; the slice-2 request that is still owed to the current unit
    int w_pk = 0;                                                   // pack of the current unit: 0 = shared half, 1 = per-frame half
    bool w_slice2_owed = false;                                     // slice 2 of the current unit's weights still has to be brought (its slot was busy)

    row_setup(0);                                                   // (nothing pending yet: an empty resource, zeros)
    quarter_request(0);
    quarter_request(1);
    for (int kt = 0; kt < nt_tiles; ++kt) {
        const int half_a = kt & 1;
        auto unit = [&](auto par) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par)::value;
            constexpr int cb = PAR;
            const unsigned char* const tile = sf_smem + cb * SF_TILE_BYTES;
            const int half_u = PAR == 0 ? half_a : half_a ^ 1;      // channel half of this unit
            // the next unit: unit B of this tile (same pack, other half), or unit A of the next tile (same half, that tile's pack)
            const int nx_pk = PAR == 0 ? w_pk : (n_f != 0);
            const int nx_half = half_a ^ 1;
            const bool w_replace = PAR == 0 || nx_pk != w_pk;       // (wave-uniform; unit B -> next A only at the two ends of a chain)
            SFC_STAMP();                                            // 0: unit start
            // operands: X[halo row of the column tap][pixel half][hi / lo'], Wv[row tap][output tile][hi / lo'] - never more than ~16 of the 28 live: every
            // register set is refilled for the next sub-step behind the MFMAs that read it last
            sfh8 X[4][2][2], Wv[3][2][2];
#ifdef SF_X_PXLIN   /* experiment: linear (conflict-free) pixel operand reads - wrong results on purpose: where do the bank conflicts come from? */
#define SF_PX16(kx_, ph_, r_, part_) (*reinterpret_cast<const sfh8*>(tile + lane * 16 + ((kx_) * 16 + (ph_) * 8 + (r_) * 2 + (part_)) * 1024))
#else
#define SF_PX16(kx_, ph_, r_, part_) (*reinterpret_cast<const sfh8*>(tile + (paddr[kx_] ^ ((part_) ? lo_xor : 0)) + (r_) * (SF_IW * 128) + (ph_) * 2048))
#endif
#define SF_WT16(kx_, ky_, ct_, part_) (*reinterpret_cast<const sfh8*>(wlane + (((kx_) * 3 + (ky_)) << 13) + (((ct_) * 2 + (part_)) << 10)))
            // (in the order the first sub-step uses them: LDS returns are in order, so its first MFMA waits for two reads, not for twelve)
            Wv[0][0][0] = SF_WT16(0, 0, 0, 0);
            X[0][0][0] = SF_PX16(0, 0, 0, 0);
            Wv[0][1][0] = SF_WT16(0, 0, 1, 0);
            Wv[0][0][1] = SF_WT16(0, 0, 0, 1);
            Wv[0][1][1] = SF_WT16(0, 0, 1, 1);
            X[0][0][1] = SF_PX16(0, 0, 0, 1);
            X[0][1][0] = SF_PX16(0, 1, 0, 0);
            X[0][1][1] = SF_PX16(0, 1, 0, 1);
            X[1][0][0] = SF_PX16(0, 0, 1, 0);
            X[1][1][0] = SF_PX16(0, 1, 1, 0);
            X[1][0][1] = SF_PX16(0, 0, 1, 1);
            X[1][1][1] = SF_PX16(0, 1, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            // slice 2 of THIS unit's weights (its slot was free only after the previous unit's closing barrier), then the next
            // unit's halo; one fence load covers both (slice 2 is first read after b0, the halo after this unit's closing barrier)
            if (w_slice2_owed) SFC_DMA_WX(w_pk, half_u, 2);
            const int q_f = PAR == 0 ? c_f : n_f, q_clip = PAR == 0 ? c_clip : n_clip, y0q = PAR == 0 ? c_y0 : n_y0, x0q = PAR == 0 ? c_x0 : n_x0;
            const float* const qsrc = q_f == 0 ? p.in2 + (size_t)q_clip * H * W * 64 : p.in + ((size_t)q_clip * T + (q_f - 1)) * H * W * 64;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qsrc), 0, item_bytes, 0x00020000);
            const int org = ((y0q - 1) * W + x0q - 1) * 256 + nx_half * 128;
            const bool interior = y0q > 0 && y0q + SF_IH - 1 <= H && x0q > 0 && x0q + SF_IW - 1 <= W;
            SFC_DMA_HALO(rs, org, interior, y0q, x0q, cb ^ 1);
            const unsigned fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
            unsigned fence_w = 0;

            auto substep = [&](auto sc) __attribute__((always_inline)) {
                constexpr int S = decltype(sc)::value;              // 3 kx + ky
                constexpr int kx = S / 3, ky = S % 3;
                // the residual values of quarters 0, 1 are requested at the unit's start and used 4 - 5 sub-steps (~1 500 cycles: an HBM round trip with the unit's halo
                // pieces in front of it in the in-order counter) later; those of quarters 2, 3 take their registers over (requested BEFORE the stores of 0, 1)
                // (quarters 0, 1 of this unit's row were requested at the END of the previous unit - row_setup + two requests in front of its closing fences: the
                // wait at the barrier and the unit's start are flight time - so the last sub-step carries no epilogue work in front of the closing barrier)
                if constexpr (S == 3 || S == 4) {
                    const sff4 v = rres[(S - 3) & 1];
                    quarter_request(S - 1);
                    quarter_finish_with(PAR, S - 3, v);
                }
                if constexpr (S == 6) quarter_finish(PAR, 2);
                if constexpr (S == 7) quarter_finish(PAR, 3);
                if constexpr (ky == 0) {
                    if constexpr (kx == 1) {
                        SFC_STAMP();                                // 1: column tap 0 done
                        asm volatile("" ::"v"(fence));              // slice 2 of this unit's weights has landed (and the halo, as it happens)
                        SFC_STAMP();                                // 2: fence passed
                        SFC_BARRIER();                              // b0: column tap 0 consumed; slice 2 complete
                        SFC_STAMP();                                // 3: past b0
                        if (w_replace) SFC_DMA_WX(nx_pk, nx_half, 0);
                        if constexpr (PAR == 0) {                   // decode the next tile (past the end: this one again - a harmless re-read)
                            const int kn = min(kt + 1, nt_tiles - 1);
                            SFC_TILE(kn, n_f, n_clip, n_y0, n_x0);
                        }
                    }
                    if constexpr (kx == 2) {
                        SFC_STAMP();                                // 4: column tap 1 done
                        SFC_BARRIER();                              // b1: column tap 1 consumed
                        SFC_STAMP();                                // 5: past b1
                        if (w_replace) SFC_DMA_WX(nx_pk, nx_half, 1);
                        fence_w = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);   // covers slices 0 and 1 of the next unit's weights
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                constexpr bool FIRST = PAR == 0 && S == 0;          // a tile's first products: C = the chain's shared half + bias (or 0 for that half itself)
                constexpr bool NX = S < 8;                          // there is a next sub-step in this unit: (kx1, ky1)
                constexpr int kx1 = (S + 1) / 3, ky1 = (S + 1) % 3;
                const sff4 zero4 = sff4{0.f, 0.f, 0.f, 0.f};
                // the three products of output row n_, output tile ct_, pixel half ph_: main += Wh Xh; cross += Wl Xh; cross += Wh Xl
#define SF_MM(n_, ct_, ph_) do { accm[n_][2 * (ph_) + (ct_)] = sf_mfma16(X[ky + (n_)][ph_][0], Wv[ky][ct_][0], FIRST ? pbv[n_][2 * (ph_) + (ct_)] : accm[n_][2 * (ph_) + (ct_)]); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SF_CA(n_, ct_, ph_) do { accc[n_][2 * (ph_) + (ct_)] = sf_mfma16(X[ky + (n_)][ph_][0], Wv[ky][ct_][1], FIRST ? zero4 : accc[n_][2 * (ph_) + (ct_)]); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SF_CB(n_, ct_, ph_) do { accc[n_][2 * (ph_) + (ct_)] = sf_mfma16(X[ky + (n_)][ph_][1], Wv[ky][ct_][0], accc[n_][2 * (ph_) + (ct_)]); __builtin_amdgcn_sched_barrier(0); } while (0)
                // refills for the next sub-step (one LDS read each, pinned between two MFMAs)
#define SF_LW(ct_, part_) do { if constexpr (NX) Wv[ky1][ct_][part_] = SF_WT16(kx1, ky1, ct_, part_); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SF_LX(r_, ph_, part_) do { if constexpr (NX) X[r_][ph_][part_] = SF_PX16(kx1, ph_, r_, part_); __builtin_amdgcn_sched_barrier(0); } while (0)
                // output row 0 (halo row ky), pixel half by pixel half; output row 1 (halo row ky + 1), output tile by output tile.  A register set is refilled
                // two or three MFMAs behind its last reader (an LDS return into a register an MFMA has just read costs wait states on a 16-cycle instruction):
                // the next sub-step's weights go where halo row ky was, the next halo row (or rows 0 / 1 of the next column tap) where this sub-step's weights were
                SF_MM(0, 0, 0); SF_MM(0, 1, 0); SF_CA(0, 0, 0); SF_CA(0, 1, 0); SF_CB(0, 0, 0); SF_CB(0, 1, 0);
                SF_MM(0, 0, 1); SF_LW(0, 0);
                SF_MM(0, 1, 1); SF_LW(0, 1);
                SF_CA(0, 0, 1); SF_CA(0, 1, 1); SF_CB(0, 0, 1); SF_CB(0, 1, 1);
                SF_MM(1, 0, 0); SF_LW(1, 0);
                SF_MM(1, 0, 1); SF_CA(1, 0, 0); SF_LW(1, 1);
                SF_CA(1, 0, 1); SF_CB(1, 0, 0); SF_CB(1, 0, 1);
                SF_MM(1, 1, 0); SF_LX(ky1 != 0 ? ky + 2 : 0, 0, 0);
                SF_MM(1, 1, 1); SF_CA(1, 1, 0); SF_LX(ky1 != 0 ? ky + 2 : 0, 0, 1);
                SF_CA(1, 1, 1); SF_CB(1, 1, 0); SF_CB(1, 1, 1);
                SF_LX(ky1 != 0 ? ky + 2 : 0, 1, 0); SF_LX(ky1 != 0 ? ky + 2 : 0, 1, 1);
                if constexpr (ky1 == 0) { SF_LX(1, 0, 0); SF_LX(1, 0, 1); SF_LX(1, 1, 0); SF_LX(1, 1, 1); }
#undef SF_MM
#undef SF_CA
#undef SF_CB
#undef SF_LW
#undef SF_LX
                __builtin_amdgcn_sched_barrier(0);
            };
            substep(std::integral_constant<int, 0>{});
            substep(std::integral_constant<int, 1>{});
            substep(std::integral_constant<int, 2>{});
            substep(std::integral_constant<int, 3>{});
            substep(std::integral_constant<int, 4>{});
            substep(std::integral_constant<int, 5>{});
            substep(std::integral_constant<int, 6>{});
            substep(std::integral_constant<int, 7>{});
            substep(std::integral_constant<int, 8>{});
#undef SF_PX16
#undef SF_WT16
            if constexpr (PAR == 1) {
                // The tile is complete.  A frame tile is handed to the epilogue; the shared half of a chain stays in registers
                // (+ bias) as the initial C of the frames that follow, and is cleared behind the chain's last frame (the next
                // tile is a shared half again: initial C = 0).  Branch-free (selects on wave-uniform conditions): arms that
                // define 32-register vectors cost the allocator live copies of both.
                const bool head = c_f == 0, last = c_f == ((SPLIT && kt >= nfull_tiles) ? sp_f1 : T);   // (the last frame of a whole chain / of this workgroup's part)
                // (the lane's two bias values: read from LDS HERE, once per tile and in front of the loop - as 32 reads inside it, each with its own wait, they
                // cost the kernel ~1 500 cycles per unit)
                const float bias_q[2] = {bl[ech], bl[ech + 16]};
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const sff4 fold = accm[n][q] + accc[n][q] * SF_ISCALE;
                        accp[n][q] = fold;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pbv[n][q][r] = head ? fold[r] + bias_q[q & 1] : (last ? 0.f : pbv[n][q][r]);
                    }
                ex0p = c_x0;
                ey0p = c_y0;
                eitemp = c_clip * T + (c_f - 1);
                pending = !head;
                c_f = n_f;
                c_clip = n_clip;
                c_y0 = n_y0;
                c_x0 = n_x0;
            }
            w_slice2_owed = w_replace;                              // slice 2 of the next unit's weights goes once this unit's is consumed: at its start
            w_pk = nx_pk;
            row_setup(PAR ^ 1);                                     // the NEXT unit's epilogue row (unit A: row 0 of the tile that has just been folded; unit B: row 1)
            quarter_request(0);
            quarter_request(1);
            SFC_STAMP();                                            // 6: groups 4-5 done
            asm volatile("" ::"v"(fence), "v"(fence_w));            // the next unit's halo and weight slices 0, 1 have landed
            SFC_STAMP();                                            // 7: fences passed
            SFC_BARRIER();                                          // b2
        };
        unit(std::integral_constant<int, 0>{});
        unit(std::integral_constant<int, 1>{});
    }
    // ---- the last tile (a frame tile): both rows
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        row_setup(n);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            quarter_request(q);
            quarter_finish(n, q);
        }
    }
#undef SFC_DMA_HALO
#undef SFC_DMA_W
#undef SFC_HALF
#undef SFC_TILE
#undef SFC_BARRIER
}

hipError_t launch_conv3x3_sf_chain(const ConvSplitParams& p, hipStream_t s) {
    if (!p.in || !p.in2 || !p.wpack || !p.wpack2 || !p.bias || !p.out || !p.resid || p.items < 1 || p.H < 1 || p.W < 1 || p.accum || p.out_sf)
        return hipErrorInvalidValue;
    if (p.add_div < 1 || p.items % p.add_div || p.out != p.resid) return hipErrorInvalidValue;   // in place: out += ...
    if ((long long)p.H * p.W * 256 >= 0x7fffffffLL) return hipErrorInvalidValue;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    const int ncu = device_cu_count();
    if (!ncu) return hipErrorUnknown;
    const int grid = ncu >= 8 ? ncu / 8 * 8 : 8;
    if (p.split_s) {                                                       // (the geometry conv3x3_c1c10_kernel was launched with: conv_split16.hip checks it in full)
        const long long nch = (long long)((p.W + SF_TW - 1) / SF_TW) * ((p.H + SF_TH - 1) / SF_TH) * (p.items / p.add_div);
        if (p.split_s < 2 || p.split_q < 1 || p.n_full % grid || p.n_full >= nch || (nch - p.n_full) * p.split_s > grid ||
            (long long)(p.split_s - 1) * p.split_q >= p.add_div || (long long)p.split_s * p.split_q < p.add_div)
            return hipErrorInvalidValue;
    }
    if ((p.wpack_m16 == nullptr) != (p.wpack2_m16 == nullptr) || (p.wpack_m16 && p.out2)) return hipErrorInvalidValue;
    if (p.wpack_m16 && !p.split_s) {                                       // the 16x16x32 form, on its own packs (split chains: the 32x32x16 kernel - its bookkeeping does not fit the registers)
        ConvSplitParams q = p;
        q.wpack = p.wpack_m16;
        q.wpack2 = p.wpack2_m16;
        static std::atomic<int> attr16[64];
        if (!attr16[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_sf_chain16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS_BYTES);
            if (e != hipSuccess) return e;
            attr16[dev] = 1;
        }
        hipLaunchKernelGGL((conv3x3_sf_chain16_kernel<false>), dim3(grid), dim3(SF_THREADS), SF_LDS_BYTES, s, q);
        return hipGetLastError();
    }
    static std::atomic<int> attr_dev[64][4];
    const int var = (p.out2 ? 1 : 0) + (p.split_s ? 2 : 0);               // out2: the split-format copy of the output (the next block's inp0)
    const void* const fns[4] = {reinterpret_cast<const void*>(conv3x3_sf_chain_kernel<false, false>), reinterpret_cast<const void*>(conv3x3_sf_chain_kernel<true, false>),
                                reinterpret_cast<const void*>(conv3x3_sf_chain_kernel<false, true>), reinterpret_cast<const void*>(conv3x3_sf_chain_kernel<true, true>)};
    if (!attr_dev[dev][var]) {
        hipError_t e = hipFuncSetAttribute(fns[var], hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_dev[dev][var] = 1;
    }
    switch (var) {
        case 0: hipLaunchKernelGGL((conv3x3_sf_chain_kernel<false, false>), dim3(grid), dim3(SF_THREADS), SF_LDS_BYTES, s, p); break;
        case 1: hipLaunchKernelGGL((conv3x3_sf_chain_kernel<true, false>), dim3(grid), dim3(SF_THREADS), SF_LDS_BYTES, s, p); break;
        case 2: hipLaunchKernelGGL((conv3x3_sf_chain_kernel<false, true>), dim3(grid), dim3(SF_THREADS), SF_LDS_BYTES, s, p); break;
        default: hipLaunchKernelGGL((conv3x3_sf_chain_kernel<true, true>), dim3(grid), dim3(SF_THREADS), SF_LDS_BYTES, s, p); break;
    }
    return hipGetLastError();
}

hipError_t launch_conv3x3_sf(const ConvSplitParams& p, hipStream_t s) {
    if (!p.in || !p.wpack || !p.bias || !p.out || p.items < 1 || p.H < 1 || p.W < 1 || p.accum || p.out_sf) return hipErrorInvalidValue;
    if ((p.addend == nullptr) != (p.resid == nullptr) || (p.addend && (p.add_div < 1 || p.items % p.add_div))) return hipErrorInvalidValue;
    if ((long long)p.H * p.W * 256 >= 0x7fffffffLL) return hipErrorInvalidValue;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    const int ncu = device_cu_count();
    if (!ncu) return hipErrorUnknown;
    const int grid = ncu >= 8 ? ncu / 8 * 8 : 8;          // whole XCDs; surplus workgroups exit at once
    static std::atomic<int> attr_dev[64][2];                               // the attribute is per device
    const int mode = p.addend ? 1 : 0;
    const void* fn = mode ? reinterpret_cast<const void*>(conv3x3_sf_kernel<1>) : reinterpret_cast<const void*>(conv3x3_sf_kernel<0>);
    if (!attr_dev[dev][mode]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_dev[dev][mode] = 1;
    }
    if (mode) hipLaunchKernelGGL(conv3x3_sf_kernel<1>, dim3(grid), dim3(SF_THREADS), SF_LDS_BYTES, s, p);
    else hipLaunchKernelGGL(conv3x3_sf_kernel<0>, dim3(grid), dim3(SF_THREADS), SF_LDS_BYTES, s, p);
    return hipGetLastError();
}

// ---- fp32 <-> SF (op-level tests and taps only: the forward never converts) -----------------------------------------------
__global__ void sf_from_f32_kernel(const float* in, uint16_t* out, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread = one (pixel, channel)
    if (i >= npix * 64) return;
    const size_t pix = i >> 6;
    const int ch = (int)(i & 63);
    const float x = in[i];
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)((x - (float)hi) * 2048.0f);
    uint16_t hb, lb;
    __builtin_memcpy(&hb, &hi, 2);
    __builtin_memcpy(&lb, &lo, 2);
    uint16_t* o = out + pix * 128 + (ch >> 5) * 64 + (ch & 31);
    o[0] = hb;
    o[32] = lb;
}
__global__ void sf_to_f32_kernel(const uint16_t* in, float* out, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 64) return;
    const size_t pix = i >> 6;
    const int ch = (int)(i & 63);
    const uint16_t* s = in + pix * 128 + (ch >> 5) * 64 + (ch & 31);
    _Float16 hi, lo;
    __builtin_memcpy(&hi, s, 2);
    __builtin_memcpy(&lo, s + 32, 2);
    out[i] = (float)hi + (float)lo * SF_ISCALE;
}
hipError_t launch_sf_from_f32(const float* in, uint16_t* out, size_t npix, hipStream_t s) {
    hipLaunchKernelGGL(sf_from_f32_kernel, dim3((unsigned)((npix * 64 + 255) / 256)), dim3(256), 0, s, in, out, npix);
    return hipGetLastError();
}
hipError_t launch_sf_to_f32(const uint16_t* in, float* out, size_t npix, hipStream_t s) {
    hipLaunchKernelGGL(sf_to_f32_kernel, dim3((unsigned)((npix * 64 + 255) / 256)), dim3(256), 0, s, in, out, npix);
    return hipGetLastError();
}

}  // namespace pfnl

#ifdef PFNL_SFC_TIMING
extern "C" int pfnl_debug_read_sfc_stamps(long long* host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pfnl::sfc_dbg), n * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

// fp32 3x3 64->64 convolution on the f16 matrix pipe with SPLIT operands (option conv3x3=split16): the 3x3 convolutions of
// the progressive-fusion blocks (conv1_i, both halves of conv2_i; reference model/pfnl.py:49-51, 66-71).
//
// gfx950 has no xf32 MFMA; v_mfma_f32_32x32x2_f32 peaks at 157 TFLOP/s while v_mfma_f32_32x32x16_f16 runs at 16x that rate
// with fp32 accumulation.  Every fp32 operand is therefore split exactly into two binary16 numbers
//     x = hi + lo' * 2^-11,   hi = f16(x),  lo' = f16((x - hi) * 2^11)        (x - hi is exact in fp32; |lo'| <= |x|)
// and a product is taken as  x w = hi_x hi_w + (hi_x lo'_w + lo'_x hi_w) 2^-11 ; the dropped lo lo term is 2^-22 relative,
// each kept product is exact in the fp32 accumulator, i.e. a product carries >= 22 mantissa bits and the result differs
// from an fp32 FMA chain by less than the chain's own summation-order noise (tests: error against the fp64 oracle equals
// the direct fp32 kernel's, below the Winograd kernel's).  The lo' parts are kept SCALED by 2^11 so that they live in
// binary16's normal range whatever the magnitude of x (no dependence on subnormal handling); the two cross terms therefore
// accumulate in a second accumulator that is folded in with 2^-11 once per tile.  Domain: |activation|, |weight| < 65504.
// 3 MFMAs of 32 cycles per 16x(32x32) products against 8 MFMAs of 64 cycles: 5.3x less matrix-pipe time than the direct
// f32 kernel, 2.4x less than the Winograd kernel - the convolution becomes bound by LDS / HBM traffic instead.
//
// Structure (the persistent, LDS-resident design of conv_bf16.hip, re-cut for twice the operand bytes):
//   * activations stay fp32 [items][H][W][64] in HBM; the split happens when a halo tile is committed to LDS;
//   * workgroup = 512 threads = 8 waves = (row pair rp, 32-channel tile mt) of an 8 x 32-pixel output tile;
//   * a UNIT is (tile, half of the 64 input channels): its 10 x 34 halo of 32 channels is 128 B per pixel in LDS
//     ([hi 64 B | lo' 64 B], 16-byte chunks XOR-swizzled like conv_bf16.hip), double-buffered (2 x 42.5 KB), next to the
//     72 KB of packed weights of that half ([kx][ks][ky][m][hi/lo'][lane] x 16 B); accumulators run through a tile's two
//     units.  Tiles walk the halves in boustrophedon order (0,1 | 1,0 | 0,1 ...), so the weight pack in LDS is replaced
//     once per tile, in three 24 KB slices that follow the column taps already consumed;
//   * per (column tap kx, 16-channel step ks): 4 halo rows x (hi, lo') + 3 row taps x (hi, lo') = 14 ds_read_b128 feed
//     18 MFMAs (0.78 reads per MFMA; the bf16 kernel needs 1.17), fetched one row tap ahead;
//   * a finished tile leaves through the halo buffer that is free in that phase, in two passes of 128 pixels x 256 B (one
//     per unit of the NEXT tile): accumulators + bias -> LDS scratch (pixel-major), then whole 256-byte pixel lines
//     to HBM with the addend / leaky-relu / residual of conv2_i applied on the way (coalesced reads of both).
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"
#include "conv_split16.h"

#ifndef CS_STORE_AUX
#define CS_STORE_AUX 17     // cache-policy bits of the output stores: sc0 sc1 = written through, not kept in L2.  Measured on
                            // configs[1] (tools/s16_variants.sh, same box, ms per step): 0 (default policy) 5.43, nt 5.43, sc1 5.33,
                            // sc0 sc1 5.30, sc1 nt 5.49 - the 117 MB an output tensor has are not read back through this L2
                            // before they are evicted anyway; kept out, they leave it to the halo rows, weights and addend tiles.
                            // (residual loads with sc1 or nt: no further gain; halo loads with nt: worse)
#endif
#ifndef CS_RESID_AUX
#define CS_RESID_AUX 0      // ... of the residual loads
#endif
#ifndef CS_LATE_FINISH
#define CS_LATE_FINISH 1    // per-frame variant: store pieces 2, 3 finished 4-5 sub-steps after their addend / residual requests
#endif
#ifndef CS_HALO_AUX
#define CS_HALO_AUX 0       // ... of the halo loads
#endif

namespace pfnl {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int CS_THREADS = 512;
constexpr int CS_TH = 8, CS_TW = 32;
constexpr int CS_IH = CS_TH + 2, CS_IW = CS_TW + 2;
constexpr int CS_TILE_BYTES = CS_IH * CS_IW * 128;                  // 43 520 per buffer
constexpr int CS_W_BYTES = 3 * 2 * 3 * 2 * 2 * 1024;                // 73 728: [kx][ks][ky][m][hi/lo][lane] x 16 B, one half
constexpr int CS_SLOT_BYTES = CS_W_BYTES / 3;                       // the weights of one column tap
constexpr int CS_LDS_BYTES = 2 * CS_TILE_BYTES + CS_W_BYTES + 64 * 4;   // 161 024 of 163 840
constexpr int CS_PIECES = CS_IH * CS_IW * 8;                        // 16-byte fp32 pieces (4 channels) of a unit's halo
constexpr int CS_ITERS = (CS_PIECES + CS_THREADS - 1) / CS_THREADS; // 6
constexpr int CS_WITERS = CS_SLOT_BYTES / 16 / CS_THREADS;          // 3 pieces of 16 B per thread and slot
constexpr float CS_SCALE = 2048.0f, CS_ISCALE = 1.0f / 2048.0f;
static_assert(CS_SLOT_BYTES % (16 * CS_THREADS) == 0, "slot copy must divide evenly");

#ifdef PFNL_S16_TIMING   /* phase timeline of the kernel (tools/s16_timing.py); not part of the product build */
__device__ long long cs_dbg[256 * 2 * 128];
#define CS_STAMP() do { if (lane == 0 && (wave == 0 || wave == 5) && dbg_n < 128) cs_dbg[(blockIdx.x * 2 + (wave != 0)) * 128 + dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define CS_STAMP() do {} while (0)
#endif

__device__ __forceinline__ f32x16 mfma_f16(h8 a, h8 b, f32x16 c) {
#ifdef CS_X_NOMFMA   /* timing experiments only (wrong results on purpose) */
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

// x -> (hi, lo') for 4 values: hi = f16(x) (round to nearest even), lo' = f16(x * 2^11 - hi * 2^11): the fused multiply-add is
// exact here (x - hi has at most 13 significant bits), so lo' = f16((x - hi) 2^11) with one rounding.  v_fma_mix*_f16 reads the
// f16 operand in place and writes the f16 result: 8 VALU per 4 values (the plain C++ form compiles to 12-14).
// f32x4 times / fma with a wave-uniform scalar as FOUR scalar instructions.  Written as v * s the compiler emits v_pk_mul_f32 / v_pk_fma_f32,
// and packed fp32 VALU is an anti-lever next to MFMAs on gfx950: it does not run in their shadow, every one stops the matrix pipe for its
// ~4.5 cycles plus a restart (tools/ubench/mfma_agpr_src: 2 v_pk_fma_f32 behind each MFMA = 53 cycles per MFMA, 6 v_fma_f32 = 34.5) -
// for BOTH waves of the SIMD.  Same arithmetic: bit-identical results.  ONLY for operands a VALU instruction produced: an inline-asm
// consumer of MFMA results gets none of the XDL -> VALU wait states the compiler pads its own instructions with (round 5: a fold written
// this way read the cross-term accumulator one MFMA after it was issued - repeatable at -O3, caught as run-to-run noise of 1e-7 by the -O1
// ASAN build); the folds of accumulators therefore stay compiler-generated.
#ifndef CS_SCALAR_F32
#define CS_SCALAR_F32 1
#endif
__device__ __forceinline__ f32x4 mul4s(f32x4 v, float s) {
#if CS_SCALAR_F32
    f32x4 r;
    asm("v_mul_f32 %0, %4, %5\n\tv_mul_f32 %1, %4, %6\n\tv_mul_f32 %2, %4, %7\n\tv_mul_f32 %3, %4, %8"
        : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w) : "s"(s), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    return r;
#else
    return v * s;
#endif
}
__device__ __forceinline__ void split4(f32x4 v, u32x2& hi, u32x2& lo, float nscale) {
    const h4 h = __builtin_convertvector(v, h4);                    // 2 x v_cvt_pk_f16_f32
    hi = __builtin_bit_cast(u32x2, h);
#ifdef CS_X_PLAINSPLIT
    const f32x4 r = (v - __builtin_convertvector(h, f32x4)) * CS_SCALE;
    lo = __builtin_bit_cast(u32x2, __builtin_convertvector(r, h4));
#else
    const f32x4 t = mul4s(v, CS_SCALE);                             // exact
    unsigned l0, l1;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi.x), "s"(nscale), "v"(t.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(hi.x), "s"(nscale), "v"(t.y));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi.y), "s"(nscale), "v"(t.z));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(hi.y), "s"(nscale), "v"(t.w));
    lo = u32x2{l0, l1};
#endif
}

// MODE 0: out = act(conv + bias).   MODE 1 (FUSE; conv2_i per-frame half): out = act(conv + bias + addend[item / add_div]) + resid.
// MODE 2 (ACCUM; convmerge1, reference model/pfnl.py:52,73-74): out[clip] = act(sum over the add_div frames f of a clip of
// conv(in[clip * add_div + f]; weights f) + bias) - the accumulators run through the chain of a clip's frames at one spatial tile,
// every unit brings its own weights (pack index 2 f + half), one epilogue per chain.
// OSF (MODE 0 only; conv1_i): the output is written in the SPLIT FORMAT the consumers' MFMA operands are made of (conv_split16.h:
// per pixel 256 B = [channel half][hi 32 x f16 | lo' 32 x f16]) - the split costs 2 VALU per value once, in the producer, instead
// of once per consumer and halo pixel, and the consumers can bring their halos in by LDS-DMA (conv_sf.hip).
// SPLIT (MODE 2 only; round 6): split chains (conv3x3_c1c10_kernel below has the scheme) - behind the whole rounds of (clip, tile) chains every
// workgroup takes ONE part (frames [sp_f0, sp_f1)) of a cut chain; its sum leaves raw (no bias, no activation) as a dense [8][32][64] fp32 tile
// in p.partial[slot], and c10_finalize_kernel adds the parts up.  A template parameter: the common instantiations keep their instruction streams.
template <int MODE, bool OSF = false, bool SPLIT = false>
__global__ __launch_bounds__(CS_THREADS, 1) void conv3x3_split16_kernel(ConvSplitParams p) {
    constexpr bool FUSE = MODE == 1, ACCUM = MODE == 2;
    static_assert(!OSF || MODE == 0, "split-format output: plain mode only");
    static_assert(!SPLIT || MODE == 2, "split chains: the accumulating mode only");
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_smem[];
    unsigned char* const wl = cs_smem + 2 * CS_TILE_BYTES;
    float* const bl = reinterpret_cast<float*>(cs_smem + 2 * CS_TILE_BYTES + CS_W_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PFNL_S16_TIMING
    int dbg_n = 0;
#endif
#ifdef CS_PRIO_LATE   /* experiment: waves 4-7 (the younger wave of every SIMD) above waves 0-3 */
    if (wave >= 4) __builtin_amdgcn_s_setprio(CS_PRIO_LATE);
#endif
    const int rp = wave >> 1;                                       // rows 2rp, 2rp+1 of the tile
    const int mt = wave & 1;                                        // output channels 32mt .. 32mt+31
    const int H = p.H, W = p.W;
    const int tiles_x = (W + CS_TW - 1) / CS_TW, tiles_y = (H + CS_TH - 1) / CS_TH;
    const int per_item = tiles_x * tiles_y;
    const int item_bytes = H * W * 256;
    // Work order (as conv_bf16.hip): chains of the gT frames of a clip at one spatial tile (the addend tile is then an L2
    // hit for all but the first), dealt out XCD by XCD so that neighbouring tiles share their halo rows in one L2.
    const int gT = (FUSE || ACCUM) ? p.add_div : 1;
    const int nchains = per_item * (p.items / gT);
    const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3, cpx = gridDim.x >> 3;
    const int n_full = SPLIT ? p.n_full : nchains;
    const int per_xcd = (n_full + 7) >> 3;
    const int cbeg = xcd * per_xcd;
    const int ccnt = min(per_xcd, n_full - cbeg);
    if (!SPLIT && xj >= ccnt) return;
    const int nfull_tiles = ((!SPLIT || xj < ccnt) ? (ccnt - xj + cpx - 1) / cpx : 0) * gT;   // tiles of this workgroup's whole chains
    [[maybe_unused]] const int slot = xcd * cpx + xj;
    const bool has_part = SPLIT && slot < (nchains - n_full) * p.split_s;
    const int sp_chain = has_part ? n_full + slot / p.split_s : 0;
    const int sp_f0 = has_part ? (slot % p.split_s) * p.split_q : 0, sp_f1 = has_part ? min(gT, sp_f0 + p.split_q) : 0;
    const int nt = SPLIT ? nfull_tiles + (sp_f1 - sp_f0) : nfull_tiles;   // tiles of this workgroup
    if (SPLIT && nt <= 0) return;
    [[maybe_unused]] const int nu = 2 * nt;                         // units: (tile, channel half); nu >= 2
    // tile k -> (item, y0, x0)
#define CS_TILE(k_, item_, y0_, x0_)                                                             \
    do {                                                                                         \
        int ci_, f_, ch_;                                                                        \
        if (!SPLIT || (k_) < nfull_tiles) {                                                      \
            ci_ = (k_) / gT;                                                                     \
            f_ = (k_) - ci_ * gT;                                                                \
            ch_ = cbeg + xj + ci_ * cpx;                                                         \
        } else {                                                                                 \
            f_ = sp_f0 + ((k_) - nfull_tiles);                                                   \
            ch_ = sp_chain;                                                                      \
        }                                                                                        \
        const int cl_ = ch_ / per_item;                                                          \
        const int sp_ = ch_ - cl_ * per_item;                                                    \
        item_ = cl_ * gT + f_;                                                                   \
        const int ty_ = sp_ / tiles_x;                                                           \
        y0_ = ty_ * CS_TH;                                                                       \
        x0_ = (sp_ - ty_ * tiles_x) * CS_TW;                                                     \
    } while (0)
    // unit u: tile u >> 1; the channel half walks 0,1 | 1,0 | 0,1 ... so that consecutive units of different tiles share it
#define CS_HALF(u_) ((((u_) >> 1) ^ (u_)) & 1)

    // weights of half 0 + bias: requested here, written to LDS in the prologue below - after the first halo has been requested
    // too, so that the two latencies of a launch's start overlap (a launch is only 20-100 us long)
    u32x4 w0reg[CS_W_BYTES / 16 / CS_THREADS];
#pragma unroll
    for (int k = 0; k < CS_W_BYTES / 16 / CS_THREADS; ++k) w0reg[k] = reinterpret_cast<const u32x4*>(p.wpack)[k * CS_THREADS + tid];
    const float bias_r = tid < 64 ? p.bias[tid] : 0.f;

    // staging map: piece id = k*512 + tid -> halo pixel id >> 3, 4-channel piece id & 7 (8 threads read one pixel's 128 B)
    // Two words per piece, constant for the life of the kernel: `grel` = byte offset of the piece relative to the halo origin
    // in HBM, `lpk` = LDS byte address of its hi half-chunk inside a halo buffer | py << 16 | px << 24 (for the border test).
    int grel[CS_ITERS], lpk[CS_ITERS];
    const int wbytes = W * 256;
#pragma unroll
    for (int k = 0; k < CS_ITERS; ++k) {
        const int id = min(k * CS_THREADS + tid, CS_PIECES - 1);    // surplus threads redo the last piece (same value)
        int pix = id >> 3;
        const int c = id & 7;
        // the two pixels of a 16-lane ds_write_b64 group are 8 apart, not neighbours: their hi halves (and then their lo' halves) fall on
        // different banks (see conv3x3_c1c10_kernel, K1_COMMIT_PAIR8: SQ_LDS_BANK_CONFLICT 1.3e6 -> 9e4 per launch)
        if (pix < (CS_IH * CS_IW & ~15)) pix = (pix & ~15) | ((pix & 1) << 3) | ((pix & 15) >> 1);
        const int py = pix / CS_IW, px = pix - py * CS_IW;
        grel[k] = py * wbytes + px * 256 + c * 16;
        lpk[k] = ((py * CS_IW + px) * 128 + 8 * (c & 1) + (((c >> 1) ^ ((px >> 1) & 7)) << 4)) | (py << 16) | (px << 24);
    }
    const float nscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -CS_SCALE)));
    f32x4 stg[CS_ITERS];
// (branch-free: one arm, so that the compiler's vmcnt bookkeeping stays exact for whatever is issued around it - with an
// interior / border pair of arms the wait for the weight slice of unit B also drained half of the halo loads just issued)
#define CS_REQUEST_ALL(rs_, org_, interior_, y0_, x0_)                                           \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < CS_ITERS; ++k_) {                                \
            const int gy_ = (y0_) + ((lpk[k_] >> 16) & 0xff) - 1, gx_ = (x0_) + ((unsigned)lpk[k_] >> 24) - 1; \
            const bool in_ = (interior_) || ((unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W); \
            stg[k_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, in_ ? (org_) + grel[k_] : 0x7fffffff, 0, CS_HALO_AUX)); \
        }                                                                                        \
    } while (0)
    // descriptor of unit u_'s halo: resource of its item, byte offset of the halo origin (+ the channel half), interior flag
#define CS_REQ_SETUP(u_, rs_, org_, interior_, y0_, x0_)                                         \
    int item_q_, y0_, x0_;                                                                       \
    CS_TILE((u_) >> 1, item_q_, y0_, x0_);                                                       \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(p.in) + (size_t)item_q_ * H * W * 64, 0, item_bytes, 0x00020000);     \
    const int org_ = ((y0_ - 1) * W + x0_ - 1) * 256 + CS_HALF(u_) * 128;                        \
    const bool interior_ = y0_ > 0 && y0_ + CS_IH - 1 <= H && x0_ > 0 && x0_ + CS_IW - 1 <= W
#ifdef CS_X_NOCOMMIT   /* timing experiments only */
#define CS_COMMIT1(k_, buf_) do { if (stg[k_].x == 1.2345e30f) *reinterpret_cast<f32x4*>(cs_smem) = stg[k_]; } while (0)
#else
#define CS_COMMIT1(k_, buf_)                                                                     \
    do {                                                                                         \
        u32x2 hi_, lo2_;                                                                         \
        split4(stg[k_], hi_, lo2_, nscale);                                                      \
        const int lo_ = lpk[k_] & 0xffff;                                                        \
        *reinterpret_cast<u32x2*>(cs_smem + (buf_) * CS_TILE_BYTES + lo_) = hi_;                 \
        *reinterpret_cast<u32x2*>(cs_smem + (buf_) * CS_TILE_BYTES + (lo_ ^ 64)) = lo2_;   /* lo' chunk = hi chunk ^ 4 */ \
    } while (0)
#endif

    // operand addresses: pixel operand of (column tap kx, k-step ks, part) = chunk 4*part + 2*ks + (lane >> 5) of halo pixel
    // (row 2*rp + ..., column (lane & 31) + kx); weights: 16 bytes per lane
    int paddr[3];                                                   // k-step 0; k-step 1 = the same with chunk bit 1 flipped (^ 32 bytes)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = (lane & 31) + kx;
        paddr[kx] = ((2 * rp) * CS_IW + col) * 128 + ((((lane >> 5)) ^ ((col >> 1) & 7)) << 4);
    }
    const int lo_xor = 4 << 4;                                      // hi chunk -> lo chunk of the same pixel: chunk index ^ 4
    const unsigned char* const wlane = wl + mt * 2048 + lane * 16;

    // Accumulator rows -> channels as in conv_bf16.hip: register r of a lane is channel 32mt + 16(lane>>5) + r.
    const int ech = 32 * mt + 16 * (lane >> 5);
    f32x16 accm[2], accc[2], accp[2];                               // [output row]: hi.hi products / cross products (x 2^11) / finished tile
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

    // ---- epilogue pieces -----------------------------------------------------------------------
    // pass n (n = 0 in the first unit of the next tile, 1 in its second): output rows 2rp + n of the 4 row pairs = 128 pixels.
    // scratch: pixel (rp*32 + j) x 256 B, 16-byte chunk c stored at (c & 8) | ((c ^ j) & 7): conflict-free for the
    // per-lane dump (32 lanes = 32 pixels, same chunk) and for the line read-back (16 lanes = one pixel).
    auto dump = [&](unsigned char* scratch, int n, int h) __attribute__((always_inline)) {   // channels ech + 8h .. + 7 of row n
        const int j = lane & 31;
        unsigned char* const pl = scratch + (rp * 32 + j) * 256;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r0 = 8 * h + 4 * q;
            const f32x4 v = f32x4{accp[n][r0], accp[n][r0 + 1], accp[n][r0 + 2], accp[n][r0 + 3]};   // (the bias is already in: initial C of the tile)
            const int c = (ech + r0) >> 2;                          // 16-byte chunk of the pixel's line
            *reinterpret_cast<f32x4*>(pl + (((c & 8) | ((c ^ j) & 7)) << 4)) = v;
        }
    };
    const float slope = p.act ? 0.2f : 1.0f;
    [[maybe_unused]] f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};            // ACCUM: the bias joins in the epilogue (chunk tid & 15 of every pixel line)
    if constexpr (ACCUM) bias4 = *reinterpret_cast<const f32x4*>(p.bias + (tid & 15) * 4);
    f32x4 radd[2], rres[2];                                         // FUSE: addend / residual pieces in flight (two store pieces ahead)
    // Store piece k of a thread = 16-byte chunk c = tid & 15 of pixel (row pair k, column tid >> 4): one byte offset per
    // thread (row pair 0), the row pair goes in the scalar offset (2 rows = wbytes2 bytes per step); rows past the image are
    // past the end of the item's buffer resource (the range check covers voffset + soffset), columns past it and "nothing
    // pending" are folded into the offset (0x7fffffff + k * wbytes2 stays out of range).
    int soff0 = 0x7fffffff;
    const int wbytes2 = 2 * W * 256;
    // OSF: a thread owns 8-channel groups instead of 4-channel pieces: store pieces (2j, 2j+1) are the two halves of group
    // cg = tid & 7 of pixel (row pair 2j + (tid >> 8), column (tid >> 3) & 31); the group leaves as one hi chunk + one lo' chunk
    auto piece_setup = [&](int n) __attribute__((always_inline)) {
        if constexpr (OSF) {
            const int sx = ex0p + ((tid >> 3) & 31);
            soff0 = (pending && sx < W) ? ((ey0p + n + 2 * (tid >> 8)) * W + sx) * 256 + ((tid >> 2) & 1) * 128 + (tid & 3) * 16 : 0x7fffffff;
        } else {
            const int sx = ex0p + (tid >> 4);
            soff0 = (pending && sx < W) ? ((ey0p + n) * W + sx) * 256 + (tid & 15) * 16 : 0x7fffffff;
        }
    };
    auto fuse_request = [&](int k) __attribute__((always_inline)) {
        if constexpr (FUSE) {
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.resid) + (size_t)eitemp * H * W * 64, 0, item_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.addend) + (size_t)(eitemp / p.add_div) * H * W * 64, 0, item_bytes, 0x00020000);
#ifdef CS_X_NOADDEND   /* timing experiments only (wrong results on purpose): the load is issued, always to the same cached line */
            radd[k & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (tid & 15) * 16, 0, 0));
#else
            radd[k & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, soff0, k * wbytes2, 0));
#endif
#ifdef CS_X_NORESID
            rres[k & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, (tid & 15) * 16, 0, 0));
#else
            rres[k & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, soff0, k * wbytes2, CS_RESID_AUX));
#endif
        }
    };
    // A store piece is taken in two steps one sub-step apart - the scratch read, then arithmetic + store - so that the LDS
    // round trip passes under that sub-step's MFMAs instead of stalling the (in-order) wave.
    f32x4 pv, pvb;                                                  // (pvb: piece 3 of the per-frame variant, see CS_LATE_FINISH)
    [[maybe_unused]] f32x4 osf_hold = {0.f, 0.f, 0.f, 0.f};
    auto piece_read_to = [&](f32x4& dst, const unsigned char* scratch, int k) __attribute__((always_inline)) {
        int pp, c;
        if constexpr (OSF) {
            pp = ((k >> 1) * CS_THREADS + tid) >> 3;
            c = 2 * (tid & 7) + (k & 1);
        } else {
            const int id = k * CS_THREADS + tid;
            pp = id >> 4;
            c = id & 15;
        }
        dst = *reinterpret_cast<const f32x4*>(scratch + pp * 256 + (((c & 8) | ((c ^ pp) & 7)) << 4));
    };
    auto piece_read = [&](const unsigned char* scratch, int k) __attribute__((always_inline)) { piece_read_to(pv, scratch, k); };
    // `next` >= 0: the addend / residual lines of piece `next` (same register slot as piece k) are requested between the arithmetic
    // of piece k and its store.  vmcnt is in order and counts stores: requested AFTER the store, the wait for those lines would also
    // be a wait for the store to reach memory.
    auto piece_finish_from = [&](const f32x4& src, int k, int next) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)eitemp * H * W * 64, 0, item_bytes, 0x00020000);
        f32x4 v = src;
        if constexpr (FUSE) v += radd[k & 1];
        if constexpr (ACCUM) v += bias4;
#ifdef CS_X_PKSLOPE
        const f32x4 sv = mul4s(v, slope);
        v.x = fmaxf(v.x, sv.x);
        v.y = fmaxf(v.y, sv.y);
        v.z = fmaxf(v.z, sv.z);
        v.w = fmaxf(v.w, sv.w);
#else
        {                                                           // leaky_relu(0.2) or identity (slope 1), branch-free
            const f32x4 sv = mul4s(v, slope);
            asm("v_max_f32 %0, %1, %2" : "=v"(v.x) : "v"(v.x), "v"(sv.x));   // (fmaxf adds a canonicalising v_max x,x,x per element)
            asm("v_max_f32 %0, %1, %2" : "=v"(v.y) : "v"(v.y), "v"(sv.y));
            asm("v_max_f32 %0, %1, %2" : "=v"(v.z) : "v"(v.z), "v"(sv.z));
            asm("v_max_f32 %0, %1, %2" : "=v"(v.w) : "v"(v.w), "v"(sv.w));
        }
#endif
        if constexpr (FUSE) v += rres[k & 1];
#ifndef CS_REQ_AFTER_STORE
        if (next >= 0) fuse_request(next);
#endif
        if constexpr (OSF) {
            if ((k & 1) == 0) {
                osf_hold = v;                                       // channels 8cg .. 8cg+3: wait for the other half of the group
            } else {
                u32x2 h0, l0, h1, l1;
                split4(osf_hold, h0, l0, nscale);
                split4(v, h1, l1, nscale);
                buffer_store_b128_guarded<CS_STORE_AUX>(u32x4{h0.x, h0.y, h1.x, h1.y}, rsO, soff0, (k >> 1) * 2 * wbytes2);
                buffer_store_b128_guarded<CS_STORE_AUX>(u32x4{l0.x, l0.y, l1.x, l1.y}, rsO, (int)((unsigned)soff0 + 64u), (k >> 1) * 2 * wbytes2);
            }
        } else {
#ifdef CS_X_NOSTORE   /* timing experiments only */
            if (v.x == 1.2345e30f)
#endif
            buffer_store_b128_guarded<CS_STORE_AUX>(__builtin_bit_cast(u32x4, v), rsO, soff0, k * wbytes2);   // (common.h: store-data hazard)
        }
    };
    auto piece_finish = [&](int k, int next = -1) __attribute__((always_inline)) {
        piece_finish_from(pv, k, next);
#ifdef CS_REQ_AFTER_STORE   /* the order before this was measured */
        if (next >= 0) fuse_request(next);
#endif
    };
    auto store_piece = [&](const unsigned char* scratch, int k) __attribute__((always_inline)) {
        piece_read(scratch, k);
        piece_finish(k);
    };

    // ---- weight replacement: a slot (24 KB, one column tap) travels L2 -> registers -> LDS, 3 x 16 B per thread.
    // (global_load_lds_dwordx4 would need no registers, but with an LDS-DMA in flight this compiler turns EVERY vmcnt wait
    // into vmcnt(0) - the halo commit would then wait for the stores issued a moment earlier: measured.)
    [[maybe_unused]] u32x4 wnx[CS_WITERS];
    auto w_request = [&](int half, int slot) __attribute__((always_inline)) {
#ifndef CS_X_NOSWAP   /* timing experiments only */
        const u32x4* src = reinterpret_cast<const u32x4*>(p.wpack) + (size_t)half * (CS_W_BYTES / 16) + slot * (CS_SLOT_BYTES / 16);
#pragma unroll
        for (int k = 0; k < CS_WITERS; ++k) wnx[k] = src[k * CS_THREADS + tid];
#endif
    };
    auto w_write = [&](int slot) __attribute__((always_inline)) {
#ifndef CS_X_NOSWAP
        u32x4* dst = reinterpret_cast<u32x4*>(wl + slot * CS_SLOT_BYTES);
#pragma unroll
        for (int k = 0; k < CS_WITERS; ++k) dst[k * CS_THREADS + tid] = wnx[k];
#endif
    };
    // Workgroup barrier of the main loop: this wave's ds_writes complete (lgkmcnt), then s_barrier - without the memory-model
    // fences of __syncthreads() (nothing here communicates through global memory); the "memory" clobber keeps the compiler
    // from moving accesses across.
#define CS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // ---- prologue: halo of unit 0 -> buffer 0 -----------------------------------------------------------
    {
        CS_REQ_SETUP(0, rs, org, interior, y0q, x0q);
        CS_REQUEST_ALL(rs, org, interior, y0q, x0q);
#pragma unroll
        for (int k = 0; k < CS_W_BYTES / 16 / CS_THREADS; ++k) reinterpret_cast<u32x4*>(wl)[k * CS_THREADS + tid] = w0reg[k];
        if (tid < 64) bl[tid] = bias_r;
#pragma unroll
        for (int k = 0; k < CS_ITERS; ++k) CS_COMMIT1(k, 0);
        if constexpr (ACCUM) w_request(0, 2);                       // (every ACCUM unit writes "its" slot 2 in group 1: identical data here)
    }
    __syncthreads();

    // One loop iteration = one tile = unit A (first channel half of the tile, parity 0) + unit B (parity 1).  Everything that
    // depends on the parity (LDS buffer, weight replacement, epilogue pass) is a compile-time constant, and the sequence of
    // vector-memory operations of an iteration is branch-free (work that does not apply - no tile awaiting its epilogue yet -
    // runs with out-of-range offsets: dropped stores, zero loads), so that the compiler's s_waitcnt counts are exact: a
    // conservative vmcnt(0) in front of the halo commit would wait for the stores issued a moment earlier (measured:
    // 120 us per conv1_i launch with runtime branches around them).
    // coordinates of the current tile and of the next one: ONE decode (three scalar divisions) per tile, done in unit A for the
    // tile after this one, where its latency is nobody's critical path
    int c_item, c_y0, c_x0, n_item, n_y0, n_x0;
    CS_TILE(0, c_item, c_y0, c_x0);
    n_item = c_item;
    n_y0 = c_y0;
    n_x0 = c_x0;
    [[maybe_unused]] int fch = 0;                                   // ACCUM: frame of the chain the current tile is
    for (int kt = 0; kt < nt; ++kt) {
        const int half_a = kt & 1;                                  // channel half of unit A; unit B: the other one
        auto unit = [&](auto par) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par)::value;               // 0: unit A, 1: unit B
            constexpr int cb = PAR;                                 // LDS buffer of this unit (u = 2 kt + PAR)
            const unsigned char* const tile = cs_smem + cb * CS_TILE_BYTES;
            unsigned char* const other = cs_smem + (cb ^ 1) * CS_TILE_BYTES;
            [[maybe_unused]] const int u = 2 * kt + PAR;
            CS_STAMP();                                             // 0: unit start
            // weights of unit B (the other channel half) follow the column taps unit A has consumed.  A slice must be in LDS
            // before the barrier that precedes its first operand prefetch (issued one sub-step before its group):
            //   tap 0: requested here, written in group 3 of unit A (free since b0; read from unit B's start, after b2)
            //   tap 1: requested in group 3, written in group 5 (free since b1; prefetched in unit B's group 1, after b2)
            //   tap 2: requested in group 5, written in group 1 of unit B (free since b2; prefetched in group 3, after b0)
            // operands: X[row][part], Wv[substep parity][part]; the first ones are asked for before anything else of the unit
            // (request set-up, tile decode: ~50 scalar instructions) so that their LDS round trip passes under it
            h8 X[4][2], Wv[2][2];
#define CS_PX(g_, r_, part_) (*reinterpret_cast<const h8*>(tile + (paddr[(g_) >> 1] ^ (((part_) ? lo_xor : 0) | (((g_) & 1) << 5))) + (r_) * (CS_IW * 128)))
#define CS_WT(g_, ky_, part_) (*reinterpret_cast<const h8*>(wlane + (((g_) * 3 + (ky_)) << 12) + ((part_) << 10)))
            X[0][0] = CS_PX(0, 0, 0);
            X[0][1] = CS_PX(0, 0, 1);
            X[1][0] = CS_PX(0, 1, 0);
            X[1][1] = CS_PX(0, 1, 1);
            Wv[0][0] = CS_WT(0, 0, 0);
            Wv[0][1] = CS_WT(0, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            // ACCUM: every unit has its own weights (pack index 2 f + half): the same three-slice replacement in BOTH units,
            // one group later - group 1: slot 2 of THIS unit written (requested in the previous unit's group 5), slot 0 of the
            // next unit requested; group 3: slot 0 written, slot 1 requested; group 5: slot 1 written, slot 2 requested.
            [[maybe_unused]] int w_next = 0;
            if constexpr (ACCUM) {
                // (SPLIT: behind this workgroup's last whole chain comes its part, which starts at frame sp_f0)
                const int f_end = (SPLIT && kt >= nfull_tiles) ? sp_f1 : gT;
                const int fn = fch + 1 == f_end ? ((SPLIT && kt + 1 == nfull_tiles) ? sp_f0 : 0) : fch + 1;
                w_next = PAR == 0 ? 2 * fch + (half_a ^ 1) : 2 * fn + (half_a ^ 1);   // unit B's half is also the next tile's first half
            }
            if constexpr (PAR == 0 && !ACCUM) w_request(half_a ^ 1, 0);
            // the NEXT unit's halo: requested here, committed in groups 4-5 of this unit (4 groups = ~2 us later), i.e. request
            // and use never straddle the loop back-edge and the compiler's vmcnt for the commit is exact
            // unit A asks for the other half of ITS tile, unit B for the first half of the next tile
            const int q_item = PAR == 0 ? c_item : n_item, y0q = PAR == 0 ? c_y0 : n_y0, x0q = PAR == 0 ? c_x0 : n_x0;
            const int q_half = half_a ^ 1;                          // (boustrophedon: unit B's half is also the next tile's first half)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.in) + (size_t)q_item * H * W * 64, 0, item_bytes, 0x00020000);
            const int org = ((y0q - 1) * W + x0q - 1) * 256 + q_half * 128;
            const bool interior = y0q > 0 && y0q + CS_IH - 1 <= H && x0q > 0 && x0q + CS_IW - 1 <= W;
#if !defined(CS_X_NOLOAD) && !defined(CS_SPREAD_REQ)
            CS_REQUEST_ALL(rs, org, interior, y0q, x0q);
#endif
            piece_setup(PAR);                                       // epilogue pass PAR of the previous tile (nothing pending: out of range)

            [[maybe_unused]] f32x16 bias16;                         // register r of a lane = channel ech + r
            if constexpr (PAR == 0 && !ACCUM) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + ech + 4 * q);
                    bias16[4 * q] = b4.x;
                    bias16[4 * q + 1] = b4.y;
                    bias16[4 * q + 2] = b4.z;
                    bias16[4 * q + 3] = b4.w;
                }
            }

            auto substep = [&](auto sc) __attribute__((always_inline)) {
                constexpr int S = decltype(sc)::value;
                constexpr int g = S / 3, ky = S % 3;
                // --- the slice of non-MFMA work that rides on this group (first sub-step of each group)
                if constexpr (ky == 0) {
                    if constexpr (g == 0) {
                        dump(other, PAR, 0);
                        fuse_request(0);
                        fuse_request(1);
                    }
                    if constexpr (g == 1) {
                        dump(other, PAR, 1);
                        if constexpr (ACCUM) {
                            w_write(2);
                            w_request(w_next, 0);
                        } else if constexpr (PAR == 1) {
                            w_write(2);
                        }
                    }
                    if constexpr (g == 2) {
                        CS_STAMP();                                 // 1: groups 0-1 done
                        CS_BARRIER();                              // b0: scratch complete; column tap 0 of the weights consumed
                        CS_STAMP();                                 // 2: past b0
                        CS_STAMP();                                 // 3
                        piece_read(other, 0);
                    }
                    if constexpr (g == 3) {
                        if constexpr (CS_LATE_FINISH && FUSE) {
                            piece_read_to(pvb, other, 3);          // (scratch reads stay in front of b1; the arithmetic + store need not)
                        } else {
                            piece_finish(2);
                            piece_read(other, 3);
                        }
                        if constexpr (ACCUM) {
                            w_write(0);
                            w_request(w_next, 1);
                        } else if constexpr (PAR == 0) {
                            w_write(0);
                            w_request(half_a ^ 1, 1);
                        }
                        if constexpr (PAR == 0) {                   // decode the next tile (past the end: this one again - a harmless re-read)
                            const int kn = min(kt + 1, nt - 1);
                            CS_TILE(kn, n_item, n_y0, n_x0);
                        }
                    }
                    if constexpr (g == 4) {
                        CS_STAMP();                                 // 4: groups 2-3 done
                        CS_BARRIER();                              // b1: scratch read; column tap 1 consumed
                        CS_STAMP();                                 // 5: past b1
                        CS_STAMP();                                 // 6
#pragma unroll
                        for (int k = 0; k < CS_ITERS; ++k) asm volatile("" : "+v"(lpk[k]));   // opaque: addresses derived from it are not hoisted into registers
#pragma unroll
                        for (int k = 0; k < CS_ITERS / 2; ++k) CS_COMMIT1(k, cb ^ 1);
                        CS_STAMP();                                 // 7: first half of the halo committed
                    }
                    if constexpr (g == 5) {
#pragma unroll
                        for (int k = CS_ITERS / 2; k < CS_ITERS; ++k) CS_COMMIT1(k, cb ^ 1);
                        if constexpr (ACCUM) {
                            w_write(1);
                            w_request(w_next, 2);
                        } else if constexpr (PAR == 0) {
                            w_write(1);
                            w_request(half_a ^ 1, 2);
                        }
                    }
                }
#if defined(CS_SPREAD_REQ) && !defined(CS_X_NOLOAD)   /* experiment: one halo request per sub-step instead of six at the unit's start */
                if constexpr (S < CS_ITERS) {
                    const int gy_ = y0q + ((lpk[S] >> 16) & 0xff) - 1, gx_ = x0q + ((unsigned)lpk[S] >> 24) - 1;
                    const bool in_ = interior || ((unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W);
                    stg[S] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, in_ ? org + grel[S] : 0x7fffffff, 0, CS_HALO_AUX));
                }
#endif
                if constexpr (S == 7) {
                    piece_finish(0, 2);
                    piece_read(other, 1);
                }
                if constexpr (S == 8) {
                    piece_finish(1, 3);
                    piece_read(other, 2);
                }
                // The addend / residual lines of pieces 2, 3 can only be requested once pieces 0, 1 have left their two register
                // slots (sub-steps 7, 8).  Finished two sub-steps later (the plain order) a CU would have to receive 32 KB in ~1.3 k
                // cycles - twice what the fabric gives it; the per-frame variant therefore finishes them in sub-steps 11 and 13
                // (piece 3 after b1: it only needs its registers).
                if constexpr (CS_LATE_FINISH && FUSE) {
                    if constexpr (S == 11) piece_finish(2);
                    if constexpr (S == 13) piece_finish_from(pvb, 3, -1);
                } else {
                    if constexpr (S == 10) piece_finish(3);
                }
                __builtin_amdgcn_sched_barrier(0);
                // --- operands of the next sub-step
#ifdef CS_X_NOREAD   /* timing experiments only */
                if constexpr (false) {
#else
                if constexpr (S < 17) {
#endif
                    constexpr int S1 = S + 1, g1 = S1 / 3, ky1 = S1 % 3;
                    Wv[S1 & 1][0] = CS_WT(g1, ky1, 0);
                    Wv[S1 & 1][1] = CS_WT(g1, ky1, 1);
                    if constexpr (ky1 == 0) {
                        X[0][0] = CS_PX(g1, 0, 0);
                        X[0][1] = CS_PX(g1, 0, 1);
                        X[1][0] = CS_PX(g1, 1, 0);
                        X[1][1] = CS_PX(g1, 1, 1);
                    } else {
                        X[ky1 + 1][0] = CS_PX(g1, ky1 + 1, 0);
                        X[ky1 + 1][1] = CS_PX(g1, ky1 + 1, 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // --- 6 MFMAs: row tap ky of both output rows
                const h8 wh = Wv[S & 1][0], wo = Wv[S & 1][1];
                if constexpr (PAR == 0 && S == 0 && !ACCUM) {       // a tile's first products: C = bias / 0 (no registers to clear)
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    accm[0] = mfma_f16(wh, X[ky][0], bias16);
                    accm[1] = mfma_f16(wh, X[ky + 1][0], bias16);
                    accc[0] = mfma_f16(wo, X[ky][0], zero);
                    accc[1] = mfma_f16(wo, X[ky + 1][0], zero);
                } else {
                    accm[0] = mfma_f16(wh, X[ky][0], accm[0]);
                    accm[1] = mfma_f16(wh, X[ky + 1][0], accm[1]);
                    accc[0] = mfma_f16(wo, X[ky][0], accc[0]);
                    accc[1] = mfma_f16(wo, X[ky + 1][0], accc[1]);
                }
                accc[0] = mfma_f16(wh, X[ky][1], accc[0]);
                accc[1] = mfma_f16(wh, X[ky + 1][1], accc[1]);
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
#undef CS_PX
#undef CS_WT
            if constexpr (PAR == 1 && ACCUM) {                      // a tile is one frame of the chain: only the last one ends a sum
                const bool last = fch + 1 == ((SPLIT && kt >= nfull_tiles) ? sp_f1 : gT);   // (wave-uniform; arithmetic only inside the branch)
                if (last) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        accp[n] = accm[n] + accc[n] * CS_ISCALE;   // (MFMA results: compiler-generated, hazard-padded)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            accm[n][r] = 0.f;
                            accc[n][r] = 0.f;
                        }
                    }
                    ex0p = c_x0;
                    ey0p = c_y0;
                    eitemp = c_item / gT;                           // output item = the clip
                }
                pending = last;
                fch = last ? ((SPLIT && kt + 1 == nfull_tiles) ? sp_f0 : 0) : fch + 1;
                c_item = n_item;
                c_y0 = n_y0;
                c_x0 = n_x0;
            }
            if constexpr (PAR == 1 && !ACCUM) {                     // the tile is complete: fold the cross terms in, hand it to the epilogue
                // (the previous tile's second pass ran in this unit: accp is free)
#pragma unroll
                for (int n = 0; n < 2; ++n) accp[n] = accm[n] + accc[n] * CS_ISCALE;   // (MFMA results: compiler-generated, hazard-padded)
                ex0p = c_x0;
                ey0p = c_y0;
                eitemp = c_item;
                c_item = n_item;                                    // on to the next tile
                c_y0 = n_y0;
                c_x0 = n_x0;
#ifndef CS_X_NOEPI   /* timing experiments only */
                pending = true;
#endif
            }
            CS_STAMP();                                             // 8: groups 4-5 done
            CS_BARRIER();                                          // b2: this unit's buffer is free, the next unit's is complete
        };
        unit(std::integral_constant<int, 0>{});
        unit(std::integral_constant<int, 1>{});
    }

    // ---- the last tile: both passes, any buffer is free now ------------------------------------------
    if constexpr (SPLIT) {
        if (has_part) {                                             // (wave-uniform) the part's sum: raw, a dense [row][column][64] fp32 tile in its slot
            float* const pt = p.partial + (size_t)slot * (CS_TH * CS_TW * 64);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                unsigned char* const scratch = cs_smem + n * CS_TILE_BYTES;
                dump(scratch, n, 0);
                dump(scratch, n, 1);
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 4; ++k) {                       // piece k = chunk tid & 15 of pixel (row pair k, column tid >> 4) of output row 2 k + n
                    piece_read(scratch, k);
                    *reinterpret_cast<f32x4*>(pt + ((size_t)(2 * k + n) * CS_TW + (tid >> 4)) * 64 + (tid & 15) * 4) = pv;
                }
            }
            return;
        }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        unsigned char* const scratch = cs_smem + n * CS_TILE_BYTES;
        piece_setup(n);
        fuse_request(0);
        fuse_request(1);
        dump(scratch, n, 0);
        dump(scratch, n, 1);
        __syncthreads();
        store_piece(scratch, 0);
        store_piece(scratch, 1);
        fuse_request(2);
        fuse_request(3);
        store_piece(scratch, 2);
        store_piece(scratch, 3);
    }
#undef CS_COMMIT1
#undef CS_REQ_SETUP
#undef CS_REQUEST_ALL
#undef CS_HALF
#undef CS_TILE
}

// ------------------------------------------------------------------------------------------------------------------------
// conv1_i AND conv10_i of a progressive-fusion block in ONE launch (reference model/pfnl.py:66-68): a chain = the T frames of a
// clip at one spatial tile.  Every frame tile is conv1_i as above (fp32 halo through registers, split on the way into LDS, the
// same unit / sub-step / weight-slice pipeline); what changes is what happens to a finished tile:
//   * it is not parked in registers for an epilogue spread over the next tile (32 registers) - it leaves at once, in a short SERIAL
//     phase: bias (initial C) + leaky-relu + split in registers -> the unit's own halo buffer, free at that point, as 128 pixel lines
//     of the split format (one output row of each of the 4 row pairs; two passes), from where
//       - whole 256-byte lines go to HBM (`out`: inp1 in the split format, what conv2_i's chain kernel reads by LDS-DMA) - picked
//         up into 8 registers per thread here, stored one piece per sub-step under the NEXT tile's MFMAs (a CU moves store data at
//         ~16 B per clock: the 32 KB of a row take 2 k cycles when waited for) - and
//       - the lines come back as the B operands (k = channel, n = pixel) of conv10_i's slice of this frame, W10[64 f .. 64 f + 63][:],
//         whose A operands (32 output channels x 16 input channels, hi / lo') stream from L2 into registers: 12 MFMAs per pass and
//         wave (K = 64) into `base_m`, the 32 registers the parked tile used to take, which run through the T frames of the chain;
//   * behind the chain's last frame `base_m` (+ bias: its initial value) takes the same road - leaky-relu, split, lines - to `out2`.
// conv10_i as a launch of its own read inp1 (117 MB at configs[1]) back for 11 % of conv1_i's MFMAs; here it reads nothing.
// The weight slices travel by LDS-DMA (no staging registers) and the halo pieces' source offsets are recomputed: 254 VGPRs.
// The serial phase is ~4.5 - 5.5 k shader cycles per tile of ~26 k (tools/k1_timing.py: barrier waits 1.5 k, two dumps 1.1 k, the
// two passes of operand / line reads + 12 MFMAs 2 - 2.7 k); rearranging it moves the launch time by nothing - the launch runs at the
// pace the power cap gives its MFMAs; what the fusion saves is the 117 MB and a launch.

// one LDS-DMA instruction: lane L's 16 bytes at (resource, voff) -> LDS [lds_dst + 16 L] (conv_sf.hip, sf_dma16: m0 saved / restored)
// + a scalar byte offset on the memory side only (the instruction's immediate offset would move the LDS side as well)
__device__ __forceinline__ void k1_dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs), "s"(soff) : "memory");
}

#ifndef K1_W10_AT
#define K1_W10_AT 16          // sub-step of a tile's second unit at which conv10_i's operands are requested
#endif
#ifndef K1_STORE_AUX
#define K1_STORE_AUX CS_STORE_AUX
#endif
#ifndef K1_COMMIT_PAIR8
#define K1_COMMIT_PAIR8 1     // halo commit: the two pixels of a 16-lane ds_write_b64 group are 8 apart (bank-conflict-free), not neighbours
#endif
constexpr int K1_LDS_BYTES = CS_LDS_BYTES + 64 * 4;                 // + conv10_i's bias
// ISF (round 6): `in` is the SPLIT-FORMAT copy of inp0 that conv3x3_sf_chain_kernel<true> writes next to the fp32 one (conv_sf.hip): the
// halo of a unit travels HBM -> LDS by LDS-DMA in operand form (43 wave instructions of 1 KB, the XOR swizzle applied to the source
// address, out-of-image pixels = out-of-range offsets = zeros: exactly conv_sf.hip's input path) - no staging registers, no split4 on
// the VALU (48 per thread and unit), no ds_write commit (12 per thread and unit).  The operands are the same binary16 pairs the fp32
// path builds (one split4 per value in the producer's epilogue instead of 1.33 per halo value here): results are bit-identical.
constexpr int K1_DMA_NDMA = (CS_IH * CS_IW + 7) / 8;                // 43 DMA instructions of 8 pixels x 128 B
constexpr int K1_DMA_ITERS = (K1_DMA_NDMA + 7) / 8;                 // 6 per wave (waves 3..7: 5)
constexpr int K1_SF_TILE_BYTES = K1_DMA_NDMA * 1024;                // 44 032: the last instruction's 4 surplus pixels land in padding
constexpr int K1_SF_LDS_BYTES = 2 * K1_SF_TILE_BYTES + CS_W_BYTES + 2 * 64 * 4;   // 162 304 of 163 840
static_assert(K1_SF_LDS_BYTES <= 160 * 1024, "LDS budget");

// SPLIT: the launch cuts the chains of its last, partial round by frames (p.split_s > 0; below).  A template parameter, not a run-time
// test: with the part bookkeeping compiled in, the whole-round launches (configs[1]) ran 1.3 us per launch slower on the same box.
template <bool ISF, bool SPLIT>
__global__ __launch_bounds__(CS_THREADS, 1) void conv3x3_c1c10_kernel(ConvSplitParams p) {
    constexpr int TILE_BYTES = ISF ? K1_SF_TILE_BYTES : CS_TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_smem[];
    unsigned char* const wl = cs_smem + 2 * TILE_BYTES;
    float* const bl = reinterpret_cast<float*>(cs_smem + 2 * TILE_BYTES + CS_W_BYTES);
    float* const bl2 = bl + 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PFNL_S16_TIMING
    int dbg_n = 0;
#endif
#if PFNL_S16_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(PFNL_S16_PRIO);      // the younger wave of every SIMD above the older one (tools/ubench/conv_core V3)
#endif
    const int rp = wave >> 1;                                       // rows 2rp, 2rp+1 of the tile
    const int mt = wave & 1;                                        // output channels 32mt .. 32mt+31 (of conv1_i and of conv10_i)
    const int H = p.H, W = p.W;
    const int tiles_x = (W + CS_TW - 1) / CS_TW, tiles_y = (H + CS_TH - 1) / CS_TH;
    const int per_item = tiles_x * tiles_y;
    const int item_bytes = H * W * 256;
    const int gT = p.add_div;                                       // frames per clip = tiles per chain
    const int nchains = per_item * (p.items / gT);
    const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3, cpx = gridDim.x >> 3;
    // SPLIT CHAINS (round 6; p.split_s > 0): the first p.n_full chains - a whole number of rounds of the grid - are dealt out as whole
    // chains, as ever; every chain behind them is cut by FRAMES into p.split_s parts of <= p.split_q frames, one part per workgroup (slot
    // = xcd * cpx + xj -> chain n_full + slot / s, part slot % s).  A launch of 1.25 rounds of chains then takes 1 chain + 2 tiles
    // instead of 2 chains.  A part cannot finish conv10_i - it only sees its own frames: it leaves its partial sum (part 0 starts from
    // the bias, the others from 0) as raw fp32 in p.partial[slot], and c10_finalize_kernel adds the parts up in fixed order.
    const int n_full = SPLIT ? p.n_full : nchains;
    const int per_xcd = (n_full + 7) >> 3;
    const int cbeg = xcd * per_xcd;
    const int ccnt = min(per_xcd, n_full - cbeg);
    if (!SPLIT && xj >= ccnt) return;
    const int nfull_tiles = ((!SPLIT || xj < ccnt) ? (ccnt - xj + cpx - 1) / cpx : 0) * gT;   // tiles of this workgroup's whole chains
    [[maybe_unused]] const int slot = xcd * cpx + xj;
    const bool has_part = SPLIT && slot < (nchains - n_full) * p.split_s;
    const int sp_chain = has_part ? n_full + slot / p.split_s : 0;
    const int sp_r = has_part ? slot % p.split_s : 0;
    const int sp_f0 = has_part ? sp_r * p.split_q : 0, sp_f1 = has_part ? min(gT, sp_f0 + p.split_q) : 0;   // frames [sp_f0, sp_f1) of chain sp_chain
    const int nt = SPLIT ? nfull_tiles + (sp_f1 - sp_f0) : nfull_tiles;   // tiles of this workgroup
    if (SPLIT && nt <= 0) return;
#define K1_TILE(k_, item_, y0_, x0_, fr_)                                                        \
    do {                                                                                         \
        int ch_;                                                                                 \
        if (!SPLIT || (k_) < nfull_tiles) {                                                      \
            const int ci_ = (k_) / gT;                                                           \
            fr_ = (k_) - ci_ * gT;                                                               \
            ch_ = cbeg + xj + ci_ * cpx;                                                         \
        } else {                                                                                 \
            fr_ = sp_f0 + ((k_) - nfull_tiles);                                                  \
            ch_ = sp_chain;                                                                      \
        }                                                                                        \
        const int cl_ = ch_ / per_item;                                                          \
        const int sp_ = ch_ - cl_ * per_item;                                                    \
        item_ = cl_ * gT + fr_;                                                                  \
        const int ty_ = sp_ / tiles_x;                                                           \
        y0_ = ty_ * CS_TH;                                                                       \
        x0_ = (sp_ - ty_ * tiles_x) * CS_TW;                                                     \
    } while (0)

    u32x4 w0reg[CS_W_BYTES / 16 / CS_THREADS];
#pragma unroll
    for (int k = 0; k < CS_W_BYTES / 16 / CS_THREADS; ++k) w0reg[k] = reinterpret_cast<const u32x4*>(p.wpack)[k * CS_THREADS + tid];
    const float bias_r = tid < 64 ? p.bias[tid] : (tid < 128 ? p.bias2[tid - 64] : 0.f);

    [[maybe_unused]] int lpk[CS_ITERS];                             // staging map: as conv3x3_split16_kernel; the source offset of a piece
    const int wbytes = W * 256;                                     // (py * wbytes + px * 256 + 16 (tid & 7)) is recomputed from it: no registers to spare
    [[maybe_unused]] int dpk[K1_DMA_ITERS];                         // ISF: py | px << 8 of the lane's halo pixel per DMA instruction (conv_sf.hip)
    if constexpr (ISF) {
#pragma unroll
        for (int k = 0; k < K1_DMA_ITERS; ++k) {
            const int pix = 8 * (wave + 8 * k) + (lane >> 3);
            const int py = pix / CS_IW, px = pix - py * CS_IW;
            dpk[k] = py | (px << 8);
        }
    }
    [[maybe_unused]] const unsigned lds0 = (unsigned)(uintptr_t)cs_smem;   // LDS byte address of halo buffer 0
#define K1_DMA_HALO(rs_, org_, interior_, y0_, x0_, buf_)                                        \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < K1_DMA_ITERS; ++k_) {                            \
            const int i_ = wave + 8 * k_;                                                        \
            if (k_ < K1_DMA_ITERS - 1 || i_ < K1_DMA_NDMA) {                                     \
                const int py_ = dpk[k_] & 0xff, px_ = dpk[k_] >> 8;                              \
                const int gy_ = (y0_) + py_ - 1, gx_ = (x0_) + px_ - 1;                          \
                const bool in_ = (interior_) || ((unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W && py_ < CS_IH); \
                const int rel_ = py_ * wbytes + px_ * 256 + (((lane & 7) ^ ((px_ >> 1) & 7)) << 4); \
                k1_dma16(rs_, lds0 + (buf_) * TILE_BYTES + i_ * 1024, in_ ? (org_) + rel_ : 0x7fffffff, 0); \
            }                                                                                    \
        }                                                                                        \
    } while (0)
#pragma unroll
    for (int k = 0; k < (ISF ? 0 : CS_ITERS); ++k) {
        // surplus threads redo a piece of the last pixel - the one with THEIR channel piece (tid & 7), so that the recomputed source
        // offset below and the LDS address agree and the duplicate writes carry the same bytes
        const int id = k * CS_THREADS + tid < CS_PIECES ? k * CS_THREADS + tid : CS_PIECES - 8 + (tid & 7);
        int pix = id >> 3;
        const int c = id & 7;
#if K1_COMMIT_PAIR8
        // a ds_write_b64 is served 16 lanes (two pixels of 8 pieces) at a time over 32 banks = one 128-byte pixel slot: with pixels p, p + 1
        // both hi halves (and then both lo' halves) fall on the same 16 banks - 2-way conflicts on every commit (SQ_LDS_BANK_CONFLICT 1.3e6
        // per launch, profiles/r04_pmc.md).  Paired as (p, p + 8) the swizzle term (px >> 1) & 7 differs in bit 2: one pixel's hi half
        // occupies the banks of the other's lo' half.  (the last 4 pixels of the 340 keep their order)
        if (pix < (CS_IH * CS_IW & ~15)) pix = (pix & ~15) | ((pix & 1) << 3) | ((pix & 15) >> 1);
#endif
        const int py = pix / CS_IW, px = pix - py * CS_IW;
        lpk[k] = ((py * CS_IW + px) * 128 + 8 * (c & 1) + (((c >> 1) ^ ((px >> 1) & 7)) << 4)) | (py << 16) | (px << 24);
    }
    const int c16 = (tid & 7) * 16;                                 // (piece id = k * 512 + tid: its 4-channel piece of the pixel is tid & 7 for every k)
    const float nscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -CS_SCALE)));
    [[maybe_unused]] f32x4 stg[CS_ITERS];
#define K1_REQUEST_ALL(rs_, org_, interior_, y0_, x0_)                                           \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < CS_ITERS; ++k_) {                                \
            const int py_ = (lpk[k_] >> 16) & 0xff, px_ = (unsigned)lpk[k_] >> 24;               \
            const int gy_ = (y0_) + py_ - 1, gx_ = (x0_) + px_ - 1;                              \
            const bool in_ = (interior_) | (((unsigned)gy_ < (unsigned)H) & ((unsigned)gx_ < (unsigned)W));   /* (no short circuit: no branches) */ \
            const int off_ = (org_) + py_ * wbytes + px_ * 256 + c16;                            \
            K1_HALO_LOAD(stg[k_], rs_, in_ ? off_ : 0x7fffffff);                                  \
        }                                                                                        \
    } while (0)
#ifdef K1_X_NOLOAD    /* timing experiments only (wrong results on purpose; tools/variant_run_r4.sh) */
#define K1_HALO_LOAD(dst_, rs_, off_) do { dst_ = f32x4{(float)(off_), 0.25f, -0.5f, 1.5f}; } while (0)
#else
#define K1_HALO_LOAD(dst_, rs_, off_) do { dst_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, off_, 0, CS_HALO_AUX)); } while (0)
#endif
#ifdef K1_X_NOCOMMIT  /* timing experiments only: no split arithmetic, no ds_write of the halo (the MFMAs run on stale LDS) */
#define K1_COMMIT1(k_, buf_) do { if (stg[k_].x == 1.2345e30f) *reinterpret_cast<f32x4*>(cs_smem) = stg[k_]; } while (0)
#else
#define K1_COMMIT1(k_, buf_)                                                                     \
    do {                                                                                         \
        u32x2 hi_, lo2_;                                                                         \
        split4(stg[k_], hi_, lo2_, nscale);                                                      \
        const int lo_ = lpk[k_] & 0xffff;                                                        \
        *reinterpret_cast<u32x2*>(cs_smem + (buf_) * TILE_BYTES + lo_) = hi_;                    \
        *reinterpret_cast<u32x2*>(cs_smem + (buf_) * TILE_BYTES + (lo_ ^ 64)) = lo2_;            \
    } while (0)
#endif

    int paddr[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = (lane & 31) + kx;
        paddr[kx] = ((2 * rp) * CS_IW + col) * 128 + ((((lane >> 5)) ^ ((col >> 1) & 7)) << 4);
    }
    const int lo_xor = 4 << 4;
    const unsigned char* const wlane = wl + mt * 2048 + lane * 16;
    const int kh = lane >> 5;
    const int ech = 32 * mt + 16 * kh;                              // register r of a lane = channel ech + r (both convolutions)
    f32x16 accm[2], accc[2], base_m[2];                             // [output row]: hi.hi / cross products (x 2^11) of conv1_i; conv10_i of the chain
    // the finished tile's 256 pixel lines as this thread's 8 pieces of 16 bytes (row n: piece id = i * 512 + tid -> pixel id >> 4 of
    // the row's 128, chunk id & 15): picked up from the scratch in the serial phase, stored one by one under the NEXT tile's MFMAs -
    // a CU moves store data at ~16 B per clock (measured: 2 k cycles for the 32 KB of a row), far too slow to wait for
    u32x4 held[8];
    int hx0 = 0, hy0 = 0, hitem = 0;                                // ... and where they go
    bool hpend = false;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accm[n][r] = 0.f;
            accc[n][r] = 0.f;
        }
    const float slope = p.act ? 0.2f : 1.0f;

    // weight replacement: a slot (24 KB, one column tap of the other channel half) travels L2 -> LDS by LDS-DMA, 3 instructions of
    // 1 KB per wave, behind the barrier that frees it; complete (fence load, see conv_sf.hip) before the barrier in front of its first read
    const unsigned ldsw = (unsigned)(uintptr_t)wl;
    const int wvoff = wave * 1024 + lane * 16;
#define K1_DMA_W(half_, slot_)                                                                   \
    do {                                                                                         \
        const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wpack), 0, 2 * CS_W_BYTES, 0x00020000); \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_)                                         \
            k1_dma16(rw_, ldsw + (slot_) * CS_SLOT_BYTES + (wave + 8 * k_) * 1024, wvoff, (half_) * CS_W_BYTES + (slot_) * CS_SLOT_BYTES + k_ * 8192); \
    } while (0)
#ifdef PFNL_X_NOWSTREAM   /* timing experiment only (wrong results on purpose): no weight replacement - what does the 73 KB per tile from L2 cost? */
#define K1_DMA_WX(half_, slot_) do {} while (0)
#else
#define K1_DMA_WX(half_, slot_) K1_DMA_W(half_, slot_)
#endif
#define K1_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // ---- a finished row (n) of the workgroup's tile: leaky-relu, split, 128 pixel lines in `scratch`; optionally conv10_i's
    // products of those lines; the lines to HBM.  Pixel slot pp = row pair * 32 + column; 16-byte chunk c of a line (conv_split16.h:
    // c = 8 M + 4 part + channel group) sits in slot c ^ (pp & 15): conflict-free for the dump (lanes = consecutive pixels, one
    // chunk), for the operand reads (the same) and for the line read-back (16 lanes = the 16 chunks of one pixel).
    unsigned char* const scratch = cs_smem + TILE_BYTES;            // unit B's halo buffer: free behind the tile's closing barrier
    struct RowHalves { u32x2 hi[4], lo[4]; };                       // one output row of a wave (16 channels of a pixel per lane) as binary16 pairs
    auto quarter_prep = [&](RowHalves& h, const f32x16& m, const f32x16& c, int q, bool fold) __attribute__((always_inline)) {
        // channels ech + 4q .. + 3: (cross terms folded in,) leaky-relu, split - 4 VALU per value
        f32x4 t = {m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]};
        // (the fold reads MFMA results: compiler-generated arithmetic, which gets the XDL -> VALU wait states - inline asm would not)
        if (fold) t += f32x4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]} * CS_ISCALE;
        else asm volatile("" : "+v"(t));                            // (not folded: t = a copy of registers a VALU wrote - base_m after its own fold)
        const f32x4 st = mul4s(t, slope);
        asm("v_max_f32 %0, %1, %2" : "=v"(t.x) : "v"(t.x), "v"(st.x));
        asm("v_max_f32 %0, %1, %2" : "=v"(t.y) : "v"(t.y), "v"(st.y));
        asm("v_max_f32 %0, %1, %2" : "=v"(t.z) : "v"(t.z), "v"(st.z));
        asm("v_max_f32 %0, %1, %2" : "=v"(t.w) : "v"(t.w), "v"(st.w));
        split4(t, h.hi[q], h.lo[q], nscale);
    };
    const int pp = rp * 32 + (lane & 31);
    const int sw = pp & 15;
    unsigned char* const pl = scratch + pp * 256;
    auto row_dump = [&](const RowHalves& h) __attribute__((always_inline)) {
        const int c0 = 8 * mt + 2 * kh;                             // hi chunk of channels ech .. ech+7; +1: ech+8 .. +15; +4: the lo' chunks
        *reinterpret_cast<u32x4*>(pl + ((c0 ^ sw) << 4)) = u32x4{h.hi[0].x, h.hi[0].y, h.hi[1].x, h.hi[1].y};
        *reinterpret_cast<u32x4*>(pl + (((c0 + 1) ^ sw) << 4)) = u32x4{h.hi[2].x, h.hi[2].y, h.hi[3].x, h.hi[3].y};
        *reinterpret_cast<u32x4*>(pl + (((c0 + 4) ^ sw) << 4)) = u32x4{h.lo[0].x, h.lo[0].y, h.lo[1].x, h.lo[1].y};
        *reinterpret_cast<u32x4*>(pl + (((c0 + 5) ^ sw) << 4)) = u32x4{h.lo[2].x, h.lo[2].y, h.lo[3].x, h.lo[3].y};
    };
    // One pass over the 128 lines in the scratch: (a) this thread's 4 pieces of them -> `pc` (stored later); (b) conv10_i, frame f:
    // base_m[n][cout][pixel] += W10_f[cout][k] X[k][pixel], k-step q = (M, h): channels 32 M + 16 kh + 8 h + e - 3 MFMAs per k-step,
    // ordered so that none waits for the one before it; (c) `fill(q)`: arithmetic of the caller that rides in the shadow of the MFMAs
    // (a wave issues in order: it has to sit BETWEEN them in the stream).
    auto row_pass = [&](int n, auto mm, const u32x4 (&w10)[8], u32x4* pc, auto&& fill) __attribute__((always_inline)) {
        constexpr bool MM = decltype(mm)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = i * CS_THREADS + tid, ppx = id >> 4, c = id & 15;
            pc[i] = *reinterpret_cast<const u32x4*>(scratch + ppx * 256 + ((c ^ (ppx & 15)) << 4));
        }
        if constexpr (MM) {
            h8 bh[4], bo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cb = 8 * (q >> 1) + 2 * kh + (q & 1);
                bh[q] = *reinterpret_cast<const h8*>(pl + ((cb ^ sw) << 4));
                bo[q] = *reinterpret_cast<const h8*>(pl + (((cb + 4) ^ sw) << 4));
            }
            f32x16 cross;
#pragma unroll
            for (int r = 0; r < 16; ++r) cross[r] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const h8 ah = __builtin_bit_cast(h8, w10[2 * q]), ao = __builtin_bit_cast(h8, w10[2 * q + 1]);
                cross = mfma_f16(ao, bh[q], cross);
                base_m[n] = mfma_f16(ah, bh[q], base_m[n]);
                cross = mfma_f16(ah, bo[q], cross);
                __builtin_amdgcn_sched_barrier(0);
                fill(q);
                __builtin_amdgcn_sched_barrier(0);
            }
            base_m[n] += cross * CS_ISCALE;                         // (MFMA results: compiler-generated, hazard-padded)
        }
    };
    // piece j (0..7: row j >> 2, piece j & 3) of the held tile -> HBM; nothing held: out of range (dropped)
    auto held_store = [&](int j, float* base_ptr) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(base_ptr + (size_t)hitem * H * W * 64, 0, item_bytes, 0x00020000);
        const int id = (j & 3) * CS_THREADS + tid, ppx = id >> 4, c = id & 15;
        const int gy = hy0 + 2 * (ppx >> 5) + (j >> 2), gx = hx0 + (ppx & 31);   // rows past the image: past the end of the resource
        const int off = (gy * W + gx) * 256 + c * 16;
        buffer_store_b128_guarded<K1_STORE_AUX>(held[j], rsO, (hpend & (gx < W)) ? off : 0x7fffffff, 0);
    };
    auto base_init = [&](bool zero) __attribute__((always_inline)) {   // conv10_i's bias: the initial value of a chain's sum (parts > 0 of a split chain: 0)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) base_m[n][r] = zero ? 0.f : bl2[ech + r];
    };

    // ---- prologue: halo of unit 0 -> buffer 0; weights of half 0, both biases -> LDS ----------------------------------------
    int c_item, c_y0, c_x0, n_item, n_y0, n_x0, fch, n_f;           // fch / n_f: frame of its chain the current / the next tile is
    K1_TILE(0, c_item, c_y0, c_x0, fch);
    n_item = c_item;
    n_y0 = c_y0;
    n_x0 = c_x0;
    n_f = fch;
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in) + (size_t)c_item * H * W * 64, 0, item_bytes, 0x00020000);
        const int org = ((c_y0 - 1) * W + c_x0 - 1) * 256;          // (unit 0: channel half 0)
        [[maybe_unused]] const bool interior = c_y0 > 0 && c_y0 + CS_IH - 1 <= H && c_x0 > 0 && c_x0 + CS_IW - 1 <= W;
        [[maybe_unused]] unsigned fence0 = 0;
        if constexpr (ISF) {
            K1_DMA_HALO(rs, org, false, c_y0, c_x0, 0);
            fence0 = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);   // fence load (conv_sf.hip): vmcnt is in order
        } else {
            K1_REQUEST_ALL(rs, org, interior, c_y0, c_x0);
        }
#pragma unroll
        for (int k = 0; k < CS_W_BYTES / 16 / CS_THREADS; ++k) reinterpret_cast<u32x4*>(wl)[k * CS_THREADS + tid] = w0reg[k];
        if (tid < 128) bl[tid] = bias_r;                            // (bl2 = bl + 64)
        if constexpr (ISF) {
            asm volatile("" ::"v"(fence0));                         // the DMAs have landed
        } else {
#pragma unroll
            for (int k = 0; k < CS_ITERS; ++k) K1_COMMIT1(k, 0);
        }
    }
    __syncthreads();
    base_init(SPLIT && nfull_tiles == 0 && sp_r > 0);

    for (int kt = 0; kt < nt; ++kt) {
        const int half_a = kt & 1;
        auto unit = [&](auto par) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par)::value;
            constexpr int cb = PAR;
            const unsigned char* const tile = cs_smem + cb * TILE_BYTES;
            h8 X[4][2], Wv[2][2];
            CS_STAMP();                                             // 0 / 1: unit A / B start
#define CS_PX(g_, r_, part_) (*reinterpret_cast<const h8*>(tile + (paddr[(g_) >> 1] ^ (((part_) ? lo_xor : 0) | (((g_) & 1) << 5))) + (r_) * (CS_IW * 128)))
#define CS_WT(g_, ky_, part_) (*reinterpret_cast<const h8*>(wlane + (((g_) * 3 + (ky_)) << 12) + ((part_) << 10)))
            X[0][0] = CS_PX(0, 0, 0);
            X[0][1] = CS_PX(0, 0, 1);
            X[1][0] = CS_PX(0, 1, 0);
            X[1][1] = CS_PX(0, 1, 1);
            Wv[0][0] = CS_WT(0, 0, 0);
            Wv[0][1] = CS_WT(0, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            // unit B: column tap 2 of ITS half (the slot was in use until unit A's closing barrier); read from group 4 on, behind b0
            [[maybe_unused]] unsigned fence_w = 0;
            const __amdgpu_buffer_rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, 256, 0x00020000);
            if constexpr (PAR == 1) {
                K1_DMA_WX(half_a ^ 1, 2);
                fence_w = __builtin_amdgcn_raw_buffer_load_b32(rsf, 0, 0, 0);
            }
            // the NEXT unit's halo: unit A asks for the other half of ITS tile, unit B for the first half of the next tile
            const int q_item = PAR == 0 ? c_item : n_item, y0q = PAR == 0 ? c_y0 : n_y0, x0q = PAR == 0 ? c_x0 : n_x0;
            const int q_half = half_a ^ 1;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.in) + (size_t)q_item * H * W * 64, 0, item_bytes, 0x00020000);
            const int org = ((y0q - 1) * W + x0q - 1) * 256 + q_half * 128;
            const bool interior = y0q > 0 && y0q + CS_IH - 1 <= H && x0q > 0 && x0q + CS_IW - 1 <= W;
            if constexpr (!ISF) K1_REQUEST_ALL(rs, org, interior, y0q, x0q);
            [[maybe_unused]] f32x16 bias16;
            if constexpr (PAR == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + ech + 4 * q);
                    bias16[4 * q] = b4.x;
                    bias16[4 * q + 1] = b4.y;
                    bias16[4 * q + 2] = b4.z;
                    bias16[4 * q + 3] = b4.w;
                }
            }
            // conv10_i's A operands of this frame: [f][k-step q][g = mt][hi / lo'][lane] x 16 B from L2, requested two sub-steps before the
            // tile is complete (the halo staging registers are free from there on) so that their latency passes under MFMAs
            [[maybe_unused]] u32x4 w10[8];
            auto substep = [&](auto sc) __attribute__((always_inline)) {
                constexpr int S = decltype(sc)::value;
                constexpr int g = S / 3, ky = S % 3;
                if constexpr (PAR == 1 && S == K1_W10_AT) {
                    const u32x4* src = reinterpret_cast<const u32x4*>(p.wpack2) + ((size_t)fch * 16 + 2 * mt) * 64 + lane;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        w10[2 * q] = src[(4 * q) * 64];
                        w10[2 * q + 1] = src[(4 * q + 1) * 64];
                    }
                }
                if constexpr (PAR == 0 && S >= 1 && S <= 8) held_store(S - 1, p.out);   // the previous tile's lines, one piece per sub-step
                if constexpr (ky == 0) {
                    if constexpr (g == 2) {
                        if constexpr (PAR == 1) asm volatile("" ::"v"(fence_w));   // tap 2 of this unit's weights has landed
                        K1_BARRIER();                               // b0: column tap 0 of the weights consumed (unit B: tap 2 complete)
                        if constexpr (PAR == 0) K1_DMA_WX(half_a ^ 1, 0);
                        if constexpr (ISF) {
                            // the next unit's halo by DMA, behind b0: unit A's target is the SCRATCH of the previous tile's serial phase, whose
                            // last line reads (row_pass(1)) no barrier followed - every wave has passed them here.  12 sub-steps to land.
                            K1_DMA_HALO(rs, org, interior, y0q, x0q, cb ^ 1);
                            if constexpr (PAR == 1) fence_w = __builtin_amdgcn_raw_buffer_load_b32(rsf, 0, 0, 0);   // (unit A: the fence behind b1 covers it)
                        }
                    }
                    if constexpr (g == 3 && PAR == 0) {
                        const int kn = min(kt + 1, nt - 1);         // decode the next tile (past the end: this one again - a harmless re-read)
                        K1_TILE(kn, n_item, n_y0, n_x0, n_f);
                    }
                    if constexpr (g == 4) {
                        K1_BARRIER();                               // b1: column tap 1 consumed
                        if constexpr (PAR == 0) {
                            K1_DMA_WX(half_a ^ 1, 1);
                            fence_w = __builtin_amdgcn_raw_buffer_load_b32(rsf, 0, 0, 0);   // covers taps 0 and 1 of the next unit's weights
                        }
                        if constexpr (!ISF) {
#pragma unroll
                            for (int k = 0; k < CS_ITERS; ++k) asm volatile("" : "+v"(lpk[k]));
#pragma unroll
                            for (int k = 0; k < CS_ITERS / 2; ++k) K1_COMMIT1(k, cb ^ 1);
                        }
                    }
                    if constexpr (g == 5 && !ISF) {
#pragma unroll
                        for (int k = CS_ITERS / 2; k < CS_ITERS; ++k) K1_COMMIT1(k, cb ^ 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#if PFNL_S16_SPREAD
                // the next sub-step's operand reads ride BETWEEN this sub-step's MFMAs (M w w M x x M x x M M M, every position pinned):
                // as a clump in front of the six MFMAs they are ~100 cycles in which this wave's share of the matrix pipe idles
                // (tools/ubench/conv_core: 5.66 -> 5.02 us per unit for the bare core)
                {
                    constexpr int S1 = S + 1, g1 = S1 / 3, ky1 = S1 % 3;
                    constexpr bool RD = S < 17;
                    constexpr bool FIRST = PAR == 0 && S == 0;
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    const h8 wh = Wv[S & 1][0], wo = Wv[S & 1][1];
                    accm[0] = mfma_f16(wh, X[ky][0], FIRST ? bias16 : accm[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (RD) {
                        Wv[S1 & 1][0] = CS_WT(g1, ky1, 0);
                        Wv[S1 & 1][1] = CS_WT(g1, ky1, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    accm[1] = mfma_f16(wh, X[ky + 1][0], FIRST ? bias16 : accm[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (RD) {
                        if constexpr (ky1 == 0) {
                            X[0][0] = CS_PX(g1, 0, 0);
                            X[0][1] = CS_PX(g1, 0, 1);
                        } else {
                            X[ky1 + 1][0] = CS_PX(g1, ky1 + 1, 0);
                            X[ky1 + 1][1] = CS_PX(g1, ky1 + 1, 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    accc[0] = mfma_f16(wo, X[ky][0], FIRST ? zero : accc[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (RD && ky1 == 0) {
                        X[1][0] = CS_PX(g1, 1, 0);
                        X[1][1] = CS_PX(g1, 1, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    accc[1] = mfma_f16(wo, X[ky + 1][0], FIRST ? zero : accc[1]);
                    accc[0] = mfma_f16(wh, X[ky][1], accc[0]);
                    accc[1] = mfma_f16(wh, X[ky + 1][1], accc[1]);
                }
#else
                if constexpr (S < 17) {
                    constexpr int S1 = S + 1, g1 = S1 / 3, ky1 = S1 % 3;
                    Wv[S1 & 1][0] = CS_WT(g1, ky1, 0);
                    Wv[S1 & 1][1] = CS_WT(g1, ky1, 1);
                    if constexpr (ky1 == 0) {
                        X[0][0] = CS_PX(g1, 0, 0);
                        X[0][1] = CS_PX(g1, 0, 1);
                        X[1][0] = CS_PX(g1, 1, 0);
                        X[1][1] = CS_PX(g1, 1, 1);
                    } else {
                        X[ky1 + 1][0] = CS_PX(g1, ky1 + 1, 0);
                        X[ky1 + 1][1] = CS_PX(g1, ky1 + 1, 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const h8 wh = Wv[S & 1][0], wo = Wv[S & 1][1];
                if constexpr (PAR == 0 && S == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    accm[0] = mfma_f16(wh, X[ky][0], bias16);
                    accm[1] = mfma_f16(wh, X[ky + 1][0], bias16);
                    accc[0] = mfma_f16(wo, X[ky][0], zero);
                    accc[1] = mfma_f16(wo, X[ky + 1][0], zero);
                } else {
                    accm[0] = mfma_f16(wh, X[ky][0], accm[0]);
                    accm[1] = mfma_f16(wh, X[ky + 1][0], accm[1]);
                    accc[0] = mfma_f16(wo, X[ky][0], accc[0]);
                    accc[1] = mfma_f16(wo, X[ky + 1][0], accc[1]);
                }
                accc[0] = mfma_f16(wh, X[ky][1], accc[0]);
                accc[1] = mfma_f16(wh, X[ky + 1][1], accc[1]);
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
#undef CS_PX
#undef CS_WT
            if constexpr (PAR == 1) {
                // The tile leaves (see the header).  Row 0's leaky-relu / split needs nothing but this wave's registers: in FRONT of the
                // closing barrier, where the wave of a SIMD that finishes first would only wait for the other one; row 1's rides between
                // row 0's conv10 MFMAs.
                RowHalves h0, h1;
#pragma unroll
                for (int q = 0; q < 4; ++q) quarter_prep(h0, accm[0], accc[0], q, true);
                CS_STAMP();                                         // 2: sub-steps done, row 0 prepared
                if constexpr (ISF) asm volatile("" ::"v"(fence_w)); // the next tile's first halo has landed
                K1_BARRIER();                                       // b2: this unit's buffer (the scratch) is free, the next unit's is complete
                CS_STAMP();                                         // 3: past b2
                row_dump(h0);
                CS_STAMP();                                         // 4: row 0 dumped
                K1_BARRIER();                                       // the 128 lines of row 0 are complete
                CS_STAMP();                                         // 5
                row_pass(0, std::true_type{}, w10, &held[0], [&](int q) __attribute__((always_inline)) { quarter_prep(h1, accm[1], accc[1], q, true); });
                CS_STAMP();                                         // 6: row 0's lines picked up, its products issued, row 1 prepared
                K1_BARRIER();                                       // ... and read
                CS_STAMP();                                         // 7
                row_dump(h1);
                K1_BARRIER();
                CS_STAMP();                                         // 8
                row_pass(1, std::true_type{}, w10, &held[4], [&](int) __attribute__((always_inline)) {});
                CS_STAMP();                                         // 9
                hx0 = c_x0;
                hy0 = c_y0;
                hitem = c_item;
                hpend = true;
                const bool is_part = SPLIT && kt >= nfull_tiles;    // (wave-uniform) a tile of this workgroup's part of a split chain
                const bool last = fch + 1 == (is_part ? sp_f1 : gT);
                if (last && is_part) {                              // a PART's sum leaves as raw fp32 (no activation: c10_finalize_kernel adds the parts up)
#pragma unroll
                    for (int j = 0; j < 8; ++j) held_store(j, p.out);
                    float* const pt = p.partial + (size_t)slot * (CS_TH * CS_TW * 64) + (size_t)(2 * rp * CS_TW + (lane & 31)) * 64 + ech;
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<f32x4*>(pt + n * (CS_TW * 64) + 4 * q) =
                                f32x4{base_m[n][4 * q], base_m[n][4 * q + 1], base_m[n][4 * q + 2], base_m[n][4 * q + 3]};
                    hpend = false;                                  // (the part is this workgroup's last item)
                } else if (last) {                                  // the chain's sum -> `base` (out2), and back to its initial value
                    // (once per chain: stored at once; the frame tile's own lines leave first - `held` is reused)
#pragma unroll
                    for (int j = 0; j < 8; ++j) held_store(j, p.out);
                    hitem = c_item / gT;
                    RowHalves hb0, hb1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        quarter_prep(hb0, base_m[0], base_m[0], q, false);
                        quarter_prep(hb1, base_m[1], base_m[1], q, false);
                    }
                    K1_BARRIER();
                    row_dump(hb0);
                    K1_BARRIER();
                    row_pass(0, std::false_type{}, w10, &held[0], [&](int) __attribute__((always_inline)) {});
                    K1_BARRIER();
                    row_dump(hb1);
                    K1_BARRIER();
                    row_pass(1, std::false_type{}, w10, &held[4], [&](int) __attribute__((always_inline)) {});
#pragma unroll
                    for (int j = 0; j < 8; ++j) held_store(j, p.out2);
                    hpend = false;
                    base_init(SPLIT && kt + 1 == nfull_tiles && sp_r > 0);   // (the next item is a part > 0 of a split chain: it starts from 0)
                }
                if constexpr (SPLIT) fch = n_f;
                else fch = last ? 0 : fch + 1;
                c_item = n_item;
                c_y0 = n_y0;
                c_x0 = n_x0;
                // (no barrier here: the scratch is next written by the halo commit of the coming unit, behind two of its barriers)
            } else {
                asm volatile("" ::"v"(fence_w));                    // taps 0, 1 of unit B's weights have landed
                K1_BARRIER();                                       // b2
            }
        };
        unit(std::integral_constant<int, 0>{});
        unit(std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) held_store(j, p.out);               // the last tile's lines (nothing held: dropped)
#undef K1_TILE
#undef K1_DMA_HALO
#undef K1_REQUEST_ALL
#undef K1_COMMIT1
#undef K1_BARRIER
#undef K1_DMA_W
}

int conv_split16_grid() {
    const int ncu = device_cu_count();
    return ncu >= 8 ? ncu / 8 * 8 : 8;                              // whole XCDs; surplus workgroups exit at once
}

// the geometry of a split-chain launch must be the one the kernels assume: whole rounds of the grid in front, one part per workgroup behind
static bool split_params_ok(const ConvSplitParams& p, int grid) {
    if (p.split_s == 0) return true;
    const int T = p.add_div;
    const long long nchains = (long long)((p.W + CS_TW - 1) / CS_TW) * ((p.H + CS_TH - 1) / CS_TH) * (p.items / T);
    if (p.split_s < 2 || p.split_q < 1 || p.n_full < 0 || p.n_full % grid || p.n_full >= nchains) return false;
    if ((long long)p.split_s * p.split_q < T || (long long)(p.split_s - 1) * p.split_q >= T) return false;   // every part non-empty, together the T frames
    return (nchains - p.n_full) * p.split_s <= grid;
}

// ---- split chains: the parts' raw conv10_i sums -> `base` in the split format (see conv3x3_c1c10_kernel).  One workgroup per split chain;
// thread = (pixel, 4-channel group) of its 8 x 32 tile, the parts added in the fixed order r = 0 .. s-1 (part 0 carries the bias).
__global__ __launch_bounds__(256) void c10_finalize_kernel(ConvSplitParams p) {
    const int H = p.H, W = p.W;
    const int tiles_x = (W + CS_TW - 1) / CS_TW, tiles_y = (H + CS_TH - 1) / CS_TH;
    const int per_item = tiles_x * tiles_y;
    const int j = blockIdx.x >> 3, ch = p.n_full + j;                // 8 workgroups per split chain: one tile row (32 pixels x 16 groups = 512 ids) each
    const int clip = ch / per_item, sp = ch - clip * per_item;
    const int ty = sp / tiles_x, y0 = ty * CS_TH, x0 = (sp - ty * tiles_x) * CS_TW;
    const float slope = p.act ? 0.2f : 1.0f;
    const int S = p.split_s;                                        // 2 .. 7
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int id = (blockIdx.x & 7) * 512 + it * 256 + threadIdx.x;
        const int pix = id >> 4, c4 = id & 15;
        const int y = y0 + pix / CS_TW, x = x0 + (pix & (CS_TW - 1));
        const float* src = p.partial + ((size_t)j * S * (CS_TH * CS_TW) + pix) * 64 + c4 * 4;
        f32x4 part[7];                                              // every part's piece requested before any is used (a loop of dependent
#pragma unroll                                                      // loads made this 16 KB-per-workgroup kernel 17 us long)
        for (int r = 0; r < 7; ++r) part[r] = r < S ? *reinterpret_cast<const f32x4*>(src + (size_t)r * (CS_TH * CS_TW * 64)) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 v = part[0];
#pragma unroll
        for (int r = 1; r < 7; ++r)
            if (r < S) v += part[r];                                // fixed order r = 0 .. S-1 (wave-uniform S: no divergence)
        if (p.out) v += *reinterpret_cast<const f32x4*>(p.bias + c4 * 4);   // convmerge1's parts carry no bias (conv10_i's part 0 does)
        v.x = fmaxf(v.x, slope * v.x);
        v.y = fmaxf(v.y, slope * v.y);
        v.z = fmaxf(v.z, slope * v.z);
        v.w = fmaxf(v.w, slope * v.w);
        const h4 hi = __builtin_convertvector(v, h4);
        const h4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * CS_SCALE, h4);   // exact before the one rounding: = split4
        if (p.out) {                                                // convmerge1: fp32 [clip][y][x][64]
            if (y < H && x < W) *reinterpret_cast<f32x4*>(p.out + (((size_t)clip * H + y) * W + x) * 64 + c4 * 4) = v;
        } else if (y < H && x < W) {
            // split format: pixel 256 B = [channel half M][hi 32 | lo' 32] binary16; channels 4 c4 .. + 3 -> half c4 >> 3, position (c4 & 7) * 4
            unsigned char* dst = reinterpret_cast<unsigned char*>(p.out2) + (((size_t)clip * H + y) * W + x) * 256 + (c4 >> 3) * 128 + (c4 & 7) * 8;
            *reinterpret_cast<h4*>(dst) = hi;
            *reinterpret_cast<h4*>(dst + 64) = lo;
        }
    }
}

hipError_t launch_c10_finalize(const ConvSplitParams& p, hipStream_t s) {
    if (!p.partial || (!p.out2 == !p.out) || (p.out && !p.bias) || p.items < 1 || p.H < 1 || p.W < 1 || p.add_div < 1 || p.items % p.add_div) return hipErrorInvalidValue;
    const int grid = conv_split16_grid();
    if (p.split_s < 2 || !split_params_ok(p, grid)) return hipErrorInvalidValue;
    const long long nchains = (long long)((p.W + CS_TW - 1) / CS_TW) * ((p.H + CS_TH - 1) / CS_TH) * (p.items / p.add_div);
    if (p.split_s > 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(c10_finalize_kernel, dim3((unsigned)(nchains - p.n_full) * 8), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_conv3x3_c1c10(const ConvSplitParams& p, hipStream_t s) {
    if (!p.in || !p.wpack || !p.bias || !p.wpack2 || !p.bias2 || !p.out || !p.out2 || p.items < 1 || p.H < 1 || p.W < 1) return hipErrorInvalidValue;
    if (p.add_div < 1 || p.add_div > 7 || p.items % p.add_div || p.addend || p.resid || p.accum) return hipErrorInvalidValue;
    if ((long long)p.H * p.W * 256 >= 0x7fffffffLL) return hipErrorInvalidValue;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    const int ncu = device_cu_count();
    if (!ncu) return hipErrorUnknown;
    const int grid = ncu >= 8 ? ncu / 8 * 8 : 8;
    if (!split_params_ok(p, grid) || (p.split_s && !p.partial)) return hipErrorInvalidValue;
    static std::atomic<int> attr_dev[64][4];
    const int isf = p.in_sf ? 1 : 0;                                // `in` is the split-format copy of inp0 (conv3x3_sf_chain_kernel<true, .>)
    const int var = isf + (p.split_s ? 2 : 0);
    const void* const fns[4] = {reinterpret_cast<const void*>(conv3x3_c1c10_kernel<false, false>), reinterpret_cast<const void*>(conv3x3_c1c10_kernel<true, false>),
                                reinterpret_cast<const void*>(conv3x3_c1c10_kernel<false, true>), reinterpret_cast<const void*>(conv3x3_c1c10_kernel<true, true>)};
    const int lds = isf ? K1_SF_LDS_BYTES : K1_LDS_BYTES;
    if (!attr_dev[dev][var]) {
        hipError_t e = hipFuncSetAttribute(fns[var], hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_dev[dev][var] = 1;
    }
    switch (var) {
        case 0: hipLaunchKernelGGL((conv3x3_c1c10_kernel<false, false>), dim3(grid), dim3(CS_THREADS), lds, s, p); break;
        case 1: hipLaunchKernelGGL((conv3x3_c1c10_kernel<true, false>), dim3(grid), dim3(CS_THREADS), lds, s, p); break;
        case 2: hipLaunchKernelGGL((conv3x3_c1c10_kernel<false, true>), dim3(grid), dim3(CS_THREADS), lds, s, p); break;
        default: hipLaunchKernelGGL((conv3x3_c1c10_kernel<true, true>), dim3(grid), dim3(CS_THREADS), lds, s, p); break;
    }
    return hipGetLastError();
}

hipError_t launch_conv3x3_split16(const ConvSplitParams& p, hipStream_t s) {
    if (!p.in || !p.wpack || !p.bias || !p.out || p.items < 1 || p.H < 1 || p.W < 1) return hipErrorInvalidValue;
    if ((p.addend == nullptr) != (p.resid == nullptr) || (p.addend && (p.add_div < 1 || p.items % p.add_div))) return hipErrorInvalidValue;
    if (p.accum && (p.addend || p.add_div < 1 || p.items % p.add_div)) return hipErrorInvalidValue;
    if ((long long)p.H * p.W * 256 >= 0x7fffffffLL) return hipErrorInvalidValue;
    const int ncu = device_cu_count();
    if (!ncu) return hipErrorUnknown;
    const int grid = ncu >= 8 ? ncu / 8 * 8 : 8;                    // whole XCDs; surplus workgroups exit at once
    static std::atomic<int> attr_dev[64][4];                               // the attribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (p.out_sf && (p.accum || p.addend)) return hipErrorInvalidValue;
    if (p.split_s) {                                                       // split chains: convmerge1's accumulating launch only
        if (!p.accum || !p.partial || !split_params_ok(p, grid)) return hipErrorInvalidValue;
        static std::atomic<int> attr_split[64];
        if (!attr_split[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split16_kernel<2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS_BYTES);
            if (e != hipSuccess) return e;
            attr_split[dev] = 1;
        }
        hipLaunchKernelGGL((conv3x3_split16_kernel<2, false, true>), dim3(grid), dim3(CS_THREADS), CS_LDS_BYTES, s, p);
        return hipGetLastError();
    }
    const int mode = p.out_sf ? 3 : p.accum ? 2 : p.addend ? 1 : 0;
    const void* fn = mode == 3 ? reinterpret_cast<const void*>(conv3x3_split16_kernel<0, true>)
                   : mode == 2 ? reinterpret_cast<const void*>(conv3x3_split16_kernel<2>)
                   : mode     ? reinterpret_cast<const void*>(conv3x3_split16_kernel<1>) : reinterpret_cast<const void*>(conv3x3_split16_kernel<0>);
    if (!attr_dev[dev][mode]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_dev[dev][mode] = 1;
    }
    if (mode == 3) hipLaunchKernelGGL((conv3x3_split16_kernel<0, true>), dim3(grid), dim3(CS_THREADS), CS_LDS_BYTES, s, p);
    else if (mode == 2) hipLaunchKernelGGL(conv3x3_split16_kernel<2>, dim3(grid), dim3(CS_THREADS), CS_LDS_BYTES, s, p);
    else if (mode) hipLaunchKernelGGL(conv3x3_split16_kernel<1>, dim3(grid), dim3(CS_THREADS), CS_LDS_BYTES, s, p);
    else hipLaunchKernelGGL(conv3x3_split16_kernel<0>, dim3(grid), dim3(CS_THREADS), CS_LDS_BYTES, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// host-side packing
static int split_row_channel(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }   // MFMA row -> channel in the tile (see the kernel)

static uint16_t f16_bits(float f) {
    const _Float16 h = (_Float16)f;                                 // round to nearest even
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}

size_t conv3x3_split16_pack_halfs() { return 2 * (size_t)CS_W_BYTES / 2; }   // both channel halves, in 16-bit units

// HWIO [3,3,cin_total,cout] rows [cin_begin, cin_begin+64) -> [half][kx][ks][ky][m][part][lane][e]:
// W[ky][kx][cin_begin + 32 half + 16 ks + 8 (lane>>5) + e][32 m + row_channel(lane&31)], part 0 = f16(w), part 1 = f16((w - hi) 2^11)
// identity_rows: MFMA row / column r of the weight operand = output channel 32 m + r (conv_sf.hip: the weights are the B operand,
// a lane of the result owns one channel); otherwise the row -> channel map of conv3x3_split16_kernel (16 consecutive channels per lane)
void conv3x3_split16_pack_weights(const float* hwio, int cin_total, int cin_begin, uint16_t* dst, int cout, bool identity_rows) {
    for (int half = 0; half < 2; ++half)
        for (int kx = 0; kx < 3; ++kx)
            for (int ks = 0; ks < 2; ++ks)
                for (int ky = 0; ky < 3; ++ky)
                    for (int m = 0; m < 2; ++m)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int ci = cin_begin + 32 * half + 16 * ks + 8 * (lane >> 5) + e;
                                const int co = 32 * m + (identity_rows ? (lane & 31) : split_row_channel(lane & 31));
                                const float w = co < cout ? hwio[((size_t)(ky * 3 + kx) * cin_total + ci) * cout + co] : 0.f;
                                const _Float16 hi = (_Float16)w;
                                const float lo = (w - (float)hi) * CS_SCALE;
                                const size_t base = (size_t)half * (CS_W_BYTES / 2) + ((((size_t)(kx * 2 + ks) * 3 + ky) * 2 + m) * 2) * 512;
                                dst[base + lane * 8 + e] = f16_bits((float)hi);
                                dst[base + 512 + lane * 8 + e] = f16_bits(lo);
                            }
}

// The same kernel in the order of the v_mfma_f32_16x16x32_f16 kernels (conv_sf.hip, conv3x3_sf_chain16_kernel: A = weights, 16 output channels x 32 input
// channels per operand): [half][kx][ky][output tile c4][part][lane][e] = W[ky][kx][cin_begin + 32 half + 8 (lane >> 4) + e][16 c4 + (lane & 15)] - the 24 KB per
// column tap kx are again one LDS-DMA slice
void conv3x3_split16_pack_weights16(const float* hwio, int cin_total, int cin_begin, uint16_t* dst) {
    for (int half = 0; half < 2; ++half)
        for (int kx = 0; kx < 3; ++kx)
            for (int ky = 0; ky < 3; ++ky)
                for (int c4 = 0; c4 < 4; ++c4)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int ci = cin_begin + 32 * half + 8 * (lane >> 4) + e;
                            const int co = 16 * c4 + (lane & 15);
                            const float w = hwio[((size_t)(ky * 3 + kx) * cin_total + ci) * 64 + co];
                            const _Float16 hi = (_Float16)w;
                            const float lo = (w - (float)hi) * CS_SCALE;
                            const size_t base = (size_t)half * (CS_W_BYTES / 2) + ((((size_t)(kx * 3 + ky) * 4 + c4) * 2) * 512);
                            dst[base + lane * 8 + e] = f16_bits((float)hi);
                            dst[base + 512 + lane * 8 + e] = f16_bits(lo);
                        }
}

size_t conv1x1_c10_pack_halfs(int T) { return (size_t)T * 8192; }      // 16 KB per frame

// conv10_i for conv3x3_c1c10_kernel: HWIO [1,1,T*64,64] -> [f][k-step q = 2M + h][g][part][lane][e] =
// W[f*64 + 32M + 16(lane>>5) + 8h + e][32g + row_channel(lane&31)] (the A operand: MFMA row -> channel as in the 3x3 packs, so that a lane
// of the result holds 16 consecutive channels), part 0 = f16(w), part 1 = f16((w - hi) 2^11)
void conv1x1_c10_pack_weights(const float* hwio, int T, uint16_t* dst) {
    for (int f = 0; f < T; ++f)
        for (int q = 0; q < 4; ++q)
            for (int g = 0; g < 2; ++g)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int ci = f * 64 + 32 * (q >> 1) + 16 * (lane >> 5) + 8 * (q & 1) + e;
                        const int co = 32 * g + split_row_channel(lane & 31);
                        const float w = hwio[(size_t)ci * 64 + co];
                        const _Float16 hi = (_Float16)w;
                        const float lo = (w - (float)hi) * CS_SCALE;
                        const size_t base = (((((size_t)f * 4 + q) * 2 + g) * 2) * 64 + lane) * 8 + e;
                        dst[base] = f16_bits((float)hi);
                        dst[base + 512] = f16_bits(lo);
                    }
}

}  // namespace pfnl

#ifdef PFNL_S16_TIMING
extern "C" int pfnl_debug_read_s16_stamps(long long* host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pfnl::cs_dbg), n * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

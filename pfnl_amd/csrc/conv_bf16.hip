// bf16 trunk (option precision=bf16; BASELINE.json configs[3]: 1080p, bf16 activations + weights, fp32 accumulation):
// the 3x3 64->64 convolutions of the progressive-fusion blocks (conv1_i, both halves of conv2_i; reference
// model/pfnl.py:49-51, 66-71) and conv10_i (1x1, T*64 -> 64; :50, :67-68) on v_mfma_f32_32x32x16_bf16, 16x the f32
// MFMA rate.
//
// At that rate the matrix pipe no longer bounds the block: layer by layer the trunk moves 5.8 KB per LR pixel and
// block in bf16 (15 GB per 1080p forward) against 3 TFLOP of DIRECT 3x3 work - 1.2 ms at the 2.5 PFLOP/s dense peak,
// ~3 ms at the 4-5 TB/s this traffic mix reaches.  So the 3x3 kernel is the direct algorithm (a Winograd transform
// would cost more VALU time than the multiplies it saves) built around the memory stream:
//
//   * persistent workgroup per CU (512 threads = 2 waves per SIMD); the launch's packed weights (9 taps x 64 x 64
//     bf16 = 72 KB) are copied into LDS once and stay there;
//   * output tile 8 rows x 32 columns x 64 channels; its 10 x 34-pixel halo tile (128 B per pixel, 42.5 KB) is
//     double-buffered in LDS next to the weights (157 of 160 KB), 16-byte channel chunks XOR-swizzled by
//     (column >> 1) & 7 so that a ds_read_b128 of one chunk of 32 consecutive pixels is bank-conflict free
//     (SQ_LDS_BANK_CONFLICT = 0);
//   * MFMA roles: A = weights (rows = 32 output channels), B = pixels (columns = 32 pixels of a tile row): an
//     accumulator lane owns ONE pixel, and - the row -> channel map being free - 16 consecutive output channels;
//   * wave = (row pair, channel tile): 2 accumulators.  For a fixed column tap and 16-channel step the 4 halo rows of
//     the pair serve all 3 row taps of both output rows: 4 pixel reads + 3 weight reads feed 6 MFMAs, requested one
//     group ahead of the MFMAs that use them;
//   * work order: chains of the T frames of a clip at one spatial tile (fused launch: the shared-half addend pieces
//     are fetched once per chain), dealt out XCD by XCD so that neighbouring tiles share their halo rows in that
//     XCD's L2 - HBM reads went from 1.31x (conv1_i) / 1.55x (conv2_i) of the compulsory bytes to 1.02x;
//   * everything that is not an MFMA is sliced over the 12 groups of a tile (see the loop): the previous tile's
//     epilogue from copied accumulators into an LDS scratch, its stores as whole 128-byte lines, the next halo's
//     commit to the other LDS buffer, the requests for the halo after next - each wait is for a request issued most of
//     a tile period earlier, and the in-order vmcnt queue never has a store in front of a load that is waited for.
//
// conv2_i keeps the shared-`base` split of the fp32 path: one launch over `base` produces
// the raw shared half per clip (stored in bf16), the per-frame launch adds it before the activation.  The same kernel
// also runs conv10_i inside the conv1_i launch (mode 2) and convmerge1 (mode 3, accumulating over the frames of a clip
// with the weight pack replaced in LDS between frames): see the mode list above the kernel.
// Measured (1x7x270x480, rocprofv3): conv1_i 72 us, conv2_i per-frame half 87 us, shared half 16 us, conv10_i 29 us
// per block; 3.2-4.3 TB/s of HBM traffic; matrix pipe 38 % busy.  -DCB_X_NOMFMA / NOSTORE / NOLOAD are timing
// experiments (wrong results on purpose) used to find what bounds the kernel; -DPFNL_BF16_TIMING adds phase stamps.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"
#include "conv_bf16.h"

#ifndef CB_STORE_AUX
#define CB_STORE_AUX 0      // cache-policy bits of the output stores (sc0 sc1 = 17 is worth 2 % in conv_split16.hip; here it measured +0.5 %)
#endif

namespace pfnl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int CB_THREADS = 512;
constexpr int CB_TH = 8, CB_TW = 32;
constexpr int CB_IH = CB_TH + 2, CB_IW = CB_TW + 2;
constexpr int CB_TILE_BYTES = CB_IH * CB_IW * 128;                  // 43 520, two buffers
constexpr int CB_W_BYTES = 9 * 4 * 2 * 1024;                        // 73 728: [tap][kstep][channel tile][lane] x 16 B
constexpr int CB_LDS_BYTES = 2 * CB_TILE_BYTES + CB_W_BYTES + 64 * 4;   // 161 024 of 163 840
constexpr int CB_CHUNKS = CB_IH * CB_IW * 8;                        // 16-byte pieces of a halo tile
constexpr int CB_ITERS = (CB_CHUNKS + CB_THREADS - 1) / CB_THREADS; // 6

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef CB_X_NOMFMA
    c[0] += (float)a[0] * (float)b[0];
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 bf16x4_to_f32(u32x2 v) {
    return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                 __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)};
}
// leaky_relu(0.2) or identity as max(v, slope v), slope 0.2 / 1: two packed multiplies + four v_max per 4 values, no branch
// (fmaxf adds a canonicalising v_max x,x,x per element: 3 VALU per value with the multiply)
__device__ __forceinline__ f32x4 lrelu4(f32x4 v, float slope) {
    const f32x4 sv = v * slope;
    asm("v_max_f32 %0, %1, %2" : "=v"(v.x) : "v"(v.x), "v"(sv.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(v.y) : "v"(v.y), "v"(sv.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(v.z) : "v"(v.z), "v"(sv.z));
    asm("v_max_f32 %0, %1, %2" : "=v"(v.w) : "v"(v.w), "v"(sv.w));
    return v;
}
__device__ __forceinline__ u32x2 f32x4_to_bf16(f32x4 v) {         // round to nearest even (v_cvt_pk_bf16_f32)
    const bf16x4 b = __builtin_convertvector(v, bf16x4);
    return __builtin_bit_cast(u32x2, b);
}

#ifdef PFNL_BF16_TIMING   /* phase timeline of the kernel (tools/bf16_timing.py); not part of the product build */
__device__ long long cb_dbg[256 * 8 * 256];
#define CB_STAMP() do { if (lane == 0 && dbg_n < 256) cb_dbg[(blockIdx.x * 8 + wave) * 256 + dbg_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define CB_STAMP() do {} while (0)
#endif

// MODE 0: out = act(conv + bias).  MODE 1 (conv2_i per-frame half): out = act(conv + bias + addend[item / add_div]) + resid.
// MODE 2 (conv1_i + conv10_i): MODE 0, and per chain of add_div frames x_out = act(sum_t W10_t out_t + x_bias), the 1x1
// contraction running from the LDS scratch each finished tile passes through anyway (conv10_i costs no launch and
// no second read of conv1_i's output: -116 MB of 763 MB per block at 1080p).
// MODE 3 (convmerge1, reference model/pfnl.py:52, :73-74): out_f32[clip] = act(sum_t conv(in[clip*T + t]; W_t) + bias) - the
// accumulators run through the chain's add_div frames, the 72 KB weight pack in LDS is replaced between frames (requested
// into registers during a tile, written after its last MFMA), one fp32 epilogue per chain.
template <int MODE>
__global__ __launch_bounds__(CB_THREADS, 1) void conv3x3_bf16_kernel(ConvBf16Params p) {
    constexpr bool FUSE = MODE == 1;
    constexpr bool WITH10 = MODE == 2;
    constexpr bool ACCUM = MODE == 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char cb_smem[];
    unsigned char* const wl = cb_smem + 2 * CB_TILE_BYTES;
    float* const bl = reinterpret_cast<float*>(cb_smem + 2 * CB_TILE_BYTES + CB_W_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PFNL_BF16_TIMING
    int dbg_n = 0;
#endif
    const int rp = wave >> 1;                                       // rows 2rp, 2rp+1 of the tile
    const int mt = wave & 1;                                        // output channels 32mt .. 32mt+31
    const int H = p.H, W = p.W;
    const int tiles_x = (W + CB_TW - 1) / CB_TW, tiles_y = (H + CB_TH - 1) / CB_TH;
    const int per_item = tiles_x * tiles_y;
    const int item_bytes = H * W * 128;
    // Work order.  A unit = one spatial tile of one item; the fused launch (conv2_i's per-frame half) chains the gT
    // frames of a clip at the same spatial tile, so the shared-half addend pieces are fetched once per chain.  Chains
    // are dealt out XCD by XCD (blockIdx & 7 = XCD): the 32 workgroups of an XCD walk a contiguous run of spatial
    // tiles together, so the halo rows two tiles share come from that XCD's L2 instead of HBM a second time
    // (measured before: 1.31x the compulsory reads, and the addend 7x).
    const int gT = (FUSE || WITH10 || ACCUM) ? p.add_div : 1;
    const int nchains = per_item * (p.items / gT);
    const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3, cpx = gridDim.x >> 3;
    const int per_xcd = (nchains + 7) >> 3;
    const int cbeg = xcd * per_xcd;
    const int ccnt = min(per_xcd, nchains - cbeg);
    if (xj >= ccnt) return;
    const int nu = ((ccnt - xj + cpx - 1) / cpx) * gT;              // units of this workgroup
    // unit u -> (item, spatial tile)
#define CB_UNIT(u_, item_, sp_)                                                                  \
    do {                                                                                         \
        const int ci_ = (u_) / gT, f_ = (u_) - ci_ * gT;                                         \
        const int ch_ = cbeg + xj + ci_ * cpx;                                                   \
        const int cl_ = ch_ / per_item;                                                          \
        sp_ = ch_ - cl_ * per_item;                                                              \
        item_ = cl_ * gT + f_;                                                                   \
    } while (0)

    // weights + bias -> LDS (once per workgroup)
#pragma unroll
    for (int k = 0; k < CB_W_BYTES / 16 / CB_THREADS; ++k)
        reinterpret_cast<u32x4*>(wl)[k * CB_THREADS + tid] = reinterpret_cast<const u32x4*>(p.wpack)[k * CB_THREADS + tid];
    if (tid < 64) bl[tid] = p.bias[tid];

    // staging map: piece id = k*512 + tid -> halo pixel id >> 3, channel chunk id & 7, kept as one packed word per
    // piece (LDS and global offsets are recomputed per tile: ~60 VALU per thread against a tile's 2.3k MFMA cycles
    // per wave - the registers are worth more)
    int spk[CB_ITERS];                                              // py << 16 | px << 3 | chunk
    // Modes with registers to spare keep two more words per piece, constant for the life of the kernel (as conv_split16.hip):
    // `grel` = byte offset of the piece relative to the halo origin in HBM, `lad` = its LDS byte address inside a halo buffer -
    // the request is then one select + one add per piece and the commit one add, instead of the packed word's decode (a
    // 32-bit multiply among it) in every tile.  MODE 3 (252 registers) stays with the decode.
    constexpr bool FASTREQ = MODE != 3;
    [[maybe_unused]] int grel[CB_ITERS], lad[CB_ITERS];
    const int wbytes = W * 128;
#pragma unroll
    for (int k = 0; k < CB_ITERS; ++k) {
        const int id = min(k * CB_THREADS + tid, CB_CHUNKS - 1);    // surplus threads redo the last piece (same value)
        const int pix = id >> 3, c = id & 7;
        const int py = pix / CB_IW, px = pix - py * CB_IW;
        spk[k] = (py << 16) | (px << 3) | c;
        if constexpr (FASTREQ) {
            grel[k] = py * wbytes + px * 128 + c * 16;
            lad[k] = (py * CB_IW + px) * 128 + ((c ^ ((px >> 1) & 7)) << 4);
        }
    }
    u32x4 stg[CB_ITERS];
#define CB_REQUEST(u_)                                                                           \
    do {                                                                                         \
        int it_, sp_;                                                                            \
        CB_UNIT(u_, it_, sp_);                                                                   \
        const int ty_ = sp_ / tiles_x;                                                           \
        const int y0_ = ty_ * CB_TH, x0_ = (sp_ - ty_ * tiles_x) * CB_TW;                        \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                    \
            const_cast<uint16_t*>(p.in) + (size_t)it_ * H * W * 64, 0, item_bytes, 0x00020000);  \
        const int org_ = ((y0_ - 1) * W + x0_ - 1) * 128;       /* halo origin */                \
        if constexpr (FASTREQ) {                                /* one arm, branch-free */       \
            const bool interior_ = y0_ > 0 && y0_ + CB_IH - 1 <= H && x0_ > 0 && x0_ + CB_IW - 1 <= W; \
            _Pragma("unroll") for (int k_ = 0; k_ < CB_ITERS; ++k_) {                            \
                const int gy_ = y0_ + (spk[k_] >> 16) - 1, gx_ = x0_ + ((spk[k_] >> 3) & 0x1fff) - 1; \
                const bool in_ = interior_ || ((unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W); \
                stg[k_] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, in_ ? org_ + grel[k_] : 0x7fffffff, 0, 0)); \
            }                                                                                    \
        } else if (y0_ > 0 && y0_ + CB_IH - 1 <= H && x0_ > 0 && x0_ + CB_IW - 1 <= W) {         \
            _Pragma("unroll") for (int k_ = 0; k_ < CB_ITERS; ++k_)                              \
                stg[k_] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(       \
                    rs_, org_ + (spk[k_] >> 16) * wbytes + (spk[k_] & 0xffff) * 16, 0, 0));      \
        } else {                                                                                 \
            _Pragma("unroll") for (int k_ = 0; k_ < CB_ITERS; ++k_) {                            \
                const int gy_ = y0_ + (spk[k_] >> 16) - 1, gx_ = x0_ + ((spk[k_] >> 3) & 0x1fff) - 1; \
                const bool in_ = (unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W;     \
                stg[k_] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(       \
                    rs_, in_ ? org_ + (spk[k_] >> 16) * wbytes + (spk[k_] & 0xffff) * 16 : 0x7fffffff, 0, 0)); \
            }                                                                                    \
        }                                                                                        \
    } while (0)
#define CB_COMMIT(buf_)                                                                          \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < CB_ITERS; ++k_) {                                \
            if constexpr (FASTREQ) {                                                             \
                *reinterpret_cast<u32x4*>(cb_smem + (buf_) * CB_TILE_BYTES + lad[k_]) = stg[k_]; \
            } else {                                                                             \
                const int py_ = spk[k_] >> 16, px_ = (spk[k_] >> 3) & 0x1fff, c_ = spk[k_] & 7;  \
                *reinterpret_cast<u32x4*>(cb_smem + (buf_) * CB_TILE_BYTES + (py_ * CB_IW + px_) * 128 + ((c_ ^ ((px_ >> 1) & 7)) << 4)) = stg[k_]; \
            }                                                                                    \
        }                                                                                        \
    } while (0)

    // operand addresses: pixel operand of (column tap kx, k-step ks) = chunk 2*ks + (lane >> 5) of halo pixel
    // (row 2*rp + ..., column (lane & 31) + kx); weights: 16 bytes per lane
    int paddr[3][4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = (lane & 31) + kx;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            paddr[kx][ks] = ((2 * rp) * CB_IW + col) * 128 + (((2 * ks + (lane >> 5)) ^ ((col >> 1) & 7)) << 4);
    }
    const unsigned char* const wlane = wl + mt * 1024 + lane * 16;

    // Accumulator rows: the MFMA row index is free to name any output channel (it only fixes which weight row sits
    // in which lane), so row i of a channel tile is channel 16((i>>2)&1) + 4(i>>3) + (i&3): register r of a lane is
    // then channel 32mt + 16(lane>>5) + r - 16 consecutive channels, two 16-byte pieces per output row.
    const int ech = 32 * mt + 16 * (lane >> 5);
    f32x16 acc[2], accp[2];                                         // [output row]: the tile being computed / awaiting its epilogue
    u32x4 radd[2][2];                                               // FUSE: addend pieces (accumulator layout; fetched once per chain)
    // FUSE: the residual tile travels as WHOLE LINES - piece k of a thread = 16-byte chunk tid & 7 of pixel (row 2k + (tid >> 8),
    // column (tid >> 3) & 31), 8 lanes per 128-byte line - into the scratch slot its output will take, and each epilogue unit
    // reads its 16 bytes from there (a quarter of the cache-line requests of reading them in accumulator layout, a lane = a
    // pixel: 32 lines per instruction).  Measured: the launch takes the same 89 us either way, and 67 us with the residual
    // not read at all (-DCB_X_NOFUSELOAD) - it is the 116 MB, not the access pattern: the per-frame launch runs at the
    // ~4.2 TB/s this chip gives a kernel that reads and writes at once.
    u32x4 rq[4];
    int rbase = 0x7fffffff;                                         // byte offset of this thread's piece 0 in the tile's item (out of range: nothing)
    int eoff[2] = {0x7fffffff, 0x7fffffff};                         // this lane's pixel of output row n (bytes into the item): addend / residual pieces
    int eitem = 0, eitemp = 0;
    bool pending = false;

    // Stores.  A lane owns 32 contiguous bytes of its pixel (two 16-byte pieces), lane + 32 the next 32, the partner
    // wave (other channel tile) the other half line: stored as they sit, a wave instruction sends 64 separate 16-byte
    // writes to L2, and HBM sees half-written lines.  Measured on the 1080p launches (rocprofv3, t = a R + b W over
    // the conv1_i / conv2_i launches): reads ~10 TB/s, such writes 1.8 TB/s - against 4 TB/s for the fully coalesced
    // stores of the cast kernel.  So the finished tile (bias, addend, activation, residual applied, rounded to bf16)
    // goes through LDS: every wave drops its pieces into a pixel-major scratch (the OTHER halo buffer, free until the
    // next halo is committed; 16-byte pieces XOR-swizzled by (pixel >> 1) & 7 like the halo tiles, conflict-free for the
    // writes, the read-back and the 1x1 operand reads of MODE 2), and after a
    // barrier the workgroup stores it as whole 128-byte lines, 8 pixels per wave instruction.
    int ex0p = 0, ey0p = 0;                                         // origin of the tile awaiting its epilogue
    [[maybe_unused]] uint16_t* ptrOp = p.out;                       // its item in `out`
    [[maybe_unused]] const uint16_t* ptrR = p.resid;                // FUSE: the current tile's item in `resid`, its chain's in `addend`
    [[maybe_unused]] const uint16_t* ptrA = p.addend;
    const float eslope = p.act ? 0.2f : 1.0f;
    auto epilogue_unit = [&](unsigned char* scratch, int n, int h) __attribute__((always_inline)) {   // bias, addend, leaky_relu, residual, bf16
        if constexpr (ACCUM) {                                      // fp32 out, straight from the accumulators (one tile in T, 48 of 64 channels used)
            const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(p.out_f32 + (size_t)(eitemp / gT) * H * W * 64, 0, 2 * item_bytes, 0x00020000);
            const int sx = ex0p + (lane & 31), sy = ey0p + 2 * rp + n;
            const int off = (sx < W && sy < H) ? ((sy * W + sx) * 64 + ech + 8 * h) * 4 : 0x7fffffff;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r0 = 8 * h + 4 * q;
                f32x4 v = f32x4{accp[n][r0], accp[n][r0 + 1], accp[n][r0 + 2], accp[n][r0 + 3]} + *reinterpret_cast<const f32x4*>(bl + ech + r0);
                v = lrelu4(v, eslope);
                buffer_store_b128_guarded<CB_STORE_AUX>(__builtin_bit_cast(u32x4, v), rsF, off, 16 * q);
            }
            return;
        }
        f32x4 v[2];                                                 // of row n, channels ech + 8h .. + 7 of the tile awaiting its epilogue
        const int j = lane & 31;
        const int c = 4 * mt + 2 * (lane >> 5) + h;                 // piece of the pixel's line
        u32x4* const slot = reinterpret_cast<u32x4*>(scratch + ((2 * rp + n) * 32 + j) * 128 + ((c ^ ((j >> 1) & 7)) << 4));
        [[maybe_unused]] u32x4 rr = {0, 0, 0, 0};
        if constexpr (FUSE) rr = *slot;                             // the residual piece sits where the output piece goes
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r0 = 8 * h + 4 * q;
            v[q] = f32x4{accp[n][r0], accp[n][r0 + 1], accp[n][r0 + 2], accp[n][r0 + 3]};   // (the bias is in: initial C of the tile)
            if (FUSE) v[q] += bf16x4_to_f32(u32x2{radd[n][h][2 * q], radd[n][h][2 * q + 1]});
            v[q] = lrelu4(v[q], eslope);
            if (FUSE) v[q] += bf16x4_to_f32(u32x2{rr[2 * q], rr[2 * q + 1]});
        }
        const u32x2 lo = f32x4_to_bf16(v[0]), hi = f32x4_to_bf16(v[1]);
        *slot = u32x4{lo.x, lo.y, hi.x, hi.y};
    };
    auto store_piece = [&](const unsigned char* scratch, int k) __attribute__((always_inline)) {   // 2048 pieces, 4 per thread, whole lines per instruction
        if constexpr (ACCUM) return;                                // (stored by epilogue_unit)
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(ptrOp, 0, item_bytes, 0x00020000);
        const int id = k * CB_THREADS + tid;
        const int pp = id >> 3, c = id & 7;
        const int sx = ex0p + (pp & 31), sy = ey0p + (pp >> 5);
        const u32x4 o = *reinterpret_cast<const u32x4*>(scratch + pp * 128 + ((c ^ ((pp >> 1) & 7)) << 4));
#ifdef CB_X_NOSTORE   /* timing experiments only */
        if (o.x == 0x12345678u)
#endif
        buffer_store_b128_guarded<CB_STORE_AUX>(o, rsO, (sx < W && sy < H) ? (sy * W + sx) * 128 + c * 16 : 0x7fffffff, 0);
    };
    auto fuse_request = [&](bool with_addend, int n, int h) __attribute__((always_inline)) {   // addend / residual piece (n, h) of the tile described by eoff / eitem
        // (base pointers of the tile's items: computed once per tile, not per request - a 64-bit product and a division each)
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ptrA), 0, item_bytes, 0x00020000);
#ifdef CB_X_NOFUSELOAD   /* timing experiments only (wrong results on purpose) */
        if (eoff[n] == 0x12345)
#endif
        if (with_addend)                                            // (same pixels for every frame of the chain)
            radd[n][h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, eoff[n], (ech + 8 * h) * 2, 0));
    };
    auto resid_request = [&](int k) __attribute__((always_inline)) {   // rows 2k, 2k+1 of the current tile (rows past the image: past the resource)
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ptrR), 0, item_bytes, 0x00020000);
#ifdef CB_X_NOFUSELOAD
        if (rbase == 0x12345)
#endif
        rq[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, rbase, k * 2 * wbytes, 0));
    };
    auto resid_stage = [&](unsigned char* scratch) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pp = k * 64 + (tid >> 3), c = tid & 7;
            *reinterpret_cast<u32x4*>(scratch + pp * 128 + ((c ^ ((pp >> 1) & 7)) << 4)) = rq[k];
        }
    };

    // MODE 2: the 1x1 over the chain's frames.  Wave (rp, mt) accumulates rows 2rp, 2rp+1 x output channels 32mt..+31
    // of `base`; per finished frame tile: 4 k-steps x 2 rows = 8 MFMAs, B = the tile in the scratch, A = W10 of that
    // frame from L2 (16 bytes per lane and k-step, requested a few groups ahead).
    f32x16 bacc[2];
    u32x4 xw[4];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) bacc[n][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rsXW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x_w), 0, WITH10 ? gT * 8192 : 0, 0x00020000);
    auto x_request = [&](int ks) __attribute__((always_inline)) {  // W10[frame of the pending tile][k-step ks][channel tile mt]
        xw[ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsXW, lane * 16 + mt * 1024, ((eitemp % gT) * 4 + ks) * 2048, 0));
    };
    auto x_step = [&](const unsigned char* scratch, int ks) __attribute__((always_inline)) {
        const int j = lane & 31;
        const unsigned char* bp = scratch + ((2 * rp) * 32 + j) * 128 + (((2 * ks + (lane >> 5)) ^ ((j >> 1) & 7)) << 4);
#pragma unroll
        for (int n = 0; n < 2; ++n)
            bacc[n] = mfma_bf16(__builtin_bit_cast(bf16x8, xw[ks]), *reinterpret_cast<const bf16x8*>(bp + n * 32 * 128), bacc[n]);
    };
    auto x_epilogue = [&]() __attribute__((always_inline)) {        // chain complete: bias, leaky_relu, bf16, store; clear
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(p.x_out + (size_t)(eitemp / gT) * H * W * 64, 0, item_bytes, 0x00020000);
        const int sx = ex0p + (lane & 31);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int sy = ey0p + 2 * rp + n;
            const int off = (sx < W && sy < H) ? ((sy * W + sx) * 64 + ech) * 2 : 0x7fffffff;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 v[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r0 = 8 * h + 4 * q;
                    v[q] = f32x4{bacc[n][r0], bacc[n][r0 + 1], bacc[n][r0 + 2], bacc[n][r0 + 3]} +
                           *reinterpret_cast<const f32x4*>(p.x_bias + ech + r0);
                    v[q] = lrelu4(v[q], 0.2f);
                }
                const u32x2 lo = f32x4_to_bf16(v[0]), hi = f32x4_to_bf16(v[1]);
                buffer_store_b128_guarded<CB_STORE_AUX>(u32x4{lo.x, lo.y, hi.x, hi.y}, rsX, off, 16 * h);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) bacc[n][r] = 0.f;
        }
    };

    u32x4 wnx[CB_W_BYTES / 16 / CB_THREADS];                        // MODE 3: the next frame's weight pack on its way to LDS

    CB_REQUEST(0);
    CB_COMMIT(0);
    CB_REQUEST(min(1, nu - 1));                                     // past the end: harmless re-read
    __syncthreads();
    for (int u = 0; u < nu; ++u) {
        const int cb = u & 1;                                       // LDS buffer of this tile; the other one receives the next tile
        const unsigned char* const tile = cb_smem + cb * CB_TILE_BYTES;
        int item, sp;
        CB_UNIT(u, item, sp);
        const bool chain_head = !FUSE || item % gT == 0;            // first frame of a chain: fetch the addend pieces
        const int ty = sp / tiles_x;
        const int y0 = ty * CB_TH, x0 = (sp - ty * tiles_x) * CB_TW;
        const int ox = x0 + (lane & 31);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int oy = y0 + 2 * rp + n;
            eoff[n] = (ox < W && oy < H) ? (oy * W + ox) * 128 : 0x7fffffff;
        }
        eitem = item;
        if constexpr (FUSE) {
            const int rcol = x0 + ((tid >> 3) & 31), rrow = y0 + (tid >> 8);
            rbase = rcol < W ? (rrow * W + rcol) * 128 + (tid & 7) * 16 : 0x7fffffff;
            ptrR = p.resid + (size_t)item * H * W * 64;
            ptrA = p.addend + (size_t)(item / p.add_div) * H * W * 64;
        }
        // modes 0-2: the tile's first MFMAs take C = bias (register r of a lane = channel ech + r) - nothing to clear, no bias
        // reads (an exposed LDS round trip in each of the four epilogue groups) and no bias adds in the epilogue
        [[maybe_unused]] f32x16 bias16;
        if constexpr (!ACCUM) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + ech + 4 * q);
                bias16[4 * q] = b4.x;
                bias16[4 * q + 1] = b4.y;
                bias16[4 * q + 2] = b4.z;
                bias16[4 * q + 3] = b4.w;
            }
        } else if (item % gT == 0) {                                // (MODE 3: the accumulators run through the chain)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        }

        // Everything that is not an MFMA is spread over the 12 groups of the tile, a slice per group, so that it issues
        // in the shadow of the matrix pipe (both waves of a SIMD run the same phase at the same time - work done in a
        // block of its own is added to the MFMA time, not hidden by it: measured 9.2k cycles per tile with the MFMAs,
        // loads and stores compiled out).  Every wait is for a request issued most of a tile period earlier.
        //   groups 0-3:  epilogue unit (row, channel half) of the previous tile into the scratch, then that unit's
        //                addend / residual pieces of THIS tile are requested into the registers just consumed
        //   group 4:     barrier;  groups 4-7: a quarter of the previous tile's stores each, whole lines
        //   group 8:     barrier (scratch read);  the halo requested a tile ago -> the other LDS buffer (over the scratch)
        //   group 10:    the halo after next is requested
        unsigned char* const other = cb_smem + (cb ^ 1) * CB_TILE_BYTES;

        // 12 groups (column tap kx, k-step ks): the 4 halo rows 2rp..2rp+3 serve the 3 row taps of both output rows -
        // 4 pixel reads + 3 weight reads feed 6 MFMAs; the operands of group g+1 are requested before the MFMAs of g
        bf16x8 px[2][4], wv[2][3];
#define CB_PX(g_, r_) (*reinterpret_cast<const bf16x8*>(tile + paddr[(g_) >> 2][(g_) & 3] + (r_) * (CB_IW * 128)))
#define CB_WT(g_, ky_) (*reinterpret_cast<const bf16x8*>(wlane + ((((ky_) * 3 + ((g_) >> 2)) * 4 + ((g_) & 3)) << 11)))
#pragma unroll
        for (int r = 0; r < 4; ++r) px[0][r] = CB_PX(0, r);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) wv[0][ky] = CB_WT(0, ky);
        // (written out through a generic lambda over integral constants: a `#pragma unroll` loop of this size can be
        // left rolled by the optimizer, which silently turns the operand arrays into scratch memory - 60x slower)
        auto group = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int cur = g & 1;
#ifdef PFNL_BF16_TIMING
            CB_STAMP();                                             // group start (16 stamps per tile: 12 groups + commit pair + barrier pair)
#endif
            // FUSE: everything one group later - group 0 puts the residual lines of the pending tile into the scratch, group 1
            // starts with the barrier that makes them visible
            constexpr int SH = FUSE ? 1 : 0;
            if constexpr (FUSE && g == 0) {
                if (pending) resid_stage(other);
            }
            if constexpr (FUSE && g == 1) __syncthreads();
            if constexpr (g >= SH && g < SH + 4) {
                constexpr int e = g - SH;
                if (pending) epilogue_unit(other, e >> 1, e & 1);
                if constexpr (FUSE) {
                    fuse_request(chain_head, e >> 1, e & 1);
                    resid_request(e);                               // (this tile's lines; staged in group 0 of the next tile)
                }
                if (WITH10 && pending) x_request(e);
            }
            if constexpr (g == SH + 4) __syncthreads();
            if constexpr (g >= SH + 4 && g < SH + 8) {
                if (pending) store_piece(other, g - SH - 4);
                if (WITH10 && pending) x_step(other, g - SH - 4);
            }
            if constexpr (g == SH + 8) {
                __syncthreads();                                       // the scratch has been read
                if (WITH10 && pending && eitemp % gT == gT - 1) x_epilogue();   // the pending tile closed its chain
                CB_STAMP();
#pragma unroll
                for (int k = 0; k < CB_ITERS; ++k) asm volatile("" : "+v"(spk[k]));   // opaque: nothing derived from it is hoisted into registers
                CB_COMMIT(cb ^ 1);
                CB_STAMP();
            }
            if constexpr (g == 10) {
#ifndef CB_X_NOLOAD   /* timing experiments only */
                CB_REQUEST(min(u + 2, nu - 1));
#endif
                if constexpr (ACCUM) {                              // weights of the next unit's frame (units are whole chains: frame + 1 mod T)
                    const size_t wo = (size_t)((item % gT + 1) % gT) * (CB_W_BYTES / 16);
#pragma unroll
                    for (int k = 0; k < CB_W_BYTES / 16 / CB_THREADS; ++k) wnx[k] = reinterpret_cast<const u32x4*>(p.wpack)[wo + k * CB_THREADS + tid];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g < 11) {
#pragma unroll
                for (int r = 0; r < 4; ++r) px[cur ^ 1][r] = CB_PX(g + 1, r);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) wv[cur ^ 1][ky] = CB_WT(g + 1, ky);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if constexpr (g == 0 && !ACCUM) {
                        if (ky == 0) {
                            acc[n] = mfma_bf16(wv[cur][ky], px[cur][n + ky], bias16);
                            continue;
                        }
                    }
                    acc[n] = mfma_bf16(wv[cur][ky], px[cur][n + ky], acc[n]);
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
#undef CB_PX
#undef CB_WT
        if (!ACCUM || item % gT == gT - 1) {                        // (MODE 3: only the chain's last frame leaves the accumulators)
            accp[0] = acc[0];
            accp[1] = acc[1];
            ex0p = x0;
            ey0p = y0;
            eitemp = eitem;
            if constexpr (!ACCUM) ptrOp = p.out + (size_t)eitem * H * W * 64;
            pending = true;
        } else {
            pending = false;
        }
        CB_STAMP();
        __syncthreads();                                               // this tile's buffer is free, the next tile's is complete
        CB_STAMP();
        if constexpr (ACCUM) {                                      // every wave is past its last MFMA with the old weights
            if (gT > 1) {
#pragma unroll
                for (int k = 0; k < CB_W_BYTES / 16 / CB_THREADS; ++k) reinterpret_cast<u32x4*>(wl)[k * CB_THREADS + tid] = wnx[k];
                __syncthreads();
            }
        }
    }
    // the last tile (its addend / residual pieces were requested in its own iteration); any buffer is free now
    if constexpr (FUSE) {
        resid_stage(cb_smem);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!ACCUM || pending) epilogue_unit(cb_smem, k >> 1, k & 1);
        if (WITH10) x_request(k);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        store_piece(cb_smem, k);
        if (WITH10) x_step(cb_smem, k);
    }
    if (WITH10) x_epilogue();                                       // (a workgroup's units are whole chains)
#undef CB_COMMIT
#undef CB_REQUEST
#undef CB_UNIT
}

hipError_t launch_conv3x3_bf16(const ConvBf16Params& p, hipStream_t s) {
    if (!p.in || !p.wpack || !p.bias || (!p.out && !p.out_f32) || p.items < 1 || p.H < 1 || p.W < 1) return hipErrorInvalidValue;
    if ((p.addend == nullptr) != (p.resid == nullptr) || (p.addend && (p.add_div < 1 || p.items % p.add_div))) return hipErrorInvalidValue;
    if ((long long)p.H * p.W * 128 >= 0x7fffffffLL) return hipErrorInvalidValue;
    const int ncu = device_cu_count();
    if (!ncu) return hipErrorUnknown;
    const int grid = ncu >= 8 ? ncu / 8 * 8 : 8;                    // whole XCDs; surplus workgroups exit at once
    const bool accum = p.out_f32 != nullptr;
    if (accum && (p.addend || p.x_out || p.add_div < 1 || p.items % p.add_div)) return hipErrorInvalidValue;
    const bool with10 = p.x_out != nullptr;
    if (with10 && (p.addend || !p.x_w || !p.x_bias || p.add_div < 1 || p.add_div > 7 || p.items % p.add_div)) return hipErrorInvalidValue;
    const int mode = accum ? 3 : (p.addend ? 1 : (with10 ? 2 : 0));
    if (mode != 3) {
        // Modes 0 - 2 run on the later generations (round 5: this file's kernel keeps the accumulating mode 3 = convmerge1 only; its modes 0 - 2
        // were reachable through PFNL_BF16_V2=0 alone and left the library).  The third-generation kernel (conv_bf16_v3.hip: the two halves
        // of the workgroup half a tile period apart) takes conv1_i + conv10_i and the per-frame half of conv2_i (1080p, same box, rocprofv3:
        // 76.3 -> 73.2 and 82.5 -> 79.0 us per launch, 4.24 -> 4.09 ms per forward); the plain mode - the shared half of conv2_i, two tiles per
        // workgroup at 1080p - stays on the second generation (conv_bf16_v2.hip), whose prologue is shorter (15.8 against 16.1 us).
        // PFNL_BF16_V3 = 0: the second generation for all three modes, = 1: the third for all three, = 2: the third for conv1_i + conv10_i only -
        // same arithmetic, other schedule: kept as the A/B switch and as the bit-equality reference of tools/soak_r04.py.
        static const int v3 = [] {
            const char* e = std::getenv("PFNL_BF16_V3");
            return e ? std::atoi(e) : 12;
        }();
        if (p.split_s && mode == 0) return hipErrorInvalidValue;         // (split chains exist in the third generation's chained modes only)
        if (p.split_s || v3 == 1 || (v3 == 12 && mode != 0) || (v3 == 2 && mode == 2)) return launch_conv3x3_bf16_v3(p, mode, s);
        return launch_conv3x3_bf16_v2(p, mode, s);
    }
    static std::atomic<int> attr_dev[64];                                  // the attribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_dev[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, CB_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_dev[dev] = 1;
    }
    hipLaunchKernelGGL(conv3x3_bf16_kernel<3>, dim3(grid), dim3(CB_THREADS), CB_LDS_BYTES, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// conv10_i in bf16: T*64 -> 64 over the concat of the T frames of a clip.  HBM-bound by a wide margin (896 B read
// per pixel for 57 kFLOP): the packed weights (T x 8 KB) sit in LDS, a wave takes 32 consecutive pixels, requests
// all T*4 16-byte pieces of its lanes' pixels at once and runs the 8T MFMAs as they arrive.
template <int T>
__global__ __launch_bounds__(256, 2) void conv1x1_bf16_kernel(const uint16_t* __restrict__ in, const uint16_t* __restrict__ wpack,
                                                              const float* __restrict__ bias, uint16_t* __restrict__ out,
                                                              int HW, int items, int act) {
    __shared__ __attribute__((aligned(16))) unsigned char wl[T * 8192 + 256];
    float* const bl = reinterpret_cast<float*>(wl + T * 8192);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < T * 512; k += 256) reinterpret_cast<u32x4*>(wl)[k] = reinterpret_cast<const u32x4*>(wpack)[k];
    if (tid < 64) bl[tid] = bias[tid];
    __syncthreads();
    const int gpi = (HW + 31) >> 5;
    const int ngroups = gpi * items;
    const int frame_bytes = HW * 128;
    const unsigned char* const wlane = wl + lane * 16;
    const int ech = 16 * (lane >> 5);
    for (int g = blockIdx.x * 4 + wave; g < ngroups; g += gridDim.x * 4) {
        const int item = g / gpi;
        const int p0 = (g - item * gpi) * 32;
        const int px = p0 + (lane & 31);
        const int voff = px < HW ? px * 128 + (lane >> 5) * 16 : 0x7fffffff;
        const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(in) + (size_t)item * T * HW * 64, 0, T * frame_bytes, 0x00020000);
        bf16x8 b[T][4];
#pragma unroll
        for (int f = 0; f < T; ++f)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                b[f][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsI, voff, f * frame_bytes + ks * 32, 0));
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[0][r] = 0.f;
            acc[1][r] = 0.f;
        }
#pragma unroll
        for (int f = 0; f < T; ++f) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m] = mfma_bf16(*reinterpret_cast<const bf16x8*>(wlane + (((f * 4 + ks) * 2 + m) << 10)), b[f][ks], acc[m]);
            __builtin_amdgcn_sched_barrier(0);                      // keep the weight reads next to their MFMAs (registers)
        }
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)item * HW * 64, 0, frame_bytes, 0x00020000);
        const int ooff = px < HW ? px * 128 + ech * 2 : 0x7fffffff;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) {                           // register r = channel 32m + 16(lane>>5) + r
                f32x4 v[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r0 = 8 * h + 4 * q;
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + 32 * m + ech + r0);
                    v[q] = f32x4{acc[m][r0], acc[m][r0 + 1], acc[m][r0 + 2], acc[m][r0 + 3]} + b4;
                    if (act) {
                        v[q].x = lrelu(v[q].x);
                        v[q].y = lrelu(v[q].y);
                        v[q].z = lrelu(v[q].z);
                        v[q].w = lrelu(v[q].w);
                    }
                }
                const u32x2 lo = f32x4_to_bf16(v[0]), hi = f32x4_to_bf16(v[1]);
                buffer_store_b128_guarded<CB_STORE_AUX>(u32x4{lo.x, lo.y, hi.x, hi.y}, rsO, ooff, (32 * m + 8 * h) * 2);
            }
    }
}

hipError_t launch_conv1x1_bf16(const uint16_t* in, const uint16_t* wpack, const float* bias, uint16_t* out, int items, int T,
                               int HW, int act, hipStream_t s) {
    if (!in || !wpack || !bias || !out || items < 1 || HW < 1) return hipErrorInvalidValue;
    if ((long long)T * HW * 128 >= 0x7fffffffLL) return hipErrorInvalidValue;
    const int ngroups = ((HW + 31) / 32) * items;
    const int grid = (ngroups + 3) / 4 < 512 ? (ngroups + 3) / 4 : 512;
    switch (T) {
        case 3: hipLaunchKernelGGL(conv1x1_bf16_kernel<3>, dim3(grid), dim3(256), 0, s, in, wpack, bias, out, HW, items, act); break;
        case 5: hipLaunchKernelGGL(conv1x1_bf16_kernel<5>, dim3(grid), dim3(256), 0, s, in, wpack, bias, out, HW, items, act); break;
        case 7: hipLaunchKernelGGL(conv1x1_bf16_kernel<7>, dim3(grid), dim3(256), 0, s, in, wpack, bias, out, HW, items, act); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// element casts between the bf16 trunk and the fp32 kernels either side of it
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const u32x4* __restrict__ in, f32x4* __restrict__ out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const u32x4 v = in[i];
        out[2 * i] = bf16x4_to_f32(u32x2{v.x, v.y});
        out[2 * i + 1] = bf16x4_to_f32(u32x2{v.z, v.w});
    }
}
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const f32x4* __restrict__ in, u32x4* __restrict__ out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const u32x2 a = f32x4_to_bf16(in[2 * i]), b = f32x4_to_bf16(in[2 * i + 1]);
        out[i] = u32x4{a.x, a.y, b.x, b.y};
    }
}

hipError_t launch_cast_bf16_f32(const uint16_t* in, float* out, size_t n, hipStream_t s) {
    if (n % 8) return hipErrorInvalidValue;
    const size_t n8 = n / 8;
    const int grid = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid ? grid : 1), dim3(256), 0, s, reinterpret_cast<const u32x4*>(in),
                       reinterpret_cast<f32x4*>(out), n8);
    return hipGetLastError();
}
hipError_t launch_cast_f32_bf16(const float* in, uint16_t* out, size_t n, hipStream_t s) {
    if (n % 8) return hipErrorInvalidValue;
    const size_t n8 = n / 8;
    const int grid = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid ? grid : 1), dim3(256), 0, s, reinterpret_cast<const f32x4*>(in),
                       reinterpret_cast<u32x4*>(out), n8);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// host-side packing (round to nearest even, like the device conversion)
uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// MFMA row i of a channel tile -> channel within the tile (see the kernel: 16 consecutive channels per lane)
static int bf16_row_channel(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }

size_t conv3x3_bf16_pack_halfs() { return CB_W_BYTES / 2; }

// HWIO [3,3,cin_total,64] rows [cin_begin, cin_begin+64) -> [tap][ks][m][lane][e]:
// W[ky][kx][cin_begin + 16 ks + 8 (lane>>5) + e][32 m + bf16_row_channel(lane&31)]
void conv3x3_bf16_pack_weights(const float* hwio, int cin_total, int cin_begin, uint16_t* dst, int cout) {
    for (int tap = 0; tap < 9; ++tap)
        for (int ks = 0; ks < 4; ++ks)
            for (int m = 0; m < 2; ++m)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int ci = cin_begin + 16 * ks + 8 * (lane >> 5) + e;
                        const int co = 32 * m + bf16_row_channel(lane & 31);
                        dst[((((size_t)tap * 4 + ks) * 2 + m) * 64 + lane) * 8 + e] =
                            co < cout ? bf16_rne(hwio[((size_t)tap * cin_total + ci) * cout + co]) : (uint16_t)0;
                    }
}

// The same kernel for the M16 form of conv3x3_bf16_v3_kernel (v_mfma_f32_16x16x32_bf16, A = weights): [ky][kx][kk][output tile c4][lane][e] =
// W[ky][kx][cin_begin + 32 kk + 8 (lane >> 4) + e][32 (c4 >> 1) + 8 (m >> 2) + 4 (c4 & 1) + (m & 3)], m = lane & 15: row m of output tile c4 is
// permuted so that lane group g = l >> 4 of the RESULT (rows 4 g + r of both tiles of a channel half) holds the 8 consecutive channels 8 g .. 8 g + 7
void conv3x3_bf16_pack_weights16(const float* hwio, int cin_total, int cin_begin, uint16_t* dst) {
    for (int tap = 0; tap < 9; ++tap)
        for (int kk = 0; kk < 2; ++kk)
            for (int c4 = 0; c4 < 4; ++c4)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int m = lane & 15;
                        const int ci = cin_begin + 32 * kk + 8 * (lane >> 4) + e;
                        const int co = 32 * (c4 >> 1) + 8 * (m >> 2) + 4 * (c4 & 1) + (m & 3);
                        dst[((((size_t)tap * 2 + kk) * 4 + c4) * 64 + lane) * 8 + e] = bf16_rne(hwio[((size_t)tap * cin_total + ci) * 64 + co]);
                    }
}

size_t conv1x1_bf16_pack_halfs(int T) { return (size_t)T * 4096; }

// HWIO [1,1,T*64,64] -> [f][ks][m][lane][e]: W[64 f + 16 ks + 8 (lane>>5) + e][32 m + bf16_row_channel(lane&31)]
void conv1x1_bf16_pack_weights(const float* hwio, int T, uint16_t* dst) {
    for (int f = 0; f < T; ++f)
        for (int ks = 0; ks < 4; ++ks)
            for (int m = 0; m < 2; ++m)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 64 * f + 16 * ks + 8 * (lane >> 5) + e;
                        const int co = 32 * m + bf16_row_channel(lane & 31);
                        dst[((((size_t)f * 4 + ks) * 2 + m) * 64 + lane) * 8 + e] = bf16_rne(hwio[(size_t)ci * 64 + co]);
                    }
}

}  // namespace pfnl

#ifdef PFNL_BF16_TIMING
extern "C" int pfnl_debug_read_bf16_stamps(long long* host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pfnl::cb_dbg), n * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

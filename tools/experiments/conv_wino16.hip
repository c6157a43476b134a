// Fused Winograd F(2x2,3x3) convolution 64 -> 64, f32 MFMA - "one wave owns everything" variant.
// EXPERIMENTAL (option conv3x3 = "winograd16"): parity-tested, but measured 2.2x SLOWER than conv_wino.hip
// (15.3 vs 6.9 ms per step): with one wave per SIMD the in-order vmcnt queue couples the L2-latency U
// loads to the HBM-latency halo loads issued just before them and no second wave covers the stall.  Kept as
// the starting point for a producer/consumer wave split.
//
// Same contract and maths as conv_wino.hip.  The measured limit of that kernel is structural: a wave
// owns 4 of the 16 Winograd positions, i.e. only 128 MFMAs per tile against fixed prologue / four
// chunk barriers / cross-wave LDS exchange / epilogue costs.  Here ONE wave owns all
// 16 positions of a (32 Winograd tiles x 32 couts) unit:
//   * 16 accumulators = 256 AGPRs of gfx950's unified 512-register file (1 wave per SIMD);
//   * no cross-wave output transform, no workgroup barrier at all: workgroup = 1 wave, the raw halo
//     tile lives in wave-private LDS (2 x 15.4 KB), LDS ordering inside a wave is program order;
//   * every wave is PERSISTENT over a static list of units and runs ONE software-pipelined stream of
//     (unit, chunk, K-step): while K-step s issues its 16 MFMAs (1024 matrix-pipe cycles) the wave
//     also issues the 16 LDS reads + 32 VALU of the input transform of K-step s+1, the U loads of
//     K-step s+2, and the global->register->LDS staging of the halo chunk two chunks ahead;
//   * per K-step the instruction stream is fixed by four inline-asm blocks of 4 MFMAs (hipcc clumps
//     MFMAs and strands the VALU work behind them otherwise); the first block also issues the LDS
//     reads of the next step, the transform VALU sits between the blocks.
// Output transform A^T M A happens in registers; stores are 128-byte channel-contiguous segments.
#include <cstdint>

#include "common.h"

namespace pfnl {

constexpr int W16_TX = 16, W16_TY = 2;
constexpr int W16_IW = 2 * W16_TX + 2;            // 34
constexpr int W16_IH = 2 * W16_TY + 2;            // 6
constexpr int W16_HALF = 20;
constexpr int W16_RP = 2 * W16_HALF;              // 40
constexpr int W16_PS = W16_IH * W16_RP + 1;       // 241
constexpr int W16_CK = 16;
constexpr int W16_NCHUNK = 64 / W16_CK;
constexpr int W16_BUF = W16_CK * W16_PS;          // 3856 floats
constexpr int W16_TBL = 2 * W16_BUF;              // per-lane staging descriptors live after the two raw buffers
constexpr int W16_ITERS_C = (W16_IH * W16_IW * (W16_CK / 4) + 63) / 64;
constexpr size_t W16_LDS_BYTES = size_t(2 * W16_BUF + 2 * W16_ITERS_C * 64) * sizeof(float);   // 37 504 B per wave
constexpr int W16_ITEMS = W16_IH * W16_IW * (W16_CK / 4);                // 816 float4 pieces per chunk
constexpr int W16_ITERS = (W16_ITEMS + 63) / 64;                         // 13 per lane
constexpr int W16_UDEPTH = 2;                     // K-steps of U in flight (16 floats per lane each)
constexpr int W16_KSTEPS = W16_NCHUNK * (W16_CK / 2);                    // 32 per unit
constexpr int w16_pack_floats_c = 16 * 64 * 64;

// block 0 of a K-step: 4 MFMAs of Winograd row 0 + the 16 LDS reads of the NEXT K-step's 4x4 patch.
// d[a*4+b] = patch row a, column b; column b lives at (b&1)*HALF + (b>>1) -> byte offsets 0, 80, 4, 84
// inside row a at 160*a (W16_RP = 40 floats, W16_HALF = 20 floats).  Rows are read in the order
// 2,0,1,3 and the block returns once 12 reads have landed: rows 2,0,1 are needed first (T0..T2).
static_assert(W16_RP * 4 == 160 && W16_HALF * 4 == 80, "literal offsets in w16_block0");
template <int OFF>
__device__ __forceinline__ void w16_block0(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3, float v0, float v1,
                                           float v2, float v3, const f32x4 u, unsigned pa, float (&d)[16]) {
    asm volatile(
        "s_nop 3\n\t"     // VALU-written V -> MFMA SrcA hazard: hipcc pads nothing inside asm
        "v_mfma_f32_32x32x2_f32 %[a0], %[v0], %[ux], %[a0]\n\t"
        "ds_read_b32 %[d8], %[pa] offset:%c[o]+320\n\t"
        "ds_read_b32 %[d9], %[pa] offset:%c[o]+400\n\t"
        "ds_read_b32 %[d10], %[pa] offset:%c[o]+324\n\t"
        "ds_read_b32 %[d11], %[pa] offset:%c[o]+404\n\t"
        "v_mfma_f32_32x32x2_f32 %[a1], %[v1], %[uy], %[a1]\n\t"
        "ds_read_b32 %[d0], %[pa] offset:%c[o]+0\n\t"
        "ds_read_b32 %[d1], %[pa] offset:%c[o]+80\n\t"
        "ds_read_b32 %[d2], %[pa] offset:%c[o]+4\n\t"
        "ds_read_b32 %[d3], %[pa] offset:%c[o]+84\n\t"
        "v_mfma_f32_32x32x2_f32 %[a2], %[v2], %[uz], %[a2]\n\t"
        "ds_read_b32 %[d4], %[pa] offset:%c[o]+160\n\t"
        "ds_read_b32 %[d5], %[pa] offset:%c[o]+240\n\t"
        "ds_read_b32 %[d6], %[pa] offset:%c[o]+164\n\t"
        "ds_read_b32 %[d7], %[pa] offset:%c[o]+244\n\t"
        "v_mfma_f32_32x32x2_f32 %[a3], %[v3], %[uw], %[a3]\n\t"
        "ds_read_b32 %[d12], %[pa] offset:%c[o]+480\n\t"
        "ds_read_b32 %[d13], %[pa] offset:%c[o]+560\n\t"
        "ds_read_b32 %[d14], %[pa] offset:%c[o]+484\n\t"
        "ds_read_b32 %[d15], %[pa] offset:%c[o]+564\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        : [a0] "+a"(a0), [a1] "+a"(a1), [a2] "+a"(a2), [a3] "+a"(a3), [d0] "=&v"(d[0]), [d1] "=&v"(d[1]),
          [d2] "=&v"(d[2]), [d3] "=&v"(d[3]), [d4] "=&v"(d[4]), [d5] "=&v"(d[5]), [d6] "=&v"(d[6]), [d7] "=&v"(d[7]),
          [d8] "=&v"(d[8]), [d9] "=&v"(d[9]), [d10] "=&v"(d[10]), [d11] "=&v"(d[11]), [d12] "=&v"(d[12]),
          [d13] "=&v"(d[13]), [d14] "=&v"(d[14]), [d15] "=&v"(d[15])
        : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [ux] "v"(u.x), [uy] "v"(u.y), [uz] "v"(u.z),
          [uw] "v"(u.w), [pa] "v"(pa), [o] "i"(OFF)
        : "memory");
}

// blocks 1..3 of a K-step: 4 MFMAs of one Winograd row; WAIT = lgkmcnt to drain before returning (-1: none)
template <int WAIT>
__device__ __forceinline__ void w16_block(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3, float v0, float v1,
                                          float v2, float v3, const f32x4 u) {
    if (WAIT >= 0) {
        asm volatile(
            "s_nop 3\n\t"
            "v_mfma_f32_32x32x2_f32 %[a0], %[v0], %[ux], %[a0]\n\t"
            "v_mfma_f32_32x32x2_f32 %[a1], %[v1], %[uy], %[a1]\n\t"
            "v_mfma_f32_32x32x2_f32 %[a2], %[v2], %[uz], %[a2]\n\t"
            "v_mfma_f32_32x32x2_f32 %[a3], %[v3], %[uw], %[a3]\n\t"
            "s_waitcnt lgkmcnt(%c[w])\n\t"
            : [a0] "+a"(a0), [a1] "+a"(a1), [a2] "+a"(a2), [a3] "+a"(a3)
            : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [ux] "v"(u.x), [uy] "v"(u.y), [uz] "v"(u.z),
              [uw] "v"(u.w), [w] "i"(WAIT < 0 ? 0 : WAIT)
            : "memory");
    } else {
        asm volatile(
            "s_nop 3\n\t"
            "v_mfma_f32_32x32x2_f32 %[a0], %[v0], %[ux], %[a0]\n\t"
            "v_mfma_f32_32x32x2_f32 %[a1], %[v1], %[uy], %[a1]\n\t"
            "v_mfma_f32_32x32x2_f32 %[a2], %[v2], %[uz], %[a2]\n\t"
            "v_mfma_f32_32x32x2_f32 %[a3], %[v3], %[uw], %[a3]\n\t"
            : [a0] "+a"(a0), [a1] "+a"(a1), [a2] "+a"(a2), [a3] "+a"(a3)
            : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [ux] "v"(u.x), [uy] "v"(u.y), [uz] "v"(u.z),
              [uw] "v"(u.w)
            : "memory");
    }
}

struct W16Unit {      // one (spatial tile, N-tile) unit
    int item, x0, y0, g;
};

template <bool FUSE>
__global__ __launch_bounds__(64, 1) void conv_wino16_kernel(WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int tx = lane & 15;
    const int ty = (lane >> 4) & 1;
    const int kh = lane >> 5;
    const int xl = lane & 31;
    const int H = p.H, W = p.W;

    // ---- unit list of this wave: XCD-contiguous regions (block b runs on XCD b & 7), the two N-tiles of
    // a spatial tile adjacent; static round-robin inside the region --------------------------------------
    const int tiles_x = (W + 2 * W16_TX - 1) / (2 * W16_TX);
    const int tiles_y = (H + 2 * W16_TY - 1) / (2 * W16_TY);
    const int per_item = tiles_x * tiles_y;
    const int nunits = 2 * per_item * p.items;
    int u, u_end, u_stride;
    if ((gridDim.x & 7) == 0) {
        const int rs = ((nunits + 15) >> 4) << 1;                    // even number of units per XCD region
        u = (blockIdx.x & 7) * rs + (int)(blockIdx.x >> 3);
        u_end = min(nunits, ((int)(blockIdx.x & 7) + 1) * rs);
        u_stride = gridDim.x >> 3;
    } else {
        u = blockIdx.x;
        u_end = nunits;
        u_stride = gridDim.x;
    }
    if (u >= u_end) return;
    const int n_my = (u_end - u + u_stride - 1) / u_stride;         // units this wave will process

#define W16_DECODE(unit_, d_)                                                                    \
    do {                                                                                         \
        const int tile_ = (unit_) >> 1;                                                          \
        (d_).g = (unit_) & 1;                                                                    \
        (d_).item = tile_ / per_item;                                                            \
        const int rem_ = tile_ - (d_).item * per_item;                                           \
        const int by_ = rem_ / tiles_x;                                                          \
        (d_).y0 = by_ * (2 * W16_TY);                                                            \
        (d_).x0 = (rem_ - by_ * tiles_x) * (2 * W16_TX);                                         \
    } while (0)

    // ---- staging: chunk c of the wave's stream = (unit c >> 2, channel chunk c & 3) ---------------------
    // Piece i of a lane: it = lane + 64 i -> halo pixel (py, px), channel quad c4.  Its descriptors are
    // computed ONCE and parked in a wave-private LDS table (w0 = global offset relative to the tile
    // origin, w1 = LDS slot | py << 16 | px << 20 | valid << 28): kept in registers they cost ~40 VGPRs
    // for the whole kernel (hipcc hoists them out of the unit loop) and push the wave into scratch.
    int* tbl = reinterpret_cast<int*>(smem + W16_TBL);
#pragma unroll
    for (int i = 0; i < W16_ITERS; ++i) {
        const int it = lane + i * 64;
        const int pix = it >> 2, c4 = it & 3;
        const int py = pix / W16_IW, px = pix - py * W16_IW;
        const int ok = it < W16_ITEMS ? 1 : 0;
        tbl[(2 * i) * 64 + lane] = (py * W + px) * 64 + c4 * 4;
        tbl[(2 * i + 1) * 64 + lane] =
            ((c4 * 4) * W16_PS + py * W16_RP + (px & 1) * W16_HALF + (px >> 1)) | (py << 16) | (px << 20) | (ok << 28);
    }
    f32x4 stg[W16_ITERS];
    unsigned stg_mask = 0;                  // bit i: piece i of the staged chunk is inside the image
#define W16_LOAD(cidx_)                                                                          \
    do {                                                                                         \
        const int un_ = u + ((cidx_) >> 2) * u_stride;                                           \
        W16Unit du_;                                                                             \
        W16_DECODE(un_ < u_end ? un_ : u, du_);                                                  \
        const float* fin_ = p.in + (size_t)du_.item * H * W * 64 + ((cidx_) & 3) * W16_CK;      \
        const int org_ = ((du_.y0 - 1) * W + (du_.x0 - 1)) * 64;   /* tile origin (may be < 0) */  \
        stg_mask = 0;                                                                            \
        _Pragma("unroll") for (int i = 0; i < W16_ITERS; ++i) {                                  \
            const int w0_ = tbl[(2 * i) * 64 + lane], w1_ = tbl[(2 * i + 1) * 64 + lane];        \
            const int gy_ = du_.y0 - 1 + ((w1_ >> 16) & 15), gx_ = du_.x0 - 1 + ((w1_ >> 20) & 63); \
            const bool ok_ = (w1_ >> 28) && gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;          \
            stg[i] = *reinterpret_cast<const f32x4*>(fin_ + (ok_ ? org_ + w0_ : 0));             \
            stg_mask |= ok_ ? (1u << i) : 0u;                                                    \
        }                                                                                        \
    } while (0)
#define W16_STORE(buf_)                                                                          \
    do {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < W16_ITERS; ++i) {                                  \
            const int w1_ = tbl[(2 * i + 1) * 64 + lane];                                        \
            if (w1_ >> 28) {                                                                     \
                const f32x4 v_ = ((stg_mask >> i) & 1u) ? stg[i] : f32x4{0.f, 0.f, 0.f, 0.f};    \
                float* d_ = (buf_) + (w1_ & 0xFFFF);                                             \
                d_[0] = v_.x;                                                                    \
                d_[W16_PS] = v_.y;                                                               \
                d_[2 * W16_PS] = v_.z;                                                           \
                d_[3 * W16_PS] = v_.w;                                                           \
            }                                                                                    \
        }                                                                                        \
    } while (0)

    // ---- U: [chunk][kk][g][lane][xi*4+nu], 4 x 16-byte buffer loads per K-step -------------------------
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.upack), 0, (int)(w16_pack_floats_c * sizeof(float)), 0x00020000);
    f32x4 uring[W16_UDEPTH][4];
    const int u_voff = lane * 64;          // the only per-lane part of a U address; step and N-tile are scalar
#define W16_ULOAD(slot_, s_, g_)                                                                 \
    do {                                                                                         \
        const int so_ = __builtin_amdgcn_readfirstlane(((s_) * 2 + (g_)) * 4096);                \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                            \
            uring[slot_][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, u_voff, so_ + e * 16, 0)); \
    } while (0)

    // ---- per-lane LDS read base (channel half, tile row, tile column) -----------------------------------
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned pa = lds0 + (unsigned)(kh * W16_PS + (2 * ty) * W16_RP + tx) * 4u;

    // input transform of a 4x4 patch d[a*4+b]:  T = Bt d  (rows), V = T B (columns)
#define W16_TROW(d_, t_)                                                                         \
    do {                                                                                         \
        _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                          \
            (t_)[0 * 4 + b] = (d_)[0 * 4 + b] - (d_)[2 * 4 + b];                                 \
            (t_)[1 * 4 + b] = (d_)[1 * 4 + b] + (d_)[2 * 4 + b];                                 \
            (t_)[2 * 4 + b] = (d_)[2 * 4 + b] - (d_)[1 * 4 + b];                                 \
            (t_)[3 * 4 + b] = (d_)[1 * 4 + b] - (d_)[3 * 4 + b];                                 \
        }                                                                                        \
    } while (0)
#define W16_VROW(t_, v_, x_)                                                                     \
    do {                                                                                         \
        (v_)[(x_) * 4 + 0] = (t_)[(x_) * 4 + 0] - (t_)[(x_) * 4 + 2];                            \
        (v_)[(x_) * 4 + 1] = (t_)[(x_) * 4 + 1] + (t_)[(x_) * 4 + 2];                            \
        (v_)[(x_) * 4 + 2] = (t_)[(x_) * 4 + 2] - (t_)[(x_) * 4 + 1];                            \
        (v_)[(x_) * 4 + 3] = (t_)[(x_) * 4 + 1] - (t_)[(x_) * 4 + 3];                            \
    } while (0)

    f32x16 acc[16];
    float va[16], vb[16], d[16], t[16];

    // ---- prologue: chunk 0 -> LDS buffer 0, chunk 1 -> staging registers, V of K-step 0, U of steps 0,1
    W16Unit cur;
    W16_DECODE(u, cur);
    W16_ULOAD(0, 0, cur.g);
    W16_ULOAD(1, 1, cur.g);
    W16_LOAD(0);
    W16_STORE(smem);
    W16_LOAD(1);
    {
        const float* c_ = smem + (kh * W16_PS + (2 * ty) * W16_RP + tx);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            d[a * 4 + 0] = c_[a * W16_RP];
            d[a * 4 + 1] = c_[a * W16_RP + W16_HALF];
            d[a * 4 + 2] = c_[a * W16_RP + 1];
            d[a * 4 + 3] = c_[a * W16_RP + W16_HALF + 1];
        }
        W16_TROW(d, t);
#pragma unroll
        for (int x = 0; x < 4; ++x) W16_VROW(t, va, x);
    }

    // one K-step: kk_ (0..7) of chunk parity par_ ; vc_ current V, vn_ next V; the next step reads from
    // LDS offset NOFF (next kk of the same buffer, or kk = 0 of the other buffer)
#define W16_KSTEP(slot_, vc_, vn_, NOFF)                                                         \
    do {                                                                                         \
        w16_block0<NOFF>(acc[0], acc[1], acc[2], acc[3], (vc_)[0], (vc_)[1], (vc_)[2], (vc_)[3], \
                         uring[slot_][0], pa, d);                /* rows 2,0,1 of the next patch landed */ \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        _Pragma("unroll") for (int b = 0; b < 4; ++b) {          /* T = Bt d : rows 0,1,2 (12 VALU) */ \
            t[0 * 4 + b] = d[0 * 4 + b] - d[2 * 4 + b];                                          \
            t[1 * 4 + b] = d[1 * 4 + b] + d[2 * 4 + b];                                          \
            t[2 * 4 + b] = d[2 * 4 + b] - d[1 * 4 + b];                                          \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        w16_block<0>(acc[4], acc[5], acc[6], acc[7], (vc_)[4], (vc_)[5], (vc_)[6], (vc_)[7], uring[slot_][1]); \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        _Pragma("unroll") for (int b = 0; b < 4; ++b) t[3 * 4 + b] = d[1 * 4 + b] - d[3 * 4 + b]; /* row 3 landed */ \
        W16_VROW(t, vn_, 0);                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        w16_block<-1>(acc[8], acc[9], acc[10], acc[11], (vc_)[8], (vc_)[9], (vc_)[10], (vc_)[11], uring[slot_][2]); \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        W16_VROW(t, vn_, 1);                                                                     \
        W16_VROW(t, vn_, 2);                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        w16_block<-1>(acc[12], acc[13], acc[14], acc[15], (vc_)[12], (vc_)[13], (vc_)[14], (vc_)[15], uring[slot_][3]); \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        W16_VROW(t, vn_, 3);                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)

    for (int k = 0; k < n_my; ++k) {
        const int un = u + k * u_stride;
        W16_DECODE(un, cur);
        W16Unit nxt;
        const int un1 = un + u_stride;
        W16_DECODE(un1 < u_end ? un1 : un, nxt);
#pragma unroll
        for (int n = 0; n < 16; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

        // chunks of this unit: stream chunk index c = 4k + q lives in LDS buffer q & 1
#define W16_CHUNK(q_)                                                                            \
    do {                                                                                         \
        constexpr int B0 = ((q_) & 1) * W16_BUF * 4;          /* byte offset of this chunk's buffer */ \
        constexpr int B1 = (((q_) + 1) & 1) * W16_BUF * 4;    /* the other buffer */              \
        /* K-steps 0..6: next step reads kk+1 of the same buffer */                              \
        W16_KSTEP(0, va, vb, B0 + 2 * 1 * W16_PS * 4);                                           \
        W16_ULOAD(0, ((q_) * 8 + 2) % W16_KSTEPS, (((q_) * 8 + 2) < W16_KSTEPS) ? cur.g : nxt.g); \
        W16_KSTEP(1, vb, va, B0 + 2 * 2 * W16_PS * 4);                                           \
        W16_ULOAD(1, ((q_) * 8 + 3) % W16_KSTEPS, (((q_) * 8 + 3) < W16_KSTEPS) ? cur.g : nxt.g); \
        W16_KSTEP(0, va, vb, B0 + 2 * 3 * W16_PS * 4);                                           \
        W16_ULOAD(0, ((q_) * 8 + 4) % W16_KSTEPS, (((q_) * 8 + 4) < W16_KSTEPS) ? cur.g : nxt.g); \
        W16_KSTEP(1, vb, va, B0 + 2 * 4 * W16_PS * 4);                                           \
        W16_ULOAD(1, ((q_) * 8 + 5) % W16_KSTEPS, (((q_) * 8 + 5) < W16_KSTEPS) ? cur.g : nxt.g); \
        W16_KSTEP(0, va, vb, B0 + 2 * 5 * W16_PS * 4);                                           \
        W16_ULOAD(0, ((q_) * 8 + 6) % W16_KSTEPS, (((q_) * 8 + 6) < W16_KSTEPS) ? cur.g : nxt.g); \
        W16_KSTEP(1, vb, va, B0 + 2 * 6 * W16_PS * 4);                                           \
        W16_ULOAD(1, ((q_) * 8 + 7) % W16_KSTEPS, (((q_) * 8 + 7) < W16_KSTEPS) ? cur.g : nxt.g); \
        W16_KSTEP(0, va, vb, B0 + 2 * 7 * W16_PS * 4);                                           \
        W16_ULOAD(0, ((q_) * 8 + 8) % W16_KSTEPS, (((q_) * 8 + 8) < W16_KSTEPS) ? cur.g : nxt.g); \
        /* before the last K-step: staged chunk c+1 -> the other buffer, request chunk c+2 */     \
        W16_STORE(smem + (((q_) + 1) & 1) * W16_BUF);                                            \
        if ((q_) < 3) W16_LOAD(4 * k + (q_) + 2);    /* q = 3: issued after the epilogue (stg is dead there) */ \
        W16_KSTEP(1, vb, va, B1 + 0);                         /* next step = kk 0 of the next chunk */ \
        W16_ULOAD(1, ((q_) * 8 + 9) % W16_KSTEPS, (((q_) * 8 + 9) < W16_KSTEPS) ? cur.g : nxt.g); \
    } while (0)
        W16_CHUNK(0);
        W16_CHUNK(1);
        W16_CHUNK(2);
        W16_CHUNK(3);
#undef W16_CHUNK

        // ---- epilogue of this unit: output transform in registers + fused NHWC epilogue -----------
        // Whole-accumulator (f32x16) arithmetic only: an element access acc[n][r] makes hipcc copy the
        // entire 16-register AGPR tuple per access.  One output column parity j at a time:
        //   R[xi] = sum_nu M[xi][nu] At[j][nu];  Y0 = R0 + R1 + R2;  Y1 = R1 - R2 - R3;  At = [[1,1,1,0],[0,1,-1,-1]]
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // last MFMA's D -> accumulator reads below (not tracked across asm)
        const int cout = cur.g * 32 + xl;
        const float bias = p.bias[cout];
        const float slope = p.act ? 0.2f : 1.0f;
        const int aitem = FUSE ? cur.item / p.add_div : 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16 Y0, Y1;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const f32x16 R = (j == 0) ? (acc[x * 4 + 0] + acc[x * 4 + 1] + acc[x * 4 + 2])
                                          : (acc[x * 4 + 1] - acc[x * 4 + 2] - acc[x * 4 + 3]);
                if (x == 0) Y0 = R;
                if (x == 1) { Y0 += R; Y1 = R; }
                if (x == 2) { Y0 += R; Y1 -= R; }
                if (x == 3) Y1 -= R;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = drow(r, lane);
                const int oy = cur.y0 + 2 * (m >> 4);
                const int ox = cur.x0 + 2 * (m & 15) + j;
                const int xc = min(ox, W - 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int y = oy + i, yc = min(y, H - 1);
                    float o = (i == 0 ? Y0[r] : Y1[r]) + bias;
                    if (FUSE) o += p.addend[(((size_t)aitem * H + yc) * W + xc) * 64 + cout];
                    o = fmaxf(o, slope * o);
                    if (FUSE) o += p.resid[(((size_t)cur.item * H + yc) * W + xc) * 64 + cout];
                    if (y < H && ox < W) p.out[(((size_t)cur.item * H + y) * W + ox) * 64 + cout] = o;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        W16_LOAD(4 * k + 3 + 2);               // stream chunk c+2 of the unit's last chunk (see W16_CHUNK)
    }
#undef W16_KSTEP
#undef W16_VROW
#undef W16_TROW
#undef W16_ULOAD
#undef W16_STORE
#undef W16_LOAD
#undef W16_DECODE
}

template <bool FUSE>
static hipError_t launch_w16_variant(const WinoParams& p, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL((conv_wino16_kernel<FUSE>), dim3(nblocks), dim3(64), W16_LDS_BYTES, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_wino16(const WinoParams& p, hipStream_t s) {
    if (!p.bias || !p.in || !p.upack || !p.out || p.items < 1) return hipErrorInvalidValue;
    const bool fuse = p.addend != nullptr || p.resid != nullptr;
    if (fuse && (!p.addend || !p.resid || p.add_div < 1)) return hipErrorInvalidValue;
    if ((p.H & 1) || (p.W & 1)) return hipErrorInvalidValue;
    static int slots[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!slots[dev]) {
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        slots[dev] = 4 * cus;                                     // one wave per SIMD
    }
    const int nunits = 2 * ((p.W + 2 * W16_TX - 1) / (2 * W16_TX)) * ((p.H + 2 * W16_TY - 1) / (2 * W16_TY)) * p.items;
    int nblocks = slots[dev];
    if (nunits < nblocks) nblocks = (nunits + 7) / 8 * 8;
    return fuse ? launch_w16_variant<true>(p, nblocks, s) : launch_w16_variant<false>(p, nblocks, s);
}

// HWIO [3,3,cin_total,64] rows [cin_begin, cin_begin+64) -> U = G g G^T packed as
// [K-step s = 8*chunk + kk][N-tile g][lane][xi*4 + nu],
// value U[xi][nu][cin = 2s + (lane>>5)][cout = 32g + (lane&31)].
void wino16_pack_weights(const float* hwio, int cin_total, int cin_begin, float* dst) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int s = 0; s < 32; ++s)
        for (int g = 0; g < 2; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int x = 0; x < 4; ++x)
                    for (int nu = 0; nu < 4; ++nu) {
                        const int ci = cin_begin + 2 * s + (lane >> 5);
                        const int co = g * 32 + (lane & 31);
                        double uu = 0.0;
                        for (int a = 0; a < 3; ++a)
                            for (int b = 0; b < 3; ++b)
                                uu += G[x][a] * G[nu][b] * (double)hwio[(((size_t)a * 3 + b) * cin_total + ci) * 64 + co];
                        dst[((((size_t)s * 2 + g) * 64 + lane) * 16) + x * 4 + nu] = (float)uu;
                    }
}

}  // namespace pfnl

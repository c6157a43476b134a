// Fused Winograd F(2x2,3x3) convolution, 64 -> 64 channels, f32 MFMA (gfx950).
//
// Same contract as conv_mfma_kernel<3,FUSE> (conv1_i and both halves of conv2_i of the reference,
// model/pfnl.py:49-51 applied at :66-71) with 2.25x fewer multiplies:
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        per 2x2 output tile / 4x4 input tile,
// the elementwise product summed over input channels being 16 independent GEMMs
// M[xi][nu] (tiles x cout) = V[xi][nu] (tiles x cin) * U[xi][nu] (cin x cout), run on
// v_mfma_f32_32x32x2_f32.  The input transform uses only +-1 (exact adds); U = G g G^T is
// computed on the host in fp64 and rounded once.
//
// Workgroup = 4 waves = 2x16 Winograd tiles (4 rows x 32 columns of output) x ONE 32-channel N-tile
// (the two N-tiles of a spatial tile are separate workgroups, adjacent in dispatch order and on the
// same XCD so the second halo read hits L2).  Wave w owns Winograd row xi = w: its 4 positions
// (nu = 0..3) = 4 accumulators (64 VGPRs) persist across all input-channel chunks.  <=128 VGPRs and
// 36.9 KB LDS allow FOUR independent workgroups per CU (4 waves/SIMD): measured phase timelines
// (tools/wino_timing.py) showed that what starves the matrix pipe is not the K-loop but the
// prologue / epilogue / barrier bubbles of a workgroup coinciding with its neighbour's, so the design
// maximises the number of independent, dynamically scheduled streams rather than their size.
//
// Per K-step (2 channels) a wave does 8 LDS reads + 12 VALU (input transform on the fly from the raw
// halo tile, stored channel-planar with even/odd columns de-interleaved so that lane = tile column
// reads consecutive banks), one coalesced 16-byte buffer load of pre-packed U (L2-resident; SGPR
// descriptor + one lane-offset VGPR + scalar step offsets; a register ring keeps WN_UDEPTH K-steps in
// flight, pinned with sched_barrier) and 4 MFMAs.  The raw tile (6x34 pixels x 16 channels) is
// double-buffered in LDS with register prefetch: one barrier per chunk.  Epilogue: column transform
// in registers, row transform across the four waves through an LDS slab that aliases the raw
// buffers, then the fused bias / shared-base addend / leaky-relu / residual NHWC epilogue with
// 16-byte channel-contiguous accesses.
#include <cstdint>
#include <cstdlib>
#include <string>

#include "common.h"
#include "wino_geom.h"

namespace pfnl {

template <bool FUSE>
__global__ __launch_bounds__(WN_THREADS, PFNL_WINO_WPS) void conv_wino_kernel(WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = Winograd row
    const int tx = lane & 15;
    const int ty = (lane >> 4) & 1;
    const int kh = lane >> 5;
    const int xl = lane & 31;
    const int H = p.H, W = p.W;

    // ---- which (tile, N-tile): block b -> XCD b&7 (observed dispatch), N-tile (b>>3)&1, tile index
    // inside the XCD's contiguous region (b>>4).  Placement only affects L2 reuse, never results.
    const int tiles_x = (W + 2 * WN_TX - 1) / (2 * WN_TX);
    const int tiles_y = (H + 2 * WN_TY - 1) / (2 * WN_TY);
    const int per_item = tiles_x * tiles_y;
    const int ntiles = per_item * p.items;
    const int rs = (ntiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int ng = (blockIdx.x >> 3) & 1;
    const int tile = xcd * rs + (blockIdx.x >> 4);
    if ((int)(blockIdx.x >> 4) >= rs || tile >= ntiles) return;
    const int item = tile / per_item;
    const int rem = tile - item * per_item;
    const int by = rem / tiles_x;
    const int y0 = by * (2 * WN_TY);
    const int x0 = (rem - by * tiles_x) * (2 * WN_TX);

#ifdef PFNL_WINO_TIMING
    long long* dbg = p.dbg ? p.dbg + (size_t)blockIdx.x * 16 : nullptr;
    int dbg_n = 0;
#define WN_STAMP() do { if (dbg && tid == 0 && dbg_n < 16) dbg[dbg_n++] = clock64(); } while (0)
#else
#define WN_STAMP() do {} while (0)
#endif
    WN_STAMP();                                                 // 0: entry

    // U ring first: its latency overlaps the halo-tile staging below.
    constexpr int KS_F4 = 64;                                   // float4 per (chunk, xi, N-tile, kk)
    constexpr int UP_CHUNK_F4 = 4 * 2 * 8 * KS_F4;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.upack), 0, (int)(wino_pack_floats_c * sizeof(float)), 0x00020000);
    const int uvoff = (((xi * 2 + ng) * 8) * KS_F4 + lane) * 16;
#define WN_USTEP(s_) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uvoff, (((s_) >> 3) * UP_CHUNK_F4 + ((s_) & 7) * KS_F4) * 16, 0))
    f32x4 ring[WN_UDEPTH];
#pragma unroll
    for (int d = 0; d < WN_UDEPTH; ++d) ring[d] = WN_USTEP(d);

    // staging descriptors of the raw halo tile (same for every chunk)
    int goff[WN_IN_ITERS];
    int loff[WN_IN_ITERS];
    unsigned ldmask = 0, stmask = 0;
#pragma unroll
    for (int i = 0; i < WN_IN_ITERS; ++i) {
        const int it = tid + i * WN_THREADS;
        goff[i] = 0;
        loff[i] = 0;
        if (it < WN_IN_ITEMS) {
            const int pix = it >> 2, c4 = it & 3;
            const int py = pix / WN_IW, px = pix - py * WN_IW;
            const int gy = y0 + py - 1, gx = x0 + px - 1;
            stmask |= 1u << i;
            loff[i] = (c4 * 4) * WN_PS + py * WN_RP + (px & 1) * WN_HALF + (px >> 1);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                ldmask |= 1u << i;
                goff[i] = (gy * W + gx) * 64 + c4 * 4;
            }
        }
    }
    const float* fin = p.in + (size_t)item * H * W * 64;
    // two register sets: the halo tile of chunk q+2 is requested while chunk q computes (its loads are
    // 64 B per pixel at a 256 B stride and take ~8k cycles under load - longer than one chunk of MFMAs)
    f32x4 rinA[WN_IN_ITERS], rinB[WN_IN_ITERS];
#define WN_LOAD_RAW(q_, rin)                                                                     \
    do {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < WN_IN_ITERS; ++i)                                  \
            rin[i] = *reinterpret_cast<const f32x4*>(fin + (q_) * WN_CK + goff[i]);              \
    } while (0)
#define WN_STORE_RAW(buf_, rin)                                                                       \
    do {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < WN_IN_ITERS; ++i) {                                \
            if ((stmask >> i) & 1u) {                                                            \
                const f32x4 v_ = ((ldmask >> i) & 1u) ? rin[i] : f32x4{0.f, 0.f, 0.f, 0.f};      \
                float* d_ = (buf_) + loff[i];                                                    \
                d_[0] = v_.x;                                                                    \
                d_[WN_PS] = v_.y;                                                                \
                d_[2 * WN_PS] = v_.z;                                                            \
                d_[3 * WN_PS] = v_.w;                                                            \
            }                                                                                    \
        }                                                                                        \
    } while (0)

    // row pair (A,B) and sign of this wave's Winograd row (see wn_kstep_asm); LDS byte addresses
    const int rowA = (xi == 0) ? 0 : 1;
    const int rowB = (xi == 3) ? 3 : 2;
    const float sgn = (xi == 1) ? 1.f : -1.f;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;            // low 32 bits of a flat LDS address = LDS offset
    const unsigned lane_off = (unsigned)(kh * WN_PS + (2 * ty) * WN_RP + tx) * 4u;
    const unsigned pa = lds0 + lane_off + rowA * WN_RP * 4;
    const unsigned pb = lds0 + lane_off + rowB * WN_RP * 4;

    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    WN_LOAD_RAW(0, rinA);
    WN_LOAD_RAW(1, rinB);
    WN_STORE_RAW(smem, rinA);
    __syncthreads();
    WN_STAMP();                                                 // 1: first halo tile in LDS

    // V of the first K-step of a chunk (plain code; every later K-step's V comes out of the asm step before it)
#define WN_VFIRST(q_, v_)                                                                        \
    do {                                                                                         \
        const float* cA_ = smem + ((q_) & 1) * WN_BUF + (lane_off >> 2) + rowA * WN_RP;          \
        const float* cB_ = smem + ((q_) & 1) * WN_BUF + (lane_off >> 2) + rowB * WN_RP;          \
        const float t0_ = fmaf(sgn, cB_[0], cA_[0]);                                             \
        const float t1_ = fmaf(sgn, cB_[WN_HALF], cA_[WN_HALF]);                                 \
        const float t2_ = fmaf(sgn, cB_[1], cA_[1]);                                             \
        const float t3_ = fmaf(sgn, cB_[WN_HALF + 1], cA_[WN_HALF + 1]);                         \
        (v_)[0] = t0_ - t2_;                                                                     \
        (v_)[1] = t1_ + t2_;                                                                     \
        (v_)[2] = t2_ - t1_;                                                                     \
        (v_)[3] = t1_ - t3_;                                                                     \
    } while (0)
#define WN_KSTEP(q_, kk_, vcur_, vnxt_)                                                          \
    do {                                                                                         \
        constexpr int s_ = (q_) * (WN_CK / 2) + (kk_);                                           \
        const f32x4 bc_ = ring[s_ % WN_UDEPTH];                                                  \
        if (s_ + WN_UDEPTH < WN_NSTEP) ring[s_ % WN_UDEPTH] = WN_USTEP(s_ + WN_UDEPTH);          \
        if ((kk_) + 1 < WN_CK / 2) {                                                             \
            wn_kstep_asm<(((q_) & 1) * WN_BUF + 2 * ((kk_) + 1) * WN_PS) * 4>(                    \
                acc[0], acc[1], acc[2], acc[3], vcur_, vnxt_, bc_, pa, pb, sgn);                 \
        } else {                                                                                 \
            acc[0] = mfma32((vcur_)[0], bc_.x, acc[0]);                                          \
            acc[1] = mfma32((vcur_)[1], bc_.y, acc[1]);                                          \
            acc[2] = mfma32((vcur_)[2], bc_.z, acc[2]);                                          \
            acc[3] = mfma32((vcur_)[3], bc_.w, acc[3]);                                          \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
#define WN_CHUNK(q_, rnext_, rfar_)  /* rnext_: holds chunk q+1 (stored at the end); rfar_: receives chunk q+2 */ \
    do {                                                                                         \
        if ((q_) + 2 < WN_NCHUNK) WN_LOAD_RAW((q_) + 2, rfar_);                                  \
        float va[4], vb[4];                                                                      \
        WN_VFIRST(q_, va);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        WN_KSTEP(q_, 0, va, vb); WN_KSTEP(q_, 1, vb, va); WN_KSTEP(q_, 2, va, vb); WN_KSTEP(q_, 3, vb, va); \
        WN_KSTEP(q_, 4, va, vb); WN_KSTEP(q_, 5, vb, va); WN_KSTEP(q_, 6, va, vb); WN_KSTEP(q_, 7, vb, va); \
        if ((q_) + 1 < WN_NCHUNK) WN_STORE_RAW(smem + (((q_) + 1) & 1) * WN_BUF, rnext_);        \
        __syncthreads();                                                                         \
        WN_STAMP();                                             /* 2..5: chunk done */           \
    } while (0)
    static_assert(WN_CK == 16 && WN_NCHUNK == 4, "WN_CHUNK is written out for 8 K-steps x 4 chunks");
    WN_CHUNK(0, rinB, rinA);
    WN_CHUNK(1, rinA, rinB);
    WN_CHUNK(2, rinB, rinA);
    WN_CHUNK(3, rinA, rinB);
#undef WN_CHUNK
#undef WN_KSTEP
#undef WN_VFIRST
#undef WN_LOAD_RAW
#undef WN_STORE_RAW
#undef WN_USTEP

    // ---- epilogue --------------------------------------------------------------------------------
    // thread items: id = k*256 + tid -> channel quad c4 (8 per N-tile), column parity j, tile
    const int c4 = tid & 7;
    const int cbase = ng * 32 + c4 * 4;
    const int aitem = FUSE ? item / p.add_div : 0;
    f32x4 av[2][2], rv[2][2];
    if (FUSE) {                                                 // issue now: overlaps the slab exchange
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int id = k * WN_THREADS + tid;
            const int j = (id >> 3) & 1, etile = id >> 4;
            const int ox = min(x0 + 2 * (etile & 15) + j, W - 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yc = min(y0 + 2 * (etile >> 4) + i, H - 1);
                av[k][i] = *reinterpret_cast<const f32x4*>(p.addend + (((size_t)aitem * H + yc) * W + ox) * 64 + cbase);
                rv[k][i] = *reinterpret_cast<const f32x4*>(p.resid + (((size_t)item * H + yc) * W + ox) * 64 + cbase);
            }
        }
    }
    // column transform over nu in registers: At = [[1,1,1,0],[0,1,-1,-1]]
    float* slab = smem + xi * (2 * 32 * WN_ES);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r];
        const int etile = drow(r, lane);
        slab[(0 * 32 + etile) * WN_ES + xl] = m0 + m1 + m2;
        slab[(1 * 32 + etile) * WN_ES + xl] = m1 - m2 - m3;
    }
    __syncthreads();
    WN_STAMP();                                                 // 6: slab exchanged
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + cbase);
    const float slope = p.act ? 0.2f : 1.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int id = k * WN_THREADS + tid;
        const int j = (id >> 3) & 1, etile = id >> 4;
        const int oy = y0 + 2 * (etile >> 4);
        const int ox = x0 + 2 * (etile & 15) + j;
        // row transform over xi across the four waves
        const float* sp = smem + (j * 32 + etile) * WN_ES + c4 * 4;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(sp);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(sp + 1 * 2 * 32 * WN_ES);
        const f32x4 r2 = *reinterpret_cast<const f32x4*>(sp + 2 * 2 * 32 * WN_ES);
        const f32x4 r3 = *reinterpret_cast<const f32x4*>(sp + 3 * 2 * 32 * WN_ES);
        f32x4 yv[2];
        yv[0] = r0 + r1 + r2;
        yv[1] = r1 - r2 - r3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = oy + i;
            f32x4 o = yv[i] + bias4;
            if (FUSE) o += av[k][i];
            o.x = fmaxf(o.x, slope * o.x);
            o.y = fmaxf(o.y, slope * o.y);
            o.z = fmaxf(o.z, slope * o.z);
            o.w = fmaxf(o.w, slope * o.w);
            if (FUSE) o += rv[k][i];
            if (ox < W && y < H)
                *reinterpret_cast<f32x4*>(p.out + (((size_t)item * H + y) * W + ox) * 64 + cbase) = o;
        }
    }
    WN_STAMP();                                                 // 7: stores issued
}

template <bool FUSE>
static hipError_t launch_wino_variant(const WinoParams& p, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL((conv_wino_kernel<FUSE>), dim3(nblocks), dim3(WN_THREADS), WN_LDS_BYTES, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_wino(const WinoParams& p, hipStream_t s) {
    if (!p.bias || !p.in || !p.upack || !p.out || p.items < 1) return hipErrorInvalidValue;
    const bool fuse = p.addend != nullptr || p.resid != nullptr;
    if (fuse && (!p.addend || !p.resid || p.add_div < 1)) return hipErrorInvalidValue;
    if ((p.H & 1) || (p.W & 1)) return hipErrorInvalidValue;     // 2x2 tiles must not straddle the border
    const int ntiles = ((p.W + 2 * WN_TX - 1) / (2 * WN_TX)) * ((p.H + 2 * WN_TY - 1) / (2 * WN_TY)) * p.items;
    const int rs = (ntiles + 7) / 8;
    const int nblocks = 16 * rs;                                  // 8 XCD regions x 2 N-tiles x rs tiles
    return fuse ? launch_wino_variant<true>(p, nblocks, s) : launch_wino_variant<false>(p, nblocks, s);
}

size_t wino_pack_floats() { return (size_t)wino_pack_floats_c; }

// HWIO [3,3,cin_total,cout<=64] rows [cin_begin, cin_begin+64) -> U = G g G^T packed as
// [chunk q][xi][N-tile g][kk][lane][nu], value U[xi][nu][cin = 16q + 2kk + (lane>>5)][cout = 32g + (lane&31)].
void wino_pack_weights(const float* hwio, int cin_total, int cin_begin, float* dst, int cout) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int q = 0; q < 4; ++q)
        for (int x = 0; x < 4; ++x)
            for (int g = 0; g < 2; ++g)
                for (int kk = 0; kk < 8; ++kk)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int nu = 0; nu < 4; ++nu) {
                            const int ci = cin_begin + q * 16 + 2 * kk + (lane >> 5);
                            const int co = g * 32 + (lane & 31);
                            double u = 0.0;
                            for (int a = 0; a < 3 && co < cout; ++a)      // output channels >= cout: zero weights
                                for (int b = 0; b < 3; ++b)
                                    u += G[x][a] * G[nu][b] *
                                         (double)hwio[(((size_t)a * 3 + b) * cin_total + ci) * cout + co];
                            if (x == 2) u = -u;   // the kernel computes d1 - d2 for this row (see wn_kstep_asm)
                            dst[(((((size_t)q * 4 + x) * 2 + g) * 8 + kk) * 64 + lane) * 4 + nu] = (float)u;
                        }
}

}  // namespace pfnl

// Fused Winograd F(2x2,3x3) convolution, 64 -> 64 channels, f32 MFMA (gfx950), persistent.
//
// Same contract as conv_mfma_kernel<3,FUSE> (conv1_i and both halves of conv2_i of the reference,
// model/pfnl.py:49-51 applied at :66-71) with 2.25x fewer multiplies:
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        per 2x2 output tile / 4x4 input tile,
// the elementwise product summed over input channels being 16 independent GEMMs
// M[xi][nu] (tiles x cout) = V[xi][nu] (tiles x cin) * U[xi][nu] (cin x cout), run on
// v_mfma_f32_32x32x2_f32.  The input transform uses only +-1 (exact adds); U = G g G^T is
// computed on the host in fp64 and rounded once.
//
// Workgroup = 8 waves; a workgroup TILE = 2x16 Winograd tiles (4 rows x 32 columns of output) x 64
// cout.  Wave (xi, g) owns Winograd row xi and N-tile g: its 4 positions (nu = 0..3) = 4
// accumulators (64 VGPRs) persist across all input-channel chunks of a tile; 4 waves per SIMD hide
// LDS/VALU latency behind the matrix pipe.  Per K-step (2 channels) a wave does 8 LDS reads + 12
// VALU (input transform on the fly from the raw halo tile, stored channel-planar with even/odd
// columns de-interleaved so that lane = tile column reads consecutive banks), one coalesced 16-byte
// global load of pre-packed U (L2-resident, identical for every tile; a 4-deep register ring keeps 4
// K-steps in flight) and 4 MFMAs.
//
// PERSISTENT: the grid is sized to the resident slots (2 workgroups per CU) and every workgroup walks
// a sequence of tiles.  The stream of (tile, chunk) pairs is software-pipelined across tile borders:
// the next chunk's halo tile (6x34 pixels x 16 channels) is fetched into registers while the current
// chunk's MFMAs run and lands in the other LDS buffer (one barrier per chunk), the U ring simply
// wraps, and epilogue stores are fire-and-forget - so no workgroup ever waits for HBM with an idle
// matrix pipe except on its very first chunk.  Tiles are dealt out XCD-contiguously (block b runs on
// XCD b % 8) so that neighbouring tiles' halos hit the same L2.
// Epilogue per tile: column transform in registers, row transform across the four xi-waves through
// an LDS slab (one output column parity j at a time), then the fused bias / shared-base addend /
// leaky-relu / residual NHWC epilogue with 16-byte channel-contiguous accesses.
#include <cstdlib>
#include <string>

#include "common.h"

namespace pfnl {

constexpr int WN_TX = 16, WN_TY = 2;              // Winograd tiles per workgroup tile (x, y)
constexpr int WN_IW = 2 * WN_TX + 2;              // 34 input columns
constexpr int WN_IH = 2 * WN_TY + 2;              // 6 input rows
constexpr int WN_HALF = 20;                       // floats per column-parity half row (17 used)
constexpr int WN_RP = 2 * WN_HALF;                // row pitch 40: two tile rows are 80 = 16 mod 32 banks apart
constexpr int WN_PS = WN_IH * WN_RP + 1;          // plane stride 241 (odd)
constexpr int WN_CK = 16;
constexpr int WN_NCHUNK = 64 / WN_CK;
constexpr int WN_BUF = WN_CK * WN_PS;             // floats per raw buffer (3856)
constexpr int WN_ES = 68;                         // slab row stride (floats), 16-byte aligned rows
constexpr int WN_SLAB = 4 * 32 * WN_ES;           // [xi][tile][cout] for ONE column parity j
constexpr int WN_SLAB_OFF = 2 * WN_BUF;           // slab lives after the two raw buffers (7712, 16B aligned)
constexpr int WN_LDS_FLOATS = WN_SLAB_OFF + WN_SLAB;
constexpr size_t WN_LDS_BYTES = size_t(WN_LDS_FLOATS) * sizeof(float);   // 65 664 B -> 2 workgroups / CU
constexpr int WN_THREADS = 512;
constexpr int WN_IN_ITEMS = WN_IH * WN_IW * (WN_CK / 4);                 // 816 float4 pieces
constexpr int WN_IN_ITERS = (WN_IN_ITEMS + WN_THREADS - 1) / WN_THREADS; // 2
constexpr int WN_UDEPTH = PFNL_WINO_UDEPTH;       // K-steps of U kept in flight
constexpr int WN_NSTEP = WN_NCHUNK * (WN_CK / 2); // 32 K-steps per tile
constexpr int wino_pack_floats_c = 16 * 64 * 64;
static_assert(WN_NSTEP % WN_UDEPTH == 0, "ring slots must line up across tiles");
static_assert((WN_SLAB_OFF * 4) % 16 == 0, "slab must be 16-byte aligned");

struct WinoTileDesc {          // per-thread staging descriptor of one workgroup tile
    int goff[WN_IN_ITERS];
    unsigned ldmask;
    int item, x0, y0;
};

template <bool FUSE>
__global__ __launch_bounds__(WN_THREADS, 4) void conv_wino_kernel(WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wv & 3;                                     // Winograd row of this wave
    const int ng = wv >> 2;                                    // N-tile of this wave
    const int tx = lane & 15;
    const int ty = (lane >> 4) & 1;
    const int kh = lane >> 5;
    const int xl = lane & 31;
    const int H = p.H, W = p.W;

    // ---- tile walk: XCD-contiguous regions, round-robin inside a region ------------------------
    const int tiles_x = (W + 2 * WN_TX - 1) / (2 * WN_TX);
    const int tiles_y = (H + 2 * WN_TY - 1) / (2 * WN_TY);
    const int per_item = tiles_x * tiles_y;
    const int ntiles = per_item * p.items;
    int t, t_end, t_stride;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
        const int rs = (ntiles + 7) >> 3;
        t = xcd * rs + j;
        t_end = min(ntiles, (xcd + 1) * rs);
        t_stride = per_xcd;
    } else {
        t = blockIdx.x;
        t_end = ntiles;
        t_stride = gridDim.x;
    }
    if (t >= t_end) return;

    // Staged float4 piece i of this thread: it = tid + i*512 -> pixel (py,px) of the 6x34 halo tile,
    // channel quad c4.  Coordinates are recomputed where needed (a few VALU per tile) instead of
    // being kept live: the kernel sits at the 128-VGPR limit of 4 waves/SIMD and a spill reload is a
    // VMEM op that would force vmcnt(0) and drain the U ring.
#define WN_PIECE(i_, py_, px_, c4_, ok_)                                                          \
    const int it_##i_ = tid + (i_) * WN_THREADS;                                                  \
    const bool ok_ = it_##i_ < WN_IN_ITEMS;                                                       \
    const int c4_ = it_##i_ & 3;                                                                  \
    const int py_ = (it_##i_ >> 2) / WN_IW;                                                       \
    const int px_ = (it_##i_ >> 2) - py_ * WN_IW;
#define WN_DESCRIBE(tile_, d_)                                                                    \
    do {                                                                                          \
        (d_).item = (tile_) / per_item;                                                           \
        const int rem_ = (tile_) - (d_).item * per_item;                                          \
        const int by_ = rem_ / tiles_x;                                                           \
        (d_).y0 = by_ * (2 * WN_TY);                                                              \
        (d_).x0 = (rem_ - by_ * tiles_x) * (2 * WN_TX);                                           \
        (d_).ldmask = 0;                                                                          \
        _Pragma("unroll") for (int i = 0; i < WN_IN_ITERS; ++i) {                                 \
            WN_PIECE(i, py_, px_, c4_, ok_)                                                       \
            const int gy_ = (d_).y0 + py_ - 1, gx_ = (d_).x0 + px_ - 1;                           \
            (d_).goff[i] = 0;                                                                     \
            if (ok_ && gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W) {                              \
                (d_).ldmask |= 1u << i;                                                           \
                (d_).goff[i] = (gy_ * W + gx_) * 64 + c4_ * 4;                                    \
            }                                                                                     \
        }                                                                                         \
    } while (0)

    f32x4 rin[WN_IN_ITERS];
    unsigned rin_mask = 0;          // ldmask of the tile whose data sits in rin
#define WN_LOAD_RAW(desc_, q_)                                                                    \
    do {                                                                                          \
        const float* fin_ = p.in + (size_t)(desc_).item * H * W * 64 + (q_) * WN_CK;             \
        _Pragma("unroll") for (int i = 0; i < WN_IN_ITERS; ++i)                                   \
            rin[i] = *reinterpret_cast<const f32x4*>(fin_ + (desc_).goff[i]);                     \
        rin_mask = (desc_).ldmask;                                                                \
    } while (0)
#define WN_STORE_RAW(buf_)                                                                        \
    do {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < WN_IN_ITERS; ++i) {                                 \
            WN_PIECE(i, py_, px_, c4_, ok_)                                                       \
            if (ok_) {                                                                            \
                const f32x4 v_ = ((rin_mask >> i) & 1u) ? rin[i] : f32x4{0.f, 0.f, 0.f, 0.f};     \
                float* d_ = (buf_) + (c4_ * 4) * WN_PS + py_ * WN_RP + (px_ & 1) * WN_HALF + (px_ >> 1); \
                d_[0] = v_.x;                                                                     \
                d_[WN_PS] = v_.y;                                                                 \
                d_[2 * WN_PS] = v_.z;                                                             \
                d_[3 * WN_PS] = v_.w;                                                             \
            }                                                                                     \
        }                                                                                         \
    } while (0)

    // B^T row xi = s0 * e[a0] + s1 * e[a1]
    const int a0 = (xi == 0) ? 0 : 1;
    const int a1 = (xi == 3) ? 3 : 2;
    const float s0 = (xi == 2) ? -1.f : 1.f;
    const float s1 = (xi == 0 || xi == 3) ? -1.f : 1.f;
    const int abase = kh * WN_PS + (2 * ty) * WN_RP + tx;      // channel half, tile row, tile column
    const int r0off = abase + a0 * WN_RP;
    const int r1off = abase + a1 * WN_RP;

    // packed U: [chunk][xi][N-tile][kk][lane][nu]  (one float4 per lane per K-step)
    constexpr int KS_F4 = 64;
    constexpr int UP_CHUNK_F4 = 4 * 2 * 8 * KS_F4;
    // buffer loads: SGPR descriptor + ONE per-lane VGPR offset + a scalar offset per K-step, so the 32
    // step addresses cost no VGPRs (plain pointers get hoisted out of the tile loop as 32 VGPR pairs)
    const __amdgpu_buffer_rsrc_t urs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.upack), 0, (int)(wino_pack_floats_c * sizeof(float)), 0x00020000);
    const int uvoff = (((xi * 2 + ng) * 8) * KS_F4 + lane) * 16;
#define WN_USTEP(s_) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uvoff, (((s_) >> 3) * UP_CHUNK_F4 + ((s_) & 7) * KS_F4) * 16, 0))
    f32x4 ring[WN_UDEPTH];
#pragma unroll
    for (int d = 0; d < WN_UDEPTH; ++d) ring[d] = WN_USTEP(d);

    // epilogue constants: thread item = (tile, channel quad), one per column parity j
    const int c4 = tid & 15;
    const int etile = tid >> 4;                                // 0..31
    const float slope = p.act ? 0.2f : 1.0f;
    float* slab = smem + WN_SLAB_OFF;

#ifdef PFNL_WINO_TIMING
    long long* dbg = p.dbg ? p.dbg + (size_t)blockIdx.x * 64 : nullptr;
    int dbg_n = 0;
#define WN_STAMP() do { if (dbg && tid == 0 && dbg_n < 64) dbg[dbg_n++] = clock64(); } while (0)
#else
#define WN_STAMP() do {} while (0)
#endif
    WN_STAMP();                                                 // 0: kernel entry (after setup)
    WinoTileDesc cur_d, nxt_d;
    WN_DESCRIBE(t, cur_d);
    WN_LOAD_RAW(cur_d, 0);
    WN_STORE_RAW(smem);
    __syncthreads();
    WN_STAMP();                                                 // 1: first halo tile in LDS

    for (;;) {
        const int tn = t + t_stride;
        const bool has_next = tn < t_end;
        const int tdesc = has_next ? tn : t;
        WN_DESCRIBE(tdesc, nxt_d);

        f32x16 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

#pragma unroll
        for (int q = 0; q < WN_NCHUNK; ++q) {
            const float* cur = smem + (q & 1) * WN_BUF;
            if (q + 1 < WN_NCHUNK) {
                WN_LOAD_RAW(cur_d, q + 1);                     // next chunk of this tile
            } else {
                WN_LOAD_RAW(nxt_d, 0);                         // first chunk of the next tile (or a harmless re-read)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < WN_CK / 2; ++kk) {
                const int s = q * (WN_CK / 2) + kk;
                const f32x4 bc = ring[s % WN_UDEPTH];
                ring[s % WN_UDEPTH] = WN_USTEP((s + WN_UDEPTH) % WN_NSTEP);   // wraps into the next tile
                __builtin_amdgcn_sched_barrier(0);             // pins one U load per K-step ahead of its MFMAs
                const float* c0 = cur + r0off + (2 * kk) * WN_PS;
                const float* c1 = cur + r1off + (2 * kk) * WN_PS;
                // t[b] = sum_a Bt[xi][a] d[a][b];  column b lives at (b&1)*HALF + (b>>1)
                const float t0 = fmaf(s1, c1[0], s0 * c0[0]);
                const float t1 = fmaf(s1, c1[WN_HALF], s0 * c0[WN_HALF]);
                const float t2 = fmaf(s1, c1[1], s0 * c0[1]);
                const float t3 = fmaf(s1, c1[WN_HALF + 1], s0 * c0[WN_HALF + 1]);
                acc[0] = mfma32(t0 - t2, bc.x, acc[0]);        // V[xi][nu] = sum_b Bt[nu][b] t[b]
                acc[1] = mfma32(t1 + t2, bc.y, acc[1]);
                acc[2] = mfma32(t2 - t1, bc.z, acc[2]);
                acc[3] = mfma32(t1 - t3, bc.w, acc[3]);
            }
            WN_STAMP();                                         // MFMAs of chunk q issued
            WN_STORE_RAW(smem + ((q + 1) & 1) * WN_BUF);       // the buffer nobody reads in this iteration
            __syncthreads();
            WN_STAMP();                                         // chunk barrier passed
        }

        // ---- epilogue of tile t (the next tile's first chunk is already in LDS) -----------------
        const int oy = cur_d.y0 + 2 * (etile >> 4);
        const int oxb = cur_d.x0 + 2 * (etile & 15);
        const int aitem = FUSE ? cur_d.item / p.add_div : 0;
        const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + c4 * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // column transform over nu in registers: At = [[1,1,1,0],[0,1,-1,-1]]
            if (j == 1) __syncthreads();                       // everyone finished reading slab(j=0)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = (j == 0) ? (acc[0][r] + acc[1][r] + acc[2][r]) : (acc[1][r] - acc[2][r] - acc[3][r]);
                slab[(xi * 32 + drow(r, lane)) * WN_ES + ng * 32 + xl] = v;
            }
            const int ox = oxb + j;
            const bool inx = ox < W;
            const int oxc = inx ? ox : W - 1;
            f32x4 av[2], rv[2];
            if (FUSE) {                                        // issue before the barrier: overlaps it
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int yc = min(oy + i, H - 1);
                    av[i] = *reinterpret_cast<const f32x4*>(p.addend + (((size_t)aitem * H + yc) * W + oxc) * 64 + c4 * 4);
                    rv[i] = *reinterpret_cast<const f32x4*>(p.resid + (((size_t)cur_d.item * H + yc) * W + oxc) * 64 + c4 * 4);
                }
            }
            __syncthreads();
            // row transform over xi across the four waves
            const float* sp = slab + etile * WN_ES + c4 * 4;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(sp);
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(sp + 1 * 32 * WN_ES);
            const f32x4 r2 = *reinterpret_cast<const f32x4*>(sp + 2 * 32 * WN_ES);
            const f32x4 r3 = *reinterpret_cast<const f32x4*>(sp + 3 * 32 * WN_ES);
            f32x4 yv[2];
            yv[0] = r0 + r1 + r2;
            yv[1] = r1 - r2 - r3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int y = oy + i;
                const int yc = min(y, H - 1);
                f32x4 o = yv[i] + bias4;
                if (FUSE) o += av[i];
                o.x = fmaxf(o.x, slope * o.x);
                o.y = fmaxf(o.y, slope * o.y);
                o.z = fmaxf(o.z, slope * o.z);
                o.w = fmaxf(o.w, slope * o.w);
                if (FUSE) o += rv[i];
                if (inx && y < H)
                    *reinterpret_cast<f32x4*>(p.out + (((size_t)cur_d.item * H + yc) * W + oxc) * 64 + c4 * 4) = o;
            }
            WN_STAMP();                                         // epilogue half j done
        }
        if (!has_next) break;
        t = tn;
        cur_d = nxt_d;
        // slab(j=1) readers vs the next tile's slab(j=0) writers are separated by the four chunk barriers
    }
#undef WN_DESCRIBE
#undef WN_PIECE
#undef WN_LOAD_RAW
#undef WN_STORE_RAW
#undef WN_USTEP
}

template <bool FUSE>
static hipError_t launch_wino_variant(const WinoParams& p, int nblocks, hipStream_t s) {
    static bool attr_set[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<FUSE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)WN_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((conv_wino_kernel<FUSE>), dim3(nblocks), dim3(WN_THREADS), WN_LDS_BYTES, s, p);
    return hipGetLastError();
}

static int wino_resident_slots() {
    static int slots[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 512;
    if (!slots[dev]) {
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        slots[dev] = 2 * cus;                                   // 2 workgroups per CU (LDS 65.7 KB, <=128 VGPRs)
    }
    return slots[dev];
}

hipError_t launch_conv_wino(const WinoParams& p, hipStream_t s) {
    if (!p.bias || !p.in || !p.upack || !p.out || p.items < 1) return hipErrorInvalidValue;
    const bool fuse = p.addend != nullptr || p.resid != nullptr;
    if (fuse && (!p.addend || !p.resid || p.add_div < 1)) return hipErrorInvalidValue;
    if ((p.H & 1) || (p.W & 1)) return hipErrorInvalidValue;     // 2x2 tiles must not straddle the border
    const int ntiles = ((p.W + 2 * WN_TX - 1) / (2 * WN_TX)) * ((p.H + 2 * WN_TY - 1) / (2 * WN_TY)) * p.items;
    int nblocks = wino_resident_slots();
    static const bool one_tile_per_wg = std::getenv("PFNL_WINO_GRID") && std::string(std::getenv("PFNL_WINO_GRID")) == "tiles";
    if (ntiles < nblocks || one_tile_per_wg) nblocks = (ntiles + 7) / 8 * 8;   // small problems: one tile per workgroup
    return fuse ? launch_wino_variant<true>(p, nblocks, s) : launch_wino_variant<false>(p, nblocks, s);
}

size_t wino_pack_floats() { return (size_t)wino_pack_floats_c; }

// HWIO [3,3,cin_total,64] rows [cin_begin, cin_begin+64) -> U = G g G^T packed as
// [chunk q][xi][N-tile g][kk][lane][nu], value U[xi][nu][cin = 16q + 2kk + (lane>>5)][cout = 32g + (lane&31)].
void wino_pack_weights(const float* hwio, int cin_total, int cin_begin, float* dst) {
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
                            for (int a = 0; a < 3; ++a)
                                for (int b = 0; b < 3; ++b)
                                    u += G[x][a] * G[nu][b] *
                                         (double)hwio[(((size_t)a * 3 + b) * cin_total + ci) * 64 + co];
                            dst[(((((size_t)q * 4 + x) * 2 + g) * 8 + kk) * 64 + lane) * 4 + nu] = (float)u;
                        }
}

}  // namespace pfnl

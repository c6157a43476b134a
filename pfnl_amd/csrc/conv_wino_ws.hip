// Fused Winograd F(2x2,3x3) convolution, 64 -> 64 channels, f32 MFMA - persistent, wave-specialised.
//
// Same arithmetic, packed U and tile geometry as conv_wino_kernel (conv_wino.hip; reference
// model/pfnl.py:49-51 applied at :66-71); what changes is who does what.  conv_wino.hip measured its
// own limit: a wave spends ~40 % of its life in the prologue (first halo chunk from HBM), the slab
// exchange and the NHWC epilogue, so on average only ~1.5 of the 3 resident waves per SIMD are inside
// the K-loop.  Here a workgroup is 8 waves that never change roles:
//
//   waves 0-3  (matrix waves, Winograd row xi = wave): nothing but the K-loop.  Their only vector-
//              memory traffic is the L2-resident U stream, so no HBM-latency load ever sits in front
//              of it in the in-order vmcnt queue.  At the end of a unit they drop the column-transformed
//              accumulators into an LDS slab and go straight on to the next unit.
//   waves 4-7  (helper waves): stage the raw halo chunks global -> registers -> LDS two phases ahead,
//              and run the previous unit's row transform + fused bias / addend / leaky-relu / residual
//              epilogue out of the slab while the matrix waves are already computing.
//
// Unit = (spatial tile of 2x16 Winograd tiles, 32-channel N-tile); a workgroup walks a strided list of
// units of its XCD's contiguous tile range (the two N-tiles of a tile run at the same time on
// neighbouring workgroups, so the second halo read hits L2).  One barrier per 16-channel chunk is the
// only synchronisation: in phase c the matrix waves read raw buffer c&1 while the helpers fill buffer
// (c+1)&1; the slab is written at the end of a unit's 4th phase and read during the next unit's first two.
// 67.7 KB LDS and <= 128 VGPRs -> two workgroups per CU = per SIMD two matrix waves + two helpers.
#include <cstdint>

#include "common.h"
#include "wino_geom.h"

namespace pfnl {

constexpr int WS_THREADS = 512;
constexpr int WS_RAW_FLOATS = 2 * WP_BUF;                                  // two raw halo buffers
constexpr int WS_LDS_FLOATS = WS_RAW_FLOATS + WN_SLAB;
constexpr size_t WS_LDS_BYTES = size_t(WS_LDS_FLOATS) * sizeof(float);     // 67 712 B
constexpr int WS_MAX_WG_PER_XCD = 64;                                      // 32 CUs x 2 workgroups
static_assert((WS_RAW_FLOATS * sizeof(float)) % 16 == 0, "slab must stay 16-byte aligned");

template <bool FUSE>
__global__ __launch_bounds__(WS_THREADS, 2) void conv_wino_ws_kernel(WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const slabm = smem + WS_RAW_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;

    // ---- this workgroup's unit list: XCD b&7 owns tiles [xcd*rs, xcd*rs+rs); its workgroup j takes
    // units j, j+wpx, ... of (tile, N-tile) pairs, N-tile minor.  wpx is even, so the N-tile is fixed.
    const int tiles_x = (W + 2 * WN_TX - 1) / (2 * WN_TX);
    const int tiles_y = (H + 2 * WN_TY - 1) / (2 * WN_TY);
    const int per_item = tiles_x * tiles_y;
    const int ntiles = per_item * p.items;
    const int rs = (ntiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int wpx = gridDim.x >> 3;
    const int tbeg = xcd * rs;
    const int nux = 2 * min(rs, ntiles - tbeg);
    if (j >= nux) return;
    const int nu = (nux - j + wpx - 1) / wpx;
    const int ng = j & 1;
#define WS_UNIT(i_, item_, y0_, x0_)                                  \
    do {                                                              \
        const int t_ = tbeg + ((j + (i_) * wpx) >> 1);                \
        item_ = t_ / per_item;                                        \
        const int rem_ = t_ - item_ * per_item;                       \
        const int by_ = rem_ / tiles_x;                               \
        y0_ = by_ * (2 * WN_TY);                                      \
        x0_ = (rem_ - by_ * tiles_x) * (2 * WN_TX);                   \
    } while (0)

#ifdef PFNL_WINO_TIMING
    long long* dbg = p.dbg ? p.dbg + (size_t)blockIdx.x * 128 + (wave < 4 ? 0 : 64) : nullptr;
    int dbg_n = 0;
    const bool dbg_on = dbg && (lane == 0) && (wave == 0 || wave == 4);
#define WS_STAMP() do { if (dbg_on && dbg_n < 64) dbg[dbg_n++] = clock64(); } while (0)
#else
#define WS_STAMP() do {} while (0)
#endif
    if (wave < 4) {
        // =================================== matrix waves ===========================================
#ifndef PFNL_WS_MPRIO
#define PFNL_WS_MPRIO 0
#endif
#ifndef PFNL_WS_HPRIO
#define PFNL_WS_HPRIO 3
#endif
        __builtin_amdgcn_s_setprio(PFNL_WS_MPRIO);
        const int xi = wave;
        const int tx = lane & 15;
        const int ty = (lane >> 4) & 1;
        const int kh = lane >> 5;
        const int xl = lane & 31;

        constexpr int KS_F4 = 64;                                   // float4 per (chunk, xi, N-tile, kk)
        constexpr int UP_CHUNK_F4 = 4 * 2 * 8 * KS_F4;
        const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.upack), 0, (int)(wino_pack_floats_c * sizeof(float)), 0x00020000);
        const int uvoff = (((xi * 2 + ng) * 8) * KS_F4 + lane) * 16;
#define WS_USTEP(s_) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uvoff, ((((s_) & 31) >> 3) * UP_CHUNK_F4 + ((s_) & 7) * KS_F4) * 16, 0))
        f32x4 ring[WN_UDEPTH];
#pragma unroll
        for (int d = 0; d < WN_UDEPTH; ++d) ring[d] = WS_USTEP(d);
#ifdef PFNL_WS_NO_U     /* timing experiment only: wrong results */
#define WS_RING_REFILL(s_) do {} while (0)
#else
#define WS_RING_REFILL(s_) ring[(s_) % WN_UDEPTH] = WS_USTEP((s_) + WN_UDEPTH)
#endif

        const int rowA = (xi == 0) ? 0 : 1;
        const int rowB = (xi == 3) ? 3 : 2;
        const float sgn = (xi == 1) ? 1.f : -1.f;
        const f32x2 sg2 = {sgn, sgn};
        const unsigned lds0 = (unsigned)(uintptr_t)smem;
        const int lane_a = kh * WP_PS + (2 * ty + rowA) * WP_RP + 2 * tx;          // floats
        const int lane_b = kh * WP_PS + (2 * ty + rowB) * WP_RP + 2 * tx;
        const unsigned pa = lds0 + (unsigned)lane_a * 4u;
        const unsigned pb = lds0 + (unsigned)lane_b * 4u;
        float* const slab = slabm + xi * (2 * 32 * WN_ES);

        f32x16 acc[4];
#define WS_VFIRST(q_, v03_, v12_)                                                                \
    do {                                                                                         \
        const f32x2* cA_ = reinterpret_cast<const f32x2*>(smem + ((q_) & 1) * WP_BUF + lane_a);   \
        const f32x2* cB_ = reinterpret_cast<const f32x2*>(smem + ((q_) & 1) * WP_BUF + lane_b);   \
        const f32x2 t01_ = cA_[0] + sg2 * cB_[0];                                                \
        const f32x2 t23_ = cA_[1] + sg2 * cB_[1];                                                \
        (v03_) = t01_ - t23_;                                    /* (t0 - t2, t1 - t3) */        \
        (v12_).x = t01_.y + t23_.x;                                                              \
        (v12_).y = t23_.x - t01_.y;                                                              \
    } while (0)
#define WS_KSTEP(q_, kk_, c03_, c12_, n03_, n12_)                                                \
    do {                                                                                         \
        constexpr int s_ = (q_) * (WN_CK / 2) + (kk_);                                           \
        const f32x4 bc_ = ring[s_ % WN_UDEPTH];                                                  \
        WS_RING_REFILL(s_);                                     /* wraps into the next unit */   \
        constexpr int off_ = (((q_) & 1) * WP_BUF + 2 * ((kk_) + 1) * WP_PS) * 4;                \
        if ((q_) == 0 && (kk_) == 0) {                                                           \
            wp_kstep_asm_zero<off_>(acc[0], acc[1], acc[2], acc[3], (c03_).x, (c12_).x, (c12_).y, (c03_).y, n03_, \
                                    n12_, bc_, pa, pb, sg2);                                     \
        } else if ((kk_) + 1 < WN_CK / 2) {                                                      \
            wp_kstep_asm<off_>(acc[0], acc[1], acc[2], acc[3], (c03_).x, (c12_).x, (c12_).y, (c03_).y, n03_, n12_, \
                               bc_, pa, pb, sg2);                                                \
        } else {                                                                                 \
            acc[0] = mfma32((c03_).x, bc_.x, acc[0]);                                            \
            acc[1] = mfma32((c12_).x, bc_.y, acc[1]);                                            \
            acc[2] = mfma32((c12_).y, bc_.z, acc[2]);                                            \
            acc[3] = mfma32((c03_).y, bc_.w, acc[3]);                                            \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
#define WS_MCHUNK(q_)                                                                            \
    do {                                                                                         \
        f32x2 pA_, pB_, pC_, pD_;                               /* (V0,V3) and (V1,V2), two sets */ \
        WS_VFIRST(q_, pA_, pB_);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        WS_KSTEP(q_, 0, pA_, pB_, pC_, pD_); WS_KSTEP(q_, 1, pC_, pD_, pA_, pB_);                \
        WS_KSTEP(q_, 2, pA_, pB_, pC_, pD_); WS_KSTEP(q_, 3, pC_, pD_, pA_, pB_);                \
        WS_KSTEP(q_, 4, pA_, pB_, pC_, pD_); WS_KSTEP(q_, 5, pC_, pD_, pA_, pB_);                \
        WS_KSTEP(q_, 6, pA_, pB_, pC_, pD_); WS_KSTEP(q_, 7, pC_, pD_, pA_, pB_);                \
    } while (0)
        static_assert(WN_CK == 16 && WN_NCHUNK == 4, "written out for 8 K-steps x 4 chunks");

        WS_STAMP();
        __syncthreads();                                            // B0: chunk 0 of the first unit is in LDS
        for (int i = 0; i < nu; ++i) {
            WS_STAMP();
            WS_MCHUNK(0);
            WS_STAMP();
            __syncthreads();
            WS_STAMP();
            WS_MCHUNK(1);
            WS_STAMP();
            __syncthreads();
            WS_STAMP();
            WS_MCHUNK(2);
            WS_STAMP();
            __syncthreads();
            WS_STAMP();
            WS_MCHUNK(3);
            WS_STAMP();
            // column transform over nu in registers (At = [[1,1,1,0],[0,1,-1,-1]]) -> slab[xi][j][tile][cout]
            {
                const f32x16 s0 = acc[0] + acc[1] + acc[2];     // whole-vector form: packed-f32 adds
                const f32x16 s1 = acc[1] - acc[2] - acc[3];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int etile = drow(r, lane);
                    slab[(0 * 32 + etile) * WN_ES + xl] = s0[r];
                    slab[(1 * 32 + etile) * WN_ES + xl] = s1[r];
                }
            }
            WS_STAMP();
            __syncthreads();
        }
#undef WS_MCHUNK
#undef WS_KSTEP
#undef WS_VFIRST
#undef WS_USTEP
        return;
    }

    // ====================================== helper waves =============================================
    // The helpers issue few instructions but every one of them is on the barrier-critical path; at
    // equal or lower priority they only got an issue slot about once per MFMA (measured: 16 LDS stores
    // + selects took ~4.5k cycles), so they run at the highest priority.
    __builtin_amdgcn_s_setprio(PFNL_WS_HPRIO);
    const int ht = tid - 4 * 64;                                    // 0..255
    // unit-independent part of the staging descriptors
    int loff[WN_IN_ITERS], pyx[WN_IN_ITERS];
    unsigned stmask = 0;
#pragma unroll
    for (int i = 0; i < WN_IN_ITERS; ++i) {
        const int it = ht + i * 256;
        loff[i] = 0;
        pyx[i] = 0;
        if (it < WN_IN_ITEMS) {
            const int pix = it >> 2, c4 = it & 3;
            const int py = pix / WN_IW, px = pix - py * WN_IW;
            stmask |= 1u << i;
            loff[i] = (c4 * 4) * WP_PS + py * WP_RP + px;
            pyx[i] = (py << 16) | (px << 4) | (c4 * 4);
        }
    }
    // descriptors of the unit whose chunks are being loaded: one buffer resource per item (its range
    // check returns 0 for the voffset of halo pixels outside the image: no selects, no masks) and a
    // byte offset per staged piece; the chunk's channel offset goes in the scalar offset.
    int goff[WN_IN_ITERS];
    __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, 0, 0x00020000);
    const int item_bytes = H * W * 64 * (int)sizeof(float);
#define WS_DESC(i_)                                                                              \
    do {                                                                                         \
        int item_, y0_, x0_;                                                                     \
        WS_UNIT(i_, item_, y0_, x0_);                                                            \
        rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) + (size_t)item_ * H * W * 64, 0, item_bytes, \
                                                0x00020000);                                     \
        _Pragma("unroll") for (int k_ = 0; k_ < WN_IN_ITERS; ++k_) {                             \
            const int gy_ = y0_ + (pyx[k_] >> 16) - 1, gx_ = x0_ + ((pyx[k_] >> 4) & 0xfff) - 1; \
            const bool in_ = gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;                         \
            goff[k_] = in_ ? ((gy_ * W + gx_) * 64 + (pyx[k_] & 15)) * 4 : 0x7fffffff;           \
        }                                                                                        \
    } while (0)
    f32x4 rin0[WN_IN_ITERS], rin1[WN_IN_ITERS];
#ifdef PFNL_WS_NO_HALO  /* timing experiment only: wrong results */
#define WS_LOAD(q_, rin) do {} while (0)
#else
#define WS_LOAD(q_, rin)                                                                         \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < WN_IN_ITERS; ++k_)                               \
            rin[k_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsL, goff[k_], (q_) * WN_CK * 4, 0)); \
    } while (0)
#endif
#define WS_STORE(buf_, rin)                                                                      \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < WN_IN_ITERS; ++k_) {                             \
            if ((stmask >> k_) & 1u) {                                                           \
                float* d_ = smem + (buf_) * WP_BUF + loff[k_];                                   \
                d_[0] = rin[k_].x;                                                               \
                d_[WP_PS] = rin[k_].y;                                                           \
                d_[2 * WP_PS] = rin[k_].z;                                                       \
                d_[3 * WP_PS] = rin[k_].w;                                                       \
            }                                                                                    \
        }                                                                                        \
    } while (0)

    // epilogue state: E = the unit whose addend/residual loads are in flight / whose slab is read
    const int c4 = ht & 7;
    const int cbase = ng * 32 + c4 * 4;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + cbase);
    const float slope = p.act ? 0.2f : 1.0f;
    int e_item = 0, e_y0 = 0, e_x0 = 0;
    f32x4 av[2][2], rv[2][2];
#define WS_EPI_LOAD(k_)                                                                          \
    do {                                                                                         \
        if (FUSE) {                                                                              \
            const int id_ = (k_) * 256 + ht;                                                     \
            const int j_ = (id_ >> 3) & 1, et_ = id_ >> 4;                                       \
            const int ox_ = min(e_x0 + 2 * (et_ & 15) + j_, W - 1);                              \
            const int ai_ = e_item / p.add_div;                                                  \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_) {                                   \
                const int yc_ = min(e_y0 + 2 * (et_ >> 4) + r_, H - 1);                          \
                av[k_][r_] = *reinterpret_cast<const f32x4*>(p.addend + (((size_t)ai_ * H + yc_) * W + ox_) * 64 + cbase); \
                rv[k_][r_] = *reinterpret_cast<const f32x4*>(p.resid + (((size_t)e_item * H + yc_) * W + ox_) * 64 + cbase); \
            }                                                                                    \
        }                                                                                        \
    } while (0)
#define WS_EPI(k_)                                                                               \
    do {                                                                                         \
        const int id_ = (k_) * 256 + ht;                                                         \
        const int j_ = (id_ >> 3) & 1, et_ = id_ >> 4;                                           \
        const int oy_ = e_y0 + 2 * (et_ >> 4);                                                   \
        const int ox_ = e_x0 + 2 * (et_ & 15) + j_;                                              \
        const float* sp_ = slabm + (j_ * 32 + et_) * WN_ES + c4 * 4;                             \
        const f32x4 r0_ = *reinterpret_cast<const f32x4*>(sp_);                                  \
        const f32x4 r1_ = *reinterpret_cast<const f32x4*>(sp_ + 1 * 2 * 32 * WN_ES);             \
        const f32x4 r2_ = *reinterpret_cast<const f32x4*>(sp_ + 2 * 2 * 32 * WN_ES);             \
        const f32x4 r3_ = *reinterpret_cast<const f32x4*>(sp_ + 3 * 2 * 32 * WN_ES);             \
        f32x4 yv_[2];                                                                            \
        yv_[0] = r0_ + r1_ + r2_;                                                                \
        yv_[1] = r1_ - r2_ - r3_;                                                                \
        _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_) {                                       \
            const int y_ = oy_ + r_;                                                             \
            f32x4 o_ = yv_[r_] + bias4;                                                          \
            if (FUSE) o_ += av[k_][r_];                                                          \
            o_.x = fmaxf(o_.x, slope * o_.x);                                                    \
            o_.y = fmaxf(o_.y, slope * o_.y);                                                    \
            o_.z = fmaxf(o_.z, slope * o_.z);                                                    \
            o_.w = fmaxf(o_.w, slope * o_.w);                                                    \
            if (FUSE) o_ += rv[k_][r_];                                                          \
            if (ox_ < W && y_ < H)                                                               \
                *reinterpret_cast<f32x4*>(p.out + (((size_t)e_item * H + y_) * W + ox_) * 64 + cbase) = o_; \
        }                                                                                        \
    } while (0)

#ifdef PFNL_WS_NO_EPI   /* timing experiment only: wrong results */
#undef WS_EPI
#undef WS_EPI_LOAD
#define WS_EPI(k_) do {} while (0)
#define WS_EPI_LOAD(k_) do {} while (0)
#endif
    // prologue: chunks 0,1 of unit 0 requested; chunk 0 stored; chunk 2 requested
    WS_DESC(0);
    WS_LOAD(0, rin0);
    WS_LOAD(1, rin1);
    WS_STORE(0, rin0);
    WS_LOAD(2, rin0);
    WS_STAMP();
    __syncthreads();                                                // B0
    for (int i = 0; i < nu; ++i) {
        WS_STAMP();
        // phase 0: chunk 1 -> buffer 1; request chunk 3; first half of the previous unit's epilogue
#ifdef PFNL_WINO_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WS_STAMP();
#endif
        WS_STORE(1, rin1);
        WS_STAMP();
        WS_LOAD(3, rin1);
        WS_STAMP();
        if (i > 0) WS_EPI(0);
        WS_STAMP();
        __syncthreads();
        WS_STAMP();
        // phase 1: chunk 2 -> buffer 0; request chunk 0 of the next unit; second half of that epilogue
        WS_STORE(0, rin0);
        WS_STAMP();
        WS_DESC(min(i + 1, nu - 1));                                // past the end: harmless re-read of the last unit
        WS_LOAD(0, rin0);
        WS_STAMP();
        if (i > 0) WS_EPI(1);
        WS_STAMP();
        __syncthreads();
        WS_STAMP();
        // phase 2: chunk 3 -> buffer 1; request chunk 1 of the next unit; addend / residual of THIS unit, first half
        WS_STORE(1, rin1);
        WS_LOAD(1, rin1);
        WS_UNIT(i, e_item, e_y0, e_x0);
        WS_EPI_LOAD(0);
        WS_STAMP();
        __syncthreads();
        WS_STAMP();
        // phase 3: next unit's chunk 0 -> buffer 0; request its chunk 2; second half of the addend / residual loads
        WS_STORE(0, rin0);
        WS_LOAD(2, rin0);
        WS_EPI_LOAD(1);
        WS_STAMP();
        __syncthreads();                                            // the matrix waves have filled the slab
    }
    WS_EPI(0);
    WS_EPI(1);
#undef WS_EPI
#undef WS_EPI_LOAD
#undef WS_STORE
#undef WS_LOAD
#undef WS_DESC
#undef WS_UNIT
}

template <bool FUSE>
static hipError_t launch_ws_variant(const WinoParams& p, int nblocks, hipStream_t s) {
    static bool attr_set[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_ws_kernel<FUSE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)WS_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((conv_wino_ws_kernel<FUSE>), dim3(nblocks), dim3(WS_THREADS), WS_LDS_BYTES, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_wino_ws(const WinoParams& p, hipStream_t s) {
    if (!p.bias || !p.in || !p.upack || !p.out || p.items < 1) return hipErrorInvalidValue;
    const bool fuse = p.addend != nullptr || p.resid != nullptr;
    if (fuse && (!p.addend || !p.resid || p.add_div < 1)) return hipErrorInvalidValue;
    if ((p.H & 1) || (p.W & 1)) return hipErrorInvalidValue;
    const int ntiles = ((p.W + 2 * WN_TX - 1) / (2 * WN_TX)) * ((p.H + 2 * WN_TY - 1) / (2 * WN_TY)) * p.items;
    const int rs = (ntiles + 7) / 8;
    const int wpx = 2 * rs < WS_MAX_WG_PER_XCD ? 2 * rs : WS_MAX_WG_PER_XCD;    // even
    return fuse ? launch_ws_variant<true>(p, 8 * wpx, s) : launch_ws_variant<false>(p, 8 * wpx, s);
}

}  // namespace pfnl

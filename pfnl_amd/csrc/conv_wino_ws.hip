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
constexpr int WS_RAW_FLOATS = 2 * WN_BUF;                                  // two raw halo buffers
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

    if (wave < 4) {
        // =================================== matrix waves ===========================================
        __builtin_amdgcn_s_setprio(2);
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

        const int rowA = (xi == 0) ? 0 : 1;
        const int rowB = (xi == 3) ? 3 : 2;
        const float sgn = (xi == 1) ? 1.f : -1.f;
        const unsigned lds0 = (unsigned)(uintptr_t)smem;
        const unsigned lane_off = (unsigned)(kh * WN_PS + (2 * ty) * WN_RP + tx) * 4u;
        const unsigned pa = lds0 + lane_off + rowA * WN_RP * 4;
        const unsigned pb = lds0 + lane_off + rowB * WN_RP * 4;
        float* const slab = slabm + xi * (2 * 32 * WN_ES);

        f32x16 acc[4];
#define WS_VFIRST(q_, v_)                                                                        \
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
#define WS_KSTEP(q_, kk_, vcur_, vnxt_)                                                          \
    do {                                                                                         \
        constexpr int s_ = (q_) * (WN_CK / 2) + (kk_);                                           \
        const f32x4 bc_ = ring[s_ % WN_UDEPTH];                                                  \
        ring[s_ % WN_UDEPTH] = WS_USTEP(s_ + WN_UDEPTH);        /* wraps into the next unit */   \
        constexpr int off_ = (((q_) & 1) * WN_BUF + 2 * ((kk_) + 1) * WN_PS) * 4;                \
        if ((q_) == 0 && (kk_) == 0) {                                                           \
            wn_kstep_asm_zero<off_>(acc[0], acc[1], acc[2], acc[3], vcur_, vnxt_, bc_, pa, pb, sgn); \
        } else if ((kk_) + 1 < WN_CK / 2) {                                                      \
            wn_kstep_asm<off_>(acc[0], acc[1], acc[2], acc[3], vcur_, vnxt_, bc_, pa, pb, sgn);  \
        } else {                                                                                 \
            acc[0] = mfma32((vcur_)[0], bc_.x, acc[0]);                                          \
            acc[1] = mfma32((vcur_)[1], bc_.y, acc[1]);                                          \
            acc[2] = mfma32((vcur_)[2], bc_.z, acc[2]);                                          \
            acc[3] = mfma32((vcur_)[3], bc_.w, acc[3]);                                          \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
#define WS_MCHUNK(q_)                                                                            \
    do {                                                                                         \
        float va[4], vb[4];                                                                      \
        WS_VFIRST(q_, va);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        WS_KSTEP(q_, 0, va, vb); WS_KSTEP(q_, 1, vb, va); WS_KSTEP(q_, 2, va, vb); WS_KSTEP(q_, 3, vb, va); \
        WS_KSTEP(q_, 4, va, vb); WS_KSTEP(q_, 5, vb, va); WS_KSTEP(q_, 6, va, vb); WS_KSTEP(q_, 7, vb, va); \
    } while (0)
        static_assert(WN_CK == 16 && WN_NCHUNK == 4, "written out for 8 K-steps x 4 chunks");

        __syncthreads();                                            // B0: chunk 0 of the first unit is in LDS
        for (int i = 0; i < nu; ++i) {
            WS_MCHUNK(0);
            __syncthreads();
            WS_MCHUNK(1);
            __syncthreads();
            WS_MCHUNK(2);
            __syncthreads();
            WS_MCHUNK(3);
            // column transform over nu in registers (At = [[1,1,1,0],[0,1,-1,-1]]) -> slab[xi][j][tile][cout]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r];
                const int etile = drow(r, lane);
                slab[(0 * 32 + etile) * WN_ES + xl] = m0 + m1 + m2;
                slab[(1 * 32 + etile) * WN_ES + xl] = m1 - m2 - m3;
            }
            __syncthreads();
        }
#undef WS_MCHUNK
#undef WS_KSTEP
#undef WS_VFIRST
#undef WS_USTEP
        return;
    }

    // ====================================== helper waves =============================================
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
            loff[i] = (c4 * 4) * WN_PS + py * WN_RP + (px & 1) * WN_HALF + (px >> 1);
            pyx[i] = (py << 16) | (px << 4) | (c4 * 4);
        }
    }
    // descriptors of the unit whose chunks are being loaded
    int goff[WN_IN_ITERS];
    unsigned maskL = 0;
    const float* finL = p.in;
#define WS_DESC(i_)                                                                              \
    do {                                                                                         \
        int item_, y0_, x0_;                                                                     \
        WS_UNIT(i_, item_, y0_, x0_);                                                            \
        finL = p.in + (size_t)item_ * H * W * 64;                                                \
        maskL = 0;                                                                               \
        _Pragma("unroll") for (int k_ = 0; k_ < WN_IN_ITERS; ++k_) {                             \
            const int gy_ = y0_ + (pyx[k_] >> 16) - 1, gx_ = x0_ + ((pyx[k_] >> 4) & 0xfff) - 1; \
            const bool in_ = ((stmask >> k_) & 1u) && gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W; \
            goff[k_] = in_ ? (gy_ * W + gx_) * 64 + (pyx[k_] & 15) : 0;                          \
            maskL |= in_ ? (1u << k_) : 0u;                                                      \
        }                                                                                        \
    } while (0)
    f32x4 rin0[WN_IN_ITERS], rin1[WN_IN_ITERS];
    unsigned mask0 = 0, mask1 = 0;
#define WS_LOAD(q_, rin, mask_)                                                                  \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < WN_IN_ITERS; ++k_)                               \
            rin[k_] = *reinterpret_cast<const f32x4*>(finL + (q_) * WN_CK + goff[k_]);           \
        mask_ = maskL;                                                                           \
    } while (0)
#define WS_STORE(buf_, rin, mask_)                                                               \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < WN_IN_ITERS; ++k_) {                             \
            if ((stmask >> k_) & 1u) {                                                           \
                const f32x4 v_ = ((mask_ >> k_) & 1u) ? rin[k_] : f32x4{0.f, 0.f, 0.f, 0.f};     \
                float* d_ = smem + (buf_) * WN_BUF + loff[k_];                                   \
                d_[0] = v_.x;                                                                    \
                d_[WN_PS] = v_.y;                                                                \
                d_[2 * WN_PS] = v_.z;                                                            \
                d_[3 * WN_PS] = v_.w;                                                            \
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

    // prologue: chunks 0,1 of unit 0 requested; chunk 0 stored; chunk 2 requested
    WS_DESC(0);
    WS_LOAD(0, rin0, mask0);
    WS_LOAD(1, rin1, mask1);
    WS_STORE(0, rin0, mask0);
    WS_LOAD(2, rin0, mask0);
    __syncthreads();                                                // B0
    for (int i = 0; i < nu; ++i) {
        // phase 0: chunk 1 -> buffer 1; request chunk 3; first half of the previous unit's epilogue
        WS_STORE(1, rin1, mask1);
        WS_LOAD(3, rin1, mask1);
        if (i > 0) WS_EPI(0);
        __syncthreads();
        // phase 1: chunk 2 -> buffer 0; request chunk 0 of the next unit; second half of that epilogue
        WS_STORE(0, rin0, mask0);
        WS_DESC(min(i + 1, nu - 1));                                // past the end: harmless re-read of the last unit
        WS_LOAD(0, rin0, mask0);
        if (i > 0) WS_EPI(1);
        __syncthreads();
        // phase 2: chunk 3 -> buffer 1; request chunk 1 of the next unit; addend / residual of THIS unit, first half
        WS_STORE(1, rin1, mask1);
        WS_LOAD(1, rin1, mask1);
        WS_UNIT(i, e_item, e_y0, e_x0);
        WS_EPI_LOAD(0);
        __syncthreads();
        // phase 3: next unit's chunk 0 -> buffer 0; request its chunk 2; second half of the addend / residual loads
        WS_STORE(0, rin0, mask0);
        WS_LOAD(2, rin0, mask0);
        WS_EPI_LOAD(1);
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

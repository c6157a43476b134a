// Fused Winograd F(2x2,3x3) convolution, 64 -> 64 channels, f32 MFMA - persistent, wave-specialised.
//
// Same arithmetic and packed U as conv_wino_kernel (conv_wino.hip; reference model/pfnl.py:49-51 applied
// at :66-71).  What shapes this kernel is a property of the gfx950 SIMD measured in tools/ubench/:
// while the matrix pipe is busy with f32 MFMAs, every VALU / LDS instruction a wave issues between its
// own MFMAs delays the next MFMA (~14 cycles per group + ~4 per instruction), and other waves on that
// SIMD get about ONE VALU issue slot per MFMA.  Non-MFMA instruction COUNT per MFMA is therefore the
// currency, not latency.  So:
//
//   * one workgroup (8 waves) per CU, roles fixed for the life of the kernel;
//   * waves 0-3, matrix waves (Winograd row xi = wave): 8 accumulators = BOTH 32-channel N-tiles of a
//     2x16-tile spatial tile, so the input transform (4 ds_read_b64 + 4 packed-f32 VALU per K-step) is
//     paid once per 8 MFMAs.  Their only vector-memory traffic is the L2-resident U stream (4 K-steps
//     in flight), so no HBM-latency load ever sits in front of it in the in-order vmcnt queue.  Every
//     K-step is the same asm block: it multiplies step s and prefetches + transforms step s+1; the
//     chunk barrier sits between K-step 6 and 7 of a chunk (by then the last raw values of the chunk
//     are in registers and the next chunk's tile is in LDS), so there is no per-chunk restart bubble;
//   * waves 4-7, helper waves: stage the raw halo chunks global -> registers -> LDS FOUR phases ahead
//     (buffer loads whose range check zero-fills the image border: no selects), and run the previous
//     tile's row transform + fused bias / addend / leaky-relu / residual epilogue out of an LDS slab,
//     with buffer stores whose range check drops the out-of-image lanes (no branches).
//
// Phase c = [barrier c-1, barrier c): the matrix waves read raw buffer c&1 (chunk c) while the helpers
// fill buffer (c+1)&1; a tile's accumulators are dropped into the slab right after its 32nd K-step and
// are read by the helpers in phases 1..3 of the next tile.  100.6 KB LDS, <= 256 VGPRs.
#include <cstdint>

#include "common.h"
#include "wino_geom.h"

namespace pfnl {

constexpr int WS_THREADS = 512;
constexpr int WS_RAW_FLOATS = 2 * WP_BUF;                                  // two raw halo buffers
constexpr int WS_ES = 68;                                                  // slab row stride: 64 couts + 4
constexpr int WS_SLAB_XI = 2 * 32 * WS_ES;                                 // floats per Winograd row: [j][tile][cout]
constexpr int WS_LDS_FLOATS = WS_RAW_FLOATS + 4 * WS_SLAB_XI;
constexpr size_t WS_LDS_BYTES = size_t(WS_LDS_FLOATS) * sizeof(float);     // 100 608 B
constexpr int WS_MAX_WG_PER_XCD = 32;                                      // one workgroup per CU
constexpr int WS_UD_MAX = 8;                                               // K-steps of U in flight (divides 32)
static_assert((WS_RAW_FLOATS * sizeof(float)) % 16 == 0, "slab must stay 16-byte aligned");

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: out = act(conv + bias).   MODE 1: out = act(conv + bias + addend[item / add_div]) + resid.
// MODE 2 (the whole of conv2_i, reference model/pfnl.py:69-71): per clip and spatial tile one workgroup runs
//   unit 0      : raw 3x3 conv of `in2` (= base, kernel rows 0..63 = upack2) -> kept in LDS (`pbl`)
//   units 1..T  : 3x3 conv of frame t of `in` (kernel rows 64..127 = upack) + pbl + bias, act, + resid -> out
// so the shared half never goes to HBM and needs no launch of its own (T = add_div).
// MODE 3 (convmerge1, model/pfnl.py:73-74: a 3x3 conv over the concat of T frames): per clip and spatial tile
//   units 0..T-1 : frame t of `in` with the weight pack upack + t*upack_stride, all into the SAME accumulators;
//   only the last unit drops the slab and gets an epilogue: out[clip] = act(sum + bias).
template <int MODE>
__global__ __launch_bounds__(WS_THREADS, 1) void conv_wino_ws_kernel(WinoParams p) {
    constexpr bool FUSE = MODE == 1 || MODE == 2, GROUPED = MODE == 2, ACCUM = MODE == 3;
    constexpr int WS_UD = ACCUM ? 4 : WS_UD_MAX;                    // (the accumulating mode is short of registers)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const slabm = smem + WS_RAW_FLOATS;
    float* const pbl = slabm + 4 * WS_SLAB_XI;                      // MODE 2: [4 rows][32 cols][64] addend tile
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;

    // ---- this workgroup's tile list: XCD b&7 owns tiles [xcd*rs, xcd*rs+rs); workgroup j of the XCD
    // takes tiles j, j+wpx, ...  (placement only affects L2 reuse, never results)
    const int tiles_x = (W + 2 * WN_TX - 1) / (2 * WN_TX);
    const int tiles_y = (H + 2 * WN_TY - 1) / (2 * WN_TY);
    const int per_item = tiles_x * tiles_y;
    const int gT = (GROUPED || ACCUM) ? p.add_div : 0;              // frames per clip
    const int upg = GROUPED ? gT + 1 : (ACCUM ? gT : 1);            // units per (clip, tile) group
    const int ntiles = (GROUPED || ACCUM) ? per_item * (p.items / p.add_div) : per_item * p.items;   // MODE 2/3: (clip, tile) groups
    const int rs = (ntiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int wpx = gridDim.x >> 3;
    const int tbeg = xcd * rs;
    const int tcnt = min(rs, ntiles - tbeg);
    if (j >= tcnt) return;
    const int nu = ((tcnt - j + wpx - 1) / wpx) * upg;
    // Tile order.  Plain launches: item-major.  Fused launches (conv2): the add_div frames of a clip at the
    // same spatial tile are consecutive, so the shared addend tile (`pb`, one per clip) is read from HBM once
    // and then from L2 (measured before: 359 MB fetched per launch against 252 MB compulsory).
    const int grp = (MODE == 1 && p.add_div > 1 && p.items % p.add_div == 0) ? p.add_div : 1;
    const int per_grp = per_item * grp;
    // unit i of this workgroup -> (input item, tile origin, is it the shared-half unit of MODE 2)
#define WS_UNIT(i_, item_, y0_, x0_, pb_, fin_)                       \
    do {                                                              \
        int sp_;                                                      \
        fin_ = true;                                                  \
        if (GROUPED || ACCUM) {                                       \
            const int gi_ = (i_) / upg;                               \
            const int u_ = (i_) - gi_ * upg;                          \
            const int t_ = tbeg + j + gi_ * wpx;                      \
            const int c_ = t_ / per_item;                             \
            sp_ = t_ - c_ * per_item;                                 \
            pb_ = GROUPED && u_ == 0;                                 \
            item_ = GROUPED ? (pb_ ? c_ : c_ * gT + u_ - 1) : c_ * gT + u_; \
            if (ACCUM) fin_ = u_ == gT - 1;                           \
        } else {                                                      \
            const int t_ = tbeg + j + (i_) * wpx;                     \
            const int c_ = t_ / per_grp;                              \
            const int r_ = t_ - c_ * per_grp;                         \
            sp_ = r_ / grp;                                           \
            item_ = c_ * grp + (r_ - sp_ * grp);                      \
            pb_ = false;                                              \
        }                                                             \
        const int by_ = sp_ / tiles_x;                                \
        y0_ = by_ * (2 * WN_TY);                                      \
        x0_ = (sp_ - by_ * tiles_x) * (2 * WN_TX);                    \
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
#ifdef PFNL_WS_MPRIO
        __builtin_amdgcn_s_setprio(PFNL_WS_MPRIO);
#endif
        const int xi = wave;
        const int tx = lane & 15;
        const int ty = (lane >> 4) & 1;
        const int kh = lane >> 5;
        const int xl = lane & 31;

        constexpr int KS_F4 = 64;                                   // float4 per (chunk, xi, N-tile, kk)
        constexpr int UP_CHUNK_F4 = 4 * 2 * 8 * KS_F4;
        // one descriptor over both weight packs of MODE 2 (they live in one blob); the pack is chosen per
        // unit through the scalar offset
        const float* ubase = (GROUPED && p.upack2 < p.upack) ? p.upack2 : p.upack;
        const int uoffB = (int)((p.upack - ubase) * sizeof(float));
        const int uoffA = GROUPED ? (int)((p.upack2 - ubase) * sizeof(float)) : uoffB;
        const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(ubase), 0,
            max(uoffA, uoffB) + (ACCUM ? (gT - 1) * (int)(p.upack_stride * sizeof(float)) : 0) + (int)(wino_pack_floats_c * sizeof(float)),
            0x00020000);
        int mu = 0;                                                 // MODE 2: position of the current unit in its group
        int uoff_cur = GROUPED ? uoffA : uoffB, uoff_nxt = uoffB;
        const int ustride = ACCUM ? (int)(p.upack_stride * sizeof(float)) : 0;
        if (ACCUM) uoff_nxt = gT > 1 ? ustride : 0;
        const int uvoff = ((xi * 2 * 8) * KS_F4 + lane) * 16;
#define WS_USTEP(s_, g_)                                                                          \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                              \
                                  urs, uvoff,                                                     \
                                  ((s_) < 32 ? uoff_cur : uoff_nxt) +                             \
                                      ((((s_) & 31) >> 3) * UP_CHUNK_F4 + ((g_) * 8 + ((s_) & 7)) * KS_F4) * 16, \
                                  0))
        f32x4 ring0[WS_UD], ring1[WS_UD];
#pragma unroll
        for (int d = 0; d < WS_UD; ++d) {
            ring0[d] = WS_USTEP(d, 0);
            ring1[d] = WS_USTEP(d, 1);
        }

        const int rowA = (xi == 0) ? 0 : 1;
        const int rowB = (xi == 3) ? 3 : 2;
        const float sgn = (xi == 1) ? 1.f : -1.f;
        const f32x2 sg2 = {sgn, sgn};
        const unsigned lds0 = (unsigned)(uintptr_t)smem;
        const int lane_a = kh * WP_PS + (2 * ty + rowA) * WP_RP + 2 * tx;          // floats
        const int lane_b = kh * WP_PS + (2 * ty + rowB) * WP_RP + 2 * tx;
        const unsigned pa = lds0 + (unsigned)lane_a * 4u;
        const unsigned pb = lds0 + (unsigned)lane_b * 4u;
        float* const slab = slabm + xi * WS_SLAB_XI;

        f32x16 acc[8];                                              // [N-tile g][nu]
        f32x2 vA, vB, vC, vD;                                       // (V0,V3), (V1,V2): current / next step

        // step s: 8 MFMAs on V_s; raw values of step s+1 read from buffer ((s+1)>>3)&1 and transformed
#define WS_STEPX(s_, z_, c03_, c12_, n03_, n12_)                                                 \
    do {                                                                                         \
        constexpr int d_ = (s_) % WS_UD;                                                         \
        const f32x4 b0_ = ring0[d_], b1_ = ring1[d_];                                            \
        ring0[d_] = WS_USTEP((s_) + WS_UD, 0);                  /* wraps into the next tile */   \
        ring1[d_] = WS_USTEP((s_) + WS_UD, 1);                                                   \
        constexpr int n_ = ((s_) + 1) & 31;                                                      \
        constexpr int off_ = (((n_ >> 3) & 1) * WP_BUF + 2 * (n_ & 7) * WP_PS) * 4;              \
        f32x2 x01_, y01_, x23_, y23_;                                                            \
        wq_kstep_a<off_, z_>(acc[0], acc[1], acc[2], acc[3], (c03_).x, (c12_).x, (c12_).y, (c03_).y, b0_, \
                                    x01_, y01_, x23_, y23_, pa, pb);                             \
        wq_kstep_b<z_>(acc[4], acc[5], acc[6], acc[7], (c03_).x, (c12_).x, (c12_).y, (c03_).y, b1_, x01_, \
                              y01_, x23_, y23_, n03_, n12_, sg2);                                \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
#define WS_STEP(s_, c03_, c12_, n03_, n12_) WS_STEPX(s_, false, c03_, c12_, n03_, n12_)
#define WS_STEP2(s_)                       \
    WS_STEP(s_, vA, vB, vC, vD);           \
    WS_STEP((s_) + 1, vC, vD, vA, vB)
        // column transform over nu (At = [[1,1,1,0],[0,1,-1,-1]]) of registers 4q..4q+3 (= channels 32g + 8q + 4kh + 0..3
        // of tile xl) of N-tile g -> slab[xi][j][tile][cout], 16 bytes per store
#define WS_DUMP(g_, q_)                                                                          \
    do {                                                                                         \
        const f32x4 m0_ = {acc[4 * (g_)][4 * (q_)], acc[4 * (g_)][4 * (q_) + 1], acc[4 * (g_)][4 * (q_) + 2], acc[4 * (g_)][4 * (q_) + 3]}; \
        const f32x4 m1_ = {acc[4 * (g_) + 1][4 * (q_)], acc[4 * (g_) + 1][4 * (q_) + 1], acc[4 * (g_) + 1][4 * (q_) + 2], acc[4 * (g_) + 1][4 * (q_) + 3]}; \
        const f32x4 m2_ = {acc[4 * (g_) + 2][4 * (q_)], acc[4 * (g_) + 2][4 * (q_) + 1], acc[4 * (g_) + 2][4 * (q_) + 2], acc[4 * (g_) + 2][4 * (q_) + 3]}; \
        const f32x4 m3_ = {acc[4 * (g_) + 3][4 * (q_)], acc[4 * (g_) + 3][4 * (q_) + 1], acc[4 * (g_) + 3][4 * (q_) + 2], acc[4 * (g_) + 3][4 * (q_) + 3]}; \
        float* const d_ = slab + xl * WS_ES + (g_) * 32 + 8 * (q_) + 4 * kh;                      \
        *reinterpret_cast<f32x4*>(d_) = m0_ + m1_ + m2_;                                          \
        *reinterpret_cast<f32x4*>(d_ + 32 * WS_ES) = m1_ - (m2_ + m3_);                           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
        // K-step 31 of a tile: block B (N-tile 1) split into its MFMAs, N-tile 0's finished accumulators transformed
        // and stored between them
#define WS_STEP31_D0(c03_, c12_, n03_, n12_)                                                     \
    do {                                                                                         \
        constexpr int d_ = 31 % WS_UD;                                                           \
        const f32x4 b0_ = ring0[d_], b1_ = ring1[d_];                                            \
        ring0[d_] = WS_USTEP(31 + WS_UD, 0);                                                     \
        ring1[d_] = WS_USTEP(31 + WS_UD, 1);                                                     \
        f32x2 x01_, y01_, x23_, y23_;                                                            \
        wq_kstep_a<0, false>(acc[0], acc[1], acc[2], acc[3], (c03_).x, (c12_).x, (c12_).y, (c03_).y, b0_, \
                             x01_, y01_, x23_, y23_, pa, pb);       /* next step = 0 of the next tile: buffer 0, kk 0 */ \
        wq_mfma1<false>(acc[4], b1_.x, (c03_).x);                                                \
        wq_results_ready(acc[0], acc[1], acc[2], acc[3]);                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        WS_DUMP(0, 0);                                                                           \
        wq_mfma1<false>(acc[5], b1_.y, (c12_).x);                                                \
        wq_b_transform(x01_, y01_, x23_, y23_, n03_, n12_, sg2);                                  \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        WS_DUMP(0, 1);                                                                           \
        wq_mfma1<false>(acc[6], b1_.z, (c12_).y);                                                \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        WS_DUMP(0, 2);                                                                           \
        wq_mfma1<false>(acc[7], b1_.w, (c03_).y);                                                \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        WS_DUMP(0, 3);                                                                           \
    } while (0)
        // K-step 0 of the NEXT tile (C = 0): block A (N-tile 0) split, the previous tile's N-tile 1 transformed and
        // stored between its MFMAs; then block B as usual
#define WS_STEP0_D1(c03_, c12_, n03_, n12_)                                                      \
    do {                                                                                         \
        constexpr int d_ = 0;                                                                    \
        const f32x4 b0_ = ring0[d_], b1_ = ring1[d_];                                            \
        ring0[d_] = WS_USTEP(WS_UD, 0);                                                          \
        ring1[d_] = WS_USTEP(WS_UD, 1);                                                          \
        constexpr int off_ = (2 * 1 * WP_PS) * 4;                   /* next step = 1: buffer 0, kk 1 */ \
        f32x2 x01_, y01_, x23_, y23_;                                                            \
        wq_a_head<off_, true>(acc[0], b0_.x, (c03_).x, x01_, y01_, x23_, y23_, pa, pb);          \
        wq_results_ready(acc[4], acc[5], acc[6], acc[7]);                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        WS_DUMP(1, 0);                                                                           \
        wq_mfma1<true>(acc[1], b0_.y, (c12_).x);                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        WS_DUMP(1, 1);                                                                           \
        wq_mfma1<true>(acc[2], b0_.z, (c12_).y);                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        WS_DUMP(1, 2);                                                                           \
        wq_mfma1<true>(acc[3], b0_.w, (c03_).y);                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        WS_DUMP(1, 3);                                                                           \
        wq_kstep_b<true>(acc[4], acc[5], acc[6], acc[7], (c03_).x, (c12_).x, (c12_).y, (c03_).y, b1_, x01_, \
                         y01_, x23_, y23_, n03_, n12_, sg2);                                     \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)

        __syncthreads();                                            // chunk 0 of the first tile is in LDS
        {   // V of the very first K-step (every later one comes out of the asm step before it)
            const f32x2* cA = reinterpret_cast<const f32x2*>(smem + lane_a);
            const f32x2* cB = reinterpret_cast<const f32x2*>(smem + lane_b);
            const f32x2 t01 = cA[0] + sg2 * cB[0];
            const f32x2 t23 = cA[1] + sg2 * cB[1];
            vA = t01 - t23;
            vB.x = t01.y + t23.x;
            vB.y = t23.x - t01.y;
            asm volatile("s_nop 4" ::: "memory");                   // VALU write -> MFMA read inside the asm block
        }
        for (int i = 0; i < nu; ++i) {
            if (ACCUM) {                                            // MODE 3: frames add into the same accumulators;
                if (mu == 0) {                                      // cleared in place before a tile's first frame
#pragma unroll
                    for (int n = 0; n < 8; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
                }
                WS_STEPX(0, false, vA, vB, vC, vD);
            } else if (i > 0) {                                     // first K-step of a tile: C = 0; the previous tile's N-tile 1 leaves
                WS_STEP0_D1(vA, vB, vC, vD);
            } else {
                WS_STEPX(0, true, vA, vB, vC, vD);
            }
            WS_STEP(1, vC, vD, vA, vB);
            WS_STEP2(2);
            WS_STEP2(4);
            WS_STEP(6, vA, vB, vC, vD);
            WS_STAMP();
            __syncthreads();                                        // chunk 1 ready / chunk 0's buffer free
            WS_STAMP();
            WS_STEP(7, vC, vD, vA, vB);
            WS_STEP2(8);
            WS_STEP2(10);
            WS_STEP2(12);
            WS_STEP(14, vA, vB, vC, vD);
            WS_STAMP();
            __syncthreads();
            WS_STAMP();
            WS_STEP(15, vC, vD, vA, vB);
            WS_STEP2(16);
            WS_STEP2(18);
            WS_STEP2(20);
            WS_STEP(22, vA, vB, vC, vD);
            WS_STAMP();
            __syncthreads();
            WS_STAMP();
            WS_STEP(23, vC, vD, vA, vB);
            WS_STEP2(24);
            WS_STEP2(26);
            WS_STEP2(28);
            WS_STEP(30, vA, vB, vC, vD);
            WS_STAMP();
            __syncthreads();                                        // next tile's chunk 0 ready
            WS_STAMP();
            if (ACCUM) {
                WS_STEP(31, vC, vD, vA, vB);
                WS_STAMP();
                // MODE 3: only a clip's last frame leaves the accumulators; both N-tiles after the last MFMA
                asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // MFMA results of the asm blocks -> VALU reads
                if (mu == gT - 1) {
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int q = 0; q < 4; ++q) WS_DUMP(g, q);
                }
            } else {
                WS_STEP31_D0(vC, vD, vA, vB);                       // N-tile 0 leaves here, N-tile 1 in the next tile's K-step 0
                WS_STAMP();
            }
            WS_STAMP();
            if (GROUPED) {                                          // weight pack of the next unit / the one after
                mu = mu == gT ? 0 : mu + 1;
                uoff_cur = uoff_nxt;
                uoff_nxt = mu == gT ? uoffA : uoffB;
            }
            if (ACCUM) {
                mu = mu == gT - 1 ? 0 : mu + 1;
                uoff_cur = uoff_nxt;
                uoff_nxt = (mu == gT - 1 ? 0 : mu + 1) * ustride;
            }
        }
        if (!ACCUM) {                                               // the last tile's N-tile 1
            wq_results_ready(acc[4], acc[5], acc[6], acc[7]);
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) WS_DUMP(1, q);
        }
        __syncthreads();                                            // the last tile's slab is complete
#undef WS_STEP0_D1
#undef WS_STEP31_D0
#undef WS_DUMP
#undef WS_STEP2
#undef WS_STEP
#undef WS_USTEP
        return;
    }

    // ====================================== helper waves =============================================
    // Few instructions, but every one is on the barrier-critical path and only gets an issue slot when the
    // matrix wave of its SIMD leaves one: highest priority.
#ifndef PFNL_WS_HPRIO
#define PFNL_WS_HPRIO 3
#endif
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
    int relk[WN_IN_ITERS];                                          // byte offset relative to the tile origin pixel
#pragma unroll
    for (int i = 0; i < WN_IN_ITERS; ++i)
        relk[i] = ((((pyx[i] >> 16) - 1) * W + ((pyx[i] >> 4) & 0xfff) - 1) * 64 + (pyx[i] & 15)) * 4;
    // descriptors of the tile whose chunks are being loaded: one buffer resource per item (its range
    // check returns 0 for the voffset given to halo pixels outside the image) and a byte offset per
    // staged piece; the chunk's channel offset goes in the scalar offset.
    const int item_bytes = H * W * 64 * (int)sizeof(float);
    const size_t in_stride = p.in_item_stride ? (size_t)p.in_item_stride : (size_t)H * W * 64;
    int goff[WN_IN_ITERS], goffN[WN_IN_ITERS];                     // current / next (computed a phase early)
    __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, item_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsLN = rsL;
#define WS_DESC(i_)                                                                              \
    do {                                                                                         \
        int item_, y0_, x0_;                                                                     \
        bool pb_, fin_;                                                                          \
        WS_UNIT(i_, item_, y0_, x0_, pb_, fin_);                                                 \
        (void)fin_;                                                                              \
        rsLN = __builtin_amdgcn_make_buffer_rsrc(                                                \
            const_cast<float*>(pb_ ? p.in2 : p.in) + (size_t)item_ * in_stride, 0, item_bytes, 0x00020000); \
        const int org_ = (y0_ * W + x0_) * 256;                                                  \
        if (y0_ > 0 && y0_ + WN_IH - 1 <= H && x0_ > 0 && x0_ + WN_IW - 1 <= W) {   /* interior tile: uniform branch */ \
            _Pragma("unroll") for (int k_ = 0; k_ < WN_IN_ITERS; ++k_) goffN[k_] = org_ + relk[k_]; \
        } else {                                                                                 \
            _Pragma("unroll") for (int k_ = 0; k_ < WN_IN_ITERS; ++k_) {                         \
                const int gy_ = y0_ + (pyx[k_] >> 16) - 1, gx_ = x0_ + ((pyx[k_] >> 4) & 0xfff) - 1; \
                const bool in_ = (unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W;     \
                goffN[k_] = in_ ? org_ + relk[k_] : 0x7fffffff;                                   \
            }                                                                                    \
        }                                                                                        \
    } while (0)
#define WS_DESC_COMMIT()                                                                         \
    do {                                                                                         \
        rsL = rsLN;                                                                              \
        _Pragma("unroll") for (int k_ = 0; k_ < WN_IN_ITERS; ++k_) goff[k_] = goffN[k_];         \
    } while (0)
    f32x4 rin[WN_NCHUNK][WN_IN_ITERS];                              // one register set per chunk index
#define WS_LOAD(q_)                                                                              \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < WN_IN_ITERS; ++k_)                               \
            rin[q_][k_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsL, goff[k_], (q_) * WN_CK * 4, 0)); \
    } while (0)
#define WS_STORE(q_)                                                                             \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < WN_IN_ITERS; ++k_) {                             \
            if ((stmask >> k_) & 1u) {                                                           \
                float* d_ = smem + ((q_) & 1) * WP_BUF + loff[k_];                               \
                d_[0] = rin[q_][k_].x;                                                           \
                d_[WP_PS] = rin[q_][k_].y;                                                       \
                d_[2 * WP_PS] = rin[q_][k_].z;                                                   \
                d_[3 * WP_PS] = rin[q_][k_].w;                                                   \
            }                                                                                    \
        }                                                                                        \
    } while (0)

    // epilogue state: E = the tile whose addend / residual loads are in flight and whose slab is read.
    // Item id = k*256 + ht -> channel quad c4 (16), column parity jj, Winograd tile et (32); two output rows.
    const int c4 = ht & 15;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + c4 * 4);
    const float slope = p.act ? 0.2f : 1.0f;
    int n_y0 = 0, n_x0 = 0;
    bool e_pb = false, n_pb = false;                                // MODE 2: the epilogue tile is the shared-half unit
    bool e_fin = true, n_fin = true;                                // MODE 3: the unit completes its tile (has an epilogue)
    __amdgpu_buffer_rsrc_t rsOut = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, item_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsRes = rsOut, rsAdd = rsOut, rsOutN = rsOut, rsResN = rsOut, rsAddN = rsOut;
    f32x4 av[4][2], rv[4][2];
#define WS_EPI_UNIT(i_)                                                                          \
    do {                                                                                         \
        int item_;                                                                               \
        WS_UNIT(i_, item_, n_y0, n_x0, n_pb, n_fin);                                             \
        if (n_pb) item_ = 0;                                    /* its result stays in LDS */     \
        if (ACCUM) item_ /= gT;                                 /* one output tile per clip */    \
        rsOutN = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)item_ * H * W * 64, 0, item_bytes, 0x00020000); \
        if (FUSE) {                                                                              \
            rsResN = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid) + (size_t)item_ * H * W * 64, 0, \
                                                      item_bytes, 0x00020000);                   \
            if (!GROUPED)                                                                        \
                rsAddN = __builtin_amdgcn_make_buffer_rsrc(                                      \
                    const_cast<float*>(p.addend) + (size_t)(item_ / p.add_div) * H * W * 64, 0, item_bytes, 0x00020000); \
        }                                                                                        \
    } while (0)
#define WS_EPI_COMMIT()                                                                          \
    do {                                                                                         \
        rsOut = rsOutN;                                                                          \
        e_pb = n_pb;                                                                             \
        e_fin = n_fin;                                                                           \
        if (FUSE) {                                                                              \
            rsRes = rsResN;                                                                      \
            if (!GROUPED) rsAdd = rsAddN;                                                        \
        }                                                                                        \
        _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) eoff[k_] = eoffN[k_];                   \
    } while (0)
    // Byte offsets of this thread's four items (row 0) inside the item, computed once per tile; out of
    // range for pixels outside the image (H, W even: row 1 is valid iff row 0 is; it is reached through the
    // scalar offset `rowb` - the range check covers voffset + soffset, and 0x7fffffff + rowb stays out of range).  Item k sits 16*(k&1) columns and
    // 2*(k>>1) rows from item 0.
    int eoff[4] = {0, 0, 0, 0}, eoffN[4] = {0, 0, 0, 0};
    const int rowb = W * 256;
    const int h5 = ht >> 5;
#define WS_EPI_OFFS()                                                                            \
    do {                                                                                         \
        const int ox_ = n_x0 + 2 * h5 + ((ht >> 4) & 1), oy_ = n_y0;                             \
        const int o0_ = ((oy_ * W + ox_) * 64 + c4 * 4) * 4;                                     \
        const bool vx0_ = ox_ < W, vx1_ = ox_ + 16 < W, vy0_ = oy_ < H, vy1_ = oy_ + 2 < H;      \
        eoffN[0] = (vx0_ && vy0_) ? o0_ : 0x7fffffff;                                             \
        eoffN[1] = (vx1_ && vy0_) ? o0_ + 16 * 256 : 0x7fffffff;                                  \
        eoffN[2] = (vx0_ && vy1_) ? o0_ + 2 * rowb : 0x7fffffff;                                  \
        eoffN[3] = (vx1_ && vy1_) ? o0_ + 2 * rowb + 16 * 256 : 0x7fffffff;                       \
    } while (0)
#define WS_EPI_LOAD(k_)                                                                          \
    do {                                                                                         \
        if (FUSE && !(GROUPED && e_pb)) {                                                        \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_) {                                   \
                if (!GROUPED)                                                                    \
                    av[k_][r_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsAdd, eoff[k_], r_ * rowb, 0)); \
                rv[k_][r_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsRes, eoff[k_], r_ * rowb, 0)); \
            }                                                                                    \
        }                                                                                        \
    } while (0)
#define WS_EPI(k_)                                                                               \
    do {                                                                                         \
        const int id_ = (k_) * 256 + ht;                                                         \
        const float* sp_ = slabm + (((id_ >> 4) & 1) * 32 + (id_ >> 5)) * WS_ES + c4 * 4;        \
        const f32x4 r0_ = *reinterpret_cast<const f32x4*>(sp_);                                  \
        const f32x4 r1_ = *reinterpret_cast<const f32x4*>(sp_ + 1 * WS_SLAB_XI);                 \
        const f32x4 r2_ = *reinterpret_cast<const f32x4*>(sp_ + 2 * WS_SLAB_XI);                 \
        const f32x4 r3_ = *reinterpret_cast<const f32x4*>(sp_ + 3 * WS_SLAB_XI);                 \
        f32x4 yv_[2];                                                                            \
        yv_[0] = r0_ + r1_ + r2_;                               /* row transform over xi */      \
        yv_[1] = r1_ - (r2_ + r3_);                                                              \
        /* MODE 2: this item's two pixels in the LDS addend tile [row][col][64] */               \
        f32x4* pl_ = reinterpret_cast<f32x4*>(pbl + ((2 * ((id_ >> 5) >> 4)) * 32 + 2 * ((id_ >> 5) & 15) + ((id_ >> 4) & 1)) * 64 + c4 * 4); \
        if (GROUPED && e_pb) {                                  /* shared-half unit: raw result -> LDS, nothing else */ \
            pl_[0] = yv_[0];                                                                     \
            pl_[32 * 16] = yv_[1];                                                               \
        } else                                                                                   \
        _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_) {                                       \
            f32x4 o_ = yv_[r_] + bias4;                                                          \
            if (GROUPED) o_ += pl_[r_ * 32 * 16];                                                \
            else if (FUSE) o_ += av[k_][r_];                                                     \
            const f32x4 so_ = o_ * slope;                                                        \
            o_.x = fmaxf(o_.x, so_.x);                                                           \
            o_.y = fmaxf(o_.y, so_.y);                                                           \
            o_.z = fmaxf(o_.z, so_.z);                                                           \
            o_.w = fmaxf(o_.w, so_.w);                                                           \
            if (FUSE) o_ += rv[k_][r_];                                                          \
            buffer_store_b128_guarded(__builtin_bit_cast(u32x4, o_), rsOut, eoff[k_], r_ * rowb);        \
        }                                                                                        \
    } while (0)

    // prologue: all four chunks of tile 0 requested; chunk 0 stored; chunk 0 of tile 1 requested
    WS_DESC(0);
    WS_DESC_COMMIT();
    WS_LOAD(0);
    WS_LOAD(1);
    WS_LOAD(2);
    WS_LOAD(3);
    WS_STORE(0);
    WS_DESC(min(1, nu - 1));                                        // past the end: harmless re-read of the last tile
    WS_DESC_COMMIT();
    WS_LOAD(0);
    __syncthreads();
    for (int i = 0; i < nu; ++i) {
        // phase 0 (the matrix waves also drop the previous tile's accumulators here, so it has slack):
        // chunk 1 -> buffer 1; request chunk 1 of the next tile; rest of the previous tile's addend /
        // residual; descriptors of the tile after next and this tile's epilogue offsets, into shadow sets
        WS_STORE(1);
        WS_LOAD(1);
        if (i > 0 && e_fin) {
            WS_EPI_LOAD(2);
            WS_EPI_LOAD(3);
        }
        WS_DESC(min(i + 2, nu - 1));
        WS_EPI_UNIT(i);
        WS_EPI_OFFS();
        WS_STAMP();
        __syncthreads();                                            // the matrix waves have filled the slab (tile i-1)
        WS_STAMP();
        // phase 1: chunk 2 -> buffer 0; half of the previous tile's epilogue
        WS_STORE(2);
        WS_LOAD(2);
        if (i > 0 && e_fin) {
            WS_EPI(0);
            WS_EPI(1);
        }
        WS_STAMP();
        __syncthreads();
        WS_STAMP();
        // phase 2: chunk 3 -> buffer 1
        WS_STORE(3);
        WS_LOAD(3);
        if (i > 0 && e_fin) WS_EPI(2);
        WS_STAMP();
        __syncthreads();
        WS_STAMP();
        // phase 3: next tile's chunk 0 -> buffer 0; request chunk 0 of the tile after; last quarter of the
        // epilogue, then the addend / residual of THIS tile (first half)
        WS_STORE(0);
        WS_DESC_COMMIT();
        WS_LOAD(0);
        if (i > 0 && e_fin) WS_EPI(3);
        WS_EPI_COMMIT();
        WS_EPI_LOAD(0);
        WS_EPI_LOAD(1);
        WS_STAMP();
        __syncthreads();
        WS_STAMP();
    }
    WS_EPI_LOAD(2);
    WS_EPI_LOAD(3);
    __syncthreads();                                                // the last tile's slab is complete
    WS_EPI(0);
    WS_EPI(1);
    WS_EPI(2);
    WS_EPI(3);
#undef WS_EPI_COMMIT
#undef WS_DESC_COMMIT
#undef WS_EPI
#undef WS_EPI_LOAD
#undef WS_EPI_OFFS
#undef WS_EPI_UNIT
#undef WS_STORE
#undef WS_LOAD
#undef WS_DESC
#undef WS_UNIT
}

template <int MODE>
static hipError_t launch_ws_variant(const WinoParams& p, int nblocks, hipStream_t s) {
    constexpr size_t WS_LDS = WS_LDS_BYTES + (MODE == 2 ? 4 * 32 * 64 * sizeof(float) : 0);
    static std::atomic<int> attr_set[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_ws_kernel<MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)WS_LDS);
        if (e != hipSuccess) return e;
        attr_set[dev] = 1;
    }
    hipLaunchKernelGGL((conv_wino_ws_kernel<MODE>), dim3(nblocks), dim3(WS_THREADS), WS_LDS, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_wino_ws(const WinoParams& p, hipStream_t s) {
    if (!p.bias || !p.in || !p.upack || !p.out || p.items < 1) return hipErrorInvalidValue;
    const bool fuse = p.addend != nullptr || p.resid != nullptr;
    if (fuse && !p.in2 && (!p.addend || !p.resid || p.add_div < 1)) return hipErrorInvalidValue;
    if ((p.H & 1) || (p.W & 1)) return hipErrorInvalidValue;
    if ((long long)p.H * p.W * 256 >= 0x7fffffffLL && !p.in_item_stride) return launch_conv_wino(p, s);   // 32-bit buffer offsets per item: per-tile kernel instead
    const bool grouped = p.in2 != nullptr;
    const bool accum = p.accum != 0;
    if (accum && (grouped || fuse || p.add_div < 1 || p.items % p.add_div || p.upack_stride < 0)) return hipErrorInvalidValue;
    if (grouped && (!p.upack2 || !p.resid || p.addend || p.add_div < 1 || p.items % p.add_div)) return hipErrorInvalidValue;
    const int per_item = ((p.W + 2 * WN_TX - 1) / (2 * WN_TX)) * ((p.H + 2 * WN_TY - 1) / (2 * WN_TY));
    const int ntiles = per_item * ((grouped || accum) ? p.items / p.add_div : p.items);   // MODE 2/3: (clip, tile) groups
    const int rs = (ntiles + 7) / 8;
    const int wpx = rs < WS_MAX_WG_PER_XCD ? rs : WS_MAX_WG_PER_XCD;
    if (grouped) return launch_ws_variant<2>(p, 8 * wpx, s);
    if (accum) return launch_ws_variant<3>(p, 8 * wpx, s);
    return fuse ? launch_ws_variant<1>(p, 8 * wpx, s) : launch_ws_variant<0>(p, 8 * wpx, s);
}

}  // namespace pfnl

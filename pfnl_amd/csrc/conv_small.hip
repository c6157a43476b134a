// Small-shape convolutions of the progressive-fusion trunk (BASELINE.json configs[0]: 7 x 32 x 32, configs[4]: 5 x 64 x 64; reference
// model/pfnl.py:24 `in_size=32`, :65-74): conv1_i, conv10_i, the WHOLE of conv2_i (3x3 over concat([base, inp1_t]), 128 -> 64, as
// the reference writes it - below ~a tile per CU the shared-base split only adds a launch) and convmerge1, as ONE kernel template.
//
// Why a kernel of its own: the persistent kernels (conv_split16.hip, conv_sf.hip, conv_wino_ws.hip) are built around 8 x 32-pixel
// tiles, 72 KB of weights per workgroup in the prologue and one workgroup per CU - at 7 x 32 x 32 a launch has 28 such tiles
// for 256 CUs and costs 15 - 27 us whatever it computes (round 2: 85 launches, 1.6 ms).  Here a workgroup is 8 waves (two per
// SIMD: they cover each other's LDS and weight-load latency) and owns R rows x 32 pixels x 64 output channels (R = 1, 2 or 3,
// picked so that a launch is a whole number of ~full rounds of workgroups), and nothing is resident: K (= taps x input
// channels) is walked source by source (a source = the 64 channels of one input tensor at the tile: a frame, or `base`), split
// over the 8 waves as (output channel tile nt, 16-channel group kq of every source); the four K parts meet in LDS at the end.
//   * arithmetic: that of conv_split16.hip - every fp32 operand split exactly into two binary16 numbers, three
//     v_mfma_f32_32x32x16_f16 per product block, fp32 accumulation, cross terms in a second accumulator (x 2^-11 at the end);
//   * activations fp32 NHWC in HBM (L2-resident at these sizes); a source's (R + 2) x 34 halo is loaded as 16-byte pieces, split on
//     the way into LDS ([pixel][half][hi | lo'][32 ch], 16-byte chunks XOR-swizzled with the pixel column), double-buffered, one
//     barrier per source; out-of-image pixels: out-of-range buffer offsets (zeros);
//   * weights never touch LDS: they are the MFMA's B operand (lane = output channel), packed per wave in the order it walks
//     them ([kq][nt][source][tap][hi / lo'][lane] x 16 B) and streamed L2 -> registers through a ring up to a source deep;
//   * MFMA roles A = pixels, B = weights: D[pixel][cout], so the K-half partials go to LDS as conflict-free 4-byte stores and
//     come back as whole 16-byte channel pieces for a coalesced NHWC epilogue (bias, leaky-relu, residual).
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"
#include "conv_small.h"

namespace pfnl {

typedef _Float16 cmh8 __attribute__((ext_vector_type(8)));
typedef _Float16 cmh4 __attribute__((ext_vector_type(4)));
typedef unsigned cmu4 __attribute__((ext_vector_type(4)));
typedef unsigned cmu2 __attribute__((ext_vector_type(2)));

constexpr int CM_NW = 8;                                            // waves per workgroup: (cout tile nt) x (channel group kq of every source: 16 channels)
constexpr int CM_THREADS = CM_NW * 64;
constexpr int CM_NQ = CM_NW / 2;                                    // K parts that meet in LDS
constexpr int CM_CGW = 8 / CM_NW;                                   // 16-channel groups of a source per wave (1)
constexpr int CM_ASUM_MAX = 7;                                      // partial tensors a summed source can have (T <= 7)

#ifdef PFNL_CM_TIMING   /* phase timeline of a workgroup (tools/cm_timing.py); not part of the product build */
__device__ long long cm_dbg[4096 * 16];
#define CM_STAMP(i_) do { if (tid == 0 && blockIdx.x < 4096) cm_dbg[blockIdx.x * 16 + (i_)] = wall_clock64(); } while (0)
#else
#define CM_STAMP(i_) do {} while (0)
#endif

template <int KS, int R>
struct CmGeom {
    static constexpr int IH = R + KS - 1, IW = 32 + KS - 1;
    static constexpr int NPIX = IH * IW;
    static constexpr int BUF_BYTES = NPIX * 256;                    // one source's halo in operand form
    static constexpr int PIECES = NPIX * 16;                        // 16-byte fp32 pieces (4 channels)
    static constexpr int ITERS = (PIECES + CM_THREADS - 1) / CM_THREADS;
    static constexpr int STEPS = CM_CGW * KS * KS;                  // k-steps (16 channels x one tap) of a wave per source
    static constexpr int RING = KS == 3 ? (R == 3 ? 3 : 9) : 1;     // weight operands in flight (divides STEPS)
    static constexpr int RED_BYTES = CM_NQ * R * 32 * 256;          // the K parts of the tile, fp32 pixel lines
    static constexpr int LDS_BYTES = (2 * BUF_BYTES > RED_BYTES ? 2 * BUF_BYTES : RED_BYTES);
    static constexpr int XT_BYTES = R * 32 * 256;                   // fused 1x1 (X stage): the activated tile in operand form, behind the rest
    static constexpr int LDS_BYTES_X = LDS_BYTES + XT_BYTES;
};

__device__ __forceinline__ void cm_split4(f32x4 v, cmu2& hi, cmu2& lo, float nscale) {   // conv_split16.hip split4
    const cmh4 h = __builtin_convertvector(v, cmh4);
    hi = __builtin_bit_cast(cmu2, h);
    const f32x4 t = v * 2048.0f;
    unsigned l0, l1;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi.x), "s"(nscale), "v"(t.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(hi.x), "s"(nscale), "v"(t.y));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi.y), "s"(nscale), "v"(t.z));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(hi.y), "s"(nscale), "v"(t.w));
    lo = cmu2{l0, l1};
}

// XS: the X stage (conv10_i's partial of this frame behind conv1_i's tile, ConvSmallParams::x_*);  ASUM: source `a` is the
// activated sum of a_nsum tensors (conv2_i taking `base` from conv10_i's per-frame partials)
template <int KS, int R, bool XS = false, bool ASUM = false>
__global__ __launch_bounds__(CM_THREADS, 1) void conv_small_kernel(ConvSmallParams p) {
    using G = CmGeom<KS, R>;
    constexpr int PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char cm_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 1;                                        // output channels 32 nt .. 32 nt + 31
    const int kq = wave >> 1;                                       // channels 16 kq .. 16 kq + 15 of every source
    const int H = p.H, W = p.W;
    const int tiles_x = (W + 31) >> 5, tiles_y = (H + R - 1) / R;
    const int per_item = tiles_x * tiles_y;
    const int item = blockIdx.x / per_item;
    const int sp = blockIdx.x - item * per_item;
    const int ty = sp / tiles_x;
    const int y0 = ty * R, x0 = (sp - ty * tiles_x) * 32;
    const size_t hw64 = (size_t)H * W * 64;
    const int item_bytes = H * W * 256;
    const int wbytes = W * 256;
    const int nsrc = p.nsrc;
    const int ntot = nsrc * G::STEPS;                               // k-steps of this wave

    // staging map: piece id = k * 256 + tid -> halo pixel id >> 4, 4-channel piece c = id & 15.  LDS: pixel * 256 B, 16-byte chunk
    // index = 8 M + 4 part + (channel group of 8 within the half), stored at chunk ^ (column & 15); a piece is half a chunk.
    int grel[G::ITERS], lpk[G::ITERS];
#pragma unroll
    for (int k = 0; k < G::ITERS; ++k) {
        const int id = min(k * CM_THREADS + tid, G::PIECES - 1);    // surplus threads redo the last piece (same value)
        const int pix = id >> 4, c = id & 15;
        const int py = pix / G::IW, px = pix - py * G::IW;
        const int chunk = (c >> 3) * 8 + ((c & 7) >> 1);            // hi part; lo' = chunk + 4
        grel[k] = py * wbytes + px * 256 + c * 16;
        lpk[k] = (pix * 256 + ((chunk ^ (px & 15)) << 4) + (c & 1) * 8) | (py << 16) | (px << 24);
    }
    const float nscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -2048.0f)));
    f32x4 stg[G::ITERS];
    auto request = [&](int s, auto first) __attribute__((always_inline)) {
        const int org = ((y0 - PAD) * W + x0 - PAD) * 256;
        if constexpr (ASUM && decltype(first)::value) {             // (only the prologue's call: source 0 = `a`; nA == 1 with ASUM)
            {                                                       // `a` = lrelu(sum of the a_nsum partial tensors + a_bias)
                // every partial's piece is requested before the first is used (ONE round trip to L2, not a_nsum of them): the loop
                // is written out over CM_ASUM_MAX tensors, the ones past a_nsum with an empty resource (zero, no memory access);
                // the sum runs in the fixed order j = 0, 1, ... (deterministic)
                const int ns = p.a_nsum;
                const float* const src0 = p.a + (size_t)(item / p.a_div) * ns * hw64;
                int off[G::ITERS];
#pragma unroll
                for (int k = 0; k < G::ITERS; ++k) {
                    const int gy = y0 + ((lpk[k] >> 16) & 0xff) - PAD, gx = x0 + ((unsigned)lpk[k] >> 24) - PAD;
                    const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                    off[k] = in ? org + grel[k] : 0x7fffffff;
                }
                constexpr int NB = CM_ASUM_MAX;                     // tensors per batch (registers: NB x ITERS x 4; the prologue holds little else)
                f32x4 sum[G::ITERS];
#pragma unroll
                for (int j0 = 0; j0 < CM_ASUM_MAX; j0 += NB) {
                    f32x4 part[NB][G::ITERS];
#pragma unroll
                    for (int jj = 0; jj < NB; ++jj) {
                        const int j = j0 + jj;
                        const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc(
                            const_cast<float*>(src0 + (size_t)min(j, ns - 1) * hw64), 0, (j < ns && j < CM_ASUM_MAX) ? item_bytes : 0, 0x00020000);
#pragma unroll
                        for (int k = 0; k < G::ITERS; ++k) part[jj][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rj, off[k], 0, 0));
                    }
#pragma unroll
                    for (int k = 0; k < G::ITERS; ++k)
#pragma unroll
                        for (int jj = 0; jj < NB; ++jj) sum[k] = (j0 + jj == 0) ? part[0][k] : sum[k] + part[jj][k];
                }
#pragma unroll
                for (int k = 0; k < G::ITERS; ++k) {
                    const int id = min(k * CM_THREADS + tid, G::PIECES - 1);
                    f32x4 v = sum[k];
                    v += *reinterpret_cast<const f32x4*>(p.a_bias + (id & 15) * 4);
                    v.x = fmaxf(v.x, 0.2f * v.x);
                    v.y = fmaxf(v.y, 0.2f * v.y);
                    v.z = fmaxf(v.z, 0.2f * v.z);
                    v.w = fmaxf(v.w, 0.2f * v.w);
                    stg[k] = off[k] == 0x7fffffff ? f32x4{0.f, 0.f, 0.f, 0.f} : v;   // SAME padding of `base`: zeros, not lrelu(bias)
                }
                return;
            }
        }
        const float* src = s < p.nA ? p.a + (size_t)(item / p.a_div) * hw64 : p.b + ((size_t)item * p.b_mul + (s - p.nA)) * hw64;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, item_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < G::ITERS; ++k) {
            const int gy = y0 + ((lpk[k] >> 16) & 0xff) - PAD, gx = x0 + ((unsigned)lpk[k] >> 24) - PAD;
            const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            stg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, in ? org + grel[k] : 0x7fffffff, 0, 0));
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < G::ITERS; ++k) {
            cmu2 hi, lo;
            cm_split4(stg[k], hi, lo, nscale);
            const int a = (lpk[k] & 0xffff) + buf * G::BUF_BYTES;
            *reinterpret_cast<cmu2*>(cm_smem + a) = hi;
            *reinterpret_cast<cmu2*>(cm_smem + (a ^ 64)) = lo;      // lo' chunk = hi chunk ^ 4
        }
    };

    // pixel operand of (tap ky kx, row r, part): chunk 8 (kq >> 1) + 4 part + 2 (kq & 1) + (lane >> 5) of halo pixel
    // (r + ky, (lane & 31) + kx): address = paddr[kx] ^ (part * 64) + (r + ky) * IW * 256
    int paddr[KS];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
        const int col = (lane & 31) + kx;
        paddr[kx] = col * 256 + ((((kq >> 1) * 8 + (kq & 1) * 2 + (lane >> 5)) ^ (col & 15)) << 4);
    }
    CM_STAMP(0);
    request(0, std::true_type{});                                   // (ahead of the weight ring: the summed source is register-hungry)
    CM_STAMP(1);
    // weights: this wave's steps are contiguous: [kq][nt][source][tap][part][lane] x 16 B
    const cmu4* const wsrc = reinterpret_cast<const cmu4*>(p.wpack) + ((size_t)(kq * 2 + nt) * ntot) * 128 + lane;
    cmu4 wring[G::RING][2];
#pragma unroll
    for (int n = 0; n < G::RING; ++n) {
        const int m = min(n, ntot - 1);
        wring[n][0] = wsrc[(size_t)m * 128];
        wring[n][1] = wsrc[(size_t)m * 128 + 64];
    }
    [[maybe_unused]] cmu4 xw[2];                                    // X stage: this wave's slice of conv10_i (source t = this frame), requested now
    if constexpr (XS) {
        const cmu4* const xs = reinterpret_cast<const cmu4*>(p.x_wpack) + ((size_t)(kq * 2 + nt) * p.x_T + (item % p.x_T)) * 128 + lane;
        xw[0] = xs[0];
        xw[1] = xs[64];
    }
    f32x16 accm[R], accc[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            accm[r][i] = 0.f;
            accc[r][i] = 0.f;
        }


    commit(0);
    CM_STAMP(2);
    __syncthreads();
    CM_STAMP(3);
    for (int c = 0; c < nsrc; ++c) {
        const unsigned char* const tile = cm_smem + (c & 1) * G::BUF_BYTES;
        const bool more = c + 1 < nsrc;                             // (wave-uniform)
        if (more) request(c + 1, std::false_type{});
        const int nbase = c * G::STEPS;
        // Written-out pipeline (left alone, hipcc sinks every weight load and every ds_read to the instruction in front of its MFMA
        // and waits for it there: measured 6.8 us for 108 MFMAs): the pixel operands of step j + 1 are read before the MFMAs of
        // step j, the ring slot a step has used is refilled right behind its MFMAs; sched_barrier fences hold the order.
        cmh8 aop[2][R][2];
        auto read_ops = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            constexpr int tap = j, ky = tap / KS, kx = tap % KS;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned char* const q = tile + (r + ky) * (G::IW * 256);
                aop[j & 1][r][0] = *reinterpret_cast<const cmh8*>(q + paddr[kx]);
                aop[j & 1][r][1] = *reinterpret_cast<const cmh8*>(q + (paddr[kx] ^ 64));
            }
        };
        read_ops(std::integral_constant<int, 0>{});
        auto step = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + 1 < G::STEPS) read_ops(std::integral_constant<int, j + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            const cmh8 wh = __builtin_bit_cast(cmh8, wring[j % G::RING][0]), wo = __builtin_bit_cast(cmh8, wring[j % G::RING][1]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                accm[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop[j & 1][r][0], wh, accm[r], 0, 0, 0);
                accc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop[j & 1][r][0], wo, accc[r], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) accc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop[j & 1][r][1], wh, accc[r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int m = min(nbase + j + G::RING, ntot - 1);       // the ring slot is free again: RING steps ahead
            wring[j % G::RING][0] = wsrc[(size_t)m * 128];
            wring[j % G::RING][1] = wsrc[(size_t)m * 128 + 64];
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{});
        if constexpr (KS == 3) {
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{});
            step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{});
            step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{});
        }
        if (c == 0) CM_STAMP(4);
        if (more) commit((c & 1) ^ 1);
        __syncthreads();                                            // the next source is complete; this one's buffer is free
    }
    CM_STAMP(5);

    // ---- the two K halves meet in LDS (the halo buffers are free now): [kh][row][pixel] x 256 B
    constexpr int EP = (R * 32 * 16) / CM_THREADS;
    f32x4 rsd[EP];
#pragma unroll
    for (int k = 0; k < EP; ++k) {                                  // the residual pieces are requested now: their latency passes under the meeting
        const int id = k * CM_THREADS + tid;
        const int pp = id >> 4, cc = id & 15;
        const int y = min(y0 + (pp / 32), H - 1), x = min(x0 + (pp & 31), W - 1);
        rsd[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.resid) rsd[k] = *reinterpret_cast<const f32x4*>(p.resid + (((size_t)item * H + y) * W + x) * 64 + cc * 4);
    }
    {
        float* const red = reinterpret_cast<float*>(cm_smem) + (kq * R * 32) * 64 + 32 * nt + (lane & 31);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(r * 32 + drow(i, lane)) * 64] = accm[r][i] + accc[r][i] * (1.0f / 2048.0f);
    }
    __syncthreads();
    CM_STAMP(6);
    const float slope = p.act ? 0.2f : 1.0f;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + (tid & 15) * 4);
#pragma unroll
    for (int k = 0; k < EP; ++k) {
        const int id = k * CM_THREADS + tid;
        const int pp = id >> 4, cc = id & 15;                       // pixel r * 32 + x of the tile, 4-channel piece
        const int y = y0 + (pp >> 5), x = x0 + (pp & 31);
        f32x4 v = bias4;
#pragma unroll
        for (int q = 0; q < CM_NQ; ++q) v += *reinterpret_cast<const f32x4*>(cm_smem + (q * R * 32 + pp) * 256 + cc * 16);
        v.x = fmaxf(v.x, v.x * slope);
        v.y = fmaxf(v.y, v.y * slope);
        v.z = fmaxf(v.z, v.z * slope);
        v.w = fmaxf(v.w, v.w * slope);
        if constexpr (XS) {                                         // the activated tile in operand form (commit()'s layout, no halo)
            cmu2 hi, lo;
            cm_split4(v, hi, lo, nscale);
            const int chunk = (cc >> 3) * 8 + ((cc & 7) >> 1);
            const int a = G::LDS_BYTES + pp * 256 + ((chunk ^ (pp & 15)) << 4) + (cc & 1) * 8;
            *reinterpret_cast<cmu2*>(cm_smem + a) = hi;
            *reinterpret_cast<cmu2*>(cm_smem + (a ^ 64)) = lo;
        }
        if (y < H && x < W) {
            const size_t o = (((size_t)item * H + y) * W + x) * 64 + cc * 4;
            v += rsd[k];
            *reinterpret_cast<f32x4*>(p.out + o) = v;
        }
    }
    CM_STAMP(7);
    if constexpr (XS) {
        // ---- X stage: conv10_i's partial of this frame, W10_t^T . tile (K = 64: one k-step per wave, three MFMAs per row); the four
        // K parts meet in the same LDS area the 3x3 used; the sum leaves as fp32 NHWC - the NEXT launch (conv2_i) adds the T partials
        __syncthreads();                                            // the tile is complete; the meeting area has been read
        const cmh8 wh = __builtin_bit_cast(cmh8, xw[0]), wo = __builtin_bit_cast(cmh8, xw[1]);
        const int col = lane & 31;
        const int xoff = col * 256 + ((((kq >> 1) * 8 + (kq & 1) * 2 + (lane >> 5)) ^ (col & 15)) << 4);   // hi part; lo' = ^ 64
        const unsigned char* const xt = cm_smem + G::LDS_BYTES;
        float* const red = reinterpret_cast<float*>(cm_smem) + (kq * R * 32) * 64 + 32 * nt + (lane & 31);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const cmh8 ah = *reinterpret_cast<const cmh8*>(xt + r * (32 * 256) + xoff);
            const cmh8 al = *reinterpret_cast<const cmh8*>(xt + r * (32 * 256) + (xoff ^ 64));
            f32x16 zm, zc;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                zm[i] = 0.f;
                zc[i] = 0.f;
            }
            zm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, zm, 0, 0, 0);
            zc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wo, zc, 0, 0, 0);
            zc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, zc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(r * 32 + drow(i, lane)) * 64] = zm[i] + zc[i] * (1.0f / 2048.0f);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EP; ++k) {
            const int id = k * CM_THREADS + tid;
            const int pp = id >> 4, cc = id & 15;
            const int y = y0 + (pp >> 5), x = x0 + (pp & 31);
            f32x4 v = *reinterpret_cast<const f32x4*>(cm_smem + pp * 256 + cc * 16);
#pragma unroll
            for (int q = 1; q < CM_NQ; ++q) v += *reinterpret_cast<const f32x4*>(cm_smem + (q * R * 32 + pp) * 256 + cc * 16);
            if (y < H && x < W) *reinterpret_cast<f32x4*>(p.x_out + (((size_t)item * H + y) * W + x) * 64 + cc * 4) = v;
        }
    }
}

// ---- conv10_i at small shapes: a 1x1 has no halo, so the A operand comes straight from HBM / L2 (as conv1x1.hip does) and the whole
// K = T x 64 is in flight at once: every source's 32 bytes per lane are requested up front, split in registers, no LDS and no
// barrier before the final meeting of the two K halves.  Tile = 32 consecutive pixels (linear over H x W) x 64 output channels.
constexpr int CM1_MAXSRC = 8, CM1_THREADS = 512;
__global__ __launch_bounds__(CM1_THREADS, 2) void conv_small_1x1_kernel(ConvSmallParams p) {
    // 8 waves = (output channel tile nt, channel half kh, source parity sh): a wave owns <= 4 sources x 2 k-steps: everything it
    // needs (4 x 64 B of pixels per lane, 8 x 2 weight operands) is requested at once, one round trip to L2 in all
    __shared__ __attribute__((aligned(16))) float red1[4 * 32 * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 1, kh = (wave >> 1) & 1, sh = wave >> 2;
    const int HW = p.H * p.W;
    const int gpi = (HW + 31) >> 5;
    const int item = blockIdx.x / gpi;
    const int p0 = (blockIdx.x - item * gpi) * 32;
    const int nsrc = p.nsrc;
    const float nscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, -2048.0f)));
    const size_t hw64 = (size_t)HW * 64;
    // lane (pixel p0 + (lane & 31), k half lane >> 5): channels 32 kh + 16 jj + 8 (lane >> 5) + e of a source
    const size_t poff = (size_t)min(p0 + (lane & 31), HW - 1) * 64 + 32 * kh + 8 * (lane >> 5);
    // weights (conv_small_pack_weights, ks = 1): [channel group kq = 2 kh + jj][nt][source][part][lane] x 16 B
    const cmu4* const wsrc = reinterpret_cast<const cmu4*>(p.wpack) + lane;
    f32x4 av[CM1_MAXSRC / 2][2][2];
    cmu4 wv[CM1_MAXSRC / 2][2][2];
#pragma unroll
    for (int i = 0; i < CM1_MAXSRC / 2; ++i) {
        const int sc = min(2 * i + sh, nsrc - 1);                   // (surplus slots re-read the last source: never used)
        const float* src = (sc < p.nA ? p.a + (size_t)(item / p.a_div) * hw64 : p.b + ((size_t)item * p.b_mul + (sc - p.nA)) * hw64) + poff;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            av[i][jj][0] = *reinterpret_cast<const f32x4*>(src + 16 * jj);
            av[i][jj][1] = *reinterpret_cast<const f32x4*>(src + 16 * jj + 4);
            const size_t wn = ((size_t)((2 * kh + jj) * 2 + nt) * nsrc + sc) * 128;
            wv[i][jj][0] = wsrc[wn];
            wv[i][jj][1] = wsrc[wn + 64];
        }
    }
    f32x16 accm, accc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        accm[r] = 0.f;
        accc[r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < CM1_MAXSRC / 2; ++i) {
        if (2 * i + sh < nsrc) {                                    // wave-uniform; no memory operation inside
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                cmu2 h0, l0, h1, l1;
                cm_split4(av[i][jj][0], h0, l0, nscale);
                cm_split4(av[i][jj][1], h1, l1, nscale);
                const cmh8 ah = __builtin_bit_cast(cmh8, cmu4{h0.x, h0.y, h1.x, h1.y}), al = __builtin_bit_cast(cmh8, cmu4{l0.x, l0.y, l1.x, l1.y});
                const cmh8 wh = __builtin_bit_cast(cmh8, wv[i][jj][0]), wo = __builtin_bit_cast(cmh8, wv[i][jj][1]);
                accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, accm, 0, 0, 0);
                accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wo, accc, 0, 0, 0);
                accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, accc, 0, 0, 0);
            }
        }
    }
    {
        float* const red = red1 + ((sh * 2 + kh) * 32) * 64 + 32 * nt + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) red[drow(r, lane) * 64] = accm[r] + accc[r] * (1.0f / 2048.0f);
    }
    __syncthreads();
    const float slope = p.act ? 0.2f : 1.0f;
    const int pp = tid >> 4, cc = tid & 15;                         // one 16-byte piece per thread
    f32x4 v = *reinterpret_cast<const f32x4*>(p.bias + cc * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) v += *reinterpret_cast<const f32x4*>(red1 + (q * 32 + pp) * 64 + cc * 4);
    v.x = fmaxf(v.x, v.x * slope);
    v.y = fmaxf(v.y, v.y * slope);
    v.z = fmaxf(v.z, v.z * slope);
    v.w = fmaxf(v.w, v.w * slope);
    if (p0 + pp < HW) {
        const size_t o = ((size_t)item * HW + p0 + pp) * 64 + cc * 4;
        if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + o);
        *reinterpret_cast<f32x4*>(p.out + o) = v;
    }
}

template <int KS, int R, bool XS = false, bool ASUM = false>
static hipError_t cm_launch(const ConvSmallParams& p, int tiles, hipStream_t s) {
    using G = CmGeom<KS, R>;
    constexpr int LDS = XS ? G::LDS_BYTES_X : G::LDS_BYTES;
    static std::atomic<int> attr_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_dev[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_small_kernel<KS, R, XS, ASUM>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_dev[dev] = 1;
    }
    hipLaunchKernelGGL((conv_small_kernel<KS, R, XS, ASUM>), dim3(tiles), dim3(CM_THREADS), LDS, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_small(const ConvSmallParams& p, hipStream_t s) {
    if (!p.b || !p.wpack || !p.bias || !p.out || p.items < 1 || p.H < 1 || p.W < 1 || p.nsrc < 1 || p.nA < 0 || p.nA > p.nsrc) return hipErrorInvalidValue;
    if ((p.nA > 0 && (!p.a || p.a_div < 1)) || p.b_mul < 1 || (p.ks != 1 && p.ks != 3)) return hipErrorInvalidValue;
    if ((long long)p.H * p.W * 256 >= 0x7fffffffLL) return hipErrorInvalidValue;
    const int tiles_x = (p.W + 31) / 32;
    // rows per workgroup (one 8-wave workgroup per CU at a time): a CU's time is ~ (workgroups it gets) x R, so minimise
    // ceil(workgroups / CUs) x R; ties go to the larger R (fewer workgroups to dispatch, fewer weight bytes).
    // 7 x 32 x 32 -> R = 1 (224 workgroups), 5 x 64 x 64 -> R = 3 (220), 7 x 64 x 64 -> R = 2 (448)
    const int ncu = device_cu_count();                              // (of the device this launch goes to)
    if (!ncu) return hipErrorUnknown;
    long long tr[4] = {0, 0, 0, 0};
    int bestR = 1;
    long long best = -1;
    for (int R = 1; R <= 3; ++R) {
        tr[R] = (long long)p.items * ((p.H + R - 1) / R) * tiles_x;
        const long long cost = ((tr[R] + ncu - 1) / ncu) * R;
        if (best < 0 || cost <= best) {
            best = cost;
            bestR = R;
        }
    }
    if (tr[bestR] > 0x7fffffffLL) return hipErrorInvalidValue;
    const long long t2 = tr[2], t1 = tr[1];
    const bool r2 = bestR >= 2;
    const bool xs = p.x_wpack != nullptr, asum = p.a_nsum > 1;
    if (xs && (p.ks != 3 || p.nsrc != 1 || p.nA != 0 || !p.x_out || p.x_T < 1 || p.items % p.x_T || p.resid)) return hipErrorInvalidValue;
    if (asum && (p.ks != 3 || p.nA != 1 || !p.a_bias || xs || p.a_nsum > CM_ASUM_MAX)) return hipErrorInvalidValue;
    if (xs) return bestR == 3 ? cm_launch<3, 3, true, false>(p, (int)tr[3], s) : bestR == 2 ? cm_launch<3, 2, true, false>(p, (int)t2, s) : cm_launch<3, 1, true, false>(p, (int)t1, s);
    if (asum) return bestR == 3 ? cm_launch<3, 3, false, true>(p, (int)tr[3], s) : bestR == 2 ? cm_launch<3, 2, false, true>(p, (int)t2, s) : cm_launch<3, 1, false, true>(p, (int)t1, s);
    if (p.ks == 3) return bestR == 3 ? cm_launch<3, 3>(p, (int)tr[3], s) : bestR == 2 ? cm_launch<3, 2>(p, (int)t2, s) : cm_launch<3, 1>(p, (int)t1, s);
    if (p.ks == 1 && p.nsrc <= CM1_MAXSRC) {                        // conv10_i: no halo, no LDS staging
        const long long g = (long long)p.items * (((long long)p.H * p.W + 31) / 32);
        if (g > 0x7fffffffLL) return hipErrorInvalidValue;
        hipLaunchKernelGGL(conv_small_1x1_kernel, dim3((unsigned)g), dim3(CM1_THREADS), 0, s, p);
        return hipGetLastError();
    }
    (void)r2;
    return cm_launch<1, 1>(p, (int)t1, s);
}

size_t conv_small_pack_halfs(int ks, int nsrc) { return (size_t)nsrc * ks * ks * 4 * 2 * 2 * 512; }   // per source: taps x 4 channel groups x 2 nt x (hi, lo') x 1 KB

// HWIO [ks, ks, 64 * nsrc, cout] -> [channel group kq][nt][source][tap][part][lane][e]:
// W[tap][64 s + 16 kq + 8 (lane >> 5) + e][32 nt + (lane & 31)], part 0 = f16(w), part 1 = f16((w - hi) 2^11); cout < 64: zero-padded
void conv_small_pack_weights(const float* hwio, int ks, int nsrc, int cout, uint16_t* dst) {
    const int steps = CM_CGW * ks * ks, cin = 64 * nsrc;
    for (int kh = 0; kh < CM_NQ; ++kh)
        for (int nt = 0; nt < 2; ++nt)
            for (int s = 0; s < nsrc; ++s)
                for (int j = 0; j < steps; ++j)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int tap = j / CM_CGW, jj = j % CM_CGW;
                            const int ci = 64 * s + 16 * (CM_CGW * kh + jj) + 8 * (lane >> 5) + e;
                            const int co = 32 * nt + (lane & 31);
                            const float w = co < cout ? hwio[((size_t)tap * cin + ci) * cout + co] : 0.f;
                            const _Float16 hi = (_Float16)w;
                            const _Float16 lo = (_Float16)((w - (float)hi) * 2048.0f);
                            uint16_t hb, lb;
                            std::memcpy(&hb, &hi, 2);
                            std::memcpy(&lb, &lo, 2);
                            const size_t n = ((size_t)(kh * 2 + nt) * nsrc + s) * steps + j;
                            dst[n * 1024 + lane * 8 + e] = hb;
                            dst[n * 1024 + 512 + lane * 8 + e] = lb;
                        }
}

}  // namespace pfnl

#ifdef PFNL_CM_TIMING
extern "C" int pfnl_debug_read_cm_stamps(long long* host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pfnl::cm_dbg), n * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

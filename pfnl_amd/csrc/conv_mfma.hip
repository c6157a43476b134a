// MFMA implicit-GEMM convolution for the progressive-fusion trunk (gfx950, f32 in / f32 acc).
//
// Replaces the tf.layers.Conv2D instances conv1_i / conv10_i / conv2_i / convmerge1 of the
// reference (model/pfnl.py:49-52, applied at :66-74): 'same' padding, stride 1, HWIO kernels,
// bias, leaky_relu(0.2).  GEMM view: M = output pixels, N = 64 output channels, K = taps x Cin.
//
// Workgroup = 4 waves = an 8-row x 32-column x 64-channel output tile.  Wave w owns rows 2w,2w+1:
// two 32-pixel M-tiles x two 32-channel N-tiles = four v_mfma_f32_32x32x2_f32 accumulators.
// K is walked in chunks of 16 input channels.  Per chunk the halo tile (10x34 pixels) is staged
// channel-planar in LDS ([c][row][col], odd plane stride) so that the A operand (lane = pixel
// column) and the B operand (lane = output channel, weights [tap][c][64]) are conflict-free
// ds_read_b32; the next chunk's global loads are issued before the current chunk's MFMAs and
// written to LDS after them (register staging), and two workgroups per CU interleave.
#include <cstdlib>

#include "common.h"

namespace pfnl {

template <int KS, int CK_, int MT>
struct ConvGeom {
    static constexpr int CK = CK_;
    static constexpr int HALO = KS - 1;
    static constexpr int TH = 4 * MT;                              // output rows per workgroup (MT per wave)
    static constexpr int IW = CONV_TW + HALO;
    static constexpr int IH = TH + HALO;
    static constexpr int IPIX = IW * IH;
    static constexpr int PS = (IPIX % 2 == 0) ? IPIX + 1 : IPIX;  // odd: planes hit distinct banks
    static constexpr int W_FLOATS = KS * KS * CK * CONV_NPAD;
    static constexpr int IN_FLOATS = CK * PS;
    static constexpr int IN_ITEMS = IPIX * (CK / 4);               // float4 pieces of the halo tile
    static constexpr int IN_ITERS = (IN_ITEMS + 255) / 256;
    static constexpr int W_ITERS = W_FLOATS / 4 / 256;
    static_assert(W_FLOATS % (4 * 256) == 0, "weight chunk must split evenly over 256 threads");
    static constexpr int EPI_STRIDE = 68;                          // epilogue slab row stride (floats)
    static constexpr int EPI_FLOATS = 4 * 32 * EPI_STRIDE;         // one 32-pixel slab per wave
    static constexpr int LDS_FLOATS = (W_FLOATS + IN_FLOATS) > EPI_FLOATS ? (W_FLOATS + IN_FLOATS) : EPI_FLOATS;
    static constexpr size_t LDS_BYTES = size_t(LDS_FLOATS) * sizeof(float);
};

// FUSE = false: out = act(conv + bias);  FUSE = true: out = act(conv + bias + addend) + resid.
// MT = M-tiles (output rows) per wave: 2 for the 64->64 3x3 (B operand reused twice); 1 where the grid
// would otherwise be <= one workgroup per CU (conv10, convmerge1: 256 tiles of 8x32 at 4 x 128 x 128).
template <int KS, int CK_, int MT, bool FUSE>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvParams p) {
    using G = ConvGeom<KS, CK_, MT>;
    constexpr int CK = G::CK, IW = G::IW, PS = G::PS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sw = smem;                    // [KS*KS][CK][64]
    float* s_in = smem + G::W_FLOATS;     // [CK][PS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int xl = lane & 31;
    const int kh = lane >> 5;
    const int item = blockIdx.z;
    const int tx0 = blockIdx.x * CONV_TW;
    const int ty0 = blockIdx.y * G::TH;
    const int H = p.H, W = p.W;

    // Per-thread staging descriptors (identical for every chunk).
    int goff[G::IN_ITERS];
    int loff[G::IN_ITERS];
    unsigned ldmask = 0;   // bit i: global load is inside the image
    unsigned stmask = 0;   // bit i: this thread owns an LDS slot in iteration i
#pragma unroll
    for (int i = 0; i < G::IN_ITERS; ++i) {
        const int it = tid + i * 256;
        goff[i] = 0;
        loff[i] = 0;
        if (it < G::IN_ITEMS) {
            const int pix = it / (CK / 4);
            const int c4 = it % (CK / 4);
            const int py = pix / IW;
            const int px = pix % IW;
            const int gy = ty0 + py - KS / 2;
            const int gx = tx0 + px - KS / 2;
            stmask |= 1u << i;
            loff[i] = (c4 * 4) * PS + py * IW + px;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                ldmask |= 1u << i;
                goff[i] = (gy * W + gx) * p.in_cstride + c4 * 4;
            }
        }
    }

    f32x4 rin[G::IN_ITERS];
    f32x4 rw[G::W_ITERS];
    const size_t frame_floats = (size_t)H * W * p.in_cstride;

#define PFNL_LOAD_CHUNK(q_)                                                                        \
    do {                                                                                           \
        const int fr_ = item * p.frames_per_item + (q_) / p.chunks_per_frame;                      \
        const float* src_ = p.in + (size_t)fr_ * frame_floats + ((q_) % p.chunks_per_frame) * CK;  \
        _Pragma("unroll") for (int i = 0; i < G::IN_ITERS; ++i) {                                  \
            rin[i] = f32x4{0.f, 0.f, 0.f, 0.f};                                                       \
            if ((ldmask >> i) & 1u) rin[i] = *reinterpret_cast<const f32x4*>(src_ + goff[i]) ;     \
        }                                                                                          \
        const f32x4* wsrc_ = reinterpret_cast<const f32x4*>(p.wpack + (size_t)(q_) * G::W_FLOATS); \
        _Pragma("unroll") for (int i = 0; i < G::W_ITERS; ++i) rw[i] = wsrc_[tid + i * 256];       \
    } while (0)

#define PFNL_STORE_CHUNK()                                                                         \
    do {                                                                                           \
        _Pragma("unroll") for (int i = 0; i < G::IN_ITERS; ++i) {                                  \
            if ((stmask >> i) & 1u) {                                                              \
                const f32x4 v_ = ((ldmask >> i) & 1u) ? rin[i] : f32x4{0.f, 0.f, 0.f, 0.f};        \
                float* d_ = s_in + loff[i];                                                        \
                d_[0] = v_.x;                                                                      \
                d_[PS] = v_.y;                                                                     \
                d_[2 * PS] = v_.z;                                                                 \
                d_[3 * PS] = v_.w;                                                                 \
            }                                                                                      \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < G::W_ITERS; ++i)                                     \
            reinterpret_cast<f32x4*>(sw)[tid + i * 256] = rw[i];                                   \
    } while (0)

    f32x16 acc[MT][2];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // lane-dependent bases: A = pixel (row 2*wave(+1), column xl), channel half kh; B = channel xl.
    const float* abase = s_in + kh * PS + (MT * wave) * IW + xl;
    const float* bbase = sw + kh * CONV_NPAD + xl;

    PFNL_LOAD_CHUNK(0);
    for (int q = 0; q < p.nchunks; ++q) {
        PFNL_STORE_CHUNK();
        __syncthreads();
        PFNL_LOAD_CHUNK(q + 1 < p.nchunks ? q + 1 : q);   // in flight while the MFMAs run (last: harmless re-read)
        __builtin_amdgcn_sched_barrier(0);                // keep the loads ahead of the MFMAs
#pragma unroll
        for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
                const float* ap = abase + dy * IW + dx;
                const float* bp = bbase + (dy * KS + dx) * CK * CONV_NPAD;
#pragma unroll
                for (int kk = 0; kk < CK / 2; ++kk) {
                    const float b0 = bp[(2 * kk) * CONV_NPAD];
                    const float b1 = bp[(2 * kk) * CONV_NPAD + 32];
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) {
                        const float a = ap[(2 * kk) * PS + mi * IW];
                        acc[mi][0] = mfma32(a, b0, acc[mi][0]);
                        acc[mi][1] = mfma32(a, b1, acc[mi][1]);
                    }
                }
            }
        }
        __syncthreads();                     // everyone finished reading this chunk
    }

    // Epilogue.  D[i = pixel column][j = channel]: lane holds channel xl (+32) at 16 columns.  Each
    // wave transposes one 32-pixel x 64-channel M-tile through its private LDS slab so that the
    // bias/addend/residual loads and the stores are 16-byte, channel-contiguous (NHWC) accesses.
    constexpr int ES = G::EPI_STRIDE;                        // slab row stride (floats), 16B aligned
    float* slab = smem + wave * (32 * ES);
    const int c4 = lane & 15;                                // this lane's channel quad
    const int prow = lane >> 4;                              // pixel = prow + 4*j
    const bool cvalid = c4 * 4 < p.cout;
    const int c4c = cvalid ? c4 : 0;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + c4c * 4);   // p.bias is never null
    const float slope = p.act ? 0.2f : 1.0f;                 // max(v, slope*v): lrelu or identity
    const int aitem = FUSE ? item / p.add_div : 0;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int y = ty0 + MT * wave + mi;
        const bool live = y < H && cvalid;
        const int yc = y < H ? y : H - 1;
        const size_t rowpix = ((size_t)item * H + yc) * W;
        // issue the addend / residual loads first (clamped addresses; masked at the store)
        f32x4 av[8], rv[8];
        if (FUSE) {
            const float* ap = p.addend + (((size_t)aitem * H + yc) * W) * CONV_NPAD + c4c * 4;
            const float* rp = p.resid + rowpix * p.out_cstride + c4c * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int x = tx0 + prow + 4 * j;
                const int xc = x < W ? x : W - 1;
                av[j] = *reinterpret_cast<const f32x4*>(ap + (size_t)xc * CONV_NPAD);
                rv[j] = *reinterpret_cast<const f32x4*>(rp + (size_t)xc * p.out_cstride);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[drow(r, lane) * ES + ni * 32 + xl] = acc[mi][ni][r];
        float* op = p.out + rowpix * p.out_cstride + c4 * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int px = prow + 4 * j;
            const int x = tx0 + px;
            f32x4 t = *reinterpret_cast<const f32x4*>(slab + px * ES + c4 * 4) + bias4;
            if (FUSE) t += av[j];
            t.x = fmaxf(t.x, slope * t.x);
            t.y = fmaxf(t.y, slope * t.y);
            t.z = fmaxf(t.z, slope * t.z);
            t.w = fmaxf(t.w, slope * t.w);
            if (FUSE) t += rv[j];
            if (live && x < W) *reinterpret_cast<f32x4*>(op + (size_t)x * p.out_cstride) = t;
        }
    }
}

template <int KS, int CK_, int MT, bool FUSE>
static hipError_t launch_variant(const ConvParams& p, int items, hipStream_t s) {
    const dim3 grid((p.W + CONV_TW - 1) / CONV_TW, (p.H + 4 * MT - 1) / (4 * MT), items);
    static std::atomic<int> attr_set[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<KS, CK_, MT, FUSE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ConvGeom<KS, CK_, MT>::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = 1;
    }
    constexpr size_t lds_bytes = ConvGeom<KS, CK_, MT>::LDS_BYTES;
    hipLaunchKernelGGL((conv_mfma_kernel<KS, CK_, MT, FUSE>), grid, dim3(256), lds_bytes, s, p);
    return hipGetLastError();
}

// K-chunk depth per kernel size.  The next chunk's global loads are issued one chunk of MFMAs ahead; a
// 1x1 chunk of 16 channels is only 32 MFMAs per wave (~2k cycles), far less than the loaded memory
// latency (~8k cycles), so the 1x1 kernel walks 32-channel chunks (64-channel chunks measured slower: 1 workgroup per CU).
int conv_ck(int ksize) { return ksize == 1 ? 32 : CONV_CK; }

hipError_t launch_conv_mfma(const ConvParams& p0, int ksize, int items, hipStream_t s) {
    if (!p0.bias || !p0.in || !p0.wpack || !p0.out) return hipErrorInvalidValue;
    const bool fuse = p0.addend != nullptr || p0.resid != nullptr;
    if (fuse && (!p0.addend || !p0.resid || p0.add_div < 1)) return hipErrorInvalidValue;
    ConvParams p = p0;
    p.chunks_per_frame = p.in_cstride / conv_ck(ksize);
    p.nchunks = p.frames_per_item * p.chunks_per_frame;
    // 8-row tiles unless that leaves the chip with fewer than two workgroups per CU
    const long tiles8 = (long)((p.W + CONV_TW - 1) / CONV_TW) * ((p.H + 7) / 8) * items;
    bool small = tiles8 < 3 * 256;
    if (const char* e = getenv("PFNL_CONV_MT")) {                 // test hook: force the tile height
        if (e[0] == '1') small = true;
        if (e[0] == '2') small = false;
    }
    if (ksize == 3) {
        if (fuse) return launch_variant<3, 16, 2, true>(p, items, s);
        return small ? launch_variant<3, 16, 1, false>(p, items, s) : launch_variant<3, 16, 2, false>(p, items, s);
    }
    if (ksize == 1) {
        if (fuse) return launch_variant<1, 32, 2, true>(p, items, s);
        return small ? launch_variant<1, 32, 1, false>(p, items, s) : launch_variant<1, 32, 2, false>(p, items, s);
    }
    return hipErrorInvalidValue;
}

size_t conv_pack_floats(int ksize, int cin) { return (size_t)cin * ksize * ksize * CONV_NPAD; }

void conv_pack_weights(const float* hwio, int ksize, int cin_total, int cin_begin, int cin, int cout,
                       float* dst) {
    const int taps = ksize * ksize;
    const int ck = conv_ck(ksize);
    const int nchunks = cin / ck;
    for (int q = 0; q < nchunks; ++q)
        for (int t = 0; t < taps; ++t)
            for (int c = 0; c < ck; ++c)
                for (int o = 0; o < CONV_NPAD; ++o) {
                    const int ci = cin_begin + q * ck + c;
                    float v = 0.f;
                    if (o < cout) v = hwio[((size_t)t * cin_total + ci) * cout + o];
                    dst[(((size_t)q * taps + t) * ck + c) * CONV_NPAD + o] = v;
                }
}

}  // namespace pfnl

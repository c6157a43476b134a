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
// Workgroup = 4 waves = 2x16 tiles (4 rows x 32 columns of output) x 64 cout.  Wave w owns the
// Winograd row xi = w: its 4 positions (nu = 0..3) x 2 N-tiles = 8 accumulators live in registers
// across ALL input-channel chunks, so nothing is re-added per chunk.  Per K-step (2 channels) a
// wave does 8 LDS reads + 12 VALU (input transform on the fly from the raw halo tile, which is
// stored channel-planar with even/odd columns de-interleaved so that lane = tile column reads
// consecutive banks), 2 coalesced 16-byte global loads of pre-packed U (L2-resident, identical for
// every workgroup, software-prefetched one K-step ahead) and 8 MFMAs.  The raw tile (6x34 pixels x
// 16 channels) is double-buffered in LDS: one barrier per chunk.  Epilogue: column transform in
// registers, row transform across the four waves through LDS, then the same fused
// bias / shared-base addend / leaky-relu / residual NHWC epilogue as the direct kernel.
#include "common.h"

namespace pfnl {

constexpr int WN_TX = 16, WN_TY = 2;              // tiles per workgroup (x, y)
constexpr int WN_IW = 2 * WN_TX + 2;              // 34 input columns
constexpr int WN_IH = 2 * WN_TY + 2;              // 6 input rows
constexpr int WN_HALF = 20;                       // floats per column-parity half row (17 used)
constexpr int WN_RP = 2 * WN_HALF;                // row pitch 40: two tile rows are 80 = 16 mod 32 banks apart
constexpr int WN_PS = WN_IH * WN_RP + 1;          // plane stride 241 (odd)
constexpr int WN_CK = 16;
constexpr int WN_BUF = WN_CK * WN_PS;             // floats per raw buffer
constexpr int WN_ES = 68;                         // epilogue slab row stride (floats)
constexpr int WN_EPI = 4 * 2 * 32 * WN_ES;        // [wave][j][tile][cout]
constexpr int WN_LDS_FLOATS = (2 * WN_BUF > WN_EPI) ? 2 * WN_BUF : WN_EPI;
constexpr size_t WN_LDS_BYTES = size_t(WN_LDS_FLOATS) * sizeof(float);
constexpr int WN_IN_ITEMS = WN_IH * WN_IW * (WN_CK / 4);   // 816 float4 pieces
constexpr int WN_IN_ITERS = (WN_IN_ITEMS + 255) / 256;     // 4

template <bool FUSE>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = Winograd row
    const int tx = lane & 15;
    const int ty = (lane >> 4) & 1;
    const int kh = lane >> 5;
    const int item = blockIdx.z;
    const int x0 = blockIdx.x * (2 * WN_TX);
    const int y0 = blockIdx.y * (2 * WN_TY);
    const int H = p.H, W = p.W;

    // staging descriptors of the raw halo tile (same for every chunk)
    int goff[WN_IN_ITERS];
    int loff[WN_IN_ITERS];
    unsigned ldmask = 0, stmask = 0;
#pragma unroll
    for (int i = 0; i < WN_IN_ITERS; ++i) {
        const int it = tid + i * 256;
        goff[i] = 0;
        loff[i] = 0;
        if (it < WN_IN_ITEMS) {
            const int pix = it >> 2, c4 = it & 3;
            const int py = pix / WN_IW, px = pix % WN_IW;
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
    f32x4 rin[WN_IN_ITERS];

#define WN_LOAD_RAW(q_)                                                                          \
    do {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < WN_IN_ITERS; ++i)                                  \
            rin[i] = *reinterpret_cast<const f32x4*>(fin + (q_) * WN_CK + goff[i]);              \
    } while (0)
#define WN_STORE_RAW(buf_)                                                                       \
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

    // B^T row xi = s0 * e[a0] + s1 * e[a1]
    const int a0 = (xi == 0) ? 0 : 1;
    const int a1 = (xi == 3) ? 3 : 2;
    const float s0 = (xi == 2) ? -1.f : 1.f;
    const float s1 = (xi == 0 || xi == 3) ? -1.f : 1.f;
    // lane part of the A address: channel half kh, tile row ty, tile column tx
    const int abase = kh * WN_PS + (2 * ty) * WN_RP + tx;
    const int r0off = abase + a0 * WN_RP;
    const int r1off = abase + a1 * WN_RP;

    f32x16 acc[4][2];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][t][r] = 0.f;

    // packed U: [chunk][xi][kk][lane][nu*2 + nt]
    const f32x4* up = reinterpret_cast<const f32x4*>(p.upack) + ((size_t)xi * 8) * 128 + lane * 2;
    constexpr int UP_CHUNK_F4 = 4 * 8 * 128;     // float4 per chunk
    constexpr int NCHUNK = 64 / WN_CK;

    WN_LOAD_RAW(0);
    WN_STORE_RAW(smem);
    __syncthreads();
    f32x4 bn0 = up[0], bn1 = up[1];
    for (int q = 0; q < NCHUNK; ++q) {
        const float* cur = smem + (q & 1) * WN_BUF;
        const int qn = q + 1 < NCHUNK ? q + 1 : q;
        WN_LOAD_RAW(qn);                                   // next chunk's halo tile, in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        const f32x4* uq = up + (size_t)q * UP_CHUNK_F4;
        const f32x4* uqn = up + (size_t)qn * UP_CHUNK_F4;
#pragma unroll
        for (int kk = 0; kk < WN_CK / 2; ++kk) {
            const f32x4 b0 = bn0, b1 = bn1;
            const f32x4* nx = (kk + 1 < WN_CK / 2) ? uq + (kk + 1) * 128 : uqn;   // next K-step's U
            bn0 = nx[0];
            bn1 = nx[1];
            const float* c0 = cur + r0off + (2 * kk) * WN_PS;
            const float* c1 = cur + r1off + (2 * kk) * WN_PS;
            // t[b] = sum_a Bt[xi][a] d[a][b];  column b lives at (b&1)*HALF + (b>>1)
            const float t0 = fmaf(s1, c1[0], s0 * c0[0]);
            const float t1 = fmaf(s1, c1[WN_HALF], s0 * c0[WN_HALF]);
            const float t2 = fmaf(s1, c1[1], s0 * c0[1]);
            const float t3 = fmaf(s1, c1[WN_HALF + 1], s0 * c0[WN_HALF + 1]);
            const float v0 = t0 - t2, v1 = t1 + t2, v2 = t2 - t1, v3 = t1 - t3;   // V[xi][nu]
            acc[0][0] = mfma32(v0, b0.x, acc[0][0]);
            acc[0][1] = mfma32(v0, b0.y, acc[0][1]);
            acc[1][0] = mfma32(v1, b0.z, acc[1][0]);
            acc[1][1] = mfma32(v1, b0.w, acc[1][1]);
            acc[2][0] = mfma32(v2, b1.x, acc[2][0]);
            acc[2][1] = mfma32(v2, b1.y, acc[2][1]);
            acc[3][0] = mfma32(v3, b1.z, acc[3][0]);
            acc[3][1] = mfma32(v3, b1.w, acc[3][1]);
        }
        WN_STORE_RAW(smem + ((q + 1) & 1) * WN_BUF);       // the buffer nobody reads in this iteration
        __syncthreads();
    }
#undef WN_LOAD_RAW
#undef WN_STORE_RAW

    // column transform (over nu) in registers: R[j] = sum_nu M[xi][nu] At[j][nu]
    // At = [[1,1,1,0],[0,1,-1,-1]]
    float* slab = smem + xi * (2 * 32 * WN_ES);
    const int xl = lane & 31;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = acc[0][t][r], m1 = acc[1][t][r], m2 = acc[2][t][r], m3 = acc[3][t][r];
            const int tile = drow(r, lane);
            slab[(0 * 32 + tile) * WN_ES + t * 32 + xl] = m0 + m1 + m2;
            slab[(1 * 32 + tile) * WN_ES + t * 32 + xl] = m1 - m2 - m3;
        }
    __syncthreads();

    // row transform across waves (over xi) + fused epilogue; thread item = (tile, j, channel quad)
    const int c4 = tid & 15;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + c4 * 4);
    const float slope = p.act ? 0.2f : 1.0f;
    const int aitem = FUSE ? item / p.add_div : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int id = k * 256 + tid;
        const int j = (id >> 4) & 1;
        const int tile = id >> 5;
        const int oy = y0 + 2 * (tile >> 4);
        const int ox = x0 + 2 * (tile & 15) + j;
        const float* sp = smem + (j * 32 + tile) * WN_ES + c4 * 4;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(sp);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(sp + 1 * 2 * 32 * WN_ES);
        const f32x4 r2 = *reinterpret_cast<const f32x4*>(sp + 2 * 2 * 32 * WN_ES);
        const f32x4 r3 = *reinterpret_cast<const f32x4*>(sp + 3 * 2 * 32 * WN_ES);
        f32x4 yv[2];
        yv[0] = r0 + r1 + r2;
        yv[1] = r1 - r2 - r3;
        const bool inx = ox < W;
        const int oxc = inx ? ox : W - 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = oy + i;
            const bool live = inx && y < H;
            const int yc = y < H ? y : H - 1;
            const size_t pix = ((size_t)item * H + yc) * W + oxc;
            f32x4 t = yv[i] + bias4;
            if (FUSE) t += *reinterpret_cast<const f32x4*>(p.addend + (((size_t)aitem * H + yc) * W + oxc) * 64 + c4 * 4);
            t.x = fmaxf(t.x, slope * t.x);
            t.y = fmaxf(t.y, slope * t.y);
            t.z = fmaxf(t.z, slope * t.z);
            t.w = fmaxf(t.w, slope * t.w);
            if (FUSE) t += *reinterpret_cast<const f32x4*>(p.resid + pix * 64 + c4 * 4);
            if (live) *reinterpret_cast<f32x4*>(p.out + pix * 64 + c4 * 4) = t;
        }
    }
}

template <bool FUSE>
static hipError_t launch_wino_variant(const WinoParams& p, dim3 grid, hipStream_t s) {
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
    hipLaunchKernelGGL((conv_wino_kernel<FUSE>), grid, dim3(256), WN_LDS_BYTES, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_wino(const WinoParams& p, int items, hipStream_t s) {
    if (!p.bias || !p.in || !p.upack || !p.out) return hipErrorInvalidValue;
    const bool fuse = p.addend != nullptr || p.resid != nullptr;
    if (fuse && (!p.addend || !p.resid || p.add_div < 1)) return hipErrorInvalidValue;
    if ((p.H & 1) || (p.W & 1)) return hipErrorInvalidValue;     // 2x2 tiles must not straddle the border
    dim3 grid((p.W + 2 * WN_TX - 1) / (2 * WN_TX), (p.H + 2 * WN_TY - 1) / (2 * WN_TY), items);
    return fuse ? launch_wino_variant<true>(p, grid, s) : launch_wino_variant<false>(p, grid, s);
}

size_t wino_pack_floats() { return (size_t)16 * 64 * 64; }

// HWIO [3,3,cin_total,64] rows [cin_begin, cin_begin+64) -> U = G g G^T packed as
// [chunk q][xi][kk][lane][nu*2+nt], value U[xi][nu][cin = 16q + 2kk + (lane>>5)][cout = 32nt + (lane&31)].
void wino_pack_weights(const float* hwio, int cin_total, int cin_begin, float* dst) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int q = 0; q < 4; ++q)
        for (int x = 0; x < 4; ++x)
            for (int kk = 0; kk < 8; ++kk)
                for (int lane = 0; lane < 64; ++lane)
                    for (int nu = 0; nu < 4; ++nu)
                        for (int nt = 0; nt < 2; ++nt) {
                            const int ci = cin_begin + q * 16 + 2 * kk + (lane >> 5);
                            const int co = nt * 32 + (lane & 31);
                            double u = 0.0;
                            for (int a = 0; a < 3; ++a)
                                for (int b = 0; b < 3; ++b)
                                    u += G[x][a] * G[nu][b] *
                                         (double)hwio[(((size_t)a * 3 + b) * cin_total + ci) * 64 + co];
                            dst[((((size_t)q * 4 + x) * 8 + kk) * 64 + lane) * 8 + nu * 2 + nt] = (float)u;
                        }
}

}  // namespace pfnl

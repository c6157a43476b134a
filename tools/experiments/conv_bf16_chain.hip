// The WHOLE of conv2_i of the bf16 trunk in one launch (precision=bf16, BASELINE.json configs[3]; reference model/pfnl.py:69-71 with the
// shared-base split of DESIGN.md section 3) - the bf16 twin of conv3x3_sf_chain_kernel (conv_sf.hip), whose structure it takes over:
//   * a chain = the T + 1 tiles of one (clip, 8 x 32-pixel tile): first the shared half (input `base`, kernel rows 0..63), whose
//     result never leaves the registers - rounded to bf16 exactly where the two-launch form stores it (`pb`), plus the bias, it
//     is the INITIAL C of the T per-frame tiles (input inp1_t, kernel rows 64..127) that follow;
//   * bf16 activations ARE MFMA operands ([items][H][W][64] bf16 = 8 chunks of 16 B per pixel): a tile's 10 x 34 halo goes HBM ->
//     LDS by LDS-DMA (43 wave instructions of 1 KB, the LDS tile's XOR swizzle applied to the source addresses, out-of-image
//     pixels = out-of-range offsets = zeros written), completion by a fence load (see conv_sf.hip);
//   * MFMA roles A = pixels, B = weights: D[pixel][cout] - a lane owns ONE output channel and 16 pixels of a row, so leaky-relu,
//     residual and the bf16 rounding run from registers with 2-byte loads / stores (32 lanes x 2 B = the 64-byte half of a pixel's
//     line): no LDS scratch tile, no barriers for it;
//   * the launch's 72 KB of weights are resident in LDS and only change at the two ends of a chain (shared-half pack <-> per-frame
//     pack): three 24 KB slices by LDS-DMA following the column taps the current tile has consumed.
// A tile is ONE pass of 18 sub-steps x 4 MFMAs (K = 64 in one go: what conv_sf.hip calls the two "parts" of a k-step are here the two
// 16-channel k-steps 2 part + ks, so the LDS addressing is literally the same); the loop is unrolled over two tiles for the
// compile-time parity of the halo buffers.
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"
#include "conv_bf16.h"

#ifndef BC_STORE_AUX
#define BC_STORE_AUX 0
#endif

namespace pfnl {

typedef __bf16 bcb8 __attribute__((ext_vector_type(8)));

constexpr int BC_THREADS = 512;
constexpr int BC_TH = 8, BC_TW = 32, BC_IH = 10, BC_IW = 34;
constexpr int BC_NDMA = (BC_IH * BC_IW + 7) / 8;                    // 43
constexpr int BC_TILE_BYTES = BC_NDMA * 1024;                       // 44 032
constexpr int BC_W_BYTES = 73728;                                   // [kx][ks][ky][m][part][lane] x 16 B: k-step 2 part + ks of tap (ky, kx)
constexpr int BC_SLOT_BYTES = BC_W_BYTES / 3;
constexpr int BC_LDS_BYTES = 2 * BC_TILE_BYTES + BC_W_BYTES + 256;
constexpr int BC_DMA_ITERS = (BC_NDMA + 7) / 8;

__device__ __forceinline__ void bc_dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}
__device__ __forceinline__ f32x16 bc_mfma(bcb8 a, bcb8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float bc_round(float v) { return (float)(__bf16)v; }   // round to nearest even bf16, back in fp32

__global__ __launch_bounds__(BC_THREADS, 1) void conv3x3_bf16_chain_kernel(ConvBf16ChainParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bc_smem[];
    unsigned char* const wl = bc_smem + 2 * BC_TILE_BYTES;
    float* const bl = reinterpret_cast<float*>(bc_smem + 2 * BC_TILE_BYTES + BC_W_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rp = wave >> 1;
    const int nt = wave & 1;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + BC_TW - 1) / BC_TW, tiles_y = (H + BC_TH - 1) / BC_TH;
    const int per_item = tiles_x * tiles_y;
    const int item_bytes = H * W * 128;
    const int wbytes = W * 128;
    const int T = p.T, gT = T + 1;
    const int nchains = per_item * (p.items / T);
    const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3, cpx = gridDim.x >> 3;
    const int per_xcd = (nchains + 7) >> 3;
    const int cbeg = xcd * per_xcd;
    const int ccnt = min(per_xcd, nchains - cbeg);
    if (xj >= ccnt) return;
    const int nt_tiles = ((ccnt - xj + cpx - 1) / cpx) * gT;
#define BC_TILE(k_, f_, clip_, y0_, x0_)                                                         \
    do {                                                                                         \
        /* (scalar integer division runs on the vector unit: readfirstlane keeps the results - and every buffer descriptor   \
           built from them, which the asm DMAs take as "s" operands - in scalar registers) */                                \
        const int ci_ = __builtin_amdgcn_readfirstlane((k_) / gT);                               \
        f_ = (k_) - ci_ * gT;                                                                    \
        const int ch_ = cbeg + xj + ci_ * cpx;                                                   \
        clip_ = __builtin_amdgcn_readfirstlane(ch_ / per_item);                                  \
        const int sp_ = ch_ - clip_ * per_item;                                                  \
        const int ty_ = __builtin_amdgcn_readfirstlane(sp_ / tiles_x);                           \
        y0_ = ty_ * BC_TH;                                                                       \
        x0_ = (sp_ - ty_ * tiles_x) * BC_TW;                                                     \
    } while (0)

    const float bias_r = tid < 64 ? p.bias[tid] : 0.f;
    int dpk[BC_DMA_ITERS];                                          // py | px << 8 of the lane's halo pixel per DMA instruction
#pragma unroll
    for (int k = 0; k < BC_DMA_ITERS; ++k) {
        const int pix = 8 * (wave + 8 * k) + (lane >> 3);
        const int py = pix / BC_IW, px = pix - py * BC_IW;
        dpk[k] = py | (px << 8);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)bc_smem;
    const unsigned ldsw = lds0 + 2 * BC_TILE_BYTES;
#define BC_DMA_HALO(rs_, org_, interior_, y0_, x0_, buf_)                                        \
    do {                                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < BC_DMA_ITERS; ++k_) {                            \
            const int i_ = wave + 8 * k_;                                                        \
            if (k_ < BC_DMA_ITERS - 1 || i_ < BC_NDMA) {                                         \
                const int py_ = dpk[k_] & 0xff, px_ = dpk[k_] >> 8;                              \
                const int gy_ = (y0_) + py_ - 1, gx_ = (x0_) + px_ - 1;                          \
                const bool in_ = (interior_) || ((unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W && py_ < BC_IH); \
                const int rel_ = py_ * wbytes + px_ * 128 + (((lane & 7) ^ ((px_ >> 1) & 7)) << 4); \
                bc_dma16(rs_, lds0 + (buf_) * BC_TILE_BYTES + i_ * 1024, in_ ? (org_) + rel_ : 0x7fffffff); \
            }                                                                                    \
        }                                                                                        \
    } while (0)
    const int wvoff = wave * 1024 + lane * 16;
#define BC_DMA_W(pk_, slot_)                                                                     \
    do {                                                                                         \
        const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc(                    \
            const_cast<uint16_t*>(__builtin_amdgcn_readfirstlane(pk_) ? p.wpack : p.wpack2), 0, BC_W_BYTES, 0x00020000); \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_)                                         \
            bc_dma16(rw_, ldsw + (slot_) * BC_SLOT_BYTES + (wave + 8 * k_) * 1024, (slot_) * BC_SLOT_BYTES + wvoff + k_ * 8192); \
    } while (0)

    int paddr[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = (lane & 31) + kx;
        paddr[kx] = ((2 * rp) * BC_IW + col) * 128 + ((((lane >> 5)) ^ ((col >> 1) & 7)) << 4);
    }
    const int part_xor = 4 << 4;                                    // k-steps 2, 3 (chunks 4..7) = chunk index ^ 4
    const unsigned char* const wlane = wl + nt * 2048 + lane * 16;
    const int ech = 32 * nt + (lane & 31);
    f32x16 acc[2], accp[2], pbv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[n][r] = 0.f;
            accp[n][r] = 0.f;
            pbv[n][r] = 0.f;
        }
    int ex0p = 0, ey0p = 0, eitemp = 0;
    bool pending = false;
    const float slope = p.act ? 0.2f : 1.0f;

    // ---- epilogue from registers: both rows of the pending tile during the next tile.  Register 4 q + j of a lane = pixel column
    // 8 q + j + 4 (lane >> 5): byte offset in the row = evoff + j * 128 (immediate) + q * 1024 (scalar offset); 2 bytes per lane.
    // out == resid (in place): one resource per ROW, built where it is used (this kernel is short of scalar registers, not of time)
    int evoff = 0;
    unsigned short rres[2][16];
    auto row_rsrc = [&](int n) __attribute__((always_inline)) {
        const int ey = ey0p + 2 * rp + n;
        const int nrec = (pending && ey < H) ? wbytes : 0;
        return __builtin_amdgcn_make_buffer_rsrc(p.out + ((size_t)eitemp * H + ey) * W * 64, 0, nrec, 0x00020000);
    };
    auto rows_setup = [&]() __attribute__((always_inline)) { evoff = (ex0p + 4 * (lane >> 5)) * 128 + ech * 2; };
    auto quarter_request = [&](int n, int q) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rsO = row_rsrc(n);
#pragma unroll
        for (int j = 0; j < 4; ++j) rres[n][4 * q + j] = __builtin_amdgcn_raw_buffer_load_b16(rsO, evoff + j * 128, q * 1024, 0);
    };
    auto quarter_finish = [&](int n, int q) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rsO = row_rsrc(n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = accp[n][4 * q + j];                           // (shared half + bias are in: initial C of the tile)
            const float sv = v * slope;
            asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(sv));
            v += __builtin_bit_cast(float, (unsigned)rres[n][4 * q + j] << 16);
            const __bf16 o = (__bf16)v;
            const unsigned short ob = __builtin_bit_cast(unsigned short, o);
            __builtin_amdgcn_raw_buffer_store_b16(ob, rsO, evoff + j * 128, q * 1024, BC_STORE_AUX);
            asm volatile("s_nop 1" ::"v"(ob));
        }
    };
#define BC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // ---- prologue: halo of tile 0 (a shared half: `base`) and the shared-half pack
    int c_f, c_clip, c_y0, c_x0, n_f, n_clip, n_y0, n_x0;
    BC_TILE(0, c_f, c_clip, c_y0, c_x0);
    n_f = c_f;
    n_clip = c_clip;
    n_y0 = c_y0;
    n_x0 = c_x0;
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(p.in2) + (size_t)c_clip * H * W * 64, 0, item_bytes, 0x00020000);
        const int org = ((c_y0 - 1) * W + c_x0 - 1) * 128;
        BC_DMA_HALO(rs, org, false, c_y0, c_x0, 0);
        BC_DMA_W(0, 0);
        BC_DMA_W(0, 1);
        BC_DMA_W(0, 2);
        const unsigned fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
        if (tid < 64) bl[tid] = bias_r;
        asm volatile("" ::"v"(fence));
    }
    __syncthreads();
    const float bias_l = bl[ech];
    int w_pk = 0;                                                   // pack in LDS: 0 = shared half, 1 = per-frame half
    bool w_slice2_owed = false;

    auto tile_pass = [&](auto par, int kt) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;                   // parity of the tile = its halo buffer
        constexpr int cb = PAR;
        const unsigned char* const tile = bc_smem + cb * BC_TILE_BYTES;
        {   // decode the next tile (past the end: this one again - a harmless re-read)
            const int kn = min(kt + 1, nt_tiles - 1);
            BC_TILE(kn, n_f, n_clip, n_y0, n_x0);
        }
        const int nx_pk = n_f != 0;
        const bool w_replace = nx_pk != w_pk;                       // (wave-uniform: only at the two ends of a chain)
        bcb8 X[4][2], Wv[2][2];
#define BC_PX(g_, r_, part_) (*reinterpret_cast<const bcb8*>(tile + (paddr[(g_) >> 1] ^ (((part_) ? part_xor : 0) | (((g_) & 1) << 5))) + (r_) * (BC_IW * 128)))
#define BC_WT(g_, ky_, part_) (*reinterpret_cast<const bcb8*>(wlane + (((g_) * 3 + (ky_)) << 12) + ((part_) << 10)))
        X[0][0] = BC_PX(0, 0, 0);
        X[0][1] = BC_PX(0, 0, 1);
        X[1][0] = BC_PX(0, 1, 0);
        X[1][1] = BC_PX(0, 1, 1);
        Wv[0][0] = BC_WT(0, 0, 0);
        Wv[0][1] = BC_WT(0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (w_slice2_owed) BC_DMA_W(w_pk, 2);                       // slice 2 of THIS tile's pack (its slot was busy until the previous tile's closing barrier)
        const uint16_t* const qsrc = n_f == 0 ? p.in2 + (size_t)n_clip * H * W * 64 : p.in + ((size_t)n_clip * T + (n_f - 1)) * H * W * 64;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qsrc), 0, item_bytes, 0x00020000);
        const int org = ((n_y0 - 1) * W + n_x0 - 1) * 128;
        const bool interior = n_y0 > 0 && n_y0 + BC_IH - 1 <= H && n_x0 > 0 && n_x0 + BC_IW - 1 <= W;
        BC_DMA_HALO(rs, org, interior, n_y0, n_x0, cb ^ 1);
        const unsigned fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
        unsigned fence_w = 0;
        rows_setup();

        auto substep = [&](auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value;
            constexpr int g = S / 3, ky = S % 3;
            // residual lines of the pending tile: all requested in the first four sub-steps, consumed from sub-step 10
            if constexpr (S < 4) {
                quarter_request(S >> 1, 2 * (S & 1));
                quarter_request(S >> 1, 2 * (S & 1) + 1);
            }
            if constexpr (S >= 10) quarter_finish((S - 10) >> 2, (S - 10) & 3);
            if constexpr (ky == 0) {
                if constexpr (g == 2) {
                    asm volatile("" ::"v"(fence));                  // slice 2 of this tile's pack has landed
                    BC_BARRIER();                                   // b0: column tap 0 consumed
                    if (w_replace) BC_DMA_W(nx_pk, 0);
                }
                if constexpr (g == 4) {
                    BC_BARRIER();                                   // b1: column tap 1 consumed
                    if (w_replace) BC_DMA_W(nx_pk, 1);
                    fence_w = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (S < 17) {
                constexpr int S1 = S + 1, g1 = S1 / 3, ky1 = S1 % 3;
                Wv[S1 & 1][0] = BC_WT(g1, ky1, 0);
                Wv[S1 & 1][1] = BC_WT(g1, ky1, 1);
                if constexpr (ky1 == 0) {
                    X[0][0] = BC_PX(g1, 0, 0);
                    X[0][1] = BC_PX(g1, 0, 1);
                    X[1][0] = BC_PX(g1, 1, 0);
                    X[1][1] = BC_PX(g1, 1, 1);
                } else {
                    X[ky1 + 1][0] = BC_PX(g1, ky1 + 1, 0);
                    X[ky1 + 1][1] = BC_PX(g1, ky1 + 1, 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const bcb8 w0 = Wv[S & 1][0], w1 = Wv[S & 1][1];
            if constexpr (S == 0) {                                 // a tile's first products: C = the chain's shared half + bias (0 for that half itself)
                acc[0] = bc_mfma(X[ky][0], w0, pbv[0]);
                acc[1] = bc_mfma(X[ky + 1][0], w0, pbv[1]);
            } else {
                acc[0] = bc_mfma(X[ky][0], w0, acc[0]);
                acc[1] = bc_mfma(X[ky + 1][0], w0, acc[1]);
            }
            acc[0] = bc_mfma(X[ky][1], w1, acc[0]);
            acc[1] = bc_mfma(X[ky + 1][1], w1, acc[1]);
            __builtin_amdgcn_sched_barrier(0);
        };
        substep(std::integral_constant<int, 0>{});
        substep(std::integral_constant<int, 1>{});
        substep(std::integral_constant<int, 2>{});
        substep(std::integral_constant<int, 3>{});
        substep(std::integral_constant<int, 4>{});
        substep(std::integral_constant<int, 5>{});
        substep(std::integral_constant<int, 6>{});
        substep(std::integral_constant<int, 7>{});
        substep(std::integral_constant<int, 8>{});
        substep(std::integral_constant<int, 9>{});
        substep(std::integral_constant<int, 10>{});
        substep(std::integral_constant<int, 11>{});
        substep(std::integral_constant<int, 12>{});
        substep(std::integral_constant<int, 13>{});
        substep(std::integral_constant<int, 14>{});
        substep(std::integral_constant<int, 15>{});
        substep(std::integral_constant<int, 16>{});
        substep(std::integral_constant<int, 17>{});
#undef BC_PX
#undef BC_WT
        {
            // The tile is complete (branch-free, see conv_sf.hip).  The shared half is rounded to bf16 where the two-launch form
            // stores it, gets the bias and becomes the initial C of the chain's frames; behind the last frame it is cleared.
            const bool head = c_f == 0, last = c_f == T;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                accp[n] = acc[n];
#pragma unroll
                for (int r = 0; r < 16; ++r) pbv[n][r] = head ? bc_round(acc[n][r]) + bias_l : (last ? 0.f : pbv[n][r]);
            }
            ex0p = c_x0;
            ey0p = c_y0;
            eitemp = c_clip * T + (c_f - 1);
            pending = !head;
            c_f = n_f;
            c_clip = n_clip;
            c_y0 = n_y0;
            c_x0 = n_x0;
        }
        w_slice2_owed = w_replace;
        w_pk = nx_pk;
        asm volatile("" ::"v"(fence), "v"(fence_w));                // the next tile's halo and weight slices 0, 1 have landed
        BC_BARRIER();                                               // b2
    };
    for (int kt = 0; kt < nt_tiles; kt += 2) {
        tile_pass(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nt_tiles) tile_pass(std::integral_constant<int, 1>{}, kt + 1);
    }
    // ---- the last tile (a frame tile): both rows
    rows_setup();
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) quarter_request(n, q);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) quarter_finish(n, q);
#undef BC_DMA_HALO
#undef BC_DMA_W
#undef BC_TILE
#undef BC_BARRIER
}

hipError_t launch_conv3x3_bf16_chain(const ConvBf16ChainParams& p, hipStream_t s) {
    if (!p.in || !p.in2 || !p.wpack || !p.wpack2 || !p.bias || !p.out || p.items < 1 || p.H < 1 || p.W < 1 || p.T < 1 || p.items % p.T) return hipErrorInvalidValue;
    if ((long long)p.H * p.W * 128 >= 0x7fffffffLL) return hipErrorInvalidValue;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    static int ncu[64] = {};
    if (!ncu[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        ncu[dev] = prop.multiProcessorCount;
    }
    const int grid = ncu[dev] >= 8 ? ncu[dev] / 8 * 8 : 8;
    static bool attr_dev[64] = {};
    if (!attr_dev[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_dev[dev] = true;
    }
    hipLaunchKernelGGL(conv3x3_bf16_chain_kernel, dim3(grid), dim3(BC_THREADS), BC_LDS_BYTES, s, p);
    return hipGetLastError();
}

size_t conv3x3_bf16_chain_pack_halfs() { return BC_W_BYTES / 2; }

// HWIO [3,3,cin_total,64] rows [cin_begin, cin_begin+64) -> [kx][ks][ky][m][part][lane][e]:
// W[ky][kx][cin_begin + 16 (2 part + ks) + 8 (lane>>5) + e][32 m + (lane & 31)]   (the weights are the B operand: identity rows)
void conv3x3_bf16_chain_pack_weights(const float* hwio, int cin_total, int cin_begin, uint16_t* dst) {
    for (int kx = 0; kx < 3; ++kx)
        for (int ks = 0; ks < 2; ++ks)
            for (int ky = 0; ky < 3; ++ky)
                for (int m = 0; m < 2; ++m)
                    for (int part = 0; part < 2; ++part)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int ci = cin_begin + 16 * (2 * part + ks) + 8 * (lane >> 5) + e;
                                const int co = 32 * m + (lane & 31);
                                const size_t base = (((((size_t)(kx * 2 + ks) * 3 + ky) * 2 + m) * 2 + part) * 64 + lane) * 8 + e;
                                dst[base] = bf16_rne(hwio[((size_t)(ky * 3 + kx) * cin_total + ci) * 64 + co]);
                            }
}

}  // namespace pfnl

// Micro-benchmark (gfx950): what does ONE workgroup per CU (512 threads, all of LDS - the residency of the persistent 3x3 kernels) pull
// through its CU's memory port when it does NOTHING but the byte streams of a convolution tile?  Per tile (8 x 32 pixels):
//   H  halo read       10 x 34 pixels x PX bytes by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per wave instruction) into a double-buffered
//                      LDS tile, rows W_IMG pixels apart in memory (as a halo is); neighbouring tiles overlap by the halo ring (L2-shared)
//   R  residual read   8 x 32 x PX bytes, dword (or 16-byte) loads per lane, consumed by an add
//   S  output store    8 x 32 x PX bytes, dword (or 16-byte) stores
// with PX = 128 (bf16 trunk) or 256 (fp32 / split format), one closing barrier per tile and the DMA waited for with a fence load -
// the skeleton of conv3x3_sf_kernel / conv3x3_bf16_kernel with every MFMA, LDS operand read and epilogue instruction removed.
// The answer is the ceiling of the "memory side" of those launches: B per shader clock per CU and TB/s over 256 CUs, next to
// the 8 TB/s spec / 6.3 TB/s copy-kernel figure.  Tiles are dealt out XCD by XCD as the kernels do; the image set is > 1 GB (no
// Infinity-Cache residency).
//   hipcc --offload-arch=gfx950 -O3 -o cu_stream_mix cu_stream_mix.hip && ./cu_stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}

constexpr int TH = 8, TW = 32, IH = 10, IW = 34;

struct Params {
    const unsigned char* in;     // [items][H][W][PX]
    const unsigned char* res;    // [items][H][W][PX]
    unsigned char* out;          // [items][H][W][PX]
    int H, W, items;
    int do_h, do_r, do_s, wide;  // streams on / off; wide: 16-byte residual loads and stores instead of dwords
    long long* cycles;           // per workgroup: shader cycles of its loop
};

template <int PX>
__global__ __launch_bounds__(512, 1) void stream_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NPIX = IH * IW;
    constexpr int CH = 8;                                           // 16-byte chunks per 128-byte (half-)pixel: a UNIT's halo, as in the kernels
    constexpr int UNITS = PX / 128;                                 // 256 B per pixel: two units per tile (the two channel halves)
    constexpr int NDMA = (NPIX * CH + 63) / 64;                     // wave instructions per unit halo (43)
    constexpr int TILE_BYTES = NDMA * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;
    const int tiles_x = W / TW, tiles_y = H / TH, per_item = tiles_x * tiles_y;
    const int nt = per_item * p.items;
    const int xcd = blockIdx.x & 7, xj = blockIdx.x >> 3, cpx = gridDim.x >> 3;
    const int per_xcd = (nt + 7) >> 3;
    const int tbeg = xcd * per_xcd, tcnt = min(per_xcd, nt - tbeg);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const long long item_bytes = (long long)H * W * PX;
    float acc = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    int buf = 0;
    for (int k = xj; k < tcnt; k += cpx, buf ^= 1) {
        const int t = tbeg + k;
        const int item = t / per_item, sp = t - item * per_item;
        const int ty = sp / tiles_x, y0 = ty * TH, x0 = (sp - ty * tiles_x) * TW;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.in) + (size_t)item * item_bytes, 0, (int)item_bytes, 0x00020000);
        unsigned fence = 0;
        if (p.do_h) {
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
#pragma unroll
                for (int i0 = 0; i0 < (NDMA + 7) / 8; ++i0) {
                    const int i = wave + 8 * i0;
                    if (i < NDMA) {
                        const int piece = i * 64 + lane, pix = piece / CH, c = piece - pix * CH;
                        const int py = pix / IW, px = pix - py * IW;
                        const int gy = y0 + py - 1, gx = x0 + px - 1;
                        const bool in = pix < NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                        dma16(rs, lds0 + ((UNITS == 1 ? buf : u) * TILE_BYTES) + i * 1024, in ? (gy * W + gx) * PX + u * 128 + c * 16 : 0x7fffffff);
                    }
                }
            }
            fence = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
        }
        // residual + store: thread = (pixel, piece); dword mode: lane = channel pair ..., 32 lanes x 4 B per pixel line as in the kernels
        const size_t tile_org = (size_t)item * item_bytes + ((size_t)y0 * W + x0) * PX;
        // (every load of the tile is requested before the first is used, every store follows: the kernels request residual lines
        //  4 - 10 sub-steps ahead of their use as well)
        if (p.wide) {
            constexpr int CH = PX / 16;
            constexpr int NP = TH * TW * CH / 512;                  // 16-byte pieces per thread
            u32x4 v[NP];
            size_t o[NP];
#pragma unroll
            for (int i0 = 0; i0 < NP; ++i0) {
                const int piece = i0 * 512 + tid, pix = piece / CH, c = piece - pix * CH;
                o[i0] = tile_org + ((size_t)(pix / TW) * W + (pix % TW)) * PX + c * 16;
                v[i0] = u32x4{1, 2, 3, 4};
                if (p.do_r) v[i0] = *reinterpret_cast<const u32x4*>(p.res + o[i0]);
            }
#pragma unroll
            for (int i0 = 0; i0 < NP; ++i0) {
                v[i0].x += 1;
                if (p.do_s) *reinterpret_cast<u32x4*>(p.out + o[i0]) = v[i0];
                else acc += v[i0].x;
            }
        } else {
            constexpr int ND = TH * TW * PX / 4 / 512;              // dwords per thread (16 / 32)
            unsigned v[ND];
            size_t o[ND];
#pragma unroll
            for (int i0 = 0; i0 < ND; ++i0) {
                const int d = i0 * 512 + tid, pix = d / (PX / 4), c = d - pix * (PX / 4);
                o[i0] = tile_org + ((size_t)(pix / TW) * W + (pix % TW)) * PX + c * 4;
                v[i0] = 7;
                if (p.do_r) v[i0] = *reinterpret_cast<const unsigned*>(p.res + o[i0]);
            }
#pragma unroll
            for (int i0 = 0; i0 < ND; ++i0) {
                v[i0] += 1;
                if (p.do_s) *reinterpret_cast<unsigned*>(p.out + o[i0]) = v[i0];
                else acc += v[i0];
            }
        }
        asm volatile("" ::"v"(fence));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (p.do_h) acc += reinterpret_cast<const float*>(smem + (UNITS == 1 ? buf : 0) * TILE_BYTES)[tid];   // (the tile is read)
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) p.cycles[blockIdx.x] = t1 - t0;
    if (acc == 1.2345e30f) p.out[0] = 1;
}

template <int PX>
static void run(const char* name, int do_h, int do_r, int do_s, int wide, int H, int W, int items, unsigned char* in, unsigned char* res, unsigned char* out,
                long long* dcyc) {
    const int lds = 160 * 1024;                                     // all of it: one workgroup per CU
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<PX>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    Params p{in, res, out, H, W, items, do_h, do_r, do_s, wide, dcyc};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(stream_kernel<PX>, dim3(256), dim3(512), lds, 0, p);
    CHECK(hipDeviceSynchronize());
    const int reps = 5;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream_kernel<PX>, dim3(256), dim3(512), lds, 0, p);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<long long> cyc(256);
    CHECK(hipMemcpy(cyc.data(), dcyc, 256 * sizeof(long long), hipMemcpyDeviceToHost));
    double cavg = 0;
    for (long long c : cyc) cavg += c;
    cavg /= 256;
    const long long ntiles = (long long)(H / TH) * (W / TW) * items;
    const double tile_b = TH * TW * PX, halo_b = (double)IH * IW * PX;
    const double cu_bytes = (do_h ? halo_b : 0) + (do_r ? tile_b : 0) + (do_s ? tile_b : 0);      // through the CU's port, per tile
    const double hbm_bytes = (do_h ? tile_b : 0) + (do_r ? tile_b : 0) + (do_s ? tile_b : 0);     // compulsory (halo ring shared through L2)
    const double tiles_per_cu = ntiles / 256.0;
    printf("%-46s %7.3f ms  %6.0f cycles/tile  %5.2f B/clk/CU  CU-port %5.2f TB/s  compulsory HBM %5.2f TB/s  (clock %.2f GHz)\n", name, ms,
           cavg / tiles_per_cu, cu_bytes * tiles_per_cu / cavg, cu_bytes * ntiles / (ms * 1e-3) / 1e12, hbm_bytes * ntiles / (ms * 1e-3) / 1e12,
           cavg / (ms * 1e-3) / 1e9);
}

int main() {
    // 56 frames of 256 x 512 pixels: 7340032 pixels -> 0.94 GB per tensor at 128 B, 1.9 GB at 256 B; 28672 tiles = 112 per CU
    const int H = 256, W = 512, items = 56;
    const size_t bytes = (size_t)H * W * items * 256;
    unsigned char *in, *res, *out;
    long long* dcyc;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&res, bytes));
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMalloc(&dcyc, 256 * sizeof(long long)));
    CHECK(hipMemset(in, 1, bytes));
    CHECK(hipMemset(res, 2, bytes));
    CHECK(hipMemset(out, 0, bytes));
    printf("one 512-thread workgroup per CU, 8x32-pixel tiles, halo by LDS-DMA, residual / store per lane; %d tiles per CU\n", (H / TH) * (W / TW) * items / 256);
    printf("--- 128 B per pixel (bf16 trunk)\n");
    run<128>("halo only", 1, 0, 0, 0, H, W, items, in, res, out, dcyc);
    run<128>("store only (dwords)", 0, 0, 1, 0, H, W, items, in, res, out, dcyc);
    run<128>("store only (16 B)", 0, 0, 1, 1, H, W, items, in, res, out, dcyc);
    run<128>("halo + store (conv1_i), dwords", 1, 0, 1, 0, H, W, items, in, res, out, dcyc);
    run<128>("halo + store (conv1_i), 16 B", 1, 0, 1, 1, H, W, items, in, res, out, dcyc);
    run<128>("halo + residual + store (conv2_i), dwords", 1, 1, 1, 0, H, W, items, in, res, out, dcyc);
    run<128>("halo + residual + store (conv2_i), 16 B", 1, 1, 1, 1, H, W, items, in, res, out, dcyc);
    printf("--- 256 B per pixel (fp32 / split format: two units of 128 B halves per tile in the kernels; here one 256 B halo)\n");
    run<256>("halo only", 1, 0, 0, 0, H, W, items, in, res, out, dcyc);
    run<256>("halo + store (conv1_i), dwords", 1, 0, 1, 0, H, W, items, in, res, out, dcyc);
    run<256>("halo + store (conv1_i), 16 B", 1, 0, 1, 1, H, W, items, in, res, out, dcyc);
    run<256>("halo + residual + store (conv2_i), dwords", 1, 1, 1, 0, H, W, items, in, res, out, dcyc);
    run<256>("halo + residual + store (conv2_i), 16 B", 1, 1, 1, 1, H, W, items, in, res, out, dcyc);
    return 0;
}

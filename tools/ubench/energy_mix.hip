// Micro-benchmark (gfx950, round 6): what TIME do f16 MFMAs, LDS operand reads and HBM bytes cost a launch that sits on the package power cap?
// DESIGN.md R6.1 found that schedule work on the trunk kernels moves shader cycles and not time: the forward runs at the 1 400 W cap, and there
// time = energy / cap.  This program measures the terms of that sum directly.  One workgroup per CU (512 threads, 8 waves), every wave runs
//      per step:  L  global_load_dwordx4 (1 KB per wave instruction, its own stream of a > 2 GB buffer: nothing comes out of a cache)
//                 M  v_mfma_f32_32x32x16_f16 (4 accumulators in rotation, operands in registers)
//                 D  ds_read_b128 (operand-sized LDS reads, conflict-free, results consumed by a later MFMA)
//                 L  global_store_dwordx4 of what the previous step loaded (a copy: bytes = 2 L KB per wave and step)
// for `steps` steps; launches are repeated for ~0.6 s per point so that the power controller has settled, and the last launches are timed.
// Points: MFMA only; copy only; MFMA + bytes at the FLOP-per-byte ratios of the trunk launches (fp32: conv1_i + conv10_i 530, conv2_i 330;
// bf16: 297 and 183); MFMA + LDS reads at 0.5 and 1 ds_read_b128 per MFMA.  If time is energy, then t = a F + b B + c D with the SAME a, b, c at
// every point: the program prints a (from MFMA only), b and c from each mixed point, and the residual of the fit - and beside it what a
// "perfect overlap" model max(t_mfma, t_bytes) would have predicted.
//   hipcc --offload-arch=gfx950 -O3 -o energy_mix energy_mix.hip && ./energy_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Params {
    const u32x4* src;         // [waves][steps][L][64] 16-byte pieces
    u32x4* dst;
    float* sink;
    long long* clk;           // per workgroup: shader cycles, constant-clock ticks (100 MHz)
    int steps, wrap;          // wrap: a wave's stream restarts every `wrap` steps (footprint = waves x wrap x L KB each way: Infinity-Cache- or L2-resident)
    int rot;                  // operand rotation: 0 = as written (A changes every 4th MFMA, B every MFMA), 1 = the same A and B for every MFMA, 2 = both change every MFMA
    unsigned amask, bmask;    // and-masks on the operands' binary16 patterns (0xffff = data-like; fewer mantissa bits / zeros: does the power follow the data?)
};

template <int M, int L, int D, bool NT = true, int SHAPE = 32>
__global__ __launch_bounds__(512, 1) void mix_kernel(Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = blockIdx.x * 8 + (tid >> 6);
    half8 a[4], b[4];                                               // random mantissas and signs, magnitudes 0.5 .. 2 (data-like toggling: power depends on it)
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) {
            const unsigned h = (unsigned)(tid * 64 + k * 8 + i) * 2654435761u, g = h * 2246822519u + 12345u;
            a[k][i] = __builtin_bit_cast(_Float16, (unsigned short)((0x3800u + ((h >> 7) & 0x07ffu) + ((h >> 3) & 0x8000u)) & p.amask));
            b[k][i] = __builtin_bit_cast(_Float16, (unsigned short)((0x3800u + ((g >> 7) & 0x07ffu) + ((g >> 3) & 0x8000u)) & p.bmask));
        }
    float16v acc[4];
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v acc16[8];                                                   // SHAPE = 16: v_mfma_f32_16x16x32_f16, two per 32x32x16's worth of FLOPs
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 4; ++i) acc16[k][i] = 0.f;
    if (D) { for (int i = tid; i < 65536 / 4; i += 512) ((unsigned*)lds)[i] = 0x38003800u + (i * 2654435761u & 0x87ff87ffu); }
    __syncthreads();
    const long long per_wave = (long long)p.steps * L * 64;
    const u32x4* s = p.src + (long long)wave * per_wave + lane;
    u32x4* d = p.dst + (long long)wave * per_wave + lane;
    u32x4 cur[L ? L : 1], nxt[L ? L : 1];
    for (int j = 0; j < (L ? L : 1); ++j) cur[j] = u32x4{0, 0, 0, 0};
    const long long w0 = wall_clock64(), t0 = clock64();
    const unsigned char* lrd = lds + (tid >> 6) * 8192 + lane * 16;
    for (int st = 0; st < p.steps; ++st) {
#pragma unroll
        for (int j = 0; j < L; ++j) { const u32x4* q = s + ((long long)(st % p.wrap) * L + j) * 64; nxt[j] = NT ? __builtin_nontemporal_load(q) : *q; }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (D && (m * D) / M != ((m + 1) * D) / M) {          // D reads spread evenly over the M MFMAs of the step
                const half8 r = *(const half8*)(lrd + (((m * 1024) + st * 2048) & 7168));
                asm volatile("" :: "v"(r));
                if ((m & 7) == 7) b[(m >> 3) & 3] = r;              // (consumed now and then: the reads must really deliver)
            }
            if constexpr (SHAPE == 32) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[p.rot == 1 ? 0 : p.rot == 2 ? (m & 3) : ((m >> 2) & 3)], b[p.rot == 1 ? 0 : ((m + (m >> 4)) & 3)], acc[m & 3], 0, 0, 0);
            } else {
                acc16[(2 * m) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(m >> 2) & 3], b[(m + (m >> 4)) & 3], acc16[(2 * m) & 7], 0, 0, 0);
                acc16[(2 * m + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(m >> 1) & 3], b[(m + 1 + (m >> 4)) & 3], acc16[(2 * m + 1) & 7], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < L; ++j) {
            if (st) { u32x4* q = d + ((long long)((st - 1) % p.wrap) * L + j) * 64; if (NT) __builtin_nontemporal_store(cur[j], q); else *q = cur[j]; }
            cur[j] = nxt[j];
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float r = 0.f;
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) r += acc[k][i];
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 4; ++i) r += acc16[k][i];
    for (int j = 0; j < (L ? L : 1); ++j) r += (float)cur[j][0];
    if (r == 123.456f) p.sink[0] = r;
    if (tid == 0) { p.clk[2 * blockIdx.x] = t1 - t0; p.clk[2 * blockIdx.x + 1] = w1 - w0; }
}

__global__ void fill_kernel(unsigned* p, long long n) {              // data-like bytes (a constant buffer would toggle no wires)
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = (unsigned)i * 2654435761u ^ (unsigned)(i >> 7) * 2246822519u;
}

struct Point { const char* name; int M, L, D; double ms, ghz; double flops, bytes, reads; };

template <int M, int L, int D, bool NT = true, int SHAPE = 32>
static Point run(const char* name, Params p, int ncu, long long cap_pieces_per_wave) {
    // steps: ~1.5 ms per launch at the expected rates, within the buffer
    long long steps = 4096;
    if (L) steps = std::min<long long>(steps, cap_pieces_per_wave / (L * 64));
    p.steps = (int)steps;
    if (p.wrap <= 0 || p.wrap > steps) p.wrap = (int)steps;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto launch = [&]() { hipLaunchKernelGGL((mix_kernel<M, L, D, NT, SHAPE>), dim3(ncu), dim3(512), 0, 0, p); };
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float one = 0.f; CHECK(hipEventElapsedTime(&one, e0, e1));
    const int settle = std::max(8, (int)(600.f / one)), timed = std::max(4, settle / 4);
    for (int i = 0; i < settle; ++i) launch();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < timed; ++i) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= timed;
    std::vector<long long> clk(2 * ncu);
    CHECK(hipMemcpy(clk.data(), p.clk, clk.size() * 8, hipMemcpyDeviceToHost));
    double ghz = 0.;
    for (int i = 0; i < ncu; ++i) ghz += (double)clk[2 * i] / ((double)clk[2 * i + 1] * 10.0);   // cycles per ns (constant clock: 10 ns per tick)
    ghz /= ncu;
    Point pt{name, M, L, D, ms, ghz, 0, 0, 0};
    const double waves = 8.0 * ncu;
    pt.flops = waves * steps * M * 32768.0;
    pt.bytes = waves * (2.0 * steps - 1) * L * 1024.0;
    pt.reads = waves * steps * D;
    printf("%-44s M=%3d L=%d D=%3d  %8.3f ms  %7.1f TFLOP/s  %6.2f TB/s  shader clock %.2f GHz  (%.0f FLOP per byte)\n", name, M, L, D, ms,
           pt.flops / ms * 1e-9, pt.bytes / ms * 1e-9, ghz, L ? pt.flops / pt.bytes : 0.);
    fflush(stdout);
    return pt;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount / 8 * 8;
    const long long waves = 8LL * ncu;
    const long long cap_pieces_per_wave = (3LL << 30) / 16 / waves;          // 3 GB each way
    Params p{};
    p.amask = p.bmask = 0xffffu;
    CHECK(hipMalloc((void**)&p.src, waves * cap_pieces_per_wave * 16));
    CHECK(hipMalloc((void**)&p.dst, waves * cap_pieces_per_wave * 16));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned*)p.src, waves * cap_pieces_per_wave * 4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMalloc((void**)&p.sink, 64));
    CHECK(hipMalloc((void**)&p.clk, 2 * ncu * 8));
    printf("# %s, %d workgroups x 512 threads, one per CU\n", prop.name, ncu);
    std::vector<Point> pts;
    pts.push_back(run<32, 0, 0>("MFMA only", p, ncu, cap_pieces_per_wave));
    pts.push_back(run<0, 4, 0>("copy only", p, ncu, cap_pieces_per_wave));
    pts.push_back(run<64, 1, 0>("MFMA + bytes, 1024 FLOP per byte", p, ncu, cap_pieces_per_wave));
    pts.push_back(run<33, 1, 0>("MFMA + bytes, 528 (fp32 conv1_i + conv10_i)", p, ncu, cap_pieces_per_wave));
    pts.push_back(run<21, 1, 0>("MFMA + bytes, 336 (fp32 conv2_i)", p, ncu, cap_pieces_per_wave));
    pts.push_back(run<37, 2, 0>("MFMA + bytes, 296 (bf16 conv1_i + conv10_i)", p, ncu, cap_pieces_per_wave));
    pts.push_back(run<23, 2, 0>("MFMA + bytes, 184 (bf16 conv2_i per frame)", p, ncu, cap_pieces_per_wave));
    pts.push_back(run<8, 1, 0>("MFMA + bytes, 128", p, ncu, cap_pieces_per_wave));
    pts.push_back(run<32, 0, 16>("MFMA + 0.5 ds_read_b128 per MFMA", p, ncu, cap_pieces_per_wave));
    pts.push_back(run<32, 0, 32>("MFMA + 1 ds_read_b128 per MFMA", p, ncu, cap_pieces_per_wave));
    pts.push_back(run<33, 1, 16>("MFMA + bytes 528 + 0.5 ds_read_b128 per MFMA", p, ncu, cap_pieces_per_wave));
    pts.push_back(run<32, 0, 0>("MFMA only (again)", p, ncu, cap_pieces_per_wave));
    // the fit
    const double a = 0.5 * (pts[0].ms / pts[0].flops + pts.back().ms / pts.back().flops);      // ms per FLOP
    const double bw = pts[1].bytes / pts[1].ms;                                                    // bytes per ms with nothing else to do
    printf("\n# t = a F + b B + c D:  a = %.4f ms per TFLOP (MFMA only = %.0f TFLOP/s);  the copy alone moves %.2f TB/s\n", a * 1e12, 1e-9 / a, bw * 1e-9);
    double c = 0.;
    for (auto& q : pts) if (q.M && !q.L && q.D) c = (q.ms - a * q.flops) / q.reads;              // (the last such point: 1 read per MFMA)
    printf("# c = %.3e ms per ds_read_b128 wave instruction (1 KB) = %.3f of an MFMA's time each\n", c, c / (a * 32768.0));
    for (auto& q : pts) {
        if (!q.M || !q.L) continue;
        const double rest = q.ms - a * q.flops - c * q.reads;
        const double b = rest / q.bytes;
        const double overlap = std::max(a * q.flops, q.bytes / bw);
        printf("%-44s b = %6.1f us per GB  (an HBM byte costs the time of %5.1f MFMA FLOPs);  measured %.3f ms, perfect-overlap model %.3f ms (%.2fx)\n",
               q.name, b * 1e12, b / a, q.ms, overlap, q.ms / overlap);
    }
    { const Point m3 = run<33, 1, 0, false>("MFMA + bytes 528, plain loads / stores", p, ncu, cap_pieces_per_wave);
      printf("    b = %.1f us per GB\n", (m3.ms - a * m3.flops) / m3.bytes * 1e12); }
    for (int wrap : {64, 8, 2}) {   // the same mixes out of the Infinity Cache (256 MB) / out of L2 (8 x 4 MB): what does a byte cost when it does not come from HBM?
        Params q = p; q.wrap = wrap;
        printf("# streams of %d steps: footprint %.0f MB each way at L = 1\n", wrap, waves * wrap * 1024.0 / 1e6);
        run<0, 4, 0>("copy only, small footprint (x4)", q, ncu, cap_pieces_per_wave);
        const Point m = run<33, 1, 0>("MFMA + bytes 528, small footprint", q, ncu, cap_pieces_per_wave);
        printf("    b = %.1f us per GB\n", (m.ms - a * m.flops) / m.bytes * 1e12);
        const Point m2 = run<21, 1, 0>("MFMA + bytes 336, small footprint", q, ncu, cap_pieces_per_wave);
        printf("    b = %.1f us per GB\n", (m2.ms - a * m2.flops) / m2.bytes * 1e12);
        const Point m3 = run<33, 1, 0, false>("MFMA + bytes 528, small footprint, plain ld/st", q, ncu, cap_pieces_per_wave);
        printf("    b = %.1f us per GB\n", (m3.ms - a * m3.flops) / m3.bytes * 1e12);
    }
    {   // the other f16 shape: v_mfma_f32_16x16x32_f16 (two per 32x32x16's worth of FLOPs; tools/ubench/mfma_shape_power.hip) at the same mixes
        printf("# v_mfma_f32_16x16x32_f16 (M counts pairs: the same FLOPs per step)\n");
        run<32, 0, 0, true, 16>("16x16x32: MFMA only", p, ncu, cap_pieces_per_wave);
        run<33, 1, 0, true, 16>("16x16x32: MFMA + bytes 528", p, ncu, cap_pieces_per_wave);
        run<21, 1, 0, true, 16>("16x16x32: MFMA + bytes 336", p, ncu, cap_pieces_per_wave);
        run<32, 0, 16, true, 16>("16x16x32: MFMA + 0.5 ds_read_b128 per 32 KFLOP", p, ncu, cap_pieces_per_wave);
        run<33, 1, 16, true, 16>("16x16x32: bytes 528 + 0.5 ds_read_b128", p, ncu, cap_pieces_per_wave);
        run<37, 2, 0, true, 16>("16x16x32: MFMA + bytes 296 (bf16 mix)", p, ncu, cap_pieces_per_wave);
        run<23, 2, 0, true, 16>("16x16x32: MFMA + bytes 184 (bf16 mix)", p, ncu, cap_pieces_per_wave);
    }
    {   // does the matrix pipe's power follow its operands?  (not part of the fit)
        Params q = p;
        q.rot = 1; run<32, 0, 0>("MFMA only, the SAME A and B every MFMA", q, ncu, cap_pieces_per_wave);
        q.rot = 2; run<32, 0, 0>("MFMA only, A and B change every MFMA", q, ncu, cap_pieces_per_wave);
        q.rot = 0;
        q.amask = q.bmask = 0xffc0u; run<32, 0, 0>("MFMA only, 5 mantissa bits in A and B", q, ncu, cap_pieces_per_wave);
        q.amask = 0xffffu; q.bmask = 0xffc0u; run<32, 0, 0>("MFMA only, 5 mantissa bits in B", q, ncu, cap_pieces_per_wave);
        q.amask = q.bmask = 0xfc00u; run<32, 0, 0>("MFMA only, powers of two in A and B", q, ncu, cap_pieces_per_wave);
        q.amask = 0xffffu; q.bmask = 0u; run<32, 0, 0>("MFMA only, B = 0", q, ncu, cap_pieces_per_wave);
        q.amask = q.bmask = 0u; run<32, 0, 0>("MFMA only, A = B = 0", q, ncu, cap_pieces_per_wave);
    }
    return 0;
}

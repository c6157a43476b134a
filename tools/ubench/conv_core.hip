// Micro-benchmark (gfx950): the MFMA + operand-read core of the split-f16 3x3 kernels (conv_split16.hip / conv_sf.hip) in
// isolation - 8 waves per workgroup (2 per SIMD), one workgroup per CU, operands from LDS with the kernels' addresses,
// 18 sub-steps of 6 v_mfma_f32_32x32x16_f16 per unit, nothing else (no global memory, no epilogue).  Question: how many
// cycles does a unit of 216 MFMAs per SIMD (6 912 cycles of matrix pipe) take under different orders of the SAME instructions?
//   V0  the kernels' order: [operand reads of the next sub-step] fence [6 MFMAs] fence
//   V1  reads spread between the MFMAs: M r r M r r M r r M M M (every position pinned with sched_barrier)
//   V2  V0 with s_setprio 1 around the MFMA cluster
//   V3  V1 with the younger half of the workgroup (waves 4-7) at s_setprio 1
//   V4  reads of sub-step S+2 (two ahead) spread between the MFMAs of S
//   V5  V0 with waves 4-7 started half a sub-step late (de-phased partners)
//   V6  NO operand reads in the loop (the first sub-step's operands, over and over): what the matrix pipe alone sustains under the
//       power cap - the difference to V0 / V1 is what the 14 ds_read_b128 per 18 MFMAs cost
//   core4: the same products with FOUR waves per workgroup (one per SIMD, up to 512 registers each): a wave owns 4 output rows x 32
//       pixels x 32 channels (128 accumulator registers), 12 MFMAs per sub-step, 18 ds_read_b128 per 36 MFMAs (0.5 per MFMA
//       instead of 0.78) - the operand-traffic diet a bigger register tile would buy
// BAR: 0 none, 1 one workgroup barrier per group of 3 sub-steps (the kernels have 3 - 5 per 18)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int IW = 34, TILE_BYTES = 10 * IW * 128, W_BYTES = 73728;

__device__ __forceinline__ f32x16 mf(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int V, int BAR>
__global__ __launch_bounds__(512, 1) void core(const unsigned* init, float* out, long long* cyc, int units) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (2 * TILE_BYTES + W_BYTES) / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = init[i];
    __syncthreads();
    const int rp = wave >> 1, mt = wave & 1;
    int paddr[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = (lane & 31) + kx;
        paddr[kx] = ((2 * rp) * IW + col) * 128 + ((((lane >> 5)) ^ ((col >> 1) & 7)) << 4);
    }
    const int lo_xor = 4 << 4;
    const unsigned char* const wlane = smem + 2 * TILE_BYTES + mt * 2048 + lane * 16;
    f32x16 accm[2], accc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accm[n][r] = 0.f;
            accc[n][r] = 0.f;
        }
    if constexpr (V == 3) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }
    if constexpr (V == 5) {
        if (wave >= 4) __builtin_amdgcn_s_sleep(2);                 // ~128 cycles
    }
    const long long w0 = wall_clock64();                            // constant 100 MHz
    const long long t0 = clock64();                                 // shader clock
    for (int u = 0; u < units; ++u) {
        const unsigned char* const tile = smem + (u & 1) * TILE_BYTES;
        h8 X[4][2], Wv[3][2];
#define PX(g_, r_, part_) (*reinterpret_cast<const h8*>(tile + (paddr[(g_) >> 1] ^ (((part_) ? lo_xor : 0) | (((g_) & 1) << 5))) + (r_) * (IW * 128)))
#define WT(g_, ky_, part_) (*reinterpret_cast<const h8*>(wlane + (((g_) * 3 + (ky_)) << 12) + ((part_) << 10)))
        X[0][0] = PX(0, 0, 0);
        X[0][1] = PX(0, 0, 1);
        X[1][0] = PX(0, 1, 0);
        X[1][1] = PX(0, 1, 1);
        Wv[0][0] = WT(0, 0, 0);
        Wv[0][1] = WT(0, 0, 1);
        if constexpr (V == 4) {                                     // two ahead: sub-step 1's operands as well
            Wv[1][0] = WT(0, 1, 0);
            Wv[1][1] = WT(0, 1, 1);
            X[2][0] = PX(0, 2, 0);
            X[2][1] = PX(0, 2, 1);
        }
        FENCE();
        auto substep = [&](auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value;
            constexpr int ky = S % 3;
            constexpr int D = V == 4 ? 2 : 1;                       // read distance in sub-steps
            constexpr int S1 = S + D, g1 = S1 / 3, ky1 = S1 % 3;
            constexpr bool RD = S1 < 18;
            constexpr int WB = V == 4 ? 3 : 2;                      // weight operand ring
            if constexpr (BAR == 1 && ky == 0 && S > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            const h8 wh = Wv[S % WB][0], wo = Wv[S % WB][1];
            if constexpr (V == 6) {
                accm[0] = mf(X[0][0], Wv[0][0], accm[0]);
                accm[1] = mf(X[1][0], Wv[0][0], accm[1]);
                accc[0] = mf(X[0][0], Wv[0][1], accc[0]);
                accc[1] = mf(X[1][0], Wv[0][1], accc[1]);
                accc[0] = mf(X[0][1], Wv[0][0], accc[0]);
                accc[1] = mf(X[1][1], Wv[0][0], accc[1]);
                FENCE();
            } else if constexpr (V == 0 || V == 2 || V == 5) {
                if constexpr (RD) {
                    Wv[S1 % WB][0] = WT(g1, ky1, 0);
                    Wv[S1 % WB][1] = WT(g1, ky1, 1);
                    if constexpr (ky1 == 0) {
                        X[0][0] = PX(g1, 0, 0);
                        X[0][1] = PX(g1, 0, 1);
                        X[1][0] = PX(g1, 1, 0);
                        X[1][1] = PX(g1, 1, 1);
                    } else {
                        X[ky1 + 1][0] = PX(g1, ky1 + 1, 0);
                        X[ky1 + 1][1] = PX(g1, ky1 + 1, 1);
                    }
                }
                FENCE();
                if constexpr (V == 2) __builtin_amdgcn_s_setprio(1);
                accm[0] = mf(X[ky][0], wh, accm[0]);
                accm[1] = mf(X[ky + 1][0], wh, accm[1]);
                accc[0] = mf(X[ky][0], wo, accc[0]);
                accc[1] = mf(X[ky + 1][0], wo, accc[1]);
                accc[0] = mf(X[ky][1], wh, accc[0]);
                accc[1] = mf(X[ky + 1][1], wh, accc[1]);
                if constexpr (V == 2) __builtin_amdgcn_s_setprio(0);
                FENCE();
            } else {
                // M r r M r r M r r M M M: weights first, then the pixel rows
                accm[0] = mf(X[ky][0], wh, accm[0]);
                FENCE();
                if constexpr (RD) {
                    Wv[S1 % WB][0] = WT(g1, ky1, 0);
                    Wv[S1 % WB][1] = WT(g1, ky1, 1);
                }
                FENCE();
                accm[1] = mf(X[ky + 1][0], wh, accm[1]);
                FENCE();
                if constexpr (RD) {
                    if constexpr (ky1 == 0) {
                        // (V4: X[0], X[1] of the next group are still in use by sub-step S when ky == 1: read them into the ring later)
                        X[0][0] = PX(g1, 0, 0);
                        X[0][1] = PX(g1, 0, 1);
                    } else {
                        X[ky1 + 1][0] = PX(g1, ky1 + 1, 0);
                        X[ky1 + 1][1] = PX(g1, ky1 + 1, 1);
                    }
                }
                FENCE();
                accc[0] = mf(X[ky][0], wo, accc[0]);
                FENCE();
                if constexpr (RD && ky1 == 0) {
                    X[1][0] = PX(g1, 1, 0);
                    X[1][1] = PX(g1, 1, 1);
                }
                FENCE();
                accc[1] = mf(X[ky + 1][0], wo, accc[1]);
                accc[0] = mf(X[ky][1], wh, accc[0]);
                accc[1] = mf(X[ky + 1][1], wh, accc[1]);
                FENCE();
            }
        };
#define SS(n_) substep(std::integral_constant<int, n_>{})
        SS(0); SS(1); SS(2); SS(3); SS(4); SS(5); SS(6); SS(7); SS(8); SS(9); SS(10); SS(11); SS(12); SS(13); SS(14); SS(15); SS(16); SS(17);
#undef SS
#undef PX
#undef WT
        if constexpr (BAR == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (tid == 0) cyc[2048 + blockIdx.x] = w1 - w0;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += accm[0][r] + accm[1][r] + accc[0][r] + accc[1][r];
    out[blockIdx.x * 512 + tid] = s;
}

// four waves per workgroup, one per SIMD: wave = (row quad rq, channel tile mt); reads of the next sub-step spread between the MFMAs
template <int BAR>
__global__ __launch_bounds__(256, 1) void core4(const unsigned* init, float* out, long long* cyc, int units) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (2 * TILE_BYTES + W_BYTES) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = init[i];
    __syncthreads();
    const int rq = wave >> 1, mt = wave & 1;
    int paddr[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = (lane & 31) + kx;
        paddr[kx] = ((4 * rq) * IW + col) * 128 + ((((lane >> 5)) ^ ((col >> 1) & 7)) << 4);
    }
    const int lo_xor = 4 << 4;
    const unsigned char* const wlane = smem + 2 * TILE_BYTES + mt * 2048 + lane * 16;
    f32x16 accm[4], accc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accm[n][r] = 0.f;
            accc[n][r] = 0.f;
        }
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int u = 0; u < units; ++u) {
        const unsigned char* const tile = smem + (u & 1) * TILE_BYTES;
        h8 X[6][2], Wv[2][2];
#define PX(g_, r_, part_) (*reinterpret_cast<const h8*>(tile + (paddr[(g_) >> 1] ^ (((part_) ? lo_xor : 0) | (((g_) & 1) << 5))) + (r_) * (IW * 128)))
#define WT(g_, ky_, part_) (*reinterpret_cast<const h8*>(wlane + (((g_) * 3 + (ky_)) << 12) + ((part_) << 10)))
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            X[r][0] = PX(0, r, 0);
            X[r][1] = PX(0, r, 1);
        }
        Wv[0][0] = WT(0, 0, 0);
        Wv[0][1] = WT(0, 0, 1);
        FENCE();
        auto substep = [&](auto sc) __attribute__((always_inline)) {
            constexpr int S = decltype(sc)::value;
            constexpr int ky = S % 3;
            constexpr int S1 = S + 1, g1 = S1 / 3, ky1 = S1 % 3;
            constexpr bool RD = S1 < 18;
            if constexpr (BAR == 1 && ky == 0 && S > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const h8 wh = Wv[S & 1][0], wo = Wv[S & 1][1];
            // 12 MFMAs; the next sub-step's reads (2 weights; 8 pixel reads when a new group starts, else 2) between them
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                accm[n] = mf(X[ky + n][0], wh, accm[n]);
                FENCE();
                if constexpr (RD) {
                    if (n == 0) {
                        Wv[S1 & 1][0] = WT(g1, ky1, 0);
                        Wv[S1 & 1][1] = WT(g1, ky1, 1);
                    }
                    if constexpr (ky1 != 0) {
                        if (n == 1) {
                            X[ky1 + 3][0] = PX(g1, ky1 + 3, 0);
                            X[ky1 + 3][1] = PX(g1, ky1 + 3, 1);
                        }
                    }
                }
                FENCE();
                accc[n] = mf(X[ky + n][0], wo, accc[n]);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                accc[n] = mf(X[ky + n][1], wh, accc[n]);
                FENCE();
                if constexpr (RD && ky1 == 0) {                      // rows n of the next group: free once this loop is past them (ky == 2: rows 2..5 in use)
                    if (n >= 2) {
                        X[n - 2][0] = PX(g1, n - 2, 0);
                        X[n - 2][1] = PX(g1, n - 2, 1);
                    }
                }
                FENCE();
            }
            if constexpr (RD && ky1 == 0) {
                X[2][0] = PX(g1, 2, 0);
                X[2][1] = PX(g1, 2, 1);
                X[3][0] = PX(g1, 3, 0);
                X[3][1] = PX(g1, 3, 1);
                FENCE();
            }
        };
#define SS(n_) substep(std::integral_constant<int, n_>{})
        SS(0); SS(1); SS(2); SS(3); SS(4); SS(5); SS(6); SS(7); SS(8); SS(9); SS(10); SS(11); SS(12); SS(13); SS(14); SS(15); SS(16); SS(17);
#undef SS
#undef PX
#undef WT
        if constexpr (BAR == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (tid == 0) cyc[2048 + blockIdx.x] = w1 - w0;
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += accm[n][r] + accc[n][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int BAR>
void run4(const char* name, const unsigned* dinit, float* dout, long long* dcyc, int units) {
    const int LDS = 2 * TILE_BYTES + W_BYTES;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(core4<BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    std::vector<long long> c(256 * 9);
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((core4<BAR>), dim3(256), dim3(256), LDS, 0, dinit, dout, dcyc, units);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost));
        double s03 = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 4; ++w) s03 += c[b * 8 + w];
        double wt = 0;
        for (int b = 0; b < 256; ++b) wt += c[2048 + b];
        const double us_unit = wt / 256 / 100.0 / units;
        if (rep == 1)
            printf("%-34s bar %d: %7.0f cycles per unit = %5.1f per MFMA of the SIMD's 216; %.2f us per unit in-kernel => shader clock %.2f GHz, "
                   "%.0f TFLOP/s executed (f16 MFMA); launch %.2f us per unit\n",
                   name, BAR, s03 / 1024 / units, s03 / 1024 / units / 216, us_unit, s03 / 1024 / units / us_unit * 1e-3,
                   256.0 * 4 * 216 * 32768 / us_unit * 1e-6, ms * 1e3 / units);
    }
}

template <int V, int BAR>
void run(const char* name, const unsigned* dinit, float* dout, long long* dcyc, int units) {
    const int LDS = 2 * TILE_BYTES + W_BYTES;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(core<V, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    std::vector<long long> c(256 * 9);
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((core<V, BAR>), dim3(256), dim3(512), LDS, 0, dinit, dout, dcyc, units);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost));
        double s03 = 0, s47 = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? s03 : s47) += c[b * 8 + w];
        double wt = 0;
        for (int b = 0; b < 256; ++b) wt += c[2048 + b];
        const double us_unit = wt / 256 / 100.0 / units;            // 100 MHz ticks -> us
        if (rep == 1)
            printf("%-34s bar %d: %7.0f / %7.0f cycles per unit (waves 0-3 / 4-7) = %5.1f / %5.1f per MFMA of the SIMD's 216; %.2f us per unit in-kernel "
                   "=> shader clock %.2f GHz, %.0f TFLOP/s executed (f16 MFMA); launch %.2f us per unit\n",
                   name, BAR, s03 / 1024 / units, s47 / 1024 / units, s03 / 1024 / units / 216, s47 / 1024 / units / 216, us_unit,
                   s47 / 1024 / units / us_unit * 1e-3, 256.0 * 4 * 216 * 32768 / us_unit * 1e-6, ms * 1e3 / units);
    }
}

int main() {
    const size_t words = (2 * TILE_BYTES + W_BYTES) / 4;
    std::vector<unsigned> h(words);
    unsigned x = 12345;
    for (auto& w : h) {                                             // two binary16 values of magnitude ~0.1 - 1 per word (realistic toggling)
        x = x * 1664525u + 1013904223u;
        const unsigned a = 0x3000u + ((x >> 8) & 0x0fffu) + ((x >> 3) & 0x8000u);
        const unsigned b = 0x3000u + ((x >> 20) & 0x0fffu) + ((x << 2) & 0x8000u);
        w = a | (b << 16);
    }
    unsigned* dinit;
    float* dout;
    long long* dcyc;
    CHECK(hipMalloc(&dinit, words * 4));
    CHECK(hipMalloc(&dout, 256 * 512 * 4));
    CHECK(hipMalloc(&dcyc, 256 * 9 * 8));
    CHECK(hipMemcpy(dinit, h.data(), words * 4, hipMemcpyHostToDevice));
    const int units = 256;                                          // ~1.3 ms per launch: long enough for the clock to settle
    run<0, 0>("V0 kernels' order", dinit, dout, dcyc, units);
    run<0, 1>("V0 kernels' order", dinit, dout, dcyc, units);
    run<1, 0>("V1 reads between MFMAs", dinit, dout, dcyc, units);
    run<1, 1>("V1 reads between MFMAs", dinit, dout, dcyc, units);
    run<2, 1>("V2 V0 + setprio around MFMAs", dinit, dout, dcyc, units);
    run<3, 1>("V3 V1 + waves 4-7 at prio 1", dinit, dout, dcyc, units);
    run<5, 0>("V5 V0 + waves 4-7 delayed", dinit, dout, dcyc, units);
    run<5, 1>("V5 V0 + waves 4-7 delayed", dinit, dout, dcyc, units);
    run<6, 0>("V6 MFMAs only (no operand reads)", dinit, dout, dcyc, units);
    run<6, 1>("V6 MFMAs only (no operand reads)", dinit, dout, dcyc, units);
    run4<0>("core4: 4 waves, 4 rows per wave", dinit, dout, dcyc, units);
    run4<1>("core4: 4 waves, 4 rows per wave", dinit, dout, dcyc, units);
    return 0;
}

// Non-local block on bf16 MFMA (option precision=bf16; BASELINE.json configs[3]: at 1080p the affinity is
// N = 32400 squared, 354 GFLOP - 3.3 ms on the f32 matrix pipe, more than the whole bf16 trunk).
//
// Same streaming-softmax structure as nonlocal.hip (reference utils.py:18-71, nltype=1), with the two contractions on
// v_mfma_f32_32x32x16_bf16 and fp32 everywhere a bf16 value would be visible in the result:
//   * logits S = X X^T (|S| <= 84, exp(S) needs ~1e-4 absolute): bf16 inputs alone would be wrong by ~16 %
//     (SURVEY.md section 7), so X is split into hi + lo bf16 parts and S = hi hi + hi lo + lo hi accumulated in fp32
//     (the dropped lo lo term is < 84 * 2^-18): 18 MFMAs of 32 cycles per 32x32 tile against 42 f32 MFMAs of 64;
//   * P = exp2(S' - running max) in fp32, rounded to bf16 only as the MFMA operand; the row sum accumulates the SAME
//     rounded values through the "ones" channel, so the normalisation is exact for what was summed;
//   * V = X also as hi + lo (12 MFMAs per tile): a query dominated by one key returns that key's fp32 value;
//   * running max / rescale, normalisation, the folded 1x1 projection (f32 MFMA) and the residual as in nonlocal.hip.
// Operand layouts (lane = (l & 31, kh = l >> 5), 8 bf16 per lane and MFMA):
//   K tile in LDS  [key][96 ch] (+ pad to 208 B: conflict-free b128 reads), hi and lo: A of S^T = K Q^T;
//   Q in registers [6 k-steps] hi and lo, pre-scaled by log2(e): B of S^T;
//   P^T straight from the S^T accumulator: register r of lane (query, kh) is key (r&3) + 8(r>>2) + 4kh, registers
//   8t..8t+7 form the B operand of k-step t - the contraction order over keys is free, so V^T is stored by nl_pack_bf16
//   with the keys of every 32-block permuted to exactly that order ([ch][block][t][kh][e]).
#include "common.h"
#include "conv_bf16.h"

namespace pfnl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NB_KT = 64;                  // keys per LDS tile
constexpr int NB_KROW = 208;               // bytes per key row of the K tiles (96 ch * 2 B + 16)
constexpr int NB_VROW = 144;               // bytes per channel row of the V^T tiles (64 keys * 2 B + 16)
constexpr int NB_CP = 96;
constexpr int NB_THREADS = 512;           // 8 waves x 32 queries share every key tile: the K / V^T stream (48 KB per 64 keys) is what
                                          // bounds this kernel - with 128 queries per workgroup it ran at 6.6 TB/s of L2 -> CU traffic
                                          // (11 B/clk/CU) and 43 % matrix-pipe use, whatever was done to its instruction schedule
constexpr int NB_QB = NB_THREADS / 2;     // queries per workgroup
constexpr int NB_TILE_BYTES = 2 * NB_KT * NB_KROW + 2 * NB_CP * NB_VROW;   // 54 272: K hi, K lo, V^T hi, V^T lo
constexpr int NB_LDS_BYTES = 3 * NB_TILE_BYTES;                             // 162 816 of 163 840: tiles t-1 (late waves' P V), t, t+1 (being filled)

__device__ __forceinline__ unsigned short bf16_bits(float f) {   // round to nearest even
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_float(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }

// X [B][N][CP] fp32 (nl_pack_kernel) -> Khi, Klo [B][N][96] bf16;  Vthi, Vtlo [B][96][Npad] bf16, keys permuted per
// 32-block, channel C = 1 (the row-sum channel), channels > C = 0
__global__ void nl_pack_bf16_kernel(const float* __restrict__ X, uint16_t* __restrict__ Khi, uint16_t* __restrict__ Klo,
                                    uint16_t* __restrict__ Vthi, uint16_t* __restrict__ Vtlo, int B, int N, int Npad, int C,
                                    int CPin) {
    const size_t total = (size_t)B * Npad * NB_CP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % NB_CP);
        const int n = (int)((i / NB_CP) % Npad);
        const int b = (int)(i / ((size_t)NB_CP * Npad));
        const float v = (n < N && c < C) ? X[((size_t)b * N + n) * CPin + c] : 0.f;
        const unsigned short hi = bf16_bits(v);
        const unsigned short lo = bf16_bits(v - bf16_float(hi));
        if (n < N) {
            Khi[((size_t)b * N + n) * NB_CP + c] = hi;
            Klo[((size_t)b * N + n) * NB_CP + c] = lo;
        }
        // position of key n inside its 32-block: key = (e&3) + 8(2t + (e>>2)) + 4kh  ->  pos = 16t + 8kh + e
        const int kb = n & 31;
        const int e = (kb & 3) | (((kb >> 3) & 1) << 2), kh = (kb >> 2) & 1, t = kb >> 4;
        const size_t vp = ((size_t)b * NB_CP + c) * Npad + (n & ~31) + 16 * t + 8 * kh + e;
        Vthi[vp] = c == C ? (unsigned short)0x3f80 : hi;             // 1.0
        Vtlo[vp] = c == C ? (unsigned short)0 : lo;
    }
}

template <int C>
__global__ __launch_bounds__(NB_THREADS, 2) void nl_attn_bf16_kernel(const float* __restrict__ X, const uint16_t* __restrict__ Khi,
                                                              const uint16_t* __restrict__ Klo, const uint16_t* __restrict__ Vthi,
                                                              const uint16_t* __restrict__ Vtlo, float* __restrict__ Xo,
                                                              const float* __restrict__ Wp, const float* __restrict__ bp,
                                                              float* __restrict__ Zp, float* __restrict__ ML, int N, int Npad, int q0, int q1) {
    constexpr int CT = 3;
    constexpr int CP = (C + 31) / 32 * 32;                          // row stride of X / Xo / Wp (nl_padded_ch)
    static_assert(C < NB_CP && C % 2 == 0, "needs a pad channel inside 96");
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];   // three tiles: K hi | K lo | V^T hi | V^T lo

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int xl = lane & 31;
    const int kh = lane >> 5;
    const int b = blockIdx.y;
    const float* Xb = X + (size_t)b * N * CP;
    float* Xob = Xo + (size_t)b * N * CP;
    const int q = q0 + blockIdx.x * NB_QB + wave * 32 + xl;         // this lane's query (queries [q0, q1): a strip of the frame)
    const int qc = q < q1 ? q : q1 - 1;

    // B operand of S^T = K Q^T: this lane's query, channels 16ks + 8kh .. +7, scaled by log2(e), split hi + lo
    constexpr float LOG2E = 1.4426950408889634f;
    bf16x8 qh[6], ql[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 16 * ks + 8 * kh + e;
            const float v = c < C ? Xb[(size_t)qc * CP + c] * LOG2E : 0.f;
            const __bf16 h = (__bf16)v;
            qh[ks][e] = h;
            ql[ks][e] = (__bf16)(v - (float)h);
        }
    constexpr int LCT = C / 32, LI = C % 32;                        // where the row-sum channel C lives in the D layout
    constexpr int LKH = (LI % 8) >= 4 ? 1 : 0, LR = (LI / 8) * 4 + (LI % 8) % 4;

    f32x16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m = -INFINITY;

    // staging: 4 x 768 16-byte pieces per 64-key tile, 6 per thread (8 slots: the second pass covers pieces 512..767)
    const uint16_t* const Khb = Khi + (size_t)b * N * NB_CP;
    const uint16_t* const Klb = Klo + (size_t)b * N * NB_CP;
    const uint16_t* const Vhb = Vthi + (size_t)b * NB_CP * Npad;
    const uint16_t* const Vlb = Vtlo + (size_t)b * NB_CP * Npad;
    constexpr int NI = 2;
    u32x4 rk[4 * NI];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int id = min(tid + i * NB_THREADS, 767);          // 0..767 (surplus threads redo the last piece)
            const int key = id / 12, c16 = id - key * 12;
            const bool ok = k0 + key < N;
            const size_t ko = ((size_t)(k0 + (ok ? key : 0)) * NB_CP + c16 * 8);
            rk[i] = ok ? *reinterpret_cast<const u32x4*>(Khb + ko) : u32x4{0, 0, 0, 0};
            rk[NI + i] = ok ? *reinterpret_cast<const u32x4*>(Klb + ko) : u32x4{0, 0, 0, 0};
            const int ch = id >> 3, kc = id & 7;                    // V^T: 96 rows x 8 pieces (k0 + 64 <= Npad + 32: rows are padded)
            const bool vok = k0 + kc * 8 < Npad;
            const size_t vo = (size_t)ch * Npad + k0 + (vok ? kc * 8 : 0);
            rk[2 * NI + i] = vok ? *reinterpret_cast<const u32x4*>(Vhb + vo) : u32x4{0, 0, 0, 0};
            rk[3 * NI + i] = vok ? *reinterpret_cast<const u32x4*>(Vlb + vo) : u32x4{0, 0, 0, 0};
        }
    };
    auto store_tile = [&](unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int id = min(tid + i * NB_THREADS, 767);
            const int key = id / 12, c16 = id - key * 12;
            *reinterpret_cast<u32x4*>(buf + key * NB_KROW + c16 * 16) = rk[i];
            *reinterpret_cast<u32x4*>(buf + NB_KT * NB_KROW + key * NB_KROW + c16 * 16) = rk[NI + i];
            const int ch = id >> 3, kc = id & 7;
            *reinterpret_cast<u32x4*>(buf + 2 * NB_KT * NB_KROW + ch * NB_VROW + kc * 16) = rk[2 * NI + i];
            *reinterpret_cast<u32x4*>(buf + 2 * NB_KT * NB_KROW + NB_CP * NB_VROW + ch * NB_VROW + kc * 16) = rk[3 * NI + i];
        }
    };

    const int ntiles = (N + NB_KT - 1) / NB_KT;
    const int ksp = gridDim.z, sp = blockIdx.z;
    const int kt0 = (int)((long long)ntiles * sp / ksp), kt1 = (int)((long long)ntiles * (sp + 1) / ksp);
    load_tile(kt0 * NB_KT);
    store_tile(sm);
    if (kt0 + 1 < kt1) load_tile((kt0 + 1) * NB_KT);
    __syncthreads();
    // The two waves of a SIMD are half a tile apart: waves 0-3 run S^T, softmax, P V of tile t; waves 4-7 run P V of
    // tile t-1 (its P^T kept in registers, its V^T in the third LDS buffer), then S^T and softmax of tile t.  Next to
    // a wave that keeps the matrix pipe busy a partner's VALU gets one issue slot per MFMA (tools/ubench) - phase-aligned,
    // the two waves' softmax blocks (150 VALU, 32 of them quarter-rate v_exp_f32) simply add to the MFMA time; skewed,
    // and with the softmax at raised priority, one wave's VALU runs under the other's MFMAs.
    const bool late = wave >= 4;
    bf16x8 pt[2][2];                                                // P^T of the tile whose P V is still to come
    bf16x8 ob[2][6];                                                // operands one MFMA step ahead (the compiler alone issues each
                                                                    // ds_read right in front of its MFMA: 60 LDS latencies per tile)
#define NB_LOAD_QK(ks_, d_)                                                                                     \
    do {                                                                                                        \
        ob[d_][0] = *reinterpret_cast<const bf16x8*>(kah + (ks_) * 32);                                         \
        ob[d_][1] = *reinterpret_cast<const bf16x8*>(kah + 32 * NB_KROW + (ks_) * 32);                          \
        ob[d_][2] = *reinterpret_cast<const bf16x8*>(kal + (ks_) * 32);                                         \
        ob[d_][3] = *reinterpret_cast<const bf16x8*>(kal + 32 * NB_KROW + (ks_) * 32);                          \
    } while (0)
#define NB_LOAD_PV(vah_, val_, j_, d_)                                                                          \
    do {                                                                                                        \
        _Pragma("unroll") for (int ct_ = 0; ct_ < CT; ++ct_) {                                                  \
            ob[d_][ct_] = *reinterpret_cast<const bf16x8*>((vah_) + ct_ * 32 * NB_VROW + ((j_) >> 1) * 64 + ((j_) & 1) * 32);     \
            ob[d_][3 + ct_] = *reinterpret_cast<const bf16x8*>((val_) + ct_ * 32 * NB_VROW + ((j_) >> 1) * 64 + ((j_) & 1) * 32); \
        }                                                                                                       \
    } while (0)
    // O^T[ch][query] += V^T[ch][keys] P^T[keys][query] for the tile in `buf`, keys in the accumulator's own order;
    // channel tile innermost (consecutive MFMAs go to different accumulators)
#define NB_PV(buf_)                                                                                             \
    do {                                                                                                        \
        const unsigned char* const vah_ = (buf_) + 2 * NB_KT * NB_KROW + xl * NB_VROW + kh * 16;                \
        const unsigned char* const val_ = vah_ + NB_CP * NB_VROW;                                               \
        NB_LOAD_PV(vah_, val_, 0, 0);                                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            if (j_ < 3) NB_LOAD_PV(vah_, val_, j_ + 1, (j_ + 1) & 1);                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            _Pragma("unroll") for (int ct_ = 0; ct_ < CT; ++ct_)                                                \
                o[ct_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ob[j_ & 1][ct_], pt[j_ >> 1][j_ & 1], o[ct_], 0, 0, 0);     \
            _Pragma("unroll") for (int ct_ = 0; ct_ < CT; ++ct_)                                                \
                o[ct_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ob[j_ & 1][3 + ct_], pt[j_ >> 1][j_ & 1], o[ct_], 0, 0, 0); \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)

    for (int kt = kt0; kt < kt1; ++kt) {
        const int bi = (kt - kt0) % 3;
        unsigned char* const cur = sm + bi * NB_TILE_BYTES;
        unsigned char* const nxt = sm + (bi == 2 ? 0 : bi + 1) * NB_TILE_BYTES;   // held tile kt-2: read by nobody any more
        const unsigned char* const prv = sm + (bi == 0 ? 2 : bi - 1) * NB_TILE_BYTES;
        if (late && kt > kt0) NB_PV(prv);
        // S^T for both 32-key halves of the tile (two independent accumulators, alternating: no MFMA waits for its
        // predecessor), then ONE running-max / rescale update for the 64 keys
        const int kbase = kt * NB_KT;
        f32x16 st[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[sub][r] = 0.f;
        const unsigned char* const kah = cur + xl * NB_KROW + kh * 16;
        const unsigned char* const kal = kah + NB_KT * NB_KROW;
        NB_LOAD_QK(0, 0);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
            if (ks < 5) NB_LOAD_QK(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int d = ks & 1;
            st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ob[d][0], qh[ks], st[0], 0, 0, 0);
            st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ob[d][1], qh[ks], st[1], 0, 0, 0);
            st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ob[d][0], ql[ks], st[0], 0, 0, 0);
            st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ob[d][1], ql[ks], st[1], 0, 0, 0);
            st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ob[d][2], qh[ks], st[0], 0, 0, 0);
            st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ob[d][3], qh[ks], st[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(2);                              // the softmax VALU goes ahead of the partner wave's MFMAs
        if (kbase + NB_KT > N) {                                    // wave-uniform: only the last, partial key tile
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + sub * 32 + drow(r, lane) >= N) st[sub][r] = -INFINITY;
        }
        float tmax = fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[1][0], st[1][1]));
#pragma unroll
        for (int r = 2; r < 16; r += 2) tmax = fmaxf(tmax, fmaxf(fmaxf(st[0][r], st[0][r + 1]), fmaxf(st[1][r], st[1][r + 1])));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);         // m = -inf on the first tile -> 0
#ifdef NB_X_NOSOFTMAX   /* timing experiment only: wrong results */
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; r += 8) pt[sub][r >> 3] = __builtin_bit_cast(bf16x8, u32x4{__builtin_bit_cast(unsigned, st[sub][r]), __builtin_bit_cast(unsigned, st[sub][r + 1]), __builtin_bit_cast(unsigned, st[sub][r + 2]), __builtin_bit_cast(unsigned, st[sub][r + 3])});
#else
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) pt[sub][r >> 3][r & 7] = (__bf16)__builtin_amdgcn_exp2f(st[sub][r] - mn);
#endif
        m = mn;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!late) NB_PV(cur);
        if (kt + 1 < kt1) {                                         // next tile (requested a tile ago) -> the third buffer; the tile
            store_tile(nxt);                                        // after it requested
            if (kt + 2 < kt1) load_tile((kt + 2) * NB_KT);
        }
        __syncthreads();                                            // this tile's S^T operands are free, the next tile is complete
    }
    if (late) NB_PV(sm + ((kt1 - 1 - kt0) % 3) * NB_TILE_BYTES);    // the late waves' last P V
#undef NB_PV
#undef NB_LOAD_PV
#undef NB_LOAD_QK

    float l = o[LCT][LR];
    {
        const float lo = __shfl_xor(l, 32);
        if (kh != LKH) l = lo;
    }
    const float inv = (ksp == 1) ? 1.0f / l : 1.0f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] *= inv;

    // Z^T = W'^T O^T on the f32 matrix pipe, as in nonlocal.hip (pad rows of W' are zero: the row-sum channel drops out)
    constexpr int CTW = CP / 32;
#pragma unroll
    for (int cot = 0; cot < CTW; ++cot) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
            const float* wa = Wp + (size_t)(ct * 32 + 4 * kh) * CP + cot * 32 + xl;
#pragma unroll
            for (int s = 0; s < 16; ++s) z = mfma32(wa[((s & 3) + 8 * (s >> 2)) * CP], o[ct][s], z);
        }
        if (q < q1) {
            if (ksp == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cot * 32 + drow(r, lane);
                    if (co < C) {
                        const size_t idx = (size_t)q * CP + co;
                        Xob[idx] = Xb[idx] + z[r] + bp[co];            // residual, model/pfnl.py:60
                    }
                }
            } else {
                float* zp = Zp + (((size_t)b * ksp + sp) * N + q) * CP;
#pragma unroll
                for (int r = 0; r < 16; ++r) zp[cot * 32 + drow(r, lane)] = z[r];
            }
        }
    }
    if (ksp > 1 && q < q1 && kh == 0) {
        float* ml = ML + (((size_t)b * ksp + sp) * N + q) * 2;
        ml[0] = m;
        ml[1] = l;
    }
}

size_t nl_bf16_scratch_halfs(int B, int N) {                        // Khi, Klo, Vthi, Vtlo
    const size_t npad = (size_t)(N + 31) / 32 * 32 + 64;            // + one tile of slack for the last tile's V^T pieces
    return 2 * (size_t)B * N * NB_CP + 2 * (size_t)B * NB_CP * npad;
}

// X, Xo as in launch_nl_attn; scratch16: nl_bf16_scratch_halfs(B, N) 16-bit elements; partial: nl_partial_floats
hipError_t launch_nl_attn_bf16(const float* X, float* Xo, const float* Wp, const float* bp, float* partial, uint16_t* scratch16,
                               int B, int N, int C, hipStream_t s, int q0, int q1) {
    if (C != 84 && C != 60 && C != 36) return hipErrorInvalidValue;
    if (q1 < 0) q1 = N;
    if (q0 < 0 || q0 >= q1 || q1 > N) return hipErrorInvalidValue;
    const int CP = nl_padded_ch(C);
    const int npad = (N + 31) / 32 * 32 + 64;
    uint16_t* Khi = scratch16;
    uint16_t* Klo = Khi + (size_t)B * N * NB_CP;
    uint16_t* Vthi = Klo + (size_t)B * N * NB_CP;
    uint16_t* Vtlo = Vthi + (size_t)B * NB_CP * npad;
    {
        const size_t total = (size_t)B * npad * NB_CP;
        const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(nl_pack_bf16_kernel, dim3(blocks), dim3(256), 0, s, X, Khi, Klo, Vthi, Vtlo, B, N, npad, C, CP);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // key splits: this kernel runs 1 workgroup per CU (LDS), so the grid should fill a whole number of 256-workgroup
    // rounds: time ~ ceil(query blocks * B * ks / 256) / ks.  (1080p: 127 blocks -> ks = 2 is one full round.)  Bounded
    // by the fp32 kernel's choice, which sized the partial-result buffer.
    const int ks_max = nl_key_splits(B, N);
    int ks = 1;
    {
        const long long qb = (long long)((q1 - q0 + NB_QB - 1) / NB_QB) * B;
        double best = 1e30;
        for (int k = 1; k <= ks_max; ++k) {
            const double t = (double)((qb * k + 255) / 256) / k;
            if (t < best - 1e-9) {
                best = t;
                ks = k;
            }
        }
    }
    if (ks > 1 && !partial) return hipErrorInvalidValue;
    float* Zp = partial;
    float* ML = partial ? partial + (size_t)B * ks * N * CP : nullptr;
    dim3 grid((q1 - q0 + NB_QB - 1) / NB_QB, B, ks);
    dim3 block(NB_THREADS);
    static bool attr_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_dev[dev]) {
        for (const void* fn : {reinterpret_cast<const void*>(nl_attn_bf16_kernel<84>), reinterpret_cast<const void*>(nl_attn_bf16_kernel<60>),
                               reinterpret_cast<const void*>(nl_attn_bf16_kernel<36>)}) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, NB_LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        attr_dev[dev] = true;
    }
    switch (C) {
        case 84: hipLaunchKernelGGL(nl_attn_bf16_kernel<84>, grid, block, NB_LDS_BYTES, s, X, Khi, Klo, Vthi, Vtlo, Xo, Wp, bp, Zp, ML, N, npad, q0, q1); break;
        case 60: hipLaunchKernelGGL(nl_attn_bf16_kernel<60>, grid, block, NB_LDS_BYTES, s, X, Khi, Klo, Vthi, Vtlo, Xo, Wp, bp, Zp, ML, N, npad, q0, q1); break;
        case 36: hipLaunchKernelGGL(nl_attn_bf16_kernel<36>, grid, block, NB_LDS_BYTES, s, X, Khi, Klo, Vthi, Vtlo, Xo, Wp, bp, Zp, ML, N, npad, q0, q1); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ks == 1) return e;
    return launch_nl_merge(X, Zp, ML, bp, Xo, B, N, C, ks, s, q0, q1);
}

}  // namespace pfnl

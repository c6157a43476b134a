// Non-local block of the fp32 path on the f16 matrix pipe with exactly split operands (option nonlocal=split16; the same idea as
// conv_split16.hip: x = hi + lo with hi = f16(x), lo = f16(x - hi), products hi hi + hi lo + lo hi in fp32 accumulators: >= 22 mantissa
// bits per product).  Derived from nonlocal_bf16.hip (same streaming-softmax structure, same operand layouts); what differs:
//   * binary16 has a 5-bit exponent, so everything is kept in its normal range by powers of two that cancel exactly:
//     K, V and Q are scaled by 2^7 (the image is in [0,1]: hi <= 185, lo normal for x >= 0.001), the logits therefore come out of
//     the MFMA scaled by 2^14 (undone inside the exp2 argument's fma), the probabilities are computed as 2^14 exp2(s - max) (<= 16384:
//     representable down to 2^-28 of the row maximum) - the row sum accumulates the same scaled values, so O / l needs only 2^-7;
//   * P is split as well (hi + lo): 3 MFMAs per V^T P^T product block instead of 2 - 72 f16 MFMAs of 32 cycles per 64 keys and 32
//     queries against 180 f32 MFMAs of 64 in nonlocal.hip.
// ---- (the description of the structure, from nonlocal_bf16.hip:)
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
//   * running max / rescale, normalisation and the residual as in nonlocal.hip; the folded 1x1 projection in the same split arithmetic (round 6:
//     W' through LDS in operand form, O^T from the accumulators as the B operand - see the kernel's epilogue).
// Operand layouts (lane = (l & 31, kh = l >> 5), 8 bf16 per lane and MFMA):
//   K tile in LDS  [key][96 ch] (+ pad to 208 B: conflict-free b128 reads), hi and lo: A of S^T = K Q^T;
//   Q in registers [6 k-steps] hi and lo, pre-scaled by log2(e): B of S^T;
//   P^T straight from the S^T accumulator: register r of lane (query, kh) is key (r&3) + 8(r>>2) + 4kh, registers
//   8t..8t+7 form the B operand of k-step t - the contraction order over keys is free, so V^T is stored by nl_pack_bf16
//   with the keys of every 32-block permuted to exactly that order ([ch][block][t][kh][e]).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_bf16.h"
#include "conv_split16.h"

namespace pfnl {

typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));   // (name kept from the bf16 kernel: 8 x binary16 here)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NF_KT = 64;                  // keys per LDS tile
constexpr int NF_KROW = 208;               // bytes per key row of the K tiles (96 ch * 2 B + 16)
constexpr int NF_VROW = 144;               // bytes per channel row of the V^T tiles (64 keys * 2 B + 16)
constexpr int NF_CP = 96;
constexpr int NF_THREADS = 512;           // 8 waves x 32 queries share every key tile: the K / V^T stream (48 KB per 64 keys) is what
                                          // bounds this kernel - with 128 queries per workgroup it ran at 6.6 TB/s of L2 -> CU traffic
                                          // (11 B/clk/CU) and 43 % matrix-pipe use, whatever was done to its instruction schedule
constexpr int NF_QB = NF_THREADS / 2;     // queries per workgroup
// The transposing-read form of nl_attn_f16_sw_kernel (round 6; TR = true below): built, parity-green on its first run (every non-local test and golden
// forward), and NOT faster - 0.0995 against 0.0990 ms at configs[1], 0.956 - 0.981 against 0.961 - 0.966 at 1080p fp32, 0.413 - 0.417 against 0.417 - 0.423 at
// 1080p bf16 (profiles/r06_nl_tr.txt): halving the bytes a key tile pulls through the CU's port does not move a kernel whose matrix pipe is 79 % busy at 1080p
// (and whose 16 tiles per workgroup at configs[1] sit between a query prologue and the 96 x 96 projection epilogue); its reads conflict on this tile's 208-byte
// rows (SQ_LDS_BANK_CONFLICT 5.0e7 per launch, 0 without).  The product library does not instantiate it: build with -DNF_TR_BUILD (tools/build_variant.sh) and
// run with PFNL_NL_TR=1 to reproduce.
#ifdef NF_TR_BUILD
#define NF_TR_INSTANCES 1
#else
#define NF_TR_INSTANCES 0
#endif

constexpr float NF_XSCALE = 128.0f;                       // 2^7 on K, V and Q
constexpr float NF_SINV = 1.0f / (128.0f * 128.0f);       // logits leave the MFMA scaled by 2^14
constexpr float NF_PSHIFT = 14.0f;                        // probabilities are kept as 2^14 exp2(s - max)
__device__ __forceinline__ unsigned short bf16_bits(float f) {   // binary16, round to nearest even
    const _Float16 b = (_Float16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_float(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }

// X [B][N][CP] fp32 (nl_pack_kernel) -> Khi, Klo [B][Npad][96] binary16;  Vthi, Vtlo [B][96][Npad] binary16, keys permuted per
// 32-block, channel C = 1 (the row-sum channel), channels > C = 0.  One workgroup per 32-key block: rows of X in, rows of K out, the
// transposed block through LDS so that V^T leaves as 64-byte row segments (round 4: the element-per-thread form wrote V^T as 2-byte
// stores a row apart - 38 us at 1080p for 31 MB).
// FROMX (round 6): the same launch ALSO does nl_pack_kernel's job - it reads the clip x [B][T][H][W][3] itself through the space_to_depth
// index map (model/pfnl.py:55-57: channel (dy 2 + dx) 3T + 3t + c of grid cell n) and writes X [B][N][CPin] fp32 (pad columns 0) next to the
// binary16 operands: one launch and one pass over the frames instead of two launches of ~10 us with X read back in between.
template <bool FROMX>
__global__ __launch_bounds__(256) void nl_pack_f16_kernel(const float* __restrict__ X, uint16_t* __restrict__ Khi, uint16_t* __restrict__ Klo,
                                                          uint16_t* __restrict__ Vthi, uint16_t* __restrict__ Vtlo, int B, int N, int Npad, int C,
                                                          int CPin, const float* __restrict__ xraw, float* __restrict__ Xout, int H, int W) {
    __shared__ uint16_t vth[NF_CP][34], vtl[NF_CP][34];             // [channel][position in the block] (+2: odd word stride)
    const int nblk = Npad / 32;
    const int b = blockIdx.x / nblk, n0 = (blockIdx.x - b * nblk) * 32;
    const int tid = threadIdx.x;
    [[maybe_unused]] const int C3 = C / 4, T = C / 12, W2 = W / 2;
#pragma unroll
    for (int j = 0; j < 32 * NF_CP / 256; ++j) {
        const int i = tid + 256 * j;
        const int kk = i / NF_CP, c = i - kk * NF_CP;
        const int n = n0 + kk;
        float raw = 0.f;
        if constexpr (FROMX) {
            if (n < N && c < C) {
                const int sub = c / C3, k = c - sub * C3;
                const int t = k / 3, cc = k - 3 * t;
                const int y = 2 * (n / W2) + (sub >> 1), xx = 2 * (n % W2) + (sub & 1);
                raw = xraw[((((size_t)b * T + t) * H + y) * W + xx) * 3 + cc];
            }
            if (n < N && c < CPin) Xout[((size_t)b * N + n) * CPin + c] = raw;
        } else {
            raw = (n < N && c < C) ? X[((size_t)b * N + n) * CPin + c] : 0.f;
        }
        const float v = raw * NF_XSCALE;
        const unsigned short hi = bf16_bits(v);
        const unsigned short lo = bf16_bits(v - bf16_float(hi));
        // key rows up to Npad (>= the last tile's end).  Keys past N carry -65504 in the pad channel C: a query operand with a
        // positive entry there (nl_attn_f16_sw_kernel: 1024) gets a logit of -4 000 for them - the key mask as DATA, no code
        // channel C + 1 = 1.0 for every key: the row-sum ("ones") channel of the TR form of the kernel, which reads its P V operands out of the K tile
        // (nl_attn_f16_sw_kernel<C, SPLIT, true>); the query operand is 0 there, so the logits of either form do not see it
        Khi[((size_t)b * Npad + n) * NF_CP + c] = (n >= N && c == C) ? (unsigned short)0xfbff : (c == C + 1 ? (unsigned short)0x3c00 : hi);
        Klo[((size_t)b * Npad + n) * NF_CP + c] = lo;
        // position of key kk inside its 32-block: key = (e&3) + 8(2t + (e>>2)) + 4kh  ->  pos = 16t + 8kh + e
        const int e = (kk & 3) | (((kk >> 3) & 1) << 2), kh = (kk >> 2) & 1, t = kk >> 4;
        const int pos = 16 * t + 8 * kh + e;
        vth[c][pos] = c == C ? (unsigned short)0x3c00 : hi;         // 1.0 (binary16)
        vtl[c][pos] = c == C ? (unsigned short)0 : lo;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16 * NF_CP / 256; ++j) {                    // 96 rows x 16 pairs of positions
        const int i = tid + 256 * j;
        const int c = i >> 4, pp = (i & 15) * 2;
        const size_t vp = ((size_t)b * NF_CP + c) * Npad + n0 + pp;
        *reinterpret_cast<unsigned*>(Vthi + vp) = (unsigned)vth[c][pp] | ((unsigned)vth[c][pp + 1] << 16);
        *reinterpret_cast<unsigned*>(Vtlo + vp) = (unsigned)vtl[c][pp] | ((unsigned)vtl[c][pp + 1] << 16);
    }
}

// SPLIT = true: the fp32 path (operands as hi + lo, 72 MFMAs per 64 keys).  SPLIT = false: the same kernel on the hi parts only
// (24 MFMAs per 64 keys) - 16-bit operands, fp32 accumulation: the non-local block of precision=bf16, whose trunk is 16-bit
// anyway (binary16 has 3 more mantissa bits than the bf16 of the trunk: logits good to ~2e-3, where bf16 logits are off by 16 %).
// (The first-generation kernel nl_attn_f16_kernel - two half-workgroups half a tile apart - left the library in round 5:
// tools/experiments/nonlocal_f16_v1_kernel.hip.)


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4: the arithmetic of the first-generation kernel, software-pipelined INSIDE a wave (nl_attn_f16_sw_kernel).  Counted on that
// first kernel at 1080p (hi parts only:
// 4.05 k cycles per 64-key tile and SIMD for 1.5 k of MFMA and 2 x 1.0 k of softmax VALU; split: 7.6 k for 4.6 k of MFMA): its half-tile
// skew lets one wave's softmax face the other wave's MFMAs - and on this chip that hides nothing: the VALU of one wave gets about one
// issue slot per MFMA of the OTHER wave of its SIMD (tools/ubench/valu_under_mfma; a two-phase ping-pong of the half-workgroups measured
// Y + X per phase, not max(Y, X): tools/experiments/README.md).  What a wave issues between two of its OWN MFMAs does ride in their
// shadow.  So the loop runs over 32-key halves h, and ONE half-body issues, from one wave and in one pinned order,
//     the MFMAs of S^T(h + 1) and P V(h - 1)   |   their operand reads, AHEAD reads in front   |   the VALU of softmax(h)
// with S^T double-buffered per half (the same 32 registers as before) and no phase difference between the waves.  Three things this
// needed beyond writing it down:
//   * every step pinned with sched_barrier(0): left alone the compiler clumps the three streams (twelve reads with a wait after each,
//     the max reduction as one block);
//   * the exponentials pinned where they are written (NP_PIN): their only use is one half-body later, and LLVM sinks a pure computation
//     to its use - the whole softmax reappeared in front of the first P V MFMA that consumes it, which is the SUM again;
//   * no branch in the body: with the key mask of the last, partial tile as a second copy of the body behind a (uniform) branch the
//     allocator kept nine 16-register accumulator blocks instead of five (250 registers hi-only, spills split).  The mask is DATA now:
//     key rows past N carry -65504 in pad channel C (nl_pack_f16_kernel), the query operand 1024 there - a logit of -4 000.
// The K / V^T tiles travel by LDS-DMA (`buffer_load_dwordx4 ... lds`; the four packed arrays are one allocation = one buffer resource,
// the padded LDS rows are produced by per-lane source offsets, the pad chunks by an out-of-range offset = zeros): no staging registers, no
// ds_write.  Ring of NSLOT tiles (3 with split operands - LDS is full - and 5 of 32 KB on the hi parts only); one workgroup barrier per
// tile, between its two half-bodies: behind it tile t + 1 is complete and the slot of tile t - 1 is free for tile t + NSLOT - 1, whose pieces
// ride on the MFMA gaps of half-body b.  A wave waits for its own pieces of tile t + 1 in front of that barrier with `s_waitcnt
// vmcnt((NSLOT - 3) x pieces per wave)`: the counter retires in issue order and every wave issues the same number of pieces per tile (past
// the last tile: against an empty resource), so the count is exact.
// TR (round 6): the tile is the K rows ONLY - the P V operands (per lane 8 keys of one channel: a column of the K tile) come out of them by the
// transposing read ds_read_b64_tr_b16 (tools/ubench/ds_tr_b16.hip has its law), so V^T is neither packed, nor moved, nor held: half the bytes a tile pulls
// through the CU's port (what bounds the split form), a ring of 5 / 8 tiles.
template <bool SPLIT, bool TR = false>
struct NfSW {
    static constexpr int KLO_OFF = NF_KT * NF_KROW;                                    // 13 312 (split only)
    static constexpr int VHI_OFF = SPLIT ? 2 * NF_KT * NF_KROW : NF_KT * NF_KROW;      // 26 624 | 13 312
    static constexpr int VLO_OFF = VHI_OFF + NF_CP * NF_VROW;                          // + 13 824 (split only)
    static constexpr int TILE_BYTES = TR ? VHI_OFF : (SPLIT ? VLO_OFF + NF_CP * NF_VROW : VLO_OFF);   // 54 272 | 27 136; TR: 26 624 | 13 312
    static constexpr int NSLOT = TR ? (SPLIT ? 5 : 8) : (SPLIT ? 3 : 5);
    static constexpr int PIECES = TR ? (SPLIT ? 32 : 16) : (SPLIT ? 53 : 32);          // 1 KB DMA instructions per tile (hi only: 27 carry data; TR: 26 | 13)
    static constexpr int SLOT_BYTES = PIECES * 1024;                                   // 54 272 | 32 768; TR: 32 768 | 16 384
    static constexpr int PW = (PIECES + 7) / 8;                                        // per wave: 7 (waves 5-7: 6) | 4; TR: 4 | 2
    static constexpr int LDS_BYTES = NSLOT * SLOT_BYTES;                               // 162 816 | 163 840; TR: 163 840 | 131 072
    static_assert(TILE_BYTES <= SLOT_BYTES && LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((SPLIT && !TR) || PIECES % 8 == 0, "the vmcnt wait of a ring deeper than 3 counts on equal shares");
};

// P V operand of a v_mfma_f32_32x32x16_f16 out of K rows: `p` = this lane's piece address for the operand's first four keys (row k0 + 4 kh + ((lane & 15) >> 2),
// channels c0 + 16 ((lane >> 4) & 1) + 4 (lane & 3) .. + 3); the second four keys are 8 rows on.  Lane (i = lane & 31, kh) receives keys k0 + 4 kh + {0..3, 8..11}
// of channel c0 + i - the order the S^T accumulator registers 8 t .. 8 t + 7 hold them in (tools/ubench/ds_tr_b16.hip checks exactly this addressing).
typedef __fp16 nf_fp4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned nf_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 nf_tr_operand(const unsigned char* p) {
    typedef __attribute__((address_space(3))) nf_fp4 lds_fp4;
    const nf_u2 a = __builtin_bit_cast(nf_u2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp4*)(p)));
    const nf_u2 b = __builtin_bit_cast(nf_u2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp4*)(p + 8 * NF_KROW)));
    return __builtin_bit_cast(bf16x8, u32x4{a.x, a.y, b.x, b.y});
}

__device__ __forceinline__ void nf_dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}
#define NF_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef PFNL_NP_TIMING   /* phase timeline (tools/np_timing.py); not part of the product build */
__device__ long long np_dbg[256 * 2 * 128];
#define NP_STAMP() do { if (SPLIT == (PFNL_NP_TIMING != 0) && lane == 0 && (wave == 0 || wave == 4) && dbg_n < 128 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 256) np_dbg[(blockIdx.x * 2 + (wave != 0)) * 128 + dbg_n++] = __builtin_readcyclecounter(); } while (0)
__device__ long long np_ph[256 * 8];                                // phases of a workgroup (tools/np_phases.py): constant-clock ticks (10 ns)
#define NP_PHASE(i_) do { if (SPLIT == (PFNL_NP_TIMING != 0) && tid == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 256) np_ph[blockIdx.x * 8 + (i_)] = wall_clock64(); } while (0)
#else
#define NP_STAMP() do {} while (0)
#define NP_PHASE(i_) do {} while (0)
#endif

// K16 = Khi (the lowest address of the scratch); rel_* = byte offsets of Klo, Vthi, Vtlo from it; scratch_bytes = the whole allocation
template <int C, bool SPLIT, bool TR>
__global__ __launch_bounds__(NF_THREADS, 2) void nl_attn_f16_sw_kernel(const float* __restrict__ X, const uint16_t* __restrict__ K16,
                                                                 unsigned rel_klo, unsigned rel_vhi, unsigned rel_vlo, unsigned scratch_bytes,
                                                                 float* __restrict__ Xo, const float* __restrict__ Wp,
                                                                 const float* __restrict__ bp, float* __restrict__ Zp,
                                                                 float* __restrict__ ML, int N, int Npad, int q0, int q1) {
    using G = NfSW<SPLIT, TR>;
    constexpr int CT = 3;
    constexpr int CP = (C + 31) / 32 * 32;                          // row stride of X / Xo / Wp (nl_padded_ch)
    static_assert(C + 1 < NF_CP && C % 2 == 0, "needs two pad channels inside 96");
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];   // NSLOT tiles: K hi | (K lo) | V^T hi | (V^T lo)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PFNL_NP_TIMING
    int dbg_n = 0;
#endif
    NP_PHASE(0);                                                    // entry
    const int xl = lane & 31;
    const int kh = lane >> 5;
    const int b = blockIdx.y;
    const float* Xb = X + (size_t)b * N * CP;
    float* Xob = Xo + (size_t)b * N * CP;
    const int q = q0 + blockIdx.x * NF_QB + (tid >> 6) * 32 + xl;   // this lane's query (queries [q0, q1): a strip of the frame)
    const int qc = q < q1 ? q : q1 - 1;

    // B operand of S^T = K Q^T: this lane's query, channels 16ks + 8kh .. +7, scaled by log2(e), split hi + lo
    constexpr float LOG2E = 1.4426950408889634f;
    bf16x8 qh[6];
    [[maybe_unused]] bf16x8 ql[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 16 * ks + 8 * kh + e;
            // (channel C: the key rows past N carry -65504 there (nl_pack_f16_kernel), every other key 0 - the key mask as data)
            const float v = c < C ? Xb[(size_t)qc * CP + c] * (LOG2E * NF_XSCALE) : (c == C ? 1024.f : 0.f);
            const _Float16 h = (_Float16)v;
            qh[ks][e] = h;
            if constexpr (SPLIT) ql[ks][e] = (_Float16)(v - (float)h);
        }
    constexpr int CS = TR ? C + 1 : C;                              // the row-sum ("ones") channel: V^T's channel C; TR: K's channel C + 1 (C is the key mask)
    constexpr int LCT = CS / 32, LI = CS % 32;                      // where it lives in the D layout
    constexpr int LKH = (LI % 8) >= 4 ? 1 : 0, LR = (LI / 8) * 4 + (LI % 8) % 4;

    f32x16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m = -INFINITY;

    // DMA map: piece i = wave + 8k writes slot bytes [1024 i, + 1024), lane L the 16 bytes at p = 1024 i + 16 L.  `srel` = byte offset of
    // that chunk's source from K16 for key tile 0 (0x7fffffff: a pad chunk - stays out of range).  Per tile: + k0 x (192 | 2); every key
    // row and V^T column of a tile exists (both arrays are padded past the last tile; keys past N are masked by their data).
    int srel[G::PW];
#pragma unroll
    for (int k = 0; k < G::PW; ++k) {
        const int p = (wave + 8 * k) * 1024 + 16 * lane;
        int sr = 0x7fffffff;
        if (p < NF_KT * NF_KROW || (SPLIT && p < 2 * NF_KT * NF_KROW)) {
            const bool lo = p >= NF_KT * NF_KROW;
            const int r = p - (lo ? NF_KT * NF_KROW : 0);
            const int row = r / NF_KROW, col = r - row * NF_KROW;
            if (col < NF_CP * 2) {
                sr = (int)((lo ? rel_klo : 0u) + (unsigned)(((size_t)b * Npad + row) * (NF_CP * 2)) + col);
            }
        } else if (!TR && p < G::TILE_BYTES) {
            const bool lo = SPLIT && p >= G::VLO_OFF;
            const int r = p - (lo ? G::VLO_OFF : G::VHI_OFF);
            const int ch = r / NF_VROW, col = r - ch * NF_VROW;
            if (col < NF_KT * 2) {
                sr = (int)((lo ? rel_vlo : rel_vhi) + (unsigned)(((size_t)b * NF_CP + ch) * Npad * 2) + col);
            }
        }
        srel[k] = sr;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)sm;
    // piece k of this wave for key tile `kt` -> ring slot `slot`; `live` false (past the last tile): an empty resource, nothing moves
    auto tile_piece = [&](int k, int kt, int slot, bool live) __attribute__((always_inline)) {
        const int i = wave + 8 * k;
        if (G::PIECES % 8 == 0 || 8 * k + 7 < G::PIECES || i < G::PIECES) {   // (wave-uniform; split: waves 5-7 have no seventh piece)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(K16), 0, live ? scratch_bytes : 0, 0x00020000);
            const int k0 = kt * NF_KT;
            const bool isk = TR || i * 1024 < (SPLIT ? 2 : 1) * NF_KT * NF_KROW;   // (the K | V^T boundary is a multiple of 1024)
            const int off = (int)((unsigned)srel[k] + (unsigned)(k0 * (isk ? NF_CP * 2 : 2)));   // (a pad chunk stays past the range)
            nf_dma16(rs, lds0 + slot * G::SLOT_BYTES + i * 1024, off);
        }
    };

    const int ntiles = (N + NF_KT - 1) / NF_KT;
    const int ksp = gridDim.z, sp = blockIdx.z;
    const int kt0 = (int)((long long)ntiles * sp / ksp), kt1 = (int)((long long)ntiles * (sp + 1) / ksp);
    // prologue: tiles kt0 .. kt0 + NSLOT - 2 -> slots 0 ..; an all-zero tile -> slot NSLOT - 1, the "tile kt0 - 1" of the first P V
    // (whose P^T is zero: no branch around it, but 0 x whatever the LDS held could be NaN)
#pragma unroll
    for (int j = 0; j < G::NSLOT - 1; ++j)
#pragma unroll
        for (int k = 0; k < G::PW; ++k) tile_piece(k, kt0 + j, j, kt0 + j < kt1);
#pragma unroll
    for (int k = 0; k < G::PW; ++k) tile_piece(k, 0, G::NSLOT - 1, false);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    bf16x8 pt[2][2];                                                // P^T of the two 32-key halves of a tile (hi, lo parts)
    [[maybe_unused]] bf16x8 pl[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pt[i][j][e] = (_Float16)0.f;
                if constexpr (SPLIT) pl[i][j][e] = (_Float16)0.f;
            }
#define NP_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, c_, 0, 0, 0)
    f32x16 st[2];                                                   // S^T of the 32-key half in flight and of the one being exponentiated
    // MFMA i of a half-body: i < NQ: S^T k-step i (split: k-step i/3, products hh, hl, lh); then P V: (key step jj, channel tile ct)
    // (split: x (Vh Ph, Vh Pl, Vl Ph) - consecutive MFMAs on one accumulator run at full rate: tools/ubench/mfma_bf16_rate).  Its A
    // operand is read q(i) of the half-body's read sequence: i (hi only) | 2 (i / 3) + (i % 3 == 2) (split: hi row, lo row per step).
    constexpr int NQ = SPLIT ? 18 : 6, NR = SPLIT ? 24 : 12;
    // (an opaque use where a value is written: see the header - LLVM would sink the softmax to the MFMA that consumes it)
#define NP_PIN(v_) asm volatile("" : "+v"(v_))
#ifndef NP_AHEAD
#define NP_AHEAD 4
#endif
    constexpr int AHEAD = NP_AHEAD;                                 // reads in flight in front of the one in use (<= 6)
    constexpr int RING = AHEAD + 2;
    bf16x8 rb[RING];
    auto half_body = [&](auto hc, const unsigned char* qbuf, const unsigned char* vbuf, int kt_fill, int s_fill) __attribute__((always_inline)) {
        constexpr int H = decltype(hc)::value;                      // 0: half a of a tile (softmax of its keys 0..31), 1: half b
        f32x16& stc = st[H];
        f32x16& stn = st[H ^ 1];
        // S^T(h + 1): key rows 32 (1 - H) .. of `qbuf`;  P V(h - 1): V^T keys 32 (1 - H) .. of `vbuf`, P^T = pt[1 - H]
        const unsigned char* const kah = qbuf + ((1 - H) * 32 + xl) * NF_KROW + kh * 16;
        const unsigned char* const vah = vbuf + G::VHI_OFF + xl * NF_VROW + kh * 16 + (1 - H) * 64;
        // TR: this lane's piece of the first P V operand of the half (keys 32 (1 - H) ..; + 16 keys per key step jj, + 64 bytes per channel tile ct)
        [[maybe_unused]] const unsigned char* const vtr = vbuf + ((1 - H) * 32 + 4 * kh + ((lane & 15) >> 2)) * NF_KROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
        auto read = [&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q < NR) {
                constexpr int d = q % RING;
                if constexpr (!SPLIT) {
                    if constexpr (q < 6) rb[d] = *reinterpret_cast<const bf16x8*>(kah + q * 32);
                    else if constexpr (TR) rb[d] = nf_tr_operand(vtr + ((q - 6) / 3) * 16 * NF_KROW + ((q - 6) % 3) * 64);
                    else rb[d] = *reinterpret_cast<const bf16x8*>(vah + ((q - 6) % 3) * 32 * NF_VROW + ((q - 6) / 3) * 32);
                } else {
                    if constexpr (q < 12) rb[d] = *reinterpret_cast<const bf16x8*>(kah + (q & 1) * G::KLO_OFF + (q >> 1) * 32);
                    else {
                        constexpr int u = (q - 12) >> 1, lo = (q - 12) & 1;   // u = 3 jj + ct
                        if constexpr (TR) rb[d] = nf_tr_operand(vtr + lo * G::KLO_OFF + (u / 3) * 16 * NF_KROW + (u % 3) * 64);
                        else rb[d] = *reinterpret_cast<const bf16x8*>(vah + lo * (G::VLO_OFF - G::VHI_OFF) + (u % 3) * 32 * NF_VROW + (u / 3) * 32);
                    }
                }
            }
        };
        auto mfma = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int q = SPLIT ? 2 * (i / 3) + (i % 3 == 2) : i;
            const bf16x8 a = rb[q % RING];
            if constexpr (i < NQ) {
                constexpr int ks = SPLIT ? i / 3 : i;
                bf16x8 bq = qh[ks];
                if constexpr (SPLIT) {
                    if constexpr (i % 3 == 1) bq = ql[ks];
                }
                if constexpr (i == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    stn = NP_MFMA(a, bq, z);
                } else {
                    stn = NP_MFMA(a, bq, stn);
                }
            } else {
                constexpr int j = i - NQ;
                constexpr int u = SPLIT ? j / 3 : j;                // 3 jj + ct
                constexpr int jj = u / 3, ct = u % 3;
                bf16x8 bp = pt[1 - H][jj];
                if constexpr (SPLIT) {
                    if constexpr (j % 3 == 1) bp = pl[1 - H][jj];
                }
                o[ct] = NP_MFMA(a, bp, o[ct]);
            }
        };
        // ---- reads in flight
#pragma unroll
        for (int q0_ = 0; q0_ <= AHEAD; ++q0_) {
            if (q0_ == 0) read(std::integral_constant<int, 0>{});
            if (q0_ == 1) read(std::integral_constant<int, 1>{});
            if (q0_ == 2) read(std::integral_constant<int, 2>{});
            if (q0_ == 3) read(std::integral_constant<int, 3>{});
            if (q0_ == 4) read(std::integral_constant<int, 4>{});
            if (q0_ == 5) read(std::integral_constant<int, 5>{});
            if (q0_ == 6) read(std::integral_constant<int, 6>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        float tmax = 0.f, tsw = 0.f, alpha = 1.f, shift = 0.f;
        // ---- the MFMAs, one gap each: the read AHEAD in front and a slice of the softmax - gaps 0..3 the running max of the 32 keys (a
        // dependent chain with a cross-lane exchange in it: as a block in front of the MFMAs it was ~300 cycles of idle matrix pipe per
        // half), then the exponentials (a pair of keys per gap: 2 fma, 2 exp, the packed conversion; split: the lo parts in the gap
        // after) - and in b a request of the next tile now and then
        [[maybe_unused]] float pv0 = 0.f, pv1 = 0.f;
        auto gap = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            mfma(ic);
            __builtin_amdgcn_sched_barrier(0);
            {   // the first MFMA that takes read q sends read q + AHEAD on its way (reads 0 .. AHEAD went out in front of the max)
                constexpr int qn = (SPLIT ? 2 * (i / 3) + (i % 3 == 2) : i) + AHEAD + 1;
                constexpr int qp = i == 0 ? AHEAD + 1 : (SPLIT ? 2 * ((i - 1) / 3) + ((i - 1) % 3 == 2) : i - 1) + AHEAD + 1;
                if constexpr (qn > qp) read(std::integral_constant<int, qn - 1 < NR ? (qn - 1 >= qp ? qn - 1 : NR) : NR>{});
            }
            if constexpr (i == 0) {
                // (v_max3_f32 by hand: fmaxf() puts a canonicalising v_max x, x in front of every MFMA output - 20 instructions for 16
                // values; the logits are finite)
                float ta, tb;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(ta) : "v"(stc[0]), "v"(stc[1]), "v"(stc[2]));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tb) : "v"(stc[8]), "v"(stc[9]), "v"(stc[10]));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(ta) : "v"(ta), "v"(stc[3]), "v"(stc[4]));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tb) : "v"(tb), "v"(stc[11]), "v"(stc[12]));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(ta) : "v"(ta), "v"(stc[5]), "v"(stc[6]));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tb) : "v"(tb), "v"(stc[13]), "v"(stc[14]));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(ta) : "v"(ta), "v"(stc[7]), "v"(stc[15]));
                asm("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(ta), "v"(tb));
                NP_PIN(tmax);
            } else if constexpr (i == 1) {
                tsw = __shfl_xor(tmax, 32);
                NP_PIN(tsw);
            } else if constexpr (i == 3) {
                asm("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(tmax), "v"(tsw));
                tmax *= NF_SINV;                                    // true base-2 logit (the scale is positive: max commutes)
                const float mn = fmaxf(m, tmax);
                alpha = __builtin_amdgcn_exp2f(m - mn);             // m = -inf on the first half -> 0
                shift = NF_PSHIFT - mn;
                m = mn;
                NP_PIN(shift);
                NP_PIN(alpha);
            }
            if constexpr (!SPLIT) {
                if constexpr (i >= 4) {
                    constexpr int r = 2 * (i - 4);
                    const float a0 = __builtin_amdgcn_exp2f(__builtin_fmaf(stc[r], NF_SINV, shift));       // 2^14 exp2(s - max)
                    const float a1 = __builtin_amdgcn_exp2f(__builtin_fmaf(stc[r + 1], NF_SINV, shift));
                    pt[H][r >> 3][r & 7] = (_Float16)a0;
                    pt[H][r >> 3][(r & 7) + 1] = (_Float16)a1;
                    NP_PIN(pt[H][r >> 3]);
                }
            } else {
                if constexpr (i >= 4 && i % 2 == 0) {
                    constexpr int r = 2 * ((i - 4) / 4);
                    if constexpr (((i - 4) / 2) % 2 == 0) {
                        pv0 = __builtin_amdgcn_exp2f(__builtin_fmaf(stc[r], NF_SINV, shift));
                        pv1 = __builtin_amdgcn_exp2f(__builtin_fmaf(stc[r + 1], NF_SINV, shift));
                        asm volatile("" : "+v"(pv0), "+v"(pv1));
                    } else {
                        const _Float16 h0 = (_Float16)pv0, h1 = (_Float16)pv1;
                        pt[H][r >> 3][r & 7] = h0;
                        pt[H][r >> 3][(r & 7) + 1] = h1;
                        pl[H][r >> 3][r & 7] = (_Float16)(pv0 - (float)h0);
                        pl[H][r >> 3][(r & 7) + 1] = (_Float16)(pv1 - (float)h1);
                        NP_PIN(pt[H][r >> 3]);
                        NP_PIN(pl[H][r >> 3]);
                    }
                }
            }
            if constexpr (H == 1) {
                constexpr int step = SPLIT ? 5 : 3;
                if constexpr (i % step == 2 && i / step < G::PW) tile_piece(i / step, kt_fill, s_fill, kt_fill < kt1);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        gap(std::integral_constant<int, 0>{});
        gap(std::integral_constant<int, 1>{});
        gap(std::integral_constant<int, 2>{});
        gap(std::integral_constant<int, 3>{});
        gap(std::integral_constant<int, 4>{});
        gap(std::integral_constant<int, 5>{});
        gap(std::integral_constant<int, 6>{});
        gap(std::integral_constant<int, 7>{});
        gap(std::integral_constant<int, 8>{});
        gap(std::integral_constant<int, 9>{});
        gap(std::integral_constant<int, 10>{});
        gap(std::integral_constant<int, 11>{});
        if constexpr (SPLIT) {
            gap(std::integral_constant<int, 12>{});
            gap(std::integral_constant<int, 13>{});
            gap(std::integral_constant<int, 14>{});
            gap(std::integral_constant<int, 15>{});
            gap(std::integral_constant<int, 16>{});
            gap(std::integral_constant<int, 17>{});
            gap(std::integral_constant<int, 18>{});
            gap(std::integral_constant<int, 19>{});
            gap(std::integral_constant<int, 20>{});
            gap(std::integral_constant<int, 21>{});
            gap(std::integral_constant<int, 22>{});
            gap(std::integral_constant<int, 23>{});
            gap(std::integral_constant<int, 24>{});
            gap(std::integral_constant<int, 25>{});
            gap(std::integral_constant<int, 26>{});
            gap(std::integral_constant<int, 27>{});
            gap(std::integral_constant<int, 28>{});
            gap(std::integral_constant<int, 29>{});
            gap(std::integral_constant<int, 30>{});
            gap(std::integral_constant<int, 31>{});
            gap(std::integral_constant<int, 32>{});
            gap(std::integral_constant<int, 33>{});
            gap(std::integral_constant<int, 34>{});
            gap(std::integral_constant<int, 35>{});
        }
        // ---- the output accumulators follow the new maximum before the next P V (rare once the maximum has settled)
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
        }
    };

#ifdef NP_PRIO_YOUNG   /* experiment: static priority for the younger half of the workgroup */
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    int s_cur = 0, s_prv = G::NSLOT - 1, s_nxt = 1;                 // ring slots of tiles t, t - 1, t + 1
    {   // S^T of the first half (plain: nothing to overlap with yet)
        const unsigned char* const kah = sm + xl * NF_KROW + kh * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[0][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(kah + ks * 32);
            st[0] = NP_MFMA(ah, qh[ks], st[0]);
            if constexpr (SPLIT) {
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(kah + G::KLO_OFF + ks * 32);
                st[0] = NP_MFMA(ah, ql[ks], st[0]);
                st[0] = NP_MFMA(al, qh[ks], st[0]);
            }
        }
    }
    NP_PHASE(1);                                                    // query operands, first tiles: the loop starts
    for (int kt = kt0; kt < kt1; ++kt) {
        NP_STAMP();                                                 // 0
        const unsigned char* const cur = sm + s_cur * G::SLOT_BYTES;
        const unsigned char* const prv = sm + s_prv * G::SLOT_BYTES;
        const unsigned char* const nxt = sm + s_nxt * G::SLOT_BYTES;
        // ---- a: softmax(2t) | S^T(2t + 1) | P V(2t - 1)
        half_body(std::integral_constant<int, 0>{}, cur, prv, 0, 0);
        NP_STAMP();                                                 // 1
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G::NSLOT - 3) * G::PW) : "memory");   // this wave's pieces of tile t + 1 have landed
        NF_BARRIER();                                               // tile t + 1 is complete; nobody reads tile t - 1 any more
        NP_STAMP();                                                 // 2
        // ---- b: softmax(2t + 1) | S^T(2t + 2) | P V(2t); the pieces of tile t + NSLOT - 1 -> the slot of tile t - 1
        // (S^T past the last tile: an all-zero tile, the result is not used)
        half_body(std::integral_constant<int, 1>{}, nxt, cur, kt + G::NSLOT - 1, s_prv);
        NP_STAMP();                                                 // 3
        s_prv = s_cur;
        s_cur = s_nxt;
        s_nxt = s_nxt + 1 == G::NSLOT ? 0 : s_nxt + 1;
    }
    NP_PHASE(2);                                                    // the loop is through
    {   // the last half's P V
        const unsigned char* const vah = sm + s_prv * G::SLOT_BYTES + G::VHI_OFF + xl * NF_VROW + kh * 16 + 64;
        [[maybe_unused]] const unsigned char* const vtr = sm + s_prv * G::SLOT_BYTES + (32 + 4 * kh + ((lane & 15) >> 2)) * NF_KROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const bf16x8 vh = TR ? nf_tr_operand(vtr + jj * 16 * NF_KROW + ct * 64) : *reinterpret_cast<const bf16x8*>(vah + ct * 32 * NF_VROW + jj * 32);
                o[ct] = NP_MFMA(vh, pt[1][jj], o[ct]);
                if constexpr (SPLIT) {
                    const bf16x8 vl = TR ? nf_tr_operand(vtr + G::KLO_OFF + jj * 16 * NF_KROW + ct * 64)
                                         : *reinterpret_cast<const bf16x8*>(vah + (G::VLO_OFF - G::VHI_OFF) + ct * 32 * NF_VROW + jj * 32);
                    o[ct] = NP_MFMA(vh, pl[1][jj], o[ct]);
                    o[ct] = NP_MFMA(vl, pt[1][jj], o[ct]);
                }
            }
    }

    float l = o[LCT][LR];
    {
        const float lo = __shfl_xor(l, 32);
        if (kh != LKH) l = lo;
    }
    // V carries 2^7; l and O share the 2^14 of P.  O is normalised by its row sum for the projection in EITHER case - a key split's partial sums reach
    // 2^14 x keys, beyond binary16 - and a partial result gets its l back behind the projection (nl_merge_kernel weighs the parts with their (m, l))
    const float inv = (1.0f / NF_XSCALE) / l;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] *= inv;

    // Z^T = W'^T O^T (the folded 1x1 projections, model/pfnl.py:58 -> utils.py:26,67) in the arithmetic of the rest of the kernel: both operands split
    // into binary16 pairs (lo' = (v - hi) 2^11), 3 v_mfma_f32_32x32x16_f16 per product block, cross terms in a second accumulator.  (Until round 6 this was
    // 16 v_mfma_f32_32x32x2_f32 per 32 x 32 x 32 block with W' fetched by every lane: 144 MFMAs of 64 cycles and 36 KB of loads PER WAVE - measured, with
    // the projection removed, at 9 - 16 us of every launch whatever its size.)  W' goes through LDS once per workgroup, in operand form: pair g = (cot, ct, t)
    // = the A operand of k-step t of channel tile ct for output tile cot, lane (m, kh) element i = W'[32 ct + 16 t + 4 kh + (i & 3) + 8 (i >> 2)][32 cot + m] -
    // the channel order registers 8 t .. 8 t + 7 of an O^T accumulator hold (the P^T trick of the loop above), so O^T is the B operand as it stands.
    constexpr int CTW = CP / 32;
    constexpr int NPAIR = CTW * CTW * 2;
    static_assert(NPAIR * 2048 <= G::SLOT_BYTES * G::NSLOT, "W' in operand form fits the ring");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the ring's last (empty) requests have landed
    __syncthreads();                                                // every wave is through its last P V: the ring is free
    NP_PHASE(3);
    for (int pp = tid; pp < NPAIR * 64; pp += NF_THREADS) {
        const int ln = pp & 63, g = pp >> 6;
        const int t = g & 1, ctp = (g >> 1) % CTW, cotp = (g >> 1) / CTW;
        const float* w = Wp + (size_t)(ctp * 32 + 16 * t + 4 * (ln >> 5)) * CP + cotp * 32 + (ln & 31);
        bf16x8 h, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = w[((i & 3) + 8 * (i >> 2)) * CP];
            const _Float16 hh = (_Float16)v;
            h[i] = hh;
            l[i] = (_Float16)((v - (float)hh) * 2048.0f);
        }
        *reinterpret_cast<bf16x8*>(sm + g * 2048 + ln * 16) = h;
        *reinterpret_cast<bf16x8*>(sm + g * 2048 + 1024 + ln * 16) = l;
    }
    bf16x8 bh[CTW][2], bl[CTW][2];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = o[ct][8 * t + i];
                const _Float16 hh = (_Float16)v;
                bh[ct][t][i] = hh;
                bl[ct][t][i] = (_Float16)((v - (float)hh) * 2048.0f);
            }
    __syncthreads();
    NP_PHASE(4);                                                    // W' is in LDS
#pragma unroll
    for (int cot = 0; cot < CTW; ++cot) {
        f32x16 z, zc;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = zc[r] = 0.f;
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned char* wa = sm + ((cot * CTW + ct) * 2 + t) * 2048 + lane * 16;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(wa), al = *reinterpret_cast<const bf16x8*>(wa + 1024);
                z = NP_MFMA(ah, bh[ct][t], z);
                zc = NP_MFMA(ah, bl[ct][t], zc);
                zc = NP_MFMA(al, bh[ct][t], zc);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = (z[r] + zc[r] * (1.0f / 2048.0f)) * (ksp == 1 ? 1.0f : l);
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
    NP_PHASE(5);                                                    // projection and stores issued
}

size_t nl_f16_scratch_halfs(int B, int N) {                        // Khi, Klo, Vthi, Vtlo
    const size_t npad = (size_t)(N + 31) / 32 * 32 + 64;            // + one tile of slack for the last tile's V^T pieces
    return 4 * (size_t)B * NF_CP * npad;
}

// X, Xo as in launch_nl_attn; scratch16: nl_f16_scratch_halfs(B, N) 16-bit elements; partial: nl_partial_floats
static size_t nl_f16_chunk_limit() {                              // bytes of packed operands one launch may address (testing hook: PFNL_NL_CHUNK_BYTES)
    static const size_t lim = [] {
        const char* e = std::getenv("PFNL_NL_CHUNK_BYTES");
        const long long v = e ? std::atoll(e) : 0;
        return v > 0 ? (size_t)v : (size_t)0x7fff0000ull;
    }();
    return lim;
}

static hipError_t nl_attn_f16_run(const float* X, float* Xo, const float* Wp, const float* bp, float* partial, uint16_t* scratch16,
                                  int B, int N, int C, hipStream_t s, int q0, int q1, bool split, int ks_cap, bool prepacked = false);

bool nl_f16_fits_one_launch(int B, int N) { return 2 * nl_f16_scratch_halfs(1, N) * (size_t)B <= nl_f16_chunk_limit(); }

// x [B][T][H][W][3] -> X [B][N][CP] fp32 AND the binary16 operand arrays of nl_attn_f16_sw_kernel in scratch16, in ONE launch (the batch must
// fit one launch: nl_f16_fits_one_launch); launch_nl_attn_f16(..., prepacked = true) then skips its own pack
hipError_t launch_nl_pack_fused(const float* x, float* X, uint16_t* scratch16, int B, int T, int H, int W, hipStream_t s) {
    const int C = 12 * T, N = (H / 2) * (W / 2);
    if (!x || !X || !scratch16 || B < 1 || (C != 84 && C != 60 && C != 36) || (H & 1) || (W & 1) || !nl_f16_fits_one_launch(B, N)) return hipErrorInvalidValue;
    const int CP = nl_padded_ch(C);
    const int npad = (N + 31) / 32 * 32 + 64;
    uint16_t* Khi = scratch16;
    uint16_t* Klo = Khi + (size_t)B * npad * NF_CP;
    uint16_t* Vthi = Klo + (size_t)B * npad * NF_CP;
    uint16_t* Vtlo = Vthi + (size_t)B * NF_CP * npad;
    hipLaunchKernelGGL(nl_pack_f16_kernel<true>, dim3(B * (npad / 32)), dim3(256), 0, s, nullptr, Khi, Klo, Vthi, Vtlo, B, N, npad, C, CP, x, X, H, W);
    return hipGetLastError();
}

hipError_t launch_nl_attn_f16(const float* X, float* Xo, const float* Wp, const float* bp, float* partial, uint16_t* scratch16,
                               int B, int N, int C, hipStream_t s, int q0, int q1, bool split, bool prepacked) {
    if (B < 1 || N < 1) return hipErrorInvalidValue;
    if (prepacked) {                                                  // launch_nl_pack_fused filled X and scratch16 for the whole batch
        if (!nl_f16_fits_one_launch(B, N)) return hipErrorInvalidValue;
        return nl_attn_f16_run(X, Xo, Wp, bp, partial, scratch16, B, N, C, s, q0, q1, split, nl_key_splits(B, N), true);
    }
    // The kernel reaches its packed operands through one buffer resource with 32-bit offsets: a batch whose operands exceed the limit runs in
    // chunks of whole clips, one after the other on the stream, through the same scratch and partial buffers (sized for the whole batch; the
    // key split is capped by what the whole batch's partial buffer was sized for).  A single clip never reaches it (N < 2.1 M keys by the
    // forward's own size limit).
    const size_t per_clip = 2 * nl_f16_scratch_halfs(1, N);
    const int ks_cap = nl_key_splits(B, N);
    int bc = (int)std::min<size_t>((size_t)B, std::max<size_t>(1, nl_f16_chunk_limit() / per_clip));
    const int CP = nl_padded_ch(C);
    for (int b0 = 0; b0 < B; b0 += bc) {
        const int nb = std::min(bc, B - b0);
        hipError_t e = nl_attn_f16_run(X + (size_t)b0 * N * CP, Xo + (size_t)b0 * N * CP, Wp, bp, partial, scratch16, nb, N, C, s, q0, q1, split, ks_cap);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static hipError_t nl_attn_f16_run(const float* X, float* Xo, const float* Wp, const float* bp, float* partial, uint16_t* scratch16,
                                  int B, int N, int C, hipStream_t s, int q0, int q1, bool split, int ks_cap, bool prepacked) {
    if (C != 84 && C != 60 && C != 36) return hipErrorInvalidValue;
    if (q1 < 0) q1 = N;
    if (q0 < 0 || q0 >= q1 || q1 > N) return hipErrorInvalidValue;
    const int CP = nl_padded_ch(C);
    const int npad = (N + 31) / 32 * 32 + 64;
    uint16_t* Khi = scratch16;
    uint16_t* Klo = Khi + (size_t)B * npad * NF_CP;
    uint16_t* Vthi = Klo + (size_t)B * npad * NF_CP;
    uint16_t* Vtlo = Vthi + (size_t)B * NF_CP * npad;
    if (!prepacked) {
        hipLaunchKernelGGL(nl_pack_f16_kernel<false>, dim3(B * (npad / 32)), dim3(256), 0, s, X, Khi, Klo, Vthi, Vtlo, B, N, npad, C, CP, nullptr, nullptr, 0, 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // key splits: this kernel runs 1 workgroup per CU (LDS), so the grid should fill a whole number of 256-workgroup
    // rounds: time ~ ceil(query blocks * B * ks / 256) / ks.  (1080p: 127 blocks -> ks = 2 is one full round.)  Bounded
    // by the fp32 kernel's choice, which sized the partial-result buffer.
    const int ks_max = std::min(nl_key_splits(B, N), ks_cap);
    int ks = 1;
    {
        const long long qb = (long long)((q1 - q0 + NF_QB - 1) / NF_QB) * B;
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
    dim3 grid((q1 - q0 + NF_QB - 1) / NF_QB, B, ks);
    dim3 block(NF_THREADS);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    // the kernel addresses the four packed arrays as ONE buffer resource with 32-bit offsets and an out-of-range sentinel of 2^31 - 1:
    // a batch whose packed operands reach 2 GB (> 2.7 M keys) runs in clip chunks (see below)
    const size_t scratch_bytes = 2 * nl_f16_scratch_halfs(B, N);
    // TR (round 6): the P V operands out of the K tile by ds_read_b64_tr_b16 - no V^T arrays in the tile (PFNL_NL_TR=0: the form with V^T tiles)
    static const bool use_tr = [] {
        const char* e = std::getenv("PFNL_NL_TR");
        return NF_TR_INSTANCES && e && e[0] != '0';
    }();
    static std::atomic<int> attr_sw[64];
    if (!attr_sw[dev]) {
        const void* const fns[12] = {
            reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<84, true, false>), reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<60, true, false>),
            reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<36, true, false>), reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<84, false, false>),
            reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<60, false, false>), reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<36, false, false>),
#if NF_TR_INSTANCES
            reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<84, true, true>), reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<60, true, true>),
            reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<36, true, true>), reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<84, false, true>),
            reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<60, false, true>), reinterpret_cast<const void*>(nl_attn_f16_sw_kernel<36, false, true>)
#endif
        };
        const int lds[4] = {NfSW<true, false>::LDS_BYTES, NfSW<false, false>::LDS_BYTES, NfSW<true, true>::LDS_BYTES, NfSW<false, true>::LDS_BYTES};
        for (int i = 0; i < (NF_TR_INSTANCES ? 12 : 6); ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, lds[i / 3]);
            if (e != hipSuccess) return e;
        }
        attr_sw[dev] = 1;
    }
    const unsigned rel_klo = (unsigned)((Klo - Khi) * 2), rel_vhi = (unsigned)((Vthi - Khi) * 2), rel_vlo = (unsigned)((Vtlo - Khi) * 2);
#define NP_LAUNCH(C_, S_, T_) hipLaunchKernelGGL((nl_attn_f16_sw_kernel<C_, S_, T_>), grid, block, (NfSW<S_, T_>::LDS_BYTES), s, X, Khi, rel_klo, rel_vhi, rel_vlo, \
                                                 (unsigned)scratch_bytes, Xo, Wp, bp, Zp, ML, N, npad, q0, q1)
#if NF_TR_INSTANCES
#define NP_LAUNCH2(C_, S_) do { if (use_tr) NP_LAUNCH(C_, S_, true); else NP_LAUNCH(C_, S_, false); } while (0)
#else
#define NP_LAUNCH2(C_, S_) do { (void)use_tr; NP_LAUNCH(C_, S_, false); } while (0)
#endif
    switch (C) {
        case 84: if (split) NP_LAUNCH2(84, true); else NP_LAUNCH2(84, false); break;
        case 60: if (split) NP_LAUNCH2(60, true); else NP_LAUNCH2(60, false); break;
        case 36: if (split) NP_LAUNCH2(36, true); else NP_LAUNCH2(36, false); break;
    }
#undef NP_LAUNCH2
#undef NP_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ks == 1) return e;
    return launch_nl_merge(X, Zp, ML, bp, Xo, B, N, C, ks, s, q0, q1);
}

}  // namespace pfnl

#ifdef PFNL_NP_TIMING
extern "C" int pfnl_debug_read_np_stamps(long long* host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pfnl::np_dbg), n * sizeof(long long)) == hipSuccess ? 0 : -1;
}
extern "C" int pfnl_debug_read_np_phases(long long* host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pfnl::np_ph), n * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

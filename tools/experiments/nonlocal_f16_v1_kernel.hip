// nl_attn_f16_kernel - the first generation of the split-f16 non-local kernel (rounds 2 - 4), removed from the product library in round 5:
// since round 4 it was reachable only through PFNL_NL_SW=0 and for batches whose packed operands exceed 2 GB; the latter are now run
// through nl_attn_f16_sw_kernel in clip chunks (launch_nl_attn_f16).  A fragment of pfnl_amd/csrc/nonlocal_f16.hip as of commit 947d47b
// (constants NF_*, nf helpers and the launcher's NF_LAUNCH arm are there); kept for the record of what the half-tile-skewed form looked like.
// SPLIT = true: the fp32 path (operands as hi + lo, 72 MFMAs per 64 keys).  SPLIT = false: the same kernel on the hi parts only
// (24 MFMAs per 64 keys) - 16-bit operands, fp32 accumulation: the non-local block of precision=bf16, whose trunk is 16-bit
// anyway (binary16 has 3 more mantissa bits than the bf16 of the trunk: logits good to ~2e-3, where bf16 logits are off by 16 %).
template <int C, bool SPLIT>
__global__ __launch_bounds__(NF_THREADS, 2) void nl_attn_f16_kernel(const float* __restrict__ X, const uint16_t* __restrict__ Khi,
                                                              const uint16_t* __restrict__ Klo, const uint16_t* __restrict__ Vthi,
                                                              const uint16_t* __restrict__ Vtlo, float* __restrict__ Xo,
                                                              const float* __restrict__ Wp, const float* __restrict__ bp,
                                                              float* __restrict__ Zp, float* __restrict__ ML, int N, int Npad, int q0, int q1) {
    constexpr int CT = 3;
    constexpr int CP = (C + 31) / 32 * 32;                          // row stride of X / Xo / Wp (nl_padded_ch)
    static_assert(C < NF_CP && C % 2 == 0, "needs a pad channel inside 96");
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];   // three tiles: K hi | K lo | V^T hi | V^T lo

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int xl = lane & 31;
    const int kh = lane >> 5;
    const int b = blockIdx.y;
    const float* Xb = X + (size_t)b * N * CP;
    float* Xob = Xo + (size_t)b * N * CP;
    const int q = q0 + blockIdx.x * NF_QB + wave * 32 + xl;         // this lane's query (queries [q0, q1): a strip of the frame)
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
            const float v = c < C ? Xb[(size_t)qc * CP + c] * (LOG2E * NF_XSCALE) : 0.f;
            const _Float16 h = (_Float16)v;
            qh[ks][e] = h;
            if constexpr (SPLIT) ql[ks][e] = (_Float16)(v - (float)h);
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
    const uint16_t* const Khb = Khi + (size_t)b * Npad * NF_CP;
    const uint16_t* const Klb = Klo + (size_t)b * Npad * NF_CP;
    const uint16_t* const Vhb = Vthi + (size_t)b * NF_CP * Npad;
    const uint16_t* const Vlb = Vtlo + (size_t)b * NF_CP * Npad;
    constexpr int NI = 2;
    u32x4 rk[4 * NI];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int id = min(tid + i * NF_THREADS, 767);          // 0..767 (surplus threads redo the last piece)
            const int key = id / 12, c16 = id - key * 12;
            const bool ok = k0 + key < N;
            const size_t ko = ((size_t)(k0 + (ok ? key : 0)) * NF_CP + c16 * 8);
            rk[i] = ok ? *reinterpret_cast<const u32x4*>(Khb + ko) : u32x4{0, 0, 0, 0};
            if constexpr (SPLIT) rk[NI + i] = ok ? *reinterpret_cast<const u32x4*>(Klb + ko) : u32x4{0, 0, 0, 0};
            const int ch = id >> 3, kc = id & 7;                    // V^T: 96 rows x 8 pieces (k0 + 64 <= Npad + 32: rows are padded)
            const bool vok = k0 + kc * 8 < Npad;
            const size_t vo = (size_t)ch * Npad + k0 + (vok ? kc * 8 : 0);
            rk[2 * NI + i] = vok ? *reinterpret_cast<const u32x4*>(Vhb + vo) : u32x4{0, 0, 0, 0};
            if constexpr (SPLIT) rk[3 * NI + i] = vok ? *reinterpret_cast<const u32x4*>(Vlb + vo) : u32x4{0, 0, 0, 0};
        }
    };
    auto store_tile = [&](unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int id = min(tid + i * NF_THREADS, 767);
            const int key = id / 12, c16 = id - key * 12;
            *reinterpret_cast<u32x4*>(buf + key * NF_KROW + c16 * 16) = rk[i];
            if constexpr (SPLIT) *reinterpret_cast<u32x4*>(buf + NF_KT * NF_KROW + key * NF_KROW + c16 * 16) = rk[NI + i];
            const int ch = id >> 3, kc = id & 7;
            *reinterpret_cast<u32x4*>(buf + 2 * NF_KT * NF_KROW + ch * NF_VROW + kc * 16) = rk[2 * NI + i];
            if constexpr (SPLIT) *reinterpret_cast<u32x4*>(buf + 2 * NF_KT * NF_KROW + NF_CP * NF_VROW + ch * NF_VROW + kc * 16) = rk[3 * NI + i];
        }
    };

    const int ntiles = (N + NF_KT - 1) / NF_KT;
    const int ksp = gridDim.z, sp = blockIdx.z;
    const int kt0 = (int)((long long)ntiles * sp / ksp), kt1 = (int)((long long)ntiles * (sp + 1) / ksp);
    load_tile(kt0 * NF_KT);
    store_tile(sm);
    if (kt0 + 1 < kt1) load_tile((kt0 + 1) * NF_KT);
    __syncthreads();
    // The two waves of a SIMD are half a tile apart: waves 0-3 run S^T, softmax, P V of tile t; waves 4-7 run P V of
    // tile t-1 (its P^T kept in registers, its V^T in the third LDS buffer), then S^T and softmax of tile t.  Next to
    // a wave that keeps the matrix pipe busy a partner's VALU gets one issue slot per MFMA (tools/ubench) - phase-aligned,
    // the two waves' softmax blocks (150 VALU, 32 of them quarter-rate v_exp_f32) simply add to the MFMA time; skewed,
    // and with the softmax at raised priority, one wave's VALU runs under the other's MFMAs.
    const bool late = wave >= 4;
    bf16x8 pt[2][2];                                                // P^T (hi, lo parts) of the tile whose P V is still to come
    [[maybe_unused]] bf16x8 pl[2][2];
    bf16x8 ob[2][6];                                                // operands one MFMA step ahead (the compiler alone issues each
                                                                    // ds_read right in front of its MFMA: 60 LDS latencies per tile)
#define NF_LOAD_QK(ks_, d_)                                                                                     \
    do {                                                                                                        \
        ob[d_][0] = *reinterpret_cast<const bf16x8*>(kah + (ks_) * 32);                                         \
        ob[d_][1] = *reinterpret_cast<const bf16x8*>(kah + 32 * NF_KROW + (ks_) * 32);                          \
        if constexpr (SPLIT) {                                                                                  \
            ob[d_][2] = *reinterpret_cast<const bf16x8*>(kal + (ks_) * 32);                                     \
            ob[d_][3] = *reinterpret_cast<const bf16x8*>(kal + 32 * NF_KROW + (ks_) * 32);                      \
        }                                                                                                       \
    } while (0)
#define NF_LOAD_PV(vah_, val_, j_, d_)                                                                          \
    do {                                                                                                        \
        _Pragma("unroll") for (int ct_ = 0; ct_ < CT; ++ct_) {                                                  \
            ob[d_][ct_] = *reinterpret_cast<const bf16x8*>((vah_) + ct_ * 32 * NF_VROW + ((j_) >> 1) * 64 + ((j_) & 1) * 32);     \
            if constexpr (SPLIT) ob[d_][3 + ct_] = *reinterpret_cast<const bf16x8*>((val_) + ct_ * 32 * NF_VROW + ((j_) >> 1) * 64 + ((j_) & 1) * 32); \
        }                                                                                                       \
    } while (0)
    // O^T[ch][query] += V^T[ch][keys] P^T[keys][query] for the tile in `buf`, keys in the accumulator's own order;
    // channel tile innermost (consecutive MFMAs go to different accumulators)
#define NF_PV(buf_)                                                                                             \
    do {                                                                                                        \
        const unsigned char* const vah_ = (buf_) + 2 * NF_KT * NF_KROW + xl * NF_VROW + kh * 16;                \
        const unsigned char* const val_ = vah_ + NF_CP * NF_VROW;                                               \
        NF_LOAD_PV(vah_, val_, 0, 0);                                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            if (j_ < 3) NF_LOAD_PV(vah_, val_, j_ + 1, (j_ + 1) & 1);                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            _Pragma("unroll") for (int ct_ = 0; ct_ < CT; ++ct_)                                                \
                o[ct_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ob[j_ & 1][ct_], pt[j_ >> 1][j_ & 1], o[ct_], 0, 0, 0);     \
            if constexpr (SPLIT) {                                                                              \
                _Pragma("unroll") for (int ct_ = 0; ct_ < CT; ++ct_)                                            \
                    o[ct_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ob[j_ & 1][3 + ct_], pt[j_ >> 1][j_ & 1], o[ct_], 0, 0, 0); \
                _Pragma("unroll") for (int ct_ = 0; ct_ < CT; ++ct_)                                            \
                    o[ct_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ob[j_ & 1][ct_], pl[j_ >> 1][j_ & 1], o[ct_], 0, 0, 0);     \
            }                                                                                                   \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)

    for (int kt = kt0; kt < kt1; ++kt) {
        const int bi = (kt - kt0) % 3;
        unsigned char* const cur = sm + bi * NF_TILE_BYTES;
        unsigned char* const nxt = sm + (bi == 2 ? 0 : bi + 1) * NF_TILE_BYTES;   // held tile kt-2: read by nobody any more
        const unsigned char* const prv = sm + (bi == 0 ? 2 : bi - 1) * NF_TILE_BYTES;
        if (late && kt > kt0) NF_PV(prv);
        // S^T for both 32-key halves of the tile (two independent accumulators, alternating: no MFMA waits for its
        // predecessor), then ONE running-max / rescale update for the 64 keys
        const int kbase = kt * NF_KT;
        f32x16 st[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[sub][r] = 0.f;
        const unsigned char* const kah = cur + xl * NF_KROW + kh * 16;
        const unsigned char* const kal = kah + NF_KT * NF_KROW;
        NF_LOAD_QK(0, 0);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
            if (ks < 5) NF_LOAD_QK(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int d = ks & 1;
            st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ob[d][0], qh[ks], st[0], 0, 0, 0);
            st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ob[d][1], qh[ks], st[1], 0, 0, 0);
            if constexpr (SPLIT) {
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ob[d][0], ql[ks], st[0], 0, 0, 0);
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ob[d][1], ql[ks], st[1], 0, 0, 0);
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ob[d][2], qh[ks], st[0], 0, 0, 0);
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ob[d][3], qh[ks], st[1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(2);                              // the softmax VALU goes ahead of the partner wave's MFMAs
        if (kbase + NF_KT > N) {                                    // wave-uniform: only the last, partial key tile
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + sub * 32 + drow(r, lane) >= N) st[sub][r] = -INFINITY;
        }
        float tmax = fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[1][0], st[1][1]));
#pragma unroll
        for (int r = 2; r < 16; r += 2) tmax = fmaxf(tmax, fmaxf(fmaxf(st[0][r], st[0][r + 1]), fmaxf(st[1][r], st[1][r + 1])));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32)) * NF_SINV;            // true base-2 logit (the scale is positive: max commutes)
        const float mn = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);         // m = -inf on the first tile -> 0
#ifdef NF_X_NOSOFTMAX   /* timing experiment only: wrong results */
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; r += 8) pl[sub][r >> 3] = pt[sub][r >> 3] = __builtin_bit_cast(bf16x8, u32x4{__builtin_bit_cast(unsigned, st[sub][r]), __builtin_bit_cast(unsigned, st[sub][r + 1]), __builtin_bit_cast(unsigned, st[sub][r + 2]), __builtin_bit_cast(unsigned, st[sub][r + 3])});
#else
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[sub][r], NF_SINV, NF_PSHIFT - mn));   // 2^14 exp2(s - max)
                const _Float16 ph = (_Float16)pv;
                pt[sub][r >> 3][r & 7] = ph;
                if constexpr (SPLIT) pl[sub][r >> 3][r & 7] = (_Float16)(pv - (float)ph);
            }
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
        if (!late) NF_PV(cur);
        if (kt + 1 < kt1) {                                         // next tile (requested a tile ago) -> the third buffer; the tile
            store_tile(nxt);                                        // after it requested
            if (kt + 2 < kt1) load_tile((kt + 2) * NF_KT);
        }
        __syncthreads();                                            // this tile's S^T operands are free, the next tile is complete
    }
    if (late) NF_PV(sm + ((kt1 - 1 - kt0) % 3) * NF_TILE_BYTES);    // the late waves' last P V
#undef NF_PV
#undef NF_LOAD_PV
#undef NF_LOAD_QK

    float l = o[LCT][LR];
    {
        const float lo = __shfl_xor(l, 32);
        if (kh != LKH) l = lo;
    }
    const float inv = (ksp == 1) ? (1.0f / NF_XSCALE) / l : (1.0f / NF_XSCALE);   // V carries 2^7; l and O share the 2^14 of P
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

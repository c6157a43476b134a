// Non-local spatio-temporal correlation block (gfx950, f32 MFMA, flash-style streaming softmax).
//
// Replaces utils.NonLocalBlock(nltype=1, sub_sample=1) of the reference (utils.py:18-71) together
// with the gather/scatter around it in PFNL.forward (model/pfnl.py:55-60):
//   X  = space_to_depth(concat_t x_t, 2)            [B, N=(H/2)(W/2), C=12T]
//   P  = exp(X X^T) / rowsum                        (utils.py:53-58; theta = phi = X at nltype 1)
//   Z  = (P (X Wg + bg)) Ww + bw                    (utils.py:26,64,67)
//      = (P X) (Wg Ww) + (bg Ww + bw)               (rows of P sum to 1; W' = Wg Ww folded on host)
//   out = X + Z   kept in the space_to_depth ("packed") layout; conv0 reads it through the
//                 depth_to_space index map, so model/pfnl.py:59-61 costs no pass over memory.
// The N x N affinity is never materialised: one wave owns 32 queries, streams 32-key tiles from an
// LDS-staged X tile, keeps a running max/sum per query (mathematically identical to the
// reference's un-stabilised exp/sum/divide wherever that does not overflow) and accumulates
// O^T = X^T P^T with MFMA.  Everything is computed transposed (keys/channels on the MFMA row axis,
// queries on the column axis = lane) so that softmax state is per-lane and the probabilities feed
// the second MFMA straight from the accumulator registers of the first.
#include "common.h"

namespace pfnl {

int nl_padded_ch(int C) { return 32 * ((C + 31) / 32); }

// x [B,T,H,W,3] -> X [B,N,CP]; channel (dy*2+dx)*3T + 3t + c (model/pfnl.py:55-57); pad columns = 0.
__global__ void nl_pack_kernel(const float* __restrict__ x, float* __restrict__ X, int B, int T, int H,
                               int W, int CP) {
    const int C3 = 3 * T;
    const int W2 = W / 2;
    const size_t N = (size_t)(H / 2) * W2;
    const size_t total = (size_t)B * N * CP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % CP);
        const size_t n = (i / CP) % N;
        const int b = (int)(i / ((size_t)CP * N));
        float v = 0.f;
        if (co < 4 * C3) {
            const int sub = co / C3, k = co % C3;
            const int t = k / 3, c = k % 3;
            const int y = 2 * (int)(n / W2) + (sub >> 1);
            const int xx = 2 * (int)(n % W2) + (sub & 1);
            v = x[((((size_t)b * T + t) * H + y) * W + xx) * 3 + c];
        }
        X[i] = v;
    }
}

// packed [B,N,CP] -> [B,H,W,3T] (depth_to_space, model/pfnl.py:59); used by the debug tap / op hook.
__global__ void nl_unpack_kernel(const float* __restrict__ Xo, float* __restrict__ out, int B, int T,
                                 int H, int W, int CP) {
    const int C3 = 3 * T;
    const int W2 = W / 2;
    const size_t total = (size_t)B * H * W * C3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % C3);
        const int xx = (int)((i / C3) % W);
        const int y = (int)((i / ((size_t)C3 * W)) % H);
        const int b = (int)(i / ((size_t)C3 * W * H));
        const size_t n = (size_t)(y >> 1) * W2 + (xx >> 1);
        const int co = ((y & 1) * 2 + (xx & 1)) * C3 + k;
        out[i] = Xo[((size_t)b * (H / 2) * W2 + n) * CP + co];
    }
}

constexpr int NL_KT = 64;   // keys per LDS tile (two 32-key MFMA sub-tiles)

// One workgroup = 4 waves x 32 queries.  C = 12T real channels, CT = ceil(C/32) channel tiles.
// DOT (utils.py:59-62, nltype 2): P = relu(theta phi^T) / rowsum instead of exp(.) / rowsum - no running maximum, the logits get
// the per-query constant theta_i . b_phi (column C of Q: it does NOT cancel under the relu) added before the clamp.
// Kx / Nk: keys = values (sub_sample > 1: the average-pooled X, utils.py:27-28,35-36 - pooling commutes with the 1x1 convs).
template <int C, bool DOT>
__global__ __launch_bounds__(256, 2) void nl_attn_kernel(const float* __restrict__ X,
                                                         const float* __restrict__ Kx, int Nk,
                                                         float* __restrict__ Xo,
                                                         const float* __restrict__ Wp,   // [CP][CP], row = ci
                                                         const float* __restrict__ bp,   // [CP]
                                                         float* __restrict__ Zp,         // [B][ks][N][CP] partial W'^T O (ks > 1)
                                                         float* __restrict__ ML,         // [B][ks][N][2]  partial (max, sum)
                                                         const float* __restrict__ Q,    // [B][N][CP] queries (nltype 0: X M + c), or X
                                                         int N, int q0, int q1) {   // queries [q0, q1) only (a strip of the frame)
    constexpr int CT = (C + 31) / 32;
    constexpr int CP = CT * 32;
    constexpr int LS = CP + 1;            // odd LDS row stride: key rows land on distinct banks
    constexpr int KSTEPS = C / 2;
    static_assert(C % 2 == 0, "channel count must be even");
    __shared__ float sk[NL_KT * LS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int xl = lane & 31;
    const int kh = lane >> 5;
    const int b = blockIdx.y;
    const float* Xb = X + (size_t)b * N * CP;
    const float* Kb = Kx + (size_t)b * Nk * CP;
    float* Xob = Xo + (size_t)b * N * CP;
    const int q = q0 + blockIdx.x * 128 + wave * 32 + xl;  // this lane's query
    const int qc = q < q1 ? q : q1 - 1;

    // B operand of S^T = Xk Xq^T: lane holds Xq[q][2s + kh], pre-scaled by log2(e) so that the logits come out
    // of the MFMA in base-2 units and the softmax needs one v_exp_f32 per element (expf expands to ~8 VALU, and
    // VALU issued between a wave's MFMAs is matrix-pipe time: tools/ubench).
    constexpr float LOG2E = 1.4426950408889634f;
    float bq[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) bq[s] = Q[((size_t)b * N + qc) * CP + 2 * s + kh] * (DOT ? 1.0f : LOG2E);
    [[maybe_unused]] const float dq = DOT ? Q[((size_t)b * N + qc) * CP + C] : 0.f;
    // The running sum l of the probabilities is not kept in VALU: pad channel C of the key tile in LDS is set to 1,
    // so row C of O^T = V^T P^T accumulates sum_k P (and is rescaled with O).  Where that row lives in the D layout:
    constexpr int LCT = C / 32, LI = C % 32;
    constexpr int LKH = (LI % 8) >= 4 ? 1 : 0, LR = (LI / 8) * 4 + (LI % 8) % 4;
    static_assert(C < CP, "needs a pad channel");

    f32x16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m = -INFINITY;   // running max of the (base-2) logits of this lane's query

    constexpr int TILE_F4 = NL_KT * CP / 4;
    constexpr int LD_ITERS = (TILE_F4 + 255) / 256;
    f32x4 rk[LD_ITERS];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LD_ITERS; ++i) {
            const int it = tid + i * 256;
            rk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (it < TILE_F4) {
                const int row = it / (CP / 4), c4 = it % (CP / 4);
                if (k0 + row < Nk)
                    rk[i] = *reinterpret_cast<const f32x4*>(Kb + (size_t)(k0 + row) * CP + c4 * 4);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < LD_ITERS; ++i) {
            const int it = tid + i * 256;
            if (it < TILE_F4) {
                const int row = it / (CP / 4), c4 = it % (CP / 4);
                float* d = sk + row * LS + c4 * 4;
                d[0] = rk[i].x;
                d[1] = rk[i].y;
                d[2] = rk[i].z;
                d[3] = rk[i].w;
                if (c4 == C / 4) d[C % 4] = 1.0f;          // the "ones" channel (see LCT / LR above)
            }
        }
    };

    // key split: workgroup z of gridDim.z handles key tiles [kt0, kt1) (flash-decoding style); the
    // partial results are merged by nl_merge_kernel.  gridDim.z == 1: everything here, final output.
    const int ntiles = (Nk + NL_KT - 1) / NL_KT;
    const int ks = gridDim.z, sp = blockIdx.z;
    const int kt0 = (int)((long long)ntiles * sp / ks), kt1 = (int)((long long)ntiles * (sp + 1) / ks);
    load_tile(kt0 * NL_KT);
    for (int kt = kt0; kt < kt1; ++kt) {
        if (kt > kt0) __syncthreads();
        store_tile();
        __syncthreads();
        if (kt + 1 < kt1) load_tile((kt + 1) * NL_KT);
#pragma unroll
        for (int sub = 0; sub < NL_KT / 32; ++sub) {
            const int kbase = kt * NL_KT + sub * 32;
            if (kbase >= Nk) break;                        // wave-uniform
            // S^T[key i][query j]: A = Xk[i = xl][c = 2s + kh] from LDS, B = bq.
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            const float* ka = sk + (sub * 32 + xl) * LS + kh;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) st = mfma32(ka[2 * s], bq[s], st);

            if constexpr (DOT) {                           // relu(f), utils.py:60; keys past the end contribute nothing
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = fmaxf(st[r] + dq, 0.f);
                if (kbase + 32 > Nk) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kbase + drow(r, lane) >= Nk) st[r] = 0.f;
                }
            } else {
            // online softmax; register r of this lane is key kbase + drow(r, lane).
            if (kbase + 32 > Nk) {                         // wave-uniform: only the last, partial key tile
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + drow(r, lane) >= Nk) st[r] = -INFINITY;
            }
            float tmax = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
#pragma unroll
            for (int r = 4; r < 16; r += 4) tmax = fmaxf(tmax, fmaxf(fmaxf(st[r], st[r + 1]), fmaxf(st[r + 2], st[r + 3])));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float mn = fmaxf(m, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);   // m = -inf on the first tile -> 0
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = __builtin_amdgcn_exp2f(st[r] - mn);
            m = mn;
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
            }
            }
            // O^T[ch i][query j] += V^T[i][key] P^T[key][j]; contraction step s uses the key that
            // accumulator register s of this lane's half already holds: key = drow(s, lane).
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float* va = sk + (sub * 32 + 4 * kh) * LS + ct * 32 + xl;
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    o[ct] = mfma32(va[((s & 3) + 8 * (s >> 2)) * LS], st[s], o[ct]);
            }
        }
    }

    // normalise (partial results stay un-normalised: the projection is linear); l = row C of O^T, held by the
    // lanes of half LKH in register LR
    float l = o[LCT][LR];
    {
        const float lo = __shfl_xor(l, 32);
        if (kh != LKH) l = lo;
    }
    const float inv = (ks == 1) ? 1.0f / l : 1.0f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] *= inv;

    // Z^T[co][query] = sum_ci W'[ci][co] O^T[ci][query]; register s of o[ct] is ci = 32ct + drow(s).
    // Pad rows/cols of W' are zero and pad channels of O^T are exactly zero (X pad columns are 0).
#pragma unroll
    for (int cot = 0; cot < CT; ++cot) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float* wa = Wp + (size_t)(ct * 32 + 4 * kh) * CP + cot * 32 + xl;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                z = mfma32(wa[((s & 3) + 8 * (s >> 2)) * CP], o[ct][s], z);
        }
        if (q < q1) {
            if (ks == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cot * 32 + drow(r, lane);
                    if (co < C) {
                        const size_t idx = (size_t)q * CP + co;
                        Xob[idx] = Xb[idx] + z[r] + bp[co];    // residual, model/pfnl.py:60
                    }
                }
            } else {
                float* zp = Zp + (((size_t)b * ks + sp) * N + q) * CP;
#pragma unroll
                for (int r = 0; r < 16; ++r) zp[cot * 32 + drow(r, lane)] = z[r];
            }
        }
    }
    if (ks > 1 && q < q1 && kh == 0) {
        float* ml = ML + (((size_t)b * ks + sp) * N + q) * 2;
        ml[0] = DOT ? 0.f : m;                            // (DOT: the partials carry no maximum - merged with weight 1)
        ml[1] = l;
    }
    // pad columns of the output are never read by conv0.
}

// out = X + sum_p e^{m_p - m} Zp_p / sum_p e^{m_p - m} l_p + b'   (merge of the key-split partials); four channels per thread (C and CP
// are multiples of 4): the weights of a query are computed once per 16 bytes moved, not once per 4
__global__ void nl_merge_kernel(const float* __restrict__ X, const float* __restrict__ Zp,
                                const float* __restrict__ ML, const float* __restrict__ bp,
                                float* __restrict__ Xo, int B, int N, int C, int CP, int ks, int q0, int q1) {
    const size_t nq = (size_t)(q1 - q0);
    const int cq = CP >> 2;
    const size_t total = (size_t)B * nq * cq;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(j % cq) * 4;
        if (co >= C) continue;
        const size_t qn = q0 + (j / cq) % nq;
        const size_t b = j / ((size_t)cq * nq);
        const size_t i = (b * N + qn) * CP + co;
        float m = -INFINITY;
        for (int p = 0; p < ks; ++p) m = fmaxf(m, ML[((b * ks + p) * N + qn) * 2]);
        f32x4 num = {0.f, 0.f, 0.f, 0.f};
        float den = 0.f;
        for (int p = 0; p < ks; ++p) {
            const float* ml = ML + ((b * ks + p) * N + qn) * 2;
            const float w = exp2f(ml[0] - m);                 // the partial maxima are base-2 logits
            const f32x4 z = *reinterpret_cast<const f32x4*>(Zp + ((b * ks + p) * N + qn) * CP + co);
            num.x = fmaf(w, z.x, num.x);
            num.y = fmaf(w, z.y, num.y);
            num.z = fmaf(w, z.z, num.z);
            num.w = fmaf(w, z.w, num.w);
            den = fmaf(w, ml[1], den);
        }
        const f32x4 x = *reinterpret_cast<const f32x4*>(X + i);    // (X, Xo, Zp: rows of CP floats from 16-byte aligned bases; bp: only float-aligned)
        *reinterpret_cast<f32x4*>(Xo + i) = f32x4{x.x + num.x / den + bp[co], x.y + num.y / den + bp[co + 1], x.z + num.z / den + bp[co + 2],
                                                  x.w + num.w / den + bp[co + 3]};
    }
}

// (Nk keys for N queries: fewer keys never give more splits, so nl_partial_floats(B, N, C) bounds the pooled-key case too)
static int nl_key_splits2(int B, int N, int Nk) {
    const int qblocks = (N + 127) / 128, ntiles = (Nk + NL_KT - 1) / NL_KT;
    int ks = (512 + qblocks * B - 1) / (qblocks * B);
    if (ks > 8) ks = 8;
    const int min_tiles = qblocks * B >= 32 ? 4 : 1;
    if (ks > ntiles / min_tiles) ks = ntiles / min_tiles;
    return ks < 1 ? 1 : ks;
}

int nl_key_splits(int B, int N) {
    const int qblocks = (N + 127) / 128, ntiles = (N + NL_KT - 1) / NL_KT;
    int ks = (512 + qblocks * B - 1) / (qblocks * B);          // aim at >= 2 workgroups per CU
    if (ks > 8) ks = 8;
    // keep >= 4 key tiles per split - except where a handful of workgroups is all there is (BASELINE.json configs[0]: 256 queries =
    // 2 workgroups walking 4 tiles of f32 MFMAs each, 40 us of pure latency): there every tile gets a workgroup of its own
    // (measured round 4: 40 -> 17 us for attention + merge at 7 x 32 x 32)
    const int min_tiles = qblocks * B >= 32 ? 4 : 1;
    if (ks > ntiles / min_tiles) ks = ntiles / min_tiles;
    return ks < 1 ? 1 : ks;
}

size_t nl_partial_floats(int B, int N, int C) {
    const int ks = nl_key_splits(B, N);
    return ks > 1 ? (size_t)B * ks * N * (nl_padded_ch(C) + 2) : 0;
}

hipError_t launch_nl_pack(const float* x, float* X, int B, int T, int H, int W, hipStream_t s) {
    const int CP = nl_padded_ch(12 * T);
    const size_t total = (size_t)B * (H / 2) * (W / 2) * CP;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(nl_pack_kernel, dim3(blocks), dim3(256), 0, s, x, X, B, T, H, W, CP);
    return hipGetLastError();
}

hipError_t launch_nl_unpack(const float* Xo, float* out, int B, int T, int H, int W, hipStream_t s) {
    const int CP = nl_padded_ch(12 * T);
    const size_t total = (size_t)B * H * W * 3 * T;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(nl_unpack_kernel, dim3(blocks), dim3(256), 0, s, Xo, out, B, T, H, W, CP);
    return hipGetLastError();
}

// Embedded-Gaussian option (utils.py nltype 0: theta = X Wt + bt, phi = X Wp + bp): the logits
//   (X_i Wt + bt) . (X_j Wp + bp) = (X_i M + c) . X_j + (terms constant in j, which cancel in the softmax over j),
//   M = Wt Wp^T, c = bt Wp^T (folded on the host in fp64) - so only the QUERIES are projected; keys and values stay X.
// Cq = C + 1 (nltype 2): column C of M / c holds Wt b_phi / bt . b_phi, i.e. Q[.][C] = theta_i . b_phi (see nl_attn_kernel<., DOT>).
__global__ void nl_qproj_kernel(const float* __restrict__ X, const float* __restrict__ M, const float* __restrict__ c,
                                float* __restrict__ Q, size_t rows, int C, int CP, int Cq) {
    const size_t total = rows * CP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % CP);
        const size_t n = i / CP;
        float acc = 0.f;
        if (co < Cq) {
            acc = c[co];
            const float* xr = X + n * CP;
            for (int ci = 0; ci < C; ++ci) acc = fmaf(xr[ci], M[(size_t)ci * CP + co], acc);
        }
        Q[i] = acc;
    }
}

hipError_t launch_nl_qproj(const float* X, const float* M, const float* c, float* Q, int B, int N, int C, hipStream_t s, bool dot_column) {
    const int CP = nl_padded_ch(C);
    const size_t total = (size_t)B * N * CP;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nl_qproj_kernel, dim3(blocks), dim3(256), 0, s, X, M, c, Q, (size_t)B * N, C, CP, dot_column ? C + 1 : C);
    return hipGetLastError();
}

hipError_t launch_nl_attn(const float* X, float* Xo, const float* Wp, const float* bp, float* partial, int B, int N,
                          int C, hipStream_t s, const float* Q, int q0, int q1) {
    return launch_nl_attn_general(X, X, N, Xo, Wp, bp, partial, B, N, C, s, Q, q0, q1, false);
}

// Kx [B][Nk][CP]: keys = values (X itself, or its average-pooled copy: launch_nl_pool); dot: the relu / rowsum form (nltype 2),
// which needs Q with the per-query constant in column C (launch_nl_qproj with Cq = C + 1).
hipError_t launch_nl_attn_general(const float* X, const float* Kx, int Nk, float* Xo, const float* Wp, const float* bp,
                                  float* partial, int B, int N, int C, hipStream_t s, const float* Q, int q0, int q1, bool dot) {
    if (dot && !Q) return hipErrorInvalidValue;
    if (!Q) Q = X;
    if (q1 < 0) q1 = N;
    if (q0 < 0 || q0 >= q1 || q1 > N || Nk < 1) return hipErrorInvalidValue;
    const int ks = nl_key_splits2(B, N, Nk);
    const int CP = nl_padded_ch(C);
    if (ks > 1 && !partial) return hipErrorInvalidValue;
    float* Zp = partial;
    float* ML = partial ? partial + (size_t)B * ks * N * CP : nullptr;
    dim3 grid((q1 - q0 + 127) / 128, B, ks);
    dim3 block(256);
#define NL_LAUNCH(C_)                                                                                                  \
    do {                                                                                                               \
        if (dot) hipLaunchKernelGGL((nl_attn_kernel<C_, true>), grid, block, 0, s, X, Kx, Nk, Xo, Wp, bp, Zp, ML, Q, N, q0, q1);   \
        else hipLaunchKernelGGL((nl_attn_kernel<C_, false>), grid, block, 0, s, X, Kx, Nk, Xo, Wp, bp, Zp, ML, Q, N, q0, q1);    \
    } while (0)
    switch (C) {
        case 84: NL_LAUNCH(84); break;
        case 60: NL_LAUNCH(60); break;
        case 36: NL_LAUNCH(36); break;
        default: return hipErrorInvalidValue;
    }
#undef NL_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ks == 1) return e;
    return launch_nl_merge(X, Zp, ML, bp, Xo, B, N, C, ks, s, q0, q1);
}

// tf.layers.average_pooling2d(pool_size = strides = sub, 'valid') of the packed X [B][h2*w2][CP] -> [B][(h2/sub)*(w2/sub)][CP]
// (reference utils.py:27-28,35-36: applied to g and phi, which are 1x1 convolutions of X - the mean commutes with both).
__global__ void nl_pool_kernel(const float* __restrict__ X, float* __restrict__ Xs, int B, int h2, int w2, int sub, int CP) {
    const int hp = h2 / sub, wp = w2 / sub;
    const size_t total = (size_t)B * hp * wp * CP;
    const float inv = 1.0f / (float)(sub * sub);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CP);
        const int px = (int)((i / CP) % wp);
        const int py = (int)((i / ((size_t)CP * wp)) % hp);
        const size_t b = i / ((size_t)CP * wp * hp);
        float acc = 0.f;
        for (int dy = 0; dy < sub; ++dy)
            for (int dx = 0; dx < sub; ++dx)
                acc += X[((b * h2 + (size_t)(py * sub + dy)) * w2 + (px * sub + dx)) * CP + c];
        Xs[i] = acc * inv;
    }
}

hipError_t launch_nl_pool(const float* X, float* Xs, int B, int h2, int w2, int sub, int C, hipStream_t s) {
    if (sub < 1 || h2 / sub < 1 || w2 / sub < 1) return hipErrorInvalidValue;
    const int CP = nl_padded_ch(C);
    const size_t total = (size_t)B * (h2 / sub) * (w2 / sub) * CP;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(nl_pool_kernel, dim3(blocks), dim3(256), 0, s, X, Xs, B, h2, w2, sub, CP);
    return hipGetLastError();
}

hipError_t launch_nl_merge(const float* X, const float* Zp, const float* ML, const float* bp, float* Xo, int B, int N, int C, int ks,
                           hipStream_t s, int q0, int q1) {
    const int CP = nl_padded_ch(C);
    if (q1 < 0) q1 = N;
    const size_t total = (size_t)B * (q1 - q0) * (CP / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(nl_merge_kernel, dim3(blocks), dim3(256), 0, s, X, Zp, ML, bp, Xo, B, N, C, CP, ks, q0, q1);
    return hipGetLastError();
}

}  // namespace pfnl

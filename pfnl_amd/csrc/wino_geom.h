// Geometry and the hand-ordered K-step shared by the fused Winograd F(2x2,3x3) kernels
// (conv_wino.hip: one workgroup per tile; conv_wino_ws.hip: persistent, wave-specialised).
#pragma once
#include "common.h"

namespace pfnl {

constexpr int WN_TX = 16, WN_TY = 2;              // Winograd tiles per workgroup tile (x, y)
constexpr int WN_IW = 2 * WN_TX + 2;              // 34 input columns
constexpr int WN_IH = 2 * WN_TY + 2;              // 6 input rows
constexpr int WN_HALF = 20;                       // floats per column-parity half row (17 used)
constexpr int WN_RP = 2 * WN_HALF;                // row pitch 40: two tile rows are 80 = 16 mod 32 banks apart
constexpr int WN_PS = WN_IH * WN_RP + 1;          // plane stride 241 (odd)
constexpr int WN_CK = 16;
constexpr int WN_NCHUNK = 64 / WN_CK;
constexpr int WN_BUF = WN_CK * WN_PS;             // floats per raw buffer (3856)
constexpr int WN_ES = 36;                         // slab row stride (floats): 32 couts + 4, 16-byte aligned rows
constexpr int WN_SLAB = 4 * 2 * 32 * WN_ES;       // [xi][j][tile][cout]  (9216 floats, aliases the raw buffers)
constexpr int WN_LDS_FLOATS = (2 * WN_BUF > WN_SLAB) ? 2 * WN_BUF : WN_SLAB;
constexpr size_t WN_LDS_BYTES = size_t(WN_LDS_FLOATS) * sizeof(float);   // 36 864 B -> 4 workgroups / CU
constexpr int WN_THREADS = 256;
constexpr int WN_IN_ITEMS = WN_IH * WN_IW * (WN_CK / 4);                 // 816 float4 pieces
constexpr int WN_IN_ITERS = (WN_IN_ITEMS + WN_THREADS - 1) / WN_THREADS; // 4
constexpr int WN_UDEPTH = PFNL_WINO_UDEPTH;       // K-steps of U kept in flight
constexpr int WN_NSTEP = WN_NCHUNK * (WN_CK / 2); // 32 K-steps
constexpr int wino_pack_floats_c = 16 * 64 * 64;


// One K-step (2 input channels) of one wave, hand-ordered: the SIMD issues VALU and MFMA through one
// port, and VALU work is only hidden under a 64-cycle f32 MFMA when it sits between two MFMAs of the
// SAME wave in program order - hipcc clumps the four MFMAs together, so the order is fixed here.
//   current step : acc[nu] += V_cur[nu] (x) U[nu]                      (4 MFMAs)
//   next step    : 8 LDS reads (immediate offsets from two row bases), t[b] = dA[b] + sgn*dB[b],
//                  V_nxt = Bt * t                                       (8 VALU, in the MFMA shadows)
// Row selection / sign per Winograd row xi: (A,B,sgn) = (0,2,-1) (1,2,+1) (1,2,-1)* (1,3,-1);
// * xi = 2 needs d2 - d1 = -(d1 - d2): the minus sign is folded into U on the host.
template <int OFF>     // byte offset of (buffer, channel pair) inside the raw LDS tile
__device__ __forceinline__ void wn_kstep_asm(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3, const float (&vc)[4],
                                             float (&vn)[4], const f32x4 bc, unsigned pa, unsigned pb, float sgn) {
    float x0, x1, x2, x3, y0, y1, y2, y3;
    asm volatile(
        "ds_read_b32 %[x0], %[pa] offset:%c[o0]\n\t"
        "ds_read_b32 %[y0], %[pb] offset:%c[o0]\n\t"
        "ds_read_b32 %[x2], %[pa] offset:%c[o2]\n\t"
        "ds_read_b32 %[y2], %[pb] offset:%c[o2]\n\t"
        "v_mfma_f32_32x32x2_f32 %[a0], %[c0], %[bx], %[a0]\n\t"
        "ds_read_b32 %[x1], %[pa] offset:%c[o1]\n\t"
        "ds_read_b32 %[y1], %[pb] offset:%c[o1]\n\t"
        "ds_read_b32 %[x3], %[pa] offset:%c[o3]\n\t"
        "ds_read_b32 %[y3], %[pb] offset:%c[o3]\n\t"
        "v_mfma_f32_32x32x2_f32 %[a1], %[c1], %[by], %[a1]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_fmac_f32 %[x0], %[sg], %[y0]\n\t"          // t0
        "v_fmac_f32 %[x2], %[sg], %[y2]\n\t"          // t2
        "v_sub_f32 %[n0], %[x0], %[x2]\n\t"           // V0 = t0 - t2
        "v_mfma_f32_32x32x2_f32 %[a2], %[c2], %[bz], %[a2]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_fmac_f32 %[x1], %[sg], %[y1]\n\t"          // t1
        "v_fmac_f32 %[x3], %[sg], %[y3]\n\t"          // t3
        "v_add_f32 %[n1], %[x1], %[x2]\n\t"           // V1 = t1 + t2
        "v_sub_f32 %[n2], %[x2], %[x1]\n\t"           // V2 = t2 - t1
        "v_sub_f32 %[n3], %[x1], %[x3]\n\t"           // V3 = t1 - t3
        "v_mfma_f32_32x32x2_f32 %[a3], %[c3], %[bw], %[a3]\n\t"
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [n0] "=&v"(vn[0]), [n1] "=&v"(vn[1]),
          [n2] "=&v"(vn[2]), [n3] "=&v"(vn[3]), [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3),
          [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3)
        : [c0] "v"(vc[0]), [c1] "v"(vc[1]), [c2] "v"(vc[2]), [c3] "v"(vc[3]), [bx] "v"(bc.x), [by] "v"(bc.y),
          [bz] "v"(bc.z), [bw] "v"(bc.w), [pa] "v"(pa), [pb] "v"(pb), [sg] "v"(sgn), [o0] "i"(OFF),
          [o1] "i"(OFF + WN_HALF * 4), [o2] "i"(OFF + 4), [o3] "i"(OFF + WN_HALF * 4 + 4)
        : "memory");
}

// Same K-step for the first step of a tile: the accumulators are written, not read (C = 0).
template <int OFF>
__device__ __forceinline__ void wn_kstep_asm_zero(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3, const float (&vc)[4],
                                             float (&vn)[4], const f32x4 bc, unsigned pa, unsigned pb, float sgn) {
    float x0, x1, x2, x3, y0, y1, y2, y3;
    asm volatile(
        "ds_read_b32 %[x0], %[pa] offset:%c[o0]\n\t"
        "ds_read_b32 %[y0], %[pb] offset:%c[o0]\n\t"
        "ds_read_b32 %[x2], %[pa] offset:%c[o2]\n\t"
        "ds_read_b32 %[y2], %[pb] offset:%c[o2]\n\t"
        "v_mfma_f32_32x32x2_f32 %[a0], %[c0], %[bx], 0\n\t"
        "ds_read_b32 %[x1], %[pa] offset:%c[o1]\n\t"
        "ds_read_b32 %[y1], %[pb] offset:%c[o1]\n\t"
        "ds_read_b32 %[x3], %[pa] offset:%c[o3]\n\t"
        "ds_read_b32 %[y3], %[pb] offset:%c[o3]\n\t"
        "v_mfma_f32_32x32x2_f32 %[a1], %[c1], %[by], 0\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_fmac_f32 %[x0], %[sg], %[y0]\n\t"          // t0
        "v_fmac_f32 %[x2], %[sg], %[y2]\n\t"          // t2
        "v_sub_f32 %[n0], %[x0], %[x2]\n\t"           // V0 = t0 - t2
        "v_mfma_f32_32x32x2_f32 %[a2], %[c2], %[bz], 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_fmac_f32 %[x1], %[sg], %[y1]\n\t"          // t1
        "v_fmac_f32 %[x3], %[sg], %[y3]\n\t"          // t3
        "v_add_f32 %[n1], %[x1], %[x2]\n\t"           // V1 = t1 + t2
        "v_sub_f32 %[n2], %[x2], %[x1]\n\t"           // V2 = t2 - t1
        "v_sub_f32 %[n3], %[x1], %[x3]\n\t"           // V3 = t1 - t3
        "v_mfma_f32_32x32x2_f32 %[a3], %[c3], %[bw], 0\n\t"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [n0] "=&v"(vn[0]), [n1] "=&v"(vn[1]),
          [n2] "=&v"(vn[2]), [n3] "=&v"(vn[3]), [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3),
          [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3)
        : [c0] "v"(vc[0]), [c1] "v"(vc[1]), [c2] "v"(vc[2]), [c3] "v"(vc[3]), [bx] "v"(bc.x), [by] "v"(bc.y),
          [bz] "v"(bc.z), [bw] "v"(bc.w), [pa] "v"(pa), [pb] "v"(pb), [sg] "v"(sgn), [o0] "i"(OFF),
          [o1] "i"(OFF + WN_HALF * 4), [o2] "i"(OFF + 4), [o3] "i"(OFF + WN_HALF * 4 + 4)
        : "memory");
}



// ---- packed-math K-step (natural column order in LDS) --------------------------------------------
// Measured on gfx950 (tools/ubench/): while a SIMD's matrix pipe is saturated with f32 MFMAs, VALU
// instructions get ~1 issue slot per MFMA, and inside a wave every VALU / LDS instruction between two
// MFMAs delays the next MFMA (~14 cycles per group + ~4 per instruction).  Instruction COUNT is the
// currency, so this K-step does the same transform with 4 ds_read_b64 + 4 packed-f32 VALU instead of
// 8 ds_read_b32 + 8 VALU:  lane (tile column tx) reads columns 2tx..2tx+3 of rows A and B as two
// 8-byte pairs each,  T01 = X01 + sgn*Y01,  T23 = X23 + sgn*Y23,
// (V0,V3) = T01 - T23,  (V1,V2) = (t1 + t2, t2 - t1) via op_sel.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int WP_RP = 40;                          // row pitch (floats), even: 8-byte aligned pairs
constexpr int WP_PS = WN_IH * WP_RP + 2;           // plane stride 242 (even)
constexpr int WP_BUF = WN_CK * WP_PS;              // floats per raw buffer (3872)

template <int OFF>     // byte offset of (buffer, channel pair of the NEXT step) inside the raw LDS tile
__device__ __forceinline__ void wp_kstep_asm(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3, const float c0,
                                             const float c1, const float c2, const float c3, f32x2& n03, f32x2& n12,
                                             const f32x4 bc, unsigned pa, unsigned pb, f32x2 sg) {
    f32x2 x01, x23, y01, y23;
    asm volatile(
        "v_mfma_f32_32x32x2_f32 %[a0], %[c0], %[bx], %[a0]\n\t"
        "ds_read_b64 %[x01], %[pa] offset:%c[o0]\n\t"
        "ds_read_b64 %[y01], %[pb] offset:%c[o0]\n\t"
        "ds_read_b64 %[x23], %[pa] offset:%c[o2]\n\t"
        "ds_read_b64 %[y23], %[pb] offset:%c[o2]\n\t"
        "v_mfma_f32_32x32x2_f32 %[a1], %[c1], %[by], %[a1]\n\t"
        "v_mfma_f32_32x32x2_f32 %[a2], %[c2], %[bz], %[a2]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_fma_f32 %[x01], %[y01], %[sg], %[x01]\n\t"                              // (t0, t1)
        "v_pk_fma_f32 %[x23], %[y23], %[sg], %[x23]\n\t"                              // (t2, t3)
        "v_pk_add_f32 %[n03], %[x01], %[x23] neg_lo:[0,1] neg_hi:[0,1]\n\t"           // (t0 - t2, t1 - t3)
        "v_pk_add_f32 %[n12], %[x01], %[x23] op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"  // (t1 + t2, t2 - t1)
        "v_mfma_f32_32x32x2_f32 %[a3], %[c3], %[bw], %[a3]\n\t"
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [n03] "=&v"(n03), [n12] "=&v"(n12),
          [x01] "=&v"(x01), [x23] "=&v"(x23), [y01] "=&v"(y01), [y23] "=&v"(y23)
        : [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [bx] "v"(bc.x), [by] "v"(bc.y), [bz] "v"(bc.z),
          [bw] "v"(bc.w), [pa] "v"(pa), [pb] "v"(pb), [sg] "v"(sg), [o0] "i"(OFF), [o2] "i"(OFF + 8)
        : "memory");
}
// First K-step of a tile: accumulators written (C = 0), not read.
template <int OFF>
__device__ __forceinline__ void wp_kstep_asm_zero(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3, const float c0,
                                             const float c1, const float c2, const float c3, f32x2& n03, f32x2& n12,
                                             const f32x4 bc, unsigned pa, unsigned pb, f32x2 sg) {
    f32x2 x01, x23, y01, y23;
    asm volatile(
        "v_mfma_f32_32x32x2_f32 %[a0], %[c0], %[bx], 0\n\t"
        "ds_read_b64 %[x01], %[pa] offset:%c[o0]\n\t"
        "ds_read_b64 %[y01], %[pb] offset:%c[o0]\n\t"
        "ds_read_b64 %[x23], %[pa] offset:%c[o2]\n\t"
        "ds_read_b64 %[y23], %[pb] offset:%c[o2]\n\t"
        "v_mfma_f32_32x32x2_f32 %[a1], %[c1], %[by], 0\n\t"
        "v_mfma_f32_32x32x2_f32 %[a2], %[c2], %[bz], 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_fma_f32 %[x01], %[y01], %[sg], %[x01]\n\t"                              // (t0, t1)
        "v_pk_fma_f32 %[x23], %[y23], %[sg], %[x23]\n\t"                              // (t2, t3)
        "v_pk_add_f32 %[n03], %[x01], %[x23] neg_lo:[0,1] neg_hi:[0,1]\n\t"           // (t0 - t2, t1 - t3)
        "v_pk_add_f32 %[n12], %[x01], %[x23] op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"  // (t1 + t2, t2 - t1)
        "v_mfma_f32_32x32x2_f32 %[a3], %[c3], %[bw], 0\n\t"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [n03] "=&v"(n03), [n12] "=&v"(n12),
          [x01] "=&v"(x01), [x23] "=&v"(x23), [y01] "=&v"(y01), [y23] "=&v"(y23)
        : [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [bx] "v"(bc.x), [by] "v"(bc.y), [bz] "v"(bc.z),
          [bw] "v"(bc.w), [pa] "v"(pa), [pb] "v"(pb), [sg] "v"(sg), [o0] "i"(OFF), [o2] "i"(OFF + 8)
        : "memory");
}

// ---- 8-MFMA K-step (both 32-channel N-tiles per wave): two asm blocks, 4 LDS + 4 packed VALU per 8 MFMAs
// Operand roles: srcA = U (rows = the N-tile's 32 output channels), srcB = V (columns = the wave's 32 Winograd tiles),
// so that an accumulator lane owns ONE tile and 4 consecutive registers are 4 consecutive channels: the column
// transform's results go to the slab as 16-byte stores (the per-lane data of both operands is the same either way).
// block A: N-tile 0's four MFMAs + the raw reads of the NEXT step;  block B: N-tile 1's four MFMAs with
// the transform of the next step's V between them.  ZERO: first step of a tile, C = 0.
#define PFNL_WQ_A(C0, C1, C2, C3)                                                                       \
    asm volatile("v_mfma_f32_32x32x2_f32 %[a0], %[bx], %[c0], " C0 "\n\t"                               \
                 "ds_read_b64 %[x01], %[pa] offset:%c[o0]\n\t"                                          \
                 "ds_read_b64 %[y01], %[pb] offset:%c[o0]\n\t"                                          \
                 "ds_read_b64 %[x23], %[pa] offset:%c[o2]\n\t"                                          \
                 "ds_read_b64 %[y23], %[pb] offset:%c[o2]\n\t"                                          \
                 "v_mfma_f32_32x32x2_f32 %[a1], %[by], %[c1], " C1 "\n\t"                               \
                 "v_mfma_f32_32x32x2_f32 %[a2], %[bz], %[c2], " C2 "\n\t"                               \
                 "v_mfma_f32_32x32x2_f32 %[a3], %[bw], %[c3], " C3 "\n\t"
#define PFNL_WQ_A_IN                                                                                    \
    [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [bx] "v"(bc.x), [by] "v"(bc.y), [bz] "v"(bc.z), \
        [bw] "v"(bc.w), [pa] "v"(pa), [pb] "v"(pb), [o0] "i"(OFF), [o2] "i"(OFF + 8)
template <int OFF, bool ZERO>
__device__ __forceinline__ void wq_kstep_a(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3, const float c0, const float c1,
                                           const float c2, const float c3, const f32x4 bc, f32x2& x01, f32x2& y01,
                                           f32x2& x23, f32x2& y23, unsigned pa, unsigned pb) {
    if constexpr (ZERO) {
        PFNL_WQ_A("0", "0", "0", "0")
                     : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [x01] "=&v"(x01), [y01] "=&v"(y01),
                       [x23] "=&v"(x23), [y23] "=&v"(y23)
                     : PFNL_WQ_A_IN
                     : "memory");
    } else {
        PFNL_WQ_A("%[a0]", "%[a1]", "%[a2]", "%[a3]")
                     : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [x01] "=&v"(x01), [y01] "=&v"(y01),
                       [x23] "=&v"(x23), [y23] "=&v"(y23)
                     : PFNL_WQ_A_IN
                     : "memory");
    }
}
#define PFNL_WQ_B(C0, C1, C2, C3)                                                                       \
    asm volatile("v_mfma_f32_32x32x2_f32 %[a0], %[bx], %[c0], " C0 "\n\t"                               \
                 "v_mfma_f32_32x32x2_f32 %[a1], %[by], %[c1], " C1 "\n\t"                               \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                             \
                 "v_pk_fma_f32 %[x01], %[y01], %[sg], %[x01]\n\t"                                       \
                 "v_pk_fma_f32 %[x23], %[y23], %[sg], %[x23]\n\t"                                       \
                 "v_pk_add_f32 %[n03], %[x01], %[x23] neg_lo:[0,1] neg_hi:[0,1]\n\t"                    \
                 "v_pk_add_f32 %[n12], %[x01], %[x23] op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"    \
                 "v_mfma_f32_32x32x2_f32 %[a2], %[bz], %[c2], " C2 "\n\t"                               \
                 "v_mfma_f32_32x32x2_f32 %[a3], %[bw], %[c3], " C3 "\n\t"
#define PFNL_WQ_B_IN                                                                                    \
    [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [bx] "v"(bc.x), [by] "v"(bc.y), [bz] "v"(bc.z), \
        [bw] "v"(bc.w), [y01] "v"(y01), [y23] "v"(y23), [sg] "v"(sg)
template <bool ZERO>
__device__ __forceinline__ void wq_kstep_b(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3, const float c0, const float c1,
                                           const float c2, const float c3, const f32x4 bc, f32x2& x01, const f32x2 y01,
                                           f32x2& x23, const f32x2 y23, f32x2& n03, f32x2& n12, const f32x2 sg) {
    if constexpr (ZERO) {
        PFNL_WQ_B("0", "0", "0", "0")
                     : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [x01] "+v"(x01), [x23] "+v"(x23),
                       [n03] "=&v"(n03), [n12] "=&v"(n12)
                     : PFNL_WQ_B_IN
                     : "memory");
    } else {
        PFNL_WQ_B("%[a0]", "%[a1]", "%[a2]", "%[a3]")
                     : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [x01] "+v"(x01), [x23] "+v"(x23),
                       [n03] "=&v"(n03), [n12] "=&v"(n12)
                     : PFNL_WQ_B_IN
                     : "memory");
    }
}

// ---- pieces of the two blocks above as separate statements, for the two K-steps around a tile boundary: the column
// transform + slab stores of one N-tile's accumulators are issued between the MFMAs of the OTHER N-tile
// (conv_wino_ws.hip), instead of after the last MFMA with the matrix pipe idle.
template <bool ZERO>
__device__ __forceinline__ void wq_mfma1(f32x16& a, const float u, const float v) {
    if constexpr (ZERO) asm volatile("v_mfma_f32_32x32x2_f32 %[a], %[u], %[v], 0" : [a] "=&v"(a) : [u] "v"(u), [v] "v"(v) : "memory");
    else asm volatile("v_mfma_f32_32x32x2_f32 %[a], %[u], %[v], %[a]" : [a] "+v"(a) : [u] "v"(u), [v] "v"(v) : "memory");
}
// first MFMA of block A + the 4 raw reads of the next step
template <int OFF, bool ZERO>
__device__ __forceinline__ void wq_a_head(f32x16& a0, const float u, const float v, f32x2& x01, f32x2& y01, f32x2& x23, f32x2& y23,
                                          unsigned pa, unsigned pb) {
#define PFNL_WQ_AH(C0)                                                          \
    asm volatile("v_mfma_f32_32x32x2_f32 %[a0], %[u], %[v], " C0 "\n\t"          \
                 "ds_read_b64 %[x01], %[pa] offset:%c[o0]\n\t"                  \
                 "ds_read_b64 %[y01], %[pb] offset:%c[o0]\n\t"                  \
                 "ds_read_b64 %[x23], %[pa] offset:%c[o2]\n\t"                  \
                 "ds_read_b64 %[y23], %[pb] offset:%c[o2]\n\t"
    if constexpr (ZERO) {
        PFNL_WQ_AH("0")
                     : [a0] "=&v"(a0), [x01] "=&v"(x01), [y01] "=&v"(y01), [x23] "=&v"(x23), [y23] "=&v"(y23)
                     : [u] "v"(u), [v] "v"(v), [pa] "v"(pa), [pb] "v"(pb), [o0] "i"(OFF), [o2] "i"(OFF + 8)
                     : "memory");
    } else {
        PFNL_WQ_AH("%[a0]")
                     : [a0] "+v"(a0), [x01] "=&v"(x01), [y01] "=&v"(y01), [x23] "=&v"(x23), [y23] "=&v"(y23)
                     : [u] "v"(u), [v] "v"(v), [pa] "v"(pa), [pb] "v"(pb), [o0] "i"(OFF), [o2] "i"(OFF + 8)
                     : "memory");
    }
#undef PFNL_WQ_AH
}
// the transform of the next step's V that sits between the 2nd and 3rd MFMA of block B
__device__ __forceinline__ void wq_b_transform(f32x2& x01, const f32x2 y01, f32x2& x23, const f32x2 y23, f32x2& n03, f32x2& n12,
                                               const f32x2 sg) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 "v_pk_fma_f32 %[x01], %[y01], %[sg], %[x01]\n\t"
                 "v_pk_fma_f32 %[x23], %[y23], %[sg], %[x23]\n\t"
                 "v_pk_add_f32 %[n03], %[x01], %[x23] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_add_f32 %[n12], %[x01], %[x23] op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
                 : [x01] "+v"(x01), [x23] "+v"(x23), [n03] "=&v"(n03), [n12] "=&v"(n12)
                 : [y01] "v"(y01), [y23] "v"(y23), [sg] "v"(sg)
                 : "memory");
}
// "the four accumulators of an N-tile are complete and may be read by VALU from here on": a few wait states after the
// MFMA issued just before (which targets the other N-tile), and a data dependence that keeps the compiler from
// scheduling the reads any earlier
__device__ __forceinline__ void wq_results_ready(f32x16& a0, f32x16& a1, f32x16& a2, f32x16& a3) {
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"memory");
}

// ---- 8-MFMA K-step split at the ONE place where compiler-issued instructions may land -------------------
// head = first MFMA + the 4 raw reads of the next step; the caller then issues the two U loads of step
// s+UD (plain builtins: the compiler keeps vmcnt and register liveness right) and whatever s_waitcnt the
// operands of `rest` need; rest = the other 7 MFMAs with the 4 packed-VALU transform before the last two.
// Non-MFMA instructions thus form two groups per K-step instead of four.  ZERO: first step of a tile (C = 0).
template <int OFF, bool ZERO>
__device__ __forceinline__ void wr_kstep_head(f32x16& a0, const float c0, const float b0x, f32x2& x01, f32x2& y01,
                                              f32x2& x23, f32x2& y23, unsigned pa, unsigned pb) {
#define PFNL_WR_HEAD(C0)                                                        \
    asm volatile("v_mfma_f32_32x32x2_f32 %[a0], %[c0], %[bx], " C0 "\n\t"       \
                 "ds_read_b64 %[x01], %[pa] offset:%c[o0]\n\t"                  \
                 "ds_read_b64 %[y01], %[pb] offset:%c[o0]\n\t"                  \
                 "ds_read_b64 %[x23], %[pa] offset:%c[o2]\n\t"                  \
                 "ds_read_b64 %[y23], %[pb] offset:%c[o2]\n\t"
    if constexpr (ZERO) {
        PFNL_WR_HEAD("0")
                     : [a0] "=&v"(a0), [x01] "=&v"(x01), [y01] "=&v"(y01), [x23] "=&v"(x23), [y23] "=&v"(y23)
                     : [c0] "v"(c0), [bx] "v"(b0x), [pa] "v"(pa), [pb] "v"(pb), [o0] "i"(OFF), [o2] "i"(OFF + 8)
                     : "memory");
    } else {
        PFNL_WR_HEAD("%[a0]")
                     : [a0] "+v"(a0), [x01] "=&v"(x01), [y01] "=&v"(y01), [x23] "=&v"(x23), [y23] "=&v"(y23)
                     : [c0] "v"(c0), [bx] "v"(b0x), [pa] "v"(pa), [pb] "v"(pb), [o0] "i"(OFF), [o2] "i"(OFF + 8)
                     : "memory");
    }
#undef PFNL_WR_HEAD
}
template <bool ZERO>
__device__ __forceinline__ void wr_kstep_rest(f32x16& a1, f32x16& a2, f32x16& a3, f32x16& a4, f32x16& a5, f32x16& a6,
                                              f32x16& a7, const float c0, const float c1, const float c2, const float c3,
                                              const f32x4 b0, const f32x4 b1, f32x2& x01, const f32x2 y01, f32x2& x23,
                                              const f32x2 y23, f32x2& n03, f32x2& n12, const f32x2 sg) {
#define PFNL_WR_REST(C1, C2, C3, C4, C5, C6, C7)                                                        \
    asm volatile("v_mfma_f32_32x32x2_f32 %[a1], %[c1], %[py], " C1 "\n\t"                               \
                 "v_mfma_f32_32x32x2_f32 %[a2], %[c2], %[pz], " C2 "\n\t"                               \
                 "v_mfma_f32_32x32x2_f32 %[a3], %[c3], %[pw], " C3 "\n\t"                               \
                 "v_mfma_f32_32x32x2_f32 %[a4], %[c0], %[qx], " C4 "\n\t"                               \
                 "v_mfma_f32_32x32x2_f32 %[a5], %[c1], %[qy], " C5 "\n\t"                               \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                             \
                 "v_pk_fma_f32 %[x01], %[y01], %[sg], %[x01]\n\t"                                       \
                 "v_pk_fma_f32 %[x23], %[y23], %[sg], %[x23]\n\t"                                       \
                 "v_pk_add_f32 %[n03], %[x01], %[x23] neg_lo:[0,1] neg_hi:[0,1]\n\t"                    \
                 "v_pk_add_f32 %[n12], %[x01], %[x23] op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"    \
                 "v_mfma_f32_32x32x2_f32 %[a6], %[c2], %[qz], " C6 "\n\t"                               \
                 "v_mfma_f32_32x32x2_f32 %[a7], %[c3], %[qw], " C7 "\n\t"
#define PFNL_WR_REST_IN                                                                                  \
    [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [py] "v"(b0.y), [pz] "v"(b0.z), [pw] "v"(b0.w),  \
        [qx] "v"(b1.x), [qy] "v"(b1.y), [qz] "v"(b1.z), [qw] "v"(b1.w), [y01] "v"(y01), [y23] "v"(y23), [sg] "v"(sg)
    if constexpr (ZERO) {
        PFNL_WR_REST("0", "0", "0", "0", "0", "0", "0")
                     : [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6),
                       [a7] "=&v"(a7), [x01] "+v"(x01), [x23] "+v"(x23), [n03] "=&v"(n03), [n12] "=&v"(n12)
                     : PFNL_WR_REST_IN
                     : "memory");
    } else {
        PFNL_WR_REST("%[a1]", "%[a2]", "%[a3]", "%[a4]", "%[a5]", "%[a6]", "%[a7]")
                     : [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6),
                       [a7] "+v"(a7), [x01] "+v"(x01), [x23] "+v"(x23), [n03] "=&v"(n03), [n12] "=&v"(n12)
                     : PFNL_WR_REST_IN
                     : "memory");
    }
#undef PFNL_WR_REST
#undef PFNL_WR_REST_IN
}
}  // namespace pfnl

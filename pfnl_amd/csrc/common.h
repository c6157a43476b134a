// Shared declarations of libpfnl_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

#ifndef PFNL_WINO_WPS
#define PFNL_WINO_WPS 3      // waves per SIMD the Winograd kernel is compiled for (= workgroups per CU)
#endif
#ifndef PFNL_WINO_UDEPTH
#define PFNL_WINO_UDEPTH 2
#endif

namespace pfnl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact f32 (fmaf chain), 64 cycles/SIMD.
// Fragment maps (cdna_hip_programming.md §3): lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D register r of lane l is D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int drow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// 16-byte buffer store that is safe to follow with anything.  Measured on gfx950 (tools/GFX950_NOTES.md, "store data"): a
// buffer_store_dwordx4 whose soffset is an SGPR still reads its data registers for a cycle or two after issue, this compiler's
// hazard recognizer knows the hazard only for the immediate-offset forms, and a VALU write to one of the registers in the next
// issue slot reached it first - one corrupted dword in 16 lanes, once in ~10^5 stores.  The s_nop takes the stored registers as
// INPUTS: they stay live up to it, so nothing the scheduler puts in between can write them.  tools/lint_store_hazard.py checks
// the emitted assembly of every kernel for the pattern.
typedef unsigned pfnl_u32x4 __attribute__((ext_vector_type(4)));
template <int AUX = 0>                                              // AUX: the cache-policy bits of the instruction (gfx940+: 1 sc0, 2 nt, 16 sc1)
__device__ __forceinline__ void buffer_store_b128_guarded(pfnl_u32x4 v, __amdgpu_buffer_rsrc_t rs, int voffset, int soffset) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voffset, soffset, AUX);
    asm volatile("s_nop 1" ::"v"(v));
}

// Launch-time facts that are per DEVICE.  One process may drive several devices from several host threads (pfnl_comm_init_all, one
// handle per device): the launchers' lazy caches are indexed by the device and their slots are std::atomic, so a first launch
// racing with another thread's is a repeated, identical initialisation - not a data race, and never another device's CU count.
inline int device_cu_count() {                                      // CUs of the CURRENT device (0: it cannot be queried)
    static std::atomic<int> ncu[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int n = ncu[dev].load(std::memory_order_relaxed);
    if (!n) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 0;
        ncu[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, 0.2f * v); }  // tf.nn.leaky_relu

// ---- MFMA implicit-GEMM convolution (conv_mfma.hip) ------------------------------------------
// Output tile of one workgroup: 8 rows x 32 columns x 64 output channels of one item.
constexpr int CONV_TW = 32;
constexpr int CONV_TH = 8;        // rows per workgroup at MT = 2 (conv_mfma.hip)
constexpr int CONV_CK = 16;       // input channels per K-chunk of the 3x3 kernel (conv_ck(): 32 for 1x1)
constexpr int CONV_NPAD = 64;     // output channels are padded to 64 in the packed weights

struct ConvParams {
    const float* in;       // [items*frames_per_item][H][W][in_cstride]
    const float* wpack;    // [nchunks][ks*ks][CONV_CK][64]
    const float* bias;     // [64] (zero padded); never null (use a zero vector)
    const float* addend;   // [items/add_div][H][W][64]  (added before the activation) \ both or
    const float* resid;    // [items][H][W][out_cstride] (added after the activation)  / neither
    float* out;            // [items][H][W][out_cstride]
    int H, W;
    int in_cstride;        // floats per input pixel (64)
    int out_cstride;       // floats per output pixel (= cout)
    int cout;              // valid output channels (<= 64)
    int chunks_per_frame;  // in_cstride / CONV_CK
    int frames_per_item;   // 1: per-frame conv; T: conv over the concat of T frames
    int nchunks;           // frames_per_item * chunks_per_frame
    int add_div;           // addend item = item / add_div
    int act;               // 1 = leaky_relu(0.2)
};

hipError_t launch_conv_mfma(const ConvParams& p, int ksize, int items, hipStream_t s);
int conv_ck(int ksize);
size_t conv_pack_floats(int ksize, int cin);                 // floats in the packed weight blob
// HWIO [k,k,cin_total,cout] rows [cin_begin, cin_begin+cin) -> [chunk][tap][CK][64]
void conv_pack_weights(const float* hwio, int ksize, int cin_total, int cin_begin, int cin,
                       int cout, float* dst);

// ---- fused Winograd F(2x2,3x3) 64->64 convolution (conv_wino.hip) -------------------------------
// conv10_i: streaming 1x1 over the concat of T frames, T*64 -> 64 (conv1x1.hip)
hipError_t launch_conv1x1_stream(const float* in, const float* wpack, const float* bias, float* out, int items, int T,
                                 int HW, int act, hipStream_t s);
hipError_t launch_conv1x1_split16(const float* in, const uint16_t* wpack, const float* bias, float* out, int items, int T,
                                  int HW, int act, hipStream_t s,    // same contract on the f16 pipe, exactly split operands
                                  bool in_sf = false, bool out_sf = false);   // in / out in the split format (conv_split16.h)
size_t conv1x1_split16_pack_halfs(int T);
void conv1x1_split16_pack_weights(const float* hwio, int T, uint16_t* dst);   // HWIO [1,1,T*64,64]
size_t conv1x1_pack_floats(int T);
void conv1x1_pack_weights(const float* hwio, int T, float* dst);   // HWIO [1,1,T*64,64]

struct WinoParams {
    const float* in;       // [items][H][W][64]
    const float* upack;    // U = G g G^T, packed [chunk][xi][kk][lane][nu*2+nt] (wino_pack_weights)
    const float* bias;     // [64]; never null
    const float* addend;   // [items/add_div][H][W][64] \ both or
    const float* resid;    // [items][H][W][64]         / neither
    float* out;            // [items][H][W][64]
    int H, W;              // both even
    int add_div;
    int act;
    int items;
    long long* dbg;        // PFNL_WINO_TIMING builds only: 64 clock64() stamps per workgroup (else null)
    long long in_item_stride;   // floats between consecutive input items; 0 = H*W*64 (conv_wino_ws only)
    const float* in2;      // conv_wino_ws MODE 2 (whole conv2_i): base [items/add_div][H][W][64]; else null
    const float* upack2;   // ... and the packed U of the kernel rows that multiply it
    int accum;             // conv_wino_ws MODE 3 (convmerge1): sum over the add_div frames of a clip, one output per clip
    long long upack_stride;   // ... floats between the packed U of consecutive frames
};
hipError_t launch_conv_wino(const WinoParams& p, hipStream_t s);
hipError_t launch_conv_wino_ws(const WinoParams& p, hipStream_t s);  // persistent wave-specialised variant (conv_wino_ws.hip), same packed U
size_t wino_pack_floats();
void wino_pack_weights(const float* hwio, int cin_total, int cin_begin, float* dst, int cout = 64);

// ---- non-local block (nonlocal.hip) ----------------------------------------------------------
int nl_padded_ch(int C);                                      // 32*ceil(C/32)
hipError_t launch_nl_pack(const float* x, float* X, int B, int T, int H, int W, hipStream_t s);
size_t nl_partial_floats(int B, int N, int C);               // scratch for the key-split partials (0 if unsplit)
hipError_t launch_nl_attn(const float* X, float* Xo, const float* Wp, const float* bp, float* partial, int B,
                          int N, int C, hipStream_t s, const float* Q = nullptr,    // Q: projected queries (nltype 0) or null = X
                          int q0 = 0, int q1 = -1);                                 // queries [q0, q1) only (-1: N): a strip of the frame
hipError_t launch_nl_qproj(const float* X, const float* M, const float* c, float* Q, int B, int N, int C, hipStream_t s,
                           bool dot_column = false);                                // + column C = theta . b_phi (nltype 2)
// the general form of the block (reference utils.py:18-71, nltype 0 / 1 / 2, sub_sample >= 1): keys = values = Kx [B][Nk][CP]
hipError_t launch_nl_attn_general(const float* X, const float* Kx, int Nk, float* Xo, const float* Wp, const float* bp,
                                  float* partial, int B, int N, int C, hipStream_t s, const float* Q, int q0, int q1, bool dot);
hipError_t launch_nl_pool(const float* X, float* Xs, int B, int h2, int w2, int sub, int C, hipStream_t s);
int nl_key_splits(int B, int N);
hipError_t launch_nl_merge(const float* X, const float* Zp, const float* ML, const float* bp, float* Xo, int B, int N, int C, int ks,
                           hipStream_t s, int q0 = 0, int q1 = -1);
hipError_t launch_nl_unpack(const float* Xo, float* out, int B, int T, int H, int W, hipStream_t s);

// ---- head / tail (misc_kernels.hip) ----------------------------------------------------------
// A strip of the frame (single-clip multi-GPU sharding): the trunk buffers hold LR rows [yoff, yoff + Hs)
// of the H-row frame; only rows [core0, core1) (strip-local) of the result are written, at their place in the full output.
struct StripGeom {
    int yoff, Hs, core0, core1;
};
hipError_t launch_conv0(const float* Xo, const float* w75x64, const float* bias, float* out, int B,
                        int T, int H, int W, hipStream_t s, const StripGeom* strip = nullptr, bool force_f32 = false);   // force_f32: the VALU fp32 kernel
hipError_t launch_tail(const float* merge, const float* x, const float* w2, const float* b2,
                       float* out, int B, int T, int H, int W, int scale, int merge_cstride, hipStream_t s,
                       const StripGeom* strip = nullptr, unsigned* nonfinite = nullptr);   // nonfinite: set to 1 if a written value is inf / NaN (device or device-mapped host memory)
hipError_t launch_blur_decimate(const float* hr, float* lr, int F, int H, int W, int scale, hipStream_t s);
hipError_t launch_bicubic(const float* x, float* out, int B, int H, int W, int scale, hipStream_t s);
hipError_t run_mfma_selftest(int* mismatches);
hipError_t launch_gather_windows(const float* frames, float* win, int F, int first, int count, int T, size_t frame_floats,
                                 hipStream_t s);
hipError_t launch_quantise_u8(const float* sr, uint8_t* out, size_t n, hipStream_t s);

}  // namespace pfnl

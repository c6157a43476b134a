// Single-op hooks of the C-ABI added for per-stage parity (include/pfnl_hip.h): conv0, the tail, the embedded-Gaussian
// non-local block, and the two harness helpers.  Each replaces one call site of the reference, cited in the header.
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pfnl_hip.h"
#include "capi_internal.h"
#include "common.h"

#define OPS_FAIL(code, msg) return pfnl_internal_fail(code, msg)
#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) OPS_FAIL(PFNL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

extern "C" {

int pfnl_op_conv0(const float* x, const float* kernel_host, const float* bias_host, float* out, int B, int T, int H, int W,
                  void* stream) {
    if (!x || !kernel_host || !out) OPS_FAIL(PFNL_ERR_INVALID, "NULL argument");
    if ((T != 3 && T != 5 && T != 7) || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) OPS_FAIL(PFNL_ERR_INVALID, "unsupported conv0 geometry");
    hipStream_t s = (hipStream_t)stream;
    const int CP = pfnl::nl_padded_ch(12 * T), N = (H / 2) * (W / 2);
    std::vector<float> wb(75 * 64 + 64, 0.f);
    std::memcpy(wb.data(), kernel_host, 75 * 64 * sizeof(float));
    if (bias_host) std::memcpy(&wb[75 * 64], bias_host, 64 * sizeof(float));
    float* d = nullptr;
    HIPCHK(hipMalloc(&d, (wb.size() + (size_t)B * N * CP) * sizeof(float)));
    float* dX = d + wb.size();
    hipError_t e = hipMemcpy(d, wb.data(), wb.size() * sizeof(float), hipMemcpyHostToDevice);
    // conv0 reads the frame stack in the packed space_to_depth layout the non-local block leaves it in
    if (e == hipSuccess) e = pfnl::launch_nl_pack(x, dX, B, T, H, W, s);
    if (e == hipSuccess) e = pfnl::launch_conv0(dX, d, d + 75 * 64, out, B, T, H, W, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) OPS_FAIL(PFNL_ERR_HIP, std::string("conv0 op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_tail(const float* merge, const float* x, const float* kernel_host, const float* bias_host, float* out, int B,
                 int T, int H, int W, int scale, void* stream) {
    if (!merge || !x || !kernel_host || !out) OPS_FAIL(PFNL_ERR_INVALID, "NULL argument");
    if (B < 1 || T < 1 || H < 1 || W < 1 || (scale != 2 && scale != 4)) OPS_FAIL(PFNL_ERR_INVALID, "unsupported tail geometry");
    hipStream_t s = (hipStream_t)stream;
    const int CO = scale == 4 ? 12 : 3;
    std::vector<float> wb(9 * 12 * CO + 64, 0.f);
    std::memcpy(wb.data(), kernel_host, (size_t)9 * 12 * CO * sizeof(float));
    if (bias_host) std::memcpy(&wb[9 * 12 * CO], bias_host, CO * sizeof(float));
    float* d = nullptr;
    HIPCHK(hipMalloc(&d, wb.size() * sizeof(float)));
    hipError_t e = hipMemcpy(d, wb.data(), wb.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pfnl::launch_tail(merge, x, d, d + 9 * 12 * CO, out, B, T, H, W, scale, 48, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) OPS_FAIL(PFNL_ERR_HIP, std::string("tail op: ") + hipGetErrorString(e));
    return 0;
}

// utils.NonLocalBlock in its general form (reference utils.py:18-71: nltype 0 embedded Gaussian, 1 Gaussian, 2 dot product; sub_sample)
// + the stack / space_to_depth / depth_to_space / residual of model/pfnl.py:55-60.  The 1x1 convolutions are folded on the host in fp64
// (like pfnl_finalize_weights): W' = Wg Ww, b' = bg Ww + bw (the rows of P sum to 1), M = Wt Wp^T, c = bt Wp^T, and for nltype 2 the
// per-query constant theta_i . b_phi as column C of M / c; average pooling commutes with the 1x1 convolutions of g and phi.
static int op_nonlocal_block(const float* x, const float* wg, const float* bg, const float* ww, const float* bw, const float* wt,
                             const float* bt, const float* wp, const float* bp, int nltype, int sub, float* out, int B, int T, int H,
                             int W, void* stream) {
    if (!x || !wg || !bg || !ww || !bw || !out) OPS_FAIL(PFNL_ERR_INVALID, "NULL argument");
    if (nltype < 0 || nltype > 2) OPS_FAIL(PFNL_ERR_INVALID, "nltype: 0 | 1 | 2 (3, 'concat', builds no graph in the reference either)");
    if (nltype != 1 && (!wt || !bt || !wp || !bp)) OPS_FAIL(PFNL_ERR_INVALID, "nltype 0 / 2 need the theta and phi projections");
    if ((T != 3 && T != 5 && T != 7) || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1))
        OPS_FAIL(PFNL_ERR_INVALID, "unsupported non-local geometry");
    if (sub < 1 || (H / 2) / sub < 1 || (W / 2) / sub < 1) OPS_FAIL(PFNL_ERR_INVALID, "sub_sample out of range for this geometry");
    hipStream_t s = (hipStream_t)stream;
    const int C = 12 * T, CP = pfnl::nl_padded_ch(C), N = (H / 2) * (W / 2);
    const int Nk = sub > 1 ? ((H / 2) / sub) * ((W / 2) / sub) : N;
    // blob: W' [CP][CP] | b' [CP] | M [CP][CP] | c [CP]
    std::vector<float> blob(2 * ((size_t)CP * CP + CP), 0.f);
    float* Wf = blob.data();
    float* bf = Wf + (size_t)CP * CP;
    float* Mf = bf + CP;
    float* cf = Mf + (size_t)CP * CP;
    for (int ci = 0; ci < C; ++ci)
        for (int co = 0; co < C; ++co) {
            double a = 0.0, m = 0.0;
            for (int cm = 0; cm < C; ++cm) {
                a += (double)wg[(size_t)ci * C + cm] * (double)ww[(size_t)cm * C + co];
                if (nltype != 1) m += (double)wt[(size_t)ci * C + cm] * (double)wp[(size_t)co * C + cm];
            }
            Wf[(size_t)ci * CP + co] = (float)a;
            Mf[(size_t)ci * CP + co] = (float)m;
        }
    for (int co = 0; co < C; ++co) {
        double a = bw[co], m = 0.0;
        for (int cm = 0; cm < C; ++cm) {
            a += (double)bg[cm] * (double)ww[(size_t)cm * C + co];
            if (nltype != 1) m += (double)bt[cm] * (double)wp[(size_t)co * C + cm];
        }
        bf[co] = (float)a;
        cf[co] = (float)m;
    }
    if (nltype == 2) {
        double d0 = 0.0;
        for (int cm = 0; cm < C; ++cm) d0 += (double)bt[cm] * (double)bp[cm];
        cf[C] = (float)d0;
        for (int ci = 0; ci < C; ++ci) {
            double a = 0.0;
            for (int cm = 0; cm < C; ++cm) a += (double)wt[(size_t)ci * C + cm] * (double)bp[cm];
            Mf[(size_t)ci * CP + C] = (float)a;
        }
    }
    const size_t nX = (size_t)B * N * CP, nK = sub > 1 ? (size_t)B * Nk * CP : 0, nP = pfnl::nl_partial_floats(B, N, C);
    float* d = nullptr;
    HIPCHK(hipMalloc(&d, (blob.size() + 3 * nX + nK + nP) * sizeof(float)));
    float* dX = d + blob.size();
    float* dXo = dX + nX;
    float* dQ = dXo + nX;
    float* dK = dQ + nX;
    float* dP = nP ? dK + nK : nullptr;
    hipError_t e = hipMemcpy(d, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pfnl::launch_nl_pack(x, dX, B, T, H, W, s);
    if (e == hipSuccess && nltype != 1)
        e = pfnl::launch_nl_qproj(dX, d + (size_t)CP * CP + CP, d + 2 * (size_t)CP * CP + CP, dQ, B, N, C, s, nltype == 2);
    if (e == hipSuccess && sub > 1) e = pfnl::launch_nl_pool(dX, dK, B, H / 2, W / 2, sub, C, s);
    if (e == hipSuccess)
        e = pfnl::launch_nl_attn_general(dX, sub > 1 ? dK : dX, Nk, dXo, d, d + (size_t)CP * CP, dP, B, N, C, s, nltype != 1 ? dQ : nullptr,
                                         0, -1, nltype == 2);
    if (e == hipSuccess) e = pfnl::launch_nl_unpack(dXo, out, B, T, H, W, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) OPS_FAIL(PFNL_ERR_HIP, std::string("non-local block op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_nonlocal_embedded(const float* x, const float* wg, const float* bg, const float* ww, const float* bw,
                              const float* wt, const float* bt, const float* wp, const float* bp, float* out, int B, int T,
                              int H, int W, void* stream) {
    if (!wt || !bt || !wp || !bp) OPS_FAIL(PFNL_ERR_INVALID, "NULL argument");
    return op_nonlocal_block(x, wg, bg, ww, bw, wt, bt, wp, bp, 0, 1, out, B, T, H, W, stream);
}

int pfnl_op_nonlocal_block(const float* x, const float* wg, const float* bg, const float* ww, const float* bw, const float* wt,
                           const float* bt, const float* wp, const float* bp, int nltype, int sub_sample, float* out, int B, int T,
                           int H, int W, void* stream) {
    return op_nonlocal_block(x, wg, bg, ww, bw, wt, bt, wp, bp, nltype, sub_sample, out, B, T, H, W, stream);
}

int pfnl_op_gather_windows(const float* frames, float* win, int F, int first, int count, int T, int H, int W, void* stream) {
    if (!frames || !win) OPS_FAIL(PFNL_ERR_INVALID, "NULL argument");
    if (F < 1 || first < 0 || count < 1 || first + count > F || T < 1 || !(T & 1) || H < 1 || W < 1 || ((size_t)H * W * 3) % 4)
        OPS_FAIL(PFNL_ERR_INVALID, "bad window geometry (H*W*3 must be a multiple of 4, T odd, first + count <= F)");
    HIPCHK(pfnl::launch_gather_windows(frames, win, F, first, count, T, (size_t)H * W * 3, (hipStream_t)stream));
    return 0;
}

int pfnl_op_quantise_u8(const float* sr, uint8_t* out, size_t n, void* stream) {
    if (!sr || !out) OPS_FAIL(PFNL_ERR_INVALID, "NULL argument");
    if (!n || n % 4) OPS_FAIL(PFNL_ERR_INVALID, "element count must be a positive multiple of 4");
    HIPCHK(pfnl::launch_quantise_u8(sr, out, n, (hipStream_t)stream));
    return 0;
}

}  // extern "C"

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

int pfnl_op_nonlocal_embedded(const float* x, const float* wg, const float* bg, const float* ww, const float* bw,
                              const float* wt, const float* bt, const float* wp, const float* bp, float* out, int B, int T,
                              int H, int W, void* stream) {
    if (!x || !wg || !bg || !ww || !bw || !wt || !bt || !wp || !bp || !out) OPS_FAIL(PFNL_ERR_INVALID, "NULL argument");
    if ((T != 3 && T != 5 && T != 7) || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1))
        OPS_FAIL(PFNL_ERR_INVALID, "unsupported non-local geometry");
    hipStream_t s = (hipStream_t)stream;
    const int C = 12 * T, CP = pfnl::nl_padded_ch(C), N = (H / 2) * (W / 2);
    // blob: W' = Wg Ww [CP][CP] | b' [CP] | M = Wt Wp^T [CP][CP] | c = bt Wp^T [CP]   (fp64 folds, like pfnl_finalize_weights)
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
                m += (double)wt[(size_t)ci * C + cm] * (double)wp[(size_t)co * C + cm];
            }
            Wf[(size_t)ci * CP + co] = (float)a;
            Mf[(size_t)ci * CP + co] = (float)m;
        }
    for (int co = 0; co < C; ++co) {
        double a = bw[co], m = 0.0;
        for (int cm = 0; cm < C; ++cm) {
            a += (double)bg[cm] * (double)ww[(size_t)cm * C + co];
            m += (double)bt[cm] * (double)wp[(size_t)co * C + cm];
        }
        bf[co] = (float)a;
        cf[co] = (float)m;
    }
    const size_t nX = (size_t)B * N * CP, nP = pfnl::nl_partial_floats(B, N, C);
    float* d = nullptr;
    HIPCHK(hipMalloc(&d, (blob.size() + 3 * nX + nP) * sizeof(float)));
    float* dX = d + blob.size();
    float* dXo = dX + nX;
    float* dQ = dXo + nX;
    float* dP = nP ? dQ + nX : nullptr;
    hipError_t e = hipMemcpy(d, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pfnl::launch_nl_pack(x, dX, B, T, H, W, s);
    if (e == hipSuccess) e = pfnl::launch_nl_qproj(dX, d + (size_t)CP * CP + CP, d + 2 * (size_t)CP * CP + CP, dQ, B, N, C, s);
    if (e == hipSuccess) e = pfnl::launch_nl_attn(dX, dXo, d, d + (size_t)CP * CP, dP, B, N, C, s, dQ);
    if (e == hipSuccess) e = pfnl::launch_nl_unpack(dXo, out, B, T, H, W, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) OPS_FAIL(PFNL_ERR_HIP, std::string("embedded non-local op: ") + hipGetErrorString(e));
    return 0;
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

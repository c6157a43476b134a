// Micro-benchmark (gfx950): one wave per SIMD issuing v_mfma_f32_32x32x16_f16 back to back - where do the operands live, and how far
// apart are MFMAs on the same accumulator?  C/D in VGPRs or AGPRs x B operand in VGPRs or AGPRs x NACC accumulators in rotation.
// Question behind it (conv_wsplit.hip): the weights sit in the AGPR half of the register file as B operands and the accumulators in
// VGPRs - does that form run at the matrix pipe's 32 cycles per instruction?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int CD_A, int B_A, int NACC, int NV, int OP = 0>
__global__ __launch_bounds__(256, 1) void k(const h8* in, float* out, long long* cyc, int iters) {
    const int tid = threadIdx.x;
    h8 a = in[tid], b[4];
    for (int i = 0; i < 4; ++i) b[i] = in[256 + 4 * tid + i];
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    if (B_A) for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(b[i]));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 fill[8], fa = {1.0f, 1.0f}, fb = {0.5f, 0.25f};
    typedef float float4v __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    for (int i = tid; i < 8192; i += 256) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    float4v rd[4], rd2[2][4];
    for (int i = 0; i < 4; ++i) { rd[i] = float4v{0, 0, 0, 0}; rd2[0][i] = rd[i]; rd2[1][i] = rd[i]; }
    int ifill[8];
    for (int q = 0; q < 8; ++q) { fill[q] = f2{(float)tid, (float)q}; ifill[q] = tid + q; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            f32x16& c = acc[u % NACC];
            if (CD_A && B_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "a"(b[u & 3]));
            else if (CD_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b[u & 3]));
            else if (B_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b[u & 3]));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b[u & 3]));
#pragma unroll
            for (int q = 0; q < NV; ++q) {   // independent VALU in the MFMA's shadow
                if (OP == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(fill[q & 7]) : "v"(fa), "v"(fb));
                if (OP == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fill[q & 7].x) : "v"(fa.x), "v"(fb.x));
                if (OP == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(ifill[q & 7]) : "v"(tid));
                if (OP == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ifill[q & 7]) : "v"(fa.x), "v"(fb.x));
                if (OP == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(fill[q & 7]) : "v"(fa), "v"(fb));
                if (OP == 5) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(ifill[q & 7]) : "v"(tid));
                if (OP == 6) {   // one ds_read_b128 (conflict-free: lane * 16) + NV-1... : reads of this gap are consumed by v_fma two gaps later
                    if (q == 0) { rd[u & 3] = *reinterpret_cast<const float4v*>(lds + ((tid * 16 + u * 1024) & 32767)); }
                    else if (q == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fill[0].x) : "v"(rd[(u + 2) & 3].x), "v"(fb.x));
                    else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fill[q & 7].x) : "v"(fa.x), "v"(fb.x));
                }
                if (OP == 7) {   // two ds_read_b128 per gap
                    if (q < 2) { rd2[q][u & 3] = *reinterpret_cast<const float4v*>(lds + ((tid * 16 + u * 1024 + q * 16384) & 32767)); }
                    else if (q == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fill[0].x) : "v"(rd2[0][(u + 1) & 3].x), "v"(rd2[1][(u + 1) & 3].y));
                    else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fill[q & 7].x) : "v"(fa.x), "v"(fb.x));
                }
            }
        }
    }
    const long long t1 = clock64();
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += fill[q].x + fill[q].y + (float)ifill[q];
    for (int n = 0; n < NACC; ++n) {
        if (CD_A) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[n]));
        else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[n]));
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    }
    out[blockIdx.x * 256 + tid] = s;
}

template <int CD_A, int B_A, int NACC, int NV = 0, int OP = 0>
void run(const h8* in, float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<CD_A, B_A, NACC, NV, OP>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
    CHECK(hipDeviceSynchronize());
    long long h[4];
    CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    printf("C/D in %s, B in %s, %d accumulator(s) in rotation, %d x %s behind each MFMA: %.1f cycles per MFMA\n", CD_A ? "AGPR" : "VGPR", B_A ? "AGPR" : "VGPR", NACC, NV, OP == 0 ? "v_pk_fma_f32" : OP == 1 ? "v_fma_f32" : OP == 2 ? "v_add_u32" : OP == 3 ? "v_cvt_pk_f16_f32" : OP == 4 ? "v_pk_mul_f32" : OP == 5 ? "v_xor_b32" : OP == 6 ? "(1 ds_read_b128 + v_fma_f32)" : "(2 ds_read_b128 + v_fma_f32)", (double)h[0] / (iters * 24.0));
}

int main() {
    h8* in; float* out; long long* cyc;
    CHECK(hipMalloc(&in, 256 * 5 * sizeof(h8)));
    CHECK(hipMemset(in, 0, 256 * 5 * sizeof(h8)));
    CHECK(hipMalloc(&out, 256 * 256 * sizeof(float)));
    CHECK(hipMalloc(&cyc, 256 * 4 * sizeof(long long)));
    run<0, 0, 1>(in, out, cyc); run<0, 0, 2>(in, out, cyc); run<0, 0, 4>(in, out, cyc); run<0, 0, 6>(in, out, cyc);
    run<0, 1, 1>(in, out, cyc); run<0, 1, 2>(in, out, cyc); run<0, 1, 4>(in, out, cyc); run<0, 1, 6>(in, out, cyc);
    run<1, 0, 1>(in, out, cyc); run<1, 0, 2>(in, out, cyc); run<1, 0, 4>(in, out, cyc);
    run<1, 1, 1>(in, out, cyc); run<1, 1, 2>(in, out, cyc); run<1, 1, 4>(in, out, cyc);
    run<0, 1, 4, 2, 6>(in, out, cyc); run<0, 1, 4, 4, 6>(in, out, cyc); run<0, 1, 4, 6, 6>(in, out, cyc);
    run<0, 1, 4, 3, 7>(in, out, cyc); run<0, 1, 4, 5, 7>(in, out, cyc); run<0, 1, 4, 7, 7>(in, out, cyc);
    run<0, 1, 1, 4, 1>(in, out, cyc); run<0, 1, 2, 4, 1>(in, out, cyc); run<0, 1, 3, 4, 1>(in, out, cyc); run<0, 1, 6, 4, 1>(in, out, cyc);
    run<0, 1, 2, 2, 1>(in, out, cyc); run<0, 1, 2, 6, 1>(in, out, cyc); run<0, 1, 3, 6, 1>(in, out, cyc); run<0, 1, 6, 6, 1>(in, out, cyc);
    run<0, 1, 4, 2, 1>(in, out, cyc); run<0, 1, 4, 4, 1>(in, out, cyc); run<0, 1, 4, 6, 1>(in, out, cyc); run<0, 1, 4, 8, 1>(in, out, cyc);
    run<0, 1, 4, 4, 2>(in, out, cyc); run<0, 1, 4, 6, 2>(in, out, cyc); run<0, 1, 4, 4, 3>(in, out, cyc); run<0, 1, 4, 6, 3>(in, out, cyc);
    run<0, 1, 4, 4, 4>(in, out, cyc); run<0, 1, 4, 6, 4>(in, out, cyc); run<0, 1, 4, 4, 5>(in, out, cyc); run<0, 1, 4, 6, 5>(in, out, cyc);
    run<1, 1, 4, 4, 1>(in, out, cyc); run<1, 1, 4, 6, 1>(in, out, cyc);
    run<0, 1, 4, 2>(in, out, cyc); run<0, 1, 4, 4>(in, out, cyc); run<0, 1, 4, 6>(in, out, cyc); run<0, 1, 4, 8>(in, out, cyc);
    run<1, 1, 4, 2>(in, out, cyc); run<1, 1, 4, 4>(in, out, cyc); run<1, 1, 4, 6>(in, out, cyc); run<1, 1, 4, 8>(in, out, cyc);
    run<1, 0, 4, 4>(in, out, cyc); run<1, 0, 4, 6>(in, out, cyc); run<0, 0, 4, 4>(in, out, cyc); run<0, 0, 4, 6>(in, out, cyc);
    return 0;
}

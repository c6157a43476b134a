// Micro-benchmark (gfx950): issue cost of the VALU / LDS-store instructions the kernels' non-MFMA slices are made of,
// alone (1 wave per SIMD) and with a second wave of the same stream on the SIMD (2 waves per SIMD).
// Each instruction runs 64x unrolled over 8 independent destination registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int OP>
__global__ void k(int iters, long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    float v[8], w[8];
    unsigned u[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    f2 p[8], q[8];
    u4 d4 = {1u, 2u, 3u, 4u};
    for (int j = 0; j < 8; ++j) {
        v[j] = lane * 0.001f + j;
        w[j] = 1.0f + j * 1e-3f;
        u[j] = 0x3c003c00u + j;
        p[j] = f2{v[j], w[j]};
        q[j] = f2{1.0f, 1.0f};
    }
    const float sc = -2048.f;
    const unsigned la = (unsigned)(uintptr_t)(lds) + tid * 16;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == 0) {
#define X(j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j]) : "v"(w[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 1) {
#define X(j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(q[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 2) {
#define X(j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(q[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 3) {
#define X(j) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[j]) : "v"(v[j]), "v"(w[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 4) {
#define X(j) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "s"(sc), "v"(v[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 5) {
#define X(j) asm volatile("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "s"(sc), "v"(v[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 6) {
#define X(j) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(w[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 7) {
#define X(j) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[j]) : "v"(v[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 8) {
#define X(j) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[j]) : "v"(u[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 9) {
#define X(j) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
            REP64(X)
#undef X
        } else if constexpr (OP == 10) {
#define X(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
            REP64(X)
#undef X
        } else if constexpr (OP == 11) {
#define X(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[j]) : "v"(q[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 12) {
#define X(j) asm volatile("ds_write_b64 %0, %1 offset:%c2" :: "v"(la), "v"(p[j]), "i"(j * 8) : "memory");
            REP64(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if constexpr (OP == 13) {
#define X(j) asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(d4) : "memory");
            REP64(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if constexpr (OP == 14) {
#define X(j) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[j]) : "v"(v[j]), "v"(w[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 15) {
#define X(j) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "s"(sc));
            REP64(X)
#undef X
        } else if constexpr (OP == 16) {
#define X(j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 17) {
#define X(j) asm volatile("v_exp_f16 %0, %0" : "+v"(u[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 18) {
#define X(j) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
            REP64(X)
#undef X
        } else if constexpr (OP == 19) {
#define X(j) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[j]) : "v"(u[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 20) {
#define X(j) asm volatile("v_fract_f32 %0, %0" : "+v"(v[j]));
            REP64(X)
#undef X
        } else if constexpr (OP == 21) {
#define X(j) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[j]));
            REP64(X)
#undef X
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += v[j] + (float)u[j] + p[j].x + p[j].y;
    if (s == 12345.f) sink[0] = s + lds[lane];
    if (lane == 0 && tid < 64) out[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, long long* out, float* sink) {
    const int blocks = 256, iters = 200;
    double r[2];
    for (int wps = 1; wps <= 2; ++wps) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256 * wps), 0, 0, iters, out, sink);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(blocks);
        (void)hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += v;
        r[wps - 1] = s / blocks / iters / 64;
    }
    printf("%-22s %6.2f cycles per instruction alone, %6.2f per instruction of one wave with 2 waves per SIMD\n", name, r[0], r[1]);
}

int main() {
    long long* out;
    float* sink;
    (void)hipMalloc(&out, 256 * 8);
    (void)hipMalloc(&sink, 64);
    run<0>("v_fma_f32", out, sink);
    run<15>("v_mul_f32 (sgpr)", out, sink);
    run<6>("v_max_f32", out, sink);
    run<10>("v_add_u32", out, sink);
    run<1>("v_pk_mul_f32", out, sink);
    run<2>("v_pk_add_f32", out, sink);
    run<11>("v_pk_fma_f32", out, sink);
    run<3>("v_cvt_pk_f16_f32", out, sink);
    run<14>("v_cvt_pk_bf16_f32", out, sink);
    run<7>("v_cvt_f16_f32", out, sink);
    run<8>("v_cvt_f32_f16", out, sink);
    run<4>("v_fma_mixlo_f16", out, sink);
    run<5>("v_fma_mixhi_f16", out, sink);
    run<9>("v_pk_mul_f16", out, sink);
    run<16>("v_exp_f32", out, sink);
    run<17>("v_exp_f16", out, sink);
    run<21>("v_rcp_f32", out, sink);
    run<18>("v_pk_fma_f16", out, sink);
    run<19>("v_ldexp_f32", out, sink);
    run<20>("v_fract_f32", out, sink);
    run<12>("ds_write_b64", out, sink);
    run<13>("ds_write_b128", out, sink);
    return 0;
}

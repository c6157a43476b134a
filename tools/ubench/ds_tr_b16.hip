// Semantics probe (gfx950): `ds_read_b64_tr_b16`, the transposing LDS read.  DESIGN.md R6.7 / section 8 1c: the non-local block keeps K [key][channel]
// AND V^T [channel][key] in LDS - the same values twice - because the P V MFMA wants, per lane, 8 KEYS of one channel (a column of the K tile).
// Before that kernel can read V^T operands out of the K tile, the read's lane -> (address, element) map has to be known exactly:
//   1. which 16-bit elements does lane l get, as a function of the 64 per-lane addresses?  (every element of the LDS image is its own index)
//   2. is the exchange confined to 16-lane groups?  does a lane's own address matter, or only its group's?
//   3. given a K tile [key][channel] with row stride S bytes: which per-lane addresses make lane (i = l & 31, kh = l >> 5) receive keys
//      {k0 .. k0 + 3} of channel c(i) - the half of a v_mfma_f32_32x32x16_f16 A operand (8 keys per lane: two reads)?
// Prints the observed map for three address patterns and checks candidate (3) against a host model.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64) void probe(const int* addr, unsigned short* out, int nelem) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    for (int i = threadIdx.x; i < nelem; i += 64) lds[i] = (unsigned short)i;      // element value = its index
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)lds + (unsigned)addr[threadIdx.x];      // per-lane BYTE address
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = (unsigned short)(v.x & 0xffff);
    out[threadIdx.x * 4 + 1] = (unsigned short)(v.x >> 16);
    out[threadIdx.x * 4 + 2] = (unsigned short)(v.y & 0xffff);
    out[threadIdx.x * 4 + 3] = (unsigned short)(v.y >> 16);
}

static std::vector<unsigned short> run(const std::vector<int>& addr, int nelem) {
    int* da;
    unsigned short* dout;
    CHECK(hipMalloc(&da, 64 * sizeof(int)));
    CHECK(hipMalloc(&dout, 256 * sizeof(unsigned short)));
    CHECK(hipMemcpy(da, addr.data(), 64 * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), nelem * 2, 0, da, dout, nelem);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned short> out(256);
    CHECK(hipMemcpy(out.data(), dout, 256 * sizeof(unsigned short), hipMemcpyDeviceToHost));
    CHECK(hipFree(da));
    CHECK(hipFree(dout));
    return out;
}

static void show(const char* what, const std::vector<int>& addr, const std::vector<unsigned short>& out, int lanes = 64) {
    printf("== %s\n", what);
    for (int l = 0; l < lanes; ++l)
        printf("  lane %2d  addr %5d (elem %5d)  ->  %5d %5d %5d %5d\n", l, addr[l], addr[l] / 2, out[4 * l], out[4 * l + 1], out[4 * l + 2], out[4 * l + 3]);
}

int main() {
    const int nelem = 16384;
    // (a) the guide's canonical image: a [4][16] row-major block per 16-lane group, lane l at element (l & 15) ... expressed as BYTE addresses of
    //     4 contiguous elements per lane: lane l -> row (l & 15) / 4 ?  Unknown: so probe the plainest pattern first - lane l at 8 l bytes
    std::vector<int> a(64);
    for (int l = 0; l < 64; ++l) a[l] = 8 * l;
    auto o = run(a, nelem);
    show("lane l -> byte address 8 l (64 lanes x 4 contiguous elements = elements 0 .. 255)", a, o);
    // (b) rows far apart: lane l -> row (l & 15) of a matrix with 1024-element rows, 4 elements at column 4 (l >> 4)
    for (int l = 0; l < 64; ++l) a[l] = ((l & 15) * 1024 + 4 * (l >> 4)) * 2;
    o = run(a, nelem);
    show("lane l -> row (l & 15), columns 4 (l >> 4) .. + 3 of a [16][1024] image", a, o, 32);
    // (c) a 16-lane group on FOUR rows x 16 columns: lane l -> row (l & 3)?, column 4 ((l & 15) >> 2): the [4][16] block of the guide
    for (int l = 0; l < 64; ++l) a[l] = (((l >> 4) * 4 + (l & 3)) * 1024 + 4 * ((l & 15) >> 2)) * 2;
    o = run(a, nelem);
    show("lane l -> row 4 (l >> 4) + (l & 3), columns 4 ((l & 15) >> 2) .. + 3 of a [16][1024] image", a, o, 32);
    // THE LAW (checked on every pattern above): out[16 q + g][j] = piece(lane 16 q + 4 j + (g >> 2))[g & 3], piece(p) = the 4 contiguous elements at lane
    // p's own address.  A 16-lane group reads a [4 rows][16 columns] block - row j = the pieces of lanes 4 j .. 4 j + 3 - and lane g gets column g.
    auto law = [&](const std::vector<int>& ad, const std::vector<unsigned short>& ou) {
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int src_lane = (l & ~15) + 4 * j + ((l & 15) >> 2);
                if (ou[4 * l + j] != (unsigned short)(ad[src_lane] / 2 + (l & 3))) ++bad;
            }
        return bad;
    };
    int total = law(a, o);
    for (int l = 0; l < 64; ++l) a[l] = 8 * l;
    total += law(a, run(a, nelem));
    for (int l = 0; l < 64; ++l) a[l] = ((l & 15) * 1024 + 4 * (l >> 4)) * 2;
    total += law(a, run(a, nelem));
    // the use the non-local block has for it: a K tile [key][channel], row stride S elements; lane (i = l & 31, kh = l >> 5) of a P V A operand wants keys
    // k0 + 4 kh + 0 .. 3 of channel c0 + i: lane p = l & 15 of its group points at row k0 + 4 kh + (p >> 2), channels c0 + 16 ((l >> 4) & 1) + 4 (p & 3)
    const int S = 104, k0 = 16, c0 = 32;                            // (208-byte rows as nonlocal_f16.hip's K tile)
    for (int l = 0; l < 64; ++l) a[l] = ((k0 + 4 * (l >> 5) + ((l & 15) >> 2)) * S + c0 + 16 * ((l >> 4) & 1) + 4 * (l & 3)) * 2;
    o = run(a, nelem);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j)
            if (o[4 * l + j] != (unsigned short)((k0 + 4 * (l >> 5) + j) * S + c0 + (l & 31))) ++bad;
    printf("the law out[16q+g][j] = piece(16q + 4j + (g>>2))[g&3]: %d mismatches on three patterns; K-tile -> P V operand addressing: %d mismatches\n", total, bad);
    return 0;
}

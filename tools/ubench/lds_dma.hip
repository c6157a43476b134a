// Micro-benchmark / semantics probe (gfx950): `buffer_load_dwordx4 ... offen lds` (LDS-DMA) issued from inline asm, as the
// 3x3 kernels use it for pre-split halo tiles and weight slices.  Questions (answers printed, "OK"/"FAIL" per check):
//   1. destination = M0 + lane * 16 for LDS addresses above 64 KB, per-lane gather source = voffset (+ soffset);
//   2. lanes whose voffset is out of the resource's range: zero written, or nothing written?
//   3. lanes disabled in EXEC: nothing written, other lanes' destinations unchanged (no compaction);
//   4. the instruction's immediate offset: added to the memory address, the LDS address, or both?
//   5. "fence load": a compiler-visible buffer load issued AFTER the asm DMAs; once the compiler's own wait for that load has
//      passed (vmcnt is in order), the DMA data is in LDS - checked with HBM-cold sources, many workgroups, many rounds.
//   6. cost: cycles for a 43.5 KB halo-sized transfer per workgroup, DMA vs registers + ds_write_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// one DMA instruction: 64 lanes x 16 B -> LDS [lds_dst, lds_dst + 1024); m0 saved / restored (it is compiler-reserved)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma16_imm64(__amdgpu_buffer_rsrc_t rs, unsigned lds_dst, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen offset:64 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}

constexpr int LDS_BASE = 100000;       // > 64 KB, multiple of 16

// mode 0: permuted gather; 1: odd lanes out of range; 2: lanes >= 32 disabled; 3: immediate offset 64; 4: scalar offset 4096
__global__ __launch_bounds__(256) void probe(const unsigned* src, unsigned* out, int nbytes, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
    for (int i = tid; i < 4096 / 4; i += 256) reinterpret_cast<unsigned*>(smem + LDS_BASE)[i] = 0xABABABABu;   // sentinel
    __syncthreads();
    const unsigned dst = (unsigned)(uintptr_t)(smem + LDS_BASE) + wave * 1024;
    int voff = (wave * 64 + (lane ^ 5)) * 16;                    // permuted 16-byte pieces
    if (mode == 1 && (lane & 1)) voff = 0x7fffffff;
    if (mode == 2) {
        if (lane < 32) dma16(rs, dst, voff, 0);
    } else if (mode == 3) {
        dma16_imm64(rs, dst, voff);
    } else if (mode == 4) {
        dma16(rs, dst, voff, 4096);
    } else {
        dma16(rs, dst, voff, 0);
    }
    const u32x4 fence = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, 0, 0);   // compiler-visible, issued after the DMA
    asm volatile("" ::"v"(fence));                                                  // its use: the compiler waits here
    __syncthreads();
    for (int i = tid; i < 4096 / 16; i += 256) reinterpret_cast<u32x4*>(out)[i] = reinterpret_cast<const u32x4*>(smem + LDS_BASE)[i];
    if (tid == 0) out[1024] = fence.x;
}

// fence-load race test: each workgroup DMAs 40 wave-instructions (40 KB) from its own cold region, then checks in-kernel.
__global__ __launch_bounds__(512) void fence_race(const unsigned* src, unsigned* bad, int nbytes, int rounds, size_t wg_stride_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned nbad = 0;
    for (int r = 0; r < rounds; ++r) {
        const size_t base = ((size_t)blockIdx.x * rounds + r) * wg_stride_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned*>(src) + base / 4, 0, (int)wg_stride_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < 5; ++k) {                            // wave w: instructions w, w + 8, ...
            const int ins = wave + 8 * k;
            dma16(rs, (unsigned)(uintptr_t)(smem + LDS_BASE) + ins * 1024, (ins * 64 + lane) * 16, 0);
        }
        const u32x4 fence = __builtin_amdgcn_raw_buffer_load_b128(rs, 0, 0, 0);
        asm volatile("" ::"v"(fence));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int id = k * 512 + tid;                        // 16-byte piece id of the 40 KB
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + LDS_BASE + id * 16);
            const unsigned w0 = (unsigned)((base / 4 + id * 4) * 2654435761u);   // the host's pattern: word i = i * 2654435761
            nbad += v.x != w0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (nbad) atomicAdd(bad, nbad);
}

// cost: 43.5 KB per workgroup per round, (a) DMA (b) registers + ds_write_b128; both followed by a barrier
template <int DMA>
__global__ __launch_bounds__(512) void halo_cost(const unsigned* src, long long* cyc, unsigned* sink, int nbytes, int rounds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
    unsigned acc = 0;
    const long long t0 = clock64();
    for (int r = 0; r < rounds; ++r) {
        const int rbase = ((blockIdx.x * 131 + r * 7) % 1024) * 45056;
        if constexpr (DMA) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int ins = wave + 8 * k;
                if (ins < 43) dma16(rs, (unsigned)(uintptr_t)(smem) + ins * 1024, rbase + (ins * 64 + lane) * 16, 0);
            }
            const u32x4 fence = __builtin_amdgcn_raw_buffer_load_b128(rs, rbase, 0, 0);
            asm volatile("" ::"v"(fence));
        } else {
            u32x4 st[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) st[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, rbase + min(k * 512 + tid, 2719) * 16, 0, 0);
#pragma unroll
            for (int k = 0; k < 6; ++k) *reinterpret_cast<u32x4*>(smem + min(k * 512 + tid, 2719) * 16) = st[k];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        acc += *reinterpret_cast<const unsigned*>(smem + ((tid * 52 + r * 16) % 43520 & ~3));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const int LDS = 160 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fence_race), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(halo_cost<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(halo_cost<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    // ---- semantics
    {
        const int nwords = 8192;
        std::vector<unsigned> h(nwords);
        for (int i = 0; i < nwords; ++i) h[i] = 0x10000000u + i;
        unsigned *dsrc, *dout;
        CHECK(hipMalloc(&dsrc, nwords * 4));
        CHECK(hipMalloc(&dout, 8192));
        CHECK(hipMemcpy(dsrc, h.data(), nwords * 4, hipMemcpyHostToDevice));
        std::vector<unsigned> o(2048);
        for (int mode = 0; mode < 5; ++mode) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(256), LDS, 0, dsrc, dout, 16384, mode);   // resource covers the first 16 KB only
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(o.data(), dout, 8192, hipMemcpyDeviceToHost));
            int ok = 0, zero = 0, sentinel = 0, shifted_mem = 0, other = 0;
            for (int w = 0; w < 4; ++w)
                for (int l = 0; l < 64; ++l) {
                    const unsigned got = o[(w * 64 + l) * 4];
                    const unsigned piece = w * 64 + (l ^ 5);
                    const unsigned want = 0x10000000u + piece * 4;
                    if (got == want) ++ok;
                    else if (got == 0) ++zero;
                    else if (got == 0xABABABABu) ++sentinel;
                    else if (got == want + 16 || got == want + 1024) ++shifted_mem;
                    else ++other;
                }
            printf("mode %d: as-expected %d, zero %d, untouched %d, source shifted by imm/soffset %d, other %d", mode, ok, zero, sentinel, shifted_mem, other);
            if (mode == 3) {   // where did lane 0 of wave 0 land, and what did it carry?
                int found = -1;
                for (int i = 0; i < 1024; ++i) if (o[i] == 0x10000000u + (0 ^ 5) * 4 + 16 || o[i] == 0x10000000u + (0 ^ 5) * 4) { found = i; break; }
                printf("  [imm 64: piece of lane 0 found at LDS word %d, value %#x]", found, found >= 0 ? o[found] : 0);
            }
            printf("\n");
        }
        CHECK(hipFree(dsrc));
        CHECK(hipFree(dout));
    }
    // ---- fence-load race
    {
        const int wgs = 512, rounds = 24;
        const size_t stride = 40960;
        const size_t bytes = (size_t)wgs * rounds * stride;          // 503 MB: cold
        std::vector<unsigned> h(bytes / 4);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
        unsigned *dsrc, *dbad;
        CHECK(hipMalloc(&dsrc, bytes));
        CHECK(hipMalloc(&dbad, 4));
        CHECK(hipMemcpy(dsrc, h.data(), bytes, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(dbad, 0, 4));
            hipLaunchKernelGGL(fence_race, dim3(wgs), dim3(512), LDS, 0, dsrc, dbad, (int)stride, rounds, stride);
            CHECK(hipDeviceSynchronize());
            unsigned bad = 0;
            CHECK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
            printf("fence-load race rep %d: %u wrong pieces of %d  %s\n", rep, bad, wgs * rounds * 2560, bad ? "FAIL" : "OK");
        }
        CHECK(hipFree(dsrc));
        CHECK(hipFree(dbad));
    }
    // ---- cost
    {
        const size_t bytes = (size_t)1024 * 45056 + 65536;
        unsigned *dsrc, *dsink;
        long long* dcyc;
        CHECK(hipMalloc(&dsrc, bytes));
        CHECK(hipMalloc(&dsink, 4));
        CHECK(hipMalloc(&dcyc, 256 * 8));
        CHECK(hipMemset(dsrc, 1, bytes));
        std::vector<long long> c(256);
        for (int dma = 0; dma < 2; ++dma)
            for (int rep = 0; rep < 2; ++rep) {
                if (dma) hipLaunchKernelGGL(halo_cost<1>, dim3(256), dim3(512), LDS, 0, dsrc, dcyc, dsink, (int)bytes, 64);
                else hipLaunchKernelGGL(halo_cost<0>, dim3(256), dim3(512), LDS, 0, dsrc, dcyc, dsink, (int)bytes, 64);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(c.data(), dcyc, 256 * 8, hipMemcpyDeviceToHost));
                double s = 0;
                for (auto v : c) s += v;
                printf("halo 43.5 KB/round, 256 WGs, %s: %.0f cycles per round\n", dma ? "LDS-DMA" : "registers + ds_write_b128", s / 256 / 64);
            }
    }
    return 0;
}

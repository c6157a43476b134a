// C-ABI of libpfnl_hip.so (see include/pfnl_hip.h): handle, weight repacking, the forward
// schedule of PFNL.forward (reference model/pfnl.py:39-80) as a sequence of HIP kernel launches.
#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pfnl_hip.h"
#include "capi_internal.h"
#include "common.h"
#include "conv_bf16.h"
#include "conv_split16.h"
#include "conv_small.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
}  // namespace

int pfnl_internal_fail(int code, const std::string& msg) { return fail(code, msg); }

namespace {

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(PFNL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)

static unsigned long long g_alloc_gen = 0;   // bumped whenever a device buffer moves: captured graphs hold raw pointers

struct DevBuf {
    float* p = nullptr;
    size_t n = 0;   // floats
    int ensure(size_t count) {
        if (count <= n) return 0;
        ++g_alloc_gen;
        if (p) hipFree(p);
        p = nullptr;
        n = 0;
        if (hipMalloc(&p, count * sizeof(float)) != hipSuccess) return -1;
        n = count;
        return 0;
    }
    void release() {
        if (p) hipFree(p);
        p = nullptr;
        n = 0;
    }
};

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

// Pinned (page-locked) host memory of the handle: the landing / take-off strip of host-pointer forwards.
struct PinBuf {
    unsigned char* p = nullptr;
    size_t n = 0;   // bytes
    int ensure(size_t bytes) {
        if (bytes <= n) return 0;
        release();
        if (hipHostMalloc(reinterpret_cast<void**>(&p), bytes, hipHostMallocDefault) != hipSuccess) {
            p = nullptr;
            return -1;
        }
        n = bytes;
        return 0;
    }
    void release() {
        if (p) hipHostFree(p);
        p = nullptr;
        n = 0;
    }
};

// A few worker threads that move bytes between a caller's PAGEABLE buffers and the pinned strip while the copy engines move the
// pinned side over PCIe (a host-pointer pfnl_forward is what the reference's sess.run timing covers, model/pfnl.py:249-253).
// One job at a time: run(n, f) hands out chunk indices 0..n-1 in order; done(i) / wait() observe completion.
struct HostPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::function<void(int)> fn;
    std::unique_ptr<std::atomic<int>[]> flag;
    int cap = 0, n = 0, next = 0, finished = 0, device = 0;   // (a job is only started after the previous one has been waited for)
    bool stop = false;
    void start(int k, int dev) {
        if (!th.empty()) return;
        device = dev;
        for (int i = 0; i < k; ++i) th.emplace_back([this] { loop(); });
    }
    void loop() {
        hipSetDevice(device);
        for (;;) {
            int i;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop || next < n; });
                if (stop) return;
                i = next++;
            }
            fn(i);
            flag[i].store(1, std::memory_order_release);
            {
                std::lock_guard<std::mutex> lk(m);
                ++finished;
            }
            cv_done.notify_all();
        }
    }
    void run(int count, std::function<void(int)> f) {
        std::lock_guard<std::mutex> lk(m);
        if (count > cap) {
            flag.reset(new std::atomic<int>[count]);
            cap = count;
        }
        for (int i = 0; i < count; ++i) flag[i].store(0, std::memory_order_relaxed);
        fn = std::move(f);
        n = count;
        next = 0;
        finished = 0;
        cv_work.notify_all();
    }
    void wait_chunk(int i) {
        while (!flag[i].load(std::memory_order_acquire)) std::this_thread::yield();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return finished >= n; });
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_work.notify_all();
        for (auto& t : th) t.join();
        th.clear();
    }
};

}  // namespace

// a 3x3 launch with fewer tiles of 8x32 pixels than this takes the small-shape trunk (conv_small.hip) under the default choices; from
// here on the split-f16 kernels' per-tile launches are faster (tools/precision_ladder.py: 210 - 224 tiles 1.99 - 2.14 -> 1.85 - 1.90 ms;
// 168 tiles 1.08 against 1.88).  256 (a tile per CU) until round 5.
// Both work-order thresholds were measured on the 256-CU part and are a fraction of the CUs a launch can occupy: they scale with
// device_cu_count() (a CPX partition or a smaller device keeps the same tiles-per-CU crossover).
static constexpr int kSmallTiles256 = 200;                  // tiles of 8x32 pixels per 3x3 launch below which the small-shape trunk runs (0.78 per CU)
static constexpr int kMidChains256 = 136;                   // (clip, tile) chains below which a block runs as four per-tile launches (0.53 per CU)
static int scaled_by_cus(int v256) {
    const int ncu = pfnl::device_cu_count();
    return ncu > 0 && ncu != 256 ? std::max(1, (int)((long long)v256 * ncu / 256)) : v256;
}

struct pfnl_handle {
    pfnl_config cfg;
    hipStream_t stream = nullptr;
    std::map<std::string, std::vector<int64_t>> expected;   // tf name -> shape
    std::map<std::string, HostTensor> host;                  // tensors received so far
    bool finalized = false;
    bool weights_external = false;                           // the device blobs came from another handle (copy / broadcast): `host` holds placeholders
    size_t wdev_used = 0, wdev16_used = 0, wdev16s_used = 0;  // floats of each blob the current weights occupy (DevBuf.n is a grow-only capacity)
    // launch-bound shapes (e.g. BASELINE.json configs[0], 1x7x32x32: ~95 launches of a few us each): the whole
    // forward is captured once per (B, H, W) into a hipGraph between the staging buffers and replayed
    struct GraphEntry {
        int B, H, W, seen;
        hipGraph_t graph;
        hipGraphExec_t exec;
        unsigned long long alloc_gen, cfg_gen;
        unsigned long long seen_cfg_gen;                          // cfg_gen at the eager run that `seen` counts
        int slot;                                                 // range-flag word the captured tail kernel writes (0 synchronous / 1 asynchronous calls)
    };
    std::vector<GraphEntry> graphs;
    unsigned long long cfg_gen = 0;                           // bumped by finalize_weights / set_option
    hipEvent_t gev = nullptr;                                  // ordering with the legacy default stream around a replay
    int graph_mode = 0;                                       // 0 off (default: measured slower), 1 auto (frames*H*W <= 65536 pixels), 2 on
    bool conv2_grouped = true;                                // winograd: conv2_i as one grouped launch (option conv2=grouped|split)
    int merge_cstride = 48;                                   // floats per pixel of `merge` as written by the last forward
    int conv1x1_algo = 2;                                     // conv10: 2 streaming kernel on the f16 pipe, split operands (default), 1 streaming f32-MFMA kernel (conv1x1.hip), 0 LDS-tiled implicit GEMM
    bool bf16_fuse10 = true;                                  // bf16 trunk: conv10_i inside the conv1_i launch (option bf16_conv10=fused|separate)
    bool bf16_m16 = true;                                     // bf16 trunk: the two chained 3x3 launches on v_mfma_f32_16x16x32_bf16 (option bf16_mfma=16|32)
    std::vector<size_t> off16_c1_m16, off16_c2b_m16;          // conv3x3_bf16_pack_weights16 of conv1_i / conv2_i's per-frame half
    bool bf16 = false;                                        // option precision=bf16: progressive-fusion trunk in bf16 (conv_bf16.hip); NL, conv0 maths, merge, tail stay fp32
    DevBuf wdev16;                                            // bf16 packs (offsets in 16-bit elements)
    std::vector<size_t> off16_c1, off16_c10, off16_c2a, off16_c2b;
    size_t off16_m1 = 0;                                      // convmerge1: T consecutive packs (cout 48 zero-padded to 64)
    size_t off16s_m1 = 0;                                     // convmerge1, split-f16 packs: T consecutive (frame, both halves) packs
    int m1_algo = 0;                                          // convmerge1 with conv3x3=split16: 0 auto (= 1), 1 the split-f16 kernel's accumulating mode, 2 Winograd
    int conv_algo = 5;                                        // conv3x3: 5 auto (4 for large shapes, 3 for small), 0 direct, 1 winograd (4 waves / tile), 3 winograd_ws (persistent, wave-specialised), 4 split16 (f16 MFMA, split fp32 operands)
    int nl_algo = 2;                                          // non-local block of the fp32 path: 0 f32 MFMA (nonlocal.hip), 1 split-f16 (nonlocal_f16.hip), 2 auto (1 from N = 1024 keys)
    DevBuf wdev16s;                                           // split-f16 packs of the 3x3 kernels (offsets in 16-bit elements)
    std::vector<size_t> off16s_c1, off16s_c2a, off16s_c2b, off16s_c10, off16s_c10f;   // (c10f: conv10_i as conv3x3_c1c10_kernel takes it)
    std::vector<size_t> off16s_c2a_sf, off16s_c2b_sf;         // ... with the identity row map conv3x3_sf_kernel takes (conv_sf.hip)
    std::vector<size_t> off16s_c2a_m16, off16s_c2b_m16;       // ... in the order of the v_mfma_f32_16x16x32_f16 chain kernel (conv3x3_split16_pack_weights16)
    bool s16_m16 = true;                                      // option split16_mfma=16|32: the chain launch of conv2_i on v_mfma_f32_16x16x32_f16
    std::vector<size_t> off16m_c1, off16m_c10, off16m_c2;     // small-shape packs (conv_small.hip), in the same blob
    size_t off16m_m1 = 0;
    // The f16-pipe kernels of the fp32 path have a DOMAIN (operands inside binary16's range; the non-local kernel: inputs x 2^7).
    // Beyond it an operand becomes inf and the result non-finite - the tail kernel, which every output value passes through, ORs
    // that into `rflag` (sticky device word).  Host-pointer calls read it before they return and, when it is set, redo the call on
    // the f32-MFMA kernels (no such domain: the reference's own arithmetic range); device-pointer calls stay asynchronous and
    // pfnl_sync reports PFNL_ERR_RANGE.  Option strict_fp32=on takes the f32-MFMA kernels from the start.
    // Two sticky words in pinned, device-mapped host memory (written by the tail kernel with a system-scope store, read by the host
    // without a copy once the stream is idle): [0] belongs to SYNCHRONOUS (host-pointer) calls, [1] to asynchronous (device-pointer)
    // calls and is what pfnl_sync / pfnl_range_flag report - a host-pointer call never consumes a flag an earlier asynchronous call left.
    unsigned* rflag_host = nullptr;
    unsigned* rflag_dev = nullptr;
    // host-pointer forwards: pinned strips + the threads that copy between them and the caller's pageable buffers
    PinBuf pin_in, pin_out;
    HostPool pool;
    std::vector<hipEvent_t> d2h_ev;
    bool strict = false, strict_once = false, weights_f16_ok = true;
    long long range_reruns = 0;
    int small_mode = 0;                                       // option small=auto|on|off: the small-shape trunk kernels (auto: when a launch has < 256 tiles of 8x32 pixels)
    bool small_c10 = true;                                    // ... with conv10_i inside the conv1_i launch (per-frame partials, summed in conv2_i's prologue): option small_c10=on|off
    DevBuf p10;                                               // ... those partials [B*T][H][W][64] fp32
    bool sf_chain = true;                                     // ... and conv2_i is ONE launch (option split16_chain=on|off)
    bool sf_c10 = true;                                       // ... and conv1_i + conv10_i are ONE launch (option split16_c10=on|off)
    bool sf_mid = true;                                       // option split16_mid=auto|off: launches with fewer (clip, tile) chains than sf_mid_chains run the block as four per-tile launches
    int sf_mid_chains = 0;                                    // 0: kMidChains256 scaled by the device's CUs (measured crossover, tools/precision_ladder.py; env PFNL_SF_MID_CHAINS for sweeps)
    bool sf0 = false;                                         // option split16_sf0=off|on: in the two-launch block the chain kernel ALSO writes the block's output in the split
                                                              // format (`inp0sf`), and the next block's conv1_i + conv10_i launch takes its halo from there by LDS-DMA (round 6).
                                                              // Bit-identical; MEASURED SLOWER (configs[1], same box: 4.86 vs 4.45 ms - the chain kernel pays 22 us for the copy,
                                                              // conv1_i + conv10_i gains 0.6: DESIGN.md R6.1), hence off
    DevBuf inp0sf;                                            // ... that copy [B*T][H][W] x 256 B
    bool split_chains = true;                                 // option split16_splitchains=auto|off: in the two-launch block, a batch that is not a whole number of rounds of
                                                              // (clip, tile) chains runs its last, partial round as PARTS of chains cut by frames (conv_split16.h "SPLIT CHAINS")
    DevBuf c10part;                                           // ... conv10_i's partial sums of those parts: [slot][8][32][64] fp32
    bool sf_path = true;                                      // option split16_sf=on|off: with conv3x3 = conv1x1 = split16, conv1_i and conv10_i write the
                                                              // split format (conv_split16.h) and both halves of conv2_i read it by LDS-DMA (conv_sf.hip)

    // device weights (offsets in floats into `wdev`)
    DevBuf wdev;
    size_t off_conv0_w = 0, off_conv0_b = 0;
    std::vector<size_t> off_c1_w, off_c1_b, off_c10_w, off_c10_b, off_c10_s, off_c2a_w, off_c2b_w, off_c2_b;
    std::vector<size_t> off_c1_u, off_c2a_u, off_c2b_u;       // Winograd-packed variants
    std::vector<size_t> off_m1_u;                             // convmerge1 per frame, Winograd pack (cout 48 padded to 64)
    size_t off_m1_w = 0, off_m1_b = 0, off_m2_w = 0, off_m2_b = 0, off_nl_w = 0, off_nl_b = 0, off_zero = 0;

    // embedded-Gaussian option of the non-local block (utils.py nltype 0): optional theta / phi projections
    std::map<std::string, std::vector<int64_t>> optional;    // tf name -> shape (all four or none)
    bool nl_theta = false;
    size_t off_nl_m = 0, off_nl_c = 0;                        // M = Wt Wp^T [CP][CP], c = bt Wp^T [CP]
    DevBuf Q;                                                 // projected queries [B][N][CP]
    DevBuf Xs;                                                // average-pooled keys / values [B][Nk][CP] (nl_sub_sample > 1)
    int nl_type = -1;                                         // utils.NonLocalBlock nltype: -1 auto (0 with theta / phi variables, else 1 = PFNL's call), 0, 1, 2
    int nl_sub = 1;                                           // ... sub_sample (utils.py:27-28,35-36); PFNL's call: 1

    // workspace
    DevBuf nl16;                                              // bf16 non-local: split K / V^T operands
    DevBuf X, Xo, nlp, inp0, inp1, base, pb, merge, stage_in, stage_out, scratch;
    int lastB = 0, lastH = 0, lastW = 0;

    // profiling: boundary events.  One event after every kernel launch (plus one at the start of a
    // forward); a launch's time = its event - the previous event, i.e. kernel + the gap before it.
    bool prof = false;
    int prof_mode = 0;                // 1: every launch; 2: every launch outside the PF blocks + every 4th PF block; 3: the launches of one PF block only
    bool prof_gate = true;            // events are recorded for the launches issued now
    bool chain_open = false;          // an event has been recorded in the current forward
    std::vector<hipEvent_t> evs;
    std::vector<int> ev_cls;          // class of the interval ENDING at event i (-1: chain start)
    std::vector<int> ev_cnt;          // launches inside that interval (1, except in mode 4)
    // mode 4 ("class runs", for launch-bound shapes where an event per launch costs a fifth of the forward): ONE event where the class
    // of consecutive launches changes (and at the end of the forward): ~8 events per forward; the run's launches are counted
    int run_cls = -1, run_n = 0;
    size_t evs_used = 0;
    double prof_ms[PFNL_K_COUNT] = {0};
    int64_t prof_n[PFNL_K_COUNT] = {0};
};

namespace {

using namespace pfnl;

// sampled profiling (mode 2): which PF blocks get events - two of twenty (blocks 3 and 13: an event costs the stream ~2 us and
// the blocks are identical), every 4th below ten blocks.  bench.py scales the in-block classes by num_block / (blocks sampled).
inline bool prof_sampled(int nb, int i) { return nb >= 10 ? (i % 10) == 3 : (i & 3) == 0; }

int prof_mark(pfnl_handle* h, hipStream_t s, int cls, int count = 1) {
    if (h->evs_used == h->evs.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        h->evs.push_back(e);
        h->ev_cls.push_back(-1);
        h->ev_cnt.push_back(1);
    }
    h->ev_cls[h->evs_used] = cls;
    h->ev_cnt[h->evs_used] = count;
    return hipEventRecord(h->evs[h->evs_used++], s) == hipSuccess ? 0 : -1;
}

// mode 4: closes the run of same-class launches that is open (at a class change, and where a forward ends)
void prof_flush_run(pfnl_handle* h, hipStream_t s) {
    if (h->prof && h->prof_mode == 4 && h->run_n > 0) prof_mark(h, s, h->run_cls, h->run_n);
    h->run_n = 0;
    h->run_cls = -1;
}

// roctx ranges around every kernel class (the tracing hook of this path): env PFNL_ROCTX=1 resolves
// roctxRangePushA / roctxRangePop at run time (librocprofiler-sdk-roctx, else libroctx64; never linked) - under
// `rocprofv3 --marker-trace --kernel-trace` the launches of a forward then sit inside named ranges (nl_pack, nl_attn, conv0,
// conv3x3, conv1x1, merge1, tail), per PF block in launch order.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
const Roctx* roctx() {
    static const Roctx r = [] {
        Roctx x;
        const char* e = std::getenv("PFNL_ROCTX");
        if (!e || !*e || *e == '0') return x;
        for (const char* n : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4",
                              "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "/opt/rocm/lib/libroctx64.so"}) {
            if (void* lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) {
                x.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
                x.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
                if (x.push && x.pop) break;
                x.push = nullptr;
                x.pop = nullptr;
            }
        }
        return x;
    }();
    return &r;
}
const char* const kClassNames[PFNL_K_COUNT] = {"pfnl:nl_pack", "pfnl:nl_attn", "pfnl:conv0", "pfnl:conv3x3", "pfnl:conv1x1", "pfnl:merge1", "pfnl:tail"};

struct ProfScope {
    pfnl_handle* h;
    hipStream_t s;
    int cls;
    bool ranged = false;
    ProfScope(pfnl_handle* h_, hipStream_t s_, int cls_) : h(h_), s(s_), cls(cls_) {
        if (roctx()->push && cls >= 0 && cls < PFNL_K_COUNT) {
            roctx()->push(kClassNames[cls]);
            ranged = true;
        }
        if (h && h->prof && h->prof_mode == 4 && h->run_n > 0 && h->run_cls != cls) prof_flush_run(h, s);
        if (h && h->prof && h->prof_gate && !h->chain_open) {
            prof_mark(h, s, -1);
            h->chain_open = true;
        }
    }
    ~ProfScope() {
        if (h && h->prof && h->prof_mode == 4) {
            h->run_cls = cls;
            ++h->run_n;
        } else if (h && h->prof && h->prof_gate) {
            prof_mark(h, s, cls);
        }
        if (ranged) roctx()->pop();
    }
};

int prof_collect(pfnl_handle* h) {
    for (size_t i = 0; i < h->evs_used; ++i) {
        const int cls = h->ev_cls[i];
        if (cls < 0 || i == 0) continue;
        if (hipEventSynchronize(h->evs[i]) != hipSuccess) return -1;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->evs[i - 1], h->evs[i]) != hipSuccess) return -1;
        h->prof_ms[cls] += ms;
        h->prof_n[cls] += h->ev_cnt[i];
    }
    h->evs_used = 0;
    h->chain_open = false;
    return 0;
}

int check_geometry(const pfnl_config& c) {
    if (c.num_frames != 3 && c.num_frames != 5 && c.num_frames != 7)
        return fail(PFNL_ERR_INVALID, "num_frames must be 3, 5 or 7");
    if (c.scale != 2 && c.scale != 4) return fail(PFNL_ERR_INVALID, "scale must be 2 or 4");
    if (c.mf != 64) return fail(PFNL_ERR_INVALID, "mf must be 64");
    if (c.num_block < 0 || c.num_block > 1024) return fail(PFNL_ERR_INVALID, "bad num_block");
    for (int r : c.reserved)
        if (r != 0) return fail(PFNL_ERR_INVALID, "reserved config fields must be 0");
    return 0;
}

void add_expected(pfnl_handle* h, const std::string& layer, int k, int cin, int cout) {
    h->expected["nlvsr/" + layer + "/kernel"] = {k, k, cin, cout};
    h->expected["nlvsr/" + layer + "/bias"] = {cout};
}

size_t numel(const std::vector<int64_t>& s) {
    size_t n = 1;
    for (auto d : s) n *= (size_t)d;
    return n;
}

// THE LAUNCH PLAN of the progressive-fusion trunk for a shape under the handle's current options: the one place the dispatch rule lives.
// forward_device runs it, pfnl_workspace_bytes sizes from it, pfnl_plan reports it (bench.py's byte model and the tests read it there).
struct TrunkPlan {
    bool bf16 = false, strict = false;
    // bf16 trunk
    bool bmid = false, fuse10 = false;
    // fp32 trunk
    int algo = 0, conv1x1_algo = 0;        // resolved 3x3 / 1x1 algorithm (conv_algo 5 = auto is resolved here)
    bool sf = false;                       // inp1 and base in the split format
    bool small = false, small_c10 = false; // conv_small.hip: 3 (2 with small_c10) launches per block
    bool mid = false;                      // four per-tile launches per block
    bool c10_fused = false;                // conv1_i + conv10_i in one launch (conv3x3_c1c10_kernel)
    bool chain = false;                    // conv2_i in one launch (conv3x3_sf_chain_kernel)
    bool sf0 = false;                      // chain2 only: split-format copy of the block output, conv1_i's halo by LDS-DMA
    bool conv2_grouped = false;            // Winograd: conv2_i as one grouped launch
    int n_full = 0, split_s = 0, split_q = 0;   // chain2 only: SPLIT CHAINS (conv_split16.h) - the chains behind the first n_full are cut into split_s parts of <= split_q frames
    int c1x1_launches = 0;                 // launches per block of class conv1x1 (conv10_i on its own / c10_finalize_kernel)
    int launches_per_block = 0;
    int tiles8x32 = 0, chains = 0;
    int mfma = 32;                         // MFMA shape of the chained 3x3 launches: 16 = v_mfma_f32_16x16x32_* (bf16: conv_bf16_v3.hip M16; fp32: the chain launch
                                           // of conv2_i, conv3x3_sf_chain16_kernel - whole rounds of at least a chain per CU only), 32 = 32x32x16
    const char* name = "";
};

TrunkPlan trunk_plan(const pfnl_handle* h, int B, int H, int W) {
    TrunkPlan pl;
    const int T = h->cfg.num_frames;
    pl.bf16 = h->bf16;
    pl.tiles8x32 = B * T * ((W + 31) / 32) * ((H + 7) / 8);
    pl.chains = pl.tiles8x32 / T;
    const int mid_chains = h->sf_mid_chains > 0 ? h->sf_mid_chains : scaled_by_cus(kMidChains256);
    const int small_tiles = scaled_by_cus(kSmallTiles256);
    const bool fits32 = (long long)H * W * 256 < 0x7fffffffLL;
    if (h->bf16) {
        // MID shapes (as in the fp32 trunk): with fewer (clip, tile) chains than mid_chains the chained launches leave most CUs idle
        pl.bmid = h->sf_mid && h->bf16_fuse10 && pl.chains < mid_chains;
        pl.fuse10 = h->bf16_fuse10 && !pl.bmid;
        pl.launches_per_block = pl.fuse10 ? 3 : 4;
        pl.c1x1_launches = pl.fuse10 ? 0 : 1;
        if (pl.fuse10 && h->split_chains && T <= 7 && fits32) {        // SPLIT CHAINS of the two chained launches (the rule of the fp32 trunk below)
            const int grid = conv_split16_grid();
            const int R = pl.chains % grid;
            if (pl.chains > grid && R > 0 && grid / R >= 2) {
                const int s0 = std::min(T, grid / R), q = (T + s0 - 1) / s0, s = (T + q - 1) / q;
                if (s >= 2) {
                    pl.n_full = pl.chains - R;
                    pl.split_s = s;
                    pl.split_q = q;
                    pl.launches_per_block += 1;                         // c10_finalize_bf16_kernel
                    pl.c1x1_launches = 1;
                }
            }
        }
        pl.name = pl.bmid ? "bf16_mid4" : (pl.fuse10 ? (pl.split_s ? "bf16_3_split" : "bf16_3") : "bf16_4");
        pl.mfma = h->bf16_m16 ? 16 : 32;
        return pl;
    }
    pl.strict = h->strict || h->strict_once || !h->weights_f16_ok;   // f32-MFMA kernels only
    // conv3x3 = auto (default): the split-f16 kernels when a launch has at least ~0.78 tiles per CU, the Winograd f32 kernel below
    const int algo0 = h->conv_algo == 5 ? ((pl.tiles8x32 >= small_tiles && fits32) ? 4 : 3) : h->conv_algo;
    pl.algo = (pl.strict && algo0 == 4) ? 3 : algo0;
    pl.conv1x1_algo = (pl.strict && h->conv1x1_algo == 2) ? 1 : h->conv1x1_algo;
    pl.sf = pl.algo == 4 && pl.conv1x1_algo == 2 && h->sf_path;
    // small shapes (BASELINE.json configs[0], configs[4]): the trunk through conv_small.hip; only under the default algorithm choices
    pl.small = !pl.strict && fits32 &&
               (h->small_mode == 1 || (h->small_mode == 0 && h->conv_algo == 5 && h->conv1x1_algo == 2 && pl.tiles8x32 < small_tiles));
    pl.small_c10 = pl.small && h->small_c10;
    if (pl.small) {
        pl.launches_per_block = pl.small_c10 ? 2 : 3;
        pl.c1x1_launches = pl.small_c10 ? 0 : 1;
        pl.name = pl.small_c10 ? "small2" : "small3";
        return pl;
    }
    pl.mid = pl.sf && h->sf_mid && h->conv_algo == 5 && h->sf_c10 && h->sf_chain && pl.chains < mid_chains;   // (only under the default choices, like `small`)
    pl.c10_fused = pl.sf && h->sf_c10 && !pl.mid;
    pl.chain = pl.sf && h->sf_chain && !pl.mid;
    pl.sf0 = pl.c10_fused && pl.chain && h->sf0;
    const int wino_groups = B * ((W + 31) / 32) * ((H + 3) / 4);
    pl.conv2_grouped = pl.algo == 3 && h->conv2_grouped && wino_groups >= 224 && fits32;
    pl.launches_per_block = (pl.c10_fused ? 1 : 2) + ((pl.chain || pl.conv2_grouped) ? 1 : 2);
    pl.c1x1_launches = pl.c10_fused ? 0 : 1;
    if (pl.c10_fused && pl.chain && h->split_chains && T <= 7) {
        // SPLIT CHAINS: with R = chains mod grid chains in a last, partial round, a workgroup with one chain more than the others sets the time of
        // both launches (5 clips of 128x128 = 1.25 rounds: 7.2 ms against 4.5 for 4).  When at least two parts of a chain fit the idle
        // workgroups (R <= grid / 2), those R chains are cut by frames: s = grid / R parts of q = ceil(T / s) frames.
        const int grid = conv_split16_grid();
        const int R = pl.chains % grid;
        if (pl.chains > grid && R > 0 && grid / R >= 2) {
            const int s0 = std::min(T, grid / R), q = (T + s0 - 1) / s0, s = (T + q - 1) / q;
            if (s >= 2) {
                pl.n_full = pl.chains - R;
                pl.split_s = s;
                pl.split_q = q;
                pl.launches_per_block += 1;                             // c10_finalize_kernel
                pl.c1x1_launches = 1;
            }
        }
    }
    // the chain launch on 16x16x32: where every CU has a chain the launch sits on the power cap and the shape's energy counts; below that (UDM10: 230
    // chains) its extra cycles do (+0.9 %); split chains and the split-format copy stay on the 32x32x16 kernel
    pl.mfma = (h->s16_m16 && pl.c10_fused && pl.chain && !pl.sf0 && !pl.split_s && pl.chains >= conv_split16_grid()) ? 16 : 32;
    pl.name = pl.mid ? "mid4" : (pl.c10_fused && pl.chain) ? (pl.sf0 ? "chain2_sf0" : (pl.split_s ? "chain2_split" : "chain2"))
            : pl.algo == 4 ? (pl.launches_per_block == 3 ? "split16_3" : "split16_4")
            : pl.algo == 3 ? (pl.conv2_grouped ? "winograd_ws3" : "winograd_ws4")
            : pl.algo == 1 ? "winograd_tile4" : "direct4";
    return pl;
}

// which kernel family the non-local block takes (the same one statement for forward_device and pfnl_plan): 0 the general form (nltype 0 / 2 or
// sub-sampling: f32 MFMA), 1 the f16 pipe with exactly split operands (fp32 precision, from 1024 keys), 2 the f16 pipe on the hi parts (precision
// bf16), 3 f32 MFMA (small key counts, strict_fp32)
int nl_family(const pfnl_handle* h, int N) {
    const int nlt = h->nl_type < 0 ? (h->nl_theta ? 0 : 1) : h->nl_type;
    if (nlt != 1 || h->nl_sub > 1) return 0;
    if (h->bf16) return 2;
    const bool nl_strict = h->strict || h->strict_once || !h->weights_f16_ok;
    return (!nl_strict && (h->nl_algo == 1 || (h->nl_algo == 2 && N >= 1024))) ? 1 : 3;
}

// forward over device buffers
// `strip` != null: only LR rows [strip->yoff + core0, strip->yoff + core1) of the result are produced (single-clip sharding,
// pfnl_forward_strip): the non-local block runs its queries [q0, q1) against ALL keys, the trunk runs on the strip + halo.
int forward_device(pfnl_handle* h, const float* in, float* out, int B, int Hfull, int W, hipStream_t s,
                   const StripGeom* strip = nullptr, int q0 = 0, int q1 = -1, int flag_slot = 1 /* 0: synchronous call, 1: asynchronous */) {
    const pfnl_config& c = h->cfg;
    const int T = c.num_frames, F = B * T;
    const int H = strip ? strip->Hs : Hfull;                       // rows the trunk buffers hold
    const size_t P = (size_t)H * W;
    const int N = (Hfull / 2) * (W / 2);
    const int C = 12 * T, CP = nl_padded_ch(C);
    const float* wd = h->wdev.p;
    const TrunkPlan pl = trunk_plan(h, B, H, W);

    if (h->X.ensure((size_t)B * N * CP) || h->Xo.ensure((size_t)B * N * CP) ||     // (N: the FULL frame - keys are global)
        h->nlp.ensure(nl_partial_floats(B, N, C)) ||
        h->inp0.ensure((size_t)F * P * 64) || h->inp1.ensure((size_t)F * P * 64) ||
        h->base.ensure((size_t)B * P * 64) || h->pb.ensure((size_t)B * P * 64) ||
        h->merge.ensure((size_t)B * P * 64))
        return fail(PFNL_ERR_NOMEM, "workspace allocation failed");
    h->lastB = B;
    h->lastH = H;
    h->lastW = W;

    const bool nl_strict = !h->bf16 && (h->strict || h->strict_once || !h->weights_f16_ok);
    // The range fence is armed only when a kernel with a binary16 DOMAIN runs (fp32 precision off the strict path: split-f16 trunk,
    // non-local block and conv0; bf16 precision: its non-local block and conv0) and a non-finite value cannot be the reference's
    // own result (nltype 2 divides 0 by 0 for a query without a positive affinity, utils.py:59-62).
    const int nlt_fence = h->nl_type < 0 ? (h->nl_theta ? 0 : 1) : h->nl_type;
    unsigned* const rflag = (nl_strict || nlt_fence == 2) ? nullptr : h->rflag_dev + flag_slot;
    // the non-local block on the f16 matrix pipe (PFNL's own call: nltype 1, no sub-sampling; fp32 precision from 1024 keys, bf16 always)
    const int nl_fam = nl_family(h, N);
    const bool nl_fused_pack = (nl_fam == 1 || nl_fam == 2) && nl_f16_fits_one_launch(B, N);   // one pack launch: x -> X fp32 + the binary16 K / V^T operands (round 6)
    {   // model/pfnl.py:55-60 (+ utils.py:18-71)
        ProfScope ps(h, s, PFNL_K_NL_PACK);
        if (nl_fused_pack) {
            if (h->nl16.ensure((nl_f16_scratch_halfs(B, N) + 1) / 2)) return fail(PFNL_ERR_NOMEM, "workspace allocation failed");
            HIPCHK(launch_nl_pack_fused(in, h->X.p, reinterpret_cast<uint16_t*>(h->nl16.p), B, T, Hfull, W, s));
        } else {
            HIPCHK(launch_nl_pack(in, h->X.p, B, T, Hfull, W, s));
        }
    }
    {
        ProfScope ps(h, s, PFNL_K_NL_ATTN);
        const int nlt = h->nl_type < 0 ? (h->nl_theta ? 0 : 1) : h->nl_type;
        if ((nlt == 0 || nlt == 2) && !h->nl_theta)
            return fail(PFNL_ERR_STATE, "nl_type 0 / 2 need the nlblock_0 theta / phi variables (reference utils.py:31-42)");
        if (nlt != 1 || h->nl_sub > 1) {
            // the general form of utils.NonLocalBlock (nltype 0 / 2: queries X M + c; sub_sample: keys = values = avg-pooled X) on
            // the f32-MFMA kernel in both precisions; PFNL's own call (nltype 1, sub_sample 1) takes the branches below
            const float* Kx = h->X.p;
            int Nk = N;
            if (h->nl_sub > 1) {
                const int h2 = Hfull / 2, w2 = W / 2;
                if (h2 / h->nl_sub < 1 || w2 / h->nl_sub < 1) return fail(PFNL_ERR_INVALID, "nl_sub_sample larger than the space_to_depth grid");
                Nk = (h2 / h->nl_sub) * (w2 / h->nl_sub);
                if (h->Xs.ensure((size_t)B * Nk * CP)) return fail(PFNL_ERR_NOMEM, "workspace allocation failed");
                HIPCHK(launch_nl_pool(h->X.p, h->Xs.p, B, h2, w2, h->nl_sub, C, s));
                Kx = h->Xs.p;
            }
            const float* Qp = nullptr;
            if (nlt != 1) {
                if (h->Q.ensure((size_t)B * N * CP)) return fail(PFNL_ERR_NOMEM, "workspace allocation failed");
                HIPCHK(launch_nl_qproj(h->X.p, wd + h->off_nl_m, wd + h->off_nl_c, h->Q.p, B, N, C, s, nlt == 2));
                Qp = h->Q.p;
            }
            HIPCHK(launch_nl_attn_general(h->X.p, Kx, Nk, h->Xo.p, wd + h->off_nl_w, wd + h->off_nl_b, h->nlp.p, B, N, C, s, Qp, q0, q1, nlt == 2));
        } else if (nl_fam == 1) {   // fp32 path on the f16 pipe, exactly split operands
            if (h->nl16.ensure((nl_f16_scratch_halfs(B, N) + 1) / 2)) return fail(PFNL_ERR_NOMEM, "workspace allocation failed");
            HIPCHK(launch_nl_attn_f16(h->X.p, h->Xo.p, wd + h->off_nl_w, wd + h->off_nl_b, h->nlp.p,
                                      reinterpret_cast<uint16_t*>(h->nl16.p), B, N, C, s, q0, q1, true, nl_fused_pack));
        } else if (nl_fam == 2) {   // 16-bit operands throughout: the f16 kernel on the hi parts only
            if (h->nl16.ensure((nl_f16_scratch_halfs(B, N) + 1) / 2)) return fail(PFNL_ERR_NOMEM, "workspace allocation failed");
            HIPCHK(launch_nl_attn_f16(h->X.p, h->Xo.p, wd + h->off_nl_w, wd + h->off_nl_b, h->nlp.p,
                                      reinterpret_cast<uint16_t*>(h->nl16.p), B, N, C, s, q0, q1, false, nl_fused_pack));
        } else {
            HIPCHK(launch_nl_attn(h->X.p, h->Xo.p, wd + h->off_nl_w, wd + h->off_nl_b, h->nlp.p, B, N, C, s, nullptr, q0, q1));
        }
    }
    {   // model/pfnl.py:61-62
        ProfScope ps(h, s, PFNL_K_CONV0);
        if (h->bf16)
            HIPCHK(launch_conv0_bf16(h->Xo.p, wd + h->off_conv0_w, wd + h->off_conv0_b, reinterpret_cast<uint16_t*>(h->inp0.p), B, T, Hfull, W, s, strip));
        else
            HIPCHK(launch_conv0(h->Xo.p, wd + h->off_conv0_w, wd + h->off_conv0_b, h->inp0.p, B, T, Hfull, W, s, strip, nl_strict));
    }
    const float* merge_in = h->inp0.p;
    if (h->bf16) {
        // bf16 trunk (conv_bf16.hip): activations bf16 in the same workspace buffers, fp32 accumulation and biases
        if ((long long)T * H * W * 128 >= 0x7fffffffLL) return fail(PFNL_ERR_INVALID, "frame too large for the bf16 trunk");
        uint16_t* const a0 = reinterpret_cast<uint16_t*>(h->inp0.p);
        uint16_t* const a1 = reinterpret_cast<uint16_t*>(h->inp1.p);
        uint16_t* const ab = reinterpret_cast<uint16_t*>(h->base.p);
        uint16_t* const ap = reinterpret_cast<uint16_t*>(h->pb.p);
        const uint16_t* const w16 = reinterpret_cast<const uint16_t*>(h->wdev16.p);
        // MID shapes (trunk_plan): the chained launches - conv1_i + conv10_i, the per-frame half of conv2_i - leave most CUs idle for T
        // tile times; conv10_i then runs as its own launch and the per-frame half deals out single tiles (bit-identical: this kernel
        // has one summation order)
        const bool bmid = pl.bmid, fuse10 = pl.fuse10;
        if (pl.split_s && h->c10part.ensure((size_t)(pl.chains - pl.n_full) * pl.split_s * 8 * 32 * 64)) return fail(PFNL_ERR_NOMEM, "workspace allocation failed");
        for (int i = 0; i < c.num_block; ++i) {   // model/pfnl.py:65-71
            if (h->prof_mode == 2) {
                h->prof_gate = prof_sampled(c.num_block, i);
                h->chain_open = false;
            } else if (h->prof_mode == 3) {
                h->prof_gate = i == c.num_block / 2;
                h->chain_open = false;
            }
            {   // conv1_i (+ conv10_i from the LDS scratch its tiles pass through: conv_bf16.hip MODE 2)
                ProfScope ps(h, s, PFNL_K_CONV3X3);
                ConvBf16Params q{a0, w16 + h->off16_c1[i], wd + h->off_c1_b[i], nullptr, nullptr, a1, H, W, F, 1, 1};
                if (fuse10) {
                    q.add_div = T;
                    q.x_w = w16 + h->off16_c10[i];
                    q.x_bias = wd + h->off_c10_b[i];
                    q.x_out = ab;
                    q.n_full = pl.n_full;
                    q.split_s = pl.split_s;
                    q.split_q = pl.split_q;
                    q.partial = pl.split_s ? h->c10part.p : nullptr;
                }
                if (pl.mfma == 16) q.wpack16 = w16 + h->off16_c1_m16[i];   // (used where the launch goes to the third-generation kernel)
                HIPCHK(launch_conv3x3_bf16(q, s));
            }
            if (fuse10 && pl.split_s) {   // split chains: the parts' raw conv10_i sums -> base (+ bias, leaky-relu, bf16) for the chains that were cut
                ProfScope ps(h, s, PFNL_K_CONV1X1);
                ConvBf16Params q{};
                q.H = H;
                q.W = W;
                q.items = F;
                q.add_div = T;
                q.n_full = pl.n_full;
                q.split_s = pl.split_s;
                q.split_q = pl.split_q;
                q.partial = h->c10part.p;
                q.x_bias = wd + h->off_c10_b[i];
                q.x_out = ab;
                HIPCHK(launch_c10_finalize_bf16(q, s));
            }
            if (!fuse10) {   // conv10_i as a launch of its own
                ProfScope ps(h, s, PFNL_K_CONV1X1);
                HIPCHK(launch_conv1x1_bf16(a1, w16 + h->off16_c10[i], wd + h->off_c10_b[i], ab, B, T, H * W, 1, s));
            }
            {   // conv2_i, shared half (raw, once per clip)
                ProfScope ps(h, s, PFNL_K_CONV3X3);
                ConvBf16Params q{ab, w16 + h->off16_c2a[i], wd + h->off_zero, nullptr, nullptr, ap, H, W, B, 1, 0};
                HIPCHK(launch_conv3x3_bf16(q, s));
            }
            {   // conv2_i, per-frame half + shared half + bias, lrelu, residual
                ProfScope ps(h, s, PFNL_K_CONV3X3);
                ConvBf16Params q{a1, w16 + h->off16_c2b[i], wd + h->off_c2_b[i], ap, a0, a0, H, W, F, T, 1};
                q.flat = bmid ? 1 : 0;
                q.n_full = pl.n_full;
                q.split_s = pl.split_s;
                q.split_q = pl.split_q;
                if (pl.mfma == 16) q.wpack16 = w16 + h->off16_c2b_m16[i];
                HIPCHK(launch_conv3x3_bf16(q, s));
            }
        }
        h->prof_gate = h->prof_mode != 3;
        if (h->prof_mode == 2 || h->prof_mode == 3) h->chain_open = false;
        {   // convmerge1 (:73-74): the accumulating mode of the bf16 3x3 kernel, fp32 out for the tail
            ProfScope ps(h, s, PFNL_K_MERGE1);
            ConvBf16Params q{a0, w16 + h->off16_m1, wd + h->off_m1_b, nullptr, nullptr, nullptr, H, W, F, T, 1};
            q.out_f32 = h->merge.p;
            HIPCHK(launch_conv3x3_bf16(q, s));
        }
        h->merge_cstride = 64;
        {   // model/pfnl.py:63,76-80
            ProfScope ps(h, s, PFNL_K_TAIL);
            HIPCHK(launch_tail(h->merge.p, in, wd + h->off_m2_w, wd + h->off_m2_b, out, B, T, Hfull, W, c.scale, 64, s, strip, rflag));
        }
        prof_flush_run(h, s);
        h->chain_open = false;
        return 0;
    }

    ConvParams p{};
    p.H = H;
    p.W = W;
    p.in_cstride = 64;
    p.chunks_per_frame = 64 / CONV_CK;
    const int wino_groups = B * ((W + 31) / 32) * ((H + 3) / 4);   // (clip, 4x32-pixel tile) groups of conv_wino_ws
    // the launch structure of a block: trunk_plan (conv3x3 = auto: the split-f16 kernels - persistent, 72 KB of weights per workgroup in
    // their prologue - when a launch has enough tiles, the small-shape trunk / the Winograd f32 kernel below that)
    const int algo = pl.algo, conv1x1_algo = pl.conv1x1_algo;
    const bool sf = pl.sf;                                          // inp1 and base in the split format (conv_split16.h)
    // small shapes (BASELINE.json configs[0], configs[4]): the trunk through conv_small.hip - conv2_i as the reference writes it
    // (3x3 over concat([base, f]))
    const bool small = pl.small;
    if (pl.sf0 && h->inp0sf.ensure((size_t)F * P * 64)) return fail(PFNL_ERR_NOMEM, "workspace allocation failed");
    if (pl.split_s && h->c10part.ensure((size_t)(pl.chains - pl.n_full) * pl.split_s * 8 * 32 * 64)) return fail(PFNL_ERR_NOMEM, "workspace allocation failed");
    const uint16_t* const w16m = reinterpret_cast<const uint16_t*>(h->wdev16s.p);
    for (int i = 0; i < (h->bf16 ? 0 : c.num_block); ++i) {   // model/pfnl.py:65-71
        if (h->prof_mode == 2) {          // sampled profiling: see prof_sampled; each sampled block with a fresh event chain
            h->prof_gate = prof_sampled(c.num_block, i);
            h->chain_open = false;
        } else if (h->prof_mode == 3) {   // the dominant class only: the launches of ONE block (3 events per forward)
            h->prof_gate = i == c.num_block / 2;
            h->chain_open = false;
        }
        if (pl.small_c10) {
            // two launches per block: conv10_i rides in the conv1_i launch as per-frame partials (W10_t^T . inp1_t), which conv2_i's
            // prologue adds up (+ bias, leaky-relu) into its `base` source - no 1x1 launch, no inter-workgroup traffic inside a launch
            if (h->p10.ensure((size_t)F * P * 64)) return fail(PFNL_ERR_NOMEM, "workspace allocation failed");
            {   // conv1_i (:66) + this frame's part of conv10_i (:67-68)
                ProfScope ps(h, s, PFNL_K_CONV3X3);
                ConvSmallParams q{nullptr, h->inp0.p, 0, 1, 1, 1, w16m + h->off16m_c1[i], wd + h->off_c1_b[i], nullptr, h->inp1.p, H, W, F, 1, 3};
                q.x_wpack = w16m + h->off16m_c10[i];
                q.x_out = h->p10.p;
                q.x_T = T;
                HIPCHK(launch_conv_small(q, s));
            }
            {   // conv2_i over concat([base, inp1_t]) + lrelu + residual (:69-71), base = lrelu(sum of the T partials + bias)
                ProfScope ps(h, s, PFNL_K_CONV3X3);
                ConvSmallParams q{h->p10.p, h->inp1.p, 1, T, 1, 2, w16m + h->off16m_c2[i], wd + h->off_c2_b[i], h->inp0.p, h->inp0.p, H, W, F, 1, 3};
                q.a_nsum = T;
                q.a_bias = wd + h->off_c10_b[i];
                HIPCHK(launch_conv_small(q, s));
            }
            continue;
        }
        if (small) {
            {   // conv1_i (:66)
                ProfScope ps(h, s, PFNL_K_CONV3X3);
                ConvSmallParams q{nullptr, h->inp0.p, 0, 1, 1, 1, w16m + h->off16m_c1[i], wd + h->off_c1_b[i], nullptr, h->inp1.p, H, W, F, 1, 3};
                HIPCHK(launch_conv_small(q, s));
            }
            {   // conv10_i (:67-68)
                ProfScope ps(h, s, PFNL_K_CONV1X1);
                ConvSmallParams q{nullptr, h->inp1.p, 0, 1, T, T, w16m + h->off16m_c10[i], wd + h->off_c10_b[i], nullptr, h->base.p, H, W, B, 1, 1};
                HIPCHK(launch_conv_small(q, s));
            }
            {   // conv2_i over concat([base, inp1_t]) + lrelu + residual (:69-71)
                ProfScope ps(h, s, PFNL_K_CONV3X3);
                ConvSmallParams q{h->base.p, h->inp1.p, 1, T, 1, 2, w16m + h->off16m_c2[i], wd + h->off_c2_b[i], h->inp0.p, h->inp0.p, H, W, F, 1, 3};
                HIPCHK(launch_conv_small(q, s));
            }
            continue;
        }
        // MID shapes: the two-launch block deals out CHAINS (a workgroup takes the T frames of a (clip, tile), + the shared half), so a
        // launch with fewer chains than CUs leaves most of the chip idle for T + 2 tile times (1 clip of 128x128: 64 chains - 3.06 ms
        // for a quarter of configs[1]'s work).  Below ~0.53 chains per CU (trunk_plan) the block runs as four launches that deal out
        // single tiles: conv1_i, conv10_i (1x1), the shared half of conv2_i, the per-frame half in flat order.
        const bool mid = pl.mid, c10_fused = pl.c10_fused;
        // chain2_sf0 (round 6): blocks 1 .. nb-1 read the split-format copy of inp0 the previous block's chain kernel wrote (halo by
        // LDS-DMA); block 0 reads conv0's fp32 output, the last block writes no copy (convmerge1 reads fp32).  Bit-identical to chain2.
        const bool in_sf0 = pl.sf0 && i > 0, out_sf0 = pl.sf0 && i + 1 < c.num_block;
        if (c10_fused) {
            // conv1_i AND conv10_i in one launch (conv_split16.hip, conv3x3_c1c10_kernel): per (clip, tile) the T frame tiles of conv1_i
            // leave as split-format lines through LDS, where conv10_i picks them up as MFMA operands; inp1 is written, never read back
            ProfScope ps(h, s, PFNL_K_CONV3X3);
            const uint16_t* const w16s = reinterpret_cast<const uint16_t*>(h->wdev16s.p);
            ConvSplitParams q{in_sf0 ? h->inp0sf.p : h->inp0.p, w16s + h->off16s_c1[i], wd + h->off_c1_b[i], nullptr, nullptr, h->inp1.p, H, W, F, T, 1};
            q.in_sf = in_sf0 ? 1 : 0;
            q.wpack2 = w16s + h->off16s_c10f[i];
            q.bias2 = wd + h->off_c10_b[i];
            q.out2 = h->base.p;
            q.n_full = pl.n_full;
            q.split_s = pl.split_s;
            q.split_q = pl.split_q;
            q.partial = pl.split_s ? h->c10part.p : nullptr;
            HIPCHK(launch_conv3x3_c1c10(q, s));
        }
        if (c10_fused && pl.split_s) {   // split chains: the parts' raw conv10_i sums -> base (+ leaky-relu, split format) for the chains that were cut
            ProfScope ps(h, s, PFNL_K_CONV1X1);
            ConvSplitParams q{};
            q.H = H;
            q.W = W;
            q.items = F;
            q.add_div = T;
            q.act = 1;
            q.n_full = pl.n_full;
            q.split_s = pl.split_s;
            q.split_q = pl.split_q;
            q.partial = h->c10part.p;
            q.out2 = h->base.p;
            HIPCHK(launch_c10_finalize(q, s));
        }
        if (!c10_fused) {   // conv1_i: per frame 3x3 64->64 + lrelu                       (:66)
            ProfScope ps(h, s, PFNL_K_CONV3X3);
            p.in = h->inp0.p;
            p.wpack = wd + h->off_c1_w[i];
            p.bias = wd + h->off_c1_b[i];
            p.addend = nullptr;
            p.resid = nullptr;
            p.out = h->inp1.p;
            p.out_cstride = 64;
            p.cout = 64;
            p.frames_per_item = 1;
            p.nchunks = p.chunks_per_frame;
            p.add_div = 1;
            p.act = 1;
            if (algo == 4) {
                ConvSplitParams q{p.in, reinterpret_cast<const uint16_t*>(h->wdev16s.p) + h->off16s_c1[i], p.bias, nullptr, nullptr, p.out, H, W, F, 1, 1};
                q.out_sf = sf ? 1 : 0;                              // inp1 in the split format: it only feeds conv10_i and conv2_i's MFMA operands
                HIPCHK(launch_conv3x3_split16(q, s));
            } else if (algo == 1 || algo == 3) {
                WinoParams wp{p.in, wd + h->off_c1_u[i], p.bias, nullptr, nullptr, p.out, H, W, 1, 1, F, nullptr};
                HIPCHK(algo == 3 ? launch_conv_wino_ws(wp, s) : launch_conv_wino(wp, s));
            } else {
                HIPCHK(launch_conv_mfma(p, 3, F, s));
            }
        }
        if (!c10_fused) {   // conv10_i: 1x1 over the concat of T frames -> base + lrelu    (:67-68)
            ProfScope ps(h, s, PFNL_K_CONV1X1);
            p.in = h->inp1.p;
            p.wpack = wd + h->off_c10_w[i];
            p.bias = wd + h->off_c10_b[i];
            p.out = h->base.p;
            p.frames_per_item = T;
            p.nchunks = T * p.chunks_per_frame;
            if (conv1x1_algo == 2)
                HIPCHK(launch_conv1x1_split16(p.in, reinterpret_cast<const uint16_t*>(h->wdev16s.p) + h->off16s_c10[i], p.bias, p.out, B, T, H * W, 1, s, sf, sf));
            else if (conv1x1_algo == 1)
                HIPCHK(launch_conv1x1_stream(p.in, wd + h->off_c10_s[i], p.bias, p.out, B, T, H * W, 1, s));
            else
                HIPCHK(launch_conv_mfma(p, 1, B, s));
        }
        // grouped / accumulating modes chain T(+1) units inside one workgroup: only worth it when there are enough
        // (clip, tile) groups to occupy the chip (below ~220 the split launches finish sooner)
        if (pl.conv2_grouped) {
            // the whole of conv2_i in one launch: per (clip, tile) the shared half stays in LDS (conv_wino_ws MODE 2)
            ProfScope ps(h, s, PFNL_K_CONV3X3);
            WinoParams wp{};
            wp.in = h->inp1.p;
            wp.in2 = h->base.p;
            wp.upack = wd + h->off_c2b_u[i];
            wp.upack2 = wd + h->off_c2a_u[i];
            wp.bias = wd + h->off_c2_b[i];
            wp.resid = h->inp0.p;
            wp.out = h->inp0.p;
            wp.H = H;
            wp.W = W;
            wp.add_div = T;
            wp.act = 1;
            wp.items = F;
            HIPCHK(launch_conv_wino_ws(wp, s));
            continue;
        }
        if (pl.chain) {
            // the whole of conv2_i in one launch (conv_sf.hip, conv3x3_sf_chain_kernel): per (clip, tile) the shared half stays in
            // registers as the initial C of the T frame tiles; in place on inp0 (residual)
            ProfScope ps(h, s, PFNL_K_CONV3X3);
            const uint16_t* const w16s = reinterpret_cast<const uint16_t*>(h->wdev16s.p);
            ConvSplitParams q{h->inp1.p, w16s + h->off16s_c2b_sf[i], wd + h->off_c2_b[i], nullptr, h->inp0.p, h->inp0.p, H, W, F, T, 1};
            q.in2 = h->base.p;
            q.wpack2 = w16s + h->off16s_c2a_sf[i];
            q.out2 = out_sf0 ? h->inp0sf.p : nullptr;
            q.n_full = pl.n_full;
            q.split_s = pl.split_s;
            q.split_q = pl.split_q;
            if (pl.mfma == 16) {                                    // (trunk_plan: whole rounds with a chain per CU)
                q.wpack_m16 = w16s + h->off16s_c2b_m16[i];
                q.wpack2_m16 = w16s + h->off16s_c2a_m16[i];
            }
            HIPCHK(launch_conv3x3_sf_chain(q, s));
            continue;
        }
        {   // conv2_i, shared half: 3x3 over `base` (kernel rows 0..63), once per clip, raw
            ProfScope ps(h, s, PFNL_K_CONV3X3);
            p.in = h->base.p;
            p.wpack = wd + h->off_c2a_w[i];
            p.bias = wd + h->off_zero;
            p.out = h->pb.p;
            p.frames_per_item = 1;
            p.nchunks = p.chunks_per_frame;
            p.act = 0;
            if (algo == 4) {
                ConvSplitParams q{p.in, reinterpret_cast<const uint16_t*>(h->wdev16s.p) + (sf ? h->off16s_c2a_sf[i] : h->off16s_c2a[i]), p.bias, nullptr, nullptr, p.out, H, W, B, 1, 0};
                HIPCHK(sf ? launch_conv3x3_sf(q, s) : launch_conv3x3_split16(q, s));
            } else if (algo == 1 || algo == 3) {
                WinoParams wp{p.in, wd + h->off_c2a_u[i], p.bias, nullptr, nullptr, p.out, H, W, 1, 0, B, nullptr};
                HIPCHK(algo == 3 ? launch_conv_wino_ws(wp, s) : launch_conv_wino(wp, s));
            } else {
                HIPCHK(launch_conv_mfma(p, 3, B, s));
            }
        }
        {   // conv2_i, per-frame half (kernel rows 64..127) + shared half + bias, lrelu, residual (:69-71)
            ProfScope ps(h, s, PFNL_K_CONV3X3);
            p.in = h->inp1.p;
            p.wpack = wd + h->off_c2b_w[i];
            p.bias = wd + h->off_c2_b[i];
            p.addend = h->pb.p;
            p.add_div = T;
            p.resid = h->inp0.p;
            p.out = h->inp0.p;
            p.act = 1;
            if (algo == 4) {
                ConvSplitParams q{p.in, reinterpret_cast<const uint16_t*>(h->wdev16s.p) + (sf ? h->off16s_c2b_sf[i] : h->off16s_c2b[i]), p.bias, p.addend, p.resid, p.out, H, W, F, T, 1};
                q.flat = mid ? 1 : 0;
                HIPCHK(sf ? launch_conv3x3_sf(q, s) : launch_conv3x3_split16(q, s));
            } else if (algo == 1 || algo == 3) {
                WinoParams wp{p.in, wd + h->off_c2b_u[i], p.bias, p.addend, p.resid, p.out, H, W, T, 1, F, nullptr};
                HIPCHK(algo == 3 ? launch_conv_wino_ws(wp, s) : launch_conv_wino(wp, s));
            } else {
                HIPCHK(launch_conv_mfma(p, 3, F, s));
            }
        }
    }
    h->prof_gate = h->prof_mode != 3;
    if (h->prof_mode == 2 || h->prof_mode == 3) h->chain_open = false;
    if (small && !h->bf16) {   // convmerge1 (:73-74): T sources, cout 48 zero-padded to 64
        {
            ProfScope ps(h, s, PFNL_K_MERGE1);
            ConvSmallParams q{nullptr, merge_in, 0, 1, T, T, w16m + h->off16m_m1, wd + h->off_m1_b, nullptr, h->merge.p, H, W, B, 1, 3};
            HIPCHK(launch_conv_small(q, s));
        }
        h->merge_cstride = 64;
        {   // model/pfnl.py:63,76-80
            ProfScope ps(h, s, PFNL_K_TAIL);
            HIPCHK(launch_tail(h->merge.p, in, wd + h->off_m2_w, wd + h->off_m2_b, out, B, T, Hfull, W, c.scale, 64, s, strip, rflag));
        }
        prof_flush_run(h, s);
        h->chain_open = false;
        h->prof_gate = h->prof_mode != 3;
        return 0;
    }
    const bool m1_s16 = algo == 4 && h->m1_algo != 2 && (long long)H * W * 256 < 0x7fffffffLL;
    const bool m1_wino = !m1_s16 && (algo == 3 || algo == 4) && wino_groups >= 224 && (long long)H * W * 256 < 0x7fffffffLL;
    const int mstride = (m1_wino || m1_s16) ? 64 : 48;
    h->merge_cstride = mstride;
    if (m1_s16) {
        // convmerge1 (:73-74) on the f16 pipe: the accumulating mode of conv3x3_split16_kernel (the T frame tiles of a clip run
        // through the same accumulators, every unit with its own weights; cout zero-padded to 64; one epilogue per clip tile)
        {
            ProfScope ps(h, s, PFNL_K_MERGE1);
            ConvSplitParams q{merge_in, reinterpret_cast<const uint16_t*>(h->wdev16s.p) + h->off16s_m1, wd + h->off_m1_b, nullptr, nullptr,
                              h->merge.p, H, W, F, T, 1, 1};
            // split chains (trunk_plan): convmerge1's chains are the trunk's - the same last, partial round is cut by frames, the parts' raw sums
            // meet in c10_finalize_kernel (+ bias, leaky-relu); c10part is free behind the last block
            q.n_full = pl.n_full;
            q.split_s = pl.split_s;
            q.split_q = pl.split_q;
            q.partial = pl.split_s ? h->c10part.p : nullptr;
            HIPCHK(launch_conv3x3_split16(q, s));
        }
        if (pl.split_s) {
            ProfScope ps(h, s, PFNL_K_MERGE1);
            ConvSplitParams f{};
            f.H = H;
            f.W = W;
            f.items = F;
            f.add_div = T;
            f.act = 1;
            f.n_full = pl.n_full;
            f.split_s = pl.split_s;
            f.split_q = pl.split_q;
            f.partial = h->c10part.p;
            f.bias = wd + h->off_m1_b;
            f.out = h->merge.p;
            HIPCHK(launch_c10_finalize(f, s));
        }
    } else if (m1_wino) {
        // convmerge1 (:73-74) = sum over the T frames of a 3x3 64->48 convolution: one launch of the persistent
        // Winograd kernel in its accumulating mode (cout zero-padded to 64; the T frame tiles of a clip add into
        // the same accumulators, one epilogue per clip tile)
        ProfScope ps(h, s, PFNL_K_MERGE1);
        WinoParams wp{};
        wp.in = merge_in;
        wp.upack = wd + h->off_m1_u[0];
        wp.upack_stride = T > 1 ? (long long)(h->off_m1_u[1] - h->off_m1_u[0]) : 0;
        wp.accum = 1;
        wp.bias = wd + h->off_m1_b;
        wp.out = h->merge.p;
        wp.H = H;
        wp.W = W;
        wp.items = F;
        wp.add_div = T;
        wp.act = 1;
        HIPCHK(launch_conv_wino_ws(wp, s));
    } else {   // convmerge1: 3x3 over the concat of T frames -> 48 + lrelu        (:73-74)
        ProfScope ps(h, s, PFNL_K_MERGE1);
        p.in = merge_in;
        p.wpack = wd + h->off_m1_w;
        p.bias = wd + h->off_m1_b;
        p.addend = nullptr;
        p.resid = nullptr;
        p.out = h->merge.p;
        p.out_cstride = 48;
        p.cout = 48;
        p.frames_per_item = T;
        p.nchunks = T * p.chunks_per_frame;
        p.add_div = 1;
        p.act = 1;
        HIPCHK(launch_conv_mfma(p, 3, B, s));
    }
    {   // model/pfnl.py:63,76-80
        ProfScope ps(h, s, PFNL_K_TAIL);
        HIPCHK(launch_tail(h->merge.p, in, wd + h->off_m2_w, wd + h->off_m2_b, out, B, T, Hfull, W, c.scale, mstride, s, strip, rflag));
    }
    prof_flush_run(h, s);
    h->chain_open = false;
    return 0;
}

}  // namespace

extern "C" {

const char* pfnl_last_error(void) { return g_err.c_str(); }
int pfnl_version(void) { return 3; }

int pfnl_device_count(int* count) {
    if (!count) return fail(PFNL_ERR_INVALID, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return 0;
}

int pfnl_create(const pfnl_config* cfg, pfnl_handle** out) {
    if (!cfg || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (int e = check_geometry(*cfg)) return e;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(PFNL_ERR_NODEVICE, "no HIP device visible: libpfnl_hip has no CPU fallback");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(PFNL_ERR_INVALID, "bad device_id");
    HIPCHK(hipSetDevice(cfg->device_id));
    pfnl_handle* h = new pfnl_handle();
    h->cfg = *cfg;
    if (const char* e = std::getenv("PFNL_SMALL")) h->small_mode = std::string(e) == "on" ? 1 : (std::string(e) == "off" ? 2 : 0);   // (A/B runs)
    if (const char* e = std::getenv("PFNL_SMALL_C10")) h->small_c10 = std::string(e) != "0" && std::string(e) != "off";   // (A/B runs)
    if (const char* e = std::getenv("PFNL_SF_CHAIN")) h->sf_chain = std::string(e) != "0" && std::string(e) != "off";   // (A/B runs)
    if (const char* e = std::getenv("PFNL_SF_C10")) h->sf_c10 = std::string(e) != "0" && std::string(e) != "off";   // (A/B runs)
    if (const char* e = std::getenv("PFNL_SF_MID")) h->sf_mid = std::string(e) != "0" && std::string(e) != "off";   // (A/B runs)
    if (const char* e = std::getenv("PFNL_SF_MID_CHAINS")) h->sf_mid_chains = std::atoi(e);   // (threshold sweeps)
    if (const char* e = std::getenv("PFNL_SPLIT_CHAINS")) h->split_chains = std::string(e) != "0" && std::string(e) != "off";   // (A/B runs)
    if (const char* e = std::getenv("PFNL_SF0")) h->sf0 = std::string(e) != "0" && std::string(e) != "off";   // (A/B runs)
    if (const char* e = std::getenv("PFNL_SPLIT16_SF")) h->sf_path = std::string(e) != "0" && std::string(e) != "off";   // (A/B runs)
    if (const char* e = std::getenv("PFNL_CONV3X3")) {
        const std::string v(e);
        h->conv_algo = v == "direct" ? 0 : (v == "winograd_tile" ? 1 : (v == "split16" ? 4 : (v == "winograd" ? 3 : 5)));
    }
    // A BLOCKING stream: it is implicitly ordered with the legacy null stream (= torch's default stream) in both
    // directions, so host-pointer calls and graph replays on it are ordered with the caller's default-stream work.
    if (hipStreamCreateWithFlags(&h->stream, hipStreamDefault) != hipSuccess) {
        delete h;
        return fail(PFNL_ERR_HIP, "hipStreamCreate failed");
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&h->rflag_host), 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&h->rflag_dev), h->rflag_host, 0) != hipSuccess) {
        if (h->rflag_host) hipHostFree(h->rflag_host);
        hipStreamDestroy(h->stream);
        delete h;
        return fail(PFNL_ERR_NOMEM, "allocation failed");
    }
    std::memset(h->rflag_host, 0, 64);
    if (const char* e = std::getenv("PFNL_STRICT_FP32")) h->strict = std::string(e) != "0" && std::string(e) != "off";
    const int T = cfg->num_frames, C = 12 * T;
    add_expected(h, "conv0", 5, 3, 64);
    for (int i = 0; i < cfg->num_block; ++i) {
        add_expected(h, "conv1_" + std::to_string(i), 3, 64, 64);
        add_expected(h, "conv10_" + std::to_string(i), 1, 64 * T, 64);
        add_expected(h, "conv2_" + std::to_string(i), 3, 128, 64);
    }
    add_expected(h, "convmerge1", 3, 64 * T, 48);
    add_expected(h, "convmerge2", 3, 12, cfg->scale == 4 ? 12 : 3);
    add_expected(h, "nlblock_0/g/g", 1, C, C);
    add_expected(h, "nlblock_0/w/w", 1, C, C);
    // utils.py:31-42: nltype 0 / 2 create theta and phi 1x1 convs; PFNL calls the block with nltype 1, where they hold
    // no variables - accepted when a caller supplies all four (embedded Gaussian), absent otherwise
    for (const char* n : {"theta/theta", "phi/phi"}) {
        h->optional[std::string("nlvsr/nlblock_0/") + n + "/kernel"] = {1, 1, C, C};
        h->optional[std::string("nlvsr/nlblock_0/") + n + "/bias"] = {C};
    }
    *out = h;
    return 0;
}

int pfnl_destroy(pfnl_handle* h) {
    if (!h) return 0;
    hipSetDevice(h->cfg.device_id);
    if (h->stream) {
        hipStreamSynchronize(h->stream);
        hipStreamDestroy(h->stream);
    }
    for (auto& e : h->evs) hipEventDestroy(e);
    if (h->gev) hipEventDestroy(h->gev);
    for (auto& g : h->graphs) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
    }
    h->pool.shutdown();
    for (auto& e : h->d2h_ev) hipEventDestroy(e);
    h->pin_in.release();
    h->pin_out.release();
    if (h->rflag_host) hipHostFree(h->rflag_host);
    for (DevBuf* b : {&h->p10, &h->Xs, &h->Q, &h->wdev16s, &h->wdev, &h->wdev16, &h->nl16, &h->X, &h->Xo, &h->nlp, &h->inp0, &h->inp0sf, &h->c10part, &h->inp1, &h->base, &h->pb, &h->merge,
                      &h->stage_in, &h->stage_out, &h->scratch})
        b->release();
    delete h;
    return 0;
}

int pfnl_set_weight(pfnl_handle* h, const char* tf_name, const float* host, const int64_t* shape,
                    int rank) {
    if (!h || !tf_name || !host || !shape) return fail(PFNL_ERR_INVALID, "NULL argument");
    std::string name(tf_name);
    // tolerate a ":0" suffix and a missing "nlvsr/" prefix (checkpoint readers differ)
    if (name.size() > 2 && name.compare(name.size() - 2, 2, ":0") == 0) name.resize(name.size() - 2);
    auto it = h->expected.find(name);
    if (it == h->expected.end()) it = h->expected.find("nlvsr/" + name);
    if (it == h->expected.end()) {
        it = h->optional.find(name);
        if (it == h->optional.end()) it = h->optional.find("nlvsr/" + name);
        if (it == h->optional.end()) return fail(PFNL_ERR_INVALID, "unknown tensor name: " + name);
    }
    if ((int)it->second.size() != rank) return fail(PFNL_ERR_INVALID, "rank mismatch for " + name);
    for (int i = 0; i < rank; ++i)
        if (shape[i] != it->second[i]) return fail(PFNL_ERR_INVALID, "shape mismatch for " + name);
    if (h->weights_external) {        // the placeholders of a received replica are not weights: a caller that sets one tensor sets them all
        h->host.clear();
        h->weights_external = false;
    }
    HostTensor t;
    t.shape = it->second;
    t.data.assign(host, host + numel(t.shape));
    h->host[it->first] = std::move(t);
    h->finalized = false;
    return 0;
}

int pfnl_set_option(pfnl_handle* h, const char* key, const char* value) {
    if (!h || !key || !value) return fail(PFNL_ERR_INVALID, "NULL argument");
    const std::string k(key), v(value);
    ++h->cfg_gen;
    if (k == "graph") {
        if (v == "auto") h->graph_mode = 1;
        else if (v == "on") h->graph_mode = 2;
        else if (v == "off") h->graph_mode = 0;
        else return fail(PFNL_ERR_INVALID, "graph must be auto, on or off");
        return 0;
    }
    if (k == "conv3x3") {
        if (v == "winograd" || v == "winograd_ws") h->conv_algo = 3;
        else if (v == "winograd_tile") h->conv_algo = 1;
        else if (v == "direct") h->conv_algo = 0;
        else if (v == "split16") h->conv_algo = 4;
        else if (v == "auto") h->conv_algo = 5;
        else return fail(PFNL_ERR_INVALID, "conv3x3 must be auto, split16, winograd, winograd_tile or direct");
        return 0;
    }
    if (k == "strict_fp32") {
        if (v == "on") h->strict = true;
        else if (v == "off") h->strict = false;
        else return fail(PFNL_ERR_INVALID, "strict_fp32 must be on or off");
        return 0;
    }
    if (k == "small") {
        if (v == "auto") h->small_mode = 0;
        else if (v == "on") h->small_mode = 1;
        else if (v == "off") h->small_mode = 2;
        else return fail(PFNL_ERR_INVALID, "small must be auto, on or off");
        return 0;
    }
    if (k == "split16_chain") {
        if (v == "on") h->sf_chain = true;
        else if (v == "off") h->sf_chain = false;
        else return fail(PFNL_ERR_INVALID, "split16_chain must be on or off");
        return 0;
    }
    if (k == "split16_c10") {
        if (v == "on") h->sf_c10 = true;
        else if (v == "off") h->sf_c10 = false;
        else return fail(PFNL_ERR_INVALID, "split16_c10 must be on or off");
        return 0;
    }
    if (k == "split16_mid") {
        if (v == "auto") h->sf_mid = true;
        else if (v == "off") h->sf_mid = false;
        else return fail(PFNL_ERR_INVALID, "split16_mid must be auto or off");
        return 0;
    }
    if (k == "split16_sf0") {
        if (v == "on") h->sf0 = true;
        else if (v == "off") h->sf0 = false;
        else return fail(PFNL_ERR_INVALID, "split16_sf0 must be on or off");
        return 0;
    }
    if (k == "split16_splitchains") {
        if (v == "auto") h->split_chains = true;
        else if (v == "off") h->split_chains = false;
        else return fail(PFNL_ERR_INVALID, "split16_splitchains must be auto or off");
        return 0;
    }
    if (k == "split16_sf") {
        if (v == "on") h->sf_path = true;
        else if (v == "off") h->sf_path = false;
        else return fail(PFNL_ERR_INVALID, "split16_sf must be on or off");
        return 0;
    }
    if (k == "conv2") {
        if (v == "grouped") h->conv2_grouped = true;
        else if (v == "split") h->conv2_grouped = false;
        else return fail(PFNL_ERR_INVALID, "conv2 must be grouped or split");
        return 0;
    }
    if (k == "split16_mfma") {
        if (v == "16") h->s16_m16 = true;
        else if (v == "32") h->s16_m16 = false;
        else return fail(PFNL_ERR_INVALID, "split16_mfma must be 16 or 32");
        return 0;
    }
    if (k == "bf16_mfma") {
        if (v == "16") h->bf16_m16 = true;
        else if (v == "32") h->bf16_m16 = false;
        else return fail(PFNL_ERR_INVALID, "bf16_mfma must be 16 or 32");
        return 0;
    }
    if (k == "bf16_conv10") {
        if (v == "fused") h->bf16_fuse10 = true;
        else if (v == "separate") h->bf16_fuse10 = false;
        else return fail(PFNL_ERR_INVALID, "bf16_conv10 must be fused or separate");
        return 0;
    }
    if (k == "precision") {
        if (v == "bf16") h->bf16 = true;
        else if (v == "fp32") h->bf16 = false;
        else return fail(PFNL_ERR_INVALID, "precision must be fp32 or bf16");
        return 0;
    }
    if (k == "merge1") {
        if (v == "auto") h->m1_algo = 0;
        else if (v == "split16") h->m1_algo = 1;
        else if (v == "winograd") h->m1_algo = 2;
        else return fail(PFNL_ERR_INVALID, "merge1 must be auto, split16 or winograd");
        ++h->cfg_gen;
        return 0;
    }
    if (k == "nl_type") {
        if (v == "auto") h->nl_type = -1;
        else if (v == "0" || v == "embedded_gaussian") h->nl_type = 0;
        else if (v == "1" || v == "gaussian") h->nl_type = 1;
        else if (v == "2" || v == "dot_product") h->nl_type = 2;
        else return fail(PFNL_ERR_INVALID, "nl_type: auto | 0 | 1 | 2 (nltype 3, 'concat', builds no graph in the reference either: utils.py:23)");
        return 0;
    }
    if (k == "nl_sub_sample") {
        const int n = atoi(v.c_str());
        if (n < 1 || n > 64) return fail(PFNL_ERR_INVALID, "nl_sub_sample: an integer >= 1");
        h->nl_sub = n;
        return 0;
    }
    if (k == "small_c10") {
        if (v == "on") h->small_c10 = true;
        else if (v == "off") h->small_c10 = false;
        else return fail(PFNL_ERR_INVALID, "small_c10 must be on or off");
        return 0;
    }
    if (k == "bf16_nonlocal") {   // (the split-bf16 kernel of round 1 left the library in round 4: tools/experiments/nonlocal_bf16.hip)
        if (v != "f16") return fail(PFNL_ERR_INVALID, "bf16_nonlocal must be f16");
        return 0;
    }
    if (k == "nonlocal") {
        if (v == "f32") h->nl_algo = 0;
        else if (v == "split16") h->nl_algo = 1;
        else if (v == "auto") h->nl_algo = 2;
        else return fail(PFNL_ERR_INVALID, "nonlocal must be auto, f32 or split16");
        return 0;
    }
    if (k == "conv1x1") {
        if (v == "stream") h->conv1x1_algo = 1;
        else if (v == "tiled") h->conv1x1_algo = 0;
        else if (v == "split16") h->conv1x1_algo = 2;
        else return fail(PFNL_ERR_INVALID, "conv1x1 must be split16, stream or tiled");
        return 0;
    }
    return fail(PFNL_ERR_INVALID, "unknown option " + k);
}

// the CURRENT value of an option as pfnl_set_option would take it back - whatever set it (pfnl_set_option, an environment variable read by
// pfnl_create, the default): a caller that changes options temporarily restores from here, not from a mirror of its own calls
int pfnl_get_option(pfnl_handle* h, const char* key, char* buf, size_t buflen) {
    if (!h || !key || !buf || buflen < 1) return fail(PFNL_ERR_INVALID, "NULL argument");
    const std::string k(key);
    std::string v;
    auto onoff = [](bool b) { return std::string(b ? "on" : "off"); };
    if (k == "graph") v = h->graph_mode == 1 ? "auto" : (h->graph_mode == 2 ? "on" : "off");
    else if (k == "conv3x3") v = h->conv_algo == 3 ? "winograd" : h->conv_algo == 1 ? "winograd_tile" : h->conv_algo == 0 ? "direct" : h->conv_algo == 4 ? "split16" : "auto";
    else if (k == "strict_fp32") v = onoff(h->strict);
    else if (k == "small") v = h->small_mode == 1 ? "on" : (h->small_mode == 2 ? "off" : "auto");
    else if (k == "split16_chain") v = onoff(h->sf_chain);
    else if (k == "split16_c10") v = onoff(h->sf_c10);
    else if (k == "split16_mid") v = h->sf_mid ? "auto" : "off";
    else if (k == "split16_sf0") v = onoff(h->sf0);
    else if (k == "split16_sf") v = onoff(h->sf_path);
    else if (k == "split16_splitchains") v = h->split_chains ? "auto" : "off";
    else if (k == "conv2") v = h->conv2_grouped ? "grouped" : "split";
    else if (k == "bf16_conv10") v = h->bf16_fuse10 ? "fused" : "separate";
    else if (k == "bf16_mfma") v = h->bf16_m16 ? "16" : "32";
    else if (k == "split16_mfma") v = h->s16_m16 ? "16" : "32";
    else if (k == "precision") v = h->bf16 ? "bf16" : "fp32";
    else if (k == "merge1") v = h->m1_algo == 1 ? "split16" : (h->m1_algo == 2 ? "winograd" : "auto");
    else if (k == "nl_type") v = h->nl_type < 0 ? "auto" : std::to_string(h->nl_type);
    else if (k == "nl_sub_sample") v = std::to_string(h->nl_sub);
    else if (k == "small_c10") v = onoff(h->small_c10);
    else if (k == "bf16_nonlocal") v = "f16";
    else if (k == "nonlocal") v = h->nl_algo == 0 ? "f32" : (h->nl_algo == 1 ? "split16" : "auto");
    else if (k == "conv1x1") v = h->conv1x1_algo == 1 ? "stream" : (h->conv1x1_algo == 0 ? "tiled" : "split16");
    else return fail(PFNL_ERR_INVALID, "unknown option " + k);
    if (v.size() + 1 > buflen) return fail(PFNL_ERR_INVALID, "buffer too small");
    std::memcpy(buf, v.c_str(), v.size() + 1);
    return 0;
}

int pfnl_missing_weights(pfnl_handle* h, int* count) {
    if (!h || !count) return fail(PFNL_ERR_INVALID, "NULL argument");
    int n = 0;
    for (auto& kv : h->expected)
        if (!h->host.count(kv.first)) ++n;
    *count = n;
    return 0;
}

int pfnl_finalize_weights(pfnl_handle* h) {
    if (!h) return fail(PFNL_ERR_INVALID, "NULL handle");
    for (auto& kv : h->expected)
        if (!h->host.count(kv.first)) return fail(PFNL_ERR_STATE, "missing tensor " + kv.first);
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const int T = h->cfg.num_frames, nb = h->cfg.num_block, C = 12 * T, CP = pfnl::nl_padded_ch(C);
    std::vector<float> blob;
    auto W = [&](const std::string& l) -> const std::vector<float>& { return h->host["nlvsr/" + l + "/kernel"].data; };
    auto Bv = [&](const std::string& l) -> const std::vector<float>& { return h->host["nlvsr/" + l + "/bias"].data; };
    auto reserve = [&](size_t n) {
        size_t off = (blob.size() + 63) / 64 * 64;   // 256-byte aligned segments
        blob.resize(off + n, 0.f);
        return off;
    };
    auto put_bias = [&](const std::vector<float>& b) {
        size_t off = reserve(64);
        std::memcpy(&blob[off], b.data(), b.size() * sizeof(float));
        return off;
    };
    auto put_wino = [&](const std::vector<float>& k, int cin_total, int cin_begin) {
        size_t off = reserve(pfnl::wino_pack_floats());
        pfnl::wino_pack_weights(k.data(), cin_total, cin_begin, &blob[off]);
        return off;
    };
    auto put_pack = [&](const std::vector<float>& k, int ks, int cin_total, int cin_begin, int cin, int cout) {
        size_t off = reserve(pfnl::conv_pack_floats(ks, cin));
        pfnl::conv_pack_weights(k.data(), ks, cin_total, cin_begin, cin, cout, &blob[off]);
        return off;
    };

    h->off_zero = reserve(64);
    h->off_conv0_w = reserve(75 * 64);
    std::memcpy(&blob[h->off_conv0_w], W("conv0").data(), 75 * 64 * sizeof(float));
    h->off_conv0_b = put_bias(Bv("conv0"));
    h->off_c1_w.assign(nb, 0);
    h->off_c1_b.assign(nb, 0);
    h->off_c10_w.assign(nb, 0);
    h->off_c10_b.assign(nb, 0);
    h->off_c10_s.assign(nb, 0);
    h->off_c2a_w.assign(nb, 0);
    h->off_c2b_w.assign(nb, 0);
    h->off_c2_b.assign(nb, 0);
    h->off_c1_u.assign(nb, 0);
    h->off_c2a_u.assign(nb, 0);
    h->off_c2b_u.assign(nb, 0);
    for (int i = 0; i < nb; ++i) {
        const std::string s = std::to_string(i);
        h->off_c1_w[i] = put_pack(W("conv1_" + s), 3, 64, 0, 64, 64);
        h->off_c1_b[i] = put_bias(Bv("conv1_" + s));
        h->off_c10_w[i] = put_pack(W("conv10_" + s), 1, 64 * T, 0, 64 * T, 64);
        h->off_c10_b[i] = put_bias(Bv("conv10_" + s));
        h->off_c10_s[i] = reserve(pfnl::conv1x1_pack_floats(T));
        pfnl::conv1x1_pack_weights(W("conv10_" + s).data(), T, &blob[h->off_c10_s[i]]);
        // conv2_i input = concat([base, inp1_t]) (model/pfnl.py:69): rows 0..63 see `base`.
        h->off_c2a_w[i] = put_pack(W("conv2_" + s), 3, 128, 0, 64, 64);
        h->off_c2b_w[i] = put_pack(W("conv2_" + s), 3, 128, 64, 64, 64);
        h->off_c2_b[i] = put_bias(Bv("conv2_" + s));
        h->off_c1_u[i] = put_wino(W("conv1_" + s), 64, 0);
        h->off_c2a_u[i] = put_wino(W("conv2_" + s), 128, 0);
        h->off_c2b_u[i] = put_wino(W("conv2_" + s), 128, 64);
    }
    h->off_m1_w = put_pack(W("convmerge1"), 3, 64 * T, 0, 64 * T, 48);
    h->off_m1_b = put_bias(Bv("convmerge1"));
    h->off_m1_u.assign(T, 0);
    for (int f = 0; f < T; ++f) {
        h->off_m1_u[f] = reserve(pfnl::wino_pack_floats());
        pfnl::wino_pack_weights(W("convmerge1").data(), 64 * T, 64 * f, &blob[h->off_m1_u[f]], 48);
    }
    {
        const auto& k = W("convmerge2");
        h->off_m2_w = reserve(k.size());
        std::memcpy(&blob[h->off_m2_w], k.data(), k.size() * sizeof(float));
        h->off_m2_b = put_bias(Bv("convmerge2"));
    }
    {   // fold the two 1x1 projections of the non-local block: W' = Wg Ww, b' = bg Ww + bw
        const auto& wg = W("nlblock_0/g/g");   // [C][C] (ci, cm)
        const auto& ww = W("nlblock_0/w/w");   // [C][C] (cm, co)
        const auto& bg = Bv("nlblock_0/g/g");
        const auto& bw = Bv("nlblock_0/w/w");
        h->off_nl_w = reserve((size_t)CP * CP);
        h->off_nl_b = reserve(CP);
        for (int ci = 0; ci < C; ++ci)
            for (int co = 0; co < C; ++co) {
                double acc = 0.0;
                for (int cm = 0; cm < C; ++cm) acc += (double)wg[(size_t)ci * C + cm] * (double)ww[(size_t)cm * C + co];
                blob[h->off_nl_w + (size_t)ci * CP + co] = (float)acc;
            }
        for (int co = 0; co < C; ++co) {
            double acc = bw[co];
            for (int cm = 0; cm < C; ++cm) acc += (double)bg[cm] * (double)ww[(size_t)cm * C + co];
            blob[h->off_nl_b + co] = (float)acc;
        }
    }
    {   // optional embedded-Gaussian projections: M = Wt Wp^T, c = bt Wp^T (see nl_qproj_kernel)
        int have = 0;
        for (auto& kv : h->optional) have += (int)h->host.count(kv.first);
        if (have != 0 && have != (int)h->optional.size())
            return fail(PFNL_ERR_STATE, "nlblock_0 theta/phi: all four tensors (two kernels, two biases) or none");
        h->nl_theta = have != 0;
        if (h->nl_theta) {
            const auto& wt = W("nlblock_0/theta/theta");   // [C][C] (ci, cm)
            const auto& wp = W("nlblock_0/phi/phi");
            const auto& bt = Bv("nlblock_0/theta/theta");
            h->off_nl_m = reserve((size_t)CP * CP);
            h->off_nl_c = reserve(CP);
            for (int ci = 0; ci < C; ++ci)
                for (int cj = 0; cj < C; ++cj) {
                    double acc = 0.0;
                    for (int cm = 0; cm < C; ++cm) acc += (double)wt[(size_t)ci * C + cm] * (double)wp[(size_t)cj * C + cm];
                    blob[h->off_nl_m + (size_t)ci * CP + cj] = (float)acc;
                }
            for (int cj = 0; cj < C; ++cj) {
                double acc = 0.0;
                for (int cm = 0; cm < C; ++cm) acc += (double)bt[cm] * (double)wp[(size_t)cj * C + cm];
                blob[h->off_nl_c + cj] = (float)acc;
            }
            // column C (a pad column: C < CP): theta_i . b_phi = X_i (Wt b_phi) + bt . b_phi - the per-query constant of the logits
            // that cancels in the softmax of nltype 0 and does not under the relu of nltype 2 (nl_attn_kernel<., DOT>)
            const auto& bph = Bv("nlblock_0/phi/phi");
            double d0 = 0.0;
            for (int cm = 0; cm < C; ++cm) d0 += (double)bt[cm] * (double)bph[cm];
            blob[h->off_nl_c + C] = (float)d0;
            for (int ci = 0; ci < C; ++ci) {
                double acc = 0.0;
                for (int cm = 0; cm < C; ++cm) acc += (double)wt[(size_t)ci * C + cm] * (double)bph[cm];
                blob[h->off_nl_m + (size_t)ci * CP + C] = (float)acc;
            }
        }
    }
    {   // bf16 packs of the trunk (precision=bf16)
        std::vector<uint16_t> b16;
        auto reserve16 = [&](size_t n) {
            size_t off = (b16.size() + 127) / 128 * 128;
            b16.resize(off + n, 0);
            return off;
        };
        h->off16_c1.assign(nb, 0);
        h->off16_c10.assign(nb, 0);
        h->off16_c2a.assign(nb, 0);
        h->off16_c2b.assign(nb, 0);
        h->off16_c1_m16.assign(nb, 0);
        h->off16_c2b_m16.assign(nb, 0);
        for (int i = 0; i < nb; ++i) {
            const std::string s = std::to_string(i);
            h->off16_c1[i] = reserve16(pfnl::conv3x3_bf16_pack_halfs());
            pfnl::conv3x3_bf16_pack_weights(W("conv1_" + s).data(), 64, 0, &b16[h->off16_c1[i]]);
            h->off16_c10[i] = reserve16(pfnl::conv1x1_bf16_pack_halfs(T));
            pfnl::conv1x1_bf16_pack_weights(W("conv10_" + s).data(), T, &b16[h->off16_c10[i]]);
            h->off16_c2a[i] = reserve16(pfnl::conv3x3_bf16_pack_halfs());
            pfnl::conv3x3_bf16_pack_weights(W("conv2_" + s).data(), 128, 0, &b16[h->off16_c2a[i]]);
            h->off16_c2b[i] = reserve16(pfnl::conv3x3_bf16_pack_halfs());
            pfnl::conv3x3_bf16_pack_weights(W("conv2_" + s).data(), 128, 64, &b16[h->off16_c2b[i]]);
            h->off16_c1_m16[i] = reserve16(pfnl::conv3x3_bf16_pack_halfs());
            pfnl::conv3x3_bf16_pack_weights16(W("conv1_" + s).data(), 64, 0, &b16[h->off16_c1_m16[i]]);
            h->off16_c2b_m16[i] = reserve16(pfnl::conv3x3_bf16_pack_halfs());
            pfnl::conv3x3_bf16_pack_weights16(W("conv2_" + s).data(), 128, 64, &b16[h->off16_c2b_m16[i]]);
        }
        h->off16_m1 = reserve16((size_t)T * pfnl::conv3x3_bf16_pack_halfs());
        for (int f = 0; f < T; ++f)
            pfnl::conv3x3_bf16_pack_weights(W("convmerge1").data(), 64 * T, 64 * f, &b16[h->off16_m1 + (size_t)f * pfnl::conv3x3_bf16_pack_halfs()], 48);
        b16.resize((b16.size() + 1) / 2 * 2 + 2, 0);
        h->wdev16_used = b16.size() / 2;
        if (h->wdev16.ensure(b16.size() / 2)) return fail(PFNL_ERR_NOMEM, "weight allocation failed");
        HIPCHK(hipMemcpy(h->wdev16.p, b16.data(), b16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    {   // split-f16 packs of the 3x3 64->64 kernels (conv3x3=split16)
        std::vector<uint16_t> b16;
        const size_t n3 = pfnl::conv3x3_split16_pack_halfs(), n1 = pfnl::conv1x1_split16_pack_halfs(T);
        h->off16s_c1.assign(nb, 0);
        h->off16s_c2a.assign(nb, 0);
        h->off16s_c2b.assign(nb, 0);
        h->off16s_c10.assign(nb, 0);
        h->off16s_c2a_sf.assign(nb, 0);
        h->off16s_c2b_sf.assign(nb, 0);
        h->off16s_c2a_m16.assign(nb, 0);
        h->off16s_c2b_m16.assign(nb, 0);
        const size_t blk = 7 * n3 + n1 + pfnl::conv1x1_c10_pack_halfs(T);   // (5 packs of the 32x32x16 kernels + 2 of the 16x16x32 chain kernel)
        h->off16s_c10f.assign(nb, 0);
        h->off16s_m1 = (size_t)nb * blk;
        const size_t m3 = pfnl::conv_small_pack_halfs(3, 1), m10 = pfnl::conv_small_pack_halfs(1, T);
        const size_t small_base = (size_t)nb * blk + (size_t)T * n3;
        h->off16m_c1.assign(nb, 0);
        h->off16m_c10.assign(nb, 0);
        h->off16m_c2.assign(nb, 0);
        h->off16m_m1 = small_base + (size_t)nb * (3 * m3 + m10);
        b16.resize(h->off16m_m1 + (size_t)T * m3 + 2, 0);
        pfnl::conv_small_pack_weights(W("convmerge1").data(), 3, T, 48, &b16[h->off16m_m1]);
        for (int f = 0; f < T; ++f)
            pfnl::conv3x3_split16_pack_weights(W("convmerge1").data(), 64 * T, 64 * f, &b16[h->off16s_m1 + (size_t)f * n3], 48);
        for (int i = 0; i < nb; ++i) {
            const std::string s = std::to_string(i);
            h->off16s_c1[i] = (size_t)i * blk;
            h->off16s_c2a[i] = h->off16s_c1[i] + n3;
            h->off16s_c2b[i] = h->off16s_c1[i] + 2 * n3;
            h->off16s_c10[i] = h->off16s_c1[i] + 3 * n3;
            h->off16s_c2a_sf[i] = h->off16s_c10[i] + n1;
            h->off16s_c2b_sf[i] = h->off16s_c2a_sf[i] + n3;
            h->off16s_c10f[i] = h->off16s_c2b_sf[i] + n3;
            h->off16s_c2a_m16[i] = h->off16s_c10f[i] + pfnl::conv1x1_c10_pack_halfs(T);
            h->off16s_c2b_m16[i] = h->off16s_c2a_m16[i] + n3;
            pfnl::conv3x3_split16_pack_weights16(W("conv2_" + s).data(), 128, 0, &b16[h->off16s_c2a_m16[i]]);
            pfnl::conv3x3_split16_pack_weights16(W("conv2_" + s).data(), 128, 64, &b16[h->off16s_c2b_m16[i]]);
            pfnl::conv1x1_c10_pack_weights(W("conv10_" + s).data(), T, &b16[h->off16s_c10f[i]]);
            pfnl::conv3x3_split16_pack_weights(W("conv2_" + s).data(), 128, 0, &b16[h->off16s_c2a_sf[i]], 64, true);
            pfnl::conv3x3_split16_pack_weights(W("conv2_" + s).data(), 128, 64, &b16[h->off16s_c2b_sf[i]], 64, true);
            h->off16m_c1[i] = small_base + (size_t)i * (3 * m3 + m10);
            h->off16m_c2[i] = h->off16m_c1[i] + m3;
            h->off16m_c10[i] = h->off16m_c1[i] + 3 * m3;
            pfnl::conv_small_pack_weights(W("conv1_" + s).data(), 3, 1, 64, &b16[h->off16m_c1[i]]);
            pfnl::conv_small_pack_weights(W("conv2_" + s).data(), 3, 2, 64, &b16[h->off16m_c2[i]]);
            pfnl::conv_small_pack_weights(W("conv10_" + s).data(), 1, T, 64, &b16[h->off16m_c10[i]]);
            pfnl::conv1x1_split16_pack_weights(W("conv10_" + s).data(), T, &b16[h->off16s_c10[i]]);
            pfnl::conv3x3_split16_pack_weights(W("conv1_" + s).data(), 64, 0, &b16[h->off16s_c1[i]]);
            pfnl::conv3x3_split16_pack_weights(W("conv2_" + s).data(), 128, 0, &b16[h->off16s_c2a[i]]);
            pfnl::conv3x3_split16_pack_weights(W("conv2_" + s).data(), 128, 64, &b16[h->off16s_c2b[i]]);
        }
        h->wdev16s_used = b16.size() / 2;
        if (h->wdev16s.ensure(b16.size() / 2)) return fail(PFNL_ERR_NOMEM, "weight allocation failed");
        HIPCHK(hipMemcpy(h->wdev16s.p, b16.data(), b16.size() / 2 * 2 * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    {   // the f16-pipe kernels split the weights into binary16 pairs: a weight beyond that range sends the handle to the f32-MFMA kernels
        float wmax = 0.f;
        for (const auto& kv : h->host)
            for (float v : kv.second.data) wmax = std::fmax(wmax, std::fabs(v));
        h->weights_f16_ok = wmax < 65504.0f;
    }
    h->wdev_used = blob.size();
    if (h->wdev.ensure(blob.size())) return fail(PFNL_ERR_NOMEM, "weight allocation failed");
    HIPCHK(hipMemcpy(h->wdev.p, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    h->finalized = true;
    ++h->cfg_gen;
    return 0;
}

int pfnl_workspace_bytes(pfnl_handle* h, int B, int H, int W, size_t* bytes) {
    if (!h || !bytes) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return fail(PFNL_ERR_INVALID, "bad shape");
    // the same expressions forward_device / pfnl_forward allocate with (every buffer a call with these arguments can
    // touch under the current options, the host-pointer staging pair included)
    const size_t T = h->cfg.num_frames, P = (size_t)H * W, N = P / 4, sc = h->cfg.scale;
    const int C = 12 * (int)T;
    const size_t CP = pfnl::nl_padded_ch(C);
    size_t f = 2 * B * N * CP                                   // X, Xo
             + pfnl::nl_partial_floats(B, (int)N, C)            // nlp
             + 2 * B * T * P * 64                               // inp0, inp1
             + 3 * B * P * 64                                   // base, pb, merge (64 floats per pixel)
             + (size_t)B * T * P * 3 + (size_t)B * P * sc * sc * 3;   // stage_in, stage_out
    if (h->bf16 || h->nl_algo != 0) f += (pfnl::nl_f16_scratch_halfs(B, (int)N) + 1) / 2;   // split K / V^T operands (bf16 or f16)
    if (h->nl_theta) f += (size_t)B * N * CP;                   // projected queries (nltype 0 / 2)
    if (h->nl_sub > 1) f += (size_t)B * ((H / 2) / h->nl_sub) * ((W / 2) / h->nl_sub) * CP;   // pooled keys
    {   // buffers that depend on the launch plan: the small-shape trunk's conv10_i partials; the split-format copy of inp0
        const TrunkPlan pl = trunk_plan(h, B, H, W);
        if (pl.small_c10) f += B * T * P * 64;
        if (pl.sf0) f += B * T * P * 64;
        if (pl.split_s) f += (size_t)(pl.chains - pl.n_full) * pl.split_s * 8 * 32 * 64;
    }
    *bytes = f * sizeof(float);
    return 0;
}

int pfnl_plan(pfnl_handle* h, int B, int H, int W, char* buf, size_t buflen) {
    if (!h || !buf || buflen < 1) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return fail(PFNL_ERR_INVALID, "bad shape");
    const TrunkPlan pl = trunk_plan(h, B, H, W);
    const int N = (H / 2) * (W / 2), nlf = nl_family(h, N);
    static const char* const nln[] = {"general_f32", "split16", "f16", "f32"};
    const int nl_fused = (nlf == 1 || nlf == 2) && pfnl::nl_f16_fits_one_launch(B, N);
    static const char* const a3[] = {"direct", "winograd_tile", "?", "winograd", "split16"};
    static const char* const a1[] = {"tiled", "stream", "split16"};
    char tmp[384];
    if (pl.bf16)
        std::snprintf(tmp, sizeof tmp, "%s launches_per_block=%d c1x1=%d precision=bf16 tiles=%d chains=%d whole_chains=%d split_parts=%d part_frames=%d nl=%s nl_pack_fused=%d mfma=%d",
                      pl.name, pl.launches_per_block, pl.c1x1_launches, pl.tiles8x32, pl.chains, pl.split_s ? pl.n_full : pl.chains, pl.split_s, pl.split_q, nln[nlf], nl_fused, pl.mfma);
    else
        std::snprintf(tmp, sizeof tmp, "%s launches_per_block=%d c1x1=%d precision=fp32 conv3x3=%s conv1x1=%s c10_fused=%d chain=%d sf0=%d strict=%d tiles=%d chains=%d "
                      "whole_chains=%d split_parts=%d part_frames=%d nl=%s nl_pack_fused=%d mfma=%d",
                      pl.name, pl.launches_per_block, pl.c1x1_launches, pl.small ? "small" : a3[pl.algo < 0 || pl.algo > 4 ? 2 : pl.algo],
                      a1[pl.conv1x1_algo < 0 || pl.conv1x1_algo > 2 ? 0 : pl.conv1x1_algo], pl.c10_fused ? 1 : 0, pl.chain ? 1 : 0, pl.sf0 ? 1 : 0,
                      pl.strict ? 1 : 0, pl.tiles8x32, pl.chains, pl.split_s ? pl.n_full : pl.chains, pl.split_s, pl.split_q, nln[nlf], nl_fused, pl.mfma);
    if (std::strlen(tmp) + 1 > buflen) return fail(PFNL_ERR_INVALID, "buffer too small");
    std::strcpy(buf, tmp);
    return 0;
}

// reads and clears a sticky range flag (pinned, device-mapped host memory).  One atomic exchange: pfnl_range_flag is documented for use
// while later forwards are in flight, and with a plain read followed by a plain write of 0 a device store of 1 landing between the two
// would be lost.  A returned 1 may therefore also cover a forward that is still running.
static int range_flag_take(pfnl_handle* h, int slot, int* flagged) {
    unsigned* const f = h->rflag_host + slot;
    *flagged = __atomic_exchange_n(f, 0u, __ATOMIC_ACQ_REL) != 0;
    return 0;
}

int pfnl_range_reruns(pfnl_handle* h, long long* count) {
    if (!h || !count) return fail(PFNL_ERR_INVALID, "NULL argument");
    *count = h->range_reruns;
    return 0;
}

// ---- host-pointer forwards: what moves between the caller's buffers and HBM (the reference times exactly this inside sess.run,
// model/pfnl.py:249-253: feed_dict copy in, fetch copy out).  Pageable buffers go through the handle's pinned strips chunk by
// chunk: worker threads copy host <-> pinned while the copy engine moves the pinned side over PCIe, so neither the runtime's own
// single-threaded pageable staging (measured 6 GB/s: 2.95 ms of a 7.55 ms call at configs[1]) nor a whole-buffer memcpy is on
// the critical path; buffers that are already page-locked (hipHostMalloc / hipHostRegister / torch pin_memory) are DMA targets as
// they are.
namespace {
constexpr size_t STAGE_POOL_MIN = 512u << 10;     // below: one memcpy on the calling thread
constexpr int STAGE_MAX_CHUNKS = 32;

bool host_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();                   // (an ordinary malloc'ed pointer is "invalid value" to the runtime: not an error here)
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

void ensure_pool(pfnl_handle* h) {
    if (!h->pool.th.empty()) return;
    int k = 4;
    if (const char* e = std::getenv("PFNL_HOST_THREADS")) k = std::atoi(e);
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 1 && k > hw - 1) k = hw - 1;
    k = k < 1 ? 1 : (k > 16 ? 16 : k);
    h->pool.start(k, h->cfg.device_id);
}

size_t stage_chunk_bytes(size_t bytes) {
    size_t c = (bytes + STAGE_MAX_CHUNKS - 1) / STAGE_MAX_CHUNKS;
    if (c < (1u << 20)) c = 1u << 20;
    return (c + 4095) & ~(size_t)4095;
}

// caller's host buffer -> device, asynchronous on `s` (the pinned strip is free again once `s` has passed the copies)
int stage_h2d(pfnl_handle* h, float* dst, const void* src, size_t bytes, hipStream_t s) {
    if (host_is_pinned(src)) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        return 0;
    }
    if (h->pin_in.ensure(bytes)) return fail(PFNL_ERR_NOMEM, "pinned staging allocation failed");
    unsigned char* const pin = h->pin_in.p;
    if (bytes < STAGE_POOL_MIN) {
        std::memcpy(pin, src, bytes);
        HIPCHK(hipMemcpyAsync(dst, pin, bytes, hipMemcpyHostToDevice, s));
        return 0;
    }
    ensure_pool(h);
    const size_t csz = stage_chunk_bytes(bytes);
    const int n = (int)((bytes + csz - 1) / csz);
    const unsigned char* const sp = static_cast<const unsigned char*>(src);
    h->pool.run(n, [=](int i) {
        const size_t off = (size_t)i * csz;
        std::memcpy(pin + off, sp + off, std::min(csz, bytes - off));
    });
    hipError_t err = hipSuccess;
    for (int i = 0; i < n; ++i) {
        h->pool.wait_chunk(i);
        const size_t off = (size_t)i * csz;
        if (err == hipSuccess)
            err = hipMemcpyAsync(reinterpret_cast<unsigned char*>(dst) + off, pin + off, std::min(csz, bytes - off), hipMemcpyHostToDevice, s);
    }
    h->pool.wait();
    if (err != hipSuccess) return fail(PFNL_ERR_HIP, std::string("H2D staging: ") + hipGetErrorString(err));
    return 0;
}

// device -> caller's host buffer; returns when `dst` is filled (`s` is idle then)
int stage_d2h(pfnl_handle* h, void* dst, const float* src, size_t bytes, hipStream_t s) {
    if (host_is_pinned(dst)) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }
    if (h->pin_out.ensure(bytes)) return fail(PFNL_ERR_NOMEM, "pinned staging allocation failed");
    unsigned char* const pin = h->pin_out.p;
    if (bytes < STAGE_POOL_MIN) {
        HIPCHK(hipMemcpyAsync(pin, src, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::memcpy(dst, pin, bytes);
        return 0;
    }
    ensure_pool(h);
    const size_t csz = stage_chunk_bytes(bytes);
    const int n = (int)((bytes + csz - 1) / csz);
    while ((int)h->d2h_ev.size() < n) {
        hipEvent_t e = nullptr;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->d2h_ev.push_back(e);
    }
    for (int i = 0; i < n; ++i) {
        const size_t off = (size_t)i * csz;
        HIPCHK(hipMemcpyAsync(pin + off, reinterpret_cast<const unsigned char*>(src) + off, std::min(csz, bytes - off), hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(h->d2h_ev[i], s));
    }
    HIPCHK(hipEventSynchronize(h->d2h_ev[0]));     // (the forward itself: waited for here, not by every worker)
    unsigned char* const dp = static_cast<unsigned char*>(dst);
    hipEvent_t* const ev = h->d2h_ev.data();
    h->pool.run(n, [=](int i) {
        const size_t off = (size_t)i * csz;
        (void)hipEventSynchronize(ev[i]);
        std::memcpy(dp + off, pin + off, std::min(csz, bytes - off));
    });
    h->pool.wait();
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

const char* const RANGE_MSG_FP32 =
    "a device-pointer forward since the last check produced non-finite values: an operand left the range of the f16-pipe kernels "
    "(|x| < 65504; the non-local block: inputs on a [0,1] scale) or the input was not finite. Re-run with "
    "pfnl_set_option(h, \"strict_fp32\", \"on\") (f32-MFMA kernels: the reference's range), or use host pointers";
const char* const RANGE_MSG_BF16 =
    "a precision=bf16 forward produced non-finite values: the input was not finite, or it left the range of the binary16 operands of "
    "this precision's non-local block and conv0 (inputs on a [0,1] scale, |x| < ~350). precision=fp32 covers the whole fp32 range";
}  // namespace

int pfnl_forward(pfnl_handle* h, const void* in, int in_is_device, void* out, int out_is_device, int B,
                 int H, int W, void* stream) {
    if (!h || !in || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (!h->finalized) return fail(PFNL_ERR_STATE, "pfnl_finalize_weights has not been called");
    if (B <= 0 || H <= 0 || W <= 0) return fail(PFNL_ERR_INVALID, "B, H, W must be positive");
    if ((H & 1) || (W & 1))
        return fail(PFNL_ERR_INVALID, "H and W must be even (space_to_depth(2), reference model/pfnl.py:57)");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const int T = h->cfg.num_frames, sc = h->cfg.scale;
    const size_t n_in = (size_t)B * T * H * W * 3, n_out = (size_t)B * H * W * sc * sc * 3;
    const float* din = (const float*)in;
    float* dout = (float*)out;
    const bool sync_call = !in_is_device || !out_is_device;     // a host pointer: the call returns with `out` filled
    const int slot = sync_call ? 0 : 1;                         // whose range flag the tail kernel of this call writes
    const bool can_rerun = !h->bf16 && !(h->strict || !h->weights_f16_ok);
    const bool want_graph = !h->prof && (h->graph_mode == 2 || (h->graph_mode == 1 && (size_t)B * T * H * W <= 65536));
    // stream == NULL: device-pointer calls are launched on the LEGACY NULL STREAM itself (what a caller that passes
    // torch.cuda.current_stream().cuda_stream == 0 means: same-stream ordering with everything it has enqueued and will
    // enqueue); host-pointer calls and graph replays (a capture cannot run on the null stream) use the handle's stream.
    hipStream_t s = stream ? (hipStream_t)stream : ((in_is_device && out_is_device && !want_graph) ? (hipStream_t) nullptr : h->stream);
    if (want_graph) {
        if (h->stage_in.ensure(n_in) || h->stage_out.ensure(n_out)) return fail(PFNL_ERR_NOMEM, "staging allocation failed");
        pfnl_handle::GraphEntry* ge = nullptr;
        for (auto& g : h->graphs)
            if (g.B == B && g.H == H && g.W == W && g.slot == slot) ge = &g;
        if (!ge) {
            h->graphs.push_back({B, H, W, 0, nullptr, nullptr, 0, 0, h->cfg_gen, slot});
            ge = &h->graphs.back();
        }
        if (ge->exec && (ge->alloc_gen != g_alloc_gen || ge->cfg_gen != h->cfg_gen)) {   // buffers moved / weights or options changed
            // a replay of this graph may still be running on the stream (an asynchronous device-pointer call followed by pfnl_set_option):
            // whether the runtime defers the destruction of an executing graph depends on its version - wait for it
            HIPCHK(hipStreamSynchronize(s));
            hipGraphExecDestroy(ge->exec);
            hipGraphDestroy(ge->graph);
            ge->exec = nullptr;
            ge->graph = nullptr;
            ge->seen = 0;
        }
        if (!ge->exec && ge->seen_cfg_gen != h->cfg_gen) {   // options / weights changed since the eager run: the path to be
            ge->seen = 0;                                     // captured has not allocated its workspaces / set its attributes yet
            ge->seen_cfg_gen = h->cfg_gen;
        }
        if (!ge->exec && ge->seen >= 1) {
            // second call with this shape (the first one ran eagerly: workspaces allocated, kernel attributes set)
            HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            const int fe = forward_device(h, h->stage_in.p, h->stage_out.p, B, H, W, s, nullptr, 0, -1, slot);
            hipGraph_t g = nullptr;
            const hipError_t ce = hipStreamEndCapture(s, &g);
            if (fe) return fe;
            if (ce != hipSuccess || !g) return fail(PFNL_ERR_HIP, std::string("graph capture: ") + hipGetErrorString(ce));
            hipGraphExec_t ex = nullptr;
            const hipError_t ie = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
            if (ie != hipSuccess) {
                hipGraphDestroy(g);
                return fail(PFNL_ERR_HIP, std::string("graph instantiate: ") + hipGetErrorString(ie));
            }
            ge->graph = g;
            ge->exec = ex;
            ge->alloc_gen = g_alloc_gen;
            ge->cfg_gen = h->cfg_gen;
        }
        if (ge->exec) {
            // No stream given: the eager path relies on the legacy default stream's implicit ordering with the handle's
            // (blocking) stream; a graph launch does not take part in it, so order explicitly in both directions.
            if (!stream) {
                if (!h->gev && hipEventCreateWithFlags(&h->gev, hipEventDisableTiming) != hipSuccess)
                    return fail(PFNL_ERR_HIP, "event creation failed");
                HIPCHK(hipEventRecord(h->gev, nullptr));
                HIPCHK(hipStreamWaitEvent(s, h->gev, 0));
            }
            if (in_is_device)
                HIPCHK(hipMemcpyAsync(h->stage_in.p, in, n_in * sizeof(float), hipMemcpyDeviceToDevice, s));
            else if (int e = stage_h2d(h, h->stage_in.p, in, n_in * sizeof(float), s))
                return e;
            HIPCHK(hipGraphLaunch(ge->exec, s));
            if (out_is_device)
                HIPCHK(hipMemcpyAsync(out, h->stage_out.p, n_out * sizeof(float), hipMemcpyDeviceToDevice, s));
            if (sync_call) {
                if (!out_is_device) {
                    if (int e = stage_d2h(h, out, h->stage_out.p, n_out * sizeof(float), s)) return e;
                } else {
                    HIPCHK(hipStreamSynchronize(s));
                }
                int flagged = 0;
                if (int e = range_flag_take(h, 0, &flagged)) return e;
                if (flagged && can_rerun) {   // once more, eagerly, on the f32-MFMA kernels
                    h->strict_once = true;
                    const int e = forward_device(h, h->stage_in.p, h->stage_out.p, B, H, W, s, nullptr, 0, -1, 0);
                    h->strict_once = false;
                    if (e) return e;
                    ++h->range_reruns;
                    if (out_is_device) {
                        HIPCHK(hipMemcpyAsync(out, h->stage_out.p, n_out * sizeof(float), hipMemcpyDeviceToDevice, s));
                        HIPCHK(hipStreamSynchronize(s));
                    } else if (int e2 = stage_d2h(h, out, h->stage_out.p, n_out * sizeof(float), s)) {
                        return e2;
                    }
                } else if (flagged && h->bf16) {
                    return fail(PFNL_ERR_RANGE, RANGE_MSG_BF16);
                }
            } else if (!stream) {
                HIPCHK(hipEventRecord(h->gev, s));
                HIPCHK(hipStreamWaitEvent(nullptr, h->gev, 0));
            }
            h->lastB = B;
            h->lastH = H;
            h->lastW = W;
            return 0;
        }
        ge->seen++;
    }
    if (!in_is_device) {
        if (h->stage_in.ensure(n_in)) return fail(PFNL_ERR_NOMEM, "staging allocation failed");
        if (int e = stage_h2d(h, h->stage_in.p, in, n_in * sizeof(float), s)) return e;
        din = h->stage_in.p;
    }
    if (!out_is_device) {
        if (h->stage_out.ensure(n_out)) return fail(PFNL_ERR_NOMEM, "staging allocation failed");
        dout = h->stage_out.p;
    }
    if (int e = forward_device(h, din, dout, B, H, W, s, nullptr, 0, -1, slot)) return e;
    if (!out_is_device) {
        if (int e = stage_d2h(h, out, dout, n_out * sizeof(float), s)) return e;
    } else if (!in_is_device) {
        HIPCHK(hipStreamSynchronize(s));   // the caller may reuse its host buffer on return
    }
    if (sync_call) {
        // a synchronous call: its range flag (slot 0, pinned) is read before it returns; set = the f16-pipe domain was left somewhere
        // (or the input itself was non-finite): once more on the f32-MFMA kernels, whose range is the reference's
        int flagged = 0;
        if (int e = range_flag_take(h, 0, &flagged)) return e;
        if (flagged && can_rerun) {
            h->strict_once = true;
            const int e = forward_device(h, din, dout, B, H, W, s, nullptr, 0, -1, 0);   // (strict: the fence is not armed; still
            h->strict_once = false;                                                    //  non-finite = so is the reference's result)
            if (e) return e;
            ++h->range_reruns;
            if (!out_is_device) {
                if (int e2 = stage_d2h(h, out, dout, n_out * sizeof(float), s)) return e2;
            } else {
                HIPCHK(hipStreamSynchronize(s));
            }
        } else if (flagged && h->bf16) {
            return fail(PFNL_ERR_RANGE, RANGE_MSG_BF16);
        }
    }
    return 0;
}

// Single-clip multi-GPU sharding: one rank's horizontal strip of the SR frame.
int pfnl_forward_strip(pfnl_handle* h, const void* in, void* out, int B, int H, int W, int row0, int nrows, void* stream) {
    if (!h || !in || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (!h->finalized) return fail(PFNL_ERR_STATE, "pfnl_finalize_weights has not been called");
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return fail(PFNL_ERR_INVALID, "B, H, W must be positive, H and W even");
    if (row0 < 0 || nrows < 1 || row0 + nrows > H) return fail(PFNL_ERR_INVALID, "strip rows out of range");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    // receptive field of an output row in LR rows: tail conv (1) + convmerge1 (1) + 2 per progressive-fusion block + conv0 (2)
    const int halo = 4 + 2 * h->cfg.num_block;
    int a = row0 - halo, b = row0 + nrows + halo;
    a = a < 0 ? 0 : (a & ~1);                                    // even: space_to_depth(2) rows, Winograd tiles
    b = b > H ? H : ((b + 1) & ~1);
    const StripGeom g{a, b - a, row0 - a, row0 - a + nrows};
    // non-local block: queries = the packed rows conv0 reads (2 LR rows of margin for its 5x5), keys = the whole frame
    const int W2 = W / 2, H2 = H / 2;
    int qa = (a - 2) >> 1, qb = (b + 2 + 1) >> 1;
    qa = qa < 0 ? 0 : qa;
    qb = qb > H2 ? H2 : qb;
    hipStream_t s = stream ? (hipStream_t)stream : (hipStream_t) nullptr;
    return forward_device(h, (const float*)in, (float*)out, B, H, W, s, &g, qa * W2, qb * W2);
}

int pfnl_sync(pfnl_handle* h) {
    if (!h) return fail(PFNL_ERR_INVALID, "NULL handle");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamSynchronize(nullptr));                     // device-pointer calls with stream == NULL run on the null stream
    // (a device-pointer call on a stream of the caller's: the caller synchronises that stream before pfnl_sync)
    int flagged = 0;
    if (int e = range_flag_take(h, 1, &flagged)) return e;
    if (flagged) return fail(PFNL_ERR_RANGE, h->bf16 ? RANGE_MSG_BF16 : RANGE_MSG_FP32);
    return 0;
}

// Page-locked host memory for a caller's input / output buffers (hipHostMalloc): a host-pointer pfnl_forward lets the copy engine read /
// write such a buffer directly instead of staging it through the handle's pinned strips.  Process-wide (no handle: a result may outlive
// the handle that filled it); the caller frees every block it took.
int pfnl_host_alloc(size_t bytes, void** out) {
    if (!out || !bytes) return fail(PFNL_ERR_INVALID, "bad host allocation");
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess || !p) {
        (void)hipGetLastError();
        return fail(PFNL_ERR_NOMEM, "hipHostMalloc failed");
    }
    *out = p;
    return 0;
}

int pfnl_host_free(void* p) {
    if (p && hipHostFree(p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(PFNL_ERR_HIP, "hipHostFree failed");
    }
    return 0;
}

int pfnl_range_flag(pfnl_handle* h, int* flagged) {
    if (!h || !flagged) return fail(PFNL_ERR_INVALID, "NULL argument");
    return range_flag_take(h, 1, flagged);
}

// Weight replica over RCCL (comm.hip holds the communicator): the packed device blobs root -> all.
// A handle that is about to RECEIVE packed weights (broadcast / copy): the blob layout depends on the geometry only, so it
// is built from zeros and the sender's bytes go over it.  Everything data-dependent lives in the three device blobs.
static int prepare_to_receive_weights(pfnl_handle* h) {
    if (h->finalized) return 0;
    for (auto& kv : h->expected)
        if (!h->host.count(kv.first)) {
            HostTensor t;
            t.shape = kv.second;
            t.data.assign(numel(t.shape), 0.f);
            h->host[kv.first] = std::move(t);
        }
    const int e = pfnl_finalize_weights(h);
    if (!e) h->weights_external = true;                         // placeholders: a later pfnl_set_weight starts from an empty set
    return e;
}

int pfnl_copy_weights(pfnl_handle* dst, pfnl_handle* src) {
    if (!dst || !src) return fail(PFNL_ERR_INVALID, "NULL handle");
    if (dst == src) return 0;
    if (!src->finalized) return fail(PFNL_ERR_STATE, "the source handle has no finalized weights");
    if (src->nl_theta) return fail(PFNL_ERR_INVALID, "theta/phi handles carry host-side state: load them through pfnl_set_weight");
    if (int e = prepare_to_receive_weights(dst)) return e;
    if (dst->wdev_used != src->wdev_used || dst->wdev16_used != src->wdev16_used || dst->wdev16s_used != src->wdev16s_used)
        return fail(PFNL_ERR_STATE, "the handles differ in geometry (weight blob sizes)");
    HIPCHK(hipSetDevice(dst->cfg.device_id));
    HIPCHK(hipMemcpy(dst->wdev.p, src->wdev.p, src->wdev_used * sizeof(float), hipMemcpyDefault));
    HIPCHK(hipMemcpy(dst->wdev16.p, src->wdev16.p, src->wdev16_used * sizeof(float), hipMemcpyDefault));
    HIPCHK(hipMemcpy(dst->wdev16s.p, src->wdev16s.p, src->wdev16s_used * sizeof(float), hipMemcpyDefault));
    dst->weights_f16_ok = src->weights_f16_ok;                  // (the range of the weights travels with them)
    dst->weights_external = true;
    ++dst->cfg_gen;
    return 0;
}

int pfnl_comm_bcast_weights(pfnl_comm* c, pfnl_handle* h, int root) {
    if (!c || !h) return fail(PFNL_ERR_INVALID, "NULL argument");
    int rank = 0, nranks = 0;
    if (int e = pfnl_comm_rank(c, &rank, &nranks)) return e;
    if (root < 0 || root >= nranks) return fail(PFNL_ERR_INVALID, "bad root");
    if (rank == root && !h->finalized) return fail(PFNL_ERR_STATE, "root has no finalized weights");
    // a rank whose local preparation fails still takes part in the agreement below (status rides in v[6]: the others must not be
    // left waiting inside the collective), then every rank returns the same verdict
    const int prep = rank != root ? prepare_to_receive_weights(h) : 0;
    const std::string prep_msg = prep ? std::string(pfnl_last_error()) : std::string();
    double v[8] = {(double)h->wdev_used, -(double)h->wdev_used, (double)h->wdev16_used, -(double)h->wdev16_used,
                   (double)h->wdev16s_used, -(double)h->wdev16s_used, prep ? 1.0 : 0.0, (rank == root && !h->weights_f16_ok) ? 1.0 : 0.0};
    if (int e = pfnl_comm_allreduce_f64(c, v, 8, PFNL_COMM_MAX)) return e;
    if (v[6] != 0.0) return prep ? fail(prep, prep_msg) : fail(PFNL_ERR_STATE, "another rank could not prepare to receive the weights");
    if (v[0] != -v[1] || v[2] != -v[3] || v[4] != -v[5]) return fail(PFNL_ERR_STATE, "ranks disagree on the weight blob size (geometry / theta-phi option)");
    if (int e = pfnl_comm_bcast(c, h->wdev.p, h->wdev_used * sizeof(float), root)) return e;
    if (int e = pfnl_comm_bcast(c, h->wdev16.p, h->wdev16_used * sizeof(float), root)) return e;
    if (int e = pfnl_comm_bcast(c, h->wdev16s.p, h->wdev16s_used * sizeof(float), root)) return e;
    if (rank != root) {
        h->weights_f16_ok = v[7] == 0.0;                            // (the root's verdict on the range of the weights travels with them)
        h->weights_external = true;
    }
    ++h->cfg_gen;
    return 0;
}

int pfnl_profile_enable(pfnl_handle* h, int enable) {
    if (!h) return fail(PFNL_ERR_INVALID, "NULL handle");
    if (!enable && h->prof) {
        if (prof_collect(h)) return fail(PFNL_ERR_HIP, "event collection failed");
    }
    h->prof = enable != 0;
    h->prof_mode = enable == 2 ? 2 : enable == 3 ? 3 : enable == 4 ? 4 : (enable ? 1 : 0);
    h->prof_gate = h->prof_mode != 3;
    h->run_n = 0;
    h->run_cls = -1;
    return 0;
}

int pfnl_profile_reset(pfnl_handle* h) {
    if (!h) return fail(PFNL_ERR_INVALID, "NULL handle");
    if (prof_collect(h)) return fail(PFNL_ERR_HIP, "event collection failed");
    for (int i = 0; i < PFNL_K_COUNT; ++i) {
        h->prof_ms[i] = 0;
        h->prof_n[i] = 0;
    }
    return 0;
}

int pfnl_profile_read(pfnl_handle* h, double* ms, int64_t* launches) {
    if (!h || !ms || !launches) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (prof_collect(h)) return fail(PFNL_ERR_HIP, "event collection failed");
    for (int i = 0; i < PFNL_K_COUNT; ++i) {
        ms[i] = h->prof_ms[i];
        launches[i] = h->prof_n[i];
    }
    return 0;
}

int pfnl_debug_tap(pfnl_handle* h, const char* name, float* host_dst, size_t count) {
    if (!h || !name || !host_dst) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (!h->lastB) return fail(PFNL_ERR_STATE, "no forward has run");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipDeviceSynchronize());
    const int B = h->lastB, H = h->lastH, W = h->lastW, T = h->cfg.num_frames;
    const std::string n(name);
    const float* src = nullptr;
    size_t need = 0;
    if (n == "nl_out") {
        need = (size_t)B * H * W * 3 * T;
        if (h->scratch.ensure(need)) return fail(PFNL_ERR_NOMEM, "scratch allocation failed");
        HIPCHK(pfnl::launch_nl_unpack(h->Xo.p, h->scratch.p, B, T, H, W, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        src = h->scratch.p;
    } else if (n == "trunk") {
        need = (size_t)B * T * H * W * 64;
        src = h->inp0.p;
        if (h->bf16) {                                              // bf16 trunk: cast on demand
            if (h->scratch.ensure(need)) return fail(PFNL_ERR_NOMEM, "scratch allocation failed");
            HIPCHK(pfnl::launch_cast_bf16_f32(reinterpret_cast<const uint16_t*>(h->inp0.p), h->scratch.p, need, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            src = h->scratch.p;
        }
    } else if (n == "merge1") {
        need = (size_t)B * H * W * 48;
        if (count != need) return fail(PFNL_ERR_INVALID, "tap size mismatch");
        HIPCHK(hipMemcpy2D(host_dst, 48 * sizeof(float), h->merge.p, h->merge_cstride * sizeof(float), 48 * sizeof(float),
                           (size_t)B * H * W, hipMemcpyDeviceToHost));
        return 0;
    } else {
        return fail(PFNL_ERR_INVALID, "unknown tap " + n);
    }
    if (count != need) return fail(PFNL_ERR_INVALID, "tap size mismatch");
    HIPCHK(hipMemcpy(host_dst, src, need * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// ---- single ops -------------------------------------------------------------------------------

int pfnl_op_conv2d(const float* in, const float* kernel_host, const float* bias_host, const float* addend,
                   int add_div, const float* resid, float* out, int items, int frames_per_item, int H, int W,
                   int ksize, int cout, int act, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if ((ksize != 1 && ksize != 3) || cout < 1 || cout > 64 || items < 1 || frames_per_item < 1 || H < 1 || W < 1)
        return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    if ((addend != nullptr) != (resid != nullptr))
        return fail(PFNL_ERR_INVALID, "addend and resid must be given together (fused conv2 epilogue) or not at all");
    if (addend && add_div < 1) return fail(PFNL_ERR_INVALID, "add_div must be >= 1");
    if (addend && cout != 64) return fail(PFNL_ERR_INVALID, "fused epilogue needs cout == 64");
    hipStream_t s = (hipStream_t)stream;
    const int cin = 64 * frames_per_item;
    std::vector<float> pack(pfnl::conv_pack_floats(ksize, cin) + 64, 0.f);
    pfnl::conv_pack_weights(kernel_host, ksize, cin, 0, cin, cout, pack.data());
    const size_t boff = pack.size() - 64;
    if (bias_host) std::memcpy(&pack[boff], bias_host, cout * sizeof(float));
    float* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(float)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        pfnl::ConvParams p{};
        p.in = in;
        p.wpack = dw;
        p.bias = dw + boff;   // zeros when bias_host is NULL
        p.addend = addend;
        p.resid = resid;
        p.out = out;
        p.H = H;
        p.W = W;
        p.in_cstride = 64;
        p.out_cstride = cout;
        p.cout = cout;
        p.chunks_per_frame = 64 / pfnl::CONV_CK;
        p.frames_per_item = frames_per_item;
        p.nchunks = frames_per_item * p.chunks_per_frame;
        p.add_div = addend ? add_div : 1;
        p.act = act;
        e = pfnl::launch_conv_mfma(p, ksize, items, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_conv2_grouped(const float* in, const float* base, const float* kernel_host, const float* bias_host,
                          const float* resid, float* out, int clips, int frames_per_clip, int H, int W, int act,
                          void* stream) {
    if (!in || !base || !kernel_host || !resid || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (clips < 1 || frames_per_clip < 1 || H < 2 || W < 2 || (H & 1) || (W & 1))
        return fail(PFNL_ERR_INVALID, "grouped conv2 needs even H, W");
    if ((long long)H * W * 256 >= 0x7fffffffLL) return fail(PFNL_ERR_INVALID, "frame too large for the grouped kernel");
    hipStream_t s = (hipStream_t)stream;
    const size_t pf = pfnl::wino_pack_floats();
    std::vector<float> pack(2 * pf + 64, 0.f);
    pfnl::wino_pack_weights(kernel_host, 128, 0, pack.data());            // rows 0..63 multiply `base`
    pfnl::wino_pack_weights(kernel_host, 128, 64, pack.data() + pf);      // rows 64..127 multiply the frame
    if (bias_host) std::memcpy(&pack[2 * pf], bias_host, 64 * sizeof(float));
    float* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(float)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        pfnl::WinoParams wp{};
        wp.in = in;
        wp.in2 = base;
        wp.upack = dw + pf;
        wp.upack2 = dw;
        wp.bias = dw + 2 * pf;
        wp.resid = resid;
        wp.out = out;
        wp.H = H;
        wp.W = W;
        wp.add_div = frames_per_clip;
        wp.act = act;
        wp.items = clips * frames_per_clip;
        e = pfnl::launch_conv_wino_ws(wp, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("grouped conv2 op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_conv3x3_accum(const float* in, const float* kernel_host, const float* bias_host, float* out, int clips,
                          int frames_per_clip, int H, int W, int cout, int act, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (clips < 1 || frames_per_clip < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || cout < 1 || cout > 64)
        return fail(PFNL_ERR_INVALID, "accumulating conv needs even H, W and cout <= 64");
    if ((long long)H * W * 256 >= 0x7fffffffLL) return fail(PFNL_ERR_INVALID, "frame too large for the persistent kernel");
    hipStream_t s = (hipStream_t)stream;
    const int T = frames_per_clip;
    const size_t pf = pfnl::wino_pack_floats();
    std::vector<float> pack(T * pf + 64, 0.f);
    for (int f = 0; f < T; ++f) pfnl::wino_pack_weights(kernel_host, 64 * T, 64 * f, pack.data() + f * pf, cout);
    if (bias_host) std::memcpy(&pack[T * pf], bias_host, cout * sizeof(float));
    float* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(float)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        pfnl::WinoParams wp{};
        wp.in = in;
        wp.upack = dw;
        wp.upack_stride = (long long)pf;
        wp.accum = 1;
        wp.bias = dw + T * pf;
        wp.out = out;
        wp.H = H;
        wp.W = W;
        wp.add_div = T;
        wp.act = act;
        wp.items = clips * T;
        e = pfnl::launch_conv_wino_ws(wp, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("accumulating conv op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_conv3x3_accum_split16(const float* in, const float* kernel_host, const float* bias_host, float* out, int clips,
                                  int frames_per_clip, int H, int W, int cout, int act, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (clips < 1 || frames_per_clip < 1 || H < 1 || W < 1 || cout < 1 || cout > 64) return fail(PFNL_ERR_INVALID, "accumulating conv needs cout <= 64");
    if ((long long)H * W * 256 >= 0x7fffffffLL) return fail(PFNL_ERR_INVALID, "frame too large for the persistent kernel");
    hipStream_t s = (hipStream_t)stream;
    const int T = frames_per_clip;
    const size_t nh = pfnl::conv3x3_split16_pack_halfs();
    std::vector<uint16_t> pack((size_t)T * nh + 128, 0);
    for (int f = 0; f < T; ++f) pfnl::conv3x3_split16_pack_weights(kernel_host, 64 * T, 64 * f, pack.data() + (size_t)f * nh, cout);
    if (bias_host) std::memcpy(&pack[(size_t)T * nh], bias_host, cout * sizeof(float));
    uint16_t* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    pfnl::ConvSplitParams q{in, dw, reinterpret_cast<const float*>(dw + (size_t)T * nh), nullptr, nullptr, out, H, W, clips * T, T, act, 1};
    if (e == hipSuccess) e = pfnl::launch_conv3x3_split16(q, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("accumulating conv (split16) op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_conv3x3_bf16(const uint16_t* in, const float* kernel_host, const float* bias_host, const uint16_t* addend,
                         int add_div, const uint16_t* resid, uint16_t* out, int items, int H, int W, int act, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (items < 1 || H < 1 || W < 1 || (addend == nullptr) != (resid == nullptr)) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    hipStream_t s = (hipStream_t)stream;
    const size_t nh = pfnl::conv3x3_bf16_pack_halfs();
    std::vector<uint16_t> pack(nh + 128, 0);
    pfnl::conv3x3_bf16_pack_weights(kernel_host, 64, 0, pack.data());
    if (bias_host) std::memcpy(&pack[nh], bias_host, 64 * sizeof(float));
    uint16_t* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    pfnl::ConvBf16Params q{in, dw, reinterpret_cast<const float*>(dw + nh), addend, resid, out, H, W, items, add_div < 1 ? 1 : add_div, act};
    if (e == hipSuccess) e = pfnl::launch_conv3x3_bf16(q, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv3x3 bf16 op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_conv3x3_split16(const float* in, const float* kernel_host, const float* bias_host, const float* addend, int add_div,
                            const float* resid, float* out, int items, int H, int W, int act, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (items < 1 || H < 1 || W < 1 || (addend == nullptr) != (resid == nullptr)) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    if (addend && (add_div < 1 || items % add_div)) return fail(PFNL_ERR_INVALID, "items must be a multiple of add_div");
    hipStream_t s = (hipStream_t)stream;
    const size_t nh = pfnl::conv3x3_split16_pack_halfs();
    std::vector<uint16_t> pack(nh + 128, 0);
    pfnl::conv3x3_split16_pack_weights(kernel_host, 64, 0, pack.data());
    if (bias_host) std::memcpy(&pack[nh], bias_host, 64 * sizeof(float));
    uint16_t* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    pfnl::ConvSplitParams q{in, dw, reinterpret_cast<const float*>(dw + nh), addend, resid, out, H, W, items, add_div < 1 ? 1 : add_div, act};
    if (e == hipSuccess) e = pfnl::launch_conv3x3_split16(q, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv3x3 split16 op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_conv1_conv10_bf16(const uint16_t* in, const float* k1_host, const float* b1_host, const float* k10_host,
                              const float* b10_host, uint16_t* out1, uint16_t* base, int clips, int frames_per_clip, int H, int W,
                              void* stream) {
    if (!in || !k1_host || !k10_host || !out1 || !base) return fail(PFNL_ERR_INVALID, "NULL argument");
    const int T = frames_per_clip;
    if (clips < 1 || (T != 3 && T != 5 && T != 7) || H < 1 || W < 1) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    hipStream_t s = (hipStream_t)stream;
    const size_t n3 = pfnl::conv3x3_bf16_pack_halfs(), n1 = pfnl::conv1x1_bf16_pack_halfs(T);
    std::vector<uint16_t> pack(n3 + n1 + 256, 0);
    pfnl::conv3x3_bf16_pack_weights(k1_host, 64, 0, pack.data());
    pfnl::conv1x1_bf16_pack_weights(k10_host, T, pack.data() + n3);
    if (b1_host) std::memcpy(&pack[n3 + n1], b1_host, 64 * sizeof(float));
    if (b10_host) std::memcpy(&pack[n3 + n1 + 128], b10_host, 64 * sizeof(float));
    uint16_t* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    pfnl::ConvBf16Params q{in, dw, reinterpret_cast<const float*>(dw + n3 + n1), nullptr, nullptr, out1, H, W, clips * T, T, 1,
                           dw + n3, reinterpret_cast<const float*>(dw + n3 + n1 + 128), base};
    if (e == hipSuccess) e = pfnl::launch_conv3x3_bf16(q, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv1+conv10 bf16 op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_conv3x3_accum_bf16(const uint16_t* in, const float* kernel_host, const float* bias_host, float* out, int clips,
                               int frames_per_clip, int H, int W, int cout, int act, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    const int T = frames_per_clip;
    if (clips < 1 || T < 1 || T > 7 || H < 1 || W < 1 || cout < 1 || cout > 64) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    hipStream_t s = (hipStream_t)stream;
    const size_t nh = pfnl::conv3x3_bf16_pack_halfs();
    std::vector<uint16_t> pack((size_t)T * nh + 128, 0);
    for (int f = 0; f < T; ++f) pfnl::conv3x3_bf16_pack_weights(kernel_host, 64 * T, 64 * f, &pack[(size_t)f * nh], cout);
    if (bias_host) std::memcpy(&pack[(size_t)T * nh], bias_host, cout * sizeof(float));
    uint16_t* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    pfnl::ConvBf16Params q{in, dw, reinterpret_cast<const float*>(dw + (size_t)T * nh), nullptr, nullptr, nullptr, H, W, clips * T, T, act};
    q.out_f32 = out;
    if (e == hipSuccess) e = pfnl::launch_conv3x3_bf16(q, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv3x3 accum bf16 op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_conv1x1_bf16(const uint16_t* in, const float* kernel_host, const float* bias_host, uint16_t* out, int items,
                         int frames_per_item, int HW, int act, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    const int T = frames_per_item;
    if (items < 1 || (T != 3 && T != 5 && T != 7) || HW < 1) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    hipStream_t s = (hipStream_t)stream;
    const size_t nh = pfnl::conv1x1_bf16_pack_halfs(T);
    std::vector<uint16_t> pack(nh + 128, 0);
    pfnl::conv1x1_bf16_pack_weights(kernel_host, T, pack.data());
    if (bias_host) std::memcpy(&pack[nh], bias_host, 64 * sizeof(float));
    uint16_t* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pfnl::launch_conv1x1_bf16(in, dw, reinterpret_cast<const float*>(dw + nh), out, items, T, HW, act, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv1x1 bf16 op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_conv1x1_split16(const float* in, const float* kernel_host, const float* bias_host, float* out, int items,
                            int frames_per_item, int HW, int act, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (items < 1 || frames_per_item < 1 || HW < 1) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    hipStream_t s = (hipStream_t)stream;
    const int T = frames_per_item;
    const size_t nh = pfnl::conv1x1_split16_pack_halfs(T);
    std::vector<uint16_t> pack(nh + 128, 0);
    pfnl::conv1x1_split16_pack_weights(kernel_host, T, pack.data());
    if (bias_host) std::memcpy(&pack[nh], bias_host, 64 * sizeof(float));
    uint16_t* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pfnl::launch_conv1x1_split16(in, dw, reinterpret_cast<const float*>(dw + nh), out, items, T, HW, act, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv1x1 split16 op: ") + hipGetErrorString(e));
    return 0;
}

// The small-shape trunk kernel (conv_small.hip; ConvSmallParams in conv_small.h says which tensor each source comes from):
// out[i] = act(sum_s conv_ks(src(i, s); kernel rows [64 s, 64 s + 64)) + bias) (+ resid[i]); kernel HWIO [ks, ks, 64 nsrc, cout]
int pfnl_op_conv_small(const float* a, const float* b, int nA, int a_div, int b_mul, int nsrc, const float* kernel_host,
                       const float* bias_host, const float* resid, float* out, int items, int H, int W, int ks, int cout, int act,
                       void* stream) {
    if (!b || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (items < 1 || H < 1 || W < 1 || nsrc < 1 || nsrc > 16 || (ks != 1 && ks != 3) || cout < 1 || cout > 64 || nA < 0 || nA > nsrc || (nA && !a))
        return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    hipStream_t s = (hipStream_t)stream;
    const size_t nh = pfnl::conv_small_pack_halfs(ks, nsrc);
    std::vector<uint16_t> pack(nh + 128, 0);
    pfnl::conv_small_pack_weights(kernel_host, ks, nsrc, cout, pack.data());
    if (bias_host) std::memcpy(&pack[nh], bias_host, cout * sizeof(float));
    uint16_t* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    pfnl::ConvSmallParams q{a, b, nA, a_div < 1 ? 1 : a_div, b_mul < 1 ? 1 : b_mul, nsrc, dw, reinterpret_cast<const float*>(dw + nh), resid, out, H, W, items, act, ks};
    if (e == hipSuccess) e = pfnl::launch_conv_small(q, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv_small op: ") + hipGetErrorString(e));
    return 0;
}

// One progressive-fusion block on the small-shape kernels as the forward launches it since round 4 (two launches, conv_small.h):
// inp1 = lrelu(conv3x3(x; k1) + b1) together with the per-frame partials of conv10_i; out = x + lrelu(conv3x3(concat([base, inp1_t]); k2)
// + b2) with base = lrelu(sum_t partial_t + b10) built in the second launch's prologue (reference model/pfnl.py:66-71).
int pfnl_op_conv_small_pf_block(const float* x, const float* k1_host, const float* b1_host, const float* k10_host, const float* b10_host,
                                const float* k2_host, const float* b2_host, float* inp1, float* out, int clips, int T, int H, int W,
                                void* stream) {
    if (!x || !k1_host || !k10_host || !k2_host || !inp1 || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (clips < 1 || T < 1 || T > 7 || H < 1 || W < 1) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    hipStream_t s = (hipStream_t)stream;
    const size_t n1 = pfnl::conv_small_pack_halfs(3, 1), n10 = pfnl::conv_small_pack_halfs(1, T), n2 = pfnl::conv_small_pack_halfs(3, 2);
    std::vector<uint16_t> pack(n1 + n10 + n2 + 3 * 128, 0);
    pfnl::conv_small_pack_weights(k1_host, 3, 1, 64, pack.data());
    pfnl::conv_small_pack_weights(k10_host, 1, T, 64, pack.data() + n1);
    pfnl::conv_small_pack_weights(k2_host, 3, 2, 64, pack.data() + n1 + n10);
    uint16_t* const bh = pack.data() + n1 + n10 + n2;
    if (b1_host) std::memcpy(bh, b1_host, 64 * sizeof(float));
    if (b10_host) std::memcpy(bh + 128, b10_host, 64 * sizeof(float));
    if (b2_host) std::memcpy(bh + 256, b2_host, 64 * sizeof(float));
    const size_t F = (size_t)clips * T, tensor = F * H * W * 64;
    uint16_t* dw = nullptr;
    float* part = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    if (hipMalloc(&part, tensor * sizeof(float)) != hipSuccess) {
        (void)hipFree(dw);
        return fail(PFNL_ERR_NOMEM, "allocation failed");
    }
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    const float* const db = reinterpret_cast<const float*>(dw + n1 + n10 + n2);
    if (e == hipSuccess) {
        pfnl::ConvSmallParams q{nullptr, x, 0, 1, 1, 1, dw, db, nullptr, inp1, H, W, (int)F, 1, 3};
        q.x_wpack = dw + n1;
        q.x_out = part;
        q.x_T = T;
        e = pfnl::launch_conv_small(q, s);
    }
    if (e == hipSuccess) {
        pfnl::ConvSmallParams q{part, inp1, 1, T, 1, 2, dw + n1 + n10, db + 128, x, out, H, W, (int)F, 1, 3};
        q.a_nsum = T;
        q.a_bias = db + 64;
        e = pfnl::launch_conv_small(q, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    (void)hipFree(part);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv_small block op: ") + hipGetErrorString(e));
    return 0;
}

// ---- the split-format ("SF", conv_split16.h) variants of the split-f16 kernels, op by op.  The hooks take and return fp32
// tensors: fp32 -> SF and SF -> fp32 (hi + lo' 2^-11) conversions bracket the kernel under test, so that each of them is checked
// against the fp64 spec at its own scale and not only inside the forward.
//   which = 0: conv3x3_sf_kernel (input SF by LDS-DMA, epilogue from registers; plain or fused with addend + resid)
//   which = 1: conv3x3_split16_kernel<0, OSF> (conv1_i: fp32 in, SF out)
static int op_conv2_chain(const float* in, const float* kernel_host, const float* bias_host, const float* addend, int add_div, const float* resid,
                          float* out, uint16_t* out_sf, int items, int H, int W, int act, void* stream) {
    {
        {
        hipStream_t s2 = (hipStream_t)stream;
        const size_t nh2 = pfnl::conv3x3_split16_pack_halfs();
        std::vector<uint16_t> pk(2 * nh2 + 128, 0);
        pfnl::conv3x3_split16_pack_weights(kernel_host, 128, 0, pk.data(), 64, true);
        pfnl::conv3x3_split16_pack_weights(kernel_host, 128, 64, pk.data() + nh2, 64, true);
        if (bias_host) std::memcpy(&pk[2 * nh2], bias_host, 64 * sizeof(float));
        const size_t npf = (size_t)items * H * W, npb = (size_t)(items / add_div) * H * W;
        uint16_t *dw2 = nullptr, *tf = nullptr, *tb = nullptr;
        HIPCHK(hipMalloc(&dw2, pk.size() * sizeof(uint16_t)));
        hipError_t e2 = hipMalloc(&tf, npf * 256);
        if (e2 == hipSuccess) e2 = hipMalloc(&tb, npb * 256);
        if (e2 == hipSuccess) e2 = hipMemcpy(dw2, pk.data(), pk.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e2 == hipSuccess) e2 = pfnl::launch_sf_from_f32(in, tf, npf, s2);
        if (e2 == hipSuccess) e2 = pfnl::launch_sf_from_f32(addend, tb, npb, s2);
        if (e2 == hipSuccess && out != resid) e2 = hipMemcpyAsync(out, resid, npf * 256, hipMemcpyDeviceToDevice, s2);   // the kernel works in place
        pfnl::ConvSplitParams q{reinterpret_cast<const float*>(tf), dw2 + nh2, reinterpret_cast<const float*>(dw2 + 2 * nh2), nullptr, out, out, H, W, items, add_div, act};
        q.in2 = reinterpret_cast<const float*>(tb);
        q.wpack2 = dw2;
        q.out2 = reinterpret_cast<float*>(out_sf);                      // (null: no split-format copy)
        if (e2 == hipSuccess) e2 = pfnl::launch_conv3x3_sf_chain(q, s2);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(s2);
        (void)hipFree(dw2);
        (void)hipFree(tf);
        (void)hipFree(tb);
        if (e2 != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv2 chain op: ") + hipGetErrorString(e2));
        return 0;
        }
    }
}
// the whole of conv2_i in one launch WITH the split-format copy of its output (conv3x3_sf_chain_kernel<true>, option split16_sf0):
// out as pfnl_op_conv3x3_split16_sf(which = 2); out_sf [items][H][W][128] binary16 bit patterns (device) = the split format of `out`
int pfnl_op_conv2_chain_sf0(const float* in, const float* kernel_host, const float* bias_host, const float* base, int add_div, const float* resid,
                            float* out, uint16_t* out_sf, int items, int H, int W, int act, void* stream) {
    if (!in || !kernel_host || !out || !out_sf || !base || !resid) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (items < 1 || H < 1 || W < 1 || add_div < 1 || items % add_div) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    return op_conv2_chain(in, kernel_host, bias_host, base, add_div, resid, out, out_sf, items, H, W, act, stream);
}

int pfnl_op_conv3x3_split16_sf(int which, const float* in, const float* kernel_host, const float* bias_host, const float* addend,
                               int add_div, const float* resid, float* out, int items, int H, int W, int act, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (which < 0 || which > 2 || items < 1 || H < 1 || W < 1 || (addend == nullptr) != (resid == nullptr)) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    if (addend && ((which != 0 && which != 2) || add_div < 1 || items % add_div)) return fail(PFNL_ERR_INVALID, "fused mode: which = 0 or 2, items a multiple of add_div");
    if (which == 2) {   // the whole of conv2_i (conv3x3_sf_chain_kernel): kernel_host = HWIO [3,3,128,64], `addend` = base [items/add_div][H][W][64] fp32
        if (!addend) return fail(PFNL_ERR_INVALID, "which = 2 needs base (addend argument) and resid");
        return op_conv2_chain(in, kernel_host, bias_host, addend, add_div, resid, out, nullptr, items, H, W, act, stream);
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t nh = pfnl::conv3x3_split16_pack_halfs();
    std::vector<uint16_t> pack(nh + 128, 0);
    pfnl::conv3x3_split16_pack_weights(kernel_host, 64, 0, pack.data(), 64, which == 0);
    if (bias_host) std::memcpy(&pack[nh], bias_host, 64 * sizeof(float));
    const size_t npix = (size_t)items * H * W;
    uint16_t *dw = nullptr, *tmp = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    hipError_t e = hipMalloc(&tmp, npix * 256);
    if (e == hipSuccess) e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (which == 0) {
        if (e == hipSuccess) e = pfnl::launch_sf_from_f32(in, tmp, npix, s);
        pfnl::ConvSplitParams q{reinterpret_cast<const float*>(tmp), dw, reinterpret_cast<const float*>(dw + nh), addend, resid, out, H, W, items, add_div < 1 ? 1 : add_div, act};
        if (e == hipSuccess) e = pfnl::launch_conv3x3_sf(q, s);
    } else {
        pfnl::ConvSplitParams q{in, dw, reinterpret_cast<const float*>(dw + nh), nullptr, nullptr, reinterpret_cast<float*>(tmp), H, W, items, 1, act};
        q.out_sf = 1;
        if (e == hipSuccess) e = pfnl::launch_conv3x3_split16(q, s);
        if (e == hipSuccess) e = pfnl::launch_sf_to_f32(tmp, out, npix, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv3x3 split16 SF op: ") + hipGetErrorString(e));
    return 0;
}

// conv1_i + conv10_i as ONE launch (conv3x3_c1c10_kernel): in fp32 [clips*T][H][W][64] -> out1 = inp1 [clips*T][H][W][64], base [clips][H][W][64];
// the kernel writes both in the split format, the hook hands them back as fp32 (hi + lo' 2^-11: what the consumers' MFMAs see)
static int op_conv1_conv10_split16(const float* in, const float* k1_host, const float* b1_host, const float* k10_host,
                                   const float* b10_host, float* out1, float* base, int clips, int frames_per_clip, int H, int W,
                                   void* stream, bool in_sf) {
    if (!in || !k1_host || !k10_host || !out1 || !base) return fail(PFNL_ERR_INVALID, "NULL argument");
    const int T = frames_per_clip;
    if (clips < 1 || T < 1 || T > 7 || H < 1 || W < 1) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    hipStream_t s = (hipStream_t)stream;
    const size_t n3 = pfnl::conv3x3_split16_pack_halfs(), n1 = pfnl::conv1x1_c10_pack_halfs(T);
    std::vector<uint16_t> pack(n3 + n1 + 256, 0);
    pfnl::conv3x3_split16_pack_weights(k1_host, 64, 0, pack.data());
    pfnl::conv1x1_c10_pack_weights(k10_host, T, pack.data() + n3);
    if (b1_host) std::memcpy(&pack[n3 + n1], b1_host, 64 * sizeof(float));
    if (b10_host) std::memcpy(&pack[n3 + n1 + 128], b10_host, 64 * sizeof(float));
    const size_t np1 = (size_t)clips * T * H * W, npb = (size_t)clips * H * W;
    uint16_t *dw = nullptr, *t1 = nullptr, *tb = nullptr, *ti = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    hipError_t e = hipMalloc(&t1, np1 * 256);
    if (e == hipSuccess) e = hipMalloc(&tb, npb * 256);
    if (e == hipSuccess && in_sf) e = hipMalloc(&ti, np1 * 256);
    if (e == hipSuccess) e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && in_sf) e = pfnl::launch_sf_from_f32(in, ti, np1, s);   // (the same split the chain kernel's epilogue applies: sf_split4)
    pfnl::ConvSplitParams q{in_sf ? reinterpret_cast<const float*>(ti) : in, dw, reinterpret_cast<const float*>(dw + n3 + n1), nullptr, nullptr, reinterpret_cast<float*>(t1), H, W, clips * T, T, 1};
    q.in_sf = in_sf ? 1 : 0;
    q.wpack2 = dw + n3;
    q.bias2 = reinterpret_cast<const float*>(dw + n3 + n1 + 128);
    q.out2 = reinterpret_cast<float*>(tb);
    if (e == hipSuccess) e = pfnl::launch_conv3x3_c1c10(q, s);
    if (e == hipSuccess) e = pfnl::launch_sf_to_f32(t1, out1, np1, s);
    if (e == hipSuccess) e = pfnl::launch_sf_to_f32(tb, base, npb, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    (void)hipFree(t1);
    (void)hipFree(tb);
    if (ti) (void)hipFree(ti);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv1+conv10 split16 op: ") + hipGetErrorString(e));
    return 0;
}
int pfnl_op_conv1_conv10_split16(const float* in, const float* k1_host, const float* b1_host, const float* k10_host,
                                 const float* b10_host, float* out1, float* base, int clips, int frames_per_clip, int H, int W,
                                 void* stream) {
    return op_conv1_conv10_split16(in, k1_host, b1_host, k10_host, b10_host, out1, base, clips, frames_per_clip, H, W, stream, false);
}
// ... with the input converted to the split format first and the halo taken from there by LDS-DMA (conv3x3_c1c10_kernel<true>, option
// split16_sf0): the same operands in the same order - bit-identical to pfnl_op_conv1_conv10_split16
int pfnl_op_conv1_conv10_split16_sf0(const float* in, const float* k1_host, const float* b1_host, const float* k10_host,
                                     const float* b10_host, float* out1, float* base, int clips, int frames_per_clip, int H, int W,
                                     void* stream) {
    return op_conv1_conv10_split16(in, k1_host, b1_host, k10_host, b10_host, out1, base, clips, frames_per_clip, H, W, stream, true);
}

// conv10_i with its input and / or output in the split format (fp32 at the hook's interface, see above)
int pfnl_op_conv1x1_split16_sf(const float* in, const float* kernel_host, const float* bias_host, float* out, int items,
                               int frames_per_item, int HW, int act, int in_sf, int out_sf, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (items < 1 || frames_per_item < 1 || HW < 1) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    hipStream_t s = (hipStream_t)stream;
    const int T = frames_per_item;
    const size_t nh = pfnl::conv1x1_split16_pack_halfs(T);
    std::vector<uint16_t> pack(nh + 128, 0);
    pfnl::conv1x1_split16_pack_weights(kernel_host, T, pack.data());
    if (bias_host) std::memcpy(&pack[nh], bias_host, 64 * sizeof(float));
    const size_t npin = (size_t)items * T * HW, npout = (size_t)items * HW;
    uint16_t *dw = nullptr, *tin = nullptr, *tout = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(uint16_t)));
    hipError_t e = hipMalloc(&tin, npin * 256);
    if (e == hipSuccess) e = hipMalloc(&tout, npout * 256);
    if (e == hipSuccess) e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && in_sf) e = pfnl::launch_sf_from_f32(in, tin, npin, s);
    if (e == hipSuccess)
        e = pfnl::launch_conv1x1_split16(in_sf ? reinterpret_cast<const float*>(tin) : in, dw, reinterpret_cast<const float*>(dw + nh),
                                         out_sf ? reinterpret_cast<float*>(tout) : out, items, T, HW, act, s, in_sf != 0, out_sf != 0);
    if (e == hipSuccess && out_sf) e = pfnl::launch_sf_to_f32(tout, out, npout, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    (void)hipFree(tin);
    (void)hipFree(tout);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv1x1 split16 SF op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_conv1x1_stream(const float* in, const float* kernel_host, const float* bias_host, float* out, int items,
                           int frames_per_item, int HW, int act, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (items < 1 || frames_per_item < 1 || HW < 1) return fail(PFNL_ERR_INVALID, "unsupported conv geometry");
    hipStream_t s = (hipStream_t)stream;
    const int T = frames_per_item;
    std::vector<float> pack(pfnl::conv1x1_pack_floats(T) + 64, 0.f);
    pfnl::conv1x1_pack_weights(kernel_host, T, pack.data());
    const size_t boff = pack.size() - 64;
    if (bias_host) std::memcpy(&pack[boff], bias_host, 64 * sizeof(float));
    float* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(float)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pfnl::launch_conv1x1_stream(in, dw, dw + boff, out, items, T, HW, act, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("conv1x1 op: ") + hipGetErrorString(e));
    return 0;
}

static int op_conv3x3_wino(bool ws, const float* in, const float* kernel_host, const float* bias_host,
                           const float* addend, int add_div, const float* resid, float* out, int items, int H,
                           int W, int act, void* stream) {
    if (!in || !kernel_host || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (items < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return fail(PFNL_ERR_INVALID, "winograd conv needs even H, W");
    if ((addend != nullptr) != (resid != nullptr))
        return fail(PFNL_ERR_INVALID, "addend and resid must be given together or not at all");
    if (addend && add_div < 1) return fail(PFNL_ERR_INVALID, "add_div must be >= 1");
    hipStream_t s = (hipStream_t)stream;
    std::vector<float> pack(pfnl::wino_pack_floats() + 64, 0.f);
    pfnl::wino_pack_weights(kernel_host, 64, 0, pack.data());
    const size_t boff = pack.size() - 64;
    if (bias_host) std::memcpy(&pack[boff], bias_host, 64 * sizeof(float));
    float* dw = nullptr;
    HIPCHK(hipMalloc(&dw, pack.size() * sizeof(float)));
    hipError_t e = hipMemcpy(dw, pack.data(), pack.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        pfnl::WinoParams wp{in, dw, dw + boff, addend, resid, out, H, W, addend ? add_div : 1, act, items, nullptr};
#ifdef PFNL_WINO_TIMING
        long long* dbg = nullptr;
        const size_t dbg_n = 4096 * 64;
        if (hipMalloc(&dbg, dbg_n * sizeof(long long)) == hipSuccess) {
            (void)hipMemset(dbg, 0, dbg_n * sizeof(long long));
            wp.dbg = dbg;
        }
#endif
        e = ws ? pfnl::launch_conv_wino_ws(wp, s) : pfnl::launch_conv_wino(wp, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
#ifdef PFNL_WINO_TIMING
        if (dbg) {
            std::vector<long long> hst(dbg_n);
            (void)hipMemcpy(hst.data(), dbg, dbg_n * sizeof(long long), hipMemcpyDeviceToHost);
            if (ws) {
                for (int b : {0, 8, 101, 200, 255})
                    for (int role = 0; role < 2; ++role) {
                        const long long* t = &hst[(size_t)b * 128 + role * 64];
                        std::fprintf(stderr, "WS_TIMING wg %d %s:", b, role ? "helper" : "matrix");
                        for (int i = 1; i < 64 && t[i]; ++i) std::fprintf(stderr, " %lld", t[i] - t[i - 1]);
                        std::fprintf(stderr, " | t0-t0[wg0] %lld\n", t[0] - hst[0]);
                    }
            } else
            for (int b : {0, 8, 16, 1024, 2048, 4096, 7000}) {
                std::fprintf(stderr, "WINO_TIMING wg %d:", b);
                for (int i = 1; i < 16 && hst[(size_t)b * 16 + i]; ++i)
                    std::fprintf(stderr, " %lld", hst[(size_t)b * 16 + i] - hst[(size_t)b * 16]);
                std::fprintf(stderr, " | t0-t0[wg0] %lld\n", hst[(size_t)b * 16] - hst[0]);
            }
            (void)hipFree(dbg);
        }
#endif
    }
    (void)hipFree(dw);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("winograd conv op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_conv3x3_winograd(const float* in, const float* kernel_host, const float* bias_host,
                             const float* addend, int add_div, const float* resid, float* out, int items, int H,
                             int W, int act, void* stream) {
    return op_conv3x3_wino(false, in, kernel_host, bias_host, addend, add_div, resid, out, items, H, W, act, stream);
}

int pfnl_op_conv3x3_winograd_ws(const float* in, const float* kernel_host, const float* bias_host,
                                const float* addend, int add_div, const float* resid, float* out, int items, int H,
                                int W, int act, void* stream) {
    return op_conv3x3_wino(true, in, kernel_host, bias_host, addend, add_div, resid, out, items, H, W, act, stream);
}


static int op_nonlocal(int bf16 /* 0 f32, 2 f16 split, 3 f16 (hi parts only) */, const float* x, const float* wg, const float* bg, const float* ww, const float* bw,
                       float* out, int B, int T, int H, int W, void* stream) {
    if (!x || !wg || !bg || !ww || !bw || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if ((T != 3 && T != 5 && T != 7) || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1))
        return fail(PFNL_ERR_INVALID, "unsupported non-local geometry");
    hipStream_t s = (hipStream_t)stream;
    const int C = 12 * T, CP = pfnl::nl_padded_ch(C), N = (H / 2) * (W / 2);
    std::vector<float> blob((size_t)CP * CP + CP, 0.f);
    for (int ci = 0; ci < C; ++ci)
        for (int co = 0; co < C; ++co) {
            double acc = 0.0;
            for (int cm = 0; cm < C; ++cm) acc += (double)wg[(size_t)ci * C + cm] * (double)ww[(size_t)cm * C + co];
            blob[(size_t)ci * CP + co] = (float)acc;
        }
    for (int co = 0; co < C; ++co) {
        double acc = bw[co];
        for (int cm = 0; cm < C; ++cm) acc += (double)bg[cm] * (double)ww[(size_t)cm * C + co];
        blob[(size_t)CP * CP + co] = (float)acc;
    }
    float* d = nullptr;
    const size_t nX = (size_t)B * N * CP;
    const size_t nP = pfnl::nl_partial_floats(B, N, C);
    const size_t n16 = bf16 ? (pfnl::nl_f16_scratch_halfs(B, N) + 1) / 2 : 0;   // in floats
    HIPCHK(hipMalloc(&d, (blob.size() + 2 * nX + nP + n16 + 64) * sizeof(float)));
    float* dX = d + blob.size();
    float* dXo = dX + nX;
    float* dP = nP ? dXo + nX : nullptr;
    uint16_t* d16 = reinterpret_cast<uint16_t*>(d + (blob.size() + 2 * nX + nP + 63) / 64 * 64);
    hipError_t e = hipMemcpy(d, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pfnl::launch_nl_pack(x, dX, B, T, H, W, s);
    if (e == hipSuccess)
        e = bf16 >= 2 ? pfnl::launch_nl_attn_f16(dX, dXo, d, d + (size_t)CP * CP, dP, d16, B, N, C, s, 0, -1, bf16 == 2)
                    : pfnl::launch_nl_attn(dX, dXo, d, d + (size_t)CP * CP, dP, B, N, C, s);
    if (e == hipSuccess) e = pfnl::launch_nl_unpack(dXo, out, B, T, H, W, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    hipFree(d);
    if (e != hipSuccess) return fail(PFNL_ERR_HIP, std::string("nonlocal op: ") + hipGetErrorString(e));
    return 0;
}

int pfnl_op_nonlocal(const float* x, const float* wg, const float* bg, const float* ww, const float* bw,
                     float* out, int B, int T, int H, int W, void* stream) {
    return op_nonlocal(0, x, wg, bg, ww, bw, out, B, T, H, W, stream);
}

int pfnl_op_nonlocal_split16(const float* x, const float* wg, const float* bg, const float* ww, const float* bw,
                             float* out, int B, int T, int H, int W, void* stream) {
    return op_nonlocal(2, x, wg, bg, ww, bw, out, B, T, H, W, stream);
}

int pfnl_op_nonlocal_f16(const float* x, const float* wg, const float* bg, const float* ww, const float* bw,
                         float* out, int B, int T, int H, int W, void* stream) {
    return op_nonlocal(3, x, wg, bg, ww, bw, out, B, T, H, W, stream);
}

int pfnl_op_bicubic(const float* x, float* out, int B, int H, int W, int scale, void* stream) {
    if (!x || !out) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (B < 1 || H < 1 || W < 1 || (scale != 2 && scale != 4)) return fail(PFNL_ERR_INVALID, "bad bicubic geometry");
    HIPCHK(pfnl::launch_bicubic(x, out, B, H, W, scale, (hipStream_t)stream));
    return 0;
}

int pfnl_op_blur_decimate(const float* hr, float* lr, int F, int H, int W, int scale, void* stream) {
    if (!hr || !lr) return fail(PFNL_ERR_INVALID, "NULL argument");
    if (F < 1 || H < 7 || W < 7 || (scale != 2 && scale != 4))
        return fail(PFNL_ERR_INVALID, "blur_decimate needs H, W >= 7 (reflect pad 6) and scale 2 or 4");
    HIPCHK(pfnl::launch_blur_decimate(hr, lr, F, H, W, scale, (hipStream_t)stream));
    return 0;
}

int pfnl_selftest_mfma(int device_id) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(PFNL_ERR_NODEVICE, "no HIP device visible");
    HIPCHK(hipSetDevice(device_id));
    int bad = -1;
    HIPCHK(pfnl::run_mfma_selftest(&bad));
    if (bad != 0) return fail(PFNL_ERR_STATE, "MFMA fragment layout mismatch: " + std::to_string(bad) + " elements");
    return 0;
}

}  // extern "C"

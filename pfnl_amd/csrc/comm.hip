// pfnl_comm_* (include/pfnl_hip.h): the multi-GPU exchanges of the path, on RCCL directly (SURVEY.md section 8(b),(e)).
//
// PFNL's forward has no cross-clip op (reference model/pfnl.py:44,55), so clips shard over ranks with NO collective on
// the data path.  What is exchanged: (1) the packed device weight blobs, once, root -> all (ncclBroadcast, 12.6 MB fp32
// + 4.6 MB bf16); (2) a few doubles (squared error, frame count: ncclSum; elapsed time: ncclMax); (3) optionally the
// SR frames to one rank (ncclAllGather of equal shards).  One communicator per process (ncclCommInitRank from an id the
// caller moves between its processes through any host channel) or one per device of a single process (ncclCommInitAll).
//
// RCCL is resolved at run time (dlopen, preferring an already loaded librccl - torch's - so that both share one HIP
// runtime): libpfnl_hip.so itself does not link it and single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pfnl_hip.h"
#include "capi_internal.h"
#include "common.h"

static_assert(PFNL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size is part of the ABI");

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;            // optional: pfnl_comm_rank asks the communicator itself when present
    decltype(&ncclCommCount) CommCount = nullptr;
    std::string err;
};

static void rccl_resolve(Rccl& r);
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;                                     // (callers may be threads of one process, one per device)
    std::call_once(once, [] { rccl_resolve(r); });
    return &r;
}
static void rccl_resolve(Rccl& r) {
    std::vector<std::string> names;
    if (const char* e = std::getenv("PFNL_RCCL_LIB")) names.push_back(e);
    for (const char* n : {"librccl.so", "librccl.so.1"}) {          // one that is already in the process (torch's)
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    }
    for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) names.push_back(n);
    for (size_t i = 0; !r.lib && i < names.size(); ++i) r.lib = dlopen(names[i].c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) {
        const char* m = dlerror();                                  // (one call: dlerror() clears the message it returns)
        r.err = std::string("RCCL not found (set PFNL_RCCL_LIB): ") + (m ? m : "?");
        return;
    }
#define PFNL_SYM(field, name)                                                       \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));               \
    if (!r.field) r.err = std::string("RCCL symbol missing: ") + name;
    PFNL_SYM(GetUniqueId, "ncclGetUniqueId")
    PFNL_SYM(CommInitRank, "ncclCommInitRank")
    PFNL_SYM(CommInitAll, "ncclCommInitAll")
    PFNL_SYM(CommDestroy, "ncclCommDestroy")
    PFNL_SYM(Broadcast, "ncclBroadcast")
    PFNL_SYM(AllReduce, "ncclAllReduce")
    PFNL_SYM(AllGather, "ncclAllGather")
    PFNL_SYM(GetErrorString, "ncclGetErrorString")
#undef PFNL_SYM
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.lib, "ncclCommUserRank"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
}

}  // namespace

struct pfnl_comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    double* dscratch = nullptr;   // device scratch for the scalar reductions
    size_t dscratch_n = 0;
};

#define COMM_FAIL(code, msg) return pfnl_internal_fail(code, msg)
#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) COMM_FAIL(PFNL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
#define NCHK(expr)                                                                                             \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess) COMM_FAIL(PFNL_ERR_COMM, std::string(#expr) + ": " + R->GetErrorString(_r));    \
    } while (0)
#define NEED_RCCL()                                                            \
    Rccl* R = rccl();                                                          \
    if (!R->lib || !R->err.empty()) COMM_FAIL(PFNL_ERR_COMM, R->err)

static int finish_init(pfnl_comm* c) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->dscratch_n = 64;
    HIPCHK(hipMalloc(&c->dscratch, c->dscratch_n * sizeof(double)));
    return 0;
}

extern "C" {

int pfnl_comm_get_unique_id(void* id) {
    if (!id) COMM_FAIL(PFNL_ERR_INVALID, "NULL argument");
    NEED_RCCL();
    ncclUniqueId u;
    NCHK(R->GetUniqueId(&u));
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int pfnl_comm_init_rank(int nranks, int rank, const void* id, int device_id, pfnl_comm** out) {
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) COMM_FAIL(PFNL_ERR_INVALID, "bad communicator arguments");
    NEED_RCCL();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) COMM_FAIL(PFNL_ERR_NODEVICE, "no HIP device visible");
    if (device_id < 0 || device_id >= ndev) COMM_FAIL(PFNL_ERR_INVALID, "bad device_id");
    HIPCHK(hipSetDevice(device_id));
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    pfnl_comm* c = new pfnl_comm();
    c->nranks = nranks;
    c->rank = rank;
    c->device = device_id;
    ncclResult_t r = R->CommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        delete c;
        COMM_FAIL(PFNL_ERR_COMM, std::string("ncclCommInitRank: ") + R->GetErrorString(r));
    }
    if (int e = finish_init(c)) {
        pfnl_comm_destroy(c);
        return e;
    }
    *out = c;
    return 0;
}

int pfnl_comm_init_all(int ndev, const int* devs, pfnl_comm** comms) {
    if (ndev < 1 || !comms) COMM_FAIL(PFNL_ERR_INVALID, "bad communicator arguments");
    NEED_RCCL();
    std::vector<ncclComm_t> cs(ndev);
    std::vector<int> dl(ndev);
    for (int i = 0; i < ndev; ++i) dl[i] = devs ? devs[i] : i;
    NCHK(R->CommInitAll(cs.data(), ndev, dl.data()));
    for (int i = 0; i < ndev; ++i) {
        pfnl_comm* c = new pfnl_comm();
        c->comm = cs[i];
        c->nranks = ndev;
        c->rank = i;
        c->device = dl[i];
        comms[i] = c;
        if (int e = finish_init(c)) return e;
    }
    return 0;
}

int pfnl_comm_destroy(pfnl_comm* c) {
    if (!c) return 0;
    Rccl* R = rccl();
    hipSetDevice(c->device);
    if (c->stream) {
        hipStreamSynchronize(c->stream);
        hipStreamDestroy(c->stream);
    }
    if (c->dscratch) hipFree(c->dscratch);
    if (c->comm && R->CommDestroy) R->CommDestroy(c->comm);
    delete c;
    return 0;
}

int pfnl_comm_rank(pfnl_comm* c, int* rank, int* nranks) {
    if (!c || !rank || !nranks) COMM_FAIL(PFNL_ERR_INVALID, "NULL argument");
    *rank = c->rank;
    *nranks = c->nranks;
    // what RCCL itself says about this communicator (ncclCommUserRank / ncclCommCount): a bench line that quotes these figures shows that
    // N ranks really joined the communicator, not what the launcher's environment claimed
    Rccl* R = rccl();
    if (c->comm && R->lib && R->CommUserRank && R->CommCount) {
        int r = -1, n = -1;
        NCHK(R->CommUserRank(c->comm, &r));
        NCHK(R->CommCount(c->comm, &n));
        if (r != c->rank || n != c->nranks) COMM_FAIL(PFNL_ERR_COMM, "RCCL reports another rank / size than the communicator was created with");
        *rank = r;
        *nranks = n;
    }
    return 0;
}

int pfnl_comm_bcast(pfnl_comm* c, void* dev_buf, size_t bytes, int root) {
    if (!c || (!dev_buf && bytes) || root < 0 || root >= c->nranks) COMM_FAIL(PFNL_ERR_INVALID, "bad broadcast arguments");
    NEED_RCCL();
    HIPCHK(hipSetDevice(c->device));
    if (bytes) NCHK(R->Broadcast(dev_buf, dev_buf, bytes, ncclChar, root, c->comm, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int pfnl_comm_allreduce_f64(pfnl_comm* c, double* vals, int n, int op) {
    if (!c || !vals || n < 1 || n > (int)c->dscratch_n || (op != PFNL_COMM_SUM && op != PFNL_COMM_MAX))
        COMM_FAIL(PFNL_ERR_INVALID, "bad all-reduce arguments (n <= 64, op sum or max)");
    NEED_RCCL();
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->dscratch, vals, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCHK(R->AllReduce(c->dscratch, c->dscratch, n, ncclDouble, op == PFNL_COMM_SUM ? ncclSum : ncclMax, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(vals, c->dscratch, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int pfnl_comm_barrier(pfnl_comm* c) {
    double one = 1.0;
    return pfnl_comm_allreduce_f64(c, &one, 1, PFNL_COMM_SUM);
}

int pfnl_comm_allgather(pfnl_comm* c, const float* send_dev, float* recv_dev, size_t count_per_rank, void* stream) {
    if (!c || !send_dev || !recv_dev || !count_per_rank) COMM_FAIL(PFNL_ERR_INVALID, "bad all-gather arguments");
    NEED_RCCL();
    HIPCHK(hipSetDevice(c->device));
    NCHK(R->AllGather(send_dev, recv_dev, count_per_rank, ncclFloat, c->comm, stream ? (hipStream_t)stream : c->stream));
    if (!stream) HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"

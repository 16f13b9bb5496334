// Data-parallel exchange behind the C-ABI: emo_comm_{unique_id,init,allreduce,broadcast,destroy}.
//
// The reference has no distributed code; the build adds ONE exchange per optimizer step — a sum all-reduce of the flat fp32
// gradient buffer between backward() and the clip (reference insertion point: stage2_accompaniment/train.py:76-81) — plus an
// initial broadcast of the parameters / FAVOR+ omega.  RCCL (ncclAllReduce / ncclBroadcast over xGMI) is bound at RUN time with
// dlopen so that libemo_hip.so neither links RCCL nor needs it on a CPU-only box; the copy already mapped into the process
// (PyTorch-ROCm ships one) is preferred so that exactly one RCCL / HIP runtime pair is live.  One communicator per process
// (one process per GPU); every call is asynchronous on the caller's stream, in place, on caller-owned device memory.
#include <dlfcn.h>

#include "emo_common.h"

namespace {
typedef struct { char internal[128]; } nccl_uid_t;        // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm_t;
enum { NCCL_SUM = 0, NCCL_INT64 = 4, NCCL_F32 = 7, NCCL_BF16 = 9 };   // rccl.h: ncclRedOp_t / ncclDataType_t values

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_uid_t*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_uid_t, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    nccl_comm_t comm = nullptr;
    int rank = 0, world = 1;
} g;

bool bind_rccl() {
    if (g.handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;      // the copy the process already uses (torch's)
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        emo_set_error("emo_comm: cannot dlopen librccl.so.1 (%s)", dlerror());
        return false;
    }
#define SYM(field, name)                                              \
    *(void**)(&g.field) = dlsym(h, name);                             \
    if (!g.field) {                                                   \
        emo_set_error("emo_comm: RCCL symbol %s missing", name);      \
        return false;                                                 \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(AllReduce, "ncclAllReduce")
    SYM(Broadcast, "ncclBroadcast")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g.handle = h;
    return true;
}

int nccl_dtype(int dtype) { return dtype == EMO_F32 ? NCCL_F32 : dtype == EMO_BF16 ? NCCL_BF16 : dtype == EMO_I64 ? NCCL_INT64 : -1; }

#define RCCL_CALL(expr, what)                                                              \
    do {                                                                                   \
        int r__ = (expr);                                                                  \
        if (r__ != 0) {                                                                    \
            emo_set_error("emo_comm: %s failed: %s", what, g.GetErrorString(r__));         \
            return EMO_ERR_LAUNCH;                                                         \
        }                                                                                  \
    } while (0)
}  // namespace

extern "C" int emo_comm_bind(void) { return bind_rccl() ? EMO_OK : EMO_ERR_UNSUPPORTED; }

extern "C" int emo_comm_unique_id(void* id128) {
    EMO_CHECK(id128, "emo_comm_unique_id: null pointer");
    if (!bind_rccl()) return EMO_ERR_UNSUPPORTED;
    nccl_uid_t id;
    RCCL_CALL(g.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id128, &id, sizeof(id));
    return EMO_OK;
}

extern "C" int emo_comm_init(const void* id128, int rank, int world) {
    EMO_CHECK(id128 && world >= 1 && rank >= 0 && rank < world, "emo_comm_init: bad rank %d / world %d", rank, world);
    EMO_CHECK(g.comm == nullptr, "emo_comm_init: communicator already initialised (call emo_comm_destroy first)");
    if (!bind_rccl()) return EMO_ERR_UNSUPPORTED;
    nccl_uid_t id;
    memcpy(&id, id128, sizeof(id));
    RCCL_CALL(g.CommInitRank(&g.comm, world, id, rank), "ncclCommInitRank");      // binds the CURRENT HIP device
    g.rank = rank;
    g.world = world;
    return EMO_OK;
}

extern "C" int emo_comm_world(void) { return g.comm ? g.world : 0; }
extern "C" int emo_comm_rank(void) { return g.comm ? g.rank : -1; }

extern "C" int emo_comm_allreduce(void* buf, int64_t count, int dtype, emo_stream_t stream) {
    EMO_CHECK(g.comm, "emo_comm_allreduce: emo_comm_init has not been called");
    EMO_CHECK(buf && count > 0 && nccl_dtype(dtype) >= 0, "emo_comm_allreduce: bad args");
    RCCL_CALL(g.AllReduce(buf, buf, (size_t)count, nccl_dtype(dtype), NCCL_SUM, g.comm, (hipStream_t)stream), "ncclAllReduce");
    return EMO_OK;
}

extern "C" int emo_comm_broadcast(void* buf, int64_t count, int dtype, int root, emo_stream_t stream) {
    EMO_CHECK(g.comm, "emo_comm_broadcast: emo_comm_init has not been called");
    EMO_CHECK(buf && count > 0 && nccl_dtype(dtype) >= 0 && root >= 0 && root < g.world, "emo_comm_broadcast: bad args");
    RCCL_CALL(g.Broadcast(buf, buf, (size_t)count, nccl_dtype(dtype), root, g.comm, (hipStream_t)stream), "ncclBroadcast");
    return EMO_OK;
}

extern "C" int emo_comm_destroy(void) {
    if (g.comm) {
        RCCL_CALL(g.CommDestroy(g.comm), "ncclCommDestroy");
        g.comm = nullptr;
        g.world = 1;
        g.rank = 0;
    }
    return EMO_OK;
}

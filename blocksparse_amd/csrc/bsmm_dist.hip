// bsmm_dist.hip -- the dw all-reduce of the data-parallel path over RCCL (include/bsmm_dist.h).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>

#include "bsmm.h"
#include "bsmm_dist.h"

namespace {

// the five RCCL entry points this path needs (rccl.h: ncclResult_t = int, 0 = success; ncclSum = 0;
// ncclFloat16 = 6, ncclFloat32 = 7, ncclBfloat16 = 9)
struct Rccl {
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, const void*, int) = nullptr;      // NOTE: real signature takes ncclUniqueId BY VALUE (128 bytes)
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    void* lib = nullptr;
    bool ok = false;
};

struct UniqueId { char internal[BSMM_DIST_ID_BYTES]; };
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<int (*)(void**, int, const void*, int)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(r.lib, "ncclAllReduce"));
        r.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(r.lib, "ncclCommDestroy"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
    });
    return r;
}

inline int nccl_type(int dtype) { return dtype == BSMM_F32 ? 7 : (dtype == BSMM_F16 ? 6 : (dtype == BSMM_BF16 ? 9 : -1)); }

}  // namespace

struct bsmm_dist {
    void* comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t produced = nullptr, reduced = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" {

int bsmm_dist_unique_id(void* id_out) {
    if (!id_out) return BSMM_ERR_ARG;
    Rccl& r = rccl();
    if (!r.ok) return BSMM_ERR_UNSUPPORTED;
    return r.GetUniqueId(id_out) == 0 ? BSMM_OK : BSMM_ERR_ARG;
}

int bsmm_dist_create(bsmm_dist** out, const void* id, int32_t rank, int32_t world, int32_t device) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return BSMM_ERR_ARG;
    Rccl& r = rccl();
    if (!r.ok) return BSMM_ERR_UNSUPPORTED;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    bsmm_dist* h = new bsmm_dist;
    h->rank = rank; h->world = world; h->device = device;
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->produced, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->reduced, hipEventDisableTiming)) != hipSuccess) {
        bsmm_dist_destroy(h);
        return (int)e;
    }
    UniqueId uid;
    __builtin_memcpy(uid.internal, id, BSMM_DIST_ID_BYTES);
    if (reinterpret_cast<CommInitRankFn>(r.CommInitRank)(&h->comm, world, uid, rank) != 0) {
        bsmm_dist_destroy(h);
        return BSMM_ERR_ARG;
    }
    *out = h;
    return BSMM_OK;
}

int bsmm_dist_allreduce_begin(bsmm_dist* h, void* buf, size_t count, int32_t dtype, void* producer_stream) {
    if (!h || !h->comm || !buf || count == 0 || nccl_type(dtype) < 0) return BSMM_ERR_ARG;
    hipError_t e = hipEventRecord(h->produced, static_cast<hipStream_t>(producer_stream));       // (src/nccl_op.cc:513)
    if (e != hipSuccess) return (int)e;
    if ((e = hipStreamWaitEvent(h->stream, h->produced, 0)) != hipSuccess) return (int)e;      // (src/nccl_op.cc:168)
    if (rccl().AllReduce(buf, buf, count, nccl_type(dtype), /*ncclSum*/ 0, h->comm, h->stream) != 0) return BSMM_ERR_ARG;
    e = hipEventRecord(h->reduced, h->stream);
    return (int)e;
}

int bsmm_dist_allreduce_end(bsmm_dist* h, void* consumer_stream) {
    if (!h) return BSMM_ERR_ARG;
    return (int)hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), h->reduced, 0);
}

void* bsmm_dist_stream(bsmm_dist* h) { return h ? h->stream : nullptr; }
int bsmm_dist_world(const bsmm_dist* h) { return h ? h->world : 0; }

int bsmm_dist_destroy(bsmm_dist* h) {
    if (!h) return BSMM_OK;
    if (h->comm) rccl().CommDestroy(h->comm);
    if (h->produced) (void)hipEventDestroy(h->produced);
    if (h->reduced) (void)hipEventDestroy(h->reduced);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return BSMM_OK;
}

}  // extern "C"

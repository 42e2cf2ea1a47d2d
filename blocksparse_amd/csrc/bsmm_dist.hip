// bsmm_dist.hip -- the dw all-reduce of the data-parallel path over RCCL (include/bsmm_dist.h).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>

#include "bsmm.h"
#include "bsmm_dist.h"

namespace {

// the RCCL entry points this path needs (rccl.h: ncclResult_t = int, 0 = success; ncclSum = 0;
// ncclFloat16 = 6, ncclFloat32 = 7, ncclBfloat16 = 9)
struct Rccl {
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, const void*, int) = nullptr;      // NOTE: real signature takes ncclUniqueId BY VALUE (128 bytes)
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
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
        r.ReduceScatter = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(r.lib, "ncclReduceScatter"));
        r.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(r.lib, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(r.lib, "ncclCommDestroy"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.ReduceScatter && r.AllGather && r.CommDestroy;
    });
    return r;
}

inline int nccl_type(int dtype) { return dtype == BSMM_F32 ? 7 : (dtype == BSMM_F16 ? 6 : (dtype == BSMM_BF16 ? 9 : -1)); }
inline size_t dw_elem_size(int dtype) { return dtype == BSMM_F32 ? 4 : 2; }

// out[i] = alpha * [gate[(i0 + i) / bsq] *] sums[i] + beta * dw_old[i], rounded once, for the elements [i0, i0 + n) of DW that this
// rank owns after the reduce-scatter (the finalize of bsmm_updat_finalize, restricted to a shard).  T16: 0 fp32, 1 fp16, 2 bf16.
template <int T16>
__global__ void __launch_bounds__(256)
dw_shard_finalize_kernel(const float* __restrict__ sums, const void* __restrict__ dw_old, void* __restrict__ out, size_t n, size_t i0, int bsq,
                         float alpha, float beta, const float* __restrict__ gate) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (i + e >= n) break;
        const float a = gate ? alpha * gate[(i0 + i + e) / bsq] : alpha;
        float v = a * sums[i + e];
        if constexpr (T16 == 0) {
            if (beta != 0.f) v += beta * static_cast<const float*>(dw_old)[i + e];
            static_cast<float*>(out)[i + e] = v;
        } else if constexpr (T16 == 1) {
            if (beta != 0.f) v += beta * (float)static_cast<const _Float16*>(dw_old)[i + e];
            static_cast<_Float16*>(out)[i + e] = (_Float16)v;
        } else {
            if (beta != 0.f) v += beta * (float)static_cast<const __bf16*>(dw_old)[i + e];
            static_cast<__bf16*>(out)[i + e] = (__bf16)v;
        }
    }
}

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

size_t bsmm_dist_dw_shard_elems(int32_t world, int32_t blocks, int32_t bsize) {
    if (world < 1 || blocks <= 0 || bsize <= 0) return 0;
    const size_t total = (size_t)blocks * bsize * bsize;
    return ((total + world - 1) / world + 7) & ~(size_t)7;          // 16-byte aligned shards for every dtype
}

int bsmm_dist_dw_begin(bsmm_dist* h, float* sums, void* dw, void* staging, const float* gate, int32_t blocks, int32_t bsize, int32_t dtype,
                       float alpha, float beta, void* producer_stream) {
    if (!h || !h->comm || !sums || !dw || !staging || blocks <= 0 || nccl_type(dtype) < 0) return BSMM_ERR_ARG;
    if (bsize != 8 && bsize != 16 && bsize != 32) return BSMM_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(sums) & 15) || (reinterpret_cast<uintptr_t>(staging) & 15)) return BSMM_ERR_ARG;
    const size_t total = (size_t)blocks * bsize * bsize, shard = bsmm_dist_dw_shard_elems(h->world, blocks, bsize);
    const size_t lo = std::min(total, (size_t)h->rank * shard), hi = std::min(total, lo + shard), es = dw_elem_size(dtype);
    hipError_t e = hipEventRecord(h->produced, static_cast<hipStream_t>(producer_stream));
    if (e != hipSuccess) return (int)e;
    if ((e = hipStreamWaitEvent(h->stream, h->produced, 0)) != hipSuccess) return (int)e;
    Rccl& r = rccl();
    // 1. every rank receives the cross-rank fp32 sum of ITS shard (in place: recv = send + rank * shard)
    if (r.ReduceScatter(sums, sums + (size_t)h->rank * shard, shard, /*ncclFloat32*/ 7, /*ncclSum*/ 0, h->comm, h->stream) != 0) return BSMM_ERR_ARG;
    // 2. alpha / beta / gate and the ONE rounding, on 1 / world of the elements
    if (hi > lo) {
        const size_t n = hi - lo;
        const unsigned grid = (unsigned)((n / 4 + 255) / 256 + 1);
        const void* old = static_cast<const char*>(dw) + lo * es;
        void* out = static_cast<char*>(staging) + lo * es;
        const int bsq = bsize * bsize;
        if (dtype == BSMM_F32)      dw_shard_finalize_kernel<0><<<grid, 256, 0, h->stream>>>(sums + lo, old, out, n, lo, bsq, alpha, beta, gate);
        else if (dtype == BSMM_F16) dw_shard_finalize_kernel<1><<<grid, 256, 0, h->stream>>>(sums + lo, old, out, n, lo, bsq, alpha, beta, gate);
        else                        dw_shard_finalize_kernel<2><<<grid, 256, 0, h->stream>>>(sums + lo, old, out, n, lo, bsq, alpha, beta, gate);
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    }
    // 3. the finished shards travel in the storage type (half the bytes of the sums), in place in the staging buffer
    if (r.AllGather(static_cast<const char*>(staging) + (size_t)h->rank * shard * es, staging, shard, nccl_type(dtype), h->comm, h->stream) != 0) return BSMM_ERR_ARG;
    if ((e = hipMemcpyAsync(dw, staging, total * es, hipMemcpyDeviceToDevice, h->stream)) != hipSuccess) return (int)e;
    e = hipEventRecord(h->reduced, h->stream);
    return (int)e;
}

int bsmm_dist_dw_end(bsmm_dist* h, void* consumer_stream) { return bsmm_dist_allreduce_end(h, consumer_stream); }

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

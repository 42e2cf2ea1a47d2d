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

int bsmm_dist_dw_layout(int32_t world, int32_t rank, int32_t blocks, int32_t bsize, size_t* shard, size_t* lo, size_t* hi, size_t* capacity) {
    if (world < 1 || rank < 0 || rank >= world || blocks <= 0 || bsize <= 0) return BSMM_ERR_ARG;
    const size_t total = (size_t)blocks * bsize * bsize, sh = bsmm_dist_dw_shard_elems(world, blocks, bsize);
    const size_t l = std::min(total, (size_t)rank * sh), h = std::min(total, l + sh);
    if (shard) *shard = sh;
    if (lo) *lo = l;
    if (hi) *hi = h;
    if (capacity) *capacity = (size_t)world * sh;
    return BSMM_OK;
}

}  // extern "C"

namespace {
// step 2 of the fused reduction: alpha / beta / gate and the ONE rounding on the elements [lo, hi) of DW, from `sums_shard` (the fp32
// sums of exactly those elements) into staging + lo.  Shared by the RCCL path and the single-device emulation.
int dw_finalize_shard(const float* sums_shard, const void* dw, void* staging, const float* gate, size_t lo, size_t hi, int bsize, int dtype,
                      float alpha, float beta, hipStream_t st) {
    if (hi <= lo) return 0;
    const size_t n = hi - lo, es = dw_elem_size(dtype);
    const unsigned grid = (unsigned)((n / 4 + 255) / 256 + 1);
    const void* old = static_cast<const char*>(dw) + lo * es;
    void* out = static_cast<char*>(staging) + lo * es;
    const int bsq = bsize * bsize;
    if (dtype == BSMM_F32)      dw_shard_finalize_kernel<0><<<grid, 256, 0, st>>>(sums_shard, old, out, n, lo, bsq, alpha, beta, gate);
    else if (dtype == BSMM_F16) dw_shard_finalize_kernel<1><<<grid, 256, 0, st>>>(sums_shard, old, out, n, lo, bsq, alpha, beta, gate);
    else                        dw_shard_finalize_kernel<2><<<grid, 256, 0, st>>>(sums_shard, old, out, n, lo, bsq, alpha, beta, gate);
    return (int)hipGetLastError();
}

// the emulation's stand-in for ncclReduceScatter(sum, fp32): dst[i] = sum over q of src[q][i], i < n (dst may be src[r])
struct PtrList16 { const float* p[16]; };
__global__ void __launch_bounds__(256) emu_sum_kernel(PtrList16 src, int world, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int q = 0; q < world; ++q) v += src.p[q][i];
    dst[i] = v;
}
}  // namespace

extern "C" {

int bsmm_dist_dw_begin(bsmm_dist* h, float* sums, size_t sums_capacity, void* dw, void* staging, const float* gate, int32_t blocks, int32_t bsize,
                       int32_t dtype, float alpha, float beta, void* producer_stream) {
    if (!h || !h->comm || !sums || !dw || !staging || blocks <= 0 || nccl_type(dtype) < 0) return BSMM_ERR_ARG;
    if (bsize != 8 && bsize != 16 && bsize != 32) return BSMM_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(sums) & 15) || (reinterpret_cast<uintptr_t>(staging) & 15)) return BSMM_ERR_ARG;
    size_t shard, lo, hi, need;
    if (int rc = bsmm_dist_dw_layout(h->world, h->rank, blocks, bsize, &shard, &lo, &hi, &need)) return rc;
    if (sums_capacity < need) return BSMM_ERR_WORKSPACE;      // the reduce-scatter reads and writes world * shard floats of `sums`
    const size_t total = (size_t)blocks * bsize * bsize, es = dw_elem_size(dtype);
    hipError_t e = hipEventRecord(h->produced, static_cast<hipStream_t>(producer_stream));
    if (e != hipSuccess) return (int)e;
    if ((e = hipStreamWaitEvent(h->stream, h->produced, 0)) != hipSuccess) return (int)e;
    Rccl& r = rccl();
    // 1. every rank receives the cross-rank fp32 sum of ITS shard (in place: recv = send + rank * shard)
    if (r.ReduceScatter(sums, sums + (size_t)h->rank * shard, shard, /*ncclFloat32*/ 7, /*ncclSum*/ 0, h->comm, h->stream) != 0) return BSMM_ERR_ARG;
    // 2. alpha / beta / gate and the ONE rounding, on 1 / world of the elements
    if (int rc = dw_finalize_shard(sums + lo, dw, staging, gate, lo, hi, bsize, dtype, alpha, beta, h->stream)) return rc;
    // 3. the finished shards travel in the storage type (half the bytes of the sums), in place in the staging buffer
    if (r.AllGather(static_cast<const char*>(staging) + (size_t)h->rank * shard * es, staging, shard, nccl_type(dtype), h->comm, h->stream) != 0) return BSMM_ERR_ARG;
    if ((e = hipMemcpyAsync(dw, staging, total * es, hipMemcpyDeviceToDevice, h->stream)) != hipSuccess) return (int)e;
    e = hipEventRecord(h->reduced, h->stream);
    return (int)e;
}

int bsmm_dist_dw_emulate(int32_t world, float* const* sums, size_t sums_capacity, void* const* dw, void* const* staging, const float* gate,
                         int32_t blocks, int32_t bsize, int32_t dtype, float alpha, float beta, void* stream) {
    if (world < 1 || world > 16 || !sums || !dw || !staging || blocks <= 0 || nccl_type(dtype) < 0) return BSMM_ERR_ARG;
    if (bsize != 8 && bsize != 16 && bsize != 32) return BSMM_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t total = (size_t)blocks * bsize * bsize, es = dw_elem_size(dtype);
    size_t shard, lo, hi, need;
    if (int rc = bsmm_dist_dw_layout(world, 0, blocks, bsize, &shard, &lo, &hi, &need)) return rc;
    if (sums_capacity < need) return BSMM_ERR_WORKSPACE;
    PtrList16 src;
    for (int q = 0; q < 16; ++q) src.p[q] = nullptr;
    for (int q = 0; q < world; ++q) {
        if (!sums[q] || !dw[q] || !staging[q] || (reinterpret_cast<uintptr_t>(sums[q]) & 15) || (reinterpret_cast<uintptr_t>(staging[q]) & 15)) return BSMM_ERR_ARG;
    }
    // 1. "reduce-scatter": rank r's shard region of ITS OWN sums buffer receives the sum of that region over all ranks -- the same
    //    world * shard floats of every buffer are read (padding past `total` included) and the same region is written as in the RCCL call
    for (int r = 0; r < world; ++r) {
        for (int q = 0; q < world; ++q) src.p[q] = sums[q] + (size_t)r * shard;
        emu_sum_kernel<<<(unsigned)((shard + 255) / 256), 256, 0, st>>>(src, world, sums[r] + (size_t)r * shard, shard);
    }
    // 2. every rank finalizes its shard with ITS rank's bounds
    for (int r = 0; r < world; ++r) {
        if (int rc = bsmm_dist_dw_layout(world, r, blocks, bsize, &shard, &lo, &hi, nullptr)) return rc;
        if (int rc = dw_finalize_shard(sums[r] + lo, dw[r], staging[r], gate, lo, hi, bsize, dtype, alpha, beta, st)) return rc;
    }
    // 3. "all-gather" of shard-sized pieces (padding included, as the RCCL call moves it), then the copy into dw
    for (int r = 0; r < world; ++r)
        for (int q = 0; q < world; ++q)
            if (q != r) {
                hipError_t e = hipMemcpyAsync(static_cast<char*>(staging[q]) + (size_t)r * shard * es, static_cast<const char*>(staging[r]) + (size_t)r * shard * es,
                                              shard * es, hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return (int)e;
            }
    for (int q = 0; q < world; ++q) {
        hipError_t e = hipMemcpyAsync(dw[q], staging[q], total * es, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipGetLastError();
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

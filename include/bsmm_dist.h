/* bsmm_dist.h -- data-parallel reduction of the weight gradient over RCCL / xGMI, C ABI (libbsmm_hip.so).
 *
 * The hot path shards over the minibatch: lookup tables and W replicated, every rank owns N / world rows,
 * fprop / bprop need no communication, updat yields a partial dw per rank -> ONE all-reduce(sum) per step.  This is the
 * replacement of the reference's AllreduceNccl op for this path (/root/reference/src/nccl_op.cc:166-201: a collective on a
 * side stream, ordered after the producer by an event recorded on the compute stream, :513; the consumer waits on the
 * collective's event).  One process per GPU; the communicator, its stream and its two events live in the opaque handle --
 * there is no global state; calls on one handle must not overlap each other.
 *
 *   rank 0:   bsmm_dist_unique_id(id)            -> 128 bytes, sent to the other ranks by the host program (any channel)
 *   all:      bsmm_dist_create(&h, id, rank, world, device)
 *   per step: ... updat enqueued on `compute` ...
 *             bsmm_dist_allreduce_begin(h, dw, count, dtype, compute)   (returns at once; the collective runs on h's stream)
 *             ... bprop enqueued on `compute`: overlaps with the collective ...
 *             bsmm_dist_allreduce_end(h, compute)                       (`compute` waits for the collective)
 *   or, fused (bsize-32 streaming updat with BSMM_FLAG_DW_SUMS; 25 % fewer bytes on the wire, 1 / world of the finalize per rank):
 *             bsmm_dist_dw_begin(h, sums, sums_capacity, dw, staging, gate, blocks, bsize, dtype, alpha, beta, compute)
 *                 = reduce-scatter of the fp32 sums -> alpha / beta / gate + ONE rounding on this rank's shard -> all-gather of
 *                   the finished dw shards in the storage type -> dw, all on h's stream
 *             ... bprop, and the NEXT step's fprop, enqueued on `compute`: dw is not needed before the optimiser ...
 *             bsmm_dist_dw_end(h, compute)
 *   end:      bsmm_dist_destroy(h)
 * Return values as in bsmm.h (0 ok, > 0 hipError_t, < 0 BSMM_ERR_*); BSMM_ERR_UNSUPPORTED when librccl cannot be loaded.
 * RCCL is bound at run time (dlopen of librccl.so.1 / librccl.so: the one PyTorch-ROCm already has in the process when there
 * is one), so the library itself carries no link-time dependency on it.
 */
#ifndef BSMM_DIST_H_
#define BSMM_DIST_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bsmm_dist bsmm_dist;

#define BSMM_DIST_ID_BYTES 128

int bsmm_dist_unique_id(void* id_out);
int bsmm_dist_create(bsmm_dist** out, const void* id, int32_t rank, int32_t world, int32_t device);
/* in-place sum of `count` elements of dtype BSMM_F32 / BSMM_F16 / BSMM_BF16 (fp32 is the default of the host classes:
 * the cross-rank sum is then not rounded to 16 bit per hop) */
int bsmm_dist_allreduce_begin(bsmm_dist* h, void* buf, size_t count, int32_t dtype, void* producer_stream);
int bsmm_dist_allreduce_end(bsmm_dist* h, void* consumer_stream);
/* Fused reduction of the weight gradient.  sums: this rank's fp32 sums (the start of the workspace of a bsmm_updat call with
 * BSMM_FLAG_DW_SUMS).  CAPACITY REQUIREMENT: the reduce-scatter works on `world` equal shards of bsmm_dist_dw_shard_elems() floats
 * (ceil(total / world) rounded up to 8), so `sums` must hold world * shard floats -- up to 8 * world - 1 more than total =
 * blocks * bsize^2; the tail is scratch (read, summed, never used).  sums_capacity = the floats the caller's buffer really holds at
 * `sums`; less than world * shard -> BSMM_ERR_WORKSPACE, nothing is enqueued (the workspace of bsmm_updat has the room:
 * bsmm_workspace_bytes() includes it; an exactly-sized copy of the sums does not).  staging: world * shard elements of `dtype`;
 * dw: [blocks][bsize][bsize] of `dtype`, read when beta != 0, written with alpha * [gate *] (sum over ranks) + beta * dw.  gate may
 * be NULL.  The cross-rank sum stays fp32 until the single rounding. */
size_t bsmm_dist_dw_shard_elems(int32_t world, int32_t blocks, int32_t bsize);
/* Host-only arithmetic of the fused reduction (what bsmm_dist_dw_begin uses): shard = elements per rank, [lo, hi) = the elements of
 * DW that `rank` finalizes (clamped to total: the last ranks may own less, or nothing), capacity = world * shard = the floats `sums`
 * and the elements `staging` must hold.  Any output pointer may be NULL. */
int bsmm_dist_dw_layout(int32_t world, int32_t rank, int32_t blocks, int32_t bsize, size_t* shard, size_t* lo, size_t* hi, size_t* capacity);
int bsmm_dist_dw_begin(bsmm_dist* h, float* sums, size_t sums_capacity, void* dw, void* staging, const float* gate, int32_t blocks, int32_t bsize,
                       int32_t dtype, float alpha, float beta, void* producer_stream);
/* Single-device emulation of the fused reduction for `world` (<= 16) VIRTUAL ranks -- test infrastructure for the path above on
 * hardware with one GPU: the same layout arithmetic, the same shard-finalize kernel with every rank's own bounds, the collectives
 * replaced by a summing kernel over the ranks' buffers (reading and writing exactly the regions ncclReduceScatter would, padding
 * included) and by shard-sized copies (as ncclAllGather moves them).  sums / dw / staging: HOST arrays of `world` device pointers
 * with the per-rank buffers of bsmm_dist_dw_begin; all work is enqueued on `stream`. */
int bsmm_dist_dw_emulate(int32_t world, float* const* sums, size_t sums_capacity, void* const* dw, void* const* staging, const float* gate,
                         int32_t blocks, int32_t bsize, int32_t dtype, float alpha, float beta, void* stream);
int bsmm_dist_dw_end(bsmm_dist* h, void* consumer_stream);
/* the handle's communication stream (a hipStream_t): host code may enqueue its own pre / post processing of the buffer there
 * (e.g. the fp32 cast of a 16-bit dw) between begin's event wait and the collective -- see blocksparse_amd/dist.py */
void* bsmm_dist_stream(bsmm_dist* h);
int bsmm_dist_world(const bsmm_dist* h);
int bsmm_dist_destroy(bsmm_dist* h);

#ifdef __cplusplus
}
#endif
#endif /* BSMM_DIST_H_ */

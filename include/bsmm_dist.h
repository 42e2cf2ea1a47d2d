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
/* the handle's communication stream (a hipStream_t): host code may enqueue its own pre / post processing of the buffer there
 * (e.g. the fp32 cast of a 16-bit dw) between begin's event wait and the collective -- see blocksparse_amd/dist.py */
void* bsmm_dist_stream(bsmm_dist* h);
int bsmm_dist_world(const bsmm_dist* h);
int bsmm_dist_destroy(bsmm_dist* h);

#ifdef __cplusplus
}
#endif
#endif /* BSMM_DIST_H_ */

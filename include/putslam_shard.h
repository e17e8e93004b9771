/* putslam_shard.h -- sharding of independent frame pairs / sequences over the GPUs of one node, for C and C++ hosts.
 *
 * The path shards with no data-path collective: every GPU runs match -> RANSAC -> refit
 * (ps_vo_pairs_device, include/putslam_hip.h) on its own pairs.  Two exchanges exist, both over RCCL (rccl/rccl.h; xGMI
 * between the GPUs of a node):
 *   * the run's parameter block, rank `root` -> every rank, once at start (ncclBroadcast, 120 bytes);
 *   * the per-pair records -- pose + counts, 72 bytes -- every rank -> rank `root` (ncclGather), after which the ONE sequential
 *     step of the reference runs on the host: VO_k = VO_{k-1} * increment_k with the 0.1 m gate
 *     (reference src/PUTSLAM/PUTSLAM.cpp:735-740; putslam_hip::VOTrajectory in the drop-in layer).
 * The reference itself has no multi-GPU counterpart; bench.py does the same two exchanges through torch.distributed
 * (backend "nccl" = RCCL).  This header is the torch-free form: libputslam_shard.so links librccl and libputslam_hip.
 *
 * Two ways to form a group:
 *   ps_shard_group_create       ONE process drives several GPUs (ncclCommInitAll): a member per listed device;
 *   ps_shard_group_create_rank  one process per GPU (the torch.distributed.run / mpirun shape): every process is one member;
 *                               the 128-byte id comes from ps_shard_unique_id on one rank and travels by whatever the host
 *                               has (a file, a socket, MPI).
 * Every member owns a PsContext (its stream carries the member's kernels AND its collectives, so they are ordered without
 * events).  All functions return PS_OK or a negative PsStatus; ps_shard_last_error gives the text (RCCL's included).
 */
#ifndef PUTSLAM_SHARD_H_
#define PUTSLAM_SHARD_H_

#include "putslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PS_SHARD_RECORD_FLOATS 18   /* pose[16] column-major + numInliers + numMatchesIn, as floats: 72 bytes per pair */
#define PS_SHARD_ID_BYTES 128       /* NCCL_UNIQUE_ID_BYTES */

typedef struct PsShardGroup PsShardGroup;

/* The parameter block of a run (what bench.py's sharding.broadcast_params sends). */
typedef struct PsShardRunParams {
    PsRansacParams params;
    float K[9];
    int32_t estimator;        /* PsEstimator */
    int32_t numHypotheses;
    uint64_t seed;            /* base seed: rank r works with seed + r (one sequence per GPU) or seed + first pair (one sequence split) */
} PsShardRunParams;

int ps_shard_group_create(const int *devices, int numDevices, PsShardGroup **out);
int ps_shard_unique_id(uint8_t id[PS_SHARD_ID_BYTES]);
int ps_shard_group_create_rank(int device, int rank, int worldSize, const uint8_t id[PS_SHARD_ID_BYTES], PsShardGroup **out);
void ps_shard_group_destroy(PsShardGroup *g);
const char *ps_shard_last_error(const PsShardGroup *g);

int ps_shard_world_size(const PsShardGroup *g);          /* ranks in the communicator */
int ps_shard_local_count(const PsShardGroup *g);         /* members this process drives */
int ps_shard_rank(const PsShardGroup *g, int local);     /* rank of local member `local` */
int ps_shard_device(const PsShardGroup *g, int local);
PsContext *ps_shard_context(PsShardGroup *g, int local); /* the member's context: pass it to ps_vo_pairs_device */

/* Contiguous share [*lo, *hi) of `total` units for `rank` of `world` (sizes differ by at most one). */
void ps_shard_range(int64_t total, int world, int rank, int64_t *lo, int64_t *hi);

/* perLocal[local]: in on the root's member, out on every member (bytes identical to the root's).  Blocks until done. */
int ps_shard_broadcast_params(PsShardGroup *g, PsShardRunParams *perLocal, int root);

/* Packs pose / numInliers / numMatchesIn of every local member's `pairsPerRank` results (DEVICE pointers, as written by
 * ps_vo_pairs_device on that member's context) into 72-byte records on the member's stream, gathers them on rank `root`
 * (ncclGather) and copies them to hostRecords[worldSize][pairsPerRank][18] in the process that drives the root.
 * hostRecords may be NULL in processes that do not drive the root.  Blocks until the records are on the host.
 * results[local] with fewer valid pairs than pairsPerRank: pass validPairs[local] (NULL = all); the rest is zero-filled. */
int ps_shard_gather_records(PsShardGroup *g, const PsPairResults *results, const int32_t *validPairs, int pairsPerRank,
                            float *hostRecords, int root);

/* Waits for everything queued on every local member's stream. */
int ps_shard_synchronize(PsShardGroup *g);

#ifdef __cplusplus
}
#endif
#endif /* PUTSLAM_SHARD_H_ */

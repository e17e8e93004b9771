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
 *   ps_shard_group_create       ONE process drives several GPUs (ncclCommInitAll): a member per listed device, and -- from two
 *                               members on -- a host thread per member inside the library: the calls below that address all
 *                               members (submit_all, gather, broadcast) run the members' shares side by side;
 *   ps_shard_group_create_rank  one process per GPU (the torch.distributed.run / mpirun shape): every process is one member;
 *                               the 128-byte id comes from ps_shard_unique_id on one rank and travels by whatever the host
 *                               has (a file, a socket, MPI).
 * Every member owns a PsBatchQueue of four launch chains (include/putslam_hip.h: batches go to the chains in turn, the chains are
 * never joined; consecutive steps run side by side: give them output blocks of their own, four used in turn) and a communication stream.  A step of a looping host is
 *       ps_shard_submit_all(g, jobs)                          every member's batch, asynchronous
 *       ps_shard_gather_records_async(g, pairs, root, &t)     records packed on the chains, gathered on the comm streams
 *       ... the next step's submit ...
 *       ps_shard_wait(g, t, &records)                         when the host wants that step's records
 * and nothing in it makes a chain wait: the records of a batch are packed by a small kernel queued on each chain behind its
 * share of the batch; once the HOST has seen that kernel's event complete -- at the next submit / gather call that finds it so, or
 * in ps_shard_wait, which waits for it -- the gather goes onto the comm stream (ncclGather, the copy to pinned host memory on the
 * root, the ticket's event), tickets in order.  No stream ever holds a wait for another one: a cross-queue wait that stays pending
 * costs the chains 5 - 19 % of their rate (profiles/r06v/pending_waits.txt).  Every rank therefore has to come back -- ps_shard_wait
 * on the ticket (or a later one), or ps_shard_synchronize -- for its share of a gather to be issued; a loop that submits, gathers
 * and waits does.  All functions return PS_OK or a negative PsStatus; ps_shard_last_error gives the text (RCCL's included).
 */
#ifndef PUTSLAM_SHARD_H_
#define PUTSLAM_SHARD_H_

#include "putslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PS_SHARD_RECORD_FLOATS 18   /* pose[16] column-major + numInliers + numMatchesIn, as floats: 72 bytes per pair */
#define PS_SHARD_ID_BYTES 128       /* NCCL_UNIQUE_ID_BYTES */
#define PS_SHARD_GATHERS_IN_FLIGHT 8 /* tickets whose records stay readable: a record block is reused eight gathers later */

typedef struct PsShardGroup PsShardGroup;

/* The parameter block of a run (what bench.py's sharding.broadcast_params sends). */
typedef struct PsShardRunParams {
    PsRansacParams params;
    float K[9];
    int32_t estimator;        /* PsEstimator */
    int32_t numHypotheses;
    uint64_t seed;            /* base seed: rank r works with seed + r (one sequence per GPU) or seed + first pair (one sequence split) */
} PsShardRunParams;

/* One member's batch: the arguments of ps_vo_pairs_device (DEVICE pointers on that member's GPU). */
typedef struct PsShardJob {
    const PsRansacParams *params;
    const PsRansacConfig *cfg;
    const float *K;
    const PsFrameSet *frames;
    const int32_t *pairs;
    int32_t P;
    const PsPairResults *out;
} PsShardJob;

int ps_shard_group_create(const int *devices, int numDevices, PsShardGroup **out);
int ps_shard_unique_id(uint8_t id[PS_SHARD_ID_BYTES]);
int ps_shard_group_create_rank(int device, int rank, int worldSize, const uint8_t id[PS_SHARD_ID_BYTES], PsShardGroup **out);
void ps_shard_group_destroy(PsShardGroup *g);
const char *ps_shard_last_error(const PsShardGroup *g);

int ps_shard_world_size(const PsShardGroup *g);          /* ranks in the communicator */
int ps_shard_local_count(const PsShardGroup *g);         /* members this process drives */
int ps_shard_rank(const PsShardGroup *g, int local);     /* rank of local member `local` */
int ps_shard_device(const PsShardGroup *g, int local);
PsContext *ps_shard_context(PsShardGroup *g, int local); /* the member's context: options set on it before the first submit reach its chains */
PsBatchQueue *ps_shard_queue(PsShardGroup *g, int local);/* the member's batch queue (four chains) */

/* Contiguous share [*lo, *hi) of `total` units for `rank` of `world` (sizes differ by at most one). */
void ps_shard_range(int64_t total, int world, int rank, int64_t *lo, int64_t *hi);

/* perLocal[local]: in on the root's member, out on every member (bytes identical to the root's).  Blocks until done. */
int ps_shard_broadcast_params(PsShardGroup *g, PsShardRunParams *perLocal, int root);

/* jobs[local] for every local member: the batch goes to the member's queue (ps_batch_queue_submit: four chains, whole batches in
 * turn -- consecutive batches run side by side, so a host that keeps two steps in flight gives consecutive steps output
 * blocks of their own) and its 72-byte records are packed behind it on the chain it ran on.  Asynchronous: returns when every member's launches are queued (with
 * two or more local members the members' submissions run on their own host threads, side by side).  A member whose job has
 * P = 0 submits nothing. */
int ps_shard_submit_all(PsShardGroup *g, const PsShardJob *jobs);

/* The records of every member's LAST submitted batch -> rank `root`, asynchronously: noted now, and queued on each member's
 * communication stream as soon as the host has seen that batch's packing complete -- in this call, in a later submit / gather call
 * or in ps_shard_wait (which waits for the packing), gathers in ticket order.  pairsPerRank >= every member's P (blocks are
 * zero-filled up to it; it must be the same on every rank).  *ticket names the gather.  At most PS_SHARD_GATHERS_IN_FLIGHT gathers
 * are outstanding: a further submit settles the oldest (on the host).  Every rank has to come back with ps_shard_wait (on this or a
 * later ticket) or ps_shard_synchronize: a gather completes when every rank has issued its share. */
int ps_shard_gather_records_async(PsShardGroup *g, int pairsPerRank, int root, int64_t *ticket);

/* Queues what is left to queue of the gathers up to `ticket`, then blocks until gather `ticket` has completed on every local member.  In the process that drives the root, *hostRecords (may be
 * NULL) points at [worldSize][pairsPerRank][18] floats in pinned host memory, valid until PS_SHARD_GATHERS_IN_FLIGHT - 1 further
 * gathers have been started; NULL elsewhere. */
int ps_shard_wait(PsShardGroup *g, int64_t ticket, const float **hostRecords);

/* The blocking form for results a host produced itself with ps_vo_pairs_device on ps_shard_context(g, local) (rounds 1 - 5's
 * call; now a wrapper: pack on the member's context stream, the same asynchronous gather, wait, copy out).
 * results[local]: DEVICE pointers; validPairs[local] <= pairsPerRank (NULL = all), the rest is zero-filled.
 * hostRecords[worldSize][pairsPerRank][18] in the process that drives the root (may be NULL elsewhere). */
int ps_shard_gather_records(PsShardGroup *g, const PsPairResults *results, const int32_t *validPairs, int pairsPerRank,
                            float *hostRecords, int root);

/* Waits for everything queued on every local member's streams (chains, context, communication). */
int ps_shard_synchronize(PsShardGroup *g);

#ifdef __cplusplus
}
#endif
#endif /* PUTSLAM_SHARD_H_ */

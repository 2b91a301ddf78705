/* spcbpt_mgpu.h -- the N-GPU host of the SPCBPT hot path: one rank per MI355X, RCCL over xGMI.
 *
 * The reference is single-GPU (no NCCL / MPI / multi-device code; its render loop is
 * optixPathTracer.cpp:791-822 with launchLVCTrace 515-522 and launchSubframe 609-635).  BASELINE.json's north_star shards
 * that loop: every rank traces its share of the light sub-paths, the light-vertex cache is ALL-GATHERED, every rank builds the
 * identical sampler and renders its 8-row bands, and the film is gathered over RCCL.  This library is that host, in C++, on
 * top of include/spcbpt.h (libspcbpt_hip.so) and librccl:
 *
 *   exchange 1 (per frame)   ncclAllGather of {vertex_count, path_count} and of the compact shards padded to an agreed
 *                            capacity, on the communicator's own high-priority stream, which waits on the DEVICE for the light
 *                            pass that fills the shard (spcbpt_lvc_export_on); a device kernel concatenates the shards in rank
 *                            order = global (path, depth) order and leaves the totals on the device
 *                            (spcbpt_lvc_import_gathered); spcbpt_build_sampler sizes itself from an upper bound.  No host
 *                            wait anywhere in a frame.
 *   exchange 2 (per read-out) ncclAllGather of every rank's bands (ceil(bands / N) x 8 x width float4 = 4.15 MB per rank at
 *                            1920 x 1080, N = 8) instead of a sum over zero-padded 33 MB images.
 *   start-up                 ncclBroadcast of the subspace tuple rank 0 trained (trees, Q, Gamma).
 *
 * A rank is a (context, communicator) pair driven by one host thread; ranks may be threads of one process
 * (tools/spcbpt_render_mgpu) or processes (bench.py under torch.distributed.run, which hands the unique id around).
 * SPCBPT_COMM_LOCAL communicators stand in for RCCL when several ranks share ONE device (RCCL refuses that): the same
 * sharding, compaction and device-count build, with device-to-device copies as the transport -- what the one-GPU test box runs.
 * Every function returns 0 or a negative spcbpt_status; spcbpt_comm_last_error gives the text.  Nothing throws. */
#ifndef SPCBPT_MGPU_H
#define SPCBPT_MGPU_H

#include "spcbpt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spcbpt_comm spcbpt_comm;

#define SPCBPT_UNIQUE_ID_BYTES 128   /* = NCCL_UNIQUE_ID_BYTES */

/* ncclGetUniqueId: called once (rank 0); the launcher copies the bytes to every rank. */
int spcbpt_comm_unique_id(char id[SPCBPT_UNIQUE_ID_BYTES]);

/* ncclCommInitRank for `ctx` (its device must be current for the calling thread; every rank calls this, it synchronises with
 * the others).  The context must have its light pass configured (spcbpt_set_light_trace) -- the default shard capacity is the
 * largest scratch capacity core_count x core_padding of any rank (agreed with an all-reduce: the ranks' core ranges differ when
 * num_core is not a multiple of world); spcbpt_comm_calibrate tightens it and should be called by every job: the context's caches
 * are sized from a measured pass (spcbpt_lvc_set_capacity) PER RANK, so an uncalibrated shard capacity may exceed a rank's cache.
 * Such a rank stages its shard through the communicator's send buffer (one device copy per exchange that calibrate saves); it is
 * never refused on one rank only -- every condition that fails an exchange is derived from the gathered counts and fails on all. */
int spcbpt_comm_create(spcbpt_ctx* ctx, int rank, int world, const char id[SPCBPT_UNIQUE_ID_BYTES], spcbpt_comm** out);

/* `world` ranks on ONE device, in one process: out[r] is the communicator of ctxs[r].  Same call sequence as the RCCL form; the
 * collective calls of the ranks may be issued from one thread in any order (each completes when the last rank has called). */
int spcbpt_comm_create_local(spcbpt_ctx* const* ctxs, int world, spcbpt_comm** out);

int spcbpt_comm_destroy(spcbpt_comm* comm);
const char* spcbpt_comm_last_error(const spcbpt_comm* comm);

/* Shard capacity of exchange 1, in vertices (the same on every rank).  calibrate: traces `passes` light passes (launch frames
 * first_frame ...), all-gathers their counts WITH a host wait (start-up only) and sets the capacity to `slack` x the largest
 * shard seen, rounded up to 1024 and capped by the scratch capacity.  A later shard that does not fit is reported by the next
 * spcbpt_sync as SPCBPT_ERR_CAPACITY (never silently truncated). */
int spcbpt_comm_set_shard_capacity(spcbpt_comm* comm, int vertices);
int spcbpt_comm_get_shard_capacity(const spcbpt_comm* comm, int* vertices);
int spcbpt_comm_calibrate(spcbpt_comm* comm, int passes, uint32_t first_frame, float slack);

/* Exchange 1 for the OLDEST pending light pass of the rank's context (spcbpt_set_light_ahead order), to be followed by
 * spcbpt_build_sampler.  Queues work only. */
int spcbpt_comm_exchange_lvc(spcbpt_comm* comm);

/* The same for the n OLDEST pending passes at once -- the passes of one spcbpt_launch_light_batch: ONE all-gather of the n packed
 * shards (n x capacity vertices per rank) and one of the n count pairs, ONE compaction kernel (grid.y = frame); the sets are left
 * as n calls of spcbpt_comm_exchange_lvc would leave them, bit for bit, and n calls of spcbpt_build_sampler follow.  1 <= n <= 32. */
int spcbpt_comm_exchange_lvc_batch(spcbpt_comm* comm, int n_frames);

/* What the communicator is: rank and size as the transport itself reports them (ncclCommUserRank / ncclCommCount), and the
 * transport. */
#define SPCBPT_COMM_RCCL 0
#define SPCBPT_COMM_LOCAL 1
int spcbpt_comm_info(const spcbpt_comm* comm, int* rank, int* world, int* transport);

/* Exchange 2: the full width x height float4 film, gathered from every rank's bands, into `out_device` (device pointer of the
 * calling rank, width x height x 4 floats) or, when out_device is null, into the rank's own accum buffer.  Waits for the
 * rank's render streams first (it is a read-out) and returns when the image is complete on this rank. */
int spcbpt_comm_gather_film(spcbpt_comm* comm, void* out_device);

/* Start-up: rank `root`'s installed subspace tuple (spcbpt_preprocess / spcbpt_set_subspace) to every rank. */
int spcbpt_comm_broadcast_subspace(spcbpt_comm* comm, int root);

/* Host barrier over the communicator's stream (a 1-element all-reduce + stream sync): for timing brackets. */
int spcbpt_comm_barrier(spcbpt_comm* comm);

/* max over ranks of a host double (timing: the slowest rank's region). */
int spcbpt_comm_max_double(spcbpt_comm* comm, double* value);

#ifdef __cplusplus
}
#endif
#endif

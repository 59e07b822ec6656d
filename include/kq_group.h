/* kq_group.h — ONE root cohort tree over several GPUs of one process, behind the C ABI.
 *
 * What kueue_amd/sharding.py ShardedCycle does over torch.distributed, for a caller that is not Python: the Go controller owns one
 * kq_group (= one kq_engine per device + an RCCL communicator over xGMI, created with ncclCommInitAll inside this process) and calls
 * kq_group_cycle_run where it would call kq_cycle_run (pkg/scheduler/scheduler.go:308-386 steps 3-5):
 *   1. every engine holds the same snapshot and gets the same heads; engine r nominates the heads h with h % n == r
 *      (kq_cycle_nominate_shard: scheduler.go:665-705 reads nothing but the cycle-start snapshot) into an int64 exchange buffer in ITS
 *      device memory that is zero outside its own heads;
 *   2. ncclAllReduce(SUM, int64) of that buffer over the group — the one collective of the cycle (disjoint supports: the sum is a gather);
 *   3. every engine runs iterator order + processEntry over ALL heads (kq_cycle_process_merged): decisions and the resident usage plane
 *      are identical on every device by construction; rank 0's decisions are returned.
 * The per-device work runs on one PERSISTENT host thread per device (the calls of include/kq_engine.h are synchronous; rank 0 is the
 * caller's thread); a cycle is one job that every rank walks on its own, meeting the others at three phase barriers; a rank that fails
 * takes every rank out of the cycle at the next barrier, and the all-reduces of all ranks are issued by one thread inside one
 * ncclGroupStart / ncclGroupEnd, so no rank is left waiting in a collective. librccl is loaded with dlopen when a group of more than
 * one device is created without KQ_GROUP_HOST_COLLECTIVE: libkq_engine.so itself keeps linking libamdhip64 only.
 * Errors: 0 or a negative KQ_E* of kq_engine.h — the ENGINE's code at every group size (KQ_ECAPACITY stays KQ_ECAPACITY: grow the
 * target / reason buffers and call again); KQ_EDEVICE also covers RCCL failures (kq_group_last_error has the text). */
#ifndef KQ_GROUP_H
#define KQ_GROUP_H
#include "kq_engine.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct kq_group kq_group;

/* flags of kq_group_create_opts */
#define KQ_GROUP_HOST_COLLECTIVE 1u  /* the exchange buffers are summed through pinned host memory instead of ncclAllReduce: no RCCL is
                                        loaded and the same device ordinal may appear more than once (several engines on ONE GPU). The
                                        seam the single-GPU test tier and the CPU emulation run the N > 1 protocol through. */
#define KQ_GROUP_FORCE_SHARDED   2u  /* a group of ONE device also goes export -> import -> kq_cycle_process_merged instead of
                                        kq_cycle_run (measures what the sharded path costs by itself) */

/* n_dev >= 1 distinct device ordinals of this process (1 .. 64); cfg->device is ignored. Flags from the environment:
 * KQ_GROUP_COLLECTIVE=host -> KQ_GROUP_HOST_COLLECTIVE, KQ_GROUP_FORCE_SHARDED=1 -> KQ_GROUP_FORCE_SHARDED. */
int  kq_group_create(const kq_config* cfg, int32_t n_dev, const int32_t* devices, kq_group** out);
/* The same with explicit flags (the environment is not read). */
int  kq_group_create_opts(const kq_config* cfg, int32_t n_dev, const int32_t* devices, uint32_t flags, kq_group** out);
void kq_group_destroy(kq_group* g);
int  kq_group_size(const kq_group* g);
/* cache.Snapshot -> every device (snapshot.go:171). */
int  kq_group_snapshot_put(kq_group* g, const kq_snapshot* s);
/* One scheduling cycle over the group; `out` as for kq_cycle_run. */
int  kq_group_cycle_run(kq_group* g, const kq_heads* h, kq_decisions* out);
/* kq_cycle_commit / kq_cycle_release on every device (the resident snapshots stay in step). */
int  kq_group_cycle_commit(kq_group* g, int32_t* n_admitted);
int  kq_group_cycle_release(kq_group* g, int32_t age);
/* The resident usage plane of device `rank` ([N * n_fr]): equal on all ranks; tests compare them. */
int  kq_group_read_usage(kq_group* g, int32_t rank, int64_t* usage);
const char* kq_group_last_error(kq_group* g);
/* What the group's exchange actually ran on — for a caller (bench.py --gpus N) that must not mistake the host seam for RCCL:
 * *rccl_ranks = communicators ncclCommInitAll created (0: the host collective, or a group of one), *allreduce_calls = ncclAllReduce
 * groups issued so far (one per cycle of a group of several devices), *host_sums = exchanges summed through host memory. */
int  kq_group_collective_info(kq_group* g, int32_t* rccl_ranks, int64_t* allreduce_calls, int64_t* host_sums);

#ifdef __cplusplus
}
#endif
#endif

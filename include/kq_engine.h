/*
 * kq_engine.h — C ABI of the MI355X-native admission / flavor-assignment engine.
 *
 * This is the drop-in boundary for ONE path of kubernetes-sigs/kueue: the decision logic of a
 * scheduling cycle, i.e. what `(*Scheduler).schedule` (pkg/scheduler/scheduler.go:308-386)
 * computes between `cache.Snapshot()` (step 2) and the side effects of step 5/6:
 *
 *   nominate            scheduler.go:665   -> getAssignments :821 -> flavorassigner.Assign
 *                                             (flavorassigner.go:696) + preemption.GetTargets
 *                                             (preemption/preemption.go:132)
 *   makeIterator        scheduler.go:1080  (classical :1110 / fair sharing fair_sharing_iterator.go:39)
 *   processEntry        scheduler.go:392   (fits :771, AddUsage, overlap skip, DeferredFit ...)
 *
 * The reference has no FFI seam (it builds with CGO_ENABLED=0, Makefile:69); the seam is cut here.
 * Everything crossing the boundary is plain C: pointers + sizes, caller-owned SoA arrays, no
 * callbacks, no pointers retained after a call returns (cgo pointer-passing rules). The Go-side
 * binding a maintainer would add is shown in INTEGRATION.md and shim/go/.
 *
 * Conventions
 *   - return 0 on success, negative KQ_E* otherwise; on any error the caller runs the stock Go
 *     path for that cycle.
 *   - single caller thread per engine (the scheduler goroutine, runtime.LockOSThread).
 *   - quantities are int64 in the reference's canonical units (milli-CPU, bytes, counts);
 *     INT64_MAX is resources.Unlimited (pkg/resources/amount.go:46-60).
 *   - node index space: ClusterQueues are nodes [0, n_cq), Cohorts are nodes [n_cq, n_cq+n_cohort).
 *     Both ranges are in CANONICAL ORDER = ascending name (SURVEY.md §8c determinism contract).
 *   - flavor-resource index fr = flavor * n_resource + resource  (pkg/resources/resource.go:29).
 */
#ifndef KQ_ENGINE_H
#define KQ_ENGINE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KQ_ABI_VERSION 3

/* ---- error codes ---------------------------------------------------------------------------- */
#define KQ_OK             0
#define KQ_EINVAL        (-1)  /* malformed input (bad index, cycle in cohort tree, ...)        */
#define KQ_ENOMEM        (-2)
#define KQ_EDEVICE       (-3)  /* HIP error; kq_strerror() carries the hipGetErrorString        */
#define KQ_EUNSUPPORTED  (-4)  /* input uses a feature the device path does not implement       */
#define KQ_ECAPACITY     (-5)  /* caller-provided output buffer too small (targets)             */
#define KQ_ENODEVICE     (-6)  /* no HIP device / extension not usable                          */

/* ---- size limits of one snapshot (checked by kq_snapshot_put before anything is allocated: larger -> KQ_EUNSUPPORTED, a negative
 *      count -> KQ_EINVAL; no entry point aborts the calling process: C++ exceptions of the host side end at the ABI as KQ_ENOMEM /
 *      KQ_EINVAL with kq_last_error) ----------------------------------------------------------- */
#define KQ_MAX_ADMITTED (1 << 26)   /* admitted workloads (rows)                                   */
#define KQ_MAX_NODES    (1 << 24)   /* ClusterQueues + Cohorts                                     */
#define KQ_MAX_FR       (1 << 16)   /* flavors x resources; nodes x flavor-resources < 2^31        */

/* ---- quantities ----------------------------------------------------------------------------- */
#define KQ_UNLIMITED   INT64_MAX  /* resources.Unlimited, amount.go:56                           */
#define KQ_NIL_LIMIT   (-1)       /* nil BorrowingLimit / LendingLimit (resource.go:26-33)       */

/* ---- quota_flags[node*n_fr+fr] -------------------------------------------------------------- */
#define KQ_QF_QUOTA    0x1  /* fr is a key of resourceNode.Quotas        (resource_node.go:33)  */
#define KQ_QF_SUBTREE  0x2  /* fr is a key of resourceNode.SubtreeQuota  (resource_node.go:38)  */

/* ---- per-ClusterQueue policy word (clusterqueue_snapshot.go:53-66) -------------------------- */
/* bits 0-1  Preemption.WithinClusterQueue : 0 Never 1 LowerPriority 2 LowerOrNewerEqualPriority */
/* bits 2-3  Preemption.ReclaimWithinCohort: 0 Never 1 LowerPriority 2 LowerOrNewerEqualPriority 3 Any */
/* bit  4    Preemption.BorrowWithinCohort.Policy : 0 Never(or nil) 1 LowerPriority              */
/* bit  5    BorrowWithinCohort.MaxPriorityThreshold != nil (value in cq_borrow_prio_threshold)  */
/* bit  6    FlavorFungibility.WhenCanBorrow  : 0 MayStopSearch(=Borrow) 1 TryNextFlavor         */
/* bit  7    FlavorFungibility.WhenCanPreempt : 0 MayStopSearch(=Preempt) 1 TryNextFlavor        */
/* bits 8-9  FlavorFungibility.Preference : 0 nil 1 BorrowingOverPreemption 2 PreemptionOverBorrowing */
/* bit  10   QueueingStrategy : 0 BestEffortFIFO 1 StrictFIFO                                    */
/* bit  11   ReclaimWithinCohort is the EMPTY string (object built without API defaulting, as the
             reference's unit tests do): behaves as Never everywhere except canPreemptWhileBorrowing,
             which tests `!= PreemptionPolicyNever` (flavorassigner.go:1386-1389)                  */
#define KQ_POL_WITHIN_CQ(p)        ((p) & 0x3u)
#define KQ_POL_RECLAIM(p)          (((p) >> 2) & 0x3u)
#define KQ_POL_BORROW_WITHIN(p)    (((p) >> 4) & 0x1u)
#define KQ_POL_HAS_THRESHOLD(p)    (((p) >> 5) & 0x1u)
#define KQ_POL_BORROW_TRYNEXT(p)   (((p) >> 6) & 0x1u)
#define KQ_POL_PREEMPT_TRYNEXT(p)  (((p) >> 7) & 0x1u)
#define KQ_POL_PREFERENCE(p)       (((p) >> 8) & 0x3u)
#define KQ_POL_STRICT_FIFO(p)      (((p) >> 10) & 0x1u)
#define KQ_POL_RECLAIM_UNSET(p)    (((p) >> 11) & 0x1u)
#define KQ_POLICY_NEVER 0
#define KQ_POLICY_LOWER_PRIORITY 1
#define KQ_POLICY_LOWER_OR_NEWER_EQUAL 2
#define KQ_POLICY_ANY 3
#define KQ_PREF_NONE 0
#define KQ_PREF_BORROWING_OVER_PREEMPTION 1
#define KQ_PREF_PREEMPTION_OVER_BORROWING 2

/* ---- feature gates (pkg/features/kube_features.go:612-910); KQ_GATES_DEFAULT = upstream ----- */
#define KQ_GATE_FLAVOR_FUNGIBILITY            (1u << 0)
#define KQ_GATE_PRESERVE_SCAN_PROGRESS        (1u << 1)  /* FlavorFungibilityPreserveScanProgress */
#define KQ_GATE_PARTIAL_ADMISSION             (1u << 2)
#define KQ_GATE_PRIORITY_SORTING_IN_COHORT    (1u << 3)  /* PrioritySortingWithinCohort           */
#define KQ_GATE_FS_PREEMPT_WITHIN_NOMINAL     (1u << 4)  /* FairSharingPreemptWithinNominal       */
#define KQ_GATE_FS_PRIORITIZE_NON_BORROWING   (1u << 5)  /* FairSharingPrioritizeNonBorrowing     */
#define KQ_GATE_RECOMPUTE_ON_OVERLAP          (1u << 6)  /* RecomputeAssignmentUponPreemptionTargetsOverlap */
#define KQ_GATE_PRIORITIZE_PREEMPTORS         (1u << 7)  /* PrioritizePreemptorWorkloads (off)    */
#define KQ_GATE_QUOTA_CHECK_STRATEGY          (1u << 8)
#define KQ_GATE_SCHEDULING_EQUIVALENCE_HASHING (1u << 9) /* SchedulingEquivalenceHashing (pending side only)  */
#define KQ_GATE_ELASTIC_JOBS                  (1u << 10) /* ElasticJobsViaWorkloadSlices (kq_heads.slice_*)        */
#define KQ_GATES_DEFAULT (KQ_GATE_FLAVOR_FUNGIBILITY | KQ_GATE_PRESERVE_SCAN_PROGRESS |            \
                          KQ_GATE_PARTIAL_ADMISSION | KQ_GATE_PRIORITY_SORTING_IN_COHORT |         \
                          KQ_GATE_FS_PREEMPT_WITHIN_NOMINAL | KQ_GATE_FS_PRIORITIZE_NON_BORROWING |\
                          KQ_GATE_RECOMPUTE_ON_OVERLAP | KQ_GATE_QUOTA_CHECK_STRATEGY |            \
                          KQ_GATE_SCHEDULING_EQUIVALENCE_HASHING | KQ_GATE_ELASTIC_JOBS)

/* fair-sharing preemption strategies (apis/config/v1beta2; preemption.go:363-379) */
#define KQ_FS_LESS_THAN_OR_EQUAL_TO_FINAL_SHARE 0  /* rule S2-a */
#define KQ_FS_LESS_THAN_INITIAL_SHARE           1  /* rule S2-b */

/* configapi.QuotaCheckStrategy */
#define KQ_QUOTA_CHECK_BLOCK_UNDECLARED  0
#define KQ_QUOTA_CHECK_IGNORE_UNDECLARED 1

/* Engine configuration = what cmd/kueue/main.go:675-688 passes as scheduler.With* options. */
typedef struct kq_config {
  int32_t  abi_version;        /* KQ_ABI_VERSION */
  int32_t  device;             /* HIP device ordinal */
  uint32_t gates;              /* KQ_GATE_* ; use KQ_GATES_DEFAULT */
  int32_t  fair_sharing;       /* fairsharing.Enabled(cfg.FairSharing) */
  int32_t  n_fs_strategies;    /* 0 => default [S2-a, S2-b] (preemption.go:364-366) */
  int32_t  fs_strategies[2];
  int32_t  quota_check_strategy;
} kq_config;

/* ---- snapshot: what cache.Snapshot() returns (pkg/cache/scheduler/snapshot.go:171), flattened - */
typedef struct kq_snapshot {
  int32_t n_cq, n_cohort, n_flavor, n_resource;
  int32_t pods_resource;            /* index of corev1.ResourcePods in the resource dictionary, -1 if absent */
  const int32_t* resource_order;    /* [n_resource] rank of each resource under SliceRequests order:
                                       FNV-1a64(name), then name (pkg/resources/slice_requests.go:35-60) */
  /* hierarchy (pkg/cache/hierarchy): N = n_cq + n_cohort nodes */
  const int32_t* parent;            /* [N] node index of the parent Cohort, -1 = none */
  const int32_t* child_cohort_off;  /* [n_cohort+1] CSR into child_cohort */
  const int32_t* child_cohort;      /* node ids, ascending name */
  const int32_t* child_cq_off;      /* [n_cohort+1] CSR into child_cq */
  const int32_t* child_cq;          /* node ids (= cq index), ascending name */
  const double*  fair_weight;       /* [N] ClusterQueueSnapshot.FairWeight / CohortSnapshot.FairWeight */
  /* resourceNode planes, [N * n_fr], n_fr = n_flavor*n_resource (resource_node.go:30-45) */
  const int64_t* nominal;           /* Quotas[fr].Nominal */
  const int64_t* borrow_limit;      /* Quotas[fr].BorrowingLimit or KQ_NIL_LIMIT */
  const int64_t* lend_limit;        /* Quotas[fr].LendingLimit   or KQ_NIL_LIMIT */
  const int64_t* subtree_quota;     /* SubtreeQuota[fr] */
  const int64_t* usage;             /* Usage[fr] */
  const uint8_t* quota_flags;       /* KQ_QF_* */
  /* per ClusterQueue */
  const int32_t* cq_rg_off;         /* [n_cq+1] CSR: resource groups of a CQ (ResourceGroups order) */
  const int32_t* rg_flavor_off;     /* [n_rg+1]  CSR: rg.Flavors in declared order */
  const int32_t* rg_flavor;         /* flavor indices */
  const int32_t* rg_res_off;        /* [n_rg+1]  CSR: rg.CoveredResources */
  const int32_t* rg_res;            /* resource indices */
  const uint32_t* cq_policy;        /* [n_cq] policy word above */
  const int32_t* cq_borrow_prio_threshold; /* [n_cq] BorrowWithinCohort.MaxPriorityThreshold */
  const int64_t* cq_generation;     /* [n_cq] AllocatableResourceGeneration */
  /* admitted workloads = union of ClusterQueueSnapshot.Workloads, rows grouped by CQ */
  int32_t n_adm;
  const int32_t* cq_adm_off;        /* [n_cq+1] rows of CQ c are [cq_adm_off[c], cq_adm_off[c+1]) */
  const int64_t* adm_priority;      /* priority.EffectivePriority (util/priority/priority.go:77) */
  const int64_t* adm_queue_ts;      /* Ordering.GetQueueOrderTimestamp, ns (workload.go:1182) */
  const int64_t* adm_reserve_ts;    /* quotaReservationTime, ns; "now" if absent (common/ordering.go:94) */
  const uint32_t* adm_uid_rank;     /* rank of Obj.UID under Go bytewise string order (ordering.go:77) */
  const uint8_t* adm_flags;         /* bit0: workloadevict.IsEvicted */
  const int32_t* adm_use_off;       /* [n_adm+1] CSR: Info.Usage().Quota.Assigned (workload.go:447) */
  const int32_t* adm_use_fr;        /* fr index; an entry exists for every (resource -> assigned flavor),
                                       zero quantities included (WorkloadUsesResources, candidate_generator.go:54) */
  const int64_t* adm_use_qty;
} kq_snapshot;

#define KQ_ADM_EVICTED 0x1

/* ---- heads: queues.Heads() (pkg/cache/queue/manager.go:903) after nominate's gatekeeping ------ */
#define KQ_HEAD_HAS_QUOTA_RESERVATION 0x1  /* workload.HasQuotaReservation (second pass)          */
#define KQ_HEAD_IS_PREEMPTOR          0x2  /* Head.IsPreemptor                                    */
#define KQ_HEAD_HAS_LAST_ASSIGNMENT   0x4  /* Info.LastAssignment != nil                          */
#define KQ_HEAD_HAS_UNHEALTHY_NODES   0x8  /* workload.HasUnhealthyNodes (kq_cycle_run_tas: the second pass after a node failure)      */
#define KQ_HEAD_UNHEALTHY_ASSIGNMENT  0x10 /* workload.HasTopologyAssignmentWithUnhealthyNode workload.go:1392 (admitted, and a
                                            * TopologyAssignment names one of Status.UnhealthyNodes)                                     */

typedef struct kq_heads {
  int32_t n;                        /* number of heads; canonical order = as given (CQ name asc) */
  int64_t cycle;                    /* Scheduler.schedulingCycle (scheduler.go:97,309) */
  const int32_t* cq;                /* [n] ClusterQueue index */
  const int64_t* priority;          /* [n] EffectivePriority */
  const int64_t* queue_ts;          /* [n] GetQueueOrderTimestamp, ns */
  const uint32_t* flags;            /* [n] KQ_HEAD_* */
  const int32_t* ps_off;            /* [n+1] CSR: podsets of a head (Info.TotalRequests, workload.go:245) */
  /* per podset (n_ps = ps_off[n]) */
  const int32_t* ps_count;          /* PodSetResources.Count */
  const int32_t* ps_min_count;      /* PodSet.MinCount, -1 = nil (partial admission) */
  const int32_t* ps_req_off;        /* [n_ps+1] CSR: PodSetResources.Requests */
  const int32_t* req_res;           /* resource index */
  const int64_t* req_qty;           /* total quantity for the whole podset (per-pod x count) */
  const uint64_t* ps_flavor_ok;     /* [n_ps * n_fwords] bit f = checkFlavorForPodSets passes for flavor f
                                       (taints / node affinity / TAS match, flavorassigner.go:1212-1261);
                                       n_fwords = (n_flavor+63)/64 */
  const int32_t* ps_last_tried;     /* [n_ps * n_resource] LastAssignment.LastTriedFlavorIdx[ps][res],
                                       -1 = absent (workload.go:226-238) */
  const int64_t* last_generation;   /* [n] LastAssignment.ClusterQueueGeneration */
  const int64_t* last_cycle;        /* [n] LastAssignment.SchedulingCycle */
  const uint64_t* last_hash;        /* [n] LastAssignment.SchedulingHash (0 = unknown) */
  const uint64_t* hash;             /* [n] Info.SchedulingHash (0 = unknown) */
  /* Workload slices (ElasticJobsViaWorkloadSlices): a head that replaces an admitted slice of the same job
   * (workloadslicing.ReplacedWorkloadSlice, scheduler.go:883: annotation lookup in queue.Workloads, same namespace — host-evaluated).
   * All NULL: no head replaces a slice. The old slice's podsets are aligned with the head's by index (replaced.TotalRequests[psID],
   * flavorassigner.go:1127; the host also resolves findOldPodSetRequest's lookup by podset NAME to that index). */
  const int32_t* slice_row;         /* [n] admitted row of the replaced slice (a row of the head's ClusterQueue), -1 = none */
  const int32_t* ps_slice_count;    /* [n_ps] replaced.TotalRequests[i].Count (Assignment.TotalRequestsFor :265) */
  const int32_t* req_slice_flavor;  /* [n_req] flavor the old slice holds for (podset, resource) (PodSetResources.Flavors), -1 = none */
  const int64_t* req_slice_qty;     /* [n_req] the old slice's request for (podset, resource) */
  const int32_t* ps_slice_pods_flavor; /* [n_ps] the same two for the `pods` request the assigner injects (flavorassigner.go:743-749) */
  const int64_t* ps_slice_pods_qty; /* [n_ps] */
  /* PodSet groups (PodSet.TopologyRequest.PodSetGroupName): assignFlavors runs ONE flavor scan per group over the sum of the members'
   * requests (flavorassigner.go:782-860), every member then keeps the group's flavors for the resources it requests itself
   * (resolvePodSetFlavors :917-945) and the group's Status. NULL: no podset of the batch is in a group (kq_cycle_run_tas: NULL = take
   * kq_cycle_tas.ps_group). The members of a group must be consecutive podsets of their head (KQ_EUNSUPPORTED otherwise), and a head
   * that replaces a workload slice holds no group of several podsets. The pending side keeps the column resident (kq_pending_put /
   * kq_pending_add / kq_pending_update) and Heads() hands it to the cycle. */
  const int32_t* ps_group;          /* [n_ps] group id (any value >= 0, equal inside a group), -1 = none */
} kq_heads;

/* ---- decisions -------------------------------------------------------------------------------- */
/* flavorassigner.FlavorAssignmentMode (flavorassigner.go:453-472) */
#define KQ_MODE_NOFIT        0
#define KQ_MODE_PREEMPT      1
#define KQ_MODE_DEFERRED_FIT 2
#define KQ_MODE_FIT          3
/* entryStatus (scheduler.go:617-630) */
#define KQ_ST_NOT_NOMINATED  0
#define KQ_ST_NOMINATED      1
#define KQ_ST_SKIPPED        2
#define KQ_ST_EVICTED        4  /* kq_cycle_run_tas: handleFailedTASReplacement scheduler.go:522 (TASFailedNodeReplacementFailFast) */
#define KQ_ST_ASSUMED        5
/* qcache.RequeueReason (cluster_queue.go:54-65) */
#define KQ_RQ_GENERIC                  0
#define KQ_RQ_FAILED_AFTER_NOMINATION  1
#define KQ_RQ_PENDING_PREEMPTION       4
#define KQ_RQ_NOFIT                    7
#define KQ_RQ_PREEMPTION_NO_CANDIDATES 8
/* what the Go side has to do for this entry */
#define KQ_ACT_NONE     0  /* requeueAndUpdate only */
#define KQ_ACT_ADMIT    1  /* Scheduler.admit (scheduler.go:991) */
#define KQ_ACT_PREEMPT  2  /* Scheduler.issuePreemptions (scheduler.go:563) on targets */
#define KQ_ACT_EVICT    3  /* evictWorkloadAfterFailedTASReplacement (scheduler.go:926): no replacement for the unhealthy node */
/* why an entry was skipped (inadmissibleMsg selector, scheduler.go:470-481) */
#define KQ_SKIP_NONE            0
#define KQ_SKIP_OVERLAP         1  /* "Workload has overlapping preemption targets with another workload" */
#define KQ_SKIP_NO_LONGER_FITS  2  /* "Workload no longer fits after processing another workload" */
/* preemption reasons (apis/kueue/v1beta2 workload_types.go; hierarchical_preemption.go:46-58) */
#define KQ_REASON_IN_CLUSTER_QUEUE               0
#define KQ_REASON_IN_COHORT_RECLAMATION          1
#define KQ_REASON_IN_COHORT_FAIR_SHARING         2
#define KQ_REASON_IN_COHORT_RECLAIM_WHILE_BORROWING 3
#define KQ_REASON_REPLACED_SLICE                 4  /* not a preemption: the old workload slice the head replaces. It is part of the
                                                      * targets while the cycle runs (scheduler.go:883-899, :771-777); FindReplacedSliceTarget
                                                      * (:492) takes it out before issuePreemptions and Scheduler.admit finishes it (:605) */

/* Why a flavor was not assigned as Fit: the operands of one Status.reasons string (flavorassigner.go:349). The caller formats
 * the text (shim/go/messages.go; kueue_amd/messages.py), so no string crosses the boundary. */
#define KQ_RSN_EXCEEDS_MAX_CAPACITY  1  /* flavorassigner.go:1353: a = previously considered podsets requests (assumedUsage), b = current
                                           podset request (requestUsage), c = maximum capacity (PotentialAvailable)              */
#define KQ_RSN_INSUFFICIENT_UNUSED   2  /* :1372: a = val - Available(fr), the "more needed" amount                              */
#define KQ_RSN_NOT_IN_NOMINATION     3  /* :1097: flavor skipped by the nomination mapping; resource = the scan's resource name   */
#define KQ_RSN_FLAVOR_INELIGIBLE     4  /* :1105-1113: checkFlavorForPodSets failed (ps_flavor_ok bit clear); the host knows the text */
#define KQ_RSN_RESOURCE_UNAVAILABLE  5  /* :1080: no resource group of the ClusterQueue covers `resource`                         */
#define KQ_RSN_SLICE_FLAVOR_MISMATCH 6  /* :1132-1137: "could not assign %s flavor since the original workload is assigned: %s";
                                         * flavor = the flavor tried, a = the old slice's flavor for `resource` (-1: none)          */
#define KQ_RSN_TAS_FAILURE         200  /* kq_cycle_run_tas only (flavorassigner.go:871-877): the placement of the podset's TAS request failed on the
                                           snapshot, psAssignment.reason(failure.Reason): flavor = the TAS flavor, a = KQ_TAS_* status, b / c = its
                                           operands (include/kq_tas.h; notFitMessage tas_flavor_snapshot.go:1997 is regenerated from them)              */
#define KQ_RSN_TRUNCATED           255  /* not a reference reason: the head produced more records than its window holds; the list
                                         * returned for that head is incomplete (flavor = resource = -1). Retry with a larger rsn_cap. */

typedef struct kq_decisions {
  /* per head [n] */
  uint8_t* status;          /* KQ_ST_*   */
  uint8_t* action;          /* KQ_ACT_*  */
  uint8_t* nominated_mode;  /* Assignment.RepresentativeMode() after nominate */
  uint8_t* mode;            /* RepresentativeMode() when processEntry decided (may be DeferredFit) */
  uint8_t* requeue_reason;  /* KQ_RQ_* as passed to queues.RequeueWorkload */
  uint8_t* skip;            /* KQ_SKIP_* */
  int32_t* borrowing;       /* Assignment.Borrowing */
  int32_t* order;           /* position in the entry iterator (0-based) */
  /* per (podset, resource) [n_ps * n_resource] */
  int32_t* flavor;          /* assigned flavor index, -1 = none */
  uint8_t* res_mode;        /* FlavorAssignment.Mode */
  int32_t* tried_idx;       /* FlavorAssignment.TriedFlavorIdx -> next LastAssignment */
  /* per podset [n_ps] */
  int32_t* ps_count;        /* PodSetAssignment.Count (differs from input under partial admission) */
  /* preemption targets, CSR over heads */
  int32_t* tgt_off;         /* [n+1] */
  int32_t  tgt_cap;         /* capacity of tgt_adm / tgt_reason */
  int32_t* tgt_adm;         /* admitted-workload row */
  uint8_t* tgt_reason;      /* KQ_REASON_* */
  /* reason records, CSR over heads, in the order the reference appends them (PodSetAssignment.Status.reasons of the assignment
   * the entry ends the cycle with; Assignment.Message = podsets joined, flavorassigner.go:229). rsn_cap == 0: not reported. */
  int32_t  rsn_cap;         /* capacity of the rsn_* arrays, in records */
  int32_t* rsn_off;         /* [n+1] */
  uint8_t* rsn_code;        /* KQ_RSN_* */
  uint8_t* rsn_podset;      /* podset index within the head */
  int16_t* rsn_flavor;      /* flavor index, -1 = none */
  int16_t* rsn_resource;    /* resource index, -1 = none */
  int64_t* rsn_a;
  int64_t* rsn_b;
  int64_t* rsn_c;
} kq_decisions;

typedef struct kq_engine kq_engine;

/* scheduler.New (scheduler.go:182): build an engine bound to one HIP device. */
int  kq_engine_create(const kq_config* cfg, kq_engine** out);
void kq_engine_destroy(kq_engine* e);

/* cache.Snapshot (snapshot.go:171): upload a full snapshot into HBM. The engine copies; the
 * caller may free its arrays on return. */
int  kq_snapshot_put(kq_engine* e, const kq_snapshot* s);

/* Incremental snapshot: the next cycle's cache.Snapshot() when the quota tree, the policies and the dictionaries are
 * the ones of the last kq_snapshot_put and only usage and / or the admitted set moved — what clusterQueue.updateWorkloadUsage
 * (pkg/cache/scheduler/clusterqueue.go:594) and updateCohortResourceNode (resource_node.go:190) change between two cycles.
 * `s` is a complete kq_snapshot image as for kq_snapshot_put; only the parts named by `what` are read (plus sizes and `parent`,
 * which must match). KQ_PATCH_USAGE: the usage plane [N * n_fr] replaces the resident one (0.57 MB at cfg 3, nothing rebuilt).
 * KQ_PATCH_ADMITTED: the admitted-workload table (cq_adm_off, adm_*) replaces the resident one and the engine's candidate
 * structures are rebuilt from it; implies KQ_PATCH_USAGE. Resident head batches and the pending set stay valid; folded commits
 * (kq_cycle_commit) are forgotten: the new plane is the truth. */
#define KQ_PATCH_USAGE     0x1u
#define KQ_PATCH_ADMITTED  0x2u
int  kq_snapshot_patch(kq_engine* e, const kq_snapshot* s, uint32_t what);

/* The O(changes) form of KQ_PATCH_ADMITTED: what clusterQueue.updateWorkloadUsage (pkg/cache/scheduler/clusterqueue.go:594) does to
 * cq.Workloads between two cycles — some admitted workloads left (finished, evicted and gone), some arrived (admitted by the last
 * cycles) — handed over as the rows that left and the rows that came. The resident row table is compacted and extended ON THE DEVICE
 * (removed rows drop out, the added rows land behind the kept rows of their ClusterQueue, in the order given) and every structure
 * derived from it — candidate rank order per tree, flavor-resource buckets, level orders, row records — is rebuilt there by key
 * sorts (kueue_amd/csrc/kq_rows.hpp). After a cycle: add = the heads it admitted (usage = their assignment), evict_rows = the targets of
 * its preemptions, remove_rows = the workloads that finished; nothing of the admitted table crosses PCIe but the added rows. The result is byte-identical to
 * kq_snapshot_put of the snapshot with that row table (tests/test_rows_device.py). Row indices of later calls (remove_rows,
 * kq_decisions.tgt_adm, kq_heads.slice_row) refer to the new table: kept rows keep their order, `new_index` (optional, [n rows before
 * the call]) receives the new index of every old row, -1 for a removed one.
 * Usage is NOT touched (unless flags has KQ_ROWS_FOLD_USAGE): fold it with kq_cycle_commit / kq_cycle_release or KQ_PATCH_USAGE as before. add_uid_rank must be comparable
 * with the resident rows' adm_uid_rank (any order-preserving 32-bit key of Obj.UID works; dense ranks do not survive insertions).
 * KQ_EUNSUPPORTED (use kq_snapshot_patch): amounts outside the plain range, sizes beyond the sort keys' fields (2^20 rows, 2^21 nodes). */
typedef struct kq_row_patch {
  int32_t n_remove;
  const int32_t* remove_rows;       /* [n_remove] distinct rows of the resident table, any order */
  int32_t n_add;
  const int32_t* add_cq;            /* [n_add] */
  const int64_t* add_priority;      /* [n_add] as kq_snapshot.adm_priority ... */
  const int64_t* add_queue_ts;
  const int64_t* add_reserve_ts;
  const uint32_t* add_uid_rank;
  const uint8_t* add_flags;
  const int32_t* add_use_off;       /* [n_add + 1] */
  const int32_t* add_use_fr;
  const int64_t* add_use_qty;
  int32_t n_evict;                  /* rows (of the resident table, not among remove_rows) that get KQ_ADM_EVICTED: the targets of the last cycle's */
  const int32_t* evict_rows;        /* preemptions stay admitted until they terminate, marked Evicted (preemption.go IssuePreemptions)            */
  uint32_t flags;                   /* KQ_ROWS_* */
} kq_row_patch;
/* clusterQueue.updateWorkloadUsage (clusterqueue.go:594) in full: the usage of the removed rows leaves the snapshot (removeUsage
 * resource_node.go:156), the usage of the added rows enters it (addUsage :144) — on the device, from the rows' own usage entries; with a
 * pending set resident, the removal is the event that sends the inadmissible workloads of the freed root cohorts back to their heaps
 * (QueueAssociatedInadmissibleWorkloadsAfter, inadmissible_workloads.go:112-147), as kq_cycle_release does. The caller then does NOT
 * kq_cycle_commit the cycle whose admissions it adds as rows. This is the closed loop of a population with preemption
 * (kueue_amd/closed_loop.py, shim/go/closed_loop.go): add = the heads the cycle admitted, evict_rows = its preemption targets,
 * remove_rows = the targets of the cycle before (terminated) + the workloads that finished. Rows of more than 56 usage entries:
 * KQ_EUNSUPPORTED. */
#define KQ_ROWS_FOLD_USAGE 0x1u
int  kq_snapshot_patch_rows(kq_engine* e, const kq_row_patch* p, int32_t* new_index);

/* One scheduling cycle: nominate + iterator + processEntry (scheduler.go:308-386, steps 3-5).
 * Synchronous. `out` arrays are caller-allocated, sized from `h`. The uploaded snapshot is left
 * unchanged (the reference mutates a per-cycle copy). */
int  kq_cycle_run(kq_engine* e, const kq_heads* h, kq_decisions* out);

/* Device-resident heads ("pending side on device"): upload a batch of heads once
 * (pkg/cache/queue/manager.go:903 Heads, pre-digested workload.Info records, workload.go:245) and run
 * cycles on it without any host->device input traffic. `batch` is a small caller-chosen id. */
int  kq_heads_put(kq_engine* e, const kq_heads* h, int32_t batch);
int  kq_cycle_run_resident(kq_engine* e, int32_t batch, kq_decisions* out);

/* Nominate-ahead over a resident batch ("nominate-all-pending"): what Scheduler.nominate (scheduler.go:665-705)
 * computes for every head of the batch — getAssignments :821 = flavorassigner.Assign (flavorassigner.go:696) +
 * preemption.GetTargets (preemption.go:132) + the partial-admission search — against the resident snapshot, WITHOUT the entry
 * iterator and processEntry. The batch may hold any number of workloads per ClusterQueue (all W pending). Fills the
 * nomination fields of `out` (nominated_mode = mode, borrowing, flavor / res_mode / tried_idx / ps_count, targets); status is
 * KQ_ST_NOT_NOMINATED and order -1 for every head. Nothing is committable afterwards. */
int  kq_nominate_run_resident(kq_engine* e, int32_t batch, kq_decisions* out);

/* ---- pending side on the device: pkg/cache/queue ---------------------------------------------
 * Every pending workload of every ClusterQueue lives in HBM; Heads() (manager.go:903,922: one ClusterQueue.Pop per
 * ClusterQueue, cluster_queue.go:657) is a segmented arg-min under baseCompareFunc (cluster_queue.go:844: sticky preemptor,
 * priority descending, queue-order timestamp ascending, UID ascending) and the requeue policy (RequeueIfNotPresent :826,
 * requeueIfNotPresent :550, handleInadmissibleHash :606, queueInadmissibleWorkloads inadmissible_workloads.go:149) runs on the
 * device from the cycle's decisions, together with the LastAssignment bookkeeping of scheduler.go:248-295,459-464.
 * Outside: namespace selectors, second-pass queue. */
#define KQ_WL_ACTIVE        0  /* in the heap                                   */
#define KQ_WL_INFLIGHT      1  /* popped by Heads(), decision pending           */
#define KQ_WL_INADMISSIBLE  2  /* parked in inadmissibleWorkloads               */
#define KQ_WL_GONE          3  /* admitted (left the queue)                     */
typedef struct kq_pending {
  kq_heads w;                 /* n = W workloads: the pre-digested workload.Info columns of a heads batch, any order;
                                 w.cycle is ignored */
  const uint32_t* uid_rank;   /* [W] rank of Obj.UID (cluster_queue.go:873) */
  /* AdmissionFairSharing ordering (queueOrderingFunc cluster_queue.go:880-904): in a ClusterQueue whose AdmissionScope is
   * UsageBasedAdmissionFairSharing the heap compares the LocalQueues' fair-sharing usage first (lower first, Go cmp.Compare on
   * float64), then baseCompareFunc. lq[w] = index of the workload's LocalQueue, -1 = ordering by baseCompareFunc only (the
   * ClusterQueue has no AFS). NULL / n_lq = 0: no AFS anywhere. The usage values (afs.CalculateUsage: consumed + pending penalty,
   * resource weights, LocalQueue weight - pkg/util/admissionfairsharing/admission_fair_sharing.go:86) are the host's:
   * kq_pending_set_lq_usage before Heads(). */
  int32_t n_lq;
  const int32_t* lq;          /* [W] or NULL */
  /* Back-off after a PodsReady timeout (backoffWaitingTimeExpired cluster_queue.go:474-485): per workload RequeueState.RequeueAt in
   * ns, KQ_REQUEUE_NONE when there is none, KQ_REQUEUE_BLOCKED while the Requeued condition is False. A workload whose back-off has
   * not expired (against kq_pending_set_clock) waits among the inadmissible workloads (:414, :568, inadmissible_workloads.go:167).
   * NULL: no workload is backing off. */
  const int64_t* requeue_at;  /* [W] or NULL */
  /* kq_pending_update only ([n] or NULL): 1 = the replacement object carries the Generation the key was set with — a status-only
   * update (ReclaimablePods, the Evicted / Requeued conditions): if the key is the ClusterQueue's preemptor, IsPreemptor
   * (cluster_queue.go:213: name AND Obj.Generation) keeps holding. 0 / NULL: the spec changed, only stickyMatches (:124) holds. */
  const uint8_t* same_generation;
} kq_pending;
#define KQ_REQUEUE_NONE    INT64_MIN
#define KQ_REQUEUE_BLOCKED INT64_MAX
/* PushOrUpdate (cluster_queue.go:379) of every workload into its ClusterQueue's heap; replaces any previous pending set.
 * Workload slices (ElasticJobsViaWorkloadSlices): p->w may carry the slice_* columns — here, in kq_pending_add and in kq_pending_update;
 * they stay resident with the workloads, travel into every Heads() batch, and kq_snapshot_patch_rows moves the slice_row of every pending
 * workload with the admitted table (the slice that left the table is simply gone: the head is then an ordinary workload,
 * workloadslicing.go:371). */
int  kq_pending_put(kq_engine* e, const kq_pending* p);
/* Heads(): pops <= 1 workload per ClusterQueue (cq_active[c] == 0: ClusterQueue skipped, manager.go:926; NULL = all active)
 * and gathers them, in canonical head order (ClusterQueue index ascending), into the engine's resident pending batch.
 * *n_heads / *n_podsets size the kq_decisions arrays of the cycle; head_wl (optional, [n_cq]) receives the workload of every
 * head (index into kq_pending.w). */
int  kq_pending_heads(kq_engine* e, int64_t cycle, const uint8_t* cq_active, int32_t* n_heads, int32_t* n_podsets, int32_t* head_wl);
/* kq_cycle_run over the batch kq_pending_heads just built (no head crosses PCIe). */
int  kq_cycle_run_pending(kq_engine* e, kq_decisions* out);
/* Step 6 of schedule() (scheduler.go:362-377) for the heads of that cycle: admitted workloads leave the queue, the others go
 * through RequeueIfNotPresent with the decision's requeue reason; their LastAssignment becomes the cycle's tried indices. */
int  kq_pending_apply(kq_engine* e);
/* ComputeLocalQueueFSUsage of every LocalQueue (workload.go:492) as the ledger stands now; read by the next kq_pending_heads. */
int  kq_pending_set_lq_usage(kq_engine* e, int32_t n_lq, const double* usage);
/* ---- the pending loop without a host round trip inside a cycle (one cycle = one enqueue) --------------------------------------
 * kq_pending_step enqueues kq_pending_heads -> kq_cycle_run_pending -> kq_cycle_commit -> kq_pending_apply (-> kq_cycle_release(
 * release_age) when release_age > 0) on the engine's stream and returns WITHOUT waiting: the head count never leaves the device (every
 * array and grid is sized by the bound of kq_pending_bounds: <= 1 head per ClusterQueue, the widest workload of every ClusterQueue),
 * and the decisions land in pinned host memory behind an event. kq_pending_step_wait blocks until the OLDEST step in flight is done
 * and unpacks its decisions; at most two steps may be in flight, so the host enqueues cycle i+1 while the device runs cycle i
 * (schedule()'s own side effects — the API writes of admit / requeueAndUpdate — trail the cycle in the reference as well,
 * scheduler.go:362-377 runs them from the decisions, not the other way round).
 * `out` of kq_pending_step_wait is sized for the bound (kq_pending_bounds). Reason records (the operands of Status.reasons /
 * the "couldn't assign flavors" messages) are produced by the steps issued after kq_pending_step_reasons(e, rsn_cap > 0): their
 * windows are staged to the host with the decisions, and kq_pending_step_wait fills out->rsn_* exactly as kq_cycle_run does when
 * out->rsn_cap > 0 (KQ_ECAPACITY when it is too small); rsn_cap = 0 switches them off again (the default: a step without them copies
 * 32 B x the window of every head less). tgt_cap as in kq_decisions.
 * A step whose cycle fails on the device (KQ_ECAPACITY of the target pool, ...) commits nothing and puts its heads back into their
 * heaps; kq_pending_step_wait returns the error for that step. KQ_ECAPACITY from the unpack (out->tgt_cap too small for the target
 * CSR) leaves the step applied on the device.
 * Not to be mixed with kq_pending_heads / kq_pending_apply while a step is in flight. */
int  kq_pending_bounds(kq_engine* e, int32_t* max_heads, int32_t* max_podsets);
int  kq_pending_step(kq_engine* e, int64_t cycle, const uint8_t* cq_active, int32_t tgt_cap, int32_t release_age, int32_t want_head_wl);
int  kq_pending_step_wait(kq_engine* e, kq_decisions* out, int32_t* n_heads, int32_t* n_podsets, int32_t* head_wl);
int  kq_pending_step_reasons(kq_engine* e, int32_t rsn_cap);

/* ---- AdmissionFairSharing ledger on the device (pkg/cache/queue/afs/usage_ledger.go, entry_penalties.go) --------------------
 * With a ledger resident, the LocalQueues' fair-sharing usage that Heads() orders by is evaluated on the device from the ledger, and the
 * scheduler's own write to it — the entry penalty pushed when a workload is assumed (scheduler.go:1064-1068 assumeWorkload ->
 * updateEntryPenalty :1337-1355 -> AfsUsageLedger.PushPenalty entry_penalties.go:30) — happens in kq_pending_apply, so a resident loop
 * needs no host round trip between cycles. kq_pending_set_lq_usage is refused while a ledger is resident.
 * Amounts are EXACT: integers in units of 10^-9 of the resource's base unit (resource.Quantity has no finer precision; MulByFloat
 * rounds to that scale, pkg/util/resource/resource.go:93-115), 128 bits as (lo, hi) two's complement words.
 * Resource dictionary of the ledger: every resource name that can appear in a consumed list or a penalty, SORTED BY NAME
 * (afs.CalculateUsage sums in sorted key order, admission_fair_sharing.go:86-103); n_res <= 64.
 * One LocalQueue feeds exactly one ClusterQueue (so the <= 1 head per ClusterQueue of a cycle never share a ledger row). */
typedef struct kq_afs_ledger {
  int32_t n_lq, n_res;
  const double* lq_weight;        /* [n_lq] afs.LQWeightAsFloat64; <= 0: usage is +Inf (:98-100) */
  const double* res_weight;       /* [n_res] fsResWeights[name], 1 when the name is not listed (:92-95) */
  /* entry.Resources (the decayed consumed history): the amount, and its AsApproximateFloat64 AS THE LEDGER HOLDS IT (a parsed "8"
   * and a decayed 8.000000000 are different float paths, quantity.go:468-483); NULL f64: every amount is in the scale-9 form */
  const uint64_t* consumed_lo; const int64_t* consumed_hi; const double* consumed_f64;   /* [n_lq * n_res] */
  /* entry.pendingPenalty at upload time (penalties of workloads that are not in the pending set any more); NULL: none.
   * present[i]: the key exists in the aggregate (a zero amount still moves the sum to scale 9) */
  const uint64_t* penalty_lo; const int64_t* penalty_hi; const uint8_t* penalty_present; /* [n_lq * n_res] */
  /* what PushPenalty would record for pending workload w (afs.CalculateEntryPenalty(SumTotalRequests, config),
   * admission_fair_sharing.go:53-60; scheduler.go:1343-1348), key set as a bit mask over the ledger's resources */
  const uint64_t* wl_penalty_lo; const int64_t* wl_penalty_hi;   /* [W * n_res] */
  const uint64_t* wl_penalty_mask;                                /* [W] */
} kq_afs_ledger;
/* Installs the ledger for the resident pending set (kq_pending_put with LocalQueue indices first). Workloads appended later by
 * kq_pending_add carry no penalty until kq_pending_afs_wl_penalty gives them one. */
int  kq_pending_afs_put(kq_engine* e, const kq_afs_ledger* l);
int  kq_pending_afs_wl_penalty(kq_engine* e, int32_t n, const int32_t* wl, const uint64_t* lo, const int64_t* hi, const uint64_t* mask);
/* AfsUsageLedger.SubPenalty (entry_penalties.go:45) for pending-set workloads: rollback of a failed admission (scheduler.go:1032),
 * deletion (workload_controller.go:1296,1475,1481). No record: nothing happens. */
int  kq_pending_afs_sub_penalty(kq_engine* e, int32_t n, const int32_t* wl);
/* A controller rewrote entry.Resources of LocalQueues lq[i] (LocalQueue reconciler decay, localqueue_controller.go:223; settlement
 * workload_controller.go:1506-1528). settle_wl (optional): workload whose recorded penalty is folded in the same write —
 * remaining, penalty := old.WithoutPenalty(wlKey); Resources = MergeResourceListKeepSum(newConsumed, penalty) — or -1. */
int  kq_pending_afs_set_consumed(kq_engine* e, int32_t n, const int32_t* lq, const uint64_t* lo, const int64_t* hi, const double* f64,
                                 const int32_t* settle_wl);
/* Read back: usage[n_lq] = afs.CalculateUsage per LocalQueue as Heads() would see it now; penalty lo/hi/present [n_lq * n_res];
 * wl_record[W]; any pointer may be NULL. */
int  kq_pending_afs_read(kq_engine* e, double* usage, uint64_t* penalty_lo, int64_t* penalty_hi, uint8_t* penalty_present,
                         uint64_t* consumed_lo, int64_t* consumed_hi, uint8_t* wl_record);
/* queueInadmissibleWorkloads for the listed ClusterQueues (cq == NULL: all) — what requeueWorkloadsCohort does for the root
 * cohorts whose quota was freed (inadmissible_workloads.go:112-175). */
int  kq_pending_queue_inadmissible(kq_engine* e, int32_t n, const int32_t* cq);
/* PushOrUpdate (cluster_queue.go:379-428) of workloads that were not pending before: they are APPENDED to the resident set — the
 * first one gets index *first_index (= the previous count), existing indices do not move — and merged into the heap order of their
 * ClusterQueues. A new workload starts in the heap, or among the inadmissible when its ClusterQueue is BestEffortFIFO, its
 * scheduling hash is known and its equivalence class was bulk-moved since the ClusterQueue's last queueInadmissibleWorkloads
 * (:419-425, hashToBulkMoveReason). `more` has the layout of kq_pending_put's argument (LocalQueue indices iff the set has them).
 * Not between kq_pending_heads and kq_pending_apply. */
int  kq_pending_add(kq_engine* e, const kq_pending* more, int32_t* first_index);
/* PushOrUpdate (cluster_queue.go:379-428) of workloads that ARE pending, with a new object (priority, timestamps, podsets, scheduling
 * hash ... changed): wl[i] is replaced by the i-th workload of `more` (layout of kq_pending_add's argument, more->w.n == n, the same
 * uid_rank as the workload it replaces). Records are variable-sized and indices append-only, so the replacement gets the index
 * *first_index + i and wl[i] leaves the set; what the reference keys by NAME follows it:
 *   - wl[i] in the heap: PushOrUpdateActive (:427) - the replacement is in the heap whatever its back-off or its class (:414 and
 *     :421 only look at keys with GetActive(key) == nil);
 *   - wl[i] inadmissible: RemoveFromInadmissible (:405), then placed like an arrival (back-off :414, bulk-moved class :419-425).
 *     The in-place branch (:396-403, nothing that matters changed: UpdateInadmissible) needs no call at all;
 *   - wl[i] gone: a plain arrival. In flight cannot be (:388): not between kq_pending_heads and kq_pending_apply;
 *   - the ClusterQueue's preemptor pointer (:109 holds a name) moves to the replacement: stickyMatches (:124) still sorts it first,
 *     IsPreemptor (:213, strict: name AND Obj.Generation) holds only if more->same_generation[i] says the generation did not change
 *     (a status-only update); a replacement in another ClusterQueue clears it (Delete :506);
 *   - AdmissionFairSharing: the replacement's entry-penalty amounts are set like an arrival's (kq_pending_afs_wl_penalty); a penalty
 *     RECORD only exists for assumed workloads, which are no longer pending.
 * Cost: kq_pending_add of the replacements + one small launch. GROWTH: the replaced records stay in every column (marked gone) and in
 * the ClusterQueues' order segments until the next kq_pending_put; Heads(), the arrival merge and queueInadmissibleWorkloads scan a
 * ClusterQueue's segment linearly, so a resident loop with frequent updates should re-put the pending set once the gone records
 * (kq_pending_read_state: counts[3]) outnumber the live ones. */
int  kq_pending_update(kq_engine* e, int32_t n, const int32_t* wl, const kq_pending* more, int32_t* first_index);
/* ClusterQueue.Delete (cluster_queue.go:488-512): the workloads leave the pending set (deleted, finished, admitted by another
 * scheduler). Not between kq_pending_heads and kq_pending_apply. */
int  kq_pending_delete(kq_engine* e, int32_t n, const int32_t* wl);
/* c.clock.Now() of the queues for every later call (ns; 0 until set). */
int  kq_pending_set_clock(kq_engine* e, int64_t now_ns);
/* The workload controller changed RequeueState / the Requeued condition of pending workloads: PushOrUpdate of an existing workload
 * whose conditions changed (cluster_queue.go:391-428). An inadmissible workload leaves the inadmissible set unless its back-off has
 * still not expired or its equivalence class is bulk-moved; a workload in the heap stays there. */
int  kq_pending_set_requeue_at(kq_engine* e, int32_t n, const int32_t* wl, const int64_t* requeue_at);
/* state[W] (KQ_WL_*) and counts[4] per state; both optional. */
int  kq_pending_read_state(kq_engine* e, uint8_t* state, int32_t* counts);

/* Closed-loop driver support ("a run" applies decisions to the snapshot between cycles).
 * kq_cycle_commit folds the usage of every workload the LAST cycle admitted into the resident snapshot — what
 * cache.AssumeWorkload leaves in the cache (pkg/cache/scheduler/clusterqueue.go:594 updateWorkloadUsage ->
 * resource_node.go:144 addUsage) — and remembers it; kq_cycle_release removes, via removeUsage (:156), what the
 * commit `age` commits ago added (age = 1: the latest), i.e. those workloads finish. Preempt-mode reservations
 * and DeferredFit usage are per-cycle simulation state and are not committed. The admitted-workload table used for
 * preemption candidates is not extended; callers that need it re-upload the snapshot.
 * *n_admitted (optional) receives the number of workloads folded in. Ring depth KQ_COMMIT_RING.
 * With a pending set resident (kq_pending_put), a release that frees quota also does what the cache's notification does in the
 * reference (QueueAssociatedInadmissibleWorkloadsAfter -> requeueWorkloadsCohort, inadmissible_workloads.go:112-147): every ClusterQueue
 * under the ROOT cohort of a ClusterQueue whose workloads finished runs queueInadmissibleWorkloads. A release of a commit that
 * admitted nothing is no event and requeues nothing. */
#define KQ_COMMIT_RING 32
int  kq_cycle_commit(kq_engine* e, int32_t* n_admitted);
int  kq_cycle_release(kq_engine* e, int32_t age);

/* ---- one root cohort tree split across GPUs -------------------------------------------------
 * Shards of one root tree (unions of the root's child subtrees) share only the ROOT's usage row: usage bubbles along the path
 * (resource_node.go:144-165). Every rank runs the cycle over the heads of ITS ClusterQueues against the same cycle-start
 * snapshot; then
 *   kq_cycle_certificate  usage_delta_dev[N * n_fr] (a buffer in the engine's device memory: the RCCL send buffer) = what this
 *                         rank's cycle added to every usage cell; root_margin[n_tree * n_fr] (host) = for every flavor-resource the
 *                         smallest slack an admitted entry of this rank had in the root term of Available (resource_node.go:106-122);
 *                         flags[n_tree] (host) != 0: some entry changed usage outside the certificate (preemption targets,
 *                         recomputation, non-plain amounts, negative reservation, fair sharing).
 *   all-reduce(sum, int64) of the deltas over the ranks. If on every rank, for every flavor-resource, the root usage the OTHER
 *   ranks added is within that rank's slack and no flag is set, every decision is the one the unsharded cycle takes (an admitted
 *   entry still fits with the others' usage in front of it; a rejected one stays rejected because Available only shrinks). Otherwise
 *   the ranks run the whole cycle redundantly (kq_cycle_run over all heads).
 *   kq_snapshot_usage_add folds the reduced ClusterQueue-level delta ([n_cq * n_fr], device memory) into the resident snapshot
 *   (sign +1) — or takes it out again when those workloads finish (sign -1); cohort usage follows from it. */
int  kq_cycle_certificate(kq_engine* e, int64_t* usage_delta_dev, int64_t* root_margin, int32_t* flags);

/* ---- sharded nominate, merged process --------------------------------------------------------
 * nominate (scheduler.go:665: flavor assignment + victim search per head) only reads the cycle-start snapshot, so the heads of a
 * cycle shard over the ranks with no exchange; processEntry (scheduler.go:392) is one dependency chain per root cohort and cheap
 * once it runs as speculative rounds (kq_spec.hpp). Every rank holds the same snapshot and the same batch of heads:
 *   kq_cycle_shard_words     size (int64 words) of the exchange buffer for this batch, these decision capacities and `world` ranks
 *   kq_cycle_nominate_shard  nominate the heads with mine[h] != 0 (host array [n]; NULL: all) and write their nomination into
 *                            xbuf_dev (device memory, e.g. the RCCL buffer). Words of the other ranks' heads are ZERO.
 *   all-reduce(SUM, int64) of xbuf over the ranks — the one collective of the cycle — merges the shards
 *   kq_cycle_process_merged  import the merged nomination of ALL heads, then iterator order + processEntry + decisions, exactly as
 *                            kq_cycle_run does after its own nominate. Identical on every rank: kq_cycle_commit / _release keep the
 *                            resident snapshots in step, no fallback path, preemption and fair sharing included.
 * `out` carries the capacities (tgt_cap per rank, rsn_cap) in both calls and receives the decisions in the second. */
int  kq_cycle_shard_words(kq_engine* e, const kq_heads* h, const kq_decisions* out, int32_t world, int64_t* words);
int  kq_cycle_nominate_shard(kq_engine* e, const kq_heads* h, const uint8_t* mine, int32_t world, int32_t rank, void* xbuf_dev, kq_decisions* out);
int  kq_cycle_process_merged(kq_engine* e, int32_t world, int32_t rank, const void* xbuf_dev, kq_decisions* out);
int  kq_snapshot_usage_add(kq_engine* e, const int64_t* delta_dev, int32_t sign);

/* Per-kernel device time of the last cycle, HIP events on the engine's stream:
 * phase_ms[0] nominate, [1] order, [2] process; phase_bytes[0] nominate, [1] process (algorithmic bytes). */
int  kq_last_cycle_phases(kq_engine* e, double* phase_ms, int64_t* phase_bytes);

/* Device time (ms, hipEvent) of the kernels of the last kq_cycle_run, and its algorithmic bytes. */
int  kq_last_cycle_stats(kq_engine* e, double* kernel_ms, int64_t* algorithmic_bytes);

/* Recompute SubtreeQuota of every node and Usage of every Cohort from Quotas + CQ usage, as
 * updateCohortResourceNode / accumulateFromChild do (resource_node.go:183-230). Operates on the
 * uploaded snapshot; results readable through kq_snapshot_read_planes. */
int  kq_snapshot_derive(kq_engine* e);
int  kq_snapshot_read_planes(kq_engine* e, int64_t* subtree_quota, int64_t* usage, uint8_t* quota_flags);

/* Diagnostics. NOT part of the drop-in boundary (a Go binding does not need them); the parity tests use them.
 * kq_debug_read_usage_work: the cycle's private usage plane [N * n_fr] as processEntry left it (what the reference's
 * per-cycle Snapshot holds after schedule() returns). kq_debug_force_exact_drs: take the saturation-safe per-cell DRS loops
 * even when the maintained sums would be exact. kq_debug_prof: 64 in-kernel segment counters (KQ_PROF builds); `out64` must hold 64 int64. */
int  kq_debug_read_usage_work(kq_engine* e, int64_t* usage_out);
int  kq_debug_force_exact_drs(kq_engine* e, int on);
int  kq_debug_disable_scan_search(kq_engine* e, int on);  /* classical victim searches walk candidate by candidate */
/* The admitted-row candidate structures (rank order per tree, flavor-resource buckets, level orders, row records ...) rebuilt on the device
 * from the resident row table, and read back one by one (`which`: kueue_amd/csrc/kq_host.hpp read_rows; *bytes: capacity in, size out;
 * KQ_ECAPACITY with the size when too small): tests compare the device-built structures with the host-built ones byte for byte. */
int  kq_debug_rows_rebuild(kq_engine* e);
/* KQ_GUARD=1 in the environment when the engine is created: every device buffer lies between two 256-byte guard zones; this reads them
 * all back. out3: [0] buffers checked, [1] buffers with a damaged guard (kq_last_error names them), [2] guard bytes read.
 * KQ_EUNSUPPORTED without KQ_GUARD. */
int  kq_debug_check_guards(kq_engine* e, int64_t* out3);
int  kq_debug_read_rows(kq_engine* e, int32_t which, void* out, int64_t* bytes);
int  kq_debug_prof(kq_engine* e, int64_t* out64, int reset);
/* last cycle's speculative process rounds (kq_spec.hpp): [0] windows, [1] rounds, [2] entries they decided, [3] trees handed (partly)
 * back to the serial kernel, [4] (entry, flavor-resource) items, [5] most rounds of one window, [6] abandoned windows, [7] truncated */
int  kq_debug_spec_stats(kq_engine* e, int64_t* out8);

const char* kq_strerror(int code);
const char* kq_last_error(kq_engine* e);
int  kq_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KQ_ENGINE_H */

/* kq_cycle_tas.h — the TAS side input / output of a scheduling cycle: Topology-Aware Scheduling INSIDE kq_cycle_run.
 *
 * kq_cycle_run_tas replaces the same call as kq_cycle_run ((*Scheduler).schedule, pkg/scheduler/scheduler.go:226) for a cycle whose
 * ClusterQueues list TAS ResourceFlavors. On top of kq_cycle_run it does, on the device:
 *   - Assign's TAS step (flavorassigner.go:864-903 assignTAS; tas_flavorassigner.go:37-83 WorkloadsTopologyRequests; clusterqueue_snapshot.go:204
 *     FindTopologyAssignmentsForWorkload -> tas_flavor_snapshot.go:578 FindTopologyAssignmentsForFlavor), normal and simulate-empty,
 *     for every Assign of the head, also inside the partial-admission search;
 *   - the TAS-aware workloadFits of the victim search (preemption.go:669-684): a candidate's TopologyDomainRequests leave the leaf
 *     usage with its quota, the placement is re-run per step;
 *   - updateAssignmentForTAS (scheduler.go:941-985), the TAS part of ClusterQueueSnapshot.Fits / AddUsage (clusterqueue_snapshot.go:107-149)
 *     inside processEntry, and the recomputation of scheduler.go:707-769 (needsTASRecompute, features.TASRecomputeAssignmentWithinSchedulingCycle).
 * The oracle (oracle/kq_oracle.cpp kqo_cycle_run_tas) consumes the same two structs.
 * Host side, as for include/kq_tas.h: topology trees, node feasibility (leaf_ok), level-key resolution per TAS flavor,
 * checkPodSetAndFlavorMatchForTAS (tas_flavorassigner.go:164 -> folded into kq_heads.ps_flavor_ok like taints and affinity).
 * A failed placement leaves a KQ_RSN_TAS_FAILURE reason record (kq_engine.h) when out->rsn_cap > 0.
 * With fair sharing (kq_config.fair_sharing) the entries are walked in the fair-sharing iterator's order — one iterator over every root
 * tree (fair_sharing_iterator.go:47-263), the lowest ClusterQueue index still waiting naming the tree that pops next — and fair
 * preemption's victim search (preemption.go:381-631) carries the leaf usage with the victims.
 * Heads that hold an admission (the second pass: a delayed topology request, or a node failure -> replacement) are part of the cycle:
 * see ps_adm_flavor / ps_ex_* below.
 * Outside (KQ_EUNSUPPORTED or left to the caller): a workload whose podsets land on more than one TAS
 * flavor (TASHandleOverlappingFlavors), TASRespectNodeAffinityPreferred. (TASBalancedPlacement: kq_tas_topology.profile_mixed, include/kq_tas.h.)
 */
#ifndef KQ_CYCLE_TAS_H
#define KQ_CYCLE_TAS_H
#include <stdint.h>
#include "kq_engine.h"
#include "kq_tas.h"
#ifdef __cplusplus
extern "C" {
#endif

#define KQ_PS_TAS_EXPLICIT 1u   /* workload.IsExplicitlyRequestingTAS(podSet) */

#define KQ_CT_NO_RECOMPUTE 1u    /* features.TASRecomputeAssignmentWithinSchedulingCycle off (default on) */
#define KQ_CT_NO_FAIL_FAST 2u    /* features.TASFailedNodeReplacementFailFast off (default on) */

#define KQ_EX_UNHEALTHY 1u       /* the domain's node is one of Status.UnhealthyNodes (PodSetAssignment.HasUnhealthyNode tas_flavorassigner.go:85) */
#define KQ_EX_FIRST     2u       /* ... and it is UnhealthyNodes[0]: the domain deleteDomain takes out (tas_flavor_snapshot.go:693, :828);
                                  * at most one domain per podset (a node's name is unique among the hostname-level domains) */

typedef struct kq_cycle_tas {
  uint32_t flags;                   /* KQ_CT_* */
  int32_t n_tas;                    /* TAS ResourceFlavors with a cached topology (ClusterQueueSnapshot.TASFlavors, snapshot.go:260) */
  const int32_t* tas_flavor;        /* [n_tas] index in the snapshot's flavor dictionary; ascending by flavor NAME (slices.Sorted,
                                       clusterqueue_snapshot.go:220) */
  const kq_tas_topology* topo;      /* [n_tas] all over ONE resource dictionary (n_resources, pods_resource equal); tas_usage holds only
                                       usage that belongs to no admitted row below (the rows' usage is added from the CSR) */
  const uint8_t* cq_tas_only;       /* [n_cq] clusterQueue.isTASOnly clusterqueue.go:746 */
  /* admitted workloads: workload.TASUsage() as TopologyDomainRequests, CSR over the snapshot's admitted rows */
  const int32_t* adm_off;           /* [n_adm+1] */
  const int32_t* adm_tas;           /* [..] index into tas_flavor */
  const int32_t* adm_leaf;          /* [..] */
  const int32_t* adm_count;         /* [..] */
  const int64_t* adm_req;           /* [..][n_resources] SinglePodRequests, dense (0 = absent) */
  /* pending heads: one record per podset of kq_heads (global podset index) */
  const uint8_t* ps_flags;          /* [n_ps] KQ_PS_TAS_EXPLICIT */
  const uint8_t* ps_kind;           /* [n_ps] KQ_TAS_REQUIRED / PREFERRED / UNCONSTRAINED (implied requests: UNCONSTRAINED) */
  const int32_t* ps_level;          /* [n_ps][n_tas] levelKeyWithImpliedFallback :1212 resolved against each TAS flavor, -1 = absent */
  const int32_t* ps_slice_size;     /* [n_ps] */
  const int32_t* ps_slice_level;    /* [n_ps][n_tas] */
  const int32_t* ps_group;          /* [n_ps] PodSetGroupName id, -1 = none */
  const int64_t* ps_req;            /* [n_ps][n_resources] SinglePodRequests from the pod spec (tas_flavorassigner.go:116) */
  /* TASMultiLayerTopology: utiltas.PodSetSliceRequiredTopologyConstraints of the podset, all layers, outermost first (layer 0 repeats
   * ps_slice_size / ps_slice_level); NULL = no podset carries more than one layer. Same meaning as kq_tas_requests.n_layers / layer_*. */
  const int32_t* ps_n_layers;       /* [n_ps] */
  const int32_t* ps_layer_level;    /* [n_ps][n_tas][KQ_TAS_MAX_LEVELS] resolved against each TAS flavor, -1 = absent */
  const int32_t* ps_layer_size;     /* [n_ps][KQ_TAS_MAX_LEVELS] */
  /* Second pass (workload.NeedsSecondPass workload.go:974; heads with KQ_HEAD_HAS_QUOTA_RESERVATION come first, manager.go:923,
   * scheduler.go:1114): what Status.Admission holds for the head's podsets. All NULL = no head holds an admission.
   *   ps_adm_flavor: PodSetAssignments[i].Flavors — Assign keeps them (flavorassigner.go:768-774: mode Fit, no flavor scan for those
   *     resources); the head consumes no new quota (netUsage scheduler.go:785-794);
   *   ps_ex_*: PodSetAssignments[i].TopologyAssignment, every domain in the assignment's order, resolved to a leaf of the podset's TAS
   *     flavor (the TAS flavor among ps_adm_flavor; -1 = the snapshot has no such leaf, IsTopologyAssignmentStale :818). A podset that
   *     holds one is placed again only when it names an unhealthy node (WorkloadsTopologyRequests tas_flavorassigner.go:50), and then
   *     through findReplacementAssignment (tas_flavor_snapshot.go:608-633, :686): the pods of UnhealthyNodes[0]'s domain are placed below
   *     the required replacement domain and merged into the rest; the entry's Usage.TAS is what the new assignment holds beyond the
   *     admitted one, per domain (ComputeTASNetUsage flavorassigner.go:106-155). A replacement that fails evicts the workload
   *     (KQ_ST_EVICTED / KQ_ACT_EVICT) unless KQ_CT_NO_FAIL_FAST. SkipReassignmentForPodOwnedWorkloads (:615) is the caller's: such a
   *     workload is not submitted. */
  const int32_t* ps_adm_flavor;     /* [n_ps][n_resource of the snapshot] flavor index, -1 = none */
  const int32_t* ps_ex_off;         /* [n_ps+1] */
  const int32_t* ps_ex_leaf;
  const int32_t* ps_ex_count;
  const uint8_t* ps_ex_flags;       /* KQ_EX_* */
  /* Node feasibility of a podset on a TAS flavor — the node's NoSchedule / NoExecute taints against the podset's and the flavor's
   * tolerations, PodSpec.NodeSelector and the required node affinity against the node's labels (tas_flavor_snapshot.go:955-963,
   * fillInCounts :1893-1905): string matching, evaluated by the host exactly as for kq_tas_requests.leaf_ok, and shared here as ROWS
   * because most podsets of a cycle carry the same few tolerations. ps_mask NULL = every podset may use every leaf.
   *   ps_mask:   [n_ps][n_tas] row of leaf_mask, -1 = every leaf of that flavor
   *   leaf_mask: [n_masks][mask_stride] 1 = the leaf's node may host the podset; mask_stride >= the n_leaves of every topology a row is
   *              used with. A masked podset starts from a phase 1 of its own (no request-class table). */
  const int32_t* ps_mask;
  const uint8_t* leaf_mask;
  int32_t n_masks;
  int32_t mask_stride;
} kq_cycle_tas;

typedef struct kq_cycle_tas_out {
  int32_t* ps_tas;                  /* [n_ps] index into tas_flavor of the podset's TopologyAssignment, -1 = none */
  int32_t* dom_off;                 /* [n_ps+1] */
  int32_t* dom_leaf;
  int32_t* dom_count;
  int32_t  dom_cap;
  int64_t* tas_usage_after;         /* optional: [n_tas][n_leaves][n_resources] concatenated, leaf usage after the cycle */
} kq_cycle_tas_out;

/* One scheduling cycle with TAS inside it, against the resident snapshot (kq_snapshot_put). Same contract as kq_cycle_run for `out`
 * (decisions, targets, reasons); tout receives the TopologyAssignment of every podset that holds one (admitted or not: the nomination's)
 * and, optionally, the leaf usage after the cycle. t->adm_* is indexed by the admitted rows of the resident snapshot, t->ps_* by the
 * podsets of `h`. stats (optional, int64[4]): [0] placements computed, [1] TAS recomputations inside processEntry, [2] 1 when the cycle met a
 * workload outside the path (two TAS flavors), [3] placements of processEntry that started from a resident request-class table.
 * KQ_EUNSUPPORTED: a workload with TAS requests on more than one TAS flavor. */
int kq_cycle_run_tas(kq_engine* e, const kq_heads* h, const kq_cycle_tas* t, kq_decisions* out, kq_cycle_tas_out* tout, int64_t* stats);

#ifdef __cplusplus
}
#endif
#endif

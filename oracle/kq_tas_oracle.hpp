// kq_tas_oracle.hpp — the TAS oracle's C++ interface for the cycle oracle (kq_oracle.cpp). TEST INFRASTRUCTURE ONLY.
//
// One tas::Flavor is one TASFlavorSnapshot (pkg/cache/scheduler/tas_flavor_snapshot.go:128): the domain tree plus the per-leaf
// free capacity and TAS usage. The cycle oracle keeps one per TAS ResourceFlavor, shared by every ClusterQueue that lists the
// flavor (snapshot.go:260), and mutates its usage as the cycle admits / simulates (clusterqueue_snapshot.go:121 updateTASUsage).
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

#include "../include/kq_tas.h"

namespace tas {

struct Snapshot;
typedef std::vector<std::pair<int, int32_t>> Assignment;  // (leaf, count), leaves ascending

struct PodSetResult {  // tasPodSetAssignmentResult :424
  int status = KQ_TAS_OK;
  int32_t a = 0, b = 0;
  Assignment domains;
  std::vector<int32_t> layerFit;  // KQ_TAS_NOT_FIT_LAYERS: fit slices per layer of the constraint list
};

Snapshot* snapshot_new(const kq_tas_topology* t);
void snapshot_free(Snapshot* s);
int snapshot_resources(const Snapshot* s);
int snapshot_levels(const Snapshot* s);
int64_t snapshot_bytes(const Snapshot* s);

// FindTopologyAssignmentsForFlavor :578 for the podset requests [p0, p1) of rq, all of ONE workload. out[i] belongs to p0 + i.
void find_workload(Snapshot& s, const kq_tas_requests* rq, int p0, int p1, bool simulateEmpty, std::vector<PodSetResult>* out);
// The HasUnhealthyNodes branch of FindTopologyAssignmentsForFlavor (:608-633) for the podsets [p0, p1) of ONE workload: x as in
// kq_tas_find_replacement (include/kq_tas.h). A podset with is_replacement == 0 gets no result (status KQ_TAS_SKIPPED, findPSA :612).
void find_workload_replacement(Snapshot& s, const kq_tas_requests* rq, const kq_tas_replacement* x, int p0, int p1, std::vector<PodSetResult>* out);
// TASFlavorSnapshot.Fits :433 for one TopologyDomainRequests{leaf, SinglePodRequests, Count}; req in the dense convention of
// kq_tas_fits (0 = absent, KQ_TAS_REQ_ZERO = present with quantity zero).
bool fits_domain(const Snapshot& s, int leaf, int32_t count, const int64_t* req);
// updateTASUsage :267: tas_usage[leaf] +/-= req * count, pods +/-= count
void usage_apply(Snapshot& s, int leaf, int32_t count, const int64_t* req, bool add);
// copies tas_usage [n_leaves][n_resources] out; returns the number of values written
size_t snapshot_export_usage(const Snapshot& s, int64_t* out);

}  // namespace tas

//go:build kq_hip

// flatten_tas.go — FlattenTAS: cache.Snapshot + the flattened snapshot / heads -> kqengine.TASCycle (include/kq_cycle_tas.h), the input of
// Engine.RunCycleTAS. The unexported half (tree, leaf capacities) comes from (*TASFlavorSnapshot).ExportFlat, which lives in
// pkg/cache/scheduler (shim/go/schdcache/flatten_tas.go). Field by field as INTEGRATION.md "Topology-Aware Scheduling inside the cycle".
package kqengine

import (
	"context"
	"slices"

	corev1 "k8s.io/api/core/v1"

	kueue "sigs.k8s.io/kueue/apis/kueue/v1beta2"

	qcache "sigs.k8s.io/kueue/pkg/cache/queue"
	schdcache "sigs.k8s.io/kueue/pkg/cache/scheduler"
	"sigs.k8s.io/kueue/pkg/features"
	utiltas "sigs.k8s.io/kueue/pkg/util/tas"
	"sigs.k8s.io/kueue/pkg/workload"
)

// FlattenTAS returns nil when the snapshot holds no TAS flavor (the cycle then goes through RunCycle).
func FlattenTAS(ctx context.Context, snap *schdcache.Snapshot, fs *FlatSnapshot, ix *Index, heads []*qcache.Head, fh *FlatHeads) *TASCycle {
	maskRows := map[string]int32{} // leaf_mask rows by content
	// every TAS flavor of the snapshot, name order; one TASFlavorSnapshot per flavor is shared by the ClusterQueues (snapshot.go:260)
	flavors := map[string]*schdcache.TASFlavorSnapshot{}
	for _, cq := range snap.ClusterQueues() {
		for name, tfs := range cq.TASFlavors {
			flavors[string(name)] = tfs
		}
	}
	if len(flavors) == 0 {
		return nil
	}
	names := sortedKeys(flavors)
	resNames := make([]corev1.ResourceName, len(fs.ResourceNames))
	for i, n := range fs.ResourceNames {
		resNames[i] = corev1.ResourceName(n)
	}
	nR := len(resNames)
	tc := &TASCycle{NoRecompute: !features.Enabled(features.TASRecomputeAssignmentWithinSchedulingCycle),
		NoFailFast: !features.Enabled(features.TASFailedNodeReplacementFailFast)}
	flats := make([]*schdcache.FlatTAS, len(names))
	tasIdx := map[string]int32{}
	for i, n := range names {
		f := flavors[n].ExportFlat(resNames)
		flats[i] = f
		tasIdx[n] = int32(i)
		tc.TASFlavor = append(tc.TASFlavor, ix.Flavor[n])
		tc.Topos = append(tc.Topos, FlatTopology{NLevels: int32(len(f.LevelKeys)), NResources: int32(nR), PodsResource: fs.PodsResource,
			ProfileMixed: features.Enabled(features.TASProfileMixed), BalancedPlacement: features.Enabled(features.TASBalancedPlacement),
			AffinityPreferred: features.Enabled(features.TASRespectNodeAffinityPreferred), LevelOff: f.LevelOff, Parent: f.Parent,
			FreeCapacity: f.FreeCapacity, TASUsage: slices.Clone(f.TASUsage), LeafValues: f.LeafValues})
	}
	// isTASOnly (clusterqueue.go:746)
	tc.CQTASOnly = make([]uint8, fs.NCQ)
	for name, cq := range snap.ClusterQueues() {
		if cq.IsTASOnly() {
			tc.CQTASOnly[ix.CQ[string(name)]] = 1
		}
	}
	// admitted rows: workload.TASUsage() as (TAS flavor, leaf, count, SinglePodRequests); the rows' usage leaves the exported tasUsage so that
	// the engine adds it back row by row (and can take a row out again in SimulateWorkloadRemoval)
	tc.AdmOff = make([]int32, fs.NAdm+1)
	rowInfo := admittedInfos(snap, fs) // workload.Info of every admitted row, in row order (flatten.go keeps AdmKeys)
	for row, wi := range rowInfo {
		for flv, usage := range wi.TASUsage() {
			ti, ok := tasIdx[string(flv)]
			if !ok {
				continue
			}
			for _, tr := range usage {
				leaf := flats[ti].LeafOfID[utiltas.DomainID(tr.Values)]
				tc.AdmTAS = append(tc.AdmTAS, ti)
				tc.AdmLeaf = append(tc.AdmLeaf, leaf)
				tc.AdmCount = append(tc.AdmCount, tr.Count)
				for r, name := range resNames {
					q := tr.SinglePodRequests.Get(name)
					tc.AdmReq = append(tc.AdmReq, q)
					tc.Topos[ti].TASUsage[int(leaf)*nR+r] -= q * int64(tr.Count)
				}
			}
		}
		tc.AdmOff[row+1] = int32(len(tc.AdmTAS))
	}
	// podsets of the heads: the topology request resolved against EVERY TAS flavor (the flavor a podset lands on is only known on the device)
	nps := int(fh.PsOff[fh.N])
	nt := len(names)
	tc.PsFlags, tc.PsKind = make([]uint8, nps), make([]uint8, nps)
	tc.PsLevel, tc.PsSliceLevel = make([]int32, nps*nt), make([]int32, nps*nt)
	tc.PsSliceSize, tc.PsGroup = make([]int32, nps), make([]int32, nps)
	tc.PsReq = make([]int64, nps*nR)
	p := 0
	for _, hd := range heads {
		groups := map[string]int32{}
		for i := range hd.Info.TotalRequests {
			psr := &hd.Info.TotalRequests[i]
			ps := &hd.Info.Obj.Spec.PodSets[i]
			if workload.IsExplicitlyRequestingTAS(*ps) {
				tc.PsFlags[p] |= 1 // KQ_PS_TAS_EXPLICIT
			}
			tr := ps.TopologyRequest
			kind, levelKey, sliceKey, sliceSize := uint8(2), "", "", int32(1) // implied request: unconstrained (tas_flavorassigner.go:92-114)
			if tr != nil {
				switch {
				case tr.Required != nil:
					kind, levelKey = 0, *tr.Required
				case tr.Preferred != nil:
					kind, levelKey = 1, *tr.Preferred
				}
				if tr.PodSetSliceRequiredTopology != nil {
					sliceKey = *tr.PodSetSliceRequiredTopology
				}
				if tr.PodSetSliceSize != nil {
					sliceSize = *tr.PodSetSliceSize
				}
				if tr.PodSetGroupName != nil {
					g, ok := groups[*tr.PodSetGroupName]
					if !ok {
						g = int32(len(groups)) + 1
						groups[*tr.PodSetGroupName] = g
					}
					tc.PsGroup[p] = g
				}
			}
			tc.PsKind[p] = kind
			tc.PsSliceSize[p] = sliceSize
			for t := range nt {
				// levelKeyWithImpliedFallback / sliceLevelKeyWithDefault (tas_flavor_snapshot.go:1197-1238): no key -> the lowest level
				keys := flats[t].LevelKeys
				lv, sl := int32(len(keys)-1), int32(len(keys)-1)
				if levelKey != "" {
					lv = int32(slices.Index(keys, levelKey)) // -1: the flavor does not know the level (the engine reports KQ_TAS_BAD_LEVEL)
				}
				if sliceKey != "" {
					sl = int32(slices.Index(keys, sliceKey))
				}
				tc.PsLevel[p*nt+t], tc.PsSliceLevel[p*nt+t] = lv, sl
			}
			// node feasibility (kq_cycle_tas.ps_mask / leaf_mask): the snapshot's own FindFeasibleNodes per TAS flavor; equal masks share a row
			for t := range nt {
				mask, err := flavors[names[t]].FeasibleLeafMask(ctx, ps, flats[t])
				if err != nil {
					tc.Unsupported = true // an unparsable selector / affinity: the reference fails the placement with its message (:958, :968) — that cycle stays in Go
					continue
				}
				if mask == nil {
					continue
				}
				if tc.PsMask == nil {
					tc.PsMask = make([]int32, len(tc.PsKind)*nt)
					for i := range tc.PsMask {
						tc.PsMask[i] = -1
					}
					for _, f := range flats {
						tc.MaskStride = max(tc.MaskStride, int32(len(f.LeafValues)))
					}
				}
				row := make([]uint8, tc.MaskStride)
				copy(row, mask)
				key := string(row)
				id, ok := maskRows[key]
				if !ok {
					id = int32(len(maskRows))
					maskRows[key] = id
					tc.LeafMask = append(tc.LeafMask, row...)
				}
				tc.PsMask[p*nt+t] = id
			}
			// SinglePodRequests = resources.NewRequestsFromPodSpec (tas_flavorassigner.go:116)
			single := psr.SinglePodRequests()
			for r, name := range resNames {
				tc.PsReq[p*nR+r] = single.Get(name)
			}
			p++
		}
	}
	flattenSecondPass(tc, fs, ix, fh, heads, flats, tasIdx)
	return tc
}

// flattenSecondPass fills TASCycle.PsAdmFlavor / PsEx* for the heads that hold an admission (workload.NeedsSecondPass workload.go:974):
// PodSetResources.Flavors is what Assign keeps (flavorassigner.go:768-774), Status.Admission's TopologyAssignment what
// WorkloadsTopologyRequests (:50) and findReplacementAssignment (tas_flavor_snapshot.go:686) read. Workloads owned by a single pod are the
// caller's to keep off the engine while SkipReassignmentForPodOwnedWorkloads is on (tas_flavor_snapshot.go:615).
func flattenSecondPass(tc *TASCycle, fs *FlatSnapshot, ix *Index, fh *FlatHeads, heads []*qcache.Head, flats []*schdcache.FlatTAS, tasIdx map[string]int32) {
	any := false
	for _, hd := range heads {
		if workload.HasQuotaReservation(hd.Info.Obj) && hd.Info.Obj.Status.Admission != nil {
			any = true
		}
	}
	if !any {
		return
	}
	nR := int(fs.NResource)
	nps := int(fh.PsOff[fh.N])
	tc.PsAdmFlavor = make([]int32, nps*nR)
	for i := range tc.PsAdmFlavor {
		tc.PsAdmFlavor[i] = -1
	}
	tc.PsExOff = make([]int32, 1, nps+1)
	p := 0
	for _, hd := range heads {
		wl := hd.Info.Obj
		second := workload.HasQuotaReservation(wl) && wl.Status.Admission != nil
		for i := range hd.Info.TotalRequests {
			if second {
				psr := &hd.Info.TotalRequests[i]
				ti := int32(-1)
				for res, flv := range psr.Flavors {
					tc.PsAdmFlavor[p*nR+int(ix.Resource[string(res)])] = ix.Flavor[string(flv)]
					if t, ok := tasIdx[string(flv)]; ok {
						ti = t
					}
				}
				if psa := findPSA(wl, psr.Name); psa != nil && psa.TopologyAssignment != nil && ti >= 0 {
					for _, d := range utiltas.InternalFrom(psa.TopologyAssignment).Domains {
						leaf, ok := flats[ti].LeafOfID[utiltas.DomainID(d.Values)]
						if !ok {
							leaf = -1 // IsTopologyAssignmentStale :818
						}
						f := uint8(0)
						if node := d.Values[len(d.Values)-1]; workload.HasUnhealthyNode(wl, node) {
							f |= ExUnhealthy
							if wl.Status.UnhealthyNodes[0].Name == node {
								f |= ExFirst
							}
						}
						tc.PsExLeaf, tc.PsExCount, tc.PsExFlags = append(tc.PsExLeaf, leaf), append(tc.PsExCount, d.Count), append(tc.PsExFlags, f)
					}
				}
			}
			tc.PsExOff = append(tc.PsExOff, int32(len(tc.PsExLeaf)))
			p++
		}
	}
}

func findPSA(wl *kueue.Workload, name kueue.PodSetReference) *kueue.PodSetAssignment {
	for i := range wl.Status.Admission.PodSetAssignments {
		if wl.Status.Admission.PodSetAssignments[i].Name == name {
			return &wl.Status.Admission.PodSetAssignments[i]
		}
	}
	return nil
}

// admittedInfos lists the workload.Info of the admitted rows in FlatSnapshot row order (ClusterQueue name order, then the order Flatten
// emitted cq.Workloads in: AdmKeys holds the keys).
func admittedInfos(snap *schdcache.Snapshot, fs *FlatSnapshot) []*workload.Info {
	out := make([]*workload.Info, 0, fs.NAdm)
	byKey := map[string]*workload.Info{}
	for _, cq := range snap.ClusterQueues() {
		for key, wi := range cq.Workloads {
			byKey[string(key)] = wi
		}
	}
	for _, k := range fs.AdmKeys {
		out = append(out, byKey[k])
	}
	return out
}

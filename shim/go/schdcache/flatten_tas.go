//go:build kq_hip

// flatten_tas.go — DROPS INTO pkg/cache/scheduler (package scheduler): the TASFlavorSnapshot's tree and leaf capacities are unexported
// (tas_flavor_snapshot.go:60-160, tas_topology_tree.go:28-109), so the export that the engine's flatten needs lives next to them. It
// returns plain slices; shim/go/flatten_tas.go (package kqengine) turns them into a kqengine.TASCycle — pkg/cache/scheduler must not import
// the binding (kqengine imports this package for Snapshot).
//
// NOT COMPILED IN THIS REPOSITORY (no Go toolchain in the build image); field and method names are those of the reference at
// /root/reference (kueue main, TAS snapshot with the shared topologyTree).
package scheduler

import (
	"context"
	"slices"

	corev1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/labels"
	"k8s.io/component-helpers/scheduling/corev1/nodeaffinity"

	kueue "sigs.k8s.io/kueue/apis/kueue/v1beta2"
	"sigs.k8s.io/kueue/pkg/resources"
	utiltas "sigs.k8s.io/kueue/pkg/util/tas"
)

// FlatTAS is one TASFlavorSnapshot in the layout of include/kq_tas.h kq_tas_topology: the domains of every level numbered in the
// lexicographic order of their levelValues (compareDomainLevelValues, tas_flavor_snapshot.go:1727 — every "levelValues ascending"
// tie-break of the placement becomes an integer compare and the children of a domain are a contiguous id range).
type FlatTAS struct {
	LevelKeys []string
	LevelOff  []int32 // [levels+1] first domain of every level in the concatenated numbering
	Parent    []int32 // [domains] index WITHIN the level above, -1 at level 0
	// leaves in canonical order: [leaf][resource] over `resourceNames`
	FreeCapacity, TASUsage []int64
	LeafValues             [][]string // levelValues of every leaf (to turn (leaf, count) back into a TopologyAssignment)
	LeafOfID               map[utiltas.TopologyDomainID]int32
	IsLowestLevelNode      bool
}

// ExportFlat flattens the snapshot for the resources `resourceNames` (the snapshot's resource dictionary, name-sorted; "pods" included when
// the ClusterQueues cover it). tasUsage is exported AS IS: the caller subtracts the usage of the admitted rows it lists itself
// (kq_cycle_tas.adm_*), so that SimulateWorkloadRemoval takes exactly what adding the row put there.
func (s *TASFlavorSnapshot) ExportFlat(resourceNames []corev1.ResourceName) *FlatTAS {
	t := s.topologyTree
	nl := len(t.levelKeys)
	out := &FlatTAS{LevelKeys: slices.Clone(t.levelKeys), LevelOff: make([]int32, nl+1), LeafOfID: map[utiltas.TopologyDomainID]int32{},
		IsLowestLevelNode: t.isLowestLevelNode}
	// canonical numbering level by level: sort by levelValues
	index := make([]map[*domain]int32, nl) // domain -> index within its level
	total := 0
	for l := range nl {
		doms := make([]*domain, 0, len(t.domainsPerLevel[l]))
		for _, d := range t.domainsPerLevel[l] {
			doms = append(doms, d)
		}
		slices.SortFunc(doms, func(a, b *domain) int { return slices.Compare(a.levelValues, b.levelValues) })
		index[l] = make(map[*domain]int32, len(doms))
		out.LevelOff[l] = int32(total)
		for i, d := range doms {
			index[l][d] = int32(i)
			if l == 0 {
				out.Parent = append(out.Parent, -1)
			} else {
				out.Parent = append(out.Parent, index[l-1][d.parent])
			}
		}
		total += len(doms)
		if l == nl-1 {
			nr := len(resourceNames)
			out.FreeCapacity = make([]int64, len(doms)*nr)
			out.TASUsage = make([]int64, len(doms)*nr)
			out.LeafValues = make([][]string, len(doms))
			for i, d := range doms {
				leaf := t.leaves[d.id]
				lc := s.leafCapacityOf(leaf)
				out.LeafValues[i] = d.levelValues
				out.LeafOfID[d.id] = int32(i)
				for r, name := range resourceNames {
					out.FreeCapacity[i*nr+r] = requestOf(lc.freeCapacity, name)
					out.TASUsage[i*nr+r] = requestOf(lc.tasUsage, name)
				}
			}
		}
	}
	out.LevelOff[nl] = int32(total)
	return out
}

func requestOf(r resources.Requests, name corev1.ResourceName) int64 {
	return r.Get(name)
}

// TASFlavorNames = the ClusterQueue's TAS flavors in the order the flavor assigner walks them (slices.Sorted, clusterqueue_snapshot.go:220).
func (c *ClusterQueueSnapshot) TASFlavorNames() []string {
	names := make([]string, 0, len(c.TASFlavors))
	for n := range c.TASFlavors {
		names = append(names, string(n))
	}
	slices.Sort(names)
	return names
}

// FeasibleLeafMask is FindFeasibleNodes for one podset on this flavor as a 0/1 per leaf of `flat` (kq_cycle_tas.leaf_mask rows): the
// requirements are built exactly as findTopologyAssignment builds them (tas_flavor_snapshot.go:955-985 — tolerations = the podset's + the
// flavor's, PodSpec.NodeSelector only when the lowest level is the node, the required node affinity) and handed to the snapshot's own
// getMatchingLeaves (:1845), so a custom simulator and the TASCacheNodeMatchResults cache keep working. nil = every leaf is feasible (no
// row needed). A selector / affinity that does not parse is the placement's failure message in the reference (:958, :968): the caller
// keeps such a head out of the engine's batch (err != nil).
func (s *TASFlavorSnapshot) FeasibleLeafMask(ctx context.Context, ps *kueue.PodSet, flat *FlatTAS) ([]uint8, error) {
	info := ps.Template.Spec
	req := &topologyAssignmentPodRequirements{}
	req.podRequirements.Tolerations = append(slices.Clone(info.Tolerations), s.tolerations...)
	if s.isLowestLevelNode {
		sel, err := labels.ValidatedSelectorFromSet(info.NodeSelector)
		if err != nil {
			return nil, err
		}
		req.podRequirements.Selector = sel
	} else {
		req.podRequirements.Selector = labels.Everything()
	}
	if a := info.Affinity; a != nil && a.NodeAffinity != nil && a.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution != nil {
		as, err := nodeaffinity.NewNodeSelector(a.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution)
		if err != nil {
			return nil, err
		}
		req.podRequirements.AffinitySelector = as
	}
	matched, _, err := s.getMatchingLeaves(ctx, req)
	if err != nil {
		return nil, err
	}
	if len(matched) == len(flat.LeafValues) {
		return nil, nil
	}
	mask := make([]uint8, len(flat.LeafValues))
	for _, m := range matched {
		if i, ok := flat.LeafOfID[m.GetID()]; ok {
			mask[i] = 1
		}
	}
	return mask, nil
}

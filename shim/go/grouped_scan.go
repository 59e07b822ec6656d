//go:build kq_hip

package kqengine

// grouped_scan.go — GroupedScanRisk, twin of kueue_amd/tas_cycle.py grouped_scan_risk.
//
// assignFlavors scans flavors once per PodSetGroupName group over the SUM of the members' requests and hands every member the group's flavors
// (flavorassigner.go:782-860, resolvePodSetFlavors :917-945); the engine's flavor scan runs per podset. The two provably agree for a head
// unless one of its groups of two or more podsets (a) has a member that requests no resource the ClusterQueue covers (it takes the group's
// Status and TAS flavors), (b) belongs to a head that carries a LastAssignment bookmark (the group resumes from its FIRST member's, :1092),
// (c) has members with different eligibility masks (checkFlavorForPodSets walks the whole group, :1234), or (d) requests a resource whose
// resource group lists more than one flavor (the sum may not fit where the parts do). On the repository's random TAS cycles 1.6 % of the
// heads with such a group differ in their decision and every one of them is listed here (tests/test_oracle_grouped_flavors.py); a scheduler
// that must match the reference exactly keeps the cycles of the listed heads on the reference's path (DESIGN §7 "grouped flavor assignment").
func GroupedScanRisk(s *FlatSnapshot, h *FlatHeads, psGroup []int32) []int {
	var out []int
	nps := int(h.PsOff[h.N])
	nw := 0
	if nps > 0 {
		nw = len(h.PsFlavorOK) / nps
	}
	for i := 0; i < int(h.N); i++ {
		cq := h.CQ[i]
		rgOf := map[int32]int32{}
		for rg := s.CQRgOff[cq]; rg < s.CQRgOff[cq+1]; rg++ {
			for k := s.RgResOff[rg]; k < s.RgResOff[rg+1]; k++ {
				if _, ok := rgOf[s.RgRes[k]]; !ok {
					rgOf[s.RgRes[k]] = rg
				}
			}
		}
		_, podsCovered := rgOf[s.PodsResource]
		podsCovered = podsCovered && s.PodsResource >= 0
		members := map[int32][]int32{}
		for p := h.PsOff[i]; p < h.PsOff[i+1]; p++ {
			if psGroup[p] >= 0 {
				members[psGroup[p]] = append(members[psGroup[p]], p)
			}
		}
		risky := false
		for _, ms := range members {
			if len(ms) < 2 {
				continue
			}
			if h.Flags[i]&headLast != 0 {
				risky = true
			}
			for _, p := range ms {
				for w := 0; w < nw; w++ {
					if h.PsFlavorOK[int(p)*nw+w] != h.PsFlavorOK[int(ms[0])*nw+w] {
						risky = true
					}
				}
				covered := podsCovered
				for k := h.PsReqOff[p]; k < h.PsReqOff[p+1]; k++ {
					if rg, ok := rgOf[h.ReqRes[k]]; ok {
						covered = true
						if s.RgFlavorOff[rg+1]-s.RgFlavorOff[rg] > 1 {
							risky = true
						}
					}
				}
				if podsCovered {
					if rg := rgOf[s.PodsResource]; s.RgFlavorOff[rg+1]-s.RgFlavorOff[rg] > 1 {
						risky = true
					}
				}
				if !covered {
					risky = true
				}
			}
		}
		if risky {
			out = append(out, i)
		}
	}
	return out
}

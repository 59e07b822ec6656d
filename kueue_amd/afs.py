"""Host side of AdmissionFairSharing ordering: afs.CalculateUsage (pkg/util/admissionfairsharing/admission_fair_sharing.go:86-102).
The engine's pending side compares LocalQueues by this number (queueOrderingFunc, pkg/cache/queue/cluster_queue.go:880-904); the number
itself is the host's to compute from its usage ledger (consumed resources with decay, pending penalties, weights)."""
import math
from typing import Dict, Optional


def calculate_usage(consumed: Dict[str, float], penalty: Optional[Dict[str, float]] = None, lq_weight: float = 1.0,
                    res_weights: Optional[Dict[str, float]] = None) -> float:
    """consumed / penalty: resource name -> quantity in the resource's standard unit (cores, bytes, counts: Quantity.AsApproximateFloat64)."""
    allr = dict(consumed)
    for k, v in (penalty or {}).items():
        allr[k] = allr.get(k, 0.0) + v
    usage = 0.0
    for name in sorted(allr):
        w = (res_weights or {}).get(name, 1.0)
        usage += w * allr[name]
    if lq_weight <= 0:
        return math.inf
    return usage / lq_weight

"""ctypes view of include/kq_engine.h (the C-ABI drop-in boundary).

Only plumbing lives here: struct layouts, numpy <-> pointer helpers and the loader of the HIP
engine library. There is NO CPU fallback: if ``libkq_engine.so`` is missing or no HIP device is
usable, the product path raises (see ``load_engine``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ENGINE_LIB = os.environ.get("KQ_ENGINE_LIB", os.path.join(HERE, "libkq_engine.so"))   # (KQ_ENGINE_LIB: an A/B or timing build of the same sources)

KQ_ABI_VERSION = 3
KQ_UNLIMITED = (1 << 63) - 1
KQ_NIL_LIMIT = -1

KQ_QF_QUOTA = 0x1
KQ_QF_SUBTREE = 0x2

# gates
KQ_GATE_FLAVOR_FUNGIBILITY = 1 << 0
KQ_GATE_PRESERVE_SCAN_PROGRESS = 1 << 1
KQ_GATE_PARTIAL_ADMISSION = 1 << 2
KQ_GATE_PRIORITY_SORTING_IN_COHORT = 1 << 3
KQ_GATE_FS_PREEMPT_WITHIN_NOMINAL = 1 << 4
KQ_GATE_FS_PRIORITIZE_NON_BORROWING = 1 << 5
KQ_GATE_RECOMPUTE_ON_OVERLAP = 1 << 6
KQ_GATE_PRIORITIZE_PREEMPTORS = 1 << 7
KQ_GATE_QUOTA_CHECK_STRATEGY = 1 << 8
KQ_GATE_SCHEDULING_EQUIVALENCE_HASHING = 1 << 9
KQ_GATE_ELASTIC_JOBS = 1 << 10
KQ_GATES_DEFAULT = (
    KQ_GATE_FLAVOR_FUNGIBILITY
    | KQ_GATE_PRESERVE_SCAN_PROGRESS
    | KQ_GATE_PARTIAL_ADMISSION
    | KQ_GATE_PRIORITY_SORTING_IN_COHORT
    | KQ_GATE_FS_PREEMPT_WITHIN_NOMINAL
    | KQ_GATE_FS_PRIORITIZE_NON_BORROWING
    | KQ_GATE_RECOMPUTE_ON_OVERLAP
    | KQ_GATE_QUOTA_CHECK_STRATEGY
    | KQ_GATE_SCHEDULING_EQUIVALENCE_HASHING
    | KQ_GATE_ELASTIC_JOBS
)
GATE_BY_NAME = {
    "SchedulingEquivalenceHashing": KQ_GATE_SCHEDULING_EQUIVALENCE_HASHING,
    "FlavorFungibility": KQ_GATE_FLAVOR_FUNGIBILITY,
    "FlavorFungibilityPreserveScanProgress": KQ_GATE_PRESERVE_SCAN_PROGRESS,
    "PartialAdmission": KQ_GATE_PARTIAL_ADMISSION,
    "PrioritySortingWithinCohort": KQ_GATE_PRIORITY_SORTING_IN_COHORT,
    "FairSharingPreemptWithinNominal": KQ_GATE_FS_PREEMPT_WITHIN_NOMINAL,
    "FairSharingPrioritizeNonBorrowing": KQ_GATE_FS_PRIORITIZE_NON_BORROWING,
    "RecomputeAssignmentUponPreemptionTargetsOverlap": KQ_GATE_RECOMPUTE_ON_OVERLAP,
    "PrioritizePreemptorWorkloads": KQ_GATE_PRIORITIZE_PREEMPTORS,
    "QuotaCheckStrategy": KQ_GATE_QUOTA_CHECK_STRATEGY,
    "ElasticJobsViaWorkloadSlices": KQ_GATE_ELASTIC_JOBS,
}

# modes / status / actions
NoFit, Preempt, DeferredFit, Fit = 0, 1, 2, 3
MODE_NAMES = {0: "NoFit", 1: "Preempt", 2: "DeferredFit", 3: "Fit"}
ST_NOT_NOMINATED, ST_NOMINATED, ST_SKIPPED, ST_ASSUMED = 0, 1, 2, 5
ST_EVICTED = 4   # kq_cycle_run_tas: handleFailedTASReplacement
RQ_GENERIC, RQ_FAILED_AFTER_NOMINATION, RQ_PENDING_PREEMPTION, RQ_NOFIT, RQ_PREEMPTION_NO_CANDIDATES = 0, 1, 4, 7, 8
ACT_NONE, ACT_ADMIT, ACT_PREEMPT = 0, 1, 2
ACT_EVICT = 3
SKIP_NONE, SKIP_OVERLAP, SKIP_NO_LONGER_FITS = 0, 1, 2
REASONS = {
    0: "InClusterQueue",
    1: "InCohortReclamation",
    2: "InCohortFairSharing",
    3: "InCohortReclaimWhileBorrowing",
    4: "ReplacedWorkloadSlice",   # not a preemption reason: the old slice the head replaces (kq_engine.h KQ_REASON_REPLACED_SLICE)
}
REASON_BY_NAME = {v: k for k, v in REASONS.items()}

HEAD_HAS_QUOTA_RESERVATION = 0x1
HEAD_IS_PREEMPTOR = 0x2
HEAD_HAS_UNHEALTHY_NODES = 0x8
HEAD_UNHEALTHY_ASSIGNMENT = 0x10
HEAD_HAS_LAST_ASSIGNMENT = 0x4
ADM_EVICTED = 0x1

KQ_OK, KQ_EINVAL, KQ_ENOMEM, KQ_EDEVICE, KQ_EUNSUPPORTED, KQ_ECAPACITY, KQ_ENODEVICE = 0, -1, -2, -3, -4, -5, -6
KQ_ERRORS = {0: "OK", -1: "EINVAL", -2: "ENOMEM", -3: "EDEVICE", -4: "EUNSUPPORTED", -5: "ECAPACITY", -6: "ENODEVICE"}

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)


class kq_config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("device", C.c_int32),
        ("gates", C.c_uint32),
        ("fair_sharing", C.c_int32),
        ("n_fs_strategies", C.c_int32),
        ("fs_strategies", C.c_int32 * 2),
        ("quota_check_strategy", C.c_int32),
    ]


class kq_snapshot(C.Structure):
    _fields_ = [
        ("n_cq", C.c_int32), ("n_cohort", C.c_int32), ("n_flavor", C.c_int32), ("n_resource", C.c_int32),
        ("pods_resource", C.c_int32),
        ("resource_order", i32p),
        ("parent", i32p),
        ("child_cohort_off", i32p), ("child_cohort", i32p),
        ("child_cq_off", i32p), ("child_cq", i32p),
        ("fair_weight", f64p),
        ("nominal", i64p), ("borrow_limit", i64p), ("lend_limit", i64p), ("subtree_quota", i64p), ("usage", i64p),
        ("quota_flags", u8p),
        ("cq_rg_off", i32p),
        ("rg_flavor_off", i32p), ("rg_flavor", i32p),
        ("rg_res_off", i32p), ("rg_res", i32p),
        ("cq_policy", u32p),
        ("cq_borrow_prio_threshold", i32p),
        ("cq_generation", i64p),
        ("n_adm", C.c_int32),
        ("cq_adm_off", i32p),
        ("adm_priority", i64p), ("adm_queue_ts", i64p), ("adm_reserve_ts", i64p),
        ("adm_uid_rank", u32p),
        ("adm_flags", u8p),
        ("adm_use_off", i32p), ("adm_use_fr", i32p), ("adm_use_qty", i64p),
    ]


class kq_heads(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
        ("cycle", C.c_int64),
        ("cq", i32p), ("priority", i64p), ("queue_ts", i64p), ("flags", u32p),
        ("ps_off", i32p),
        ("ps_count", i32p), ("ps_min_count", i32p),
        ("ps_req_off", i32p), ("req_res", i32p), ("req_qty", i64p),
        ("ps_flavor_ok", u64p),
        ("ps_last_tried", i32p),
        ("last_generation", i64p), ("last_cycle", i64p), ("last_hash", u64p), ("hash", u64p),
        ("slice_row", i32p), ("ps_slice_count", i32p), ("req_slice_flavor", i32p), ("req_slice_qty", i64p),
        ("ps_slice_pods_flavor", i32p), ("ps_slice_pods_qty", i64p),
        ("ps_group", i32p),
    ]


class kq_pending(C.Structure):
    _fields_ = [("w", kq_heads), ("uid_rank", u32p), ("n_lq", C.c_int32), ("lq", i32p), ("requeue_at", i64p), ("same_generation", u8p)]


class kq_afs_ledger(C.Structure):
    _fields_ = [("n_lq", C.c_int32), ("n_res", C.c_int32), ("lq_weight", f64p), ("res_weight", f64p),
                ("consumed_lo", u64p), ("consumed_hi", i64p), ("consumed_f64", f64p),
                ("penalty_lo", u64p), ("penalty_hi", i64p), ("penalty_present", u8p),
                ("wl_penalty_lo", u64p), ("wl_penalty_hi", i64p), ("wl_penalty_mask", u64p)]


REQUEUE_NONE, REQUEUE_BLOCKED = -(1 << 63), (1 << 63) - 1
WL_ACTIVE, WL_INFLIGHT, WL_INADMISSIBLE, WL_GONE = 0, 1, 2, 3
PATCH_USAGE, PATCH_ADMITTED = 1, 2


class kq_row_patch(C.Structure):
    _fields_ = [("n_remove", C.c_int32), ("remove_rows", i32p), ("n_add", C.c_int32), ("add_cq", i32p), ("add_priority", i64p), ("add_queue_ts", i64p),
                ("add_reserve_ts", i64p), ("add_uid_rank", u32p), ("add_flags", u8p), ("add_use_off", i32p), ("add_use_fr", i32p), ("add_use_qty", i64p),
                ("n_evict", C.c_int32), ("evict_rows", i32p), ("flags", C.c_uint32)]


ROWS_FOLD_USAGE = 1


class kq_decisions(C.Structure):
    _fields_ = [
        ("status", u8p), ("action", u8p), ("nominated_mode", u8p), ("mode", u8p),
        ("requeue_reason", u8p), ("skip", u8p),
        ("borrowing", i32p), ("order", i32p),
        ("flavor", i32p), ("res_mode", u8p), ("tried_idx", i32p),
        ("ps_count", i32p),
        ("tgt_off", i32p), ("tgt_cap", C.c_int32), ("tgt_adm", i32p), ("tgt_reason", u8p),
        ("rsn_cap", C.c_int32), ("rsn_off", i32p), ("rsn_code", u8p), ("rsn_podset", u8p),
        ("rsn_flavor", C.POINTER(C.c_int16)), ("rsn_resource", C.POINTER(C.c_int16)),
        ("rsn_a", i64p), ("rsn_b", i64p), ("rsn_c", i64p),
    ]


_NP2C = {
    np.dtype(np.int32): C.c_int32,
    np.dtype(np.int16): C.c_int16,
    np.dtype(np.int64): C.c_int64,
    np.dtype(np.uint8): C.c_uint8,
    np.dtype(np.uint32): C.c_uint32,
    np.dtype(np.uint64): C.c_uint64,
    np.dtype(np.float64): C.c_double,
}


def ptr(a: np.ndarray):
    """Pointer to a C-contiguous numpy array (the array must outlive the call)."""
    assert a.flags["C_CONTIGUOUS"], "array must be contiguous"
    return a.ctypes.data_as(C.POINTER(_NP2C[a.dtype]))


def fill_struct(struct, arrays: dict, scalars: dict):
    """Populate a ctypes struct from numpy arrays / python scalars by field name."""
    for name, ctype in struct._fields_:
        if name in arrays:
            a = arrays[name]
            if a.size == 0:
                # keep a valid (non-NULL) pointer for empty arrays
                a = np.zeros(1, dtype=a.dtype)
                arrays[name + "__pad"] = a
            setattr(struct, name, ptr(a))
        elif name in scalars:
            setattr(struct, name, scalars[name])
    return struct


class EngineUnavailable(RuntimeError):
    pass


_engine_lib = None


def load_engine():
    """dlopen the HIP engine. Raises EngineUnavailable if the in-tree .so is missing."""
    global _engine_lib
    if _engine_lib is not None:
        return _engine_lib
    if not os.path.exists(ENGINE_LIB):
        raise EngineUnavailable(
            f"{ENGINE_LIB} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path."
        )
    lib = C.CDLL(ENGINE_LIB)
    lib.kq_engine_create.argtypes = [C.POINTER(kq_config), C.POINTER(C.c_void_p)]
    lib.kq_engine_create.restype = C.c_int
    lib.kq_engine_destroy.argtypes = [C.c_void_p]
    lib.kq_engine_destroy.restype = None
    lib.kq_snapshot_put.argtypes = [C.c_void_p, C.POINTER(kq_snapshot)]
    lib.kq_snapshot_put.restype = C.c_int
    lib.kq_cycle_run.argtypes = [C.c_void_p, C.POINTER(kq_heads), C.POINTER(kq_decisions)]
    lib.kq_cycle_run.restype = C.c_int
    lib.kq_heads_put.argtypes = [C.c_void_p, C.POINTER(kq_heads), C.c_int32]
    lib.kq_heads_put.restype = C.c_int
    lib.kq_cycle_run_resident.argtypes = [C.c_void_p, C.c_int32, C.POINTER(kq_decisions)]
    lib.kq_cycle_run_resident.restype = C.c_int
    lib.kq_nominate_run_resident.argtypes = [C.c_void_p, C.c_int32, C.POINTER(kq_decisions)]
    lib.kq_nominate_run_resident.restype = C.c_int
    lib.kq_pending_put.argtypes = [C.c_void_p, C.POINTER(kq_pending)]
    lib.kq_pending_heads.argtypes = [C.c_void_p, C.c_int64, u8p, i32p, i32p, i32p]
    lib.kq_cycle_run_pending.argtypes = [C.c_void_p, C.POINTER(kq_decisions)]
    lib.kq_pending_apply.argtypes = [C.c_void_p]
    lib.kq_pending_queue_inadmissible.argtypes = [C.c_void_p, C.c_int32, i32p]
    lib.kq_pending_add.argtypes = [C.c_void_p, C.POINTER(kq_pending), i32p]
    lib.kq_pending_add.restype = C.c_int
    lib.kq_pending_set_clock.argtypes = [C.c_void_p, C.c_int64]
    lib.kq_pending_set_clock.restype = C.c_int
    lib.kq_pending_set_requeue_at.argtypes = [C.c_void_p, C.c_int32, i32p, i64p]
    lib.kq_pending_set_requeue_at.restype = C.c_int
    lib.kq_pending_update.argtypes = [C.c_void_p, C.c_int32, i32p, C.POINTER(kq_pending), i32p]
    lib.kq_pending_update.restype = C.c_int
    lib.kq_pending_delete.argtypes = [C.c_void_p, C.c_int32, i32p]
    lib.kq_pending_delete.restype = C.c_int
    lib.kq_pending_bounds.argtypes = [C.c_void_p, i32p, i32p]
    lib.kq_pending_step.argtypes = [C.c_void_p, C.c_int64, u8p, C.c_int32, C.c_int32, C.c_int32]
    lib.kq_pending_step_wait.argtypes = [C.c_void_p, C.POINTER(kq_decisions), i32p, i32p, i32p]
    lib.kq_pending_step_reasons.argtypes = [C.c_void_p, C.c_int32]
    for f in ("kq_pending_bounds", "kq_pending_step", "kq_pending_step_wait", "kq_pending_step_reasons"):
        getattr(lib, f).restype = C.c_int
    lib.kq_pending_afs_put.argtypes = [C.c_void_p, C.POINTER(kq_afs_ledger)]
    lib.kq_pending_afs_wl_penalty.argtypes = [C.c_void_p, C.c_int32, i32p, u64p, i64p, u64p]
    lib.kq_pending_afs_sub_penalty.argtypes = [C.c_void_p, C.c_int32, i32p]
    lib.kq_pending_afs_set_consumed.argtypes = [C.c_void_p, C.c_int32, i32p, u64p, i64p, f64p, i32p]
    lib.kq_pending_afs_read.argtypes = [C.c_void_p, f64p, u64p, i64p, u8p, u64p, i64p, u8p]
    for f in ("kq_pending_afs_put", "kq_pending_afs_wl_penalty", "kq_pending_afs_sub_penalty", "kq_pending_afs_set_consumed", "kq_pending_afs_read"):
        getattr(lib, f).restype = C.c_int
    lib.kq_pending_set_lq_usage.argtypes = [C.c_void_p, C.c_int32, f64p]
    lib.kq_pending_set_lq_usage.restype = C.c_int
    lib.kq_pending_read_state.argtypes = [C.c_void_p, u8p, i32p]
    for f in ("kq_pending_put", "kq_pending_heads", "kq_cycle_run_pending", "kq_pending_apply", "kq_pending_queue_inadmissible", "kq_pending_read_state"):
        getattr(lib, f).restype = C.c_int
    lib.kq_snapshot_patch.argtypes = [C.c_void_p, C.POINTER(kq_snapshot), C.c_uint32]
    lib.kq_snapshot_patch.restype = C.c_int
    lib.kq_cycle_certificate.argtypes = [C.c_void_p, C.c_void_p, i64p, i32p]
    lib.kq_cycle_certificate.restype = C.c_int
    lib.kq_snapshot_usage_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.kq_snapshot_usage_add.restype = C.c_int
    lib.kq_last_cycle_phases.argtypes = [C.c_void_p, f64p, i64p]
    lib.kq_last_cycle_phases.restype = C.c_int
    lib.kq_last_cycle_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.kq_last_cycle_stats.restype = C.c_int
    lib.kq_cycle_commit.argtypes = [C.c_void_p, i32p]
    lib.kq_cycle_commit.restype = C.c_int
    lib.kq_cycle_release.argtypes = [C.c_void_p, C.c_int32]
    lib.kq_cycle_release.restype = C.c_int
    lib.kq_snapshot_derive.argtypes = [C.c_void_p]
    lib.kq_snapshot_derive.restype = C.c_int
    lib.kq_snapshot_read_planes.argtypes = [C.c_void_p, i64p, i64p, u8p]
    lib.kq_snapshot_read_planes.restype = C.c_int
    lib.kq_strerror.argtypes = [C.c_int]
    lib.kq_strerror.restype = C.c_char_p
    lib.kq_last_error.argtypes = [C.c_void_p]
    lib.kq_last_error.restype = C.c_char_p
    lib.kq_abi_version.argtypes = []
    lib.kq_abi_version.restype = C.c_int
    _engine_lib = lib
    return lib


# every symbol include/kq_engine.h declares (checked by tests/test_abi.py without a GPU)
ABI_SYMBOLS = [
    "kq_engine_create", "kq_engine_destroy", "kq_snapshot_put", "kq_cycle_run", "kq_last_cycle_stats",
    "kq_cycle_commit", "kq_cycle_release", "kq_snapshot_derive", "kq_snapshot_read_planes", "kq_strerror", "kq_last_error", "kq_abi_version",
    "kq_heads_put", "kq_cycle_run_resident", "kq_nominate_run_resident", "kq_last_cycle_phases",
    "kq_cycle_certificate", "kq_snapshot_usage_add", "kq_snapshot_patch", "kq_snapshot_patch_rows", "kq_cycle_shard_words", "kq_cycle_nominate_shard", "kq_cycle_process_merged",
    "kq_pending_bounds", "kq_pending_step", "kq_pending_step_wait", "kq_pending_step_reasons",
    "kq_pending_afs_put", "kq_pending_afs_wl_penalty", "kq_pending_afs_sub_penalty", "kq_pending_afs_set_consumed", "kq_pending_afs_read",
    "kq_pending_set_lq_usage", "kq_pending_add", "kq_pending_update", "kq_pending_delete", "kq_pending_set_clock", "kq_pending_set_requeue_at", "kq_pending_put", "kq_pending_heads", "kq_cycle_run_pending", "kq_pending_apply", "kq_pending_queue_inadmissible", "kq_pending_read_state",
    "kq_debug_read_usage_work", "kq_debug_force_exact_drs", "kq_debug_prof", "kq_debug_spec_stats", "kq_debug_disable_scan_search",
    "kq_debug_rows_rebuild", "kq_debug_read_rows", "kq_debug_check_guards",
]

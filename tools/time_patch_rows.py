"""kq_snapshot_patch_rows in a loop (100 rows out, the same 100 back in) — for rocprofv3 --kernel-trace --stats: what the call is made of."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kueue_amd.api import make_config
from kueue_amd.engine import Engine, row_patch_struct
from kueue_amd.population import generate
cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 3
snap = generate(cfgn).snapshot
eng = Engine(make_config()); eng.put(snap)
a, n = snap.arrays, snap.n_adm
rows = np.arange(0, n, n // 100)[:100]
cq_of = np.repeat(np.arange(snap.n_cq), np.diff(a["cq_adm_off"]))
u0, u1 = a["adm_use_off"][rows], a["adm_use_off"][rows + 1]
idx = np.concatenate([np.arange(x, y) for x, y in zip(u0, u1)]).astype(np.int64)
add = dict(cq=cq_of[rows], priority=a["adm_priority"][rows], queue_ts=a["adm_queue_ts"][rows], reserve_ts=a["adm_reserve_ts"][rows], uid_rank=a["adm_uid_rank"][rows],
           flags=a["adm_flags"][rows], use_off=np.concatenate([[0], np.cumsum(u1 - u0)]), use_fr=a["adm_use_fr"][idx], use_qty=a["adm_use_qty"][idx])
off = np.asarray(a["cq_adm_off"])
tail = np.array([off[c + 1] - 1 - k for c, k in bench._tail_slots(cq_of[rows])], np.int32)
new_index = np.zeros(n, np.int32)
cur, ms = rows, []
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    p, keep = row_patch_struct(cur, add)
    t1 = time.perf_counter()
    eng._check(eng._lib.kq_snapshot_patch_rows(eng._h, C.byref(p), new_index.ctypes.data_as(C.POINTER(C.c_int32))))
    ms.append((time.perf_counter() - t1) * 1e3)
    cur = tail
print("kq_snapshot_patch_rows ms: median", float(np.median(ms[1:])), "min", min(ms[1:]))

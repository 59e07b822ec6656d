import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from kueue_amd.api import make_config, Decisions
from kueue_amd.engine import Engine
from kueue_amd.population import generate, BASE_SEED
from kueue_amd.sharding import ShardedCycle
pop = generate(4, seed=BASE_SEED); snap = pop.snapshot
cfg = make_config()
a, b, a2 = Engine(cfg), Engine(cfg), Engine(cfg); a.put(snap); b.put(snap); a2.put(snap)
sc = ShardedCycle(b, None, 0, 1, device="cuda:0")
for c in range(5):
    h = pop.heads_for_cycle(c, cycle=c + 1)
    want = a.run(h, tgt_cap=4 * snap.n_adm)
    got = sc.cycle(h, tgt_cap=4 * snap.n_adm)
    w2 = a2.run(h, tgt_cap=4 * snap.n_adm); print('plain vs plain', want.equal(w2)); a2.commit()
    bad = want.equal(got)
    print("cycle", c, "bad", bad)
    for k in bad:
        x, y = want.a[k], got.a[k]
        if len(x) == len(y):
            d = np.nonzero(x != y)[0]
            print(k, len(d), d[:10], x[d[:10]], y[d[:10]])
            if k == 'tgt_adm':
                off = want.a['tgt_off']; hs = np.unique(np.searchsorted(off, d, side='right') - 1)
                print(' heads', hs[:10], 'status', want.a['status'][hs[:10]], got.a['status'][hs[:10]], 'skip', want.a['skip'][hs[:10]], 'mode', want.a['mode'][hs[:10]], got.a['mode'][hs[:10]], 'action', want.a['action'][hs[:10]])
    print("usage equal", np.array_equal(a.read_usage_work() if hasattr(a,'read_usage_work') else 0, b.read_usage_work() if hasattr(b,'read_usage_work') else 0))
    a.commit(); b.commit()
    print('resident equal', np.array_equal(a.read_usage(), b.read_usage()))

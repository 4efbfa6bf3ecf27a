import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from smqtk_indexing_amd import _lib
from oracle import cpu_ref as O
rng = np.random.default_rng(3)
n, d, nq, k = 200_000, 128, int(os.environ.get("NQ", 100)), 100
db = rng.standard_normal((n, d)).astype(np.float32)
qs = rng.standard_normal((nq, d)).astype(np.float32)
idx = _lib.DenseIndex(db)
idx.set_option("dense_int8", 0)
res = {}
for t in (2, 1, 0):
    idx.set_option("dense_tighten", t)
    dd, ii = idx.search(qs, k)
    st = idx.stats()
    res[t] = ii
    print("tighten", t, "cands/q", st["candidates"] / nq, "fallbacks", st["fallback_queries"], "mid", st["mid_tier_queries"])
ref = np.stack([O.dense_topk(db, q, k)[1] for q in qs[:8]])
for t in (2, 1, 0):
    bad = (res[t][:8] != ref).any(axis=1)
    print("tighten", t, "queries wrong among first 8:", bad.sum(), "all vs t=0 mismatch rows:", (res[t] != res[0]).any(axis=1).sum())
bad_q = np.nonzero((res[1] != res[0]).any(axis=1))[0]
for qi in bad_q[:4]:
    rd, ri = O.dense_topk(db, qs[qi], k)
    print("query", qi, "t=0 equals oracle:", np.array_equal(res[0][qi], ri), " t=1 equals oracle:", np.array_equal(res[1][qi], ri),
          "positions differing:", np.nonzero(res[1][qi] != ri)[0][:10], "missing ids:", sorted(set(ri.tolist()) - set(res[1][qi].tolist()))[:5])
    miss = sorted(set(ri.tolist()) - set(res[1][qi].tolist()))
    for m in miss[:3]:
        print("   missing row", m, "rank in oracle", int(np.nonzero(ri == m)[0][0]), "row tile", m // 32, "row in tile", m % 32)

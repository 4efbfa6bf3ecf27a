#!/usr/bin/env python3
"""Measurement helper: ItqFunctor.fit products on the device vs numpy (host) for n x 128 -> 64 bits."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smqtk_indexing_amd import _lib

n, d, bits, iters = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("BITS", 64)), 50
rng = np.random.default_rng(0)
x = rng.standard_normal((n, d)).astype(np.float32)
t0 = time.perf_counter()
fit = _lib.ItqFit(x)
fit.set_mean(fit.mean.astype(np.float32))
t1 = time.perf_counter()
cov = fit.cov()
t2 = time.perf_counter()
evals, evecs = np.linalg.eigh(cov)
pc = evecs[:, np.argsort(evals)[::-1][:bits]]
fit.project(pc)
t3 = time.perf_counter()
r, _ = np.linalg.qr(rng.standard_normal((bits, bits)))
for _ in range(iters):
    ub, _, ua = np.linalg.svd(fit.iterate(r))
    r = ua @ ub.T
t4 = time.perf_counter()
fit.close()
out = {"n": n, "d": d, "bits": bits, "iterations": iters, "device_upload_and_mean_s": t1 - t0, "device_cov_s": t2 - t1,
       "device_project_s": t3 - t2, "device_iterations_s": t4 - t3, "device_total_s": t4 - t0}
if n <= 1_000_000:
    t0 = time.perf_counter()
    m = x.mean(axis=0); xc = x - m; c = np.cov(xc.T); v = xc @ pc
    t1 = time.perf_counter()
    r2 = r.copy()
    for _ in range(5):
        b = np.where(v @ r2 >= 0, 1.0, -1.0); ub, _, ua = np.linalg.svd(b.T @ v); r2 = ua @ ub.T
    t2 = time.perf_counter()
    out.update({"host_mean_cov_project_s": t1 - t0, "host_iterations_s_extrapolated": (t2 - t1) / 5 * iters,
                "host_threads": len(os.sched_getaffinity(0))})
print(json.dumps(out))

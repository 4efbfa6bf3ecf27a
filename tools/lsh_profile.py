#!/usr/bin/env python3
"""Measurement helper: cProfile of HipLSHNearestNeighborIndex.nn_many / nn (where the host time goes)."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smqtk_indexing_amd._compat import DescriptorMemoryElement, MemoryDescriptorSet, MemoryKeyValueStore
from smqtk_indexing_amd.impls.hash_index.hip_linear import HipLinearHashIndex
from smqtk_indexing_amd.impls.lsh_functor.hip_itq import HipItqFunctor
from smqtk_indexing_amd.impls.nn_index.hip_lsh import HipLSHNearestNeighborIndex

n, d, bits, nn = int(os.environ.get("N", 1_000_000)), 128, int(os.environ.get("BITS", 64)), 100
rng = np.random.default_rng(1)
x = rng.standard_normal((n, d)).astype(np.float32)
f = HipItqFunctor(bit_length=bits)
q, _ = np.linalg.qr(rng.standard_normal((d, d)))
f.mean_vec, f.rotation = x[:10000].mean(axis=0).astype(np.float64), np.ascontiguousarray(q[:, :bits])
elems = [DescriptorMemoryElement(i).set_vector(v) for i, v in enumerate(x)]
qs = [DescriptorMemoryElement(f"q{i}").set_vector(v) for i, v in enumerate(rng.standard_normal((256, d)).astype(np.float32))]
idx = HipLSHNearestNeighborIndex(f, MemoryDescriptorSet(), MemoryKeyValueStore(), HipLinearHashIndex(),
                                 distance_method="euclidean", device_rerank=True)
idx.build_index(elems)
idx.nn_many(qs, nn)
t0 = time.perf_counter(); idx.nn_many(qs, nn); print("nn_many(256) qps", 256 / (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable(); idx.nn_many(qs, nn); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
idx.nn(qs[0], nn)
pr = cProfile.Profile(); pr.enable()
for qd in qs[:32]:
    idx.nn(qd, nn)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

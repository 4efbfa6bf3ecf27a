#!/usr/bin/env python3
"""Measurement helper: LSH pipeline queries/s (HipLSHNearestNeighborIndex) with the device
re-rank mirror vs the host re-rank path, single queries and nn_many batches."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smqtk_indexing_amd._compat import DescriptorMemoryElement, MemoryDescriptorSet, MemoryKeyValueStore
from smqtk_indexing_amd.impls.hash_index.hip_linear import HipLinearHashIndex
from smqtk_indexing_amd.impls.lsh_functor.hip_itq import HipItqFunctor
from smqtk_indexing_amd.impls.nn_index.hip_lsh import HipLSHNearestNeighborIndex

def main():
    n, d, bits, nn = int(os.environ.get("N", 200_000)), 128, int(os.environ.get("BITS", 64)), 100
    rng = np.random.default_rng(1)
    x = rng.standard_normal((n, d)).astype(np.float32)
    f = HipItqFunctor(bit_length=bits)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    f.mean_vec, f.rotation = x[:10000].mean(axis=0).astype(np.float64), np.ascontiguousarray(q[:, :bits])
    elems = [DescriptorMemoryElement(i).set_vector(v) for i, v in enumerate(x)]
    qs = [DescriptorMemoryElement(f"q{i}").set_vector(v) for i, v in enumerate(rng.standard_normal((256, d)).astype(np.float32))]
    out = {"n": n, "d": d, "bits": bits, "n_neighbors": nn}
    for name, device in (("device_rerank", True), ("host_rerank", False)):
        idx = HipLSHNearestNeighborIndex(f, MemoryDescriptorSet(), MemoryKeyValueStore(), HipLinearHashIndex(),
                                         distance_method="euclidean", device_rerank=device)
        t0 = time.perf_counter(); idx.build_index(elems); out[f"{name}_build_s"] = time.perf_counter() - t0
        idx.nn(qs[0], nn)
        t0 = time.perf_counter()
        for qd in qs[:64]:
            idx.nn(qd, nn)
        out[f"{name}_nn_qps"] = 64 / (time.perf_counter() - t0)
        idx.nn_many(qs, nn)
        t0 = time.perf_counter(); idx.nn_many(qs, nn); out[f"{name}_nn_many256_qps"] = 256 / (time.perf_counter() - t0)
    print(json.dumps(out))

if __name__ == "__main__":
    main()

"""
CPU tests of the host side: bit packing, interface template methods (ports of
the reference's interface tests), and the build / update / remove / config /
cache state machines of the plugin classes.  Nothing here launches a kernel;
searches are covered by the GPU tests.
"""
from io import BytesIO

import os

import numpy as np
import pytest

from oracle import cpu_ref as O
from smqtk_indexing_amd import HashIndex, LshFunctor, NearestNeighborsIndex, _lib
from smqtk_indexing_amd._compat import (DataMemoryElement, DescriptorMemoryElement,
                                        MemoryDescriptorSet, MemoryKeyValueStore,
                                        ReadOnlyError, from_config_dict, to_config_dict)
from smqtk_indexing_amd.impls.hash_index.hip_linear import HipLinearHashIndex
from smqtk_indexing_amd.impls.lsh_functor.hip_itq import HipItqFunctor
from smqtk_indexing_amd.impls.nn_index.hip_bruteforce import HipBruteForceNearestNeighborsIndex
from smqtk_indexing_amd.impls.nn_index.hip_lsh import HipLSHNearestNeighborIndex
from smqtk_indexing_amd.utils import bits as B
from smqtk_indexing_amd.utils.iter_validation import check_empty_iterable

NO_GPU = not _lib.usable()


# -------------------------------------------------------------------- bits
def test_bits_match_oracle():
    rng = np.random.default_rng(0)
    for b in (1, 3, 63, 64, 65, 128, 200, 256):
        v = rng.random((9, b)) > 0.5
        p = B.pack_bits_msb(v)
        np.testing.assert_array_equal(p, O.pack_bits_msb(v))
        np.testing.assert_array_equal(B.unpack_bits_msb(p, b), v)
        ints = B.packed_to_ints(p)
        assert ints == [O.bit_vector_to_int_large(r) for r in v]
        np.testing.assert_array_equal(B.ints_to_packed(ints, p.shape[1]), p)
        assert B.bit_vector_to_int_large(v[0]) == ints[0]
        np.testing.assert_array_equal(B.int_to_bit_vector_large(ints[0], b), v[0])
    with pytest.raises(ValueError):
        B.int_to_bit_vector_large(2 ** 10, 5)          # tests/utils/test_bits.py:45-54
    with pytest.raises(ValueError):
        B.ints_to_packed([2 ** 64], 1)
    assert B.int_to_bit_vector_large(0).tolist() == [False]


def test_check_empty_iterable():
    with pytest.raises(IndexError):
        check_empty_iterable([], lambda it: None, IndexError("empty"))
    assert check_empty_iterable(iter([1, 2, 3]), list, ValueError()) == [1, 2, 3]


# --------------------------------------------------- interface template tests
class _DummyNN(NearestNeighborsIndex):
    # tests/interfaces/test_nearest_neighbors_index.py:13-40
    @classmethod
    def is_usable(cls):
        return True

    def get_config(self):
        return {}

    def __init__(self):
        self.n = 0
        self.seen = None

    def count(self):
        return self.n

    def _build_index(self, d):
        self.seen = list(d)

    def _update_index(self, d):
        self.seen = list(d)

    def _remove_from_index(self, u):
        self.seen = list(u)

    def _nn(self, d, n=1):
        return (d,), (0.0,)


def test_nn_interface_templates():
    i = _DummyNN()
    for fn in (i.build_index, i.update_index, i.remove_from_index):
        with pytest.raises(ValueError, match="No DescriptorElement instances"):
            fn([])
        fn(iter([1, 2]))
        assert i.seen == [1, 2]
    q = DescriptorMemoryElement(0)
    with pytest.raises(ValueError, match="did not have a vector set"):
        i.nn(q)
    q.set_vector(np.zeros(3))
    with pytest.raises(ValueError, match="No index currently set"):
        i.nn(q)
    i.n = 5
    assert i.nn(q, 3) == ((q,), (0.0,))
    assert len(i) == 5


def test_hash_index_interface_templates():
    # tests/interfaces/test_hash_index.py:37-142
    idx = HipLinearHashIndex()
    for fn in (idx.build_index, idx.update_index, idx.remove_from_index):
        with pytest.raises(ValueError, match="No hash vectors"):
            fn([])
    with pytest.raises(ValueError, match="No index currently set"):
        idx.nn(np.zeros(8, bool))
    assert isinstance(idx, HashIndex) and isinstance(HipItqFunctor(), LshFunctor)


# -------------------------------------------------------- HipLinearHashIndex
def test_linear_build_update_remove():
    # tests/impls/hash_index/test_linear.py:47-139
    i = HipLinearHashIndex()
    i.build_index([[0, 1, 0], [1, 0, 0], [0, 1, 1], [0, 0, 1]])
    assert i.index == {1, 2, 3, 4} and i.count() == 4 and i.cache_element is None
    i.build_index([[0, 1, 0], [0, 1, 0]])
    assert i.index == {2}                                   # rebuild replaces, dedups
    i.update_index([[1, 0, 0], [0, 1, 1]])
    assert i.index == {2, 3, 4}
    i.remove_from_index([[0, 1, 0]])
    assert i.index == {3, 4}
    with pytest.raises(KeyError):
        i.remove_from_index([[0, 1, 1], [1, 1, 1]])         # 7 unknown -> untouched
    assert i.index == {3, 4}
    j = HipLinearHashIndex()
    j.update_index([[0, 1], [1, 1]])                        # update on empty == build
    assert j.index == {1, 3}


def test_linear_cache_roundtrip_and_readonly():
    # tests/impls/hash_index/test_linear.py:157-255
    cache = DataMemoryElement()
    i = HipLinearHashIndex(cache)
    assert cache.is_empty()
    i.build_index([[0, 1, 0], [1, 0, 0], [0, 1, 1], [0, 0, 1]])
    assert not cache.is_empty()
    assert HipLinearHashIndex(cache).index == {1, 2, 3, 4}
    i.update_index([[1, 1, 1]])
    assert HipLinearHashIndex(cache).index == {1, 2, 3, 4, 7}
    i.remove_from_index([[0, 0, 1]])
    assert HipLinearHashIndex(cache).index == {2, 3, 4, 7}
    # the reference's own cache format: numpy.save(tuple(ints)) (linear.py:140)
    buf = BytesIO()
    np.save(buf, (1, 2, 3, 4))
    assert HipLinearHashIndex(DataMemoryElement(buf.getvalue())).index == {1, 2, 3, 4}
    ro = DataMemoryElement(readonly=True)
    with pytest.raises(ValueError, match="read-only"):
        HipLinearHashIndex(ro).build_index([[0, 1]])
    # wide codes keep integer identity
    rng = np.random.default_rng(1)
    v = rng.random((20, 200)) > 0.5
    w = HipLinearHashIndex(DataMemoryElement())
    w.build_index(v)
    assert w.index == {O.bit_vector_to_int_large(r) for r in v}
    assert HipLinearHashIndex(w.cache_element).index == w.index


def test_linear_config_roundtrip():
    i = HipLinearHashIndex()
    c = i.get_config()
    assert c["cache_element"]["type"] is None
    assert HipLinearHashIndex.from_config(c).cache_element is None
    i2 = HipLinearHashIndex(DataMemoryElement(content_type="x"))
    c2 = i2.get_config()
    assert c2["cache_element"]["type"].endswith("DataMemoryElement")
    assert isinstance(HipLinearHashIndex.from_config(c2).cache_element, DataMemoryElement)
    assert isinstance(from_config_dict(to_config_dict(i2), [HipLinearHashIndex]), HipLinearHashIndex)


@pytest.mark.skipif(not NO_GPU, reason="checks the no-GPU failure mode")
def test_search_without_gpu_fails_loudly():
    i = HipLinearHashIndex()
    i.build_index([[0, 1], [1, 1]])
    assert not HipLinearHashIndex.is_usable()
    with pytest.raises(_lib.HipError, match="no CPU fallback"):
        i.nn([0, 0], 1)
    f = HipItqFunctor()
    f.mean_vec, f.rotation = np.zeros(2), np.eye(2)
    with pytest.raises(_lib.HipError):
        f.get_hash(np.ones(2))


# ------------------------------------------------------------- HipItqFunctor
def test_itq_functor_config_model_and_errors():
    # tests/impls/lsh_functor/test_itq.py:27-130
    f = HipItqFunctor()
    assert (f.bit_length, f.itq_iterations, f.normalize, f.random_seed) == (8, 50, None, None)
    assert not f.has_model()
    with pytest.raises(Exception, match="mean vector is none"):
        f.get_hash(np.zeros(4))
    f.mean_vec = np.zeros(4)
    with pytest.raises(Exception, match="rotation matrix is none"):
        f.get_hash(np.zeros(4))
    # every order numpy.linalg.norm accepts for a vector is accepted, like the reference (itq.py:162-164, 172-191) ...
    for ordv, code in ((1, _lib.SQ_NORM_L1), (2, _lib.SQ_NORM_L2), (0, _lib.SQ_NORM_L0), (np.inf, _lib.SQ_NORM_INF),
                       (-np.inf, _lib.SQ_NORM_NEG_INF), (3, _lib.SQ_NORM_NONE), (0.5, _lib.SQ_NORM_NONE)):
        fn = HipItqFunctor(normalize=ordv)
        assert fn._norm_ord() == code and fn._norm_on_host() == (ordv in (3, 0.5))
        v = np.array([[3., -4., 0.], [0., 0., 0.]])
        np.testing.assert_array_equal(fn._norm_vector(v), O.itq_norm_vector(v, ordv))
    # ... and what numpy rejects is rejected in the constructor
    with pytest.raises(ValueError):
        HipItqFunctor(normalize="foobar")
    with pytest.raises(ValueError):
        HipItqFunctor(normalize="fro")
    m, r = DataMemoryElement(), DataMemoryElement()
    g = HipItqFunctor(m, r, bit_length=3, itq_iterations=7, normalize=2, random_seed=4)
    g.mean_vec, g.rotation = np.arange(3.), np.eye(3)
    g.save_model()
    h = HipItqFunctor.from_config(g.get_config())
    assert h.has_model() and (h.bit_length, h.itq_iterations, h.normalize, h.random_seed) == (3, 7, 2, 4)
    np.testing.assert_array_equal(h.rotation, np.eye(3))
    with pytest.raises(RuntimeError, match="already been loaded"):
        h.fit([DescriptorMemoryElement(0).set_vector(np.zeros(3))])
    few = [DescriptorMemoryElement(i).set_vector(np.zeros(2)) for i in range(3)]
    with pytest.raises(ValueError, match="fewer features"):
        HipItqFunctor(bit_length=8).fit(few)


# ------------------------------------------- brute-force / LSH state machines
def _elems(x, base=0):
    return [DescriptorMemoryElement(base + i).set_vector(v) for i, v in enumerate(x)]


def test_bruteforce_state_machine():
    x = np.random.default_rng(0).random((10, 4))
    i = HipBruteForceNearestNeighborsIndex()
    assert i.count() == 0
    i.build_index(_elems(x))
    assert i.count() == 10 and i._matrix.dtype == np.float32
    i.update_index(_elems(x[:3] + 1, base=8))               # 8, 9 replaced; 10 added
    assert i.count() == 11
    np.testing.assert_array_equal(i._matrix[i._row_of[8]], (x[0] + 1).astype(np.float32))
    with pytest.raises(KeyError):
        i.remove_from_index([0, 99])
    assert i.count() == 11
    i.remove_from_index([0, 10])
    assert i.count() == 9 and 0 not in i._row_of
    with pytest.raises(ValueError):
        HipBruteForceNearestNeighborsIndex("hik")
    ro = HipBruteForceNearestNeighborsIndex(read_only=True)
    with pytest.raises(ReadOnlyError):
        ro.build_index(_elems(x))
    assert HipBruteForceNearestNeighborsIndex.from_config(i.get_config()).distance_method == "euclidean"


class _BitsOfSum(LshFunctor):
    """Hash = bits of int(sum(v)) (tests/impls/nn_index/test_lsh.py:28-50)."""

    def __init__(self, bits=8):
        self.bits = bits

    @classmethod
    def is_usable(cls):
        return True

    def get_config(self):
        return {"bits": self.bits}

    def get_hash(self, descriptor):
        return B.int_to_bit_vector_large(int(np.sum(descriptor)) % (2 ** self.bits), self.bits)


def test_lsh_state_machine():
    # tests/impls/nn_index/test_lsh.py:145-450 (kvstore / descriptor-set consistency)
    ds, kv, hi = MemoryDescriptorSet(), MemoryKeyValueStore(), HipLinearHashIndex()
    idx = HipLSHNearestNeighborIndex(_BitsOfSum(), ds, kv, hi, distance_method="euclidean")
    assert idx.count() == 0
    d = _elems([[0], [1], [2], [2], [5]])
    idx.build_index(d)
    assert idx.count() == 5 and len(ds) == 5
    assert kv._table == {0: {0}, 1: {1}, 2: {2, 3}, 5: {4}}
    assert hi.index == {0, 1, 2, 5}
    idx.update_index(_elems([[5], [7]], base=5))
    assert kv._table[5] == {4, 5} and kv._table[7] == {6} and idx.count() == 7
    assert hi.index == {0, 1, 2, 5, 7}
    with pytest.raises(KeyError):
        idx.remove_from_index([0, 42])
    assert idx.count() == 7
    idx.remove_from_index([2, 6])                            # bucket 2 shrinks, bucket 7 empties
    assert kv._table == {0: {0}, 1: {1}, 2: {3}, 5: {4, 5}}
    assert hi.index == {0, 1, 2, 5} and len(ds) == 5
    idx.build_index(_elems([[3]]))                           # build replaces everything
    assert kv._table == {3: {0}} and hi.index == {3} and idx.count() == 1
    ro = HipLSHNearestNeighborIndex(_BitsOfSum(), ds, kv, hi, "euclidean", read_only=True)
    for fn, arg in ((ro.build_index, d), (ro.update_index, d), (ro.remove_from_index, [0])):
        with pytest.raises(ReadOnlyError):
            fn(arg)
    with pytest.raises(ValueError):
        HipLSHNearestNeighborIndex(_BitsOfSum(), ds, kv, None, "hik")


def test_lsh_device_mirror_bucket_expansion():
    """CSR expansion of nearest-code ids into candidate rows (host side of the device re-rank)."""
    from smqtk_indexing_amd.impls.nn_index.hip_lsh import _DeviceMirror
    m = object.__new__(_DeviceMirror)
    # 4 codes; rows grouped by code: code0 -> rows [2, 5], code1 -> [], code2 -> [0], code3 -> [1, 3, 4]
    m.csr_rows = np.array([2, 5, 0, 1, 3, 4], dtype=np.int64)
    m.csr_off = np.array([0, 2, 2, 3, 6], dtype=np.int64)
    cand, off = m.expand(np.array([[3, 0], [1, -1], [2, 2]]))
    assert off.tolist() == [0, 5, 5, 7]
    assert cand.tolist() == [1, 3, 4, 2, 5, 0, 0]            # buckets in the given order, rows in row order
    cand, off = m.expand(np.zeros((2, 0), dtype=np.int64))
    assert cand.shape == (0,) and off.tolist() == [0, 0, 0]


def test_lsh_float_vectors_build_without_gpu_keeps_containers():
    """Without a device the mirror is skipped at build time; the reference containers are maintained."""
    if not NO_GPU:
        pytest.skip("needs a machine without a GPU")
    ds, kv = MemoryDescriptorSet(), MemoryKeyValueStore()
    idx = HipLSHNearestNeighborIndex(_BitsOfSum(), ds, kv, HipLinearHashIndex(), distance_method="euclidean")
    idx.build_index(_elems([[0.0], [1.0], [2.0]]))
    assert idx.count() == 3 and idx._mirror is None


def test_lsh_config_roundtrip():
    idx = HipLSHNearestNeighborIndex(HipItqFunctor(bit_length=4), MemoryDescriptorSet(), MemoryKeyValueStore(),
                                     HipLinearHashIndex(), "euclidean")
    c = idx.get_config()
    assert c["lsh_functor"]["type"].endswith("HipItqFunctor")
    if NO_GPU:
        # plugin discovery filters by is_usable(): without libsmqtk_hip + GPU the
        # HIP impls are not offered (smqtk_core Pluggable semantics)
        with pytest.raises(ValueError, match="not an available implementation"):
            HipLSHNearestNeighborIndex.from_config(c)
        return
    j = HipLSHNearestNeighborIndex.from_config(c)
    assert isinstance(j.hash_index, HipLinearHashIndex) and j.lsh_functor.bit_length == 4
    c["hash_index"] = None
    assert HipLSHNearestNeighborIndex.from_config(c).hash_index is None


def test_entry_points_name_importable_plugin_modules():
    """pyproject.toml registers the HIP implementations the way the reference registers its own
    (entry-point group "smqtk_plugins", name -> module: reference pyproject.toml:71-82); every listed module
    imports and holds a concrete subclass of one of the three interfaces, and the SMQTK_PLUGIN_PATH route
    (the other discovery mechanism of smqtk_core's Pluggable) finds the same classes."""
    import importlib
    import os
    import tomli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "pyproject.toml"), "rb") as f:
        cfg = tomli.load(f)
    eps = cfg["project"]["entry-points"]["smqtk_plugins"]
    assert len(eps) == 4
    found = {}
    for name, module in eps.items():
        assert name == module
        mod = importlib.import_module(module)
        impls = [c for c in vars(mod).values() if isinstance(c, type) and c.__module__ == module
                 and issubclass(c, (NearestNeighborsIndex, HashIndex, LshFunctor))]
        assert len(impls) == 1, (module, impls)
        found[module] = impls[0]
    assert set(found.values()) == {HipLinearHashIndex, HipItqFunctor, HipBruteForceNearestNeighborsIndex,
                                   HipLSHNearestNeighborIndex}
    old = os.environ.get("SMQTK_PLUGIN_PATH")
    os.environ["SMQTK_PLUGIN_PATH"] = ":".join(eps.values())
    try:
        # get_impls() lists USABLE implementations only (smqtk_core's Pluggable): with libsmqtk_hip.so and a GPU the
        # four classes are discovered; without, is_usable() is False (it must not raise) and they are filtered out
        usable = _lib.usable()
        assert all(c.is_usable() == usable for c in found.values())
        assert (HipLinearHashIndex in HashIndex.get_impls()) == usable
        assert (HipItqFunctor in LshFunctor.get_impls()) == usable
        assert ({HipBruteForceNearestNeighborsIndex, HipLSHNearestNeighborIndex} <= set(NearestNeighborsIndex.get_impls())) == usable
    finally:
        if old is None:
            del os.environ["SMQTK_PLUGIN_PATH"]
        else:
            os.environ["SMQTK_PLUGIN_PATH"] = old
    for pkg in cfg["tool"]["setuptools"]["packages"]:
        importlib.import_module(pkg)


def test_lsh_bucket_construction_and_refresh():
    """hash2uuids built one step per bucket (not per descriptor) equals the per-descriptor walk of lsh.py:316-323;
    codes wider than 64 bits take the plain walk.  refresh() / the store-version key drop the caches when the
    stores changed underneath the index."""
    rng = np.random.default_rng(3)
    keys = [int(x) for x in rng.integers(0, 50, size=400)] + [2 ** 64 - 1, 2 ** 63]
    uids = [f"u{i}" for i in range(len(keys))]
    want = {}
    for u, key in zip(uids, keys):
        want.setdefault(key, set()).add(u)
    assert HipLSHNearestNeighborIndex._buckets(uids, keys) == want
    wide = keys[:20] + [2 ** 70 + 3, 2 ** 70 + 3]
    got = HipLSHNearestNeighborIndex._buckets(uids[:22], wide)
    assert got[2 ** 70 + 3] == {"u20", "u21"} and sum(map(len, got.values())) == 22
    assert HipLSHNearestNeighborIndex._buckets([], []) == {}
    f = HipItqFunctor(bit_length=2)
    f.mean_vec, f.rotation = np.zeros(3), np.eye(3)[:, :2]
    idx = HipLSHNearestNeighborIndex(f, MemoryDescriptorSet(), MemoryKeyValueStore(), None, distance_method="euclidean")
    idx.hash2uuids_kvstore.add_many({1: {"a", "b"}, 2: {"c"}})
    idx.descriptor_set.add_many_descriptors([DescriptorMemoryElement(u).set_vector(np.ones(3)) for u in "abc"])
    assert idx.count() == 3
    idx.hash2uuids_kvstore.add_many({3: {"d"}})                # another writer: a new bucket -> the key changes
    assert idx.count() == 4
    idx.hash2uuids_kvstore.add_many({1: {"a", "b", "e"}})      # same number of buckets, same descriptor count:
    assert idx.count() == 4                                     # not visible to the cheap key ...
    idx.refresh()
    assert idx.count() == 5                                     # ... refresh() re-reads the stores


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` without a launcher starts its N ranks as child processes (torch.distributed.run,
    127.0.0.1 rendezvous on a free port), relays rank 0's one JSON line to stdout and returns the launcher's exit
    code.  --launch-selftest does no GPU work (gloo), so the launch itself is covered here; without it the ranks
    fail on a box without GPUs and the failure must come back as a non-zero exit code with no result line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-selftest"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    got = json.loads(lines[0])
    assert got == {"launch_selftest": True, "ranks": 2, "sum": 2}
    try:
        import torch
        has_gpu = torch.cuda.device_count() > 0
    except Exception:
        has_gpu = False
    if not has_gpu:
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root,
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode != 0
        assert not [ln for ln in p.stdout.decode().splitlines() if ln.strip().startswith("{")]

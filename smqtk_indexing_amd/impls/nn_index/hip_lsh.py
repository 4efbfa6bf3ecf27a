"""
LSH approximate ``NearestNeighborsIndex`` whose three hot loops run on MI355X.

Drop-in counterpart of ``LSHNearestNeighborIndex``
(smqtk_indexing/impls/nn_index/lsh.py:38-519): same constructor
(``lsh_functor``, ``descriptor_set``, ``hash2uuids_kvstore``, ``hash_index``,
``distance_method``, ``read_only``), same container semantics
(``hash2uuids``: python-int code -> set of UIDs; ``count()`` = sum of bucket
sizes, lsh.py:271-281; ``ReadOnlyError`` on mutation when read-only,
lsh.py:300-302; KeyError from unknown UIDs before anything is removed,
lsh.py:402-416).

What moved to the GPU:
  * hashing of descriptors in build / update / remove / query
    (lsh.py:316-321, 364-375, 412-416, 473) -> one batched ``sq_itq_hash`` call
    when the functor offers ``get_hash_packed`` (``HipItqFunctor``);
  * the nearest-code search (lsh.py:480-487) -> ``HipLinearHashIndex``
    (``sq_hamming_search``), also for the on-the-fly index the reference builds
    per query when ``hash_index`` is None;
  * the per-candidate distance calls (lsh.py:511) -> one ``sq_dense_distances``
    call in the reference's arithmetic.
Bucket expansion and the final stable sort + slice (lsh.py:489-519) stay on the
host: they are dictionary lookups over at most a few thousand candidates.
"""
import collections
import itertools
import threading
from typing import (Any, Deque, Dict, Hashable, Iterable, List, Optional, Set,
                    Tuple, Type, TypeVar)

import numpy as np

from ... import _lib
from ..._compat import (DescriptorElement, DescriptorSet, KeyValueStore,
                        ReadOnlyError, from_config_dict, make_default_config,
                        merge_dict, to_config_dict)
from ...interfaces.hash_index import HashIndex
from ...interfaces.lsh_functor import LshFunctor
from ...interfaces.nearest_neighbor_index import NearestNeighborsIndex
from ...utils.bits import (ints_to_packed, pack_bits_msb, packed_to_ints,
                           words_for_bits)
from ..hash_index.hip_linear import HipLinearHashIndex

T = TypeVar("T", bound="HipLSHNearestNeighborIndex")

_METRICS = {"euclidean": _lib.SQ_METRIC_L2, "cosine": _lib.SQ_METRIC_COSINE}


class HipLSHNearestNeighborIndex(NearestNeighborsIndex):
    """Hash -> nearest codes -> bucket expansion -> exact re-rank, on HIP kernels."""

    @classmethod
    def is_usable(cls) -> bool:
        return _lib.usable()

    @classmethod
    def get_default_config(cls) -> Dict[str, Any]:
        c = super().get_default_config()
        c["lsh_functor"] = make_default_config(LshFunctor.get_impls())
        c["descriptor_set"] = make_default_config(DescriptorSet.get_impls())
        c["hash2uuids_kvstore"] = make_default_config(KeyValueStore.get_impls())
        c["hash_index"] = make_default_config(HashIndex.get_impls())
        c["hash_index_comment"] = "'hash_index' may also be null to default to a linear index built at query time."
        return c

    @classmethod
    def from_config(cls: Type[T], config_dict: Dict, merge_default: bool = True) -> T:
        if merge_default:
            config_dict = merge_dict(cls.get_default_config(), config_dict)
        config_dict["lsh_functor"] = from_config_dict(config_dict["lsh_functor"], LshFunctor.get_impls())
        config_dict["descriptor_set"] = from_config_dict(config_dict["descriptor_set"], DescriptorSet.get_impls())
        config_dict["hash2uuids_kvstore"] = from_config_dict(config_dict["hash2uuids_kvstore"],
                                                             KeyValueStore.get_impls())
        hi = config_dict.get("hash_index")
        config_dict["hash_index"] = (from_config_dict(hi, HashIndex.get_impls())
                                     if hi and hi.get("type") else None)
        config_dict.pop("hash_index_comment", None)
        return super().from_config(config_dict, False)

    def __init__(self, lsh_functor: LshFunctor, descriptor_set: DescriptorSet,
                 hash2uuids_kvstore: KeyValueStore, hash_index: Optional[HashIndex] = None,
                 distance_method: str = "cosine", read_only: bool = False):
        super().__init__()
        self.lsh_functor = lsh_functor
        self.descriptor_set = descriptor_set
        self.hash_index = hash_index
        self.hash2uuids_kvstore = hash2uuids_kvstore
        self.distance_method = distance_method
        self.read_only = read_only
        self._model_lock = threading.RLock()
        if distance_method not in _METRICS:
            # 'hik' of the reference (lsh.py:250-251) has no device kernel
            raise ValueError("Invalid distance method label. Must be one of "
                             "['euclidean' | 'cosine'] for the HIP backend")
        self._metric = _METRICS[distance_method]

    def get_config(self) -> Dict[str, Any]:
        return {
            "lsh_functor": to_config_dict(self.lsh_functor),
            "descriptor_set": to_config_dict(self.descriptor_set),
            "hash_index": to_config_dict(self.hash_index) if self.hash_index is not None else None,
            "hash2uuids_kvstore": to_config_dict(self.hash2uuids_kvstore),
            "distance_method": self.distance_method,
            "read_only": self.read_only,
        }

    # ---------------------------------------------------------------- helpers
    def _hash_many(self, vectors: List[np.ndarray]) -> Tuple[np.ndarray, List[int]]:
        """bool hash vectors ``[n, bits]`` and their python-int keys (MSB first)."""
        if hasattr(self.lsh_functor, "get_hash_packed") and vectors:
            mat = np.asarray(vectors)
            if mat.ndim == 2:
                bits = self.lsh_functor.rotation.shape[1]      # type: ignore[attr-defined]
                packed = self.lsh_functor.get_hash_packed(mat)  # type: ignore[attr-defined]
                from ...utils.bits import unpack_bits_msb
                return unpack_bits_msb(packed, bits), packed_to_ints(packed)
        hv = np.vstack([np.asarray(self.lsh_functor.get_hash(v)).astype(bool) for v in vectors])
        return hv, packed_to_ints(pack_bits_msb(hv))

    def _guard(self) -> None:
        if self.read_only:
            raise ReadOnlyError("Cannot modify container attributes due "
                                "to being in read-only mode.")

    # -------------------------------------------------------------- interface
    def count(self) -> int:
        with self._model_lock:
            return sum(len(s) for s in self.hash2uuids_kvstore.values())

    def _build_index(self, descriptors: Iterable[DescriptorElement]) -> None:
        with self._model_lock:
            self._guard()
            self.descriptor_set.clear()
            self.descriptor_set.add_many_descriptors(descriptors)
            self.hash2uuids_kvstore.clear()
            elems = list(self.descriptor_set)
            hv, keys = self._hash_many([d.vector() for d in elems])
            update: Dict[Hashable, Set[Hashable]] = collections.defaultdict(set)
            for d, key in zip(elems, keys):
                update[key].add(d.uuid())
            self.hash2uuids_kvstore.add_many(update)
            if self.hash_index is not None:
                self.hash_index.build_index(hv)

    def _update_index(self, descriptors: Iterable[DescriptorElement]) -> None:
        with self._model_lock:
            self._guard()
            for_set, for_hash = itertools.tee(descriptors, 2)
            self.descriptor_set.add_many_descriptors(for_set)
            elems = list(for_hash)
            hv, keys = self._hash_many([d.vector() for d in elems])
            update: Dict[Hashable, Set[Hashable]] = {}
            for d, key in zip(elems, keys):
                if key not in update:
                    update[key] = self.hash2uuids_kvstore.get(key, set())
                update[key] |= {d.uuid()}
            self.hash2uuids_kvstore.add_many(update)
            if self.hash_index is not None:
                self.hash_index.update_index(hv)

    def _remove_from_index(self, uids: Iterable[Hashable]) -> None:
        with self._model_lock:
            self._guard()
            uids = list(uids)
            # KeyError here (unknown uid) leaves everything untouched
            elems = list(self.descriptor_set.get_many_descriptors(uids))
            hv, keys = self._hash_many([d.vector() for d in elems])
            update: Dict[Hashable, Set[Hashable]] = {}
            remove_keys: Set[Hashable] = set()
            gone: Deque[np.ndarray] = collections.deque()
            for uid, key, h in zip(uids, keys, hv):
                if key not in update:
                    update[key] = self.hash2uuids_kvstore.get(key, set())
                update[key] -= {uid}
                if not update[key]:
                    del update[key]
                    remove_keys.add(key)
                    gone.append(h)
            self.hash2uuids_kvstore.add_many(update)
            self.hash2uuids_kvstore.remove_many(remove_keys)
            if self.hash_index and gone:
                self.hash_index.remove_from_index(gone)
            self.descriptor_set.remove_many_descriptors(uids)

    def _nn(self, d: DescriptorElement, n: int = 1
            ) -> Tuple[Tuple[DescriptorElement, ...], Tuple[float, ...]]:
        d_v = np.asarray(d.vector())
        hv, _ = self._hash_many([d_v])
        d_h = hv[0]
        with self._model_lock:
            hi = self.hash_index
            if hi is None:
                # the reference builds a throw-away LinearHashIndex from the kvstore keys (lsh.py:481-486)
                hi = HipLinearHashIndex()
                keys = [int(k) for k in self.hash2uuids_kvstore.keys()]
                hi.set_codes_packed(ints_to_packed(keys, words_for_bits(len(d_h))))
            near_hashes, _ = hi.nn(d_h, n)
            neighbor_uuids: List[Hashable] = []
            for key in packed_to_ints(pack_bits_msb(np.asarray(near_hashes))):
                neighbor_uuids.extend(self.hash2uuids_kvstore.get(key, set()))
            neighbors = list(self.descriptor_set.get_many_descriptors(neighbor_uuids))
        vectors = np.asarray([e.vector() for e in neighbors])
        # exact re-rank: the reference's distance function per candidate row (lsh.py:511)
        dists = _lib.dense_distances(d_v, vectors, self._metric) if len(neighbors) else np.zeros(0)
        order = np.argsort(dists, kind="stable")[:n]
        r_descrs, r_dists = zip(*((neighbors[i], float(dists[i])) for i in order))
        return r_descrs, r_dists

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
With ``device_rerank`` (default) the index also keeps a device mirror of the
descriptor matrix (``sq_rows_*``) and a CSR map code -> rows, so a query's
bucket expansion is a vectorised gather of row ids and the re-rank (gather of
candidate rows, distances, stable top-n: lsh.py:499-519) is one device call that
returns only the n winners; ``nn_many`` answers a
batch of queries with one hash, one Hamming and one re-rank call.  The mirror is
rebuilt lazily after ``update_index`` / ``remove_from_index``.  Without a mirror
(mixed dtypes, ragged vectors, a foreign ``hash_index``) the host path below is
used: dictionary lookups + one ``sq_dense_distances`` call per query.
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


class _DeviceMirror:
    """Row-ordered device copy of the descriptors + CSR map from code id to rows."""

    def __init__(self, uuids: List[Hashable], matrix: np.ndarray, packed: np.ndarray):
        self.uuids = uuids
        self.dtype = matrix.dtype
        self.rows = _lib.RowMatrix(matrix)
        self.packed = np.ascontiguousarray(packed)               # [N, W] code of every row
        self.dead = np.zeros(len(uuids), dtype=bool)             # rows removed in place (tombstones: `remove`)
        self._row_of: Optional[Dict[Hashable, int]] = None       # uuid -> row, built at the first removal
        self._index_codes()

    def _index_codes(self) -> None:
        """Unique ascending codes and the CSR map code id -> rows, from the per-row codes of the live rows."""
        live_rows = np.flatnonzero(~self.dead)
        codes, inverse = np.unique(self.packed[live_rows], axis=0, return_inverse=True)
        inverse = np.asarray(inverse).reshape(-1)
        self.codes = np.ascontiguousarray(codes)                 # [C, W] ascending (= HipLinearHashIndex order)
        order = np.argsort(inverse, kind="stable")               # rows grouped by code id, row order inside a bucket
        self.csr_rows = live_rows[order].astype(np.int64)
        self.csr_off = np.searchsorted(inverse[order], np.arange(codes.shape[0] + 1)).astype(np.int64)
        self.rows.set_buckets(self.csr_off, self.csr_rows)       # the bucket map next to the rows (sq_lsh_query)
        self.own_index: Optional[HipLinearHashIndex] = None      # for hash_index=None
        self.checked_codes: Optional[np.ndarray] = None          # hash_index code array last compared with self.codes
        self.checked_ok = False

    def append(self, uuids: List[Hashable], matrix: np.ndarray, packed: np.ndarray) -> None:
        """New descriptors behind the resident ones: only they are uploaded (sq_rows_append); the code list
        and the CSR map are rebuilt on the host (code ids shift when a new code lands between old ones)."""
        self.rows.append(np.ascontiguousarray(matrix, dtype=self.dtype))
        if self._row_of is not None:
            self._row_of.update({u: len(self.uuids) + i for i, u in enumerate(uuids)})
        self.uuids.extend(uuids)
        self.packed = np.vstack([self.packed, np.ascontiguousarray(packed)])
        self.dead = np.concatenate([self.dead, np.zeros(len(uuids), dtype=bool)])
        self._index_codes()

    def remove(self, uuids: List[Hashable]) -> bool:
        """Descriptors leave the mirror in place: their rows stay in the resident matrix as tombstones that no bucket
        refers to any more (the bucket map is re-derived from the live rows' codes and re-uploaded; the descriptors
        are not).  False = too many tombstones (over a quarter) or an unknown uuid: the caller drops the mirror."""
        if self._row_of is None:
            self._row_of = {u: i for i, u in enumerate(self.uuids) if u is not None}
        uuids = list(dict.fromkeys(uuids))       # a uid listed twice leaves once (everything below mutates per entry)
        rows = [self._row_of.get(u) for u in uuids]
        if any(r is None for r in rows) or 4 * (int(self.dead.sum()) + len(rows)) > len(self.uuids):
            return False
        for u, r in zip(uuids, rows):
            del self._row_of[u]
            self.uuids[r] = None                                  # type: ignore[call-overload]
        self.dead[np.asarray(rows, dtype=np.int64)] = True
        if not (~self.dead).any():
            return False
        self._index_codes()
        return True

    def expand(self, code_ids: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """code ids ``[nq, m]`` (-1 = none) -> (candidate rows concatenated, offsets ``[nq+1]``),
        buckets in the given order, rows of a bucket in row order."""
        ids = np.asarray(code_ids, dtype=np.int64)
        valid = ids >= 0
        safe = np.where(valid, ids, 0)
        lens = np.where(valid, self.csr_off[safe + 1] - self.csr_off[safe], 0)
        starts = self.csr_off[safe]
        flat_len, flat_start = lens.reshape(-1), starts.reshape(-1)
        total = int(flat_len.sum())
        seg_end = np.cumsum(flat_len)
        # position inside its bucket for every candidate, then the CSR entry
        within = np.arange(total, dtype=np.int64) - np.repeat(seg_end - flat_len, flat_len)
        cand = self.csr_rows[np.repeat(flat_start, flat_len) + within]
        off = np.zeros(ids.shape[0] + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens.sum(axis=1))
        return cand, off

    def close(self) -> None:
        self.rows.close()

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
                 distance_method: str = "cosine", read_only: bool = False, device_rerank: bool = True):
        super().__init__()
        self.device_rerank = bool(device_rerank)
        self._mirror: Optional[_DeviceMirror] = None
        self._mirror_tried = False
        self._count_cache: Optional[Tuple[Tuple[int, int], int]] = None
        self._mirror_key: Optional[Tuple[int, int]] = None
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
            "device_rerank": self.device_rerank,
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

    @staticmethod
    def _buckets(uids: List[Hashable], keys: List[int]) -> Dict[int, Set[Hashable]]:
        """hash code -> set of uuids for descriptors (uids) with integer codes (keys): one python step per BUCKET
        (argsort + run boundaries), not per descriptor as lsh.py:316-323 walks them."""
        if not uids:
            return {}
        try:
            ka = np.asarray(keys, dtype=np.uint64)                    # codes up to 64 bits: sortable as machine words
        except OverflowError:
            ka = None
        if ka is None or ka.dtype != np.uint64:
            out: Dict[int, Set[Hashable]] = collections.defaultdict(set)
            for u, key in zip(uids, keys):
                out[key].add(u)
            return dict(out)
        order = np.argsort(ka, kind="stable")
        sk = ka[order]
        starts = np.flatnonzero(np.concatenate(([True], sk[1:] != sk[:-1])))
        ends = np.append(starts[1:], len(sk))
        ol = order.tolist()
        return {int(sk[a]): {uids[i] for i in ol[a:b]} for a, b in zip(starts.tolist(), ends.tolist())}

    # ---------------------------------------------------------- device mirror
    def _drop_mirror(self) -> None:
        if self._mirror is not None:
            self._mirror.close()
        self._mirror = None
        self._mirror_tried = False
        self._mirror_key = None

    def _state_key(self) -> Tuple[int, int]:
        """A cheap version of the two stores this index reads: (hash codes in the key-value store, descriptors in the
        set).  The caches below (bucket-size sum, device mirror) are only valid for the state they were built from;
        stores shared with another writer (a database-backed DescriptorSet / KeyValueStore) change underneath, and
        every change this key sees drops the caches.  A change it cannot see -- a uuid moving between existing
        buckets of a shared store with both counts unchanged -- needs an explicit :meth:`refresh`."""
        return len(self.hash2uuids_kvstore), int(self.descriptor_set.count())

    def refresh(self) -> None:
        """Forget everything cached from the stores (device mirror, bucket-size sum): the next query re-reads them.
        Call after the descriptor set or the key-value store was modified by someone other than this index."""
        with self._model_lock:
            self._drop_mirror()
            self._count_cache = None

    def _set_mirror(self, elems: List[DescriptorElement], vectors: List[np.ndarray], hv: np.ndarray) -> None:
        """(Re)build the device mirror from descriptors in row order, their vectors and bool codes."""
        self._drop_mirror()
        if not _lib.usable():
            return  # no device here: the containers are still maintained; a query will fail loudly in the kernels
        self._mirror_tried = True
        if not self.device_rerank or not elems:
            return
        try:
            mat = np.asarray(vectors)
        except ValueError:
            return
        if mat.ndim != 2 or mat.dtype not in (np.float32, np.float64):
            return
        self._mirror = _DeviceMirror([d.uuid() for d in elems], mat, pack_bits_msb(np.asarray(hv).astype(bool)))
        self._mirror_key = self._state_key()

    def _ensure_mirror(self) -> Optional[_DeviceMirror]:
        if self._mirror is not None and self._mirror_key != self._state_key():
            self._drop_mirror()            # the stores changed underneath (another writer): rebuild from them
        if self._mirror is None and not self._mirror_tried and self.device_rerank:
            elems = list(self.descriptor_set)
            if elems:
                vectors = [d.vector() for d in elems]
                hv, _ = self._hash_many(vectors)
                self._set_mirror(elems, vectors, hv)
            self._mirror_tried = True
        return self._mirror

    def _mirror_hash_index(self, m: _DeviceMirror) -> Optional[HipLinearHashIndex]:
        """The index whose row ids are the mirror's code ids, or None (host path)."""
        hi = self.hash_index
        if hi is None:
            if m.own_index is None:
                m.own_index = HipLinearHashIndex()
                m.own_index.set_codes_packed(m.codes)
            return m.own_index
        if isinstance(hi, HipLinearHashIndex):
            cp = hi.codes_packed()
            if m.checked_codes is cp:                      # the same array object as last time: already compared
                return hi if m.checked_ok else None
            m.checked_codes = cp
            m.checked_ok = bool(cp.shape == m.codes.shape and np.array_equal(cp, m.codes))
            return hi if m.checked_ok else None
        return None

    def _nn_device(self, vectors: np.ndarray, n: int
                   ) -> Optional[List[Tuple[Tuple[DescriptorElement, ...], Tuple[float, ...]]]]:
        """Batched query through the device mirror; None when the host path has to answer."""
        m = self._ensure_mirror()
        if m is None or vectors.dtype != m.dtype or vectors.ndim != 2 or vectors.shape[1] != m.rows.d:
            return None
        hi = self._mirror_hash_index(m)
        if hi is None:
            return None
        f = self.lsh_functor
        if (hasattr(f, "_device_model") and hasattr(f, "_norm_on_host") and not f._norm_on_host() and f.has_model()
                and words_for_bits(f.rotation.shape[1]) == m.codes.shape[1] and f.rotation.shape[0] == m.rows.d):
            # the whole query path in one device call (sq_lsh_query): hash -> nearest codes -> bucket expansion ->
            # exact re-rank; only the queries go up and the n winners per query come down (lsh.py:473-519)
            k = int(min(n, m.rows.n))
            dist, rows = m.rows.lsh_query(hi._device(), f._device_model(), vectors, n, self._metric, k)
            out = []
            for qi in range(vectors.shape[0]):
                good = rows[qi] >= 0
                uuids = [m.uuids[int(r)] for r in rows[qi][good]]
                descrs = tuple(self.descriptor_set.get_many_descriptors(uuids))
                out.append((descrs, tuple(float(x) for x in dist[qi][good])))
            return out
        if hasattr(self.lsh_functor, "get_hash_packed"):
            qp = self.lsh_functor.get_hash_packed(vectors)      # packed codes straight from the device  # type: ignore[attr-defined]
        else:
            hv, _ = self._hash_many(list(vectors))
            qp = pack_bits_msb(np.asarray(hv).astype(bool))
        w = m.codes.shape[1]
        if qp.shape[1] < w:
            qp = np.pad(qp, ((0, 0), (w - qp.shape[1], 0)))
        _, code_ids = hi.nn_packed(np.ascontiguousarray(qp), n)
        cand, off = m.expand(code_ids)
        k = int(min(n, max(1, int((off[1:] - off[:-1]).max()))))
        dist, pos = m.rows.rerank(vectors, self._metric, cand, off, k)
        out = []
        for qi in range(vectors.shape[0]):
            good = pos[qi] >= 0
            rows = cand[off[qi] + pos[qi][good]]
            uuids = [m.uuids[int(r)] for r in rows]
            descrs = tuple(self.descriptor_set.get_many_descriptors(uuids))
            out.append((descrs, tuple(float(x) for x in dist[qi][good])))
        return out

    def nn_many(self, descriptors: Iterable[DescriptorElement], n: int = 1
                ) -> List[Tuple[Tuple[DescriptorElement, ...], Tuple[float, ...]]]:
        """``nn`` for a batch of query descriptors: one hashing, one Hamming and one
        re-rank call for all of them (device mirror), else a loop over ``nn``."""
        descriptors = list(descriptors)
        vecs = [d.vector() for d in descriptors]
        if any(v is None for v in vecs):
            raise ValueError("Query descriptor did not have a vector set!")
        with self._model_lock:
            if not self.count():
                raise ValueError("No index currently set to query from!")
            try:
                mat = np.asarray(vecs)
            except ValueError:
                mat = None
            if mat is not None and mat.ndim == 2:
                res = self._nn_device(mat, n)
                if res is not None and all(len(r[0]) for r in res):
                    return res
        return [self.nn(d, n) for d in descriptors]

    def _guard(self) -> None:
        if self.read_only:
            raise ReadOnlyError("Cannot modify container attributes due "
                                "to being in read-only mode.")

    # -------------------------------------------------------------- interface
    def count(self) -> int:
        # sum of bucket sizes (lsh.py:271-281).  The reference walks every bucket on each call, and
        # NearestNeighborsIndex.nn calls count() per query; here the sum is cached and recomputed only
        # after this index mutated the store or the store's key count changed underneath it.
        with self._model_lock:
            key = self._state_key()
            if self._count_cache is None or self._count_cache[0] != key:
                self._count_cache = (key, sum(len(s) for s in self.hash2uuids_kvstore.values()))
            return self._count_cache[1]

    def _build_index(self, descriptors: Iterable[DescriptorElement]) -> None:
        with self._model_lock:
            self._guard()
            self.descriptor_set.clear()
            self.descriptor_set.add_many_descriptors(descriptors)
            self.hash2uuids_kvstore.clear()
            elems = list(self.descriptor_set)
            vectors = [d.vector() for d in elems]
            hv, keys = self._hash_many(vectors)
            self.hash2uuids_kvstore.add_many(self._buckets([d.uuid() for d in elems], keys))
            self._count_cache = None
            if self.hash_index is not None:
                self.hash_index.build_index(hv)
            self._set_mirror(elems, vectors, hv)

    def _update_index(self, descriptors: Iterable[DescriptorElement]) -> None:
        with self._model_lock:
            self._guard()
            elems = list(descriptors)
            # a pure append (no uuid replaced, no uuid twice) can extend the device mirror in place
            uids = [d.uuid() for d in elems]
            fresh = len(set(uids)) == len(uids) and not any(self.descriptor_set.has_descriptor(u) for u in uids)
            self.descriptor_set.add_many_descriptors(elems)
            vectors = [d.vector() for d in elems]
            hv, keys = self._hash_many(vectors)
            update: Dict[Hashable, Set[Hashable]] = self._buckets(uids, keys)
            for key in update:                                       # union with what the store holds (lsh.py:364-378)
                update[key] |= self.hash2uuids_kvstore.get(key, set())
            self.hash2uuids_kvstore.add_many(update)
            self._count_cache = None
            if self.hash_index is not None:
                self.hash_index.update_index(hv)
            m = self._mirror
            appended = False
            if m is not None and fresh and elems:
                try:
                    mat = np.asarray(vectors)
                    if mat.ndim == 2 and mat.dtype == m.dtype and mat.shape[1] == m.rows.d:
                        # only the new descriptors are uploaded (sq_rows_append); lsh.py:364-378 likewise touches
                        # only the new descriptors' buckets
                        m.append(uids, mat, pack_bits_msb(np.asarray(hv).astype(bool)))
                        self._mirror_key = self._state_key()
                        appended = True
                except Exception:      # whatever went wrong half-way: the mirror is rebuilt from the stores, never kept stale
                    appended = False
            if not appended:
                self._drop_mirror()  # rebuilt from the descriptor set at the next query

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
            self._count_cache = None
            if self.hash_index and gone:
                self.hash_index.remove_from_index(gone)
            self.descriptor_set.remove_many_descriptors(uids)
            m = self._mirror
            kept = False
            try:
                kept = m is not None and m.remove(uids)
            except Exception:          # the stores have changed already: a half-updated mirror must not answer queries
                kept = False
            if kept:
                self._mirror_key = self._state_key()   # the resident descriptors stay; only the bucket map moved
            else:
                self._drop_mirror()

    def _nn(self, d: DescriptorElement, n: int = 1
            ) -> Tuple[Tuple[DescriptorElement, ...], Tuple[float, ...]]:
        d_v = np.asarray(d.vector())
        with self._model_lock:
            if d_v.ndim == 1:
                res = self._nn_device(d_v.reshape(1, -1), n)
                if res is not None and len(res[0][0]):
                    return res[0]
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

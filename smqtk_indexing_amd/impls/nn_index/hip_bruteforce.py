"""
Exact brute-force ``NearestNeighborsIndex`` on MI355X.

The reference has no stand-alone brute-force class; the contract implemented
here is the one its FAISS ``"IDMap,Flat"`` wrapper expresses
(smqtk_indexing/impls/nn_index/faiss.py:486-559 build, 681-701 float32 matrix,
751-831 search ``k = min(n, ntotal)``, ascending true Euclidean distance) and
the exact tail of ``LSHNearestNeighborIndex._nn``
(impls/nn_index/lsh.py:505-519): distance of every row by
``metrics.euclidean_distance`` / ``cosine_distance``
(utils/metrics.py:73-86, 120-137), stable ascending sort, first ``n``.

The descriptor matrix lives in HBM as float32 (faiss.py:696-698 casts the same
way); ``nn`` calls ``sq_dense_search``.  Ties are returned in insertion order
(stable sort over rows), which is the canonical (distance, row) order.  When
the descriptors are not float32 (SMQTK's default is float64) the wrapper's
last step is kept as well: the distances of the n results are recomputed from
the ORIGINAL vectors against the float32 query (faiss.py:776, 818-824) and the
results are ordered by those.
"""
import threading
from typing import Any, Dict, Hashable, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .. import _require_usable
from ... import _lib
from ..._compat import DescriptorElement, ReadOnlyError
from ...interfaces.nearest_neighbor_index import NearestNeighborsIndex


class HipBruteForceNearestNeighborsIndex(NearestNeighborsIndex):
    """Exact L2 / cosine kNN over all indexed descriptors (HIP kernels)."""

    METRICS = {"euclidean": _lib.SQ_METRIC_L2, "cosine": _lib.SQ_METRIC_COSINE}

    @classmethod
    def is_usable(cls) -> bool:
        return _lib.usable()

    def __init__(self, distance_method: str = "euclidean", read_only: bool = False):
        super().__init__()
        if distance_method not in self.METRICS:
            raise ValueError("Invalid distance method label. Must be one of "
                             "['euclidean' | 'cosine']")
        self.distance_method = distance_method
        self.read_only = read_only
        self._lock = threading.RLock()
        self._elements: List[DescriptorElement] = []
        self._row_of: Dict[Hashable, int] = {}
        # host copy of the float32 rows, kept as the list of blocks it arrived in: an append adds a block (no O(N)
        # re-stacking per update); the blocks are only joined when the device copy has to be rebuilt
        self._blocks: List[np.ndarray] = []
        self._all_f32 = True     # every indexed vector is float32: the float32 search distances are final
        self._dev: Optional[_lib.DenseIndex] = None
        # rows removed while a device index is resident stay in it as tombstones (their `_elements` entry is None):
        # a search asks for k + len(_dead) rows and drops them; the index is rebuilt from the live rows once they pass
        # a quarter of it (what distributed.MutableShardedIndex does per shard)
        self._dead: set = set()

    def get_config(self) -> Dict[str, Any]:
        return {"distance_method": self.distance_method, "read_only": self.read_only}

    # ------------------------------------------------------------------ state
    @property
    def _matrix(self) -> np.ndarray:
        if not self._blocks:
            return np.zeros((0, 0), dtype=np.float32)
        if len(self._blocks) > 1:
            self._blocks = [np.ascontiguousarray(np.vstack(self._blocks), dtype=np.float32)]
        return self._blocks[0]

    def _dim(self) -> int:
        return int(self._blocks[0].shape[1]) if self._blocks else 0

    def _set(self, elements: List[DescriptorElement], matrix: np.ndarray) -> None:
        self._elements = elements
        self._dead = set()
        self._row_of = {e.uuid(): i for i, e in enumerate(elements)}
        self._blocks = [matrix] if matrix.shape[0] else []
        self._all_f32 = all(np.asarray(e.vector()).dtype == np.float32 for e in elements)
        if self._dev is not None:
            self._dev.close()
            self._dev = None

    def _device(self) -> _lib.DenseIndex:
        if self._dev is None:
            _require_usable(self)
            self._dev = _lib.DenseIndex(self._matrix, metric=self.METRICS[self.distance_method])
        return self._dev

    @staticmethod
    def _to_matrix(elements: Sequence[DescriptorElement]) -> np.ndarray:
        vecs = [e.vector() for e in elements]
        if any(v is None for v in vecs):
            raise ValueError("descriptor without a vector cannot be indexed")
        return np.ascontiguousarray(np.vstack(vecs), dtype=np.float32)

    def _guard(self) -> None:
        if self.read_only:
            raise ReadOnlyError("Cannot modify container attributes due to "
                                "being in read-only mode.")

    # -------------------------------------------------------------- interface
    def count(self) -> int:
        with self._lock:
            return len(self._elements) - len(self._dead)

    def _build_index(self, descriptors: Iterable[DescriptorElement]) -> None:
        with self._lock:
            self._guard()
            uniq: Dict[Hashable, DescriptorElement] = {}
            for d in descriptors:
                uniq[d.uuid()] = d
            elements = list(uniq.values())
            self._set(elements, self._to_matrix(elements))

    def _update_index(self, descriptors: Iterable[DescriptorElement]) -> None:
        with self._lock:
            self._guard()
            new: Dict[Hashable, DescriptorElement] = {}
            for d in descriptors:
                new[d.uuid()] = d
            add = list(new.values())
            add_m = self._to_matrix(add)
            if self._elements and self._dev is not None and add_m.shape[1] == self._dim() \
                    and not any(u in self._row_of for u in new):
                # nothing replaced: the new rows go behind the resident matrix (sq_dense_append, what
                # faiss.py:561-640 does with add_with_ids), no rebuild and no re-upload of the old rows
                self._dev.append(add_m)
                base = len(self._elements)
                self._elements.extend(add)
                self._row_of.update({e.uuid(): base + i for i, e in enumerate(add)})
                self._blocks.append(add_m)
                self._all_f32 = self._all_f32 and all(np.asarray(e.vector()).dtype == np.float32 for e in add)
                return
            kept = [e for e in self._elements if e is not None and e.uuid() not in new]
            keep_rows = [self._row_of[e.uuid()] for e in kept]
            if kept:
                matrix = np.vstack([self._matrix[keep_rows], add_m])
            else:
                matrix = add_m
            self._set(kept + add, np.ascontiguousarray(matrix, dtype=np.float32))

    def _remove_from_index(self, uids: Iterable[Hashable]) -> None:
        with self._lock:
            self._guard()
            uids = list(uids)
            for u in uids:
                if u not in self._row_of:
                    raise KeyError(u)          # nothing modified yet
            drop = set(uids)
            if self._dev is not None and 4 * (len(self._dead) + len(drop)) <= len(self._elements):
                # tombstones: the resident matrix is neither re-uploaded nor re-indexed
                for u in drop:
                    r = self._row_of.pop(u)
                    self._elements[r] = None       # type: ignore[call-overload]
                    self._dead.add(r)
                return
            kept = [e for e in self._elements if e is not None and e.uuid() not in drop]
            rows = [self._row_of[e.uuid()] for e in kept]
            self._set(kept, np.ascontiguousarray(self._matrix[rows], dtype=np.float32))

    def nn_many(self, vectors: np.ndarray, n: int = 1) -> Tuple[np.ndarray, np.ndarray]:
        """Batched search: ``[nq, d]`` -> (row ids ``[nq, k]``, distances ``[nq, k]``)."""
        with self._lock:
            if not self.count():
                raise ValueError("No index currently set to query from!")
            k = min(int(n), self.count())
            q = np.asarray(vectors, dtype=np.float32)
            if not self._dead:
                dist, idx = self._device().search(q, k)
                return idx, dist
            # tombstoned rows: ask for as many more, drop them (the (distance, row) order of the rest is unchanged)
            dist, idx = self._device().search(q, min(k + len(self._dead), len(self._elements)))
            dead = np.fromiter(self._dead, dtype=np.int64, count=len(self._dead))
            live = ~np.isin(idx, dead)
            pos = np.argsort(~live, axis=1, kind="stable")[:, :k]      # live entries first, in their order
            return np.take_along_axis(idx, pos, axis=1), np.take_along_axis(dist, pos, axis=1)

    def elements_of(self, rows: Sequence[int]) -> Tuple[DescriptorElement, ...]:
        with self._lock:
            return tuple(self._elements[int(r)] for r in rows)

    def _nn(self, d: DescriptorElement, n: int = 1
            ) -> Tuple[Tuple[DescriptorElement, ...], Tuple[float, ...]]:
        with self._lock:
            q32 = np.asarray(d.vector()).reshape(1, -1).astype(np.float32)      # faiss.py:776
            idx, dist = self.nn_many(q32, n)
            elems = self.elements_of(idx[0])
            if self._all_f32:
                return elems, tuple(float(v) for v in dist[0])
            # faiss.py:818-824: distances from the original-dtype vectors, results ordered by them
            rows = np.vstack([e.vector() for e in elems])
            exact = _lib.dense_distances(q32[0], rows, self.METRICS[self.distance_method])
            order = np.argsort(exact, kind="stable")
            return tuple(elems[i] for i in order), tuple(float(exact[i]) for i in order)
